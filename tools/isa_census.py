#!/usr/bin/env python
"""Instruction census of one kernel from the compiler's assembly (hipcc -S --cuda-device-only), by basic block:
vector / packed / scalar / LDS / vector-memory / MFMA / wait instructions.  What the phases of a frame loop cost
in issue slots before any counter is read.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Iinclude -Ishennong_amd/csrc -S \\
        --cuda-device-only -o /tmp/k.s shennong_amd/csrc/kernels_fbank2048.hip
    python tools/isa_census.py /tmp/k.s 'fbank2048_kernelILi9ELi1ELb0ELb1E' [min block size]
"""
import collections
import re
import sys

path, symbol = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 15
lines = open(path).read().split('\n')
start = next(i for i, line in enumerate(lines) if line.startswith('_Z') and symbol in line and line.rstrip().endswith(':') or
             (line.startswith('_Z') and symbol in line and ': ' in line))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, current = [], ('entry', [])
for line in lines[start + 1:end]:
    label = re.match(r'^(\.LBB\d+_\d+):', line)
    if label:
        blocks.append(current)
        current = (label.group(1) + (' (loop header)' if 'Loop Header' in line else ''), [])
        continue
    word = line.strip().split()
    if word and not word[0].startswith(('.', ';', '//')):
        current[1].append(word[0])
blocks.append(current)
total = collections.Counter()
print('%-28s %6s  %s' % ('block', 'instr', 'vector packed scalar lds vmem mfma wait'))
for name, ops in blocks:
    kinds = collections.Counter()
    for op in ops:
        kind = ('mfma' if op.startswith('v_mfma') else 'packed' if op.startswith('v_pk_') else
                'vector' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else
                'scalar' if op.startswith('s_') else 'lds' if op.startswith('ds_') else
                'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
        kinds[kind] += 1
    total.update(kinds)
    if len(ops) >= least:
        print('%-28s %6d  %s' % (name, len(ops), ' '.join('%d' % kinds[k] for k in
                                                        ('vector', 'packed', 'scalar', 'lds', 'vmem', 'mfma', 'wait'))))
print('%-28s %6d  %s' % ('whole function (static)', sum(total.values()), dict(total)))
