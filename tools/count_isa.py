#!/usr/bin/env python3
"""Tallies the instructions of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only) by
issue class: what the 512-point kernels' time is made of (profiles/NOTEBOOK.md 4.1b).

    python tools/count_isa.py listing.s <substring of the mangled kernel name> [...] [-v]
"""
import collections
import re
import sys

HALF = ('_dpp', '_sdwa', 'v_cndmask', 'v_dot2', 'v_cvt_', 'v_med3', 'v_mul_lo', 'v_lshl_add_u64',
        'v_mov_b64', 'v_readfirstlane', 'v_readlane', 'v_permlane', 'v_cmp_', 'v_mad_u64', 'v_mul_hi')
QUARTER = ('v_log_f32', 'v_exp_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32')


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('v_'):
        if any(op.startswith(q) for q in QUARTER):
            return 'valu_quarter'
        if any(h in op for h in HALF):
            return 'valu_half'
        return 'valu_full'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_nop'):
        return 's_nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def kernel_lines(path, needle):
    out, on = [], False
    for line in open(path):
        if not on and re.match(r'^_Z\w*:', line) and needle in line:
            on = True
            continue
        if on:
            if line.startswith('.Lfunc_end'):
                break
            out.append(line.rstrip('\n'))
    return out


def main():
    path = sys.argv[1]
    for needle in [a for a in sys.argv[2:] if a != '-v']:
        lines = kernel_lines(path, needle)
        # the main loop: the longest backward branch
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = i
        best = (0, 0)
        for i, l in enumerate(lines):
            m = re.match(r'^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                if i - labels[m.group(1)] > best[1] - best[0]:
                    best = (labels[m.group(1)], i)
        body = lines[best[0]:best[1] + 1]
        tally, ops = collections.Counter(), collections.Counter()
        for l in body:
            l = l.split(';')[0].strip()
            if not l or l.startswith('.') or l.endswith(':'):
                continue
            op = l.split()[0]
            tally[classify(op)] += 1
            ops[re.sub(r'_e(32|64)$', '', op)] += 1
        valu = tally['valu_full'] + tally['valu_half'] + tally['valu_quarter']
        slots = tally['valu_full'] + 2 * tally['valu_half'] + 4 * tally['valu_quarter']
        print('%s: main loop %d lines' % (needle, len(body)))
        print('  valu %d (full %d, half %d, quarter %d; %d full-rate slots)  mfma %d  lds %d  vmem %d  salu %d  '
              'waitcnt %d  s_nop %d' % (valu, tally['valu_full'], tally['valu_half'], tally['valu_quarter'], slots,
                                        tally['mfma'], tally['lds'], tally['vmem'], tally['salu'], tally['waitcnt'],
                                        tally['s_nop']))
        if '-v' in sys.argv:
            print('  ' + ', '.join('%s %d' % kv for kv in ops.most_common(40)))


if __name__ == '__main__':
    main()
