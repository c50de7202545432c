#!/bin/bash
# PMC passes over the A/B harness (tools/ab_fbank512.cpp): one rocprofv3 run per counter group (counters
# and kernel trace only), averaged per kernel name and launch configuration.
# usage: tools/pmc_ab.sh <tag> <kind> "<variant specs>" "<group 1>" "<group 2>" ...  -> gpurun_out/pmc_<tag>.txt
tag=$1; kind=$2; variants=$3; shift 3
export TMPDIR=/tmp
root=$(pwd)
export LD_LIBRARY_PATH=$root/shennong_amd:$LD_LIBRARY_PATH
mkdir -p gpurun_out
out=$root/gpurun_out/pmc_$tag.txt
: > $out
i=0
for grp in "$@"; do
  i=$((i+1))
  d=/tmp/pmc_${tag}_$i
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- \
     $root/scratch/ab512 10000 $kind 3 -- $variants > $d.log 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if 'fbank512' not in name: continue
    m = re.search(r'fbank512b_kernel<(.*?)>', name)
    key = ('b<%s>' % m.group(1).replace(' ', '')) if m else 'fbank512_kernel(old)'
    key += ' wg%s lds%s vgpr%s' % (r.get('Workgroup_Size', '?'), r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')))
    if key not in order: order.append(key)
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in order:
    print('##', key)
    for k, v in sorted(acc[key].items()):
        print('  %-34s %.4e  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
done
cat $out
