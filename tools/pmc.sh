#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, kernel-trace only) for the bench kernel.
# usage: tools/pmc.sh <tag> "<counters group 1>" "<group 2>" ...   -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out
out=$root/gpurun_out/pmc_$tag.txt
: > $out
i=0
for grp in "$@"; do
  i=$((i+1))
  d=/tmp/pmc_${tag}_$i
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- \
     python $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > /dev/null 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'fbank512' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-28s %.4e  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
done
cat $out
