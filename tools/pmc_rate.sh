#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, kernel-trace only) for the kernel of one sample rate
# (tools/profile_rates.py <reps> <rate>).
# usage: tools/pmc_rate.sh <rate> <kernel name part> "<counters group 1>" ...   -> gpurun_out/pmc_rate_<rate>.txt
rate=$1; kern=$2; shift; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out
out=$root/gpurun_out/pmc_rate_$rate.txt
: > $out
i=0
for grp in "$@"; do
  i=$((i+1))
  d=/tmp/pmc_rate_${rate}_$i
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- \
     python $root/tools/profile_rates.py 3 $rate > /dev/null 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python - "$f" "$kern" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-42s %-28s %.4e  (n=%d)' % (k[0], k[1], sum(v) / len(v), len(v)))
PY
done
cat $out
