#!/bin/bash
# PMC passes for the pitch kernel (tools/profile_pitch.py N): usage tools/pmc_pitch.sh N "<group1>" ...
n=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out
out=$root/gpurun_out/pmc_pitch.txt
: > $out
i=0
for grp in "$@"; do
  i=$((i+1)); d=/tmp/pmcp_$i; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- \
     python $root/tools/profile_pitch.py $n > /dev/null 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'pitch_track' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-28s %.4e  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
done
cat $out
