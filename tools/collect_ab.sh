#!/bin/bash
# A/B of the round-2 (fbank512_kernel, SNF_FBANK512_OLD=1) and round-3 (fbank512b_kernel) forms of the
# 512-point kernel in ONE collection on ONE box (run through gpurun): rocprofv3 kernel statistics over the 20
# timed launches of `bench.py --no-extra` for each, and one --pmc pass each (kernel-trace only) with the four
# counters VERDICT r02 names.  Writes gpurun_out/profiles_ab/.
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/profiles_ab
mkdir -p $out
cd /tmp
for which in new old; do
  [ $which = old ] && export SNF_FBANK512_OLD=1 || unset SNF_FBANK512_OLD
  rm -rf /tmp/ab_stats_$which
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_stats_$which -- \
    python $root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extra > $out/bench_$which.json 2> /dev/null
  python $root/tools/timed_launch_stats.py $(find /tmp/ab_stats_$which -name '*kernel_trace.csv' | head -1) 20 \
    > $out/timed_launches_$which.csv
  rm -rf /tmp/ab_pmc_$which
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d /tmp/ab_pmc_$which -- \
    python $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > /dev/null 2>&1
  f=$(find /tmp/ab_pmc_$which -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pmc_$which.csv
done
unset SNF_FBANK512_OLD
cd $root
python - <<'PY' | tee gpurun_out/profiles_ab/summary.txt
import collections, csv, json
print('fbank-40, 10 000 x 3 s utterances, one box, one collection (tools/collect_ab.sh)')
for which, tag in (('old', 'fbank512_kernel<13, 1,'), ('new', 'fbank512b_kernel<13, 1,')):
    line = json.load(open('gpurun_out/profiles_ab/bench_%s.json' % which))
    rows = [r for r in csv.DictReader(open('gpurun_out/profiles_ab/timed_launches_%s.csv' % which)) if tag in r['Name']]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open('gpurun_out/profiles_ab/pmc_%s.csv' % which)):
        if tag in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    print('%-4s %-28s rocprofv3, 20 timed launches: %.4f ms (min %.4f)   HIP events: %.4f ms   value %.4ge9 frames/s' % (
        which, rows[0]['Name'].split('(')[0][10:], float(rows[0]['AverageNs']) / 1e6, float(rows[0]['MinNs']) / 1e6,
        line['roofline']['kernel_ms'], line['value'] / 1e9))
    print('     ' + '  '.join('%s %.4e' % (k, mean[k]) for k in sorted(mean)))
PY
