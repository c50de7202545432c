#!/bin/bash
# Collects the artefacts that profiles/ holds for a round (run on the GPU box through gpurun):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic counters (separate --pmc passes)
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/profiles
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
cd /tmp
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
  python $root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extra > $out/bench_profiled.json 2> /dev/null
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $out/bench_fbank40_rocprofv3_kernel_stats.csv
rm -rf /tmp/prof_pitch
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pitch -- \
  python $root/tools/profile_pitch.py 4000 > $out/pitch_plp_run.txt 2> /dev/null
cp $(find /tmp/prof_pitch -name '*kernel_stats.csv' | head -1) $out/pitch_plp_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- \
    python $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > /dev/null 2>&1
  cp $(find /tmp/prof_$c -name '*counter_collection.csv' | head -1) $out/pmc_$c.csv
done
cd $root
python - <<'PY'
import csv, json, collections
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f'gpurun_out/profiles/pmc_{c}.csv')):
        if 'fbank512' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    res[c] = {k: sum(v) / len(v) for k, v in acc.items()}
print(json.dumps(res))
PY
tail -c 600 $out/bench_default.json
