#!/bin/bash
# Collects the artefacts that profiles/ holds for a round (run on the GPU box through gpurun):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic counters and SQ counters (one
#   --pmc group per run, kernel-trace only), pitch / PLP kernel stats.  Every step is bounded.
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/profiles
mkdir -p $out
timeout -s KILL 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
cd /tmp
rm -rf /tmp/prof_stats
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
  python $root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extra > $out/bench_profiled.json 2> /dev/null
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $out/bench_fbank40_rocprofv3_kernel_stats.csv
# the same trace restricted to the 20 timed launches of each kernel (settle and warm-up launches dropped)
# (20 steps x 120 passes per step = 2400 timed launches per leg since round 5)
python $root/tools/timed_launch_stats.py $(find /tmp/prof_stats -name '*kernel_trace.csv' | head -1) 2400 > $out/bench_fbank40_timed_launches_stats.csv
# the same bench as the driver launches it for N > 1 (torch.distributed.run as a process spawner, one rank):
# RcclComm.from_env(), the barrier / max all-reduce and the gather of the Features block over RCCL
timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29533 $root/bench.py --gpus 1 --steps 20 --warmup 3 --cpu-sample 0 --no-extra \
  > $out/bench_torchrun_1rank.json 2> $out/bench_torchrun_1rank.err
# parity log of the whole GPU suite (tests/conftest.py::assert_close) -> per-family worst errors and the
# fraction of elements inside the pure 1e-4 relative tolerance
rm -f /tmp/parity_log.txt
(cd $root && SNF_PARITY_LOG=/tmp/parity_log.txt timeout -s KILL 600 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1)
python $root/tools/parity_errors.py /tmp/parity_log.txt > $out/parity_errors.txt 2>&1
rm -rf /tmp/prof_pitch
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pitch -- \
  python $root/tools/profile_pitch.py 4000 > $out/pitch_plp_run.txt 2> /dev/null
cp $(find /tmp/prof_pitch -name '*kernel_stats.csv' | head -1) $out/pitch_plp_rocprofv3_kernel_stats.csv
rm -rf /tmp/prof_rates
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rates -- \
  python $root/tools/profile_rates.py 10 > $out/other_rates_run.txt 2> /dev/null
cp $(find /tmp/prof_rates -name '*kernel_stats.csv' | head -1) $out/other_rates_rocprofv3_kernel_stats.csv
# SQ counters of the pitch kernels (one --pmc group per run)
j=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  j=$((j+1)); rm -rf /tmp/prof_pmcp_$j
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/prof_pmcp_$j -- \
    python $root/tools/profile_pitch.py 4000 pitch-only > /dev/null 2>&1
  f=$(find /tmp/prof_pmcp_$j -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pmc_pitch_group_$j.csv
done
# SKIP_PMC=1: the counters of the mel kernels are kept from the last collection (kernels unchanged)
if [ -z "$SKIP_PMC" ]; then
# SQ counters of the long-frame and dual kernels (one --pmc group per run)
j=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  j=$((j+1)); rm -rf /tmp/prof_pmcr_$j
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/prof_pmcr_$j -- \
    python $root/tools/profile_rates.py 2 > /dev/null 2>&1
  f=$(find /tmp/prof_pmcr_$j -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pmc_rates_group_$j.csv
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/prof_pmc_$i
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/prof_pmc_$i -- \
    python $root/bench.py --steps 3 --warmup 1 --inner 1 --cpu-sample 0 --no-extra > /dev/null 2>&1
  f=$(find /tmp/prof_pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $out/pmc_group_$i.csv
done
fi
cd $root
python - <<'PY' | tee gpurun_out/profiles/pmc_pitch_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for name in sorted(glob.glob('gpurun_out/profiles/pmc_pitch_group_*.csv')):
    for r in csv.DictReader(open(name)):
        k = r['Kernel_Name']
        for tag in ('pitch_viterbi_kernel', 'pitch_nccf_kernel', 'pitch_resample_kernel'):
            if tag in k:
                acc[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
for (tag, c), v in sorted(acc.items()):
    print('%-24s %-24s %.5e per launch (n=%d)' % (tag, c, sum(v) / len(v), len(v)))
PY
[ -n "$SKIP_PMC" ] && { tail -c 400 $out/bench_default.json; exit 0; }
python - <<'PY' | tee gpurun_out/profiles/pmc_fbank512_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for name in sorted(glob.glob('gpurun_out/profiles/pmc_group_*.csv')):
    for r in csv.DictReader(open(name)):
        for tag in ('fbank512b_kernel<13, 1,', 'fbank512b_kernel<13, 2,'):   # fbank-40, MFCC-13
            if tag in r['Kernel_Name']:
                acc[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
for (tag, k), v in sorted(acc.items()):
    print('%-26s %-28s %.5e  (n=%d)' % (tag, k, sum(v) / len(v), len(v)))
PY
python - <<'PY' | tee gpurun_out/profiles/pmc_other_rates_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for name in sorted(glob.glob('gpurun_out/profiles/pmc_rates_group_*.csv')):
    for r in csv.DictReader(open(name)):
        k = r['Kernel_Name']
        for tag in ('fbank2048_kernel<9, 1', 'fbank1024x2_kernel<13, 1', 'fbank256x2_kernel<13, 1', 'delta_flat_o2w2_kernel<13'):
            if tag in k:
                acc[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
for (tag, c), v in sorted(acc.items()):
    print('%-28s %-24s %.5e  (n=%d)' % (tag, c, sum(v) / len(v), len(v)))
PY
tail -c 400 $out/bench_default.json
