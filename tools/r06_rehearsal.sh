#!/bin/bash
# Dress rehearsal of bench.py's N > 1 path on a ONE-GPU box: 2 / 4 / 8 ranks as processes
# that share GPU 0, snf_comm_* replaced by the host-staged socket stand-in (bench.py --transport stub).
# Lines land in gpurun_out/r06_stub_N<k>_<scaling>.json; never a measurement.
set -u
out=${1:-gpurun_out}
mkdir -p "$out"
port=29610
for scaling in weak strong; do
  for n in 2 4 8; do
    port=$((port + 7))
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus $n --steps 3 --warmup 1 --inner 8 --settle 5 --transport stub \
      --scaling $scaling --no-extra --cpu-sample 0 > "$out/r06_stub_N${n}_${scaling}.json" 2> "$out/r06_stub_N${n}_${scaling}.err"
    echo "N=$n $scaling rc=$? $(head -c 300 "$out/r06_stub_N${n}_${scaling}.json")"
  done
done
