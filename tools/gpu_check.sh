#!/bin/bash
# quick GPU check used during kernel iteration: parity tests, then the bench line
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300) > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest.log
(timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 "$@") > gpurun_out/bench.log 2>&1
tail -n 6 gpurun_out/pytest.log
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
    print('value %.4g frames/s  ms/step %.3f  kernel_ms %.3f  frac %.4f' % (
        d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
    for k, v in d['extra'].items():
        print(' ', k, v if not isinstance(v, dict) else {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print('bench parse failed', e)
    print(open('gpurun_out/bench.log').read()[-2000:])
PY
