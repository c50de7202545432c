"""Dev helper (GPU box): what `process_all` on 10 000 pinned utterances is made of - the whole call, the plan call
alone, the bare uploads / downloads of the same bytes on 1-4 streams and both directions side by side.
   python tools/pa_phases.py"""
import os
import sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import Audio, Utterances, _backend
from shennong_amd.processor import FilterbankProcessor
import bench
n = 10000
waves = bench.make_batch(0, n, 48000)
index = Utterances([(f'u{i:05d}', Audio(waves[i], 16000, validate=False)) for i in range(n)]).pin()
corpus = index._pinned
proc = FilterbankProcessor(num_bins=40, dither=0)
plan = _backend.get_plan(proc._build_options())
L = _backend.lib()
def med(f, reps=7):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return '%.1f (min %.1f)' % (np.median(ts), min(ts))
print('process_all           ', med(lambda: proc.process_all(index)))
print('run_pinned (no wrap)  ', med(lambda: plan.run_pinned(corpus, None, check_finite=True)))
print('run_pinned no check   ', med(lambda: plan.run_pinned(corpus, None, check_finite=False)))
# raw copies
total = int(corpus.soff[-1]) * 2
d = _backend.DeviceBuffer(total)
st = [C.c_void_p() for _ in range(4)]
for s in st: _backend.check(L.snf_stream_create(C.byref(s)))
def h2d(pieces, nstreams):
    step = total // pieces
    for k in range(pieces):
        _backend.check(L.snf_memcpy_h2d_async(C.c_void_p(d.ptr + k * step), C.c_void_p(corpus.owner.address + k * step), step, st[k % nstreams]))
    for s in st: L.snf_stream_synchronize(s)
for pieces, ns in ((1, 1), (16, 1), (16, 4), (16, 2)):
    print('H2D 960 MB pieces %2d streams %d' % (pieces, ns), med(lambda: h2d(pieces, ns)))
out = _backend.result_array((2980000, 40), np.float32)
do = _backend.DeviceBuffer(out.nbytes)
def d2h(pieces, nstreams):
    step = out.nbytes // pieces
    for k in range(pieces):
        _backend.check(L.snf_memcpy_d2h_async(C.c_void_p(out.ctypes.data + k * step), C.c_void_p(do.ptr + k * step), step, st[k % nstreams]))
    for s in st: L.snf_stream_synchronize(s)
print('D2H 477 MB pieces 1 ', med(lambda: d2h(1, 1)))
def both():
    step = total // 16; so = out.nbytes // 16
    for k in range(16):
        _backend.check(L.snf_memcpy_h2d_async(C.c_void_p(d.ptr + k * step), C.c_void_p(corpus.owner.address + k * step), step, st[k % 2]))
        _backend.check(L.snf_memcpy_d2h_async(C.c_void_p(out.ctypes.data + k * so), C.c_void_p(do.ptr + k * so), so, st[2 + k % 2]))
    for s in st: L.snf_stream_synchronize(s)
print('H2D 960 + D2H 477 concurrently', med(both))
