"""Stage-by-stage comparison of the GPU pitch tracker with the oracle + timing"""
import os, sys, ctypes as C
import numpy as np, scipy.io.wavfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shennong_amd import _backend, _abi, synth, Audio
from shennong_amd.processor import KaldiPitchProcessor
from oracle import oracle as orc

def fetch(ptr, shape, dtype):
    a = np.empty(shape, dtype)
    _backend.check(_backend.lib().snf_memcpy_d2h(a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes))
    return a

def compare(name, wave, **opts):
    proc = KaldiPitchProcessor(**opts)
    po = proc._options
    plan = _backend.get_plan(proc._build_options())
    got = proc.process(Audio(wave, int(po.samp_freq))).data
    want, down, res, pov, states = orc.pitch_debug(po, wave)
    ptrs = [C.c_void_p() for _ in range(4)]
    _backend.check(_backend.lib().snf_debug_pitch_scratch(plan.handle, *[C.byref(p) for p in ptrs]))
    T, S = res.shape
    L = pov.shape[1]
    g_down = fetch(ptrs[0].value, down.shape, np.float32)
    g_res = fetch(ptrs[1].value, res.shape, np.float32)
    g_pov = fetch(ptrs[2].value, pov.shape, np.float32)
    g_states = fetch(ptrs[3].value, states.shape, np.int32)
    def rep(tag, a, b):
        bad = a != b
        print('  %-10s mismatches %7d / %-8d max abs %.3e' % (tag, bad.sum(), a.size, np.abs(a.astype(np.float64) - b).max() if a.size else 0))
        if bad.any():
            idx = np.argwhere(bad)[:3]
            for i in idx:
                print('      at', tuple(i), 'gpu', a[tuple(i)], 'oracle', b[tuple(i)])
    print(name, 'T', T, 'S', S, 'L', L)
    rep('down', g_down, down); rep('nccf_res', g_res, res); rep('pov', g_pov, pov); rep('states', g_states, states)
    rep('out pov', got[:, 0], want[:, 0]); rep('out f0', got[:, 1], want[:, 1])

wave = scipy.io.wavfile.read(os.path.join(ROOT, 'tests/golden/test.wav'))[1]
compare('test.wav', wave)
compare('test.wav shift 20', wave, frame_shift=0.02)
compare('test.wav 50 ms', wave, frame_shift=0.02, frame_length=0.05)
compare('test.wav f0 60-350', wave, min_f0=60, max_f0=350, penalty_factor=0.2)
for i, w in enumerate(synth.ragged_utterances(1000, 4, min_s=0.3, max_s=1.2)):
    compare('ragged %d' % i, w)
compare('3 s synth', synth.utterances(5, 1, 48000)[0])
compare('7 s synth (>500 frames)', synth.utterances(6, 1, 112000)[0])

# timing: 4000 x 3 s
n_utts, ns = 4000, 48000
waves = synth.utterances(0, n_utts, ns)
d_wave = _backend.DeviceBuffer(waves.nbytes); d_wave.upload(waves)
plan = _backend.get_plan(KaldiPitchProcessor()._build_options())
pf = plan.num_frames(ns)
soff = np.arange(n_utts + 1, dtype=np.int64) * ns
foff = np.arange(n_utts + 1, dtype=np.int64) * pf
d_out = _backend.DeviceBuffer(pf * n_utts * 2 * 4)
for _ in range(2): plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
ks = []
for _ in range(5):
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr); ks.append(plan.last_kernel_ms(0))
print('pitch 4000 x 3 s: %.3f ms (min %.3f) for %d frames' % (np.median(ks), np.min(ks), pf * n_utts))
got = np.empty((pf * 3, 2), np.float32); d_out.download(got)
want = np.concatenate([orc.pitch(_abi.default_pitch_options(), waves[i]) for i in range(3)])
print('batch first 3 utts equal:', np.array_equal(got, want), (got != want).sum())
