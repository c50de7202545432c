"""How many Viterbi candidates an EXACT filter would leave (round 5): state j can be dropped
when it lies above the chord of two other states by more than the float32 error bound of the costs (it then loses to
one of them for every state i), or when it loses to j - r / j + r for every i of the state range (slope test).  Run
on real forward-cost rows from the CPU oracle; asserts that no true argmin is ever dropped.
    python tools/pitch_hull_filter.py [utterances]"""
import sys, os, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import oracle as orc
from shennong_amd import _abi, synth
f32 = np.float32
opts = _abi.default_pitch_options()
lags, first, last = orc.pitch_lags(opts)
S = lags.shape[0]
factor = f32(f32(float(np.log(f32(1.005), dtype=f32)) ** 2) * f32(0.1))
jj = np.arange(S)
D2 = ((jj[None, :] - jj[:, None]) ** 2).astype(f32)
TC = (D2 * factor).astype(f32)
soft = (f32(opts.soft_min_f0) * lags).astype(f32)
F = float(factor)

def survivors(fwd, chord_r, slope_r, margin):
    alive = np.ones(S, bool)
    g = fwd.astype(np.float64)
    for r in chord_r:
        a = np.full(S, np.inf); b = np.full(S, np.inf)
        a[r:] = g[:-r]; b[:-r] = g[r:]
        alive &= ~(g - 0.5 * (a + b) - F * r * r > margin)
    for r in slope_r:
        a = np.full(S, np.inf); b = np.full(S, np.inf)
        a[r:] = g[:-r]; b[:-r] = g[r:]
        # beaten by j-r for every i <= S-1
        alive &= ~(g - a - F * r * (2.0 * (S - 1 - jj) + r) > margin)
        # beaten by j+r for every i >= 0
        alive &= ~(g - b - F * r * (2.0 * jj + r) > margin)
    return alive

stats = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for u in range(n):
    wave = synth.utterances(u, 1, 48000)[0] if u % 2 == 0 else synth.ragged_utterances(u, 1)[0]
    if u == n - 1:
        wave = (np.random.default_rng(5).standard_normal(48000) * 500).astype(np.int16)   # noise only
    out, down, res, pov, states = orc.pitch_debug(opts, wave)
    T = res.shape[0]
    fwd = np.zeros(S, f32)
    for t in range(T):
        C = (TC + fwd[None, :]).astype(f32)
        bp = C.argmin(axis=1)
        best = C[jj, bp]
        row = {'distinct': len(np.unique(bp))}
        margin = max(float(fwd.max()) + F * S * S, 1.0) * 2.0 ** -20
        for name, cr, sr in (('c1 s1', [1], [1]), ('c12 s12', [1, 2], [1, 2]), ('c124 s124', [1, 2, 4], [1, 2, 4]),
                             ('c1248 s1248', [1, 2, 4, 8], [1, 2, 4, 8]), ('c1-16 s1-16', [1, 2, 4, 8, 16], [1, 2, 4, 8, 16]),
                             ('s1248', [], [1, 2, 4, 8]), ('c1 s1248', [1], [1, 2, 4, 8]), ('c1 s1-32', [1], [1, 2, 4, 8, 16, 32])):
            al = survivors(fwd, cr, sr, margin)
            assert al[bp].all(), 'filter dropped an argmin!'
            row[name] = int(al.sum())
        stats.append(row)
        v = res[t]
        local = (f32(1.0) - v).astype(f32)
        local = (local + (soft * v).astype(f32)).astype(f32)
        nf = (best + local).astype(f32)
        fwd = (nf + f32(-nf.min())).astype(f32)
print('frames', len(stats))
for k in stats[0].keys():
    a = np.array([s[k] for s in stats])
    print('%-14s mean %.1f  p50 %d  p90 %d  max %d' % (k, a.mean(), np.median(a), np.percentile(a, 90), a.max()))
