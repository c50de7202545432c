"""Developer tool: window statistics of the Viterbi refinement levels of the pitch tracker, from the oracle's
per-frame NCCF rows (candidate evaluations per level, long windows, distinct backpointers per frame):
python tools/viterbi_window_stats.py synth|noise"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc
from shennong_amd import synth
from shennong_amd.processor import KaldiPitchProcessor
proc = KaldiPitchProcessor()
opts = proc._build_options()
po = opts.pitch
kind = sys.argv[1]
wave = synth.utterances(3, 1, 48000)[0]
if kind == 'noise':
    wave = np.random.default_rng(1).integers(-3000, 3000, size=48000).astype(np.int16)
out, down, res, pov, states = orc.pitch_debug(po, wave)
lags, first, last = orc.pitch_lags(po)
T, S = res.shape
factor = np.float32(po.penalty_factor) * np.float32(np.log(1.0 + po.delta_pitch)) ** 2
print('T', T, 'S', S, 'factor', factor)
fwd = np.zeros(S, np.float32)
idx = np.arange(S, dtype=np.float32)
tot = {'L3': 0, 'L4': 0, 'L4long': 0, 'L3long': 0, 'bisect': 0, 'pruned': 0}
nj = []
for t in range(T):
    local = (np.float32(1.0) - res[t]) + np.float32(po.soft_min_f0) * lags * res[t]
    d = idx[None, :] - idx[:, None]          # [i, j] = j - i
    cost = (d * d) * factor + fwd[None, :]
    bp = cost.argmin(axis=1)
    # window stats
    r8 = np.arange(0, S, 8); r32 = np.arange(0, S, 32)
    def win(i, gap):
        below = i & ~(gap - 1); above = below + gap
        return bp[below], (bp[above] if above < S else S - 1)
    l3 = [i for i in range(8, S, 8) if i % 32]
    for i in l3:
        lo, hi = win(i, 32); tot['L3'] += hi - lo + 1; tot['L3long'] += (hi - lo >= 64)
    for i in range(S):
        if i % 8:
            lo, hi = win(i, 8); tot['L4'] += hi - lo + 1; tot['L4long'] += (hi - lo >= 24)
    # pure bisection total
    h = 256
    while h >= 1:
        for i in range(h, S, 2 * h):
            lo = bp[i - h]; hi = bp[i + h] if i + h < S else S - 1
            tot['bisect'] += hi - lo + 1
        h //= 2
    nj.append(len(np.unique(bp)))
    newf = cost[np.arange(S), bp] + local
    fwd = (newf - newf.min()).astype(np.float32)
for k, v in tot.items(): print(k, v / T)
print('distinct backpointers per frame: mean %.1f max %d' % (np.mean(nj), np.max(nj)))
