"""Replay of the Viterbi search of the pitch tracker on the CPU (round 5): for real forward-cost rows (the CPU
oracle gives the resampled NCCF, the exact argmin is brute force in float32), how many candidates every level of
csrc/kernels_pitch.hip evaluates, how many scan iterations its lane-per-state passes need and how many long
windows go to the 8-lane teams.  DESIGN.md 4.5 quotes its output.   python tools/pitch_search_replay.py [utterances]"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import oracle as orc
from shennong_amd import _abi, synth
f32 = np.float32
opts = _abi.default_pitch_options()
lags, first, last = orc.pitch_lags(opts)
S = lags.shape[0]
factor = f32(f32(float(np.log(f32(1.005), dtype=f32)) ** 2) * f32(0.1))
jj = np.arange(S)
TC = (((jj[None, :] - jj[:, None]) ** 2).astype(f32) * factor).astype(f32)
soft = (f32(opts.soft_min_f0) * lags).astype(f32)
acc = dict(frames=0, l3_iters=0, l3_long=0, l4_iters=0, l4_long=0, l5_iters=0, l5_long=0, l5_pass_iters=[],
           team_rounds=0, team_iters=0, l2_iters=0, cand_lane=0, cand_team=0, l5_sum_cand=0, l5_max_sum=0)
def level_states(level):
    if level == 3: return [i for i in range(8, S, 8) if i % 32], 32, 32
    if level == 4: return list(range(4, S, 8)), 8, 12
    return [i for i in range(S) if i % 4], 4, 12
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for u in range(n):
    wave = synth.utterances(u, 1, 48000)[0]
    out, down, res, pov, states = orc.pitch_debug(opts, wave)
    fwd = np.zeros(S, f32)
    for t in range(res.shape[0]):
        C = (TC + fwd[None, :]).astype(f32)
        bp = C.argmin(axis=1)
        acc['frames'] += 1
        # level 2: states multiple of 32 not of 128: window between level-1 neighbours (128 apart); 4 lanes x 8 per step = 32 per iteration
        for i in range(32, S, 32):
            if i % 128 == 0: continue
            lo = bp[i & ~127]; hi = bp[min((i & ~127) + 128, S - 1)] if (i & ~127) + 128 < S else S - 1
        l2 = [(bp[i & ~127], (bp[(i & ~127) + 128] if (i & ~127) + 128 < S else S - 1)) for i in range(32, S, 32) if i % 128]
        acc['l2_iters'] += max(-(-(hi - lo + 1) // 32) for lo, hi in l2)
        for level in (3, 4, 5):
            st, gap, long_range = level_states(level)
            ws = []
            for i in st:
                below = i & ~(gap - 1); above = below + gap
                lo = bp[below]; hi = bp[above] if above < S else S - 1
                ws.append(hi - lo)
            ws = np.array(ws)
            longs = ws >= long_range
            key = 'l%d' % level
            acc[key + '_long'] += int(longs.sum())
            for p0 in range(0, len(ws), 64):
                w = ws[p0:p0 + 64]; lg = longs[p0:p0 + 64]
                it = int(max([0] + [-(-x // 4) for x in w[~lg]]))
                acc[key + '_iters'] += it
                acc['cand_lane'] += int((w[~lg] + 1).sum())
                if level == 5:
                    acc['l5_pass_iters'].append(it)
            lw = ws[longs]
            for r0 in range(0, len(lw), 8):
                grp = lw[r0:r0 + 8]
                acc['team_rounds'] += 1
                acc['team_iters'] += int(max(-(-(x + 1) // 32) for x in grp))
                acc['cand_team'] += int((grp + 1).sum())
        best = C[jj, bp]
        v = res[t]
        local = (f32(1.0) - v).astype(f32); local = (local + (soft * v).astype(f32)).astype(f32)
        nf = (best + local).astype(f32); fwd = (nf + f32(-nf.min())).astype(f32)
F = acc['frames']
print('frames', F)
for k in ('l2_iters', 'l3_iters', 'l3_long', 'l4_iters', 'l4_long', 'l5_iters', 'l5_long', 'team_rounds', 'team_iters', 'cand_lane', 'cand_team'):
    print('%-12s %.2f per frame' % (k, acc[k] / F))
pi = np.array(acc['l5_pass_iters']); print('l5 pass iterations histogram', np.bincount(pi) / len(pi))
