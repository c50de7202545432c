"""CPU model of the lane-per-candidate search: per level, every candidate j that lies strictly inside the window of
exactly one gap offers its cost to the states of that gap through a min over the 64-bit key (cost bits << 32 | j);
the two window ends are offered by the state itself.  Must reproduce the brute-force argmin (lowest index on ties)."""
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

def key(cost, j):
    return (cost.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64(j)

def search(fwd, S):
    INF = np.uint64(0xffffffffffffffff)
    slot = np.full(S + 64, INF, dtype=np.uint64)
    def cost(u, j):
        d = f32(j - u)
        return f32(f32(f32(d * d) * factor) + fwd[j])
    # level 1: exact for the multiples of 128
    for u in range(0, S, 128):
        for j in range(S):
            k = key(np.array(cost(u, j), f32), j)
            if k < slot[u]: slot[u] = k
    evals = 4 * S
    for st_known, st_new in ((128, 32), (32, 8), (8, 4), (4, 1)):
        known = list(range(0, S, st_known))
        bp = [int(slot[k] & np.uint64(0xffffffff)) for k in known]
        M = np.zeros(S + 1, dtype=np.int64)
        for b in bp: M[b] += 1
        c = np.cumsum(M)[:S]
        n_gaps = len(known) if known[-1] + st_new < S else len(known) - 1   # a last partial gap exists
        # recompute: gap g spans (known[g], known[g] + st_known); it exists if it holds an unknown state < S
        n_gaps = sum(1 for k in known if any(u < S for u in range(k + st_new, k + st_known, st_new)))
        m = st_known // st_new - 1
        # endpoints
        for g in range(n_gaps):
            lo = bp[g]; hi = bp[g + 1] if g + 1 < len(known) else S - 1
            for s in range(m):
                u = known[g] + (s + 1) * st_new
                if u >= S: continue
                a = key(np.array(cost(u, lo), f32), lo); b = key(np.array(cost(u, hi), f32), hi)
                slot[u] = min(a, b); evals += 2
        # interior candidates
        for j in range(S):
            g = int(c[j]) - 1
            if M[j] != 0 or g < 0 or g >= n_gaps: continue
            for s in range(m):
                u = known[g] + (s + 1) * st_new
                if u >= S: continue
                k = key(np.array(cost(u, j), f32), j); evals += 1
                if k < slot[u]: slot[u] = k
    bp = (slot[:S] & np.uint64(0xffffffff)).astype(np.int64)
    best = (slot[:S] >> np.uint64(32)).astype(np.uint32).view(f32)
    return bp, best, evals

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tot = 0; frames = 0
for u in range(n):
    wave = synth.utterances(u, 1, 24000)[0] if u else (np.random.default_rng(3).standard_normal(24000) * 300).astype(np.int16)
    out, down, res, pov, states = orc.pitch_debug(opts, wave)
    fwd = np.zeros(S, f32)
    for t in range(res.shape[0]):
        C = (TC + fwd[None, :]).astype(f32)
        bp_ref = C.argmin(axis=1); best_ref = C[jj, bp_ref]
        bp, best, ev = search(fwd, S)
        assert np.array_equal(bp, bp_ref), (u, t, np.nonzero(bp != bp_ref)[0][:5])
        assert np.array_equal(best.view(np.uint32), best_ref.view(np.uint32))
        tot += ev; frames += 1
        v = res[t]; local = (f32(1.0) - v).astype(f32); local = (local + (soft * v).astype(f32)).astype(f32)
        nf = (best_ref + local).astype(f32); fwd = (nf + f32(-nf.min())).astype(f32)
print('frames', frames, 'all equal to brute force; evaluations per frame %.0f' % (tot / frames))
