"""Seeded synthetic 16 kHz utterances for benchmarks and parity tests:
``clip(round(3000 N(0,1) + 8000 sum_{h=1..5} sin(2 pi h f0 t) / h), +-32767)`` with a per-utterance
``f0 ~ U(80, 300) Hz``.  Every utterance is unique (the batch must not fit in the 256 MiB Infinity
Cache by repetition)."""

import numpy as np

_TABLE_SIZE = 8192
_TABLE = None


def _harmonic_table():
    global _TABLE
    if _TABLE is None:
        phase = np.arange(_TABLE_SIZE, dtype=np.float64) / _TABLE_SIZE
        _TABLE = sum(np.sin(2 * np.pi * h * phase) / h
                     for h in range(1, 6)).astype(np.float32) * 8000.0
    return _TABLE


def utterances(first_id, count, nsamples=48000, sample_rate=16000,
               seed=20260927):
    """Returns an int16 array [count, nsamples]; row i depends only on (seed, first_id + i)"""
    table = _harmonic_table()
    out = np.empty((count, nsamples), dtype=np.int16)
    t = np.arange(nsamples, dtype=np.float64)
    for i in range(count):
        rng = np.random.default_rng(seed + first_id + i)
        f0 = rng.uniform(80.0, 300.0)
        noise = rng.standard_normal(nsamples, dtype=np.float32) * 3000.0
        idx = ((t * (f0 / sample_rate)) % 1.0 * _TABLE_SIZE).astype(np.int64)
        x = np.rint(noise + table[idx])
        np.clip(x, -32767, 32767, out=x)
        out[i] = x.astype(np.int16)
    return out


def ragged_utterances(first_id, count, min_s=1.0, max_s=6.0,
                      sample_rate=16000, seed=20260927):
    """List of int16 arrays with seeded lengths in [min_s, max_s] seconds"""
    rng = np.random.default_rng(seed ^ 0x5EED)
    res = []
    for i in range(count):
        n = int(rng.integers(int(min_s * sample_rate),
                             int(max_s * sample_rate) + 1))
        res.append(utterances(first_id + i, 1, n, sample_rate, seed)[0])
    return res
