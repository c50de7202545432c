"""The N > 1 path on CPU: world_size-2 gloo processes exercise sharding and the Features gather
(the GPU kernels are not involved; rank-local 'features' are synthetic matrices)."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from shennong_amd.distributed import shard_utterances


def test_shard_utterances_balanced():
    rng = np.random.default_rng(0)
    lengths = rng.integers(16000, 96000, size=1001)
    for world in (1, 2, 4, 8):
        shards = shard_utterances(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(1001))
        totals = [int(lengths[s].sum()) for s in shards]
        assert max(totals) - min(totals) <= lengths.max()
    assert shard_utterances([], 2) == [[], []]
    assert shard_utterances([5, 5, 5], 4) == [[0], [1], [2], []]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from shennong_amd.distributed import gather_features, shard_utterances
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    nframes = rng.integers(0, 50, size=11)
    mats = {f'utt{i}': np.random.default_rng(100 + i).standard_normal(
        (int(n), 5)).astype(np.float32) for i, n in enumerate(nframes)}
    shards = shard_utterances(nframes, world)
    local = {f'utt{i}': mats[f'utt{i}'] for i in shards[rank]}
    merged = gather_features(local, dst=0)
    ok = True
    if rank == 0:
        ok = sorted(merged) == sorted(mats) and all(
            np.array_equal(merged[k], mats[k]) for k in mats)
    else:
        ok = merged is None
    # a rank with nothing to send must not deadlock the gather
    merged = gather_features(local if rank == 0 else {}, dst=0)
    if rank == 0:
        ok = ok and sorted(merged) == sorted(local)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, f'ok{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.timeout(120)
def test_gather_features_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f'ok{r}').read() for r in range(2)] == ['1', '1']
