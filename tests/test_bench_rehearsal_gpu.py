"""bench.py's N > 1 path on the ONE GPU of the test box: two ranks as processes that share
GPU 0, launched the way the driver launches the scaling runs (torch.distributed.run as a process spawner), the
exchange through the host-staged socket stand-in of snf_comm_* (`--transport stub`).  What runs for real: the
rendezvous, the socket barriers of the compute-only value, sharding under --scaling strong, the gather leg with its
two streams and two buffers, the digests of every rank's block, the assembly of the line - and, with a transport that
stalls, the watchdog: the line still appears, with the error in it, and both processes leave."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _launch(extra_args, extra_env=None, timeout=300):
    import importlib.util
    # (find_spec, not import: torch brings its own HIP runtime and RCCL into the process that imports it, and this
    # process has the system's loaded through libshennong_hip.so - a later ncclCommInitRank here then fails)
    if importlib.util.find_spec('torch') is None:
        pytest.skip('torch.distributed.run (the process spawner) is not installed')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '2', '--warmup', '1', '--inner', '3', '--settle', '2', '--utts', '400',
           '--transport', 'stub', '--no-extra', '--cpu-sample', '0'] + extra_args
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [x for x in done.stdout.splitlines() if x.startswith('{')]
    assert len(lines) == 1, done.stdout[-2000:]     # (rank 0 prints ONE line, rank 1 none)
    return json.loads(lines[0])


@pytest.mark.timeout(600)
@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_two_ranks_on_one_gpu(gpu, scaling):
    line = _launch(['--scaling', scaling])
    assert line['n_gpus'] == 2 and line['not_a_measurement'] is True and line['transport'].startswith('stub')
    assert line['scaling'] == scaling and line['value'] > 0 and line['value_mfcc13'] > 0
    frames = 400 * 298 * (2 if scaling == 'weak' else 1)
    assert line['config']['frames_per_pass_all_gpus'] == frames
    gathered = line['with_gather']
    assert 'error' not in gathered, gathered
    assert gathered['gathered_blocks_ok'] is True and gathered['gathered_blocks_checked'] == 2
    assert gathered['rccl_ranks_seen'] == 2
    assert gathered['gather_bytes_per_pass_at_root'] == (frames // 2) * 40 * 4
    assert line['value_with_gather'] == gathered['value'] > 0
    assert set(gathered['predicted_ms']) == {'153_GBps_per_link', '76.5_GBps_per_link'}


@pytest.mark.timeout(600)
def test_stalled_transport_costs_the_gather_not_the_line(gpu):
    line = _launch(['--comm-timeout', '4'], {'SNF_STUB_HANG_AFTER': '1'})
    assert line['value'] > 0 and line['value_mfcc13'] > 0 and line['roofline']['kernel_ms'] > 0
    assert line['value_with_gather'] is None
    assert 'did not finish within' in line['with_gather']['error']
    assert line['with_gather']['watchdog_abandoned_a_call'] is True
