"""Test infrastructure: opens one of the two CPU-side transports in a worker process.

``gloo``      - a torch.distributed gloo group behind the transport interface (torch_transport.py)
``rccl_stub`` - the product's own ``shennong_amd.comm.RcclComm.from_env()`` (TCP rendezvous, id broadcast,
                object all-gather, gatherv with counts and offsets, float64 all-reduce) over the
                socket-backed stand-in of the ``snf_comm_*`` entry points (fake_comm.py)
"""

import os
import sys

KINDS = ('gloo', 'rccl_stub')


def open_transport(kind, rank, world, port):
    """-> (transport, close()) in a freshly spawned worker process"""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if kind == 'gloo':
        import torch.distributed as dist
        from torch_transport import TorchTransport
        dist.init_process_group('gloo', rank=rank, world_size=world)
        transport = TorchTransport()

        def close():
            dist.barrier()
            dist.destroy_process_group()
        return transport, close
    assert kind == 'rccl_stub', kind
    import fake_comm
    from shennong_amd.comm import RcclComm
    fake = fake_comm.install()
    os.environ.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world),
                       'SNF_COMM_TOKEN': 'test-%d' % port})
    comm = RcclComm.from_env()
    comm.fake = fake

    def close():
        comm.barrier()
        comm.close()
    return comm, close
