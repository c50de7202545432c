"""Torch-free multi-GPU transport: RCCL through the C ABI (``snf_comm_*``, include/shennong_amd.h).

One process per GPU.  The data path (feature blocks, CMVN statistics) goes over RCCL / xGMI with
device pointers; what little host metadata the ranks need to agree on (names and shapes of the blocks,
the RCCL unique id) travels over TCP sockets to rank 0 - every frame authenticated with the job's token
(`_job_key`), size-capped, and parsed as plain data only (`_DataUnpickler`) - opened from the variables a
launcher such as ``torch.distributed.run`` exports (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``,
``MASTER_ADDR``, ``MASTER_PORT``) - the launcher is only a process spawner here, torch is never
imported.

An ``RcclComm`` can be passed wherever ``shennong_amd.distributed`` takes a ``group``.
"""

import ctypes as C
import hashlib
import hmac
import io
import os
import pickle
import socket
import struct
import time

import numpy as np

from shennong_amd import _backend

_PORT_OFFSET = 1017   # default rendezvous port = MASTER_PORT + this (MASTER_PORT itself belongs to the launcher)
_MAGIC = b'SNFC'
_HELLO = struct.Struct('<4sI')


def rendezvous_port(master_port):
    """Port of this module's own rendezvous: ``SNF_COMM_PORT`` when set, else next to the launcher's,
    inside the valid range"""
    if os.environ.get('SNF_COMM_PORT'):
        return int(os.environ['SNF_COMM_PORT'])
    port = master_port + _PORT_OFFSET
    return port if port <= 65535 else master_port - _PORT_OFFSET


def _job_key():
    """Key of the message authentication codes: every frame on the rendezvous sockets carries an
    HMAC-SHA256 over its length and payload, so only processes that hold the job's token are accepted
    as peers and nothing they did not send is ever parsed.  The token is ``SNF_COMM_TOKEN`` (export the
    same random string to every rank) or, failing that, the launcher's ``TORCHELASTIC_RUN_ID``; with
    neither the key is a constant and the socket must not be reachable by anyone else (the default
    address is the loopback interface)."""
    token = os.environ.get('SNF_COMM_TOKEN') or os.environ.get('TORCHELASTIC_RUN_ID', '')
    return hashlib.sha256(b'shennong_amd.comm/1:' + token.encode('utf-8')).digest()


def _max_message():
    return int(os.environ.get('SNF_COMM_MAX_MSG', str(4 << 30)))   # bytes; a bogus length never allocates more


def _send_msg(sock, payload, key):
    head = struct.pack('<Q', len(payload))
    sock.sendall(head + hmac.new(key, head + payload, hashlib.sha256).digest() + payload)


def _recv_msg(sock, key, limit=None):
    def read(n):
        chunks = []
        while n:
            chunk = sock.recv(min(n, 1 << 20))
            if not chunk:
                raise ConnectionError('peer closed the rendezvous socket')
            chunks.append(chunk)
            n -= len(chunk)
        return b''.join(chunks)
    head = read(8)
    (size,) = struct.unpack('<Q', head)
    if size > (limit if limit is not None else _max_message()):
        raise ConnectionError('rendezvous message of %d bytes exceeds the limit' % size)
    mac, payload = read(32), read(size)
    if not hmac.compare_digest(mac, hmac.new(key, head + payload, hashlib.sha256).digest()):
        raise ConnectionError('rendezvous message failed authentication (SNF_COMM_TOKEN differs?)')
    return payload


class _DataUnpickler(pickle.Unpickler):
    """Unpickles plain data only - builtin containers and scalars, numpy arrays and scalars - whatever the
    peer sent: the metadata the ranks exchange (names, shapes, times, properties) needs nothing else, and a
    pickle that names any other global is refused instead of executed."""
    _ALLOWED = {
        ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'range'),
        ('builtins', 'complex'), ('builtins', 'bytearray'), ('collections', 'OrderedDict'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'),
        ('numpy.core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
        ('numpy._core.multiarray', '_reconstruct'), ('numpy._core.multiarray', 'scalar'),
        ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED or (module == 'numpy.dtypes' and name.endswith('DType')):
            return super().find_class(module, name)
        raise pickle.UnpicklingError('rendezvous payload names %s.%s: only plain data is accepted'
                                     % (module, name))


def _loads(payload):
    return _DataUnpickler(io.BytesIO(payload)).load()


class RcclComm:
    """Communicator of `world_size` processes, one GPU each"""
    def __init__(self, rank, world_size, device=None, addr='127.0.0.1', port=29500, timeout=120.0,
                 connect=True):
        """`timeout` bounds every wait on the rendezvous sockets (a peer that died shows up as an error after
        that long, not as a hang).  With `connect` false only the sockets are opened - the host-side collectives
        (:func:`all_gather_object`, :func:`host_allreduce`, :func:`host_barrier`) work, the RCCL communicator is
        made later by :func:`connect` - so that a caller can bound the time RCCL's own bootstrap may take and
        still hold results that never needed it (bench.py)."""
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = _backend.get_device() if device is None else int(device)
        self._peers = {}     # rank 0: rank -> socket; others: {0: socket}
        self._key = _job_key()
        self._handle = C.c_void_p()
        lib = _backend.lib()
        ident = (C.c_char * 128)()
        if self.rank == 0:
            if connect:
                _backend.check(lib.snf_comm_unique_id(ident))
            if self.world_size > 1:
                server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                server.bind((addr, port))
                server.listen(self.world_size)
                server.settimeout(timeout)
                deadline = time.time() + timeout
                while len(self._peers) < self.world_size - 1:
                    if time.time() > deadline:
                        raise TimeoutError('rendezvous: %d of %d peers arrived'
                                           % (len(self._peers), self.world_size - 1))
                    conn, _ = server.accept()
                    conn.settimeout(timeout)
                    # a fixed-format, authenticated hello: anything else (a port scanner, a process of
                    # another job, a second claim on a rank) is dropped and does not take a slot
                    try:
                        magic, peer = _HELLO.unpack(_recv_msg(conn, self._key, limit=_HELLO.size))
                        if magic != _MAGIC or not 0 < peer < self.world_size or peer in self._peers:
                            raise ConnectionError('bad hello')
                    except (ConnectionError, struct.error, OSError):
                        conn.close()
                        continue
                    self._peers[peer] = conn
                server.close()
                if connect:
                    for conn in self._peers.values():
                        _send_msg(conn, bytes(ident), self._key)
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    conn = socket.create_connection((addr, port), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            conn.settimeout(timeout)
            _send_msg(conn, _HELLO.pack(_MAGIC, self.rank), self._key)
            if connect:
                C.memmove(ident, _recv_msg(conn, self._key, limit=128), 128)
            self._peers[0] = conn
        if connect:
            _backend.check(lib.snf_comm_init(ident, self.world_size, self.rank, self.device,
                                             C.byref(self._handle)))

    def connect(self):
        """Makes the RCCL communicator of a ``connect=False`` instance: rank 0 draws the unique id, the sockets
        carry it to the peers, every rank joins (ncclCommInitRank blocks until all have; a caller that must not
        wait for ever runs this on a thread it can abandon)"""
        if self._handle:
            return self
        lib = _backend.lib()
        ident = (C.c_char * 128)()
        if self.rank == 0:
            _backend.check(lib.snf_comm_unique_id(ident))
            for conn in self._peers.values():
                _send_msg(conn, bytes(ident), self._key)
        else:
            C.memmove(ident, _recv_msg(self._peers[0], self._key, limit=128), 128)
        _backend.check(lib.snf_comm_init(ident, self.world_size, self.rank, self.device,
                                         C.byref(self._handle)))
        return self

    @classmethod
    def from_env(cls, device=None, **kwargs):
        """Communicator described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT"""
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', str(rank)))
        return cls(rank, int(os.environ.get('WORLD_SIZE', '1')),
                   device=local if device is None else device,
                   addr=os.environ.get('MASTER_ADDR', '127.0.0.1'),
                   port=rendezvous_port(int(os.environ.get('MASTER_PORT', '29500'))), **kwargs)

    def close(self):
        if self._handle:
            _backend.lib().snf_comm_destroy(self._handle)
            self._handle = C.c_void_p()
        for conn in self._peers.values():
            conn.close()
        self._peers = {}

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: nocover
            pass

    # ---- host metadata (small python objects) over the rendezvous sockets ---------------------------
    def all_gather_object(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank"""
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [None] * (self.world_size - 1)
            for peer, conn in self._peers.items():
                objs[peer] = _loads(_recv_msg(conn, self._key))
            payload = pickle.dumps(objs)
            for conn in self._peers.values():
                _send_msg(conn, payload, self._key)
            return objs
        _send_msg(self._peers[0], pickle.dumps(obj), self._key)
        return _loads(_recv_msg(self._peers[0], self._key))

    def host_allreduce(self, array, op='sum'):
        """:func:`allreduce` over the rendezvous sockets instead of RCCL, applied in rank order: the barrier
        and the max-over-ranks of a timed region that must not depend on the transport it is about to measure
        (a stalled RCCL then costs the exchange's numbers, not the compute-only ones)"""
        host = np.ascontiguousarray(array, dtype=np.float64)
        parts = self.all_gather_object(host.tolist())
        total = np.asarray(parts[0], dtype=np.float64)
        for part in parts[1:]:
            part = np.asarray(part, dtype=np.float64)
            total = total + part if op == 'sum' else np.maximum(total, part)
        return total.reshape(host.shape)

    def host_barrier(self):
        self.all_gather_object(None)

    # ---- device data over RCCL -------------------------------------------------------------------------
    def gatherv_device(self, d_send, send_count, d_recv, recv_counts, root=0, stream=None):
        """Device pointers in and out (see snf_comm_gatherv); `recv_counts` is needed on the root.  With a
        `stream` (a handle of snf_stream_create) the exchange is enqueued on it, behind the kernels that
        produce `d_send`, and this returns without waiting; without one it is complete on return."""
        counts = None
        if self.rank == root:
            counts = np.ascontiguousarray(recv_counts, dtype=np.int64)
        _backend.check(_backend.lib().snf_comm_gatherv(
            self._handle, C.c_void_p(d_send), int(send_count), C.c_void_p(d_recv),
            counts.ctypes.data_as(C.POINTER(C.c_int64)) if counts is not None else None,
            int(root), C.c_void_p(stream) if stream else None))

    def ranks_seen(self):
        """World size as the RCCL communicator itself reports it (snf_comm_world_size)"""
        return int(_backend.lib().snf_comm_world_size(self._handle))

    def allreduce(self, array, op='sum'):
        """Element-wise sum / max of a float64 array over the ranks (through a device buffer)"""
        host = np.ascontiguousarray(array, dtype=np.float64)
        if host.size == 0:
            return host.copy()
        buf = _backend.DeviceBuffer(host.nbytes, device=self.device)
        try:
            buf.upload(host)
            _backend.check(_backend.lib().snf_comm_allreduce_f64(
                self._handle, C.c_void_p(buf.ptr), host.size, {'sum': 0, 'max': 1}[op], None))
            out = np.empty_like(host)
            buf.download(out)
        finally:
            buf.free()
        return out

    def barrier(self):
        self.allreduce(np.zeros(1), 'max')

    def gather_features(self, local, dst=0):
        """``{name: float32 [nframes, ndims]}`` of every rank merged on rank `dst` (None elsewhere):
        one contiguous block per peer, point to point to the root"""
        names = list(local.keys())
        shapes = [tuple(local[n].shape) for n in names]
        meta = self.all_gather_object((names, shapes))
        sizes = [sum(int(np.prod(s)) for s in m[1]) for m in meta]
        flat = np.concatenate([np.ascontiguousarray(local[n], dtype=np.float32).reshape(-1)
                               for n in names]) if names else np.zeros(0, np.float32)
        d_send = _backend.DeviceBuffer(max(flat.nbytes, 16), device=self.device)
        d_recv = None
        try:
            if flat.size:
                d_send.upload(flat)
            if self.rank == dst:
                d_recv = _backend.DeviceBuffer(max(4 * sum(sizes), 16), device=self.device)
            self.gatherv_device(d_send.ptr, flat.size, d_recv.ptr if d_recv else None, sizes, dst)
            if self.rank != dst:
                return None
            host = np.empty(sum(sizes), dtype=np.float32)
            if host.size:
                d_recv.download(host)
        finally:
            d_send.free()
            if d_recv is not None:
                d_recv.free()
        merged, pos = {}, 0
        for names_r, shapes_r in meta:
            for name, shape in zip(names_r, shapes_r):
                n = int(np.prod(shape))
                merged[name] = host[pos:pos + n].reshape(shape).copy()
                pos += n
        return merged
