"""Test infrastructure: a socket-backed stand-in for the ``snf_comm_*`` entry points of libshennong_hip.so
and for ``_backend.DeviceBuffer``, so that the REAL ``shennong_amd.comm.RcclComm`` - its TCP rendezvous, the
broadcast of the unique id, the object all-gather, ``gather_features`` with its counts and offsets, the
float64 all-reduce, the barrier - runs in two or more processes on a box without GPUs.

"Device" pointers are host addresses - or, with ``install(device=True)`` on a box that HAS a GPU, real device
pointers that the stand-in stages through host memory (snf_memcpy_*): bench.py's ``--transport stub`` runs 2 / 4 /
8 ranks as processes that share ONE GPU this way, so that everything around the exchange - sharding, counts,
offsets, double buffering, the JSON line - executes on hardware before a node with several GPUs is available.
Never a measurement: the blocks cross PCIe twice and a socket in between.  ``hang_after`` (tests): the n-th gather
of this process never returns - what a stalled peer looks like to the caller.  The exchange runs over TCP connections to rank 0 whose port travels
INSIDE the 128-byte unique id: a rank can only join if the id that rank 0 made reached it through
RcclComm's own rendezvous, as with ncclGetUniqueId / ncclCommInitRank.  The semantics restate
shennong_amd/csrc/comm.cpp (argument checks, the root's receive offsets = prefix sums of recv_counts in
rank order, the root's own block is a local copy, reductions applied in rank order); a count that the root
and a peer disagree on - which RCCL would answer with a hang - is an error here."""

import ctypes as C
import socket
import struct

import numpy as np

from shennong_amd import _abi

_MAGIC = b'FAKERCCL'


def _read(sock, n):
    chunks = []
    while n:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ConnectionError('peer closed the fake RCCL socket')
        chunks.append(chunk)
        n -= len(chunk)
    return b''.join(chunks)


def _value(x):
    """int address of a ctypes pointer-like argument (c_void_p, array, byref, int, None)"""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, C.c_void_p):
        return x.value or 0
    return C.cast(x, C.c_void_p).value or 0


class _Comm:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.peers = {}   # rank 0: rank -> socket; others: {0: socket}


class FakeCommLib:
    """Wraps the loaded library: everything except the communicator entry points goes to the real one"""
    def __init__(self, real, device=False, hang_after=None):
        self._real = real
        self._device = bool(device)
        self._hang_after = hang_after
        self._gathers = 0
        self._server = None
        self._comms = {}
        self._error = b''
        self.calls = []     # (name, details): what the communicator class asked for, for assertions

    # ---- where the blocks live ---------------------------------------------------------------------------
    def _fetch(self, ptr, nbytes, stream=None):
        """bytes of a block; device mode: behind everything enqueued on `stream` (the kernel that makes it)"""
        if nbytes <= 0:
            return b''
        if not self._device:
            return C.string_at(ptr, nbytes)
        if stream is not None and _value(stream):
            self._real.snf_stream_synchronize(C.c_void_p(_value(stream)))
        else:
            self._real.snf_device_synchronize()
        host = np.empty(nbytes, dtype=np.uint8)
        if self._real.snf_memcpy_d2h(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), nbytes) != 0:
            raise RuntimeError(self._real.snf_last_error().decode())
        return host.tobytes()

    def _store(self, ptr, data):
        if not data:
            return
        if not self._device:
            C.memmove(ptr, data, len(data))
            return
        host = np.frombuffer(data, dtype=np.uint8)
        if self._real.snf_memcpy_h2d(C.c_void_p(ptr), C.c_void_p(host.ctypes.data), len(data)) != 0:
            raise RuntimeError(self._real.snf_last_error().decode())

    def __getattr__(self, name):
        return getattr(self._real, name)

    def _fail(self, code, msg):
        self._error = msg.encode()
        return code

    def snf_last_error(self):
        return self._error or self._real.snf_last_error()

    def snf_device_synchronize(self):
        return self._real.snf_device_synchronize() if self._device else _abi.SNF_OK

    def snf_set_device(self, device_id):
        return self._real.snf_set_device(device_id) if self._device else _abi.SNF_OK

    # ---- streams and events without a device (host mode): every enqueue of the stand-in completes before it
    # returns, so a stream is always drained and an event has always happened ------------------------------------
    def _handle_out(self, out):
        self._handles = getattr(self, '_handles', 0) + 1
        out._obj.value = 0x57AE0000 + self._handles
        return _abi.SNF_OK

    def snf_stream_create(self, out):
        return self._real.snf_stream_create(out) if self._device else self._handle_out(out)

    def snf_event_create(self, out):
        return self._real.snf_event_create(out) if self._device else self._handle_out(out)

    def _noop(name, result=_abi.SNF_OK):   # noqa: N805 (a method factory, evaluated in the class body)
        def method(self, *args):
            return getattr(self._real, name)(*args) if self._device else result
        method.__name__ = name
        return method
    snf_stream_destroy = _noop('snf_stream_destroy')
    snf_stream_synchronize = _noop('snf_stream_synchronize')
    snf_stream_query = _noop('snf_stream_query', 0)
    snf_stream_wait_event = _noop('snf_stream_wait_event')
    snf_event_destroy = _noop('snf_event_destroy')
    snf_event_record = _noop('snf_event_record')
    del _noop

    def snf_event_elapsed_ms(self, start, stop, ms):
        if self._device:
            return self._real.snf_event_elapsed_ms(start, stop, ms)
        ms._obj.value = 0.0
        return _abi.SNF_OK

    # ---- the communicator ------------------------------------------------------------------------------
    def snf_comm_unique_id(self, ident):
        self._server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._server.bind(('127.0.0.1', 0))
        self._server.listen(64)
        blob = _MAGIC + struct.pack('<I', self._server.getsockname()[1])
        C.memmove(ident, blob + bytes(128 - len(blob)), 128)
        self.calls.append(('unique_id', self._server.getsockname()[1]))
        return _abi.SNF_OK

    def snf_comm_init(self, ident, world, rank, device, out):
        blob = bytes(C.string_at(_value(ident), 128))
        if blob[:8] != _MAGIC:
            return self._fail(_abi.SNF_E_RUNTIME, 'ncclCommInitRank: the unique id did not come from rank 0')
        if world < 1 or not 0 <= rank < world:
            return self._fail(_abi.SNF_E_INVALID, 'bad rank / world size')
        comm = _Comm(world, rank)
        (port,) = struct.unpack('<I', blob[8:12])
        if rank == 0:
            if world > 1 and (self._server is None or self._server.getsockname()[1] != port):
                return self._fail(_abi.SNF_E_RUNTIME, 'rank 0 lost its own unique id')
            while len(comm.peers) < world - 1:
                conn, _ = self._server.accept()
                conn.settimeout(60)
                (peer,) = struct.unpack('<I', _read(conn, 4))
                comm.peers[peer] = conn
            if self._server is not None:
                self._server.close()
                self._server = None
        else:
            conn = socket.create_connection(('127.0.0.1', port), timeout=60)
            conn.settimeout(60)
            conn.sendall(struct.pack('<I', rank))
            comm.peers[0] = conn
        handle = 0x5AFE0000 + len(self._comms) + 1
        self._comms[handle] = comm
        out._obj.value = handle
        self.calls.append(('init', world, rank, device))
        return _abi.SNF_OK

    def snf_comm_rank(self, handle):
        return self._comms[_value(handle)].rank

    def snf_comm_world_size(self, handle):
        return self._comms[_value(handle)].world

    def snf_comm_destroy(self, handle):
        comm = self._comms.pop(_value(handle), None)
        if comm is not None:
            for conn in comm.peers.values():
                conn.close()
        return _abi.SNF_OK

    def snf_comm_gatherv(self, handle, d_send, send_count, d_recv, recv_counts, root, stream):
        comm = self._comms.get(_value(handle))
        if comm is None:
            return self._fail(_abi.SNF_E_INVALID, 'null communicator')
        if send_count < 0 or not 0 <= root < comm.world:
            return self._fail(_abi.SNF_E_INVALID, 'bad argument')
        send, recv = _value(d_send), _value(d_recv)
        counts = None
        if comm.rank == root:
            if recv_counts is None or (not recv and send_count > 0):
                return self._fail(_abi.SNF_E_INVALID, 'null receive buffer')
            counts = [int(recv_counts[r]) for r in range(comm.world)]
            if counts[root] != send_count:
                return self._fail(_abi.SNF_E_INVALID, "recv_counts[root] differs from the root's send_count")
            if any(n < 0 for n in counts):
                return self._fail(_abi.SNF_E_INVALID, 'negative receive count')
        self._gathers += 1
        if self._hang_after is not None and self._gathers > self._hang_after:
            import threading
            threading.Event().wait()   # (a stalled transport: this call never comes back)
        mine = self._fetch(send, 4 * send_count, stream)
        # the sockets form a star around rank 0: blocks travel peer -> rank 0 (-> root)
        blocks = None
        if comm.rank == 0:
            blocks = {0: mine}
            for peer in range(1, comm.world):
                (n,) = struct.unpack('<q', _read(comm.peers[peer], 8))
                blocks[peer] = _read(comm.peers[peer], 4 * n) if n else b''
            if root != 0:
                for peer in range(comm.world):
                    if peer != root:
                        comm.peers[root].sendall(struct.pack('<q', len(blocks[peer]) // 4) + blocks[peer])
        else:
            comm.peers[0].sendall(struct.pack('<q', send_count) + mine)
            if comm.rank == root:
                blocks = {root: mine}
                for peer in range(comm.world):
                    if peer != root:
                        (n,) = struct.unpack('<q', _read(comm.peers[0], 8))
                        blocks[peer] = _read(comm.peers[0], 4 * n) if n else b''
        if comm.rank == root:
            offset = 0
            for peer in range(comm.world):
                n = counts[peer]
                if len(blocks[peer]) != 4 * n:   # (RCCL would hang or corrupt: make the mismatch loud)
                    return self._fail(_abi.SNF_E_RUNTIME, 'peer %d sends %d floats, root expects %d'
                                      % (peer, len(blocks[peer]) // 4, n))
                if n > 0 and not (peer == root and recv + 4 * offset == send):
                    self._store(recv + 4 * offset, blocks[peer])
                offset += n
            self.calls.append(('gatherv', 'root', counts))
        else:
            self.calls.append(('gatherv', 'send' if send_count > 0 else 'idle', int(send_count)))
        return _abi.SNF_OK

    def snf_comm_allreduce_f64(self, handle, d_buf, count, op, stream):
        comm = self._comms.get(_value(handle))
        if comm is None or (not _value(d_buf) and count > 0):
            return self._fail(_abi.SNF_E_INVALID, 'null pointer')
        if count < 0 or op not in (0, 1):
            return self._fail(_abi.SNF_E_INVALID, 'bad argument')
        self.calls.append(('allreduce', int(count), int(op)))
        if count == 0 or comm.world == 1:
            return _abi.SNF_OK
        buf = _value(d_buf)
        mine = np.frombuffer(self._fetch(buf, 8 * count, stream), dtype=np.float64).copy()
        if comm.rank == 0:
            total = mine
            for peer in range(1, comm.world):     # rank order: the result does not depend on arrival order
                head = struct.unpack('<qi', _read(comm.peers[peer], 12))
                if head != (count, op):
                    return self._fail(_abi.SNF_E_RUNTIME, 'ranks disagree on the all-reduce: %s vs %s'
                                      % (head, (count, op)))
                part = np.frombuffer(_read(comm.peers[peer], 8 * count), dtype=np.float64)
                total = total + part if op == 0 else np.maximum(total, part)
            for peer in range(1, comm.world):
                comm.peers[peer].sendall(total.tobytes())
        else:
            comm.peers[0].sendall(struct.pack('<qi', count, op) + mine.tobytes())
            total = np.frombuffer(_read(comm.peers[0], 8 * count), dtype=np.float64)
        self._store(buf, total.tobytes())
        return _abi.SNF_OK


class HostBuffer:
    """``_backend.DeviceBuffer`` on host memory (same surface: ptr, nbytes, upload, download, free)"""
    def __init__(self, nbytes, device=None):
        self.device = 0 if device is None else int(device)
        self.nbytes = int(nbytes)
        self._mem = np.zeros(max(self.nbytes, 16), dtype=np.uint8)
        self.ptr = self._mem.ctypes.data

    def upload(self, array):
        array = np.ascontiguousarray(array)
        C.memmove(self.ptr, array.ctypes.data, array.nbytes)

    def download(self, array):
        C.memmove(array.ctypes.data, self.ptr, array.nbytes)
        return array

    def free(self, synced=False):
        self.ptr = None


def install(device=False, hang_after=None):
    """Routes ``_backend.lib()`` through the stand-in in THIS process (a test worker, a rank of bench.py
    --transport stub) and, unless `device`, ``_backend.DeviceBuffer`` to host memory; returns the FakeCommLib (its
    ``calls`` record what the communicator asked for)"""
    from shennong_amd import _backend
    fake = FakeCommLib(_backend.lib(), device=device, hang_after=hang_after)
    _backend._LIB = fake
    if not device:
        _backend.DeviceBuffer = HostBuffer
    return fake
