// Multi-GPU exchange steps of the features path over RCCL: the variable-length gather
// of the per-rank feature blocks to a root, and the sum / max reductions the by-speaker CMVN statistics
// and the benchmark barrier need.  One process per GPU; device pointers in and out; no framework.
//
// The reference gathers the per-utterance results of its thread pool into one dict on the host
// (shennong/processor/base.py:104-107) and sums the CMVN statistics of a speaker's utterances
// (postprocessor/cmvn.py:145-164); these entry points are what those two steps become when the
// utterances are sharded over the 8 GPUs of a node.
//
// xGMI is point to point (7 links per GPU): the gather is ncclSend / ncclRecv pairs inside ONE group -
// every peer uses its own direct link to the root, nothing is relayed around a ring and the root
// receives exactly the bytes it needs (a ring all-gather would move 8x as much).
//
// RCCL is loaded at the first snf_comm_* call (dlopen "librccl.so.1"), so single-GPU users of
// libshennong_hip.so never map it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "snf_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;   // (optional: snf_comm_world_size falls back to the stored value)
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
#define SNF_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #sym))
    SNF_SYM(GetUniqueId, ncclGetUniqueId);
    SNF_SYM(CommInitRank, ncclCommInitRank);
    SNF_SYM(CommDestroy, ncclCommDestroy);
    SNF_SYM(GroupStart, ncclGroupStart);
    SNF_SYM(GroupEnd, ncclGroupEnd);
    SNF_SYM(Send, ncclSend);
    SNF_SYM(Recv, ncclRecv);
    SNF_SYM(AllReduce, ncclAllReduce);
    SNF_SYM(GetErrorString, ncclGetErrorString);
    SNF_SYM(CommCount, ncclCommCount);
#undef SNF_SYM
  });
  const bool ok = r.handle && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart &&
                  r.GroupEnd && r.Send && r.Recv && r.AllReduce && r.GetErrorString;
  return ok ? &r : nullptr;
}

int no_rccl() { return snf::set_error(SNF_E_RUNTIME, "RCCL (librccl.so.1) could not be loaded"); }

#define SNF_NCCL_CHECK(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess)                                                                          \
      return snf::set_error(SNF_E_RUNTIME, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
  } while (0)

}  // namespace

struct snf_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
};

extern "C" {

int snf_comm_unique_id(void* id128) {
  if (!id128) return snf::set_error(SNF_E_INVALID, "null pointer");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  static_assert(sizeof(ncclUniqueId) == SNF_COMM_ID_BYTES, "ncclUniqueId size");
  SNF_NCCL_CHECK(r->GetUniqueId(static_cast<ncclUniqueId*>(id128)));
  return SNF_OK;
}

int snf_comm_init(const void* id128, int32_t world_size, int32_t rank, int32_t device_id,
                  snf_comm** out) {
  if (!id128 || !out) return snf::set_error(SNF_E_INVALID, "null pointer");
  if (world_size < 1 || rank < 0 || rank >= world_size)
    return snf::set_error(SNF_E_INVALID, "bad rank / world size");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  SNF_HIP_CHECK(hipSetDevice(device_id));
  snf_comm* c = new snf_comm;
  c->world = world_size;
  c->rank = rank;
  c->device = device_id;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclResult_t rc = r->CommInitRank(&c->comm, world_size, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return snf::set_error(SNF_E_RUNTIME, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  }
  hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (he != hipSuccess) {
    (void)r->CommDestroy(c->comm);
    delete c;
    return snf::set_error(SNF_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(he));
  }
  *out = c;
  return SNF_OK;
}

int snf_comm_rank(const snf_comm* c) { return c ? c->rank : -1; }
// the number of ranks the RCCL communicator itself reports (ncclCommCount): what bench.py prints as
// `rccl_ranks_seen`, so that a line claiming N GPUs shows that N processes really met in one communicator
int snf_comm_world_size(const snf_comm* c) {
  if (!c) return -1;
  Rccl* r = rccl();
  int n = 0;
  if (r && r->CommCount && c->comm && r->CommCount(c->comm, &n) == ncclSuccess) return n;
  return c->world;
}

int snf_comm_destroy(snf_comm* c) {
  if (!c) return SNF_OK;
  Rccl* r = rccl();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (r && c->comm) (void)r->CommDestroy(c->comm);
  delete c;
  return SNF_OK;
}

// Every rank contributes `send_count` floats from its device buffer; on `root` they land in `d_recv` at
// the offsets prefix_sum(recv_counts) in rank order (recv_counts[world] is read on the root only; the
// root's own block is a device-to-device copy).  Stream-ordered on `stream` (NULL: the communicator's own
// stream, synchronised before returning).
int snf_comm_gatherv(snf_comm* c, const float* d_send, int64_t send_count, float* d_recv,
                     const int64_t* recv_counts, int32_t root, void* stream) {
  if (!c) return snf::set_error(SNF_E_INVALID, "null communicator");
  if (send_count < 0 || root < 0 || root >= c->world) return snf::set_error(SNF_E_INVALID, "bad argument");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  SNF_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
  if (c->rank == root) {
    if (!recv_counts || (!d_recv && send_count > 0)) return snf::set_error(SNF_E_INVALID, "null receive buffer");
    if (recv_counts[root] != send_count)
      return snf::set_error(SNF_E_INVALID, "recv_counts[root] differs from the root's send_count");
    for (int peer = 0; peer < c->world; ++peer)
      if (recv_counts[peer] < 0) return snf::set_error(SNF_E_INVALID, "negative receive count");
    // (no return between GroupStart and GroupEnd: a group left open would swallow every later call on this
    // thread; the first failure is kept and reported after the group has been closed)
    int64_t offset = 0;
    ncclResult_t failed = ncclSuccess;
    const char* where = "ncclRecv";
    SNF_NCCL_CHECK(r->GroupStart());
    for (int peer = 0; peer < c->world && failed == ncclSuccess; ++peer) {
      const int64_t n = recv_counts[peer];
      if (peer != root && n > 0)
        failed = r->Recv(d_recv + offset, static_cast<size_t>(n), ncclFloat32, peer, c->comm, s);
      offset += n;
    }
    const ncclResult_t closed = r->GroupEnd();
    if (failed == ncclSuccess && closed != ncclSuccess) {
      failed = closed;
      where = "ncclGroupEnd";
    }
    if (failed != ncclSuccess)
      return snf::set_error(SNF_E_RUNTIME, std::string(where) + " (gather, root): " + r->GetErrorString(failed));
    int64_t own = 0;
    for (int peer = 0; peer < root; ++peer) own += recv_counts[peer];
    if (send_count > 0 && d_recv + own != d_send)
      SNF_HIP_CHECK(hipMemcpyAsync(d_recv + own, d_send, sizeof(float) * send_count,
                                   hipMemcpyDeviceToDevice, s));
  } else if (send_count > 0) {
    SNF_NCCL_CHECK(r->GroupStart());
    ncclResult_t failed = r->Send(d_send, static_cast<size_t>(send_count), ncclFloat32, root, c->comm, s);
    const char* where = "ncclSend";
    const ncclResult_t closed = r->GroupEnd();
    if (failed == ncclSuccess && closed != ncclSuccess) {
      failed = closed;
      where = "ncclGroupEnd";
    }
    if (failed != ncclSuccess)
      return snf::set_error(SNF_E_RUNTIME, std::string(where) + " (gather, peer): " + r->GetErrorString(failed));
  }
  if (!stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
  return SNF_OK;
}

// In-place all-reduce of `count` float64 on the device: op 0 = sum (CMVN statistics of the speakers,
// [n_speakers, 2, dim + 1]), 1 = max (slowest rank of a timed region; doubles as a barrier).
int snf_comm_allreduce_f64(snf_comm* c, double* d_buf, int64_t count, int32_t op, void* stream) {
  if (!c || (!d_buf && count > 0)) return snf::set_error(SNF_E_INVALID, "null pointer");
  if (count < 0 || (op != 0 && op != 1)) return snf::set_error(SNF_E_INVALID, "bad argument");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  SNF_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
  if (count > 0)
    SNF_NCCL_CHECK(r->AllReduce(d_buf, d_buf, static_cast<size_t>(count), ncclFloat64,
                                op == 0 ? ncclSum : ncclMax, c->comm, s));
  if (!stream) SNF_HIP_CHECK(hipStreamSynchronize(s));
  return SNF_OK;
}

}  // extern "C"
