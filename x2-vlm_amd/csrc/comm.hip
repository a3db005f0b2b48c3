// RCCL communicator behind the C ABI (x2_comm_{init,allreduce_bucket,allgather,broadcast,destroy}).
// Replaces, for a host that is not PyTorch, what accelerators/apex_ddp_accelerator.py:57-97 (NCCL init, per-tensor
// broadcast, apex DDP's flat all-reduce + average) and models/xvlm.py:140-160 (ITC all_gather) do in the reference.
// One communicator per process (one process per GPU); every call is enqueued on the stream handed in - the caller's
// communication side stream - and optionally records a HIP event behind it, so gradient buckets overlap with the
// backward running on the compute stream.  xGMI is point-to-point: a ring all-reduce is bound by one link
// (~153 GB/s), so callers send few large buckets (a layer's flat gradient arena: 28-50 MB), never per-tensor messages.
//
// librccl.so.1 is resolved at run time (dlopen) instead of link time: the kernels of this library load and work on a
// box without RCCL, and inside a PyTorch process the already-loaded RCCL (same soname) is shared, never duplicated.
#include "x2_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
// the slice of the RCCL 2.x API used here (rccl/rccl.h: ncclResult_t == int, ncclComm_t == opaque pointer,
// ncclUniqueId == 128 bytes, ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0, ncclAvg = 4)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*CommDestroy_t)(void*);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*Broadcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*GetErrorString_t)(int);
typedef int (*CommCount_t)(const void*, int*);

struct Rccl {
  void* handle = nullptr;
  GetUniqueId_t get_unique_id = nullptr;
  CommInitRank_t comm_init_rank = nullptr;
  CommDestroy_t comm_destroy = nullptr;
  AllReduce_t all_reduce = nullptr;
  AllGather_t all_gather = nullptr;
  Broadcast_t broadcast = nullptr;
  GetErrorString_t error_string = nullptr;
  CommCount_t comm_count = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return X2_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  X2_REQUIRE(h, "x2_comm: cannot load librccl.so.1: %s", dlerror());
#define X2_SYM(field, type, name) g_rccl.field = (type)dlsym(h, name); X2_REQUIRE(g_rccl.field, "x2_comm: %s missing in RCCL", name)
  X2_SYM(get_unique_id, GetUniqueId_t, "ncclGetUniqueId");
  X2_SYM(comm_init_rank, CommInitRank_t, "ncclCommInitRank");
  X2_SYM(comm_destroy, CommDestroy_t, "ncclCommDestroy");
  X2_SYM(all_reduce, AllReduce_t, "ncclAllReduce");
  X2_SYM(all_gather, AllGather_t, "ncclAllGather");
  X2_SYM(broadcast, Broadcast_t, "ncclBroadcast");
  X2_SYM(error_string, GetErrorString_t, "ncclGetErrorString");
  X2_SYM(comm_count, CommCount_t, "ncclCommCount");
#undef X2_SYM
  g_rccl.handle = h;
  return X2_OK;
}

struct X2Comm { void* nccl; int rank; int world; };

int nccl_dtype(int dtype) { return dtype == 0 ? 7 : dtype == 1 ? 9 : -1; }     // 0: fp32, 1: bf16

#define X2_NCCL(call, what) do { const int r_ = (call); if (r_ != 0) { x2_set_error("%s: RCCL error %d: %s", what, r_, g_rccl.error_string(r_)); return X2_ERR_LAUNCH; } } while (0)

int record(void* done_event, void* stream, const char* what) {
  if (!done_event) return X2_OK;
  const hipError_t e = hipEventRecord((hipEvent_t)done_event, (hipStream_t)stream);
  if (e != hipSuccess) { x2_set_error("%s: hipEventRecord: %s", what, hipGetErrorString(e)); return X2_ERR_LAUNCH; }
  return X2_OK;
}
}  // namespace

// Rank 0 creates the 128-byte id and shares it out of band (the reference shares its rendezvous through
// MASTER_ADDR/MASTER_PORT, Pretrain.py:563-570; the Python host below uses a torch.distributed store or a file).
extern "C" int x2_comm_unique_id(void* out128) {
  X2_REQUIRE(out128, "x2_comm_unique_id: null output");
  if (load_rccl() != X2_OK) return X2_ERR_ARG;
  UniqueId id;
  X2_NCCL(g_rccl.get_unique_id(&id), "x2_comm_unique_id");
  memcpy(out128, &id, sizeof(id));
  return X2_OK;
}

// Collective over all `world` processes; the calling thread's current HIP device is the rank's GPU.
extern "C" int x2_comm_init(const void* id128, int rank, int world, void** comm_out) {
  X2_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "x2_comm_init: rank %d / world %d", rank, world);
  if (load_rccl() != X2_OK) return X2_ERR_ARG;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* nccl = nullptr;
  X2_NCCL(g_rccl.comm_init_rank(&nccl, world, id, rank), "x2_comm_init");
  *comm_out = new X2Comm{nccl, rank, world};
  return X2_OK;
}

// In-place all-reduce of one flat gradient bucket; average != 0 divides by the world size inside RCCL (ncclAvg), which
// is what apex DDP's allreduce + `/ world_size` leaves in .grad (apex_ddp_accelerator.py:87-97).  dtype: 0 fp32, 1 bf16.
// done_event: optional hipEvent_t recorded on `stream` behind the collective.
extern "C" int x2_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, int average, void* done_event, void* stream) {
  X2Comm* c = (X2Comm*)comm;
  X2_REQUIRE(c && buf && count > 0 && nccl_dtype(dtype) >= 0, "x2_comm_allreduce_bucket: count=%ld dtype=%d", count, dtype);
  X2_NCCL(g_rccl.all_reduce(buf, buf, (size_t)count, nccl_dtype(dtype), average ? 4 : 0, c->nccl, (hipStream_t)stream),
          "x2_comm_allreduce_bucket");
  return record(done_event, stream, "x2_comm_allreduce_bucket");
}

// recv[r * count_per_rank ...] = rank r's send (xvlm.py:140-160 forward; the backward keeps the local slice and needs
// no collective).
extern "C" int x2_comm_allgather(void* comm, const void* send, void* recv, long count_per_rank, int dtype, void* done_event, void* stream) {
  X2Comm* c = (X2Comm*)comm;
  X2_REQUIRE(c && send && recv && count_per_rank > 0 && nccl_dtype(dtype) >= 0, "x2_comm_allgather: count=%ld dtype=%d", count_per_rank, dtype);
  X2_NCCL(g_rccl.all_gather(send, recv, (size_t)count_per_rank, nccl_dtype(dtype), c->nccl, (hipStream_t)stream), "x2_comm_allgather");
  return record(done_event, stream, "x2_comm_allgather");
}

// One flat message from `root` (the reference broadcasts ~600 tensors one by one, apex_ddp_accelerator.py:70-77).
extern "C" int x2_comm_broadcast(void* comm, void* buf, long count, int dtype, int root, void* done_event, void* stream) {
  X2Comm* c = (X2Comm*)comm;
  X2_REQUIRE(c && buf && count > 0 && nccl_dtype(dtype) >= 0 && root >= 0 && root < c->world, "x2_comm_broadcast: count=%ld root=%d", count, root);
  X2_NCCL(g_rccl.broadcast(buf, buf, (size_t)count, nccl_dtype(dtype), root, c->nccl, (hipStream_t)stream), "x2_comm_broadcast");
  return record(done_event, stream, "x2_comm_broadcast");
}

extern "C" int x2_comm_info(void* comm, int* rank, int* world) {
  X2Comm* c = (X2Comm*)comm;
  X2_REQUIRE(c, "x2_comm_info: null communicator");
  int n = 0;
  X2_NCCL(g_rccl.comm_count(c->nccl, &n), "x2_comm_info");
  X2_REQUIRE(n == c->world, "x2_comm_info: RCCL reports %d ranks, expected %d", n, c->world);
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return X2_OK;
}

extern "C" int x2_comm_destroy(void* comm) {
  X2Comm* c = (X2Comm*)comm;
  X2_REQUIRE(c, "x2_comm_destroy: null communicator");
  X2_NCCL(g_rccl.comm_destroy(c->nccl), "x2_comm_destroy");
  delete c;
  return X2_OK;
}
