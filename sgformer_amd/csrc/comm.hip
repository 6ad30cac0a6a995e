// comm.hip — SURVEY.md §8b / §8e: the collectives of the node-sharded run behind the C ABI (sgf_comm_*), for a consumer that
// binds libsgf.so from C / C++ (the Python host side keeps using torch.distributed, whose `nccl` backend IS RCCL: both end in
// the same library).  The reference has no multi-GPU path at all; what is exchanged is defined by the kernels'
// partial-sum layouts (include/sgf.h): the attention statistics [K^T V | sum K | ||Q||^2 | ||K||^2] and the BatchNorm sums
// (all-reduce), the halo / whole-shard rows of the SpMM operand (all-gather, all-to-all), the parameter gradients (all-reduce).
//
// librccl.so is loaded with dlopen() at the first sgf_comm_* call — libsgf.so itself carries no link-time dependency on it,
// so a single-GPU consumer never needs RCCL installed.  Thin by design: one communicator per handle, every call takes the
// caller's stream, nothing is staged or copied here, errors come back as SGF_E_HIP with RCCL's own text.
#include "common.h"

#include <dlfcn.h>
// The library is bound with dlopen(), so its headers are not a build requirement either: without them the few types and
// enumerators used here are declared locally (values fixed by NCCL's / RCCL's public ABI, nccl.h).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#endif

namespace sgf {
namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
    auto sym = [&](const char* s) { return dlsym(x.lib, s); };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.Send && x.Recv &&
           x.GroupStart && x.GroupEnd && x.GetErrorString;
    return x;
  }();
  return r;
}

struct Comm {
  ncclComm_t comm;
  int world, rank;
};

#define SGF_RCCL(expr)                                                                       \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    if (_r != ncclSuccess) {                                                                 \
      ::sgf::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(_r)); \
      return SGF_E_HIP;                                                                      \
    }                                                                                        \
  } while (0)

int need_rccl(const char* fn) {
  SGF_REQUIRE(rccl().ok, SGF_E_UNSUPPORTED, "%s: librccl.so (or one of its entry points) could not be loaded with dlopen()", fn);
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

extern "C" int32_t sgf_comm_available(void) { return sgf::rccl().ok ? 1 : 0; }

extern "C" int32_t sgf_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int sgf_comm_unique_id(void* id_host) {
  using namespace sgf;
  int rc = need_rccl("sgf_comm_unique_id");
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(id_host, SGF_E_INVALID, "sgf_comm_unique_id: null pointer");
  SGF_RCCL(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id_host)));
  return SGF_OK;
}

extern "C" int sgf_comm_create(void** comm_out, int32_t world, int32_t rank, const void* id_host) {
  using namespace sgf;
  int rc = need_rccl("sgf_comm_create");
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(comm_out && id_host && world >= 1 && rank >= 0 && rank < world, SGF_E_INVALID,
              "sgf_comm_create: bad arguments (world %d, rank %d)", world, rank);
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  SGF_RCCL(rccl().CommInitRank(&c, world, id, rank));      // on the CURRENT device of the calling thread
  *comm_out = new Comm{c, world, rank};
  return SGF_OK;
}

extern "C" int sgf_comm_destroy(void* comm) {
  using namespace sgf;
  if (!comm) return SGF_OK;
  Comm* c = static_cast<Comm*>(comm);
  ncclResult_t r = rccl().ok ? rccl().CommDestroy(c->comm) : ncclSuccess;
  delete c;
  SGF_REQUIRE(r == ncclSuccess, SGF_E_HIP, "sgf_comm_destroy: %s", rccl().GetErrorString(r));
  return SGF_OK;
}

extern "C" int sgf_comm_all_reduce_f32(void* comm, float* buf, int64_t count, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(comm && (buf || count == 0) && count >= 0, SGF_E_INVALID, "sgf_comm_all_reduce_f32: bad arguments");
  if (count == 0) return SGF_OK;
  Comm* c = static_cast<Comm*>(comm);
  SGF_RCCL(rccl().AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat32, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
  return SGF_OK;
}

extern "C" int sgf_comm_all_gather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(comm && bytes_per_rank >= 0 && (bytes_per_rank == 0 || (send && recv)), SGF_E_INVALID,
              "sgf_comm_all_gather: bad arguments");
  if (bytes_per_rank == 0) return SGF_OK;
  Comm* c = static_cast<Comm*>(comm);
  SGF_RCCL(rccl().AllGather(send, recv, static_cast<size_t>(bytes_per_rank), ncclInt8, c->comm, static_cast<hipStream_t>(stream)));
  return SGF_OK;
}

extern "C" int sgf_comm_all_to_all(void* comm, const void* send, const int64_t* send_offset_host, const int64_t* send_bytes_host,
                                   void* recv, const int64_t* recv_offset_host, const int64_t* recv_bytes_host, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(comm && send_offset_host && send_bytes_host && recv_offset_host && recv_bytes_host, SGF_E_INVALID,
              "sgf_comm_all_to_all: null pointer");
  Comm* c = static_cast<Comm*>(comm);
  hipStream_t st = static_cast<hipStream_t>(stream);
  SGF_RCCL(rccl().GroupStart());
  // An error inside the group must not return with the group open (every later collective of this thread would be queued
  // and never launched): remember the first one, close the group, then report it.
  ncclResult_t first = ncclSuccess;
  for (int p = 0; p < c->world && first == ncclSuccess; ++p) {
    if (send_bytes_host[p] > 0)
      first = rccl().Send(static_cast<const char*>(send) + send_offset_host[p], static_cast<size_t>(send_bytes_host[p]), ncclInt8,
                          p, c->comm, st);
    if (first == ncclSuccess && recv_bytes_host[p] > 0)
      first = rccl().Recv(static_cast<char*>(recv) + recv_offset_host[p], static_cast<size_t>(recv_bytes_host[p]), ncclInt8, p,
                          c->comm, st);
  }
  const ncclResult_t end = rccl().GroupEnd();
  SGF_REQUIRE(first == ncclSuccess, SGF_E_HIP, "sgf_comm_all_to_all: send / recv -> %s", rccl().GetErrorString(first));
  SGF_REQUIRE(end == ncclSuccess, SGF_E_HIP, "sgf_comm_all_to_all: group end -> %s", rccl().GetErrorString(end));
  return SGF_OK;
}
