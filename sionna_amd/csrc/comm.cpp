// The one collective of the path: sum of the int64 error counters over the ranks of a Monte-Carlo run.
//
// Replaces the cross-replica gathers of the reference's driver (src/sionna/phy/utils/misc.py:546-547 inside the
// tf.distribute strategy of :616-655: every replica's full bit tensors are gathered before counting - 1.48 GB per GPU
// and iteration at config C3) by ONE RCCL allReduce(sum) over {bit_errors, block_errors, num_bits, num_blocks}: 32 bytes,
// latency-bound, xGMI bandwidth unused.  The Python host reaches RCCL through torch.distributed (backend "nccl");
// THIS file is the same collective for a host that is not Python: a C client (tests/cabi/) creates one communicator per
// process / GPU from a 128-byte id that rank 0 makes and hands to the others out of band (file, socket, MPI, environment).
//
// librccl is bound lazily with dlopen (like rocFFT in ofdm_time.hip): libsionna_amd.so carries no link-time dependency on
// it, single-GPU users never load it, and inside a PyTorch process the already-mapped librccl.so.1 is reused instead of a
// second copy.
#include "common.h"

#include <dlfcn.h>
// The library is bound at run time, so the header only supplies declarations: without the RCCL development package the
// few this file uses are stated here (rccl.h of ROCm 7: opaque communicator, 128-byte id, the enum values of the one call).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <cstring>
#include <mutex>

static_assert(SAMD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "samd_comm id size must be RCCL's");

namespace samd {
namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) get_error_string = nullptr;
  bool ok = false;
  std::string err;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

template <typename F>
bool bind_sym(void* h, const char* name, F* out, std::string* err) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  if (*out) return true;
  *err = std::string("librccl: missing symbol ") + name;
  return false;
}

bool load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  if (!g_rccl.handle) {
    // a copy that the process has mapped already (PyTorch ships its own librccl.so with this soname) wins
    g_rccl.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !g_rccl.handle && i < 3; ++i) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) {
      const char* e = dlerror();
      g_rccl.err = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
      return false;
    }
  }
  void* h = g_rccl.handle;
  g_rccl.ok = bind_sym(h, "ncclGetUniqueId", &g_rccl.get_unique_id, &g_rccl.err) &&
              bind_sym(h, "ncclCommInitRank", &g_rccl.comm_init_rank, &g_rccl.err) &&
              bind_sym(h, "ncclCommDestroy", &g_rccl.comm_destroy, &g_rccl.err) &&
              bind_sym(h, "ncclAllReduce", &g_rccl.all_reduce, &g_rccl.err) &&
              bind_sym(h, "ncclGetErrorString", &g_rccl.get_error_string, &g_rccl.err);
  return g_rccl.ok;
}

int rccl_status(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return SAMD_OK;
  set_error(std::string(what) + ": " + g_rccl.get_error_string(r));
  return SAMD_ERR_HIP;
}

}  // namespace
}  // namespace samd

struct samd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

using namespace samd;

extern "C" int samd_comm_unique_id(void* id_out) {
  SAMD_REQUIRE(id_out, "null argument");
  if (!load_rccl()) {
    set_error(g_rccl.err);
    return SAMD_ERR_UNSUPPORTED;
  }
  ncclUniqueId id;
  const int rc = rccl_status(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
  if (rc != SAMD_OK) return rc;
  memcpy(id_out, &id, sizeof(id));
  return SAMD_OK;
}

extern "C" int samd_comm_create(const void* id, int rank, int world_size, samd_comm_t** out) {
  SAMD_REQUIRE(id && out, "null argument");
  SAMD_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "rank must be in [0, world_size)");
  *out = nullptr;
  if (!load_rccl()) {
    set_error(g_rccl.err);
    return SAMD_ERR_UNSUPPORTED;
  }
  samd_comm* c = new samd_comm;
  c->rank = rank;
  c->world = world_size;
  if (hipGetDevice(&c->device) != hipSuccess) {          // the communicator is bound to the calling thread's current device
    delete c;
    set_error("hipGetDevice failed");
    return SAMD_ERR_HIP;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const int rc = rccl_status(g_rccl.comm_init_rank(&c->comm, world_size, uid, rank), "ncclCommInitRank");
  if (rc != SAMD_OK) {
    delete c;
    return rc;
  }
  *out = c;
  return SAMD_OK;
}

extern "C" int samd_comm_rank(const samd_comm_t* c) { return c ? c->rank : -1; }
extern "C" int samd_comm_world_size(const samd_comm_t* c) { return c ? c->world : -1; }

extern "C" int samd_comm_allreduce_sum_i64(samd_comm_t* c, int64_t* counters, int64_t count, void* stream) {
  SAMD_REQUIRE(c && c->comm && counters && count >= 0, "bad argument");
  if (count == 0) return SAMD_OK;
  return rccl_status(g_rccl.all_reduce(counters, counters, (size_t)count, ncclInt64, ncclSum, c->comm, (hipStream_t)stream),
                     "ncclAllReduce");
}

extern "C" void samd_comm_destroy(samd_comm_t* c) {
  if (!c) return;
  if (c->comm && g_rccl.ok) g_rccl.comm_destroy(c->comm);
  delete c;
}
