// rccl_dl.hip -- optional in-library RCCL: librccl.so is dlopen'ed at first use (no link-time dependency), so the one
// exchange step of the landmark-sharded bundle adjustments (the all-reduce of the packed reduced pose
// system per LM trial) can be issued on the library's own stream, between k_lba_pack and k_lba_assemble, without a
// host round trip.  The host only has to carry the 128-byte unique id from rank 0 to the other ranks (torch.distributed
// broadcast, MPI, a file ...).  The callback form of the sharded entries stays for hosts with their own collective.
#include <dlfcn.h>

#include <mutex>

#include "common.h"
#include "rccl_dl.h"

namespace vieo {

namespace {
// the slice of rccl.h that is used (RCCL 2.x ABI: ncclUniqueId is a 128-byte struct passed by value)
struct NcclId {
  char internal[128];
};
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(void**, int, NcclId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
const int kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_errstr errstr = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

const Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (!g_rccl.tried) {
    g_rccl.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (g_rccl.lib) {
      g_rccl.get_id = (fn_get_id)dlsym(g_rccl.lib, "ncclGetUniqueId");
      g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
      g_rccl.destroy = (fn_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
      g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.lib, "ncclAllReduce");
      g_rccl.errstr = (fn_errstr)dlsym(g_rccl.lib, "ncclGetErrorString");
      if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce) {
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
      }
    }
  }
  return g_rccl.lib ? &g_rccl : nullptr;
}
}  // namespace

int rccl_allreduce_sum_f64(void* comm, double* d_buf, size_t n, hipStream_t st) {
  const Rccl* R = rccl();
  if (!R || !comm) {
    set_error("RCCL is not available (librccl.so could not be loaded)");
    return VIEO_E_INVALID;
  }
  const int rc = R->allreduce(d_buf, d_buf, n, kNcclFloat64, kNcclSum, comm, st);
  if (rc != 0) {
    set_error("ncclAllReduce failed: %s", R->errstr ? R->errstr(rc) : "?");
    return VIEO_E_HIP;
  }
  return VIEO_OK;
}

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_rccl_available(void) { return rccl() ? 1 : 0; }

int vieo_rccl_unique_id(uint8_t* id128) {
  if (!id128) return VIEO_E_INVALID;
  const Rccl* R = rccl();
  if (!R) {
    set_error("RCCL is not available (librccl.so could not be loaded)");
    return VIEO_E_INVALID;
  }
  NcclId id;
  const int rc = R->get_id(&id);
  if (rc != 0) {
    set_error("ncclGetUniqueId failed: %s", R->errstr ? R->errstr(rc) : "?");
    return VIEO_E_HIP;
  }
  memcpy(id128, id.internal, 128);
  return VIEO_OK;
}

int vieo_rccl_comm_create(void** comm, const uint8_t* id128, int n_ranks, int rank) {
  if (!comm || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  const Rccl* R = rccl();
  if (!R) {
    set_error("RCCL is not available (librccl.so could not be loaded)");
    return VIEO_E_INVALID;
  }
  NcclId id;
  memcpy(id.internal, id128, 128);
  void* c = nullptr;
  const int nrc = R->init_rank(&c, n_ranks, id, rank);
  if (nrc != 0) {
    set_error("ncclCommInitRank failed: %s", R->errstr ? R->errstr(nrc) : "?");
    return VIEO_E_HIP;
  }
  *comm = c;
  return VIEO_OK;
}

int vieo_rccl_comm_destroy(void* comm) {
  const Rccl* R = rccl();
  if (R && comm) R->destroy(comm);
  return VIEO_OK;
}

}  // extern "C"
