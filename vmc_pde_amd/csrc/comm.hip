// The one collective of a TDVP step: sum of the packed (S, F, moments) buffer over the ranks of a node, RCCL over xGMI.
//
// Replaces the reference's reducers mpi_wrapper.py:129-163 (global_sum: pmap psum + host-buffer MPI.Allreduce, called ten
// times per TDVP step, two of them on P x P matrices with a device -> host -> device round trip each).  Here every rank
// packs all partial sums of a step into ONE device buffer (vmcpde.h, vmcpde_gram) and this entry point reduces it in place
// on the caller's stream.
//
// RCCL is resolved at run time (dlsym): a process that already carries an RCCL (PyTorch ships its own librccl.so) must not
// get a second copy linked in, and the ncclComm_t handed to vmcpde_allreduce must belong to the library that executes the
// call -- so the symbols are taken from what the process has loaded, falling back to dlopen("librccl.so").  The library is
// not needed (and not touched) by single-GPU runs.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

using namespace vmcpde;

namespace {

// ABI subset of rccl.h (ROCm 7.2 / NCCL 2.x): stable C enums and signatures
constexpr int kNcclSuccess = 0;
constexpr int kNcclFloat64 = 8;   // ncclFloat64 / ncclDouble
constexpr int kNcclSum = 0;       // ncclSum
struct UniqueId { char internal[128]; };   // NCCL_UNIQUE_ID_BYTES

typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*CommIntFn)(void*, int*);
typedef int (*GetVersionFn)(int*);

struct Rccl {
  AllReduceFn all_reduce = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  CommIntFn comm_count = nullptr, comm_user_rank = nullptr, comm_device = nullptr;
  GetVersionFn get_version = nullptr;
  bool tried = false, ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

void* lookup(void* handle, const char* name) { return handle ? dlsym(handle, name) : dlsym(RTLD_DEFAULT, name); }

bool load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  void* handle = nullptr;                                   // nullptr: whatever the process already has (e.g. PyTorch's)
  if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
    const char* override_path = getenv("VMCPDE_RCCL_LIB");
    for (const char* cand : {override_path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (!cand) continue;
      handle = dlopen(cand, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) {
      set_error("RCCL not found: no ncclAllReduce in the process and dlopen(librccl.so) failed (%s)", dlerror());
      return false;
    }
  }
  g_rccl.all_reduce = (AllReduceFn)lookup(handle, "ncclAllReduce");
  g_rccl.get_unique_id = (GetUniqueIdFn)lookup(handle, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)lookup(handle, "ncclCommInitRank");
  g_rccl.comm_destroy = (CommDestroyFn)lookup(handle, "ncclCommDestroy");
  g_rccl.error_string = (GetErrorStringFn)lookup(handle, "ncclGetErrorString");
  g_rccl.comm_count = (CommIntFn)lookup(handle, "ncclCommCount");
  g_rccl.comm_user_rank = (CommIntFn)lookup(handle, "ncclCommUserRank");
  g_rccl.comm_device = (CommIntFn)lookup(handle, "ncclCommCuDevice");
  g_rccl.get_version = (GetVersionFn)lookup(handle, "ncclGetVersion");
  g_rccl.ok = g_rccl.all_reduce && g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy;
  if (!g_rccl.ok) set_error("RCCL is loaded but lacks ncclAllReduce / ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy");
  return g_rccl.ok;
}

int fail(const char* what, int rc) {
  set_error("%s failed: %s (ncclResult %d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
  return -3;
}

}  // namespace

extern "C" int vmcpde_comm_unique_id(void* id128) {
  VMC_REQUIRE(id128, "vmcpde_comm_unique_id: null argument");
  if (!load_rccl()) return -3;
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != kNcclSuccess) return fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int vmcpde_comm_init_rank(void** comm, int32_t n_ranks, const void* id128, int32_t rank) {
  VMC_REQUIRE(comm && id128, "vmcpde_comm_init_rank: null argument");
  VMC_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank %d outside [0,%d)", rank, n_ranks);
  if (!load_rccl()) return -3;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, n_ranks, id, rank);       // binds the communicator to the current HIP device
  if (rc != kNcclSuccess) return fail("ncclCommInitRank", rc);
  *comm = c;
  return 0;
}

extern "C" int vmcpde_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (!load_rccl()) return -3;
  const int rc = g_rccl.comm_destroy(comm);
  if (rc != kNcclSuccess) return fail("ncclCommDestroy", rc);
  return 0;
}

extern "C" int vmcpde_allreduce(double* packed, int64_t count, void* comm, void* stream) {
  VMC_REQUIRE(packed && comm, "vmcpde_allreduce: null argument");
  VMC_REQUIRE(count >= 1, "vmcpde_allreduce: empty buffer");
  if (!load_rccl()) return -3;
  const int rc = g_rccl.all_reduce(packed, packed, (size_t)count, kNcclFloat64, kNcclSum, comm, (hipStream_t)stream);
  if (rc != kNcclSuccess) return fail("ncclAllReduce", rc);
  return 0;
}

extern "C" int vmcpde_comm_info(void* comm, int32_t* out4) {
  VMC_REQUIRE(comm && out4, "vmcpde_comm_info: null argument");
  if (!load_rccl()) return -3;
  VMC_REQUIRE(g_rccl.comm_count && g_rccl.comm_user_rank && g_rccl.comm_device && g_rccl.get_version,
              "RCCL lacks ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion");
  int v[4] = {0, 0, 0, 0};
  int rc = g_rccl.comm_count(comm, &v[0]);
  if (rc != kNcclSuccess) return fail("ncclCommCount", rc);
  rc = g_rccl.comm_user_rank(comm, &v[1]);
  if (rc != kNcclSuccess) return fail("ncclCommUserRank", rc);
  rc = g_rccl.comm_device(comm, &v[2]);
  if (rc != kNcclSuccess) return fail("ncclCommCuDevice", rc);
  rc = g_rccl.get_version(&v[3]);
  if (rc != kNcclSuccess) return fail("ncclGetVersion", rc);
  for (int i = 0; i < 4; ++i) out4[i] = v[i];
  return 0;
}

extern "C" int vmcpde_device_bus_id(char* out, int32_t len) {
  VMC_REQUIRE(out && len >= 16, "vmcpde_device_bus_id: need a buffer of at least 16 bytes");
  int dev = 0;
  VMC_CHECK_HIP(hipGetDevice(&dev));
  VMC_CHECK_HIP(hipDeviceGetPCIBusId(out, len, dev));
  return 0;
}
