// pkv_coll.hip — the one exchange step of the head-sharded path: an all-gather of the selected int32 indices over
// RCCL / xGMI.  The reference has no collective (run_longbench.py:390 places layers with
// device_map="auto" only); this is what a tensor-parallel host calls after the local pkv_compress.
//
// libpkv does not link RCCL: a process must talk to ONE RCCL (PyTorch ships its own librccl.so), so the three entry
// points needed are resolved at first use from whichever RCCL the process already has (global scope, then the
// already-loaded library by soname; only when there is none: PKV_RCCL_LIB, then a plain dlopen of the system library).  No
// communicator is created here: the caller owns it.
#include "../../include/pkv.h"
#include "pkv_kernels.hpp"

#include <dlfcn.h>
#include <stdlib.h>
#include <mutex>

namespace {

typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef int (*nccl_count_fn)(const void*, int*);
typedef const char* (*nccl_errstr_fn)(int);

struct Rccl {
  nccl_allgather_fn all_gather = nullptr;
  nccl_count_fn comm_count = nullptr;
  nccl_errstr_fn err_string = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local int g_last_nccl = 0;

void* lookup(void* h, const char* name) { return h ? dlsym(h, name) : nullptr; }

const Rccl& rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl;
  g_rccl.tried = true;
  void* handles[4] = {RTLD_DEFAULT, nullptr, nullptr, nullptr};
  const char* names[] = {"librccl.so", "librccl.so.1"};
  int nh = 1;
  for (const char* n : names) handles[nh++] = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process
  for (int i = 0; i < nh && !g_rccl.all_gather; ++i) {
    if (i > 0 && !handles[i]) continue;
    void* ag = i == 0 ? dlsym(RTLD_DEFAULT, "ncclAllGather") : lookup(handles[i], "ncclAllGather");
    if (!ag) continue;
    void* h = handles[i];
    g_rccl.all_gather = reinterpret_cast<nccl_allgather_fn>(ag);
    g_rccl.comm_count = reinterpret_cast<nccl_count_fn>(i == 0 ? dlsym(RTLD_DEFAULT, "ncclCommCount") : lookup(h, "ncclCommCount"));
    g_rccl.err_string = reinterpret_cast<nccl_errstr_fn>(i == 0 ? dlsym(RTLD_DEFAULT, "ncclGetErrorString") : lookup(h, "ncclGetErrorString"));
  }
  if (!g_rccl.all_gather) {                       // nothing loaded yet: PKV_RCCL_LIB (explicit path) first, then the system RCCL.
    const char* env = getenv("PKV_RCCL_LIB");     // Never consulted when the process already holds an RCCL: a second copy
    for (const char* n : {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {   // next to it corrupts the heap at exit
      if (!n || !*n) continue;
      void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_rccl.all_gather = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
      g_rccl.comm_count = reinterpret_cast<nccl_count_fn>(dlsym(h, "ncclCommCount"));
      g_rccl.err_string = reinterpret_cast<nccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
      if (g_rccl.all_gather) break;
    }
  }
  return g_rccl;
}

// rank-major [N][B][Hl][k] -> head-major [B][N*Hl][k] (only needed when B > 1; B == 1 is already in place)
__global__ __launch_bounds__(256) void regroup_kernel(const int32_t* in, int32_t* out, int N, int B, int Hl, int k) {
  const int64_t total = (int64_t)N * B * Hl * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t j = i % k, t = i / k;              // out index = ((b * N + r) * Hl + h) * k + j
    const int64_t h = t % Hl, t2 = t / Hl;
    const int64_t r = t2 % N, b = t2 / N;
    out[i] = in[((r * B + b) * Hl + h) * k + j];
  }
}

}  // namespace

extern "C" {

int pkv_last_nccl_error(void) { return g_last_nccl; }

int pkv_allgather_indices(void* nccl_comm, const int32_t* idx_local, int32_t* idx_all, int32_t B, int32_t H_local,
                          int32_t k, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  if (!nccl_comm || !idx_local || !idx_all) return PKV_ERR_NULL;
  if (B < 1 || H_local < 1 || k < 1) return PKV_ERR_SHAPE;
  const Rccl& r = rccl();
  if (!r.all_gather || !r.comm_count) return PKV_ERR_UNSUPPORTED;        // no RCCL in this process and none loadable
  int n = 0;
  int rc = r.comm_count(nccl_comm, &n);
  if (rc != 0 || n < 1) { g_last_nccl = rc; return PKV_ERR_COLLECTIVE; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t count = (size_t)B * H_local * k;
  const bool regroup = B > 1 && n > 1;
  if (regroup && (!ws || ws_bytes < (size_t)n * count * 4)) return PKV_ERR_WORKSPACE;
  rc = r.all_gather(idx_local, regroup ? ws : idx_all, count, /*ncclInt32*/ 2, nccl_comm, st);
  if (rc != 0) { g_last_nccl = rc; return PKV_ERR_COLLECTIVE; }
  if (regroup) {
    const int64_t total = (int64_t)n * count;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(regroup_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const int32_t*>(ws), idx_all, n, B, H_local, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pkv_set_last_hip_error((int)e); return PKV_ERR_HIP; }
  }
  return PKV_OK;
}

}  // extern "C"
