// host_cabi.cpp - a host that is NOT Python: SnapKV update_kv through the C ABI of include/pkv.h with nothing but
// the HIP runtime.  Fills Q/K/V [B,H,S,128] bf16 with a fixed LCG, runs pkv_compress,
// and checks on the host what can be checked without a second implementation: indices valid / distinct, scores of
// the selected tokens descending with index-ascending ties (pkv_score_window re-run), K_c/V_c = exact gather of the
// selected rows + the window tail.  Exit code 0 = all good.  Build: hipcc -I include examples/host_cabi.cpp
//   -L pyramidkv_amd -lpkv -lrccl -Wl,-rpath,'$ORIGIN/../pyramidkv_amd'
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "pkv.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define PKV_OK_(x) do { int r_ = (x); if (r_ != 0) { printf("pkv error %d (%s) at line %d\n", r_, pkv_strerror(r_), __LINE__); return 3; } } while (0)

static uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const bool with_rccl = !(argc > 1 && !strcmp(argv[1], "--no-rccl"));   // bisecting aid: skip the communicator
  const int B = 1, H = 4, S = 4096, D = 128, w = 8, cap = 64, k = cap - w, L = S - w;
  const size_t n = (size_t)B * H * S * D;
  std::vector<uint16_t> hq(n), hk(n), hv(n);
  uint64_t st = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((int64_t)(st >> 40) - (1 << 23)) / (float)(1 << 22); };
  for (size_t i = 0; i < n; ++i) { hq[i] = f32_to_bf16(rnd()); hk[i] = f32_to_bf16(rnd()); hv[i] = f32_to_bf16(rnd()); }

  pkv_desc d;
  memset(&d, 0, sizeof d);
  d.struct_size = sizeof d;      // the layout THIS host was compiled against: the library reads no further
  d.dtype = PKV_BF16; d.B = B; d.H = H; d.S = S; d.D = D; d.kv_group = 1;
  const int64_t strides[3] = {(int64_t)H * S * D, (int64_t)S * D, D};
  for (int i = 0; i < 3; ++i) d.q_stride[i] = d.k_stride[i] = d.v_stride[i] = strides[i];
  d.window = w; d.pool_kind = PKV_POOL_MAX; d.pool_kernel = 7; d.reduce = PKV_REDUCE_SUM; d.scale_mode = PKV_SCALE_DIV; d.topk = k;

  void *q, *kk, *v, *ko, *vo, *ws, *sc; int32_t* idx;
  const size_t wsb = pkv_workspace_bytes(&d);
  const int64_t sstride = (L + 7) / 8 * 8;
  HIP_OK(hipMalloc(&q, n * 2)); HIP_OK(hipMalloc(&kk, n * 2)); HIP_OK(hipMalloc(&v, n * 2));
  HIP_OK(hipMalloc(&ko, (size_t)B * H * cap * D * 2)); HIP_OK(hipMalloc(&vo, (size_t)B * H * cap * D * 2));
  HIP_OK(hipMalloc(&ws, wsb)); HIP_OK(hipMalloc(&sc, (size_t)B * H * sstride * 2)); HIP_OK(hipMalloc((void**)&idx, (size_t)B * H * k * 4));
  HIP_OK(hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(kk, hk.data(), n * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  PKV_OK_(pkv_compress(&d, q, kk, v, ko, vo, idx, ws, wsb, stream));
  PKV_OK_(pkv_score_window(&d, q, kk, sc, sstride, ws, wsb, stream));
  HIP_OK(hipStreamSynchronize(stream));

  std::vector<int32_t> hidx((size_t)B * H * k);
  std::vector<uint16_t> hko((size_t)B * H * cap * D), hvo(hko.size()), hsc((size_t)B * H * sstride);
  HIP_OK(hipMemcpy(hidx.data(), idx, hidx.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hko.data(), ko, hko.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hvo.data(), vo, hvo.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hsc.data(), sc, hsc.size() * 2, hipMemcpyDeviceToHost));

  int bad = 0;
  for (int bh = 0; bh < B * H && !bad; ++bh) {
    std::vector<char> seen(L, 0);
    const uint16_t* srow = &hsc[(size_t)bh * sstride];
    float kth = 0.f;
    for (int j = 0; j < k; ++j) {
      const int i = hidx[(size_t)bh * k + j];
      if (i < 0 || i >= L || seen[i]) { printf("row %d: bad/duplicate index %d\n", bh, i); bad = 1; break; }
      seen[i] = 1;
      const float s = bf16_to_f32(srow[i]);
      if (j) {
        const int ip = hidx[(size_t)bh * k + j - 1];
        const float sp = bf16_to_f32(srow[ip]);
        if (s > sp || (s == sp && i < ip)) { printf("row %d: order violated at %d\n", bh, j); bad = 1; break; }
      }
      kth = s;
      if (memcmp(&hko[((size_t)bh * cap + j) * D], &hk[((size_t)bh * S + i) * D], D * 2) ||
          memcmp(&hvo[((size_t)bh * cap + j) * D], &hv[((size_t)bh * S + i) * D], D * 2)) { printf("row %d: gather mismatch at %d\n", bh, j); bad = 1; break; }
    }
    for (int i = 0; i < L && !bad; ++i)
      if (!seen[i] && bf16_to_f32(srow[i]) > kth) { printf("row %d: unselected token %d beats the k-th score\n", bh, i); bad = 1; }
    for (int r = 0; r < w && !bad; ++r)
      if (memcmp(&hko[((size_t)bh * cap + k + r) * D], &hk[((size_t)bh * S + L + r) * D], D * 2) ||
          memcmp(&hvo[((size_t)bh * cap + k + r) * D], &hv[((size_t)bh * S + L + r) * D], D * 2)) { printf("row %d: window tail mismatch\n", bh); bad = 1; }
  }
  // the one exchange step of the head-sharded path over a real RCCL communicator (this box has one GPU: nranks = 1);
  // a tensor-parallel host calls exactly this after its local pkv_compress (include/pkv.h: pkv_allgather_indices)
  if (with_rccl) {
    ncclUniqueId uid;
    ncclComm_t comm;
    if (ncclGetUniqueId(&uid) != ncclSuccess || ncclCommInitRank(&comm, 1, uid, 0) != ncclSuccess) { printf("RCCL init failed\n"); bad = 1; }
    else {
      int32_t* all;
      HIP_OK(hipMalloc((void**)&all, (size_t)B * H * k * 4));
      HIP_OK(hipMemsetAsync(all, 0xff, (size_t)B * H * k * 4, stream));
      PKV_OK_(pkv_allgather_indices(comm, idx, all, B, H, k, nullptr, 0, stream));
      HIP_OK(hipStreamSynchronize(stream));
      std::vector<int32_t> hall((size_t)B * H * k);
      HIP_OK(hipMemcpy(hall.data(), all, hall.size() * 4, hipMemcpyDeviceToHost));
      if (memcmp(hall.data(), hidx.data(), hall.size() * 4)) { printf("pkv_allgather_indices: gathered indices differ\n"); bad = 1; }
      HIP_OK(hipFree(all));
      // libpkv resolved its RCCL entry points during that call: it must have taken THIS process's copy and loaded no second one
      // (PKV_RCCL_LIB pointing at another build must be ignored here; two RCCLs in one process corrupt the heap at exit)
      {
        std::vector<std::string> copies;
        if (FILE* maps = fopen("/proc/self/maps", "r")) {
          char line[1024];
          while (fgets(line, sizeof line, maps)) {
            const char* path = strchr(line, '/');
            if (!path || !strstr(path, "librccl")) continue;
            std::string s(path);
            while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
            bool known = false;
            for (const std::string& c : copies) known |= c == s;
            if (!known) copies.push_back(s);
          }
          fclose(maps);
        }
        if (copies.size() != 1) {
          printf("%zu RCCL libraries mapped in this process (expected exactly one):\n", copies.size());
          for (const std::string& c : copies) printf("  %s\n", c.c_str());
          bad = 1;
        }
      }
      if (ncclCommDestroy(comm) != ncclSuccess) { printf("ncclCommDestroy failed\n"); bad = 1; }
    }
  }
  // error convention: a bad descriptor is reported, nothing aborts
  pkv_desc e = d; e.D = 100;
  if (pkv_compress(&e, q, kk, v, ko, vo, idx, ws, wsb, stream) != PKV_ERR_SHAPE) { printf("D=100 not rejected\n"); bad = 1; }
  e = d; e.struct_size = 0;      // a host that never heard of struct_size (or an uninitialised descriptor)
  if (pkv_compress(&e, q, kk, v, ko, vo, idx, ws, wsb, stream) != PKV_ERR_ABI) { printf("struct_size 0 not rejected\n"); bad = 1; }
  // orderly teardown: nothing of this host is left for the runtimes' exit handlers to find
  for (void* p : {q, kk, v, ko, vo, ws, sc, (void*)idx}) HIP_OK(hipFree(p));
  HIP_OK(hipStreamDestroy(stream));
  printf(bad ? "host_cabi: FAILED\n" : "host_cabi: ok (pkv_version %d, %d heads x top-%d of %d, workspace %zu bytes)\n", pkv_version(), B * H, k, L, wsb);
  fflush(stdout);
  return bad;
}
