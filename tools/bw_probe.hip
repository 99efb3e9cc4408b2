// bw_probe.hip - achievable HBM bandwidth on this MI355X for the access patterns the pkv kernels use.
// The honest denominator next to the 8 TB/s spec peak.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t n, uint32_t* sink) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// the logits kernel's fragment pattern: a wave reads 16 rows x 64 B per instruction, 16 loads per lane
__global__ __launch_bounds__(256) void frag_kernel(const uint16_t* __restrict__ src, size_t rows, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const size_t tile = blockIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 v[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const size_t r = tile * 256 + wave * 64 + t * 16 + li;
    const uint16_t* row = src + (r < rows ? r : rows - 1) * 128 + lg * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) v[t * 4 + kk] = *reinterpret_cast<const u32x4*>(row + kk * 32);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) acc ^= v[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}


// ---- candidate K-stream patterns for the logits kernel (one 256-row tile of 256-B rows per workgroup) ----
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// frag pattern with nontemporal loads
__global__ __launch_bounds__(256) void frag_nt_kernel(const uint16_t* __restrict__ src, size_t rows, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const size_t tile = blockIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 v[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const size_t r = tile * 256 + wave * 64 + t * 16 + li;
    const uint16_t* row = src + (r < rows ? r : rows - 1) * 128 + lg * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) v[t * 4 + kk] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + kk * 32));
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) acc ^= v[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// row-major: every wave instruction reads 1 KB contiguous (4 whole rows); optional LDS transpose to the MFMA layout
template <bool NT, bool LDS>
__global__ __launch_bounds__(256) void rowmajor_kernel(const uint16_t* __restrict__ src, size_t rows, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[LDS ? 4096 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const size_t tile = blockIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int rl = 4 * j + lg;                       // row within the wave's 64
    const size_t r = tile * 256 + wave * 64 + rl;
    const int c = li ^ (rl & 15);                    // pre-swizzled source chunk -> linear LDS slot li
    const u32x4* ptr = reinterpret_cast<const u32x4*>(src + (r < rows ? r : rows - 1) * 128) + c;
    v[j] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
  }
  if (LDS) {
    u32x4* w = lds + wave * 1024;
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j * 64 + lane] = v[j];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rl = t * 16 + li;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) v[t * 4 + kk] = w[rl * 16 + ((kk * 4 + lg) ^ (rl & 15))];
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) acc ^= v[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// LDS-direct loads (global_load_lds_dwordx4), AUX = 0 default policy, 2 = nt; then MFMA-layout reads
template <int AUX>
__global__ __launch_bounds__(256) void glds_kernel(const uint16_t* __restrict__ src, size_t rows, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const size_t tile = blockIdx.x;
  u32x4* w = lds + wave * 1024;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int rl = 4 * j + lg;
    const size_t r = tile * 256 + wave * 64 + rl;
    const int c = li ^ (rl & 15);
    const u32x4* ptr = reinterpret_cast<const u32x4*>(src + (r < rows ? r : rows - 1) * 128) + c;
    __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lptr_t)(w + j * 64), 16, 0, AUX);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rl = t * 16 + li;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc ^= w[rl * 16 + ((kk * 4 + lg) ^ (rl & 15))];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u];
    }
  }
}

template <typename F> float time_ms(F f, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB > 256 MB Infinity Cache
  const size_t n = bytes / 16;
  u32x4 *src, *dst; uint32_t* sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 64);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  printf("{\n");
  int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
  for (int g : grids) {
    float r4 = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(g), dim3(256), 0, 0, src, n, sink); });
    float r8 = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(g), dim3(256), 0, 0, src, n, sink); });
    float r8n = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(g), dim3(256), 0, 0, src, n, sink); });
    float c4 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(g), dim3(256), 0, 0, src, dst, n); });
    float c4n = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(g), dim3(256), 0, 0, src, dst, n); });
    printf(" \"grid%d\": {\"read_u4_GBps\": %.0f, \"read_u8_GBps\": %.0f, \"read_u8_nt_GBps\": %.0f, \"copy_u4_GBps\": %.0f, \"copy_u4_nt_GBps\": %.0f},\n",
           g, bytes / r4 / 1e6, bytes / r8 / 1e6, bytes / r8n / 1e6, 2.0 * bytes / c4 / 1e6, 2.0 * bytes / c4n / 1e6);
  }
  // one-pass grids like the pkv kernels (268 MB = the K tensor of one layer at S=32k, and 1 GiB)
  for (size_t rows : {(size_t)32 * 32768, (size_t)4 * 1024 * 1024}) {
    float f = time_ms([&] { hipLaunchKernelGGL(frag_kernel, dim3((rows + 255) / 256), dim3(256), 0, 0, (const uint16_t*)src, rows, sink); });
    printf(" \"frag_pattern_rows%zu_GBps\": %.0f,\n", rows, rows * 256.0 / f / 1e6);
    const size_t nn = rows * 16;
    float o = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3((nn + 1023) / 1024), dim3(256), 0, 0, src, nn, sink); });
    printf(" \"onepass_read_rows%zu_GBps\": %.0f,\n", rows, rows * 256.0 / o / 1e6);
  }
  for (size_t rows : {(size_t)32 * 32768, (size_t)4 * 1024 * 1024}) {
    const dim3 g((rows + 255) / 256), b(256);
    const uint16_t* s16 = (const uint16_t*)src;
#define PROBE(name, kern) do { float t_ = time_ms([&] { hipLaunchKernelGGL(kern, g, b, 0, 0, s16, rows, sink); }); \
    printf(" \"%s_rows%zu_GBps\": %.0f,\n", name, rows, rows * 256.0 / t_ / 1e6); } while (0)
    PROBE("frag_nt", frag_nt_kernel);
    PROBE("rowmajor", (rowmajor_kernel<false, false>));
    PROBE("rowmajor_nt", (rowmajor_kernel<true, false>));
    PROBE("rowmajor_lds", (rowmajor_kernel<false, true>));
    PROBE("rowmajor_nt_lds", (rowmajor_kernel<true, true>));
    PROBE("glds", glds_kernel<0>);
    PROBE("glds_nt", glds_kernel<2>);
#undef PROBE
  }
  // Re-reading a buffer that fits the 256 MB Infinity Cache (MALL): the rate a SECOND pass over one layer's K would see
  // (16.8 MB = the logits of a layer, 67 MB = un-expanded K at S = 32k, 134 MB, 268 MB = expanded K).  Back-to-back launches
  // over the same bytes, one-pass grids; "cold" = the same launch right after 1 GiB of other traffic.
  for (size_t mb : {(size_t)16, (size_t)64, (size_t)128, (size_t)256}) {
    const size_t nn = mb * 1024 * 1024 / 16;
    const dim3 g((nn + 2047) / 2048), b(256);
    float warm = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, false>), g, b, 0, 0, src, nn, sink); });
    float warm_nt = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, true>), g, b, 0, 0, src, nn, sink); });
    float cold = time_ms([&] {
      hipLaunchKernelGGL((read_kernel<8, true>), dim3(4096), dim3(256), 0, 0, src + nn, n - nn, sink);     // evict
      hipLaunchKernelGGL((read_kernel<8, false>), g, b, 0, 0, src, nn, sink);
    }, 10);
    float evict = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(4096), dim3(256), 0, 0, src + nn, n - nn, sink); }, 10);
    printf(" \"reread_%zuMiB\": {\"warm_GBps\": %.0f, \"warm_nt_GBps\": %.0f, \"warm_us\": %.1f, \"after_1GiB_of_other_reads_us\": %.1f},\n",
           mb, nn * 16.0 / warm / 1e6, nn * 16.0 / warm_nt / 1e6, warm * 1e3, (cold - evict) * 1e3);
  }
  printf(" \"bytes\": %zu\n}\n", bytes);
  return 0;
}
