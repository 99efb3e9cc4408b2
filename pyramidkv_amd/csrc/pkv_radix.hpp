// pkv_radix.hpp - 8-bit radix-select building blocks shared by topk_kernel (pkv_topk.hip) and the Ada-SnapKV budget
// kernels that work on un-sorted score rows (pkv_ada.hip): one 1024-thread workgroup per row, bank-spread LDS counters.
#pragma once
#include "pkv_common.hpp"

namespace pkv {

constexpr int TK_THREADS = 1024;
constexpr int TK_WAVES = 16;
constexpr int TK_CNT_WORDS = 256 * 32;   // counters[bin][lane&31], lo16 = lanes 0-31, hi16 = lanes 32-63

// sum the bank-spread counters X[bin][32] (lo16/hi16 halves) into hist[256]; all 1024 threads, 4 per bin
__device__ __forceinline__ void reduce_counters(const uint32_t* X, uint32_t* hist, int tid) {
  const int bin = tid >> 2, part = tid & 3;
  const uint4* r4 = reinterpret_cast<const uint4*>(X + bin * 32 + part * 8);
  const uint4 a = r4[0], b = r4[1];
  uint32_t s = (a.x & 0xffffu) + (a.x >> 16) + (a.y & 0xffffu) + (a.y >> 16) + (a.z & 0xffffu) + (a.z >> 16) +
               (a.w & 0xffffu) + (a.w >> 16) + (b.x & 0xffffu) + (b.x >> 16) + (b.y & 0xffffu) + (b.y >> 16) +
               (b.z & 0xffffu) + (b.z >> 16) + (b.w & 0xffffu) + (b.w >> 16);
  s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  if (part == 0) hist[bin] = s;
}

// bin b with  above(b) < need <= above(b) + hist[b],  above(b) = sum_{b' > b} hist[b'].
// One wavefront: lane l owns bins 4l..4l+3; suffix sums from a DPP prefix scan.
__device__ __forceinline__ void find_bin(const uint32_t* hist, uint32_t need, int* out_bin, int* out_above, int tid) {
  if (tid < 64) {
    const uint4 h = reinterpret_cast<const uint4*>(hist)[tid];
    const uint32_t hv[4] = {h.x, h.y, h.z, h.w};
    const uint32_t own = h.x + h.y + h.z + h.w;
    const uint32_t incl = wave_incl_scan_u32(own);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t above = total - incl;               // bins of higher lanes
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      if (above < need && need <= above + hv[i]) { *out_bin = tid * 4 + i; *out_above = (int)above; }
      above += hv[i];
    }
  }
}

__device__ __forceinline__ void select_bin(const uint32_t* X, uint32_t* hist, uint32_t need, int* out_bin, int* out_above, int tid) {
  reduce_counters(X, hist, tid);
  __syncthreads();
  find_bin(hist, need, out_bin, out_above, tid);
}

}  // namespace pkv
