// pkv_ada.hip — Ada-SnapKV head budgets (gfx950).
//
//   reference pyramidkv_utils.py:706-719:
//     sorted scores per head -> optional normalisation (x sum(top base)/sum(all), all in model dtype)
//     -> flatten [H*L] -> topk(H*base) -> head id = idx // L -> per-head counts
//     -> cap_h = round(count_h * (1 - floor) + int(base*floor))            (fp32, half-to-even)
//
// The flattened top-(H*base) is never materialised: every head's adaptive scores are already sorted
// (multiplying a non-increasing sequence by a positive ratio and rounding keeps it non-increasing), so
// "how many entries of head h are >= x" is a binary search, and the global threshold is an exact
// two-level (8+8 bit) radix select over those counts.  Ties at the threshold go to the lowest
// flattened index (head-major), i.e. a stable descending sort of the flattened tensor.
// Integer result; the only floating point is the normalisation ratio and the final fp32 rounding.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

struct AdaWs {           // layout of the workspace handed to pkv_ada_budget
  float* ratio;          // [H]
  int32_t* cum_hi;       // [H][256]  #entries of head h with adaptive key >= (b<<8)
  int32_t* cum_lo;       // [H][256]  #entries with key >= (b1<<8 | c)
};

template <typename T>
__device__ __forceinline__ uint32_t adaptive_key(const uint16_t* v, int i, float ratio, int normalize) {
  uint16_t h = v[i];
  if (normalize) h = Elem<T>::from_f32(Elem<T>::to_f32(h) * ratio);   // adaptive_attn_score*ratio_weight (:711)
  return order_key<T>(h);
}

// number of leading entries with key >= x (sequence is non-increasing)
template <typename T>
__device__ __forceinline__ int count_ge(const uint16_t* v, int L, float ratio, int normalize, uint32_t x) {
  int lo = 0, hi = L;   // invariant: entries [0,lo) >= x, entries [hi,L) < x
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (adaptive_key<T>(v, mid, ratio, normalize) >= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// the head's sorted scores (<= 64 KB) are staged in LDS: the 256 binary searches then cost ~15 LDS
// round trips instead of 15 global-memory round trips
__device__ __forceinline__ const uint16_t* stage_row(const uint16_t* g, int L, uint16_t* lds, int tid) {
  for (int i = tid; i < L; i += 256) lds[i] = g[i];
  __syncthreads();
  return lds;
}

template <typename T>
__global__ __launch_bounds__(256) void ada_stats_kernel(BudgetParams p, AdaWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ada_smem[];
  __shared__ double red[2][4];
  __shared__ float s_ratio;
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* v = stage_row(reinterpret_cast<const uint16_t*>(p.sorted_val) + (int64_t)h * p.L, p.L,
                                reinterpret_cast<uint16_t*>(ada_smem), tid);
  float ratio = 1.0f;
  if (p.normalize) {
    double st = 0.0, sa = 0.0;
    for (int i = tid; i < p.L; i += 256) {
      const double x = (double)Elem<T>::to_f32(v[i]);
      sa += x;
      if (i < p.base) st += x;
    }
    for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o, 64); sa += __shfl_xor(sa, o, 64); }
    if (lane == 0) { red[0][wave] = st; red[1][wave] = sa; }
    __syncthreads();
    if (tid == 0) {
      const double t = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      const double a = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      const float tq = Elem<T>::to_f32(Elem<T>::from_f32((float)t));   // .sum() result in model dtype (:710)
      const float aq = Elem<T>::to_f32(Elem<T>::from_f32((float)a));
      s_ratio = Elem<T>::to_f32(Elem<T>::from_f32(tq / aq));           // model-dtype division (:710)
      ws.ratio[h] = s_ratio;
    }
    __syncthreads();
    ratio = s_ratio;
  } else if (tid == 0) {
    ws.ratio[h] = 1.0f;
  }
  ws.cum_hi[h * 256 + tid] = count_ge<T>(v, p.L, ratio, p.normalize, (uint32_t)tid << 8);
}

// largest b with  sum_h cum[h][b] >= total   (sums are non-increasing in b, sum at b=0 >= total)
__device__ __forceinline__ int find_level(const int32_t* cum, int H, int64_t total, int64_t* s_sum, int* s_b, int tid) {
  int64_t s = 0;
  for (int h = 0; h < H; ++h) s += cum[h * 256 + tid];
  s_sum[tid] = s;
  __syncthreads();
  if (s >= total && (tid == 255 || s_sum[tid + 1] < total)) *s_b = tid;
  __syncthreads();
  return *s_b;
}

template <typename T>
__global__ __launch_bounds__(256) void ada_lo_kernel(BudgetParams p, AdaWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ada_smem[];
  __shared__ int64_t s_sum[256];
  __shared__ int s_b;
  const int h = blockIdx.x, tid = threadIdx.x;
  const int64_t total = (int64_t)p.H * p.base;
  const uint16_t* v = stage_row(reinterpret_cast<const uint16_t*>(p.sorted_val) + (int64_t)h * p.L, p.L,
                                reinterpret_cast<uint16_t*>(ada_smem), tid);
  const int b1 = find_level(ws.cum_hi, p.H, total, s_sum, &s_b, tid);
  ws.cum_lo[h * 256 + tid] = count_ge<T>(v, p.L, ws.ratio[h], p.normalize, ((uint32_t)b1 << 8) | (uint32_t)tid);
}

__global__ __launch_bounds__(256) void ada_final_kernel(BudgetParams p, AdaWs ws, float one_minus_floor) {
  __shared__ int64_t s_sum[256];
  __shared__ int s_b;
  const int tid = threadIdx.x;
  const int64_t total = (int64_t)p.H * p.base;
  const int b1 = find_level(ws.cum_hi, p.H, total, s_sum, &s_b, tid);
  const int b2 = find_level(ws.cum_lo, p.H, total, s_sum, &s_b, tid);
  if (tid == 0) {
    int64_t n_gt = 0;
    for (int h = 0; h < p.H; ++h) {
      const int gt = b2 < 255 ? ws.cum_lo[h * 256 + b2 + 1] : (b1 < 255 ? ws.cum_hi[h * 256 + b1 + 1] : 0);
      n_gt += gt;
    }
    int64_t need = total - n_gt;     // ties at the threshold, handed out in flattened (head-major) order
    for (int h = 0; h < p.H; ++h) {
      const int gt = b2 < 255 ? ws.cum_lo[h * 256 + b2 + 1] : (b1 < 255 ? ws.cum_hi[h * 256 + b1 + 1] : 0);
      const int eq = ws.cum_lo[h * 256 + b2] - gt;
      const int take = (int)(need < eq ? need : eq);
      need -= take;
      const float cnt = (float)(gt + take);
      const float cap = __fadd_rn(__fmul_rn(cnt, one_minus_floor), (float)p.floor_capacity);   // :719, fp32
      p.head_capacity[h] = (int)rintf(cap);                                                    // torch.round: half to even
    }
  }
}

__global__ void ada_metadata_kernel(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    for (int h = 0; h < H; ++h) {
      const int n = cap[h] + w;
      head_lens[h] = n;       // :684
      cu_klen[h] = run;       // :689 exclusive prefix
      run += n;
    }
    cu_klen[H] = run;         // :690-691 total
  }
}

hipError_t launch_budget(int dtype, const BudgetParams& p, hipStream_t st) {
  AdaWs ws;
  char* base = reinterpret_cast<char*>(p.ws);
  ws.ratio = reinterpret_cast<float*>(base);
  ws.cum_hi = reinterpret_cast<int32_t*>(base + 1024);
  ws.cum_lo = ws.cum_hi + (size_t)p.H * 256;
  const float omf = p.one_minus_floor;
  const size_t lds = ((size_t)p.L * 2 + 15) & ~(size_t)15;
  auto k_stats = dtype == 0 ? ada_stats_kernel<BF16> : ada_stats_kernel<F16>;
  auto k_lo = dtype == 0 ? ada_lo_kernel<BF16> : ada_lo_kernel<F16>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stats), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_stats, dim3(p.H), dim3(256), lds, st, p, ws);
  hipLaunchKernelGGL(k_lo, dim3(p.H), dim3(256), lds, st, p, ws);
  hipLaunchKernelGGL(ada_final_kernel, dim3(1), dim3(256), 0, st, p, ws, omf);
  return hipGetLastError();
}

hipError_t launch_ada_metadata(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen, hipStream_t st) {
  hipLaunchKernelGGL(ada_metadata_kernel, dim3(1), dim3(64), 0, st, H, w, cap, head_lens, cu_klen);
  return hipGetLastError();
}

}  // namespace pkv
