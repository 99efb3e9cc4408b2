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
//
// The sorted rows need not be complete: one head can take at most H*base entries of the global top-(H*base), so the
// first M = min(L, H*base) entries of every head's descending order decide everything (counts are clamped at M, which
// changes neither the threshold nor any head's share - see DESIGN.md).  The host therefore hands over the TOP-M indices
// of every head (pkv_topk, no full sort) plus the un-sorted score rows: the sorted values are looked up while the list
// is staged in LDS, and the sum over ALL scores of the row (:710) is taken from the row itself.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

struct AdaWs {           // layout of the workspace handed to pkv_ada_budget
  float* ratio;          // [H]
  int32_t* cum_hi;       // [H][256]  #entries of head h with adaptive key >= (b<<8)
  int32_t* cum_lo;       // [H][256]  #entries with key >= (b1<<8 | c)
  uint16_t* list;        // [H][Lpad] sorted values of every head as staged by the first kernel (null: re-stage from the inputs)
  int Lpad;
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
__device__ __forceinline__ const uint16_t* stage_row(const BudgetParams& p, int h, uint16_t* lds, int tid) {
  if (p.sorted_idx) {     // top-M index list + un-sorted scores: sorted value i = scores[h][idx[h][i]]
    const int32_t* ix = p.sorted_idx + (int64_t)h * p.idx_stride;
    const uint16_t* row = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)h * p.scores_stride;
    // 8 index loads, then 8 dependent score loads, in flight together (clamped, masked): two round trips per 2048
    // entries instead of two per 256
    for (int i0 = 0; i0 < p.L; i0 += 8 * 256) {
      int id[8];
      uint16_t val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = i0 + j * 256 + tid; id[j] = ix[i < p.L ? i : p.L - 1]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = row[id[j]];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = i0 + j * 256 + tid; if (i < p.L) lds[i] = val[j]; }
    }
  } else {
    const uint16_t* g = reinterpret_cast<const uint16_t*>(p.sorted_val) + (int64_t)h * p.L;
    for (int i = tid; i < p.L; i += 256) lds[i] = g[i];
  }
  __syncthreads();
  return lds;
}

template <typename T>
__global__ __launch_bounds__(256) void ada_stats_kernel(BudgetParams p, AdaWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ada_smem[];
  __shared__ double red[2][4];
  __shared__ float s_ratio;
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* v = stage_row(p, h, reinterpret_cast<uint16_t*>(ada_smem), tid);
  if (ws.list && p.sorted_idx) {        // hand the looked-up list to ada_lo_kernel as one contiguous row (16-B stores)
    uint4* dst = reinterpret_cast<uint4*>(ws.list + (int64_t)h * ws.Lpad);
    const uint4* src = reinterpret_cast<const uint4*>(v);
    for (int c = tid; c < (ws.Lpad >> 3); c += 256) dst[c] = src[c];      // LDS is padded to Lpad by the launch
  }
  float ratio = 1.0f;
  if (p.normalize) {
    double st = 0.0, sa = 0.0;
    for (int i = tid; i < p.L; i += 256) {
      const double x = (double)Elem<T>::to_f32(v[i]);
      sa += x;
      if (i < p.base) st += x;
    }
    if (p.sorted_idx) {   // the list is the top M only: the sum over ALL scores (:710) comes from the row itself
      sa = 0.0;
      const uint16_t* row = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)h * p.scores_stride;
      int i0 = 0;
      if (((reinterpret_cast<uintptr_t>(row) & 15) == 0)) {           // 16-B loads, 8 scores each, all of a lane's loads in flight
        const int nv = p.Lrow >> 3;
        for (int c0 = 0; c0 < nv; c0 += 16 * 256) {                     // 16 loads in flight per lane (one round trip at S = 32k)
          U4 u[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) { const int c = c0 + j * 256 + tid; u[j].v = reinterpret_cast<const uint4*>(row)[c < nv ? c : 0]; }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if (c0 + j * 256 + tid < nv) {
#pragma unroll
              for (int e = 0; e < 8; ++e) sa += (double)Elem<T>::to_f32(u[j].h[e]);
            }
          }
        }
        i0 = nv << 3;
      }
      for (int i = i0 + tid; i < p.Lrow; i += 256) sa += (double)Elem<T>::to_f32(row[i]);
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
  if (p.adaptive_out) {   // head-sharded Ada-SnapKV: only the head's ADAPTIVE list (:711) is wanted - it is what the ranks exchange
    uint16_t* out = reinterpret_cast<uint16_t*>(p.adaptive_out) + (int64_t)h * p.L;
    for (int i = tid; i < p.L; i += 256)
      out[i] = p.normalize ? Elem<T>::from_f32(Elem<T>::to_f32(v[i]) * ratio) : v[i];
    return;
  }
  ws.cum_hi[h * 256 + tid] = count_ge<T>(v, p.L, ratio, p.normalize, (uint32_t)tid << 8);
}

// largest b with  sum_h cum[h][b] >= total   (sums are non-increasing in b, sum at b=0 >= total)
__device__ __forceinline__ int find_level(const int32_t* cum, int H, int64_t total, int64_t* s_sum, int* s_b, int tid) {
  int64_t s = 0;
  for (int h0 = 0; h0 < H; h0 += 32) {      // 32 loads in flight (a plain loop waits for every load before the next one)
    int32_t v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = cum[(h0 + j < H ? h0 + j : H - 1) * 256 + tid];
#pragma unroll
    for (int j = 0; j < 32; ++j) s += (h0 + j < H) ? v[j] : 0;
  }
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
  const uint16_t* v;
  if (ws.list && p.sorted_idx) {        // the list as staged by ada_stats_kernel: contiguous, all loads in flight together
    uint16_t* lds = reinterpret_cast<uint16_t*>(ada_smem);
    const uint4* src = reinterpret_cast<const uint4*>(ws.list + (int64_t)h * ws.Lpad);
    const int nv = ws.Lpad >> 3;
    for (int c0 = 0; c0 < nv; c0 += 8 * 256) {
      uint4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c = c0 + j * 256 + tid; t[j] = src[c < nv ? c : 0]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c = c0 + j * 256 + tid; if (c < nv) reinterpret_cast<uint4*>(lds)[c] = t[j]; }
    }
    __syncthreads();
    v = lds;
  } else {
    v = stage_row(p, h, reinterpret_cast<uint16_t*>(ada_smem), tid);
  }
  const int b1 = find_level(ws.cum_hi, p.H, total, s_sum, &s_b, tid);
  ws.cum_lo[h * 256 + tid] = count_ge<T>(v, p.L, ws.ratio[h], p.normalize, ((uint32_t)b1 << 8) | (uint32_t)tid);
}

// Final step, one workgroup: thread h owns head h.  gt_h = entries above the global threshold, eq_h = entries equal to it;
// the ties are handed out in flattened (head-major) order: head h takes min(eq_h, need - ties taken by the heads before it),
// an exclusive prefix sum over the heads.  Optionally writes the var-len metadata of :682-691 as well (one launch less).
__global__ __launch_bounds__(256) void ada_final_kernel(BudgetParams p, AdaWs ws, float one_minus_floor, int window,
                                                        int32_t* head_lens, int32_t* cu_klen, int32_t* cu_headlens) {
  __shared__ int64_t s_sum[256];
  __shared__ int s_b;
  __shared__ int64_t s_red[4];
  __shared__ int s_scan[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)p.H * p.base;
  const int b1 = find_level(ws.cum_hi, p.H, total, s_sum, &s_b, tid);
  const int b2 = find_level(ws.cum_lo, p.H, total, s_sum, &s_b, tid);
  int gt = 0, eq = 0;
  if (tid < p.H) {
    gt = b2 < 255 ? ws.cum_lo[tid * 256 + b2 + 1] : (b1 < 255 ? ws.cum_hi[tid * 256 + b1 + 1] : 0);
    eq = ws.cum_lo[tid * 256 + b2] - gt;
  }
  // need = total - sum_h gt_h
  int64_t g = gt;
  for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
  if (lane == 0) s_red[wave] = g;
  // exclusive prefix of eq over the heads
  const uint32_t incl = wave_incl_scan_u32((uint32_t)eq);
  if (lane == 63) s_scan[wave] = (int)incl;
  __syncthreads();
  const int64_t need = total - (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  int64_t before = (int64_t)incl - eq;
  for (int w2 = 0; w2 < wave; ++w2) before += s_scan[w2];
  int cap = 0;
  if (tid < p.H) {
    const int64_t left = need - before;
    const int take = (int)(left <= 0 ? 0 : (left < eq ? left : eq));
    const float cnt = (float)(gt + take);
    const float capf = __fadd_rn(__fmul_rn(cnt, one_minus_floor), (float)p.floor_capacity);   // :719, fp32
    cap = (int)rintf(capf);                                                                     // torch.round: half to even
    p.head_capacity[tid] = cap;
  }
  if (p.host_mirror) {
    // The boundary exposes klen_sum / max_seqlen_k as Python ints (:685-686), i.e. the host needs the capacities (the
    // reference syncs for the same reason, :718).  They are written straight into pinned host memory, made visible
    // system-wide, then flagged: the host polls the flag instead of paying a memcpy + a stream synchronise.
    if (tid < p.H) {
      __hip_atomic_store(p.host_mirror + tid, cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.host_mirror + p.H, p.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (head_lens && cu_klen) {                     // :684, :689-691: head_lens = cap + w, cu_klen = exclusive prefix + total
    __syncthreads();
    const int n = tid < p.H ? cap + window : 0;
    const uint32_t in2 = wave_incl_scan_u32((uint32_t)n);
    if (lane == 63) s_scan[wave] = (int)in2;
    __syncthreads();
    int off = (int)in2 - n;
    for (int w2 = 0; w2 < wave; ++w2) off += s_scan[w2];
    if (tid < p.H) { head_lens[tid] = n; cu_klen[tid] = off; if (cu_headlens) cu_headlens[tid] = off + n; }
    if (tid == p.H - 1) cu_klen[p.H] = off + n;
  }
}

__global__ void ada_metadata_kernel(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen, int32_t* cu_headlens) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    for (int h = 0; h < H; ++h) {
      const int n = cap[h] + w;
      head_lens[h] = n;       // :684
      cu_klen[h] = run;       // :689 exclusive prefix
      run += n;
      if (cu_headlens) cu_headlens[h] = run;   // :687 inclusive prefix
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
  ws.Lpad = (int)(lds / 2);
  ws.list = p.list_ws ? reinterpret_cast<uint16_t*>(p.list_ws) : nullptr;       // [H][Lpad], 16-B aligned rows
  auto k_stats = dtype == 0 ? ada_stats_kernel<BF16> : ada_stats_kernel<F16>;
  auto k_lo = dtype == 0 ? ada_lo_kernel<BF16> : ada_lo_kernel<F16>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stats), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_stats, dim3(p.H), dim3(256), lds, st, p, ws);
  if (p.adaptive_out) return hipGetLastError();
  hipLaunchKernelGGL(k_lo, dim3(p.H), dim3(256), lds, st, p, ws);
  hipLaunchKernelGGL(ada_final_kernel, dim3(1), dim3(256), 0, st, p, ws, omf, p.window, p.head_lens_out, p.cu_klen_out,
                     p.cu_headlens_out);
  return hipGetLastError();
}

hipError_t launch_ada_metadata(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen, hipStream_t st, int32_t* cu_headlens) {
  hipLaunchKernelGGL(ada_metadata_kernel, dim3(1), dim3(64), 0, st, H, w, cap, head_lens, cu_klen, cu_headlens);
  return hipGetLastError();
}

}  // namespace pkv
