// pkv_merge.hip — LOOK-M pivot merge behind every dense policy (gfx950).
//
//   reference pyramidkv_utils.py:119-170  merge_kv(key_states, value_states, indices, window_size, "pivot"),
//   called from :242,:274 (PyramidKV), :338 (SnapKV), :566 (H2O), :611 (StreamingLLM) when merge is set.
//
// The reference's behaviour, quirks included, is the specification (oracle/pkv_oracle.py: merge_kv restates it op for op,
// merge_kv_explicit spells the arithmetic out; both are pinned to the real reference by tests/golden/*_merge.npz):
//   * a position is DROPPED iff no (batch, head) selected it - torch.isin over the flattened indices of all heads (:131) -
//     and the observation-window positions count as dropped too (arange(k_len), :128): they merge onto themselves;
//   * kept keys are ordered [window, selected] (:146), kept values [selected, window] (:148), and the value merge (:162)
//     uses the KEY order's pivot numbers on the VALUE order's rows;
//   * pivot(i) = FIRST maximum over the kept keys of dtype(dot(dtype(x_i/|x_i|), dtype(t_j/|t_j|))) (:150-151);
//   * out_j = dtype( dtype(t_j + sum_i dtype(dtype(x_i + t_pivot)/2)  [fp32, ascending i]) / dtype(1 + n_j) )  (:158-162):
//     ATen's scatter_reduce(mean, include_self) accumulates in fp32, rounds the SUM, then divides by a model-dtype count.
//
// Kernels: mark (union bitmap) -> droplist (ordered compaction) -> targets (unit-norm kept keys) -> pivot (MFMA cosine
// similarity + first-maximum argmax) -> scatter (per kept row: ordered walk over the dropped rows that chose it).
// Roofline: the dominant traffic is every DROPPED key and value row read once by pivot (K) and once by scatter (K, V):
// 3 * (S - |union|) * D * e per head; the contraction is 2*(S-|union|)*(k+w)*D flop per head (MFMA, far from bound).
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"
#include "pkv_mfma.hpp"

#include <algorithm>

namespace pkv {

// ---- 1. union bitmap: mask[p] = 1 iff some (b,h) selected p ----
__global__ __launch_bounds__(256) void merge_mark_kernel(MergeParams p) {
  const int64_t total = (int64_t)p.B * p.H * p.k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / p.k;
    const int j = (int)(i - row * p.k);
    const int s = p.idx[row * p.idx_stride + j];
    if (s >= 0 && s < p.S) p.mask[s] = 1;
  }
}

// ---- 2. dropped positions in ascending order (ordered compaction by one workgroup) ----
__global__ __launch_bounds__(1024) void merge_droplist_kernel(MergeParams p) {
  __shared__ uint32_t wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t run = 0;
  for (int s0 = 0; s0 < p.S; s0 += 1024) {
    const int s = s0 + tid;
    const bool drop = s < p.S && p.mask[s] == 0;
    const uint64_t bal = __ballot(drop);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t lower = 0, all = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { const uint32_t c = wtot[w2]; all += c; lower += w2 < wave ? c : 0u; }
    if (drop) p.drop[run + lower + before] = s;
    run += all;
    __syncthreads();
  }
  if (tid == 0) *p.ndrop = (int32_t)run;
}

// kept row j of the KEY order [window, selected] (:146): source position in K
__device__ __forceinline__ int key_target_pos(const MergeParams& p, const int32_t* idx_row, int j) {
  return j < p.w ? p.S - p.w + j : idx_row[j - p.w];
}
// kept row j of the VALUE order [selected, window] (:148): source position in V
__device__ __forceinline__ int val_target_pos(const MergeParams& p, const int32_t* idx_row, int j) {
  return j < p.k ? idx_row[j] : p.S - p.w + (j - p.k);
}

// ---- 3. unit-norm kept keys: tn[bh][j][:] = dtype(t_j / dtype(|t_j|)) ----
template <typename T>
__global__ __launch_bounds__(256) void merge_targets_kernel(MergeParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x * 4 + wave;
  if (j >= p.k + p.w) return;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int pos = key_target_pos(p, idx_row, j);
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h + (int64_t)pos * p.ks_s;
  const uint32_t two = reinterpret_cast<const uint32_t*>(src)[lane];                 // 2 of the 128 elements per lane
  const float x0 = Elem<T>::to_f32((uint16_t)(two & 0xffffu)), x1 = Elem<T>::to_f32((uint16_t)(two >> 16));
  const float n2 = wave_sum(x0 * x0 + x1 * x1);
  const float n = Elem<T>::to_f32(Elem<T>::from_f32(sqrtf(n2)));                    // torch.norm -> model dtype
  const uint32_t out = round_pack2<T>(x0 / n, x1 / n);                               // k / norm -> model dtype
  reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p.tn) + ((int64_t)bh * p.ntp + j) * 128)[lane] = out;
}

// ---- 4. pivot: for every dropped row the kept row with the largest cosine similarity (first maximum) ----
// One wave = 16 dropped rows (MFMA A operand, unit-normalised in registers); the kept rows stream through LDS in tiles
// of MP_TN targets as B operands; a workgroup walks MP_ITER row groups against the staged tile before the next tile.
constexpr int MP_TN = 144;                // kept rows per LDS tile (39 KB): budget 128 + window 8 in ONE tile
constexpr int MP_ROWS = 64;               // dropped rows per workgroup pass (4 waves x 16)
constexpr int MP_ITER = 4;                // passes per workgroup: 256 dropped rows per workgroup
constexpr int MP_TROW = 136;              // LDS row stride in elements (272 B: the 16 rows of a B fragment hit different banks)

template <typename T>
__global__ __launch_bounds__(256) void merge_pivot_kernel(MergeParams p) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[MP_TN * MP_TROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int n = *p.ndrop;
  const int row_wg = blockIdx.x * MP_ROWS * MP_ITER;
  if (row_wg >= n) return;
  const int nt = p.k + p.w;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const uint16_t* tn = reinterpret_cast<const uint16_t*>(p.tn) + (int64_t)bh * p.ntp * 128;

  // A operands of this wave's MP_ITER row groups: row li of group it = dropped row row_wg + it*64 + wave*16 + li.
  // All 16 row loads of a lane are issued before any of them is used (one round trip instead of four).
  u32x4 af[MP_ITER][4];
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
    int r = row_wg + it * MP_ROWS + wave * 16 + li;
    r = r < n ? r : n - 1;                                               // clamp: rows past n are never written
    const uint16_t* row = kbase + (int64_t)p.drop[r] * p.ks_s + lg * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) af[it][kk] = *reinterpret_cast<const u32x4*>(row + kk * 32);
  }
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
    float xs[32];
    float n2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      U4 u;
      u.v = make_uint4(af[it][kk].x, af[it][kk].y, af[it][kk].z, af[it][kk].w);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xs[kk * 8 + e] = Elem<T>::to_f32(u.h[e]); n2 += xs[kk * 8 + e] * xs[kk * 8 + e]; }
    }
    n2 += __shfl_xor(n2, 16, 64);                                         // the row's 128 elements sit in 4 lanes (lg)
    n2 += __shfl_xor(n2, 32, 64);
    const float nr = Elem<T>::to_f32(Elem<T>::from_f32(sqrtf(n2)));
    const float rn = 1.0f / nr;                                           // one division per row; div_const = correctly rounded x / nr
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 a;
      a.x = round_pack2<T>(div_const(xs[kk * 8 + 0], nr, rn), div_const(xs[kk * 8 + 1], nr, rn));
      a.y = round_pack2<T>(div_const(xs[kk * 8 + 2], nr, rn), div_const(xs[kk * 8 + 3], nr, rn));
      a.z = round_pack2<T>(div_const(xs[kk * 8 + 4], nr, rn), div_const(xs[kk * 8 + 5], nr, rn));
      a.w = round_pack2<T>(div_const(xs[kk * 8 + 6], nr, rn), div_const(xs[kk * 8 + 7], nr, rn));
      af[it][kk] = a;
    }
  }
  // running best per (row group, accumulator register): lane holds rows 4*lg + r, column li of every 16-target tile
  float bestv[MP_ITER][4];
  int besti[MP_ITER][4];
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bestv[it][r] = -INFINITY; besti[it][r] = 0x7fffffff; }

  for (int t0 = 0; t0 < nt; t0 += MP_TN) {
    const int tcnt = min(MP_TN, nt - t0);
    __syncthreads();
    for (int c = tid; c < MP_TN * 16; c += 256) {                         // 16-B chunks; rows past tcnt are zero
      const int rr = c >> 4;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (rr < tcnt) val = reinterpret_cast<const uint4*>(tn + (int64_t)(t0 + rr) * 128)[c & 15];
      *reinterpret_cast<uint4*>(tile + rr * MP_TROW + (c & 15) * 8) = val;
    }
    __syncthreads();
    for (int n16 = 0; n16 * 16 < tcnt; ++n16) {
      u32x4 bf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bf[kk] = *reinterpret_cast<const u32x4*>(tile + (n16 * 16 + li) * MP_TROW + kk * 32 + lg * 8);
      const int col = t0 + n16 * 16 + li;
#pragma unroll
      for (int it = 0; it < MP_ITER; ++it) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(af[it][kk], bf[kk], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = Elem<T>::to_f32(Elem<T>::from_f32(acc[r]));    // similarity in the model dtype (:150)
          // strict >: first maximum.  A NaN similarity (zero-norm row: 0/0 at :146) ranks above everything and the FIRST
          // NaN wins, as torch.max propagates it (:151)
          const float bv = bestv[it][r];
          if (col < nt && (sv > bv || (sv != sv && bv == bv))) { bestv[it][r] = sv; besti[it][r] = col; }
        }
      }
    }
  }
  // first maximum across the 16 columns (lanes of one lg group): larger value, ties to the smaller kept-row number
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = bestv[it][r];
      int ix = besti[it][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float v2 = __shfl_xor(v, o, 64);
        const int i2 = __shfl_xor(ix, o, 64);
        const bool n1 = v != v, n2 = v2 != v2;
        if ((n2 && !n1) || (!n1 && !n2 && v2 > v) || ((n1 == n2) && (n1 || v2 == v) && i2 < ix)) { v = v2; ix = i2; }
      }
      const int row = row_wg + it * MP_ROWS + wave * 16 + lg * 4 + r;
      if (li == 0 && row < n) p.pivot[(int64_t)bh * p.S + row] = ix;
    }
  }
}

// ---- 5. scatter-mean: kept row j collects, in ascending order, the dropped rows that chose it ----
// Workgroup = one kept row of one (b,h): threads 0..127 = the 128 elements of the KEY row, 128..255 of the VALUE row.
// The ordered source list is built without barriers: wave w scans the w-th quarter of a chunk of MS_CHUNK pivots (8 loads in
// flight per lane, ballot + mbcnt compaction into its own LDS segment); the four segments in wave order are the ascending
// list.  The walk keeps two batches of 16 row loads in flight (the next batch is issued before the current one is consumed):
// the fp32 accumulation order is the list order, whatever the latency of a row.
constexpr int MS_CHUNK = 8192;            // pivots per pass (2048 per wave)
constexpr int MS_SEG = MS_CHUNK / 4;      // LDS list segment per wave
constexpr int MS_B = 32;                  // rows per batch of the walk (all in flight together)

template <typename T>
__global__ __launch_bounds__(256) void merge_scatter_kernel(MergeParams p) {
  __shared__ int32_t list[MS_CHUNK];
  __shared__ uint32_t wcount[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = tid & 127, is_v = tid >> 7;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x;
  const int n = *p.ndrop;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int32_t* piv = p.pivot + (int64_t)bh * p.S;
  const uint16_t* base = is_v ? reinterpret_cast<const uint16_t*>(p.vptr) + (int64_t)b * p.vs_b + (int64_t)hk * p.vs_h
                              : reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int64_t sstride = is_v ? p.vs_s : p.ks_s;
  const int tpos = is_v ? val_target_pos(p, idx_row, j) : key_target_pos(p, idx_row, j);
  const float t = Elem<T>::to_f32(base[(int64_t)tpos * sstride + d]);     // the kept row BEFORE merging (gather at :157/:160)
  float acc = t;                                                          // include_self
  int cnt = 1;
  for (int c0 = 0; c0 < n; c0 += MS_CHUNK) {
    __syncthreads();                                                      // the previous chunk's list has been walked
    // wave-private scan of [c0 + wave*MS_SEG, +MS_SEG): 8 pivot loads in flight per lane, no barrier inside
    uint32_t run = 0;
    const int w0 = c0 + wave * MS_SEG;
    for (int s0 = w0; s0 < min(n, w0 + MS_SEG); s0 += 8 * 64) {
      int pv[8], dr[8];      // pivots AND drop positions up front: a drop[] load behind a hit would stall every hit for a round trip
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = s0 + u * 64 + lane; pv[u] = piv[i < n ? i : n - 1]; dr[u] = p.drop[i < n ? i : n - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = s0 + u * 64 + lane;
        const bool hit = i < n && i < w0 + MS_SEG && pv[u] == j;
        const uint64_t bal = __ballot(hit);
        if (bal != 0ull) {
          const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          if (hit) list[wave * MS_SEG + run + before] = dr[u];
          run += (uint32_t)__popcll(bal);
        }
      }
    }
    if (lane == 0) wcount[wave] = run;
    __syncthreads();
    // ascending walk over the four segments as ONE list (entry l lives in segment sg at l - first[sg]); batches of MS_B rows,
    // the next batch in flight while the current one is consumed: one exposed row latency per pass, not one per segment
    const uint32_t f1 = wcount[0], f2 = f1 + wcount[1], f3 = f2 + wcount[2], m = f3 + wcount[3];
    auto entry = [&](uint32_t l) -> int32_t {
      l = l < m ? l : (m ? m - 1 : 0u);
      const uint32_t sg = (l >= f1) + (l >= f2) + (l >= f3);
      const uint32_t first = sg == 0 ? 0u : (sg == 1 ? f1 : (sg == 2 ? f2 : f3));
      return list[sg * MS_SEG + (l - first)];
    };
    // Every lane would otherwise repeat the (wave-uniform) list look-up and 64-bit address arithmetic for every row: ~15
    // vector instructions per row and lane made this walk VALU-bound.  Instead each lane looks up ONE entry of a block of 64
    // rows; the row numbers then reach the scalar unit through v_readlane and the loads use a scalar base.
    for (uint32_t l64 = 0; l64 < m; l64 += 64) {
      const int myrow = entry(l64 + (uint32_t)lane);
      const uint32_t left = m - l64 < 64u ? m - l64 : 64u;
      for (uint32_t b0 = 0; b0 < left; b0 += MS_B) {
        uint16_t x[MS_B];
#pragma unroll
        for (int u = 0; u < MS_B; ++u) {
          const int r = __builtin_amdgcn_readlane(myrow, (int)(b0 + u < 64u ? b0 + u : 63u));   // rows past `left` repeat the last entry (masked below)
          x[u] = base[(int64_t)r * sstride + d];
        }
#pragma unroll
        for (int u = 0; u < MS_B; ++u) {
          if (b0 + u < left) {
            const float half = Elem<T>::to_f32(Elem<T>::from_f32(Elem<T>::to_f32(x[u]) + t)) * 0.5f;   // (x + t) -> dtype, / 2 (:158)
            acc = __fadd_rn(acc, Elem<T>::to_f32(Elem<T>::from_f32(half)));
          }
        }
      }
    }
    cnt += (int)m;
  }
  const float sum_q = Elem<T>::to_f32(Elem<T>::from_f32(acc));            // the scattered SUM in the model dtype
  const float cnt_q = Elem<T>::to_f32(Elem<T>::from_f32((float)cnt));     // the count in the model dtype (rounds above 256 / 2048)
  uint16_t* out = reinterpret_cast<uint16_t*>(is_v ? p.v_out : p.k_out) + ((int64_t)bh * (p.k + p.w) + j) * 128 + d;
  *out = Elem<T>::from_f32(sum_q / cnt_q);
}

hipError_t launch_merge(int dtype, const MergeParams& p, hipStream_t st) {
  hipError_t e = hipMemsetAsync(p.mask, 0, (size_t)p.S, st);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)p.B * p.H * p.k;
  const int mb = (int)std::min<int64_t>((total + 255) / 256, 1024);
  hipLaunchKernelGGL(merge_mark_kernel, dim3(mb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(merge_droplist_kernel, dim3(1), dim3(1024), 0, st, p);
  const int nt = p.k + p.w;
  const dim3 gt((nt + 3) / 4, p.B * p.H), gp((p.S + MP_ROWS * MP_ITER - 1) / (MP_ROWS * MP_ITER), p.B * p.H), gs(nt, p.B * p.H);
  if (dtype == 0) {
    hipLaunchKernelGGL(merge_targets_kernel<BF16>, gt, dim3(256), 0, st, p);
    hipLaunchKernelGGL(merge_pivot_kernel<BF16>, gp, dim3(256), 0, st, p);
    hipLaunchKernelGGL(merge_scatter_kernel<BF16>, gs, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(merge_targets_kernel<F16>, gt, dim3(256), 0, st, p);
    hipLaunchKernelGGL(merge_pivot_kernel<F16>, gp, dim3(256), 0, st, p);
    hipLaunchKernelGGL(merge_scatter_kernel<F16>, gs, dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

}  // namespace pkv
