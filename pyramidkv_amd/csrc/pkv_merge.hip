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
// similarity + first-maximum argmax) -> bucket (per head: the dropped rows grouped by the kept row they chose, counting
// sort) -> scatter (per kept row: its group put in ascending order through an LDS bitmap, then the ordered walk).
// Head sizes 64 / 128 / 256: the kernels are templated on the MFMA k-steps KS = D / 32, like the H2O kernels.
// Roofline: the dominant traffic is every DROPPED key and value row read once by pivot (K) and once by scatter (K, V):
// 3 * (S - |union|) * D * e per head; the contraction is 2*(S-|union|)*(k+w)*D flop per head (MFMA, far from bound).
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"
#include "pkv_mfma.hpp"

#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace pkv {

static int merge_env(const char* name, int dflt);

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
// Thread t owns the positions [t*per, (t+1)*per), per a multiple of 16: it counts its un-selected positions from 16-byte
// loads of the mask, one block-wide exclusive scan places the threads, and a second walk over the same (cached) bytes writes
// the positions.  (The first version walked S in rounds of 1024 with two barriers each: 25 us at S = 32768.)
__global__ __launch_bounds__(1024) void merge_droplist_kernel(MergeParams p) {
  __shared__ uint32_t wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (((p.S + 1023) >> 10) + 15) & ~15;
  const int s0 = tid * per;
  auto zero_bytes = [&](int s, uint32_t (&wd)[4]) -> bool {               // the 16 mask bytes at s (positions >= S read as selected)
    if (s + 16 <= p.S) {
      const uint4 v = *reinterpret_cast<const uint4*>(p.mask + s);        // mask is 256-byte aligned, s a multiple of 16
      wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w;
      return true;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) { const int sp = s + q * 4 + bq; x |= (uint32_t)(sp < p.S ? p.mask[sp] : 1) << (8 * bq); }
      wd[q] = x;
    }
    return s < p.S;
  };
  uint32_t cnt = 0;
  for (int c = 0; c < per && s0 + c < p.S; c += 16) {
    uint32_t wd[4];
    zero_bytes(s0 + c, wd);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) cnt += ((wd[q] >> (8 * bq)) & 0xffu) == 0u ? 1u : 0u;
  }
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t lower = 0, all = 0;
#pragma unroll
  for (int w2 = 0; w2 < 16; ++w2) { const uint32_t c = wtot[w2]; all += c; lower += w2 < wave ? c : 0u; }
  uint32_t at = lower + incl - cnt;
  for (int c = 0; c < per && s0 + c < p.S; c += 16) {
    uint32_t wd[4];
    zero_bytes(s0 + c, wd);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int bq = 0; bq < 4; ++bq)
        if (((wd[q] >> (8 * bq)) & 0xffu) == 0u) p.drop[at++] = s0 + c + q * 4 + bq;
  }
  if (tid == 0) *p.ndrop = (int32_t)all;
}

// ---- 2b. the same list from several workgroups (round 6; the one-workgroup form above took 17 us at S = 32768) ----
// Workgroup g owns the positions [4096 g, 4096 (g + 1)): it first counts the dropped positions BELOW its slice (the mask
// is S bytes, L2-resident: at most S / 16 sixteen-byte loads spread over 256 threads), then compacts its own slice in order.
// A mask byte is 0 or 1, so a dword holds 4 - popcount dropped positions.  The last workgroup writes the total.
constexpr int MD_SLICE = 4096;
__global__ __launch_bounds__(256) void merge_droplist2_kernel(MergeParams p) {
  __shared__ uint32_t wsum[4], wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s_begin = blockIdx.x * MD_SLICE;
  uint32_t below = 0;
  for (int s = tid * 16; s < s_begin; s += 256 * 16) {                       // s + 16 <= s_begin <= S: whole 16-byte words
    const uint4 v = *reinterpret_cast<const uint4*>(p.mask + s);
    below += 16u - (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w));
  }
  below = wave_sum_u32(below);
  if (lane == 0) wsum[wave] = below;
  const int s0 = s_begin + tid * 16;
  uint32_t wd[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};     // positions >= S read as selected
  if (s0 + 16 <= p.S) {
    const uint4 v = *reinterpret_cast<const uint4*>(p.mask + s0);
    wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w;
  } else if (s0 < p.S) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) { const int sp = s0 + q * 4 + bq; x |= (uint32_t)(sp < p.S ? p.mask[sp] : 1) << (8 * bq); }
      wd[q] = x;
    }
  }
  const uint32_t cnt = 16u - (uint32_t)(__popc(wd[0] & 0x01010101u) + __popc(wd[1] & 0x01010101u) + __popc(wd[2] & 0x01010101u) + __popc(wd[3] & 0x01010101u));
  const uint32_t incl = wave_incl_scan_u32(cnt);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t lower = wsum[0] + wsum[1] + wsum[2] + wsum[3], all = 0;
#pragma unroll
  for (int w2 = 0; w2 < 4; ++w2) { const uint32_t c = wtot[w2]; all += c; lower += w2 < wave ? c : 0u; }
  uint32_t at = lower + incl - cnt;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
      if (((wd[q] >> (8 * bq)) & 1u) == 0u) p.drop[at++] = s0 + q * 4 + bq;
  if (blockIdx.x == gridDim.x - 1 && tid == 0) *p.ndrop = (int32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3] + all);
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
template <typename T, int KS>
__global__ __launch_bounds__(256) void merge_targets_kernel(MergeParams p) {
  constexpr int D = KS * 32, EPL = D / 64;                                         // elements per lane: 1, 2 or 4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x * 4 + wave;
  if (j >= p.k + p.w) return;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int pos = key_target_pos(p, idx_row, j);
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h + (int64_t)pos * p.ks_s + lane * EPL;
  uint16_t raw[EPL];
  if constexpr (EPL == 1) raw[0] = src[0];
  else if constexpr (EPL == 2) { const uint32_t two = *reinterpret_cast<const uint32_t*>(src); raw[0] = (uint16_t)(two & 0xffffu); raw[1] = (uint16_t)(two >> 16); }
  else { const uint2 four = *reinterpret_cast<const uint2*>(src); raw[0] = (uint16_t)(four.x & 0xffffu); raw[1] = (uint16_t)(four.x >> 16); raw[2] = (uint16_t)(four.y & 0xffffu); raw[3] = (uint16_t)(four.y >> 16); }
  float x[EPL], sq = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { x[e] = Elem<T>::to_f32(raw[e]); sq += x[e] * x[e]; }
  const float n2 = wave_sum(sq);
  const float n = Elem<T>::to_f32(Elem<T>::from_f32(sqrtf(n2)));                    // torch.norm -> model dtype
  // a kept row whose unit-norm form holds NaN / inf (zero norm, non-finite key): the pivot kernel then canonicalises NaN
  // similarities in its inner loop (torch.max's rules); otherwise every similarity of the head is finite and it skips that
  if (lane == 0 && (!(n2 < INFINITY) || !(n > 0.f))) atomicOr(p.kept_bad + bh, 1);
  uint16_t* dst = reinterpret_cast<uint16_t*>(p.tn) + ((int64_t)bh * p.ntp + j) * D + lane * EPL;
  if constexpr (EPL == 1) dst[0] = Elem<T>::from_f32(x[0] / n);                     // k / norm -> model dtype
  else if constexpr (EPL == 2) *reinterpret_cast<uint32_t*>(dst) = round_pack2<T>(x[0] / n, x[1] / n);
  else *reinterpret_cast<uint2*>(dst) = make_uint2(round_pack2<T>(x[0] / n, x[1] / n), round_pack2<T>(x[2] / n, x[3] / n));
}

// ---- 4. pivot: for every dropped row the kept row with the largest cosine similarity (first maximum) ----
// One wave = 16 dropped rows (MFMA A operand, unit-normalised in registers); the kept rows stream through LDS in tiles
// of MP_TN targets as B operands; a workgroup walks MP_ITER row groups against the staged tile before the next tile.
constexpr int MP_ROWS = 64;               // dropped rows per workgroup pass (4 waves x 16)
template <int KS> struct MergeShape {
  static constexpr int D = KS * 32;
  static constexpr int TN = KS == 8 ? 80 : 144;      // kept rows per LDS tile (<= 42 KB; a multiple of the 16 rows of a B fragment): budget 128 + window 8 in ONE tile at D <= 128
  static constexpr int ITER = KS == 8 ? 2 : 4;       // passes per workgroup (64 dropped rows each): A operands = ITER x KS x 4 registers
  static constexpr int TROW = D + 8;                 // LDS row stride in elements (+16 B: the 16 rows of a B fragment hit different banks)
};

template <typename T, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS == 8 ? 3 : 4))) void merge_pivot_kernel(MergeParams p) {
  using Sh = MergeShape<KS>;
  constexpr int D = Sh::D, MP_TN = Sh::TN, MP_ITER = Sh::ITER, MP_TROW = Sh::TROW, CPR = D / 8;   // CPR: 16-B chunks per row
  __shared__ __attribute__((aligned(16))) uint16_t tile[MP_TN * MP_TROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int n = *p.ndrop;
  const int row_wg = blockIdx.x * MP_ROWS * MP_ITER;
  if (row_wg >= n) return;
  const int nt = p.k + p.w;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const uint16_t* tn = reinterpret_cast<const uint16_t*>(p.tn) + (int64_t)bh * p.ntp * D;

  // A operands of this wave's MP_ITER row groups: row li of group it = dropped row row_wg + it*64 + wave*16 + li.
  // All row loads of a lane are issued before any of them is used (one round trip instead of MP_ITER).
  u32x4 af[MP_ITER][KS];
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
    int r = row_wg + it * MP_ROWS + wave * 16 + li;
    r = r < n ? r : n - 1;                                               // clamp: rows past n are never written
    const uint16_t* row = kbase + (int64_t)p.drop[r] * p.ks_s + lg * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) af[it][kk] = *reinterpret_cast<const u32x4*>(row + kk * 32);
  }
  bool bad_row = false;
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
    float xs[KS * 8];
    float n2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      U4 u;
      u.v = make_uint4(af[it][kk].x, af[it][kk].y, af[it][kk].z, af[it][kk].w);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xs[kk * 8 + e] = Elem<T>::to_f32(u.h[e]); n2 += xs[kk * 8 + e] * xs[kk * 8 + e]; }
    }
    n2 += __shfl_xor(n2, 16, 64);                                         // the row's D elements sit in 4 lanes (lg)
    n2 += __shfl_xor(n2, 32, 64);
    const float nr = Elem<T>::to_f32(Elem<T>::from_f32(sqrtf(n2)));
    bad_row |= !(n2 < INFINITY) || !(nr > 0.f);                           // the unit-norm row holds NaN / inf
    const float rn = 1.0f / nr;                                           // one division per row; div_const = correctly rounded x / nr
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      u32x4 a;
      a.x = round_pack2<T>(div_const(xs[kk * 8 + 0], nr, rn), div_const(xs[kk * 8 + 1], nr, rn));
      a.y = round_pack2<T>(div_const(xs[kk * 8 + 2], nr, rn), div_const(xs[kk * 8 + 3], nr, rn));
      a.z = round_pack2<T>(div_const(xs[kk * 8 + 4], nr, rn), div_const(xs[kk * 8 + 5], nr, rn));
      a.w = round_pack2<T>(div_const(xs[kk * 8 + 6], nr, rn), div_const(xs[kk * 8 + 7], nr, rn));
      af[it][kk] = a;
    }
  }
  // Running best per (row group, accumulator register): lane holds rows 4*lg + r, column li of every 16-target tile.
  // A similarity in the model dtype is 16 bits, so (order-preserving key of the value) << 16 | (0xffff - kept-row number)
  // is ONE unsigned word whose maximum is torch.max's answer (:151): the larger value, ties to the FIRST column; a NaN
  // similarity (zero-norm row: 0/0 at :146) gets the largest key, so it ranks above everything and the first NaN wins, as
  // torch.max propagates it.  One v_max_u32 per similarity instead of compare + two selects, half the registers.
  uint32_t bestk[MP_ITER][4];
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) bestk[it][r] = 0u;

  // NaN similarities can only come from a unit-norm row that is not finite: without one in reach (the usual case) the
  // NaN canonicalisation is skipped - the loop is bound by its vector instructions per similarity
  const bool exact = p.kept_bad[bh] != 0 || __ballot(bad_row) != 0ull;     // wave-uniform
  for (int t0 = 0; t0 < nt; t0 += MP_TN) {
    const int tcnt = min(MP_TN, nt - t0);
    __syncthreads();
    for (int c = tid; c < MP_TN * CPR; c += 256) {                        // 16-B chunks; rows past tcnt are zero
      const int rr = c / CPR, ch = c - rr * CPR;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (rr < tcnt) val = reinterpret_cast<const uint4*>(tn + (int64_t)(t0 + rr) * D)[ch];
      *reinterpret_cast<uint4*>(tile + rr * MP_TROW + ch * 8) = val;
    }
    __syncthreads();
    for (int n16 = 0; n16 * 16 < tcnt; ++n16) {
      u32x4 bf[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) bf[kk] = *reinterpret_cast<const u32x4*>(tile + (n16 * 16 + li) * MP_TROW + kk * 32 + lg * 8);
      const int col = t0 + n16 * 16 + li;
      const uint32_t colkey = col < t0 + tcnt ? 0xffffu - (uint32_t)col : 0xffffffffu;   // all ones: a column past the tile (masked below)
#pragma unroll
      for (int it = 0; it < MP_ITER; ++it) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = Mfma<T>::run(af[it][kk], bf[kk], acc);
        uint32_t pk[2] = {round_pack2<T>(acc[0], acc[1]), round_pack2<T>(acc[2], acc[3])};   // similarities in the model dtype (:150)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          // both halves at once: sign-magnitude -> unsigned order (negative: all bits flipped, else the sign bit set)
          const uint32_t neg = (pk[h2] >> 15) & 0x00010001u;
          uint32_t m = pk[h2] ^ ((neg * 0xffffu) | 0x80008000u);
          if (exact) {
            if (Elem<T>::is_nan((uint16_t)(pk[h2] & 0xffffu))) m |= 0x0000ffffu;
            if (Elem<T>::is_nan((uint16_t)(pk[h2] >> 16))) m |= 0xffff0000u;
          }
          const uint32_t k0 = (m << 16) | colkey, k1 = (m & 0xffff0000u) | colkey;
          if (colkey != 0xffffffffu) {
            bestk[it][2 * h2] = max(bestk[it][2 * h2], k0);
            bestk[it][2 * h2 + 1] = max(bestk[it][2 * h2 + 1], k1);
          }
        }
      }
    }
  }
  // first maximum across the 16 columns (lanes of one lg group): the largest key
#pragma unroll
  for (int it = 0; it < MP_ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t kx = bestk[it][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) kx = max(kx, (uint32_t)__shfl_xor((int)kx, o, 64));
      const int row = row_wg + it * MP_ROWS + wave * 16 + lg * 4 + r;
      if (li == 0 && row < n) p.pivot[(int64_t)bh * p.S + row] = (int32_t)(0xffffu - (kx & 0xffffu));
    }
  }
}

// ---- 4b. pivot, pipelined (round 6): the same result as merge_pivot_kernel, organised like the K scan (logits2_kernel) ----
// Head size 128 and at most MP2_TN kept rows (budget 128 + window 8: what the runners use) - everything else keeps the kernel
// above.  What was slow there (151 us for 268 MB at S = 32768 = 0.22 of HBM): every workgroup paid three dependent round
// trips (ndrop -> drop list -> rows) before its first instruction of arithmetic, all workgroups of a round did so in lockstep,
// the row loads fetched 64-byte pieces of 16 different rows per instruction, and ~1800 vector instructions per wave went
// into the normalisation and the arg-max keys.  Here a workgroup owns `nst` consecutive stages of 128 dropped rows of one
// head; per stage and wave 8 nontemporal row-major 1-KB loads (4 whole rows per instruction) land in registers while the
// previous stage is processed, the drop-list entries of the stage after that are fetched at the same time (one pipeline
// instead of three round trips per 256 rows), the rows go through a wave-private XOR-swizzled LDS transpose into the SAME
// MFMA fragments the kernel above builds, and from there on the arithmetic is the same instruction for instruction - the
// pivots are bit-identical by construction - with two cheaper forms:
//   * bf16: dtype(x / n) == dtype(x * fp32(1/n)) for every bf16 x and every NORMAL bf16 n: a quotient of two 8-bit
//     significands is never a bf16 rounding midpoint (that needs a 9-bit odd factor) and is at least 2^-17 (relative) away
//     from one, the multiply's error is below 2^-23.  Rows whose norm is zero / subnormal / not finite (wave-uniform test)
//     take div_const as above.  fp16 (11-bit significands: the gap shrinks to the error) always divides.
//   * arg-max keys: raw 16-bit pattern << 16 | (0xffff - column) compared as SIGNED integers orders all non-negative
//     similarities correctly and puts every negative one below them; a row whose maximum is not above +0 (or a wave that can
//     see a NaN) is redone with the exact order-preserving keys of the kernel above (wave-uniform, ~never for real keys: the
//     observation-window rows are among the kept rows).
constexpr int MP2_TN = 144;               // kept rows of the one LDS tile: [MP2_TN][16 chunks of 16 B], chunk c of row r at slot c ^ (r & 15)
constexpr int MP2_HT = 128;               // dropped rows per stage (4 waves x 32)
constexpr int MP2_STAGE_BYTES = 4 * 4096; // wave-private transpose areas: 16 rows x 256 B each (a stage goes through in two halves)
constexpr int MP2_LDS_BYTES = MP2_STAGE_BYTES + MP2_TN * 256;   // 53 248 B: three workgroups per CU

__device__ __forceinline__ int dpp_max16_i32(int x) {          // maximum over the 16 lanes of a DPP row, in every lane
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));      // row_half_mirror
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));      // row_mirror
  return x;
}
__device__ __forceinline__ uint32_t dpp_max16_u32(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true));
  return x;
}

template <typename T, bool NT, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PF == 1 ? 3 : 2))) void merge_pivot2_kernel(MergeParams p, int nst, uint32_t row_bytes) {
  constexpr int KS = 4, D = 128, CPR = D / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char mp2_smem[];
  u32x4* kst_all = reinterpret_cast<u32x4*>(mp2_smem);                          // [4 waves][16 rows][16 chunks]
  u32x4* tile = reinterpret_cast<u32x4*>(mp2_smem + MP2_STAGE_BYTES);           // [MP2_TN][16 chunks], swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  u32x4* kst = kst_all + wave * 256;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int n = *p.ndrop;
  const int row_wg = blockIdx.x * nst * MP2_HT;
  if (row_wg >= n) return;
  const int nh = min(nst, (n - row_wg + MP2_HT - 1) / MP2_HT);
  const int nt = p.k + p.w;                                                     // <= MP2_TN (launch_merge_t)
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const uint16_t* tn = reinterpret_cast<const uint16_t*>(p.tn) + (int64_t)bh * p.ntp * D;
  int32_t* pivot = p.pivot + (int64_t)bh * p.S;

  // the pipeline: positions of stage s+PF+1 | rows of stages s+1 .. s+PF (PF register sets) | arithmetic of stage s
  auto issue_idx = [&](int s, int (&id)[8]) {
    const int r0 = row_wg + s * MP2_HT + wave * 32 + lg;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = r0 + 4 * j; id[j] = p.drop[r < n ? r : n - 1]; }   // clamp: rows past n are never written
  };
  // row j of a lane's eight: rows 4j + lg of the wave's 32, chunk li ^ ((4j + lg) & 15) - the swizzle only depends on j & 3
  const char* lane_base[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lane_base[j] = reinterpret_cast<const char*>(kbase) + ((li ^ ((4 * j + lg) & 15)) << 4);
  auto issue_rows = [&](const int (&id)[8], u32x4 (&pre)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4* ptr = reinterpret_cast<const u32x4*>(lane_base[j & 3] + (uint64_t)(uint32_t)id[j] * row_bytes);   // v_mad_u64_u32
      pre[j] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    }
  };
  int idn[8];
  u32x4 preA[8], preB[PF == 2 ? 8 : 1];
  {
    int id0[8], id1[8];
    issue_idx(0, id0);                                                          // all position loads of the prologue leave together
    if (nh > 1) issue_idx(1, id1);
    if (PF == 2 && nh > 2) issue_idx(2, idn);
    issue_rows(id0, preA);
    if constexpr (PF == 2) {
      if (nh > 1) issue_rows(id1, preB);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) idn[j] = id1[j];
    }
  }
  for (int c = tid; c < MP2_TN * CPR; c += 256) {                               // the head's unit-norm kept keys, once; rows past nt are zero
    const int rr = c / CPR, ch = c - rr * CPR;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (rr < nt) val = reinterpret_cast<const uint4*>(tn + (int64_t)rr * D)[ch];
    tile[rr * 16 + (ch ^ (rr & 15))] = __builtin_bit_cast(u32x4, val);
  }
  const bool kept_bad = p.kept_bad[bh] != 0;
  __syncthreads();
  const int nfull = nt >> 4, ntail = nt & 15;
  // B fragments: lane (li, lg) reads row 16 n + li, chunk 4 kk + lg -> slot (4 kk + lg) ^ li (16 lanes of a row group: 16 different slots)
  const u32x4* tile_l[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) tile_l[kk] = tile + li * 16 + ((kk * 4 + lg) ^ li);
  // the lane of each 16-lane group that stores pivot (t, r): li = 4 t + r
  const int st_t = (li >> 2) & 1, st_r = li & 3;

  auto stage = [&](int s, u32x4 (&pre)[8]) {
    for (int j = 0; j < 4; ++j) kst[j * 64 + lane] = pre[j];                    // rows 0..15 of the wave's 32
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32x4 af[2][KS];                                                            // row li of group t = dropped row (stage) + wave*32 + t*16 + li
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) af[0][kk] = kst[li * 16 + ((kk * 4 + lg) ^ li)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                            // every lane has its fragments: the area takes the second half
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) kst[j * 64 + lane] = pre[4 + j];                // rows 16..31
    if (s + PF < nh) {
      issue_rows(idn, pre);
      if (s + PF + 1 < nh) issue_idx(s + PF + 1, idn);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) af[1][kk] = kst[li * 16 + ((kk * 4 + lg) ^ li)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                            // the staging area is free for the next stage

    // unit-norm rows in registers: the summation order of merge_pivot_kernel (8 elements x 4 k-steps per lane, then the 4 lanes of a row)
    bool bad_row = false;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float xs[KS * 8];
      float n2 = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        U4 u;
        u.v = make_uint4(af[t][kk].x, af[t][kk].y, af[t][kk].z, af[t][kk].w);
#pragma unroll
        for (int e = 0; e < 8; ++e) { xs[kk * 8 + e] = Elem<T>::to_f32(u.h[e]); n2 += xs[kk * 8 + e] * xs[kk * 8 + e]; }
      }
      n2 += __shfl_xor(n2, 16, 64);
      n2 += __shfl_xor(n2, 32, 64);
      const float nr = Elem<T>::to_f32(Elem<T>::from_f32(sqrtf(n2)));
      const bool bad = !(n2 < INFINITY) || !(nr > 0.f);
      bad_row |= bad;
      const float rn = 1.0f / nr;
      bool divide = true;
      if constexpr (std::is_same<T, BF16>::value)                              // bf16 (see the head of this kernel)
        divide = __ballot(bad || !(nr >= 1.17549435e-38f)) != 0ull;
      if (divide) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          u32x4 a;
          a.x = round_pack2<T>(div_const(xs[kk * 8 + 0], nr, rn), div_const(xs[kk * 8 + 1], nr, rn));
          a.y = round_pack2<T>(div_const(xs[kk * 8 + 2], nr, rn), div_const(xs[kk * 8 + 3], nr, rn));
          a.z = round_pack2<T>(div_const(xs[kk * 8 + 4], nr, rn), div_const(xs[kk * 8 + 5], nr, rn));
          a.w = round_pack2<T>(div_const(xs[kk * 8 + 6], nr, rn), div_const(xs[kk * 8 + 7], nr, rn));
          af[t][kk] = a;
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          u32x4 a;
          a.x = round_pack2<T>(xs[kk * 8 + 0] * rn, xs[kk * 8 + 1] * rn);
          a.y = round_pack2<T>(xs[kk * 8 + 2] * rn, xs[kk * 8 + 3] * rn);
          a.z = round_pack2<T>(xs[kk * 8 + 4] * rn, xs[kk * 8 + 5] * rn);
          a.w = round_pack2<T>(xs[kk * 8 + 6] * rn, xs[kk * 8 + 7] * rn);
          af[t][kk] = a;
        }
      }
    }
    bool exact = kept_bad || __ballot(bad_row) != 0ull;                         // wave-uniform
    uint32_t res[2][4];                                                         // low 16 bits: 0xffff - kept-row number of the first maximum
    if (!exact) {
      int best[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) best[t][r] = (int)0x80000000;
      // The B fragments of tile n+1 are read while the MFMAs of tile n run, and the keys of tile n-1 are formed behind the
      // MFMAs of tile n (they do not depend on them): LDS latency, matrix pipe and vector pipe overlap inside one wave.
      auto load_b = [&](u32x4 (&bf)[KS], int n16) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bf[kk] = tile_l[kk][n16 * 256];
      };
      auto mma = [&](const u32x4 (&bf)[KS], f32x4 (&acc)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KS; ++kk) acc[t] = Mfma<T>::run(af[t][kk], bf[kk], acc[t]);
        }
      };
      auto keys = [&](const f32x4 (&acc)[2], int n16, bool tail) {
        const uint32_t colkey = 0xffffu - (uint32_t)(n16 * 16 + li);
        const bool valid = !tail || li < ntail;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint32_t p01 = round_pack2<T>(acc[t][0], acc[t][1]), p23 = round_pack2<T>(acc[t][2], acc[t][3]);   // similarities in the model dtype (:150)
          int k0 = (int)((p01 << 16) | colkey), k1 = (int)((p01 & 0xffff0000u) | colkey);
          int k2 = (int)((p23 << 16) | colkey), k3 = (int)((p23 & 0xffff0000u) | colkey);
          if (tail) {
            k0 = valid ? k0 : (int)0x80000000; k1 = valid ? k1 : (int)0x80000000;
            k2 = valid ? k2 : (int)0x80000000; k3 = valid ? k3 : (int)0x80000000;
          }
          best[t][0] = max(best[t][0], k0); best[t][1] = max(best[t][1], k1);
          best[t][2] = max(best[t][2], k2); best[t][3] = max(best[t][3], k3);
        }
      };
      const int ntile = nfull + (ntail ? 1 : 0);                                // >= 1
      u32x4 bfA[KS], bfB[KS];
      f32x4 accA[2], accB[2];
      load_b(bfA, 0);
      int n16 = 0;
      for (; n16 + 1 < ntile; n16 += 2) {                                       // tiles n16 (A) and n16 + 1 (B)
        load_b(bfB, n16 + 1);
        mma(bfA, accA);
        if (n16 > 0) keys(accB, n16 - 1, false);
        if (n16 + 2 < ntile) load_b(bfA, n16 + 2);
        mma(bfB, accB);
        keys(accA, n16, false);
      }
      if (n16 < ntile) {                                                        // odd tile count: the last tile is in A
        mma(bfA, accA);
        if (n16 > 0) keys(accB, n16 - 1, false);
        keys(accA, n16, ntail != 0);
      } else {
        keys(accB, n16 - 1, ntail != 0);
      }
      bool redo = false;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kx = dpp_max16_i32(best[t][r]);
          redo |= kx < 0x00010000;                                              // the row's maximum is not above +0: signed order is not enough
          res[t][r] = (uint32_t)kx;
        }
      exact = __ballot(redo) != 0ull;
    }
    if (exact) {                                                                // merge_pivot_kernel's keys: total order incl. negative values, NaN first
      uint32_t bestk[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bestk[t][r] = 0u;
      for (int n16 = 0; n16 * 16 < nt; ++n16) {
        u32x4 bf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bf[kk] = tile_l[kk][n16 * 256];
        const int col = n16 * 16 + li;
        const uint32_t colkey = col < nt ? 0xffffu - (uint32_t)col : 0xffffffffu;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KS; ++kk) acc = Mfma<T>::run(af[t][kk], bf[kk], acc);
          uint32_t pk[2] = {round_pack2<T>(acc[0], acc[1]), round_pack2<T>(acc[2], acc[3])};
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const uint32_t neg = (pk[h2] >> 15) & 0x00010001u;
            uint32_t m = pk[h2] ^ ((neg * 0xffffu) | 0x80008000u);
            if (Elem<T>::is_nan((uint16_t)(pk[h2] & 0xffffu))) m |= 0x0000ffffu;
            if (Elem<T>::is_nan((uint16_t)(pk[h2] >> 16))) m |= 0xffff0000u;
            const uint32_t k0 = (m << 16) | colkey, k1 = (m & 0xffff0000u) | colkey;
            if (colkey != 0xffffffffu) {
              bestk[t][2 * h2] = max(bestk[t][2 * h2], k0);
              bestk[t][2 * h2 + 1] = max(bestk[t][2 * h2 + 1], k1);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[t][r] = dpp_max16_u32(bestk[t][r]);
    }
    // every lane of a 16-lane group holds the group's 8 results: lane li = 4 t + r stores pivot (t, r) - one store instruction
    {
      const uint32_t r0 = st_t ? res[1][0] : res[0][0], r1 = st_t ? res[1][1] : res[0][1];
      const uint32_t r2 = st_t ? res[1][2] : res[0][2], r3 = st_t ? res[1][3] : res[0][3];
      const uint32_t rv = st_r == 0 ? r0 : (st_r == 1 ? r1 : (st_r == 2 ? r2 : r3));
      const int row = row_wg + s * MP2_HT + wave * 32 + st_t * 16 + lg * 4 + st_r;
      if (li < 8 && row < n) pivot[row] = (int32_t)(0xffffu - (rv & 0xffffu));
    }
  };
  if constexpr (PF == 2) {
    for (int s = 0; s < nh; s += 2) {
      stage(s, preA);
      if (s + 1 < nh) stage(s + 1, preB);
    }
  } else {
    for (int s = 0; s < nh; ++s) stage(s, preA);
  }
}

// ---- 5. bucket: per (b,h) the dropped rows grouped by the kept row they chose (counting sort over the pivots) ----
// One workgroup of 1024 threads per head: LDS histogram over a range of MB_RANGE kept rows, exclusive scan (written out as
// bstart[bh][j]), placement through LDS cursors.  blist holds the dropped POSITIONS; within a group the order is whatever
// the atomics produced - the scatter kernel orders every group itself.  More than MB_RANGE kept rows: one pass per range.
constexpr int MB_RANGE = 8192;

__global__ __launch_bounds__(1024) void merge_bucket_kernel(MergeParams p, int fast) {
  extern __shared__ __attribute__((aligned(16))) uint16_t mb_list[];      // fast path: the head's grouped positions (S <= 32768 entries, 16 bits each)
  __shared__ int32_t cnt[MB_RANGE];
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x;
  const int n = *p.ndrop, nt = p.k + p.w;
  const int32_t* piv = p.pivot + (int64_t)bh * p.S;
  int32_t* list = p.blist + (int64_t)bh * p.S;
  int32_t* start = p.bstart + (int64_t)bh * (nt + 1);
  if (fast) {                                                             // host: nt * 32 <= MB_RANGE && S <= 32768
    // the usual size (round 6): every thread keeps its <= 32 pivots and positions in registers (all loads of the kernel leave
    // together), and every counter exists 32 times - copy (lane & 31) sits in bank (lane & 31), so the 64 atomics of a wave
    // instruction touch each bank at most twice whatever the pivots are (136 counters shared by 1024 threads serialised:
    // the one-counter form spent ~10 us per pass at S = 32768).  The scan runs over (kept row, copy) pairs: the rows of a
    // group land grouped by copy - the scatter kernel orders every group itself.  The placement goes into an LDS copy of
    // the list (positions fit 16 bits) that is written out with coalesced stores.
    int pv[32], pos[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int i = u * 1024 + tid; pv[u] = i < n ? piv[i] : -1; }
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int i = u * 1024 + tid; pos[u] = p.drop[i < n ? i : 0]; }
    const int cp = lane & 31;
    for (int j = tid; j < MB_RANGE; j += 1024) cnt[j] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if ((unsigned)pv[u] < (unsigned)nt) atomicAdd(&cnt[pv[u] * 32 + cp], 1);
    __syncthreads();
    int c[8], sum = 0;                                                    // 8 consecutive (row, copy) counters per thread
#pragma unroll
    for (int e = 0; e < 8; ++e) { c[e] = cnt[tid * 8 + e]; sum += c[e]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) woff += w2 < wave ? wsum[w2] : 0;
    int run = woff + incl - sum;
    if ((tid & 3) == 0 && tid / 4 < nt) start[tid / 4] = run;             // copy 0 of kept row tid / 4: the group's first entry
#pragma unroll
    for (int e = 0; e < 8; ++e) { cnt[tid * 8 + e] = run; run += c[e]; } // the cursors
    if (tid == 0) start[nt] = n;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if ((unsigned)pv[u] < (unsigned)nt) mb_list[atomicAdd(&cnt[pv[u] * 32 + cp], 1)] = (uint16_t)pos[u];
    __syncthreads();
    // the grouped list leaves LDS in order: 31k scattered 4-byte stores from one CU took ~13 us, the coalesced copy ~1
    for (int i = tid; i < n; i += 1024) list[i] = (int32_t)mb_list[i];
    return;
  }
  int base = 0;                                                           // rows placed by the earlier ranges
  for (int lo = 0; lo < nt; lo += MB_RANGE) {
    const int hi = min(nt, lo + MB_RANGE);
    for (int j = tid; j < MB_RANGE; j += 1024) cnt[j] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {                             // 8 loads in flight per thread (a plain loop pays one round trip per element)
      int pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024 + tid; pv[u] = piv[i < n ? i : n - 1]; if (i >= n) pv[u] = -1; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (pv[u] >= lo && pv[u] < hi) atomicAdd(&cnt[pv[u] - lo], 1);
    }
    __syncthreads();
    int c[8], sum = 0;                                                    // 8 consecutive kept rows per thread
#pragma unroll
    for (int e = 0; e < 8; ++e) { c[e] = cnt[tid * 8 + e]; sum += c[e]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { const int t2 = wsum[w2]; total += t2; woff += w2 < wave ? t2 : 0; }
    int run = base + woff + incl - sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = lo + tid * 8 + e;
      if (j < hi) start[j] = run;
      cnt[tid * 8 + e] = run;                                             // the group's cursor
      run += c[e];
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {
      int pv[8], pos[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 1024 + tid;
        pv[u] = piv[i < n ? i : n - 1];
        pos[u] = p.drop[i < n ? i : n - 1];
        if (i >= n) pv[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (pv[u] >= lo && pv[u] < hi) list[atomicAdd(&cnt[pv[u] - lo], 1)] = pos[u];
    }
    base += total;
    __syncthreads();
  }
  if (tid == 0) start[nt] = base;
}

// ---- 6. scatter-mean: kept row j collects, in ascending order, the dropped rows that chose it ----
// Workgroup = one kept row of one (b,h): threads 0..D/2-1 = the KEY row (two adjacent elements each), D/2..D-1 the VALUE row.  The group
// arrives unordered; the reference accumulates in ascending position (fp32, scatter_reduce walks the source in order), so
// the positions are set as bits of an LDS bitmap over [0, S) and read back in order: a popcount prefix gives every set bit
// its rank, MS_CAP ranks at a time become the list of a walk.  The walk issues the MS_B row loads of a batch together and
// consumes them in list order: the accumulation order is the list order, whatever the latency of a row.
constexpr int MS_CAP = 2048;              // list entries per pass
constexpr int MS_B = 32;                  // rows per batch of the walk (all in flight together)

template <typename T, int KS>
__global__ __launch_bounds__(KS * 32) void merge_scatter_kernel(MergeParams p) {
  constexpr int D = KS * 32, NTH = D, NW = NTH / 64;        // two adjacent elements per thread: D/2 threads per row, K then V
  extern __shared__ __attribute__((aligned(16))) uint32_t ms_bitmap[];    // [(S + 31) / 32]
  __shared__ int32_t lst[MS_CAP];
  __shared__ int32_t wsum[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = (tid % (D / 2)) * 2, is_v = tid / (D / 2);
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x;
  const int nt = p.k + p.w;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int32_t* start = p.bstart + (int64_t)bh * (nt + 1);
  const int beg = start[j], m = start[j + 1] - beg;
  const int32_t* group = p.blist + (int64_t)bh * p.S + beg;
  const uint16_t* base = is_v ? reinterpret_cast<const uint16_t*>(p.vptr) + (int64_t)b * p.vs_b + (int64_t)hk * p.vs_h
                              : reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int64_t sstride = is_v ? p.vs_s : p.ks_s;
  const int tpos = is_v ? val_target_pos(p, idx_row, j) : key_target_pos(p, idx_row, j);
  const uint32_t traw = *reinterpret_cast<const uint32_t*>(base + (int64_t)tpos * sstride + d);   // the kept row BEFORE merging (gather at :157/:160)
  const pkv_f32x2 t = {Elem<T>::to_f32((uint16_t)(traw & 0xffffu)), Elem<T>::to_f32((uint16_t)(traw >> 16))};
  pkv_f32x2 acc = t;                                                      // include_self
  if (m > 0) {                                                            // workgroup-uniform
    const int nw = (p.S + 31) >> 5;
    for (int wd = tid; wd < nw; wd += NTH) ms_bitmap[wd] = 0u;
    __syncthreads();
    for (int e = tid; e < m; e += NTH) { const int s = group[e]; atomicOr(&ms_bitmap[s >> 5], 1u << (s & 31)); }
    __syncthreads();
    // thread t owns the words [t*wpt, (t+1)*wpt): rank of its first set bit = exclusive prefix of the popcounts
    const int wpt = (nw + NTH - 1) / NTH;
    const int w0 = min(nw, tid * wpt), w1 = min(nw, w0 + wpt);
    int pc = 0;
    for (int wd = w0; wd < w1; ++wd) pc += __popc(ms_bitmap[wd]);
    int incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) woff += w2 < wave ? wsum[w2] : 0;
    const int first = woff + incl - pc;
    for (int pass0 = 0; pass0 < m; pass0 += MS_CAP) {
      int r = first;
      for (int wd = w0; wd < w1 && r < pass0 + MS_CAP; ++wd) {
        uint32_t bits = ms_bitmap[wd];
        const int c = __popc(bits);
        if (r + c <= pass0) { r += c; continue; }
        while (bits) {
          const int bpos = __builtin_ctz(bits);
          bits &= bits - 1;
          if (r >= pass0 && r < pass0 + MS_CAP) lst[r - pass0] = wd * 32 + bpos;
          ++r;
        }
      }
      __syncthreads();
      const uint32_t mm = (uint32_t)min(MS_CAP, m - pass0);
      // Every lane would otherwise repeat the (wave-uniform) list look-up and 64-bit address arithmetic for every row: ~15
      // vector instructions per row and lane made this walk VALU-bound.  Instead each lane looks up ONE entry of a block of 64
      // rows; the row numbers then reach the scalar unit through v_readlane and the loads use a scalar base.
      for (uint32_t l64 = 0; l64 < mm; l64 += 64) {
        const int myrow = lst[l64 + (uint32_t)lane < mm ? l64 + (uint32_t)lane : mm - 1];
        const uint32_t left = mm - l64 < 64u ? mm - l64 : 64u;
        for (uint32_t b0 = 0; b0 < left; b0 += MS_B) {
          uint32_t x[MS_B];
#pragma unroll
          for (int u = 0; u < MS_B; ++u) {
            const int r2 = __builtin_amdgcn_readlane(myrow, (int)(b0 + u < 64u ? b0 + u : 63u));   // rows past `left` repeat the last entry (masked below)
            x[u] = *reinterpret_cast<const uint32_t*>(base + (int64_t)r2 * sstride + d);
          }
#pragma unroll
          for (int u = 0; u < MS_B; ++u) {
            if (b0 + u < left) {
              // two elements per packed fp32 instruction (IEEE per component, no contraction): (x + t) -> dtype, / 2 -> dtype (:158)
              const pkv_f32x2 xv = {Elem<T>::to_f32((uint16_t)(x[u] & 0xffffu)), Elem<T>::to_f32((uint16_t)(x[u] >> 16))};
              const pkv_f32x2 sm = xv + t;
              const uint32_t r1 = round_pack2<T>(sm.x, sm.y);
              const pkv_f32x2 hf = pkv_f32x2{Elem<T>::to_f32((uint16_t)(r1 & 0xffffu)), Elem<T>::to_f32((uint16_t)(r1 >> 16))} * pkv_f32x2{0.5f, 0.5f};
              const uint32_t r2b = round_pack2<T>(hf.x, hf.y);
              acc = acc + pkv_f32x2{Elem<T>::to_f32((uint16_t)(r2b & 0xffffu)), Elem<T>::to_f32((uint16_t)(r2b >> 16))};
            }
          }
        }
      }
      __syncthreads();                                                    // the list has been walked
    }
  }
  const int cnt = 1 + m;
  const uint32_t sq = round_pack2<T>(acc.x, acc.y);                       // the scattered SUM in the model dtype
  const float cnt_q = Elem<T>::to_f32(Elem<T>::from_f32((float)cnt));     // the count in the model dtype (rounds above 256 / 2048)
  uint16_t* out = reinterpret_cast<uint16_t*>(is_v ? p.v_out : p.k_out) + ((int64_t)bh * nt + j) * D + d;
  *reinterpret_cast<uint32_t*>(out) = round_pack2<T>(Elem<T>::to_f32((uint16_t)(sq & 0xffffu)) / cnt_q, Elem<T>::to_f32((uint16_t)(sq >> 16)) / cnt_q);
}

// knobs of the pipelined pivot kernel (identical results; defaults are the measured best, env vars exist for A/B runs)
static int merge_env(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
static int merge_pivot2_on() { static int t = merge_env("PKV_MERGE_PIVOT2", 1); return t; }
static int merge_pivot2_nst() { static int t = merge_env("PKV_MERGE_NST", 0); return t; }       // 0 = by size
static int merge_pivot2_pf() { static int t = merge_env("PKV_MERGE_PF", 1); return t; }         // stages of rows in flight per wave
static int merge_pivot2_nt() { static int t = merge_env("PKV_MERGE_NT", 1); return t; }         // nontemporal row loads
static int merge_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

template <typename T, int KS>
hipError_t launch_merge_t(const MergeParams& p, hipStream_t st) {
  using Sh = MergeShape<KS>;
  const int nt = p.k + p.w;
  const dim3 gt((nt + 3) / 4, p.B * p.H), gp((p.S + MP_ROWS * Sh::ITER - 1) / (MP_ROWS * Sh::ITER), p.B * p.H), gs(nt, p.B * p.H);
  const size_t bitmap_bytes = (size_t)((p.S + 31) / 32) * 4;
  hipLaunchKernelGGL((merge_targets_kernel<T, KS>), gt, dim3(256), 0, st, p);
  bool pipelined = false;
  if constexpr (KS == 4) {
    if (nt <= MP2_TN && merge_pivot2_on() && p.ks_s > 0 && p.ks_s < (int64_t)1 << 30) {
      // stages per workgroup: whole rounds of the chip's 3-per-CU residency (LDS: 52 KB each), ~12 stages per workgroup
      const int64_t sph = (p.S + MP2_HT - 1) / MP2_HT, stages = (int64_t)p.B * p.H * sph, slots = (int64_t)3 * merge_cus();
      int nst = merge_pivot2_nst();
      if (nst <= 0) {
        const int64_t rounds = std::max<int64_t>(1, (stages + slots * 6) / (slots * 12));
        nst = (int)std::max<int64_t>(1, std::min<int64_t>(sph, (stages + slots * rounds - 1) / (slots * rounds)));
      }
      const dim3 g2((p.S + nst * MP2_HT - 1) / (nst * MP2_HT), p.B * p.H);
      const size_t lds = MP2_LDS_BYTES;
      hipError_t e = hipSuccess;
#define PKV_MP2(NTL, PFD)                                                                                   \
  do {                                                                                                      \
    e = dyn_lds(reinterpret_cast<const void*>(merge_pivot2_kernel<T, NTL, PFD>), lds);                      \
    if (e != hipSuccess) return e;                                                                          \
    hipLaunchKernelGGL((merge_pivot2_kernel<T, NTL, PFD>), g2, dim3(256), lds, st, p, nst, (uint32_t)(p.ks_s * 2));                 \
  } while (0)
      if (merge_pivot2_pf() == 2) { if (merge_pivot2_nt()) PKV_MP2(true, 2); else PKV_MP2(false, 2); }
      else                        { if (merge_pivot2_nt()) PKV_MP2(true, 1); else PKV_MP2(false, 1); }
#undef PKV_MP2
      pipelined = true;
    }
  }
  if (!pipelined) hipLaunchKernelGGL((merge_pivot_kernel<T, KS>), gp, dim3(256), 0, st, p);
  {
    const int fast = (nt * 32 <= MB_RANGE && p.S <= 32768) ? 1 : 0;
    const size_t blds = fast ? (size_t)32768 * 2 : 0;
    if (fast) {
      const hipError_t eb = dyn_lds(reinterpret_cast<const void*>(merge_bucket_kernel), blds);
      if (eb != hipSuccess) return eb;
    }
    hipLaunchKernelGGL(merge_bucket_kernel, dim3(p.B * p.H), dim3(1024), blds, st, p, fast);
  }
  hipLaunchKernelGGL((merge_scatter_kernel<T, KS>), gs, dim3(KS * 32), bitmap_bytes, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fp32 tensors (round 4; the reference's merge_kv is dtype-generic): the same six steps without any rounding to a model
// dtype.  mark / droplist / bucket are shared; targets, pivot and scatter have fp32 forms on v_mfma_f32_16x16x4_f32 and
// plain fp32 arithmetic.  Built for completeness (the matrix pipe runs fp32 at 1/16 of its bf16 rate): no LDS tiling of the
// kept rows - they are read from the L2-resident unit-norm buffer.  The fp32 dot products are not pinned to an order by the
// reference: a dropped row whose two best similarities agree to ~1e-7 may choose the other kept row (tests allow that).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float mf32x4;

template <int KSF>                                                         // float4 pieces per lane and row: D / 16
__global__ __launch_bounds__(256) void merge_targets_f32_kernel(MergeParams p) {
  constexpr int D = KSF * 16, EPL = D / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x * 4 + wave;
  if (j >= p.k + p.w) return;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int pos = key_target_pos(p, idx_row, j);
  const float* src = reinterpret_cast<const float*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h + (int64_t)pos * p.ks_s + lane * EPL;
  float x[EPL], sq = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { x[e] = src[e]; sq += x[e] * x[e]; }
  const float n2 = wave_sum(sq);
  const float n = sqrtf(n2);                                               // torch.norm
  if (lane == 0 && (!(n2 < INFINITY) || !(n > 0.f))) atomicOr(p.kept_bad + bh, 1);
  float* dst = reinterpret_cast<float*>(p.tn) + ((int64_t)bh * p.ntp + j) * D + lane * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) dst[e] = x[e] / n;                         // k / norm
}

// order-preserving 32-bit key of an fp32 similarity; NaN ranks above everything (torch.max propagates the first NaN)
__device__ __forceinline__ uint32_t sim_key_f32(float x) {
  if (x != x) return 0xffffffffu;
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int KSF>
__global__ __launch_bounds__(256) void merge_pivot_f32_kernel(MergeParams p) {
  constexpr int D = KSF * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int n = *p.ndrop;
  const int row0 = blockIdx.x * 64 + wave * 16;
  if (blockIdx.x * 64 >= n) return;
  const int nt = p.k + p.w;
  const float* kbase = reinterpret_cast<const float*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float* tn = reinterpret_cast<const float*>(p.tn) + (int64_t)bh * p.ntp * D;
  // A operand: dropped row row0 + li (clamped), unit-normalised in registers
  mf32x4 af[KSF];
  {
    const int r = row0 + li < n ? row0 + li : n - 1;
    const float* row = kbase + (int64_t)p.drop[r] * p.ks_s + 4 * lg;
    float n2 = 0.f;
#pragma unroll
    for (int jj = 0; jj < KSF; ++jj) {
      af[jj] = *reinterpret_cast<const mf32x4*>(row + 16 * jj);
#pragma unroll
      for (int i = 0; i < 4; ++i) n2 += af[jj][i] * af[jj][i];
    }
    n2 += __shfl_xor(n2, 16, 64);
    n2 += __shfl_xor(n2, 32, 64);
    const float nr = sqrtf(n2);
#pragma unroll
    for (int jj = 0; jj < KSF; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[jj][i] = af[jj][i] / nr;
  }
  // lane holds rows 4*lg + r of the wave's 16 and column li of every 16-target tile: strictly-greater updates keep the FIRST
  // maximum of the lane's columns; the final reduction over the 16 lanes prefers the smaller column on equal keys
  uint32_t bestk[4] = {0u, 0u, 0u, 0u};
  int bestc[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  for (int t0 = 0; t0 < nt; t0 += 16) {
    const int col = t0 + li;
    mf32x4 bf[KSF];
    const float* trow = tn + (int64_t)(col < nt ? col : nt - 1) * D + 4 * lg;
#pragma unroll
    for (int jj = 0; jj < KSF; ++jj) bf[jj] = *reinterpret_cast<const mf32x4*>(trow + 16 * jj);
    mf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < KSF; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[jj][i], bf[jj][i], acc, 0, 0, 0);   // D[dropped row][kept row]
    if (col < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t key = sim_key_f32(acc[r]);
        if (key > bestk[r] || bestc[r] == 0x7fffffff) { bestk[r] = key; bestc[r] = col; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint32_t kx = bestk[r];
    int cx = bestc[r];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const uint32_t ko = (uint32_t)__shfl_xor((int)kx, o, 64);
      const int co = __shfl_xor(cx, o, 64);
      if (ko > kx || (ko == kx && co < cx)) { kx = ko; cx = co; }
    }
    const int row = row0 + lg * 4 + r;
    if (li == 0 && row < n) p.pivot[(int64_t)bh * p.S + row] = cx;
  }
}

template <int D>
__global__ __launch_bounds__(2 * D) void merge_scatter_f32_kernel(MergeParams p) {
  constexpr int NTH = 2 * D, NW = NTH / 64;                                // one element per thread: D threads for K, D for V
  extern __shared__ __attribute__((aligned(16))) uint32_t ms_bitmap_f32[];
  __shared__ int32_t lst[MS_CAP];
  __shared__ int32_t wsum[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = tid % D, is_v = tid / D;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int j = blockIdx.x;
  const int nt = p.k + p.w;
  const int32_t* idx_row = p.idx + (int64_t)bh * p.idx_stride;
  const int32_t* start = p.bstart + (int64_t)bh * (nt + 1);
  const int beg = start[j], m = start[j + 1] - beg;
  const int32_t* group = p.blist + (int64_t)bh * p.S + beg;
  const float* base = is_v ? reinterpret_cast<const float*>(p.vptr) + (int64_t)b * p.vs_b + (int64_t)hk * p.vs_h
                           : reinterpret_cast<const float*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int64_t sstride = is_v ? p.vs_s : p.ks_s;
  const int tpos = is_v ? val_target_pos(p, idx_row, j) : key_target_pos(p, idx_row, j);
  const float t = base[(int64_t)tpos * sstride + d];                      // the kept row BEFORE merging (gather at :157/:160)
  float acc = t;                                                           // include_self
  if (m > 0) {                                                             // workgroup-uniform
    const int nw = (p.S + 31) >> 5;
    for (int wd = tid; wd < nw; wd += NTH) ms_bitmap_f32[wd] = 0u;
    __syncthreads();
    for (int e = tid; e < m; e += NTH) { const int s = group[e]; atomicOr(&ms_bitmap_f32[s >> 5], 1u << (s & 31)); }
    __syncthreads();
    const int wpt = (nw + NTH - 1) / NTH;
    const int w0 = min(nw, tid * wpt), w1 = min(nw, w0 + wpt);
    int pc = 0;
    for (int wd = w0; wd < w1; ++wd) pc += __popc(ms_bitmap_f32[wd]);
    int incl = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) woff += w2 < wave ? wsum[w2] : 0;
    const int first = woff + incl - pc;
    for (int pass0 = 0; pass0 < m; pass0 += MS_CAP) {
      int r = first;
      for (int wd = w0; wd < w1 && r < pass0 + MS_CAP; ++wd) {
        uint32_t bits = ms_bitmap_f32[wd];
        const int c = __popc(bits);
        if (r + c <= pass0) { r += c; continue; }
        while (bits) {
          const int bpos = __builtin_ctz(bits);
          bits &= bits - 1;
          if (r >= pass0 && r < pass0 + MS_CAP) lst[r - pass0] = wd * 32 + bpos;
          ++r;
        }
      }
      __syncthreads();
      const int mm = min(MS_CAP, m - pass0);
      for (int e0 = 0; e0 < mm; e0 += 8) {                                 // ascending order, 8 row loads in flight
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = base[(int64_t)lst[e0 + u < mm ? e0 + u : mm - 1] * sstride + d];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u < mm) acc += (x[u] + t) / 2.0f;                       // (x + t) / 2 (:158), fp32 accumulation in ascending order
      }
      __syncthreads();
    }
  }
  float* out = reinterpret_cast<float*>(is_v ? p.v_out : p.k_out) + ((int64_t)bh * nt + j) * D + d;
  *out = acc / (float)(1 + m);                                             // scatter_reduce(mean, include_self) (:159-162)
}

template <int KSF>
hipError_t launch_merge_f32_t(const MergeParams& p, hipStream_t st) {
  constexpr int D = KSF * 16;
  const int nt = p.k + p.w;
  const dim3 gt((nt + 3) / 4, p.B * p.H), gp((p.S + 63) / 64, p.B * p.H), gs(nt, p.B * p.H);
  const size_t bitmap_bytes = (size_t)((p.S + 31) / 32) * 4;
  hipLaunchKernelGGL((merge_targets_f32_kernel<KSF>), gt, dim3(256), 0, st, p);
  hipLaunchKernelGGL((merge_pivot_f32_kernel<KSF>), gp, dim3(256), 0, st, p);
  {
    const int fast = (nt * 32 <= MB_RANGE && p.S <= 32768) ? 1 : 0;
    const size_t blds = fast ? (size_t)32768 * 2 : 0;
    if (fast) {
      const hipError_t eb = dyn_lds(reinterpret_cast<const void*>(merge_bucket_kernel), blds);
      if (eb != hipSuccess) return eb;
    }
    hipLaunchKernelGGL(merge_bucket_kernel, dim3(p.B * p.H), dim3(1024), blds, st, p, fast);
  }
  hipLaunchKernelGGL((merge_scatter_f32_kernel<D>), gs, dim3(2 * D), bitmap_bytes, st, p);
  return hipGetLastError();
}

size_t merge_max_seq() { return 49152u * 8u; }    // the scatter kernel's position bitmap (one bit per position): 48 KB of dynamic LDS

hipError_t launch_merge(int dtype, const MergeParams& p, hipStream_t st) {
  // the union bitmap and, directly behind it, the per-head flags of the targets kernel
  hipError_t e = hipMemsetAsync(p.mask, 0, (size_t)(reinterpret_cast<char*>(p.kept_bad) - reinterpret_cast<char*>(p.mask)) + (size_t)p.B * p.H * 4, st);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)p.B * p.H * p.k;
  const int mb = (int)std::min<int64_t>((total + 255) / 256, 1024);
  hipLaunchKernelGGL(merge_mark_kernel, dim3(mb), dim3(256), 0, st, p);
  static const int droplist2 = merge_env("PKV_MERGE_DROPLIST2", 1);
  if (droplist2) hipLaunchKernelGGL(merge_droplist2_kernel, dim3((p.S + MD_SLICE - 1) / MD_SLICE), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(merge_droplist_kernel, dim3(1), dim3(1024), 0, st, p);
  if (dtype == 2) return p.D == 64 ? launch_merge_f32_t<4>(p, st) : (p.D == 256 ? launch_merge_f32_t<16>(p, st) : launch_merge_f32_t<8>(p, st));
  if (dtype == 0) {
    if (p.D == 64) return launch_merge_t<BF16, 2>(p, st);
    if (p.D == 256) return launch_merge_t<BF16, 8>(p, st);
    return launch_merge_t<BF16, 4>(p, st);
  }
  if (p.D == 64) return launch_merge_t<F16, 2>(p, st);
  if (p.D == 256) return launch_merge_t<F16, 8>(p, st);
  return launch_merge_t<F16, 4>(p, st);
}

}  // namespace pkv
