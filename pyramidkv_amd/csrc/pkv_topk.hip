// pkv_topk.hip — token selection kernels (gfx950).
//
//   topk_kernel       reference pyramidkv_utils.py:334 (:238,:270,:562)  scores.topk(k).indices
//   sort_rows_kernel  reference :706  attn_score.sort(dim=-1, descending=True)   (AdaKV/HeadKV)
//
// Order is pinned to (value desc, index asc): what ATen's GPU topk/sort produce (radix select of the
// k-th value, ">" then "==" in index order, stable block radix sort) and what oracle/topk_canonical
// restates.  Scores are 16-bit; the whole score row of a head lives in LDS (S-w <= 65536 keys),
// selection is an exact two-level (8+8 bit) radix select with bank-spread LDS counters, winners are
// compacted with wavefront scans in index order and ordered in LDS.
//
// Roofline: one workgroup per (b,h) row, 2*L bytes read once from L2/HBM; the kernel is
// latency/LDS-bound, not HBM-bound (64 KB per head at S=32k).
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"
#include "pkv_radix.hpp"

namespace pkv {

constexpr int TK_RANK_MAX = 512;         // k <= this: order by rank counting (O(k^2), no barriers)
constexpr int TK_RADIX_MAX = 4096;       // k <= this (and LDS allows): stable 2-pass LSD radix ordering
constexpr int TK_FAST_K = 512;           // k <= this: try the chunk-maxima prefilter first
constexpr int TK_FAST_C = 448;           // at most this many candidates: rank them directly (O(C^2)); above, the exact select inside the list is cheaper (measured crossover ~440)
constexpr int TK_MID_C = 4096;           // at most this many: exact select inside the candidate list first
constexpr size_t TK_LDS_LIMIT = 160 * 1024;
// one-level fast path (algo 1): 13-bit histogram of the chunk maxima, passing chunks staged in LDS, bucket counting sort
constexpr int TK2_PMAX = 2048;           // staged chunks (16 key slots per thread)
constexpr int TK2_CMAX = 4096;           // candidates
constexpr int TK2_NB = 2048;             // key buckets of the counting sort
constexpr int TK2_BMAX = 512;            // a bucket may hold at most this many candidates: ranking inside a bucket is quadratic (round 6: a row
                                         // with ~1000 candidates on one fp16 subnormal value took 55 us here, 2x the full path it now takes)

// LDS layout: keys u16[16*Lw] | X u32[max(8192,kpad)] | hist u32[256] | misc u32[64] | X2 u32[8192] (if it fits)
size_t topk_lds_bytes(int L, int k, int* Lw_out, int* kpad_out) {
  int per_wave = (L + TK_WAVES - 1) / TK_WAVES;
  int Lw = ((per_wave + 511) / 512) * 512;
  if (Lw < 512) Lw = 512;
  int kpad;
  if (k <= TK_RADIX_MAX) kpad = (k + 15) & ~15;
  else { kpad = 1; while (kpad < k) kpad <<= 1; }
  if (Lw_out) *Lw_out = Lw;
  if (kpad_out) *kpad_out = kpad;
  size_t xwords = (size_t)(kpad > TK_CNT_WORDS ? kpad : TK_CNT_WORDS);
  size_t base = (size_t)2 * TK_WAVES * Lw + 4 * xwords + 4 * 256 + 4 * 64;
  if (base + 4 * TK_CNT_WORDS <= TK_LDS_LIMIT) base += 4 * TK_CNT_WORDS;   // second counter / radix scratch region
  return base;
}

// One stable LSD radix pass over k composites (key<<16 | ~idx) by descending key byte `byte` (0/1).
// Element i lives in wave i / (ept*64): list order == (wave, e, lane) order, which makes the pass stable.
// table: u32[16][256] scratch; tot: u32[256] scratch.
__device__ __forceinline__ void radix_pass(const uint32_t* src, uint32_t* dst, uint32_t* table, uint32_t* tot,
                                           int k, int ept, int byte, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < TK_WAVES * 256; i += TK_THREADS) table[i] = 0;
  __syncthreads();
  uint32_t comp[4], dig[4], lrank[4];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < ept) {
      const int i = wave * ept * 64 + e * 64 + lane;
      const bool valid = i < k;
      comp[e] = valid ? src[i] : 0u;
      const uint32_t d = 255u - ((comp[e] >> (16 + 8 * byte)) & 255u);     // ascending in d == descending in key
      dig[e] = d;
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bb = __ballot(valid && bit);
        peers &= bit ? bb : ~bb;
      }
      const uint32_t before = (uint32_t)__popcll(peers & lt);
      const uint32_t cnt = (uint32_t)__popcll(peers);
      uint32_t prev = 0;
      if (valid) prev = table[wave * 256 + d];            // every peer reads the same running count ...
      lrank[e] = prev + before;
      if (valid && before == 0) table[wave * 256 + d] = prev + cnt;   // ... then the group leader advances it
    }
  }
  __syncthreads();
  if (tid < 256) {                                        // per digit: exclusive prefix over the 16 waves
    uint32_t run = 0;
#pragma unroll
    for (int w2 = 0; w2 < TK_WAVES; ++w2) {
      const uint32_t c = table[w2 * 256 + tid];
      table[w2 * 256 + tid] = run;
      run += c;
    }
    tot[tid] = run;
  }
  __syncthreads();
  if (tid < 64) {                                         // exclusive prefix over the 256 digit totals
    const uint4 h = reinterpret_cast<const uint4*>(tot)[tid];
    const uint32_t own = h.x + h.y + h.z + h.w;
    const uint32_t excl = wave_incl_scan_u32(own) - own;
    reinterpret_cast<uint4*>(tot)[tid] = make_uint4(excl, excl + h.x, excl + h.x + h.y, excl + h.x + h.y + h.z);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < ept) {
      const int i = wave * ept * 64 + e * 64 + lane;
      if (i < k) dst[tot[dig[e]] + table[wave * 256 + dig[e]] + lrank[e]] = comp[e];
    }
  }
  __syncthreads();
}

// ADA (round 5, pkv_ada_select): the launch also leaves, per row, the head's ADAPTIVE list behind (p.list_out): the winners'
// raw scores in output order are emitted with the indices, and an epilogue - one key per thread, on the 32 CUs the heads
// already occupy - turns them into what Ada-SnapKV's budget step (:709-719) works on: ratio = sum(first `base`) / sum(row)
// (:710, model-dtype roundings as there; the row total from finalize_kernel's fp64 partials), adaptive = round(raw * ratio)
// (:711), stored as order-preserving keys.  In the single-workgroup budget kernel the same arithmetic was 390 of 510
// instructions per wave of its longest phase (16 waves on ONE CU: every instruction costs 16 cycles there).
// A template parameter: the plain selection carries none of it.
// NI (round 6): 512-key chunks per wave as a compile-time constant (1, 2 or 4; 0 = read from p.Lw: rows beyond 32 768 keys - eight
// chunks fully unrolled need more than the kernel's 128 registers).  Every loop over a lane's
// chunks is `for j < 8: if (j < niter)`: with a run-time niter that is eight scalar compare-and-branch pairs per loop, ~20 loops,
// and eight condition masks the compiler keeps in scalar registers for all of them (the kernel sat at its 104-register limit and
// spilled into the small-k path's loops once the zero-tied rows were added).  S = 4k / 8k / 16k / 32k rows are 1 / 1 / 2 / 4 chunks.
template <typename T, bool ADA, int NI>
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(TopkParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int vrow = blockIdx.x;         // output row
  int row = vrow, seg_off = 0, Lrow = p.L;
  if (p.nseg > 1) {                    // long rows: this workgroup owns one segment, see TopkParams::nseg
    row = vrow / p.nseg;
    seg_off = (vrow - row * p.nseg) * p.seg_len;
    Lrow = min(p.seg_len, p.L - seg_off);
  }
  const int L = Lrow;
  int k = p.k;
  if (p.k_per_row) { const int kr = p.k_per_row[row]; k = kr < k ? kr : k; }
  if (k <= 0) return;
  if (k > L) k = L;

  const int Lw = p.Lw;
  const int Lk = TK_WAVES * Lw;
  uint16_t* keys = reinterpret_cast<uint16_t*>(smem);
  uint32_t* X = reinterpret_cast<uint32_t*>(smem + (size_t)2 * Lk);        // counters, later `sel`
  const int xwords = p.kpad > TK_CNT_WORDS ? p.kpad : TK_CNT_WORDS;
  uint32_t* hist = X + xwords;
  uint32_t* miscu = hist + 256;
  int* misc = reinterpret_cast<int*>(miscu);
  uint32_t* wcnt = miscu + 16;
  uint32_t* X2 = miscu + 64;                                               // only if p.dual
  const bool dual = p.dual != 0;

  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)row * p.scores_stride + seg_off;
  int32_t* const orow = p.idx_out + (int64_t)vrow * p.idx_stride;
  uint16_t* const lrow = ADA ? reinterpret_cast<uint16_t*>(p.list_out) + (int64_t)vrow * p.list_stride : nullptr;
  // output position `pos` <- the winner with composite key<<16 | (0xffff - index)
  auto emit = [&](int pos, uint32_t comp) {
    orow[pos] = seg_off + (int32_t)(0xffffu - (comp & 0xffffu));
    if (ADA) lrow[pos] = key_to_raw<T>(comp >> 16);
  };
  // ADA: the row total (:710), every wave on its own from finalize_kernel's partials - long before the epilogue needs it
  double ada_rowsum = 0.0;
  if (ADA && p.ada_normalize) {
    const double* part = p.rowsum_part + (int64_t)vrow * p.rowsum_np;
    for (int j = lane; j < p.rowsum_np; j += 64) ada_rowsum += part[j];
    ada_rowsum = wave_sum_f64(ada_rowsum);
  }
  const bool vec_ok = ((p.scores_stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.scores) & 15) == 0);
  const uint32_t inc = lane < 32 ? 1u : 65536u;
  const int cslot = lane & 31;
  const int niter = NI > 0 ? NI : Lw / 512;

#define PKV_STAMP(i) do { if (PKV_TRACE(p) && tid == 0 && row == 0) PKV_TRACE(p)[i] = (unsigned long long)clock64(); } while (0)
  PKV_STAMP(0);
  const unsigned long long t_start = PKV_WGTRACE(p) ? wall_clock64() : 0ull;
  // ---- pass A: HBM/L2 -> ordered keys in LDS, histogram of the high byte.  All (<= 8) 16-B loads of a
  //      lane are issued first; the counter arrays are zeroed while they are in flight. ----
  const int nch = (L + 7) >> 3;                                   // 8-key chunks in the row
  const bool fast_ok = k <= TK_FAST_K && 2 * k <= nch;            // prefilter applicable (see below)
  const bool use_cmax = fast_ok && p.cmax != nullptr && vec_ok;   // chunk maxima precomputed by finalize_kernel
  U4 raw[8];
  uint16_t cm[8];
  auto load_all = [&]() {
    if (vec_ok) {
      // stride % 8 == 0 and stride >= L  =>  stride >= roundup(L, 8): the 16-B load of the last, partial
      // chunk stays inside the row; out-of-range lanes re-read the row start and are masked below.
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < niter) {
          const int base = wave * Lw + j * 512 + lane * 8;
          raw[j].v = *reinterpret_cast<const uint4*>(src + (base < L ? base : 0));
        }
      }
    } else {
      // unaligned rows (external callers only): unconditional clamped 2-byte loads, masked below
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < niter) {
          const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) raw[j].h[e] = src[base + e < L ? base + e : L - 1];
        }
      }
    }
  };
  if (use_cmax) {
    const uint16_t* cmp = reinterpret_cast<const uint16_t*>(p.cmax) + (int64_t)row * p.cmax_stride + (seg_off >> 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < niter) {
        const int c = wave * (Lw >> 3) + j * 64 + lane;             // chunk (j, lane) of this wave
        cm[j] = cmp[c < nch ? c : 0];
      }
    }
  }
  // the whole row goes in flight now in every path: with the prefilter only the chunks reaching x* are looked at
  // later, but fetching them after x* is known would put a second cold round trip (~2.5 us) on the critical path
  load_all();
  for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) { X[i] = 0; if (dual) X2[i] = 0; }
  __syncthreads();
  if (PKV_TRACE(p) && tid == 0 && row == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PKV_TRACE(p)[7] = (unsigned long long)clock64(); }
  U4 kreg[8];     // this lane's ordered keys (niter chunks of 8), kept in registers for every later pass
  auto transform_all = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < niter) {
        const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) kreg[j].w[q] = order_key_pk<T>(raw[j].w[q]);
        if (base + 8 > L) {                       // partial or out-of-range chunk: padding keys are 0
#pragma unroll
          for (int e = 0; e < 8; ++e) if (base + e >= L) kreg[j].h[e] = 0;
        }
        *reinterpret_cast<uint4*>(keys + base) = kreg[j].v;
      }
    }
  };
  // ADA epilogue (all three selection paths end here): raw scores of the list -> adaptive keys, in place
  auto ada_epilogue = [&]() {
    __syncthreads();                                   // every emit of this workgroup is visible to it (one CU, write-through L1)
    float ratio = 1.0f;
    if (p.ada_normalize) {
      double st = 0.0;
      for (int i = tid; i < p.ada_base && i < k; i += TK_THREADS) st += (double)Elem<T>::to_f32(lrow[i]);
      st = wave_sum_f64(st);
      double* red = reinterpret_cast<double*>(hist);   // the histogram scratch is free by now (64-byte aligned)
      if (lane == 0) red[wave] = st;
      __syncthreads();
      double t = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < TK_WAVES; ++w2) t += red[w2];
      const float tq = Elem<T>::to_f32(Elem<T>::from_f32((float)t));            // .sum() result in model dtype (:710)
      const float aq = Elem<T>::to_f32(Elem<T>::from_f32((float)ada_rowsum));
      ratio = Elem<T>::to_f32(Elem<T>::from_f32(tq / aq));                      // model-dtype division (:710)
    }
    for (int i = tid; i < k; i += TK_THREADS) {
      uint16_t h = lrow[i];
      if (p.ada_normalize) h = Elem<T>::from_f32(Elem<T>::to_f32(h) * ratio);   // adaptive_attn_score * ratio_weight (:711)
      lrow[i] = (uint16_t)order_key<T>(h);
    }
  };
  const bool fast2 = fast_ok && dual && vec_ok && p.algo == 1;      // one-level fast path, see below
  if (!use_cmax && !fast2) transform_all();

  // ---- one-level fast path (algo 1, small k).  Cost model: 1024 threads on one CU = every wave-instruction costs
  //      ~16 cycles of wall time, so nothing here touches all 32 keys of a lane.
  //      (1) 13-bit histogram of the chunk maxima (4 per lane at S = 32k) -> x* = lower edge of the bin in which the
  //          count of chunk maxima from the top reaches k: at least k keys are >= x*, so the top-k are among the keys >= x*;
  //      (2) the chunks whose maximum reaches x* (about k of 4096) are staged in LDS by their owners; all threads then
  //          look at the 8 keys of each staged chunk (dense), candidates (key >= x*) stay in registers as composites
  //          key<<16 | (0xffff - index) and are counted per key bucket;
  //      (3) counting sort by bucket (descending key), then every candidate ranks itself inside its bucket by comparing
  //          composites: rank == output position, rank < k == selected.  Canonical order (value desc, index asc) by
  //          construction.  Heavy ties (too many staged chunks / candidates) fall through to the exact full path. ----
  if (fast2) {
    uint32_t* H13 = X;                                        // [8192], zeroed above
    uint32_t* HB = X2;                                        // [2048] bucket counters, zeroed above
    uint32_t* tmpl = X2 + TK2_NB;                             // [4096] candidates in bucket order
    uint4* stage_raw = reinterpret_cast<uint4*>(smem);        // [pmax] raw 16-B chunks (the `keys` region is unused here)
    const int pmax = Lw < TK2_PMAX ? Lw : TK2_PMAX;           // 18 * pmax <= 32 * Lw bytes
    uint16_t* stage_id = reinterpret_cast<uint16_t*>(smem + (size_t)16 * pmax);
    uint32_t gm[8];
    if (tid == 0) miscu[4] = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < niter) {
        const int c = wave * (Lw >> 3) + j * 64 + lane;
        uint32_t m = 0;
        if (use_cmax) {
          m = order_key<T>(cm[j]);
        } else {
          const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t kk = order_key_pk<T>(raw[j].w[q]);
            const uint32_t lo = (base + 2 * q < L) ? (kk & 0xffffu) : 0u, hi = (base + 2 * q + 1 < L) ? (kk >> 16) : 0u;
            m = m > lo ? m : lo;
            m = m > hi ? m : hi;
          }
        }
        gm[j] = c < nch ? m : 0u;
        if (c < nch) atomicAdd(&H13[m >> 3], 1u);
      }
    }
    __syncthreads();
    // bin b* with  above(b*) < k <= above(b*) + H13[b*]  (above = count in higher bins).  A lane owns bins 8*tid .. 8*tid+7,
    // four lanes a coarse bin of 32; the 256 coarse sums go to `hist`, and every wave then finds the coarse bin and, inside
    // it, the fine bin on its own (no LDS hand-off of the result, no third barrier).
    uint32_t xstar;
    {
      const uint4 h0 = reinterpret_cast<const uint4*>(H13)[2 * tid], h1 = reinterpret_cast<const uint4*>(H13)[2 * tid + 1];
      uint32_t s8 = h0.x + h0.y + h0.z + h0.w + h1.x + h1.y + h1.z + h1.w;
      s8 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s8, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      s8 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s8, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
      if ((tid & 3) == 0) hist[tid >> 2] = s8;
      __syncthreads();
      const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
      const uint32_t hv[4] = {h.x, h.y, h.z, h.w};
      const uint32_t own = h.x + h.y + h.z + h.w;
      const uint32_t incl = wave_incl_scan_u32(own);
      uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) - incl;     // count in the bins of higher lanes
      uint32_t cb = 0, cab = 0;
      bool hit = false;
#pragma unroll
      for (int i = 3; i >= 0; --i) {
        if (above < (uint32_t)k && (uint32_t)k <= above + hv[i]) { hit = true; cb = (uint32_t)(lane * 4 + i); cab = above; }
        above += hv[i];
      }
      const int src = __builtin_ctzll(__ballot(hit));                                    // exactly one lane hits (nch >= k chunks)
      cb = (uint32_t)__builtin_amdgcn_readlane((int)cb, src);
      cab = (uint32_t)__builtin_amdgcn_readlane((int)cab, src);
      const uint32_t cf = lane < 32 ? H13[cb * 32 + lane] : 0u;
      const uint32_t incl2 = wave_incl_scan_u32(cf);
      const uint32_t above2 = cab + (uint32_t)__builtin_amdgcn_readlane((int)incl2, 63) - incl2;
      const bool hit2 = above2 < (uint32_t)k && (uint32_t)k <= above2 + cf;
      const int src2 = __builtin_ctzll(__ballot(hit2));
      xstar = (cb * 32 + (uint32_t)src2) << 3;
    }
    PKV_STAMP(1);
    // stage the passing chunks: one LDS atomic per wave reserves the slots of all its passing chunks
    {
      // (the pass masks are recomputed in the second loop instead of being kept: 8 live 64-bit masks push this kernel over
      // its scalar-register budget and every one of them then travels through v_writelane / v_readlane)
      uint32_t npass = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < niter) npass += (uint32_t)__popcll(__ballot(gm[j] != 0u && gm[j] >= xstar));
      uint32_t slot0 = 0;
      if (lane == 0 && npass) slot0 = atomicAdd(&miscu[4], npass);
      slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
      PKV_STAMP(4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < niter) {
          const bool pass = gm[j] != 0u && gm[j] >= xstar;
          const uint64_t mk = __ballot(pass);
          if (mk != 0ull) {
            const int c = wave * (Lw >> 3) + j * 64 + lane;
            const uint32_t q = slot0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (pass && q < (uint32_t)pmax) { stage_raw[q] = raw[j].v; stage_id[q] = (uint16_t)c; }
            slot0 += (uint32_t)__popcll(mk);
          }
        }
      }
      PKV_STAMP(13);
    }
    __syncthreads();
    PKV_STAMP(2);
    const uint32_t P = miscu[4];
    bool ok2 = P <= (uint32_t)pmax;
    if (ok2) {
      // dense look at the 8 keys of every staged chunk; candidates stay in registers
      constexpr int SH = sizeof(typename KeyShift<T>::tag) == 1 ? 0 : 3;          // bf16: one key per bucket; fp16: 8
      const uint16_t* stage_h = reinterpret_cast<const uint16_t*>(stage_raw);
      const int nslots = (int)P * 8;
      const int ni = (nslots + TK_THREADS - 1) / TK_THREADS;          // workgroup-uniform trip count (<= 16)
      uint32_t comp[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        comp[i] = 0u;
        const int sl = tid + i * TK_THREADS;
        if (i < ni && sl < nslots) {
          const uint32_t key = order_key<T>(stage_h[sl]);
          const uint32_t idx = (uint32_t)stage_id[sl >> 3] * 8u + (uint32_t)(sl & 7);
          if ((int)idx < L && key >= xstar) {
            comp[i] = (key << 16) | (0xffffu - idx);
            const uint32_t db = (key - xstar) >> SH;
            atomicAdd(&HB[TK2_NB - 1 - (db < (uint32_t)TK2_NB - 1 ? db : (uint32_t)TK2_NB - 1)], 1u);
          }
        }
      }
      __syncthreads();
      // exclusive prefix over the 2048 buckets (ascending bucket = descending key); lane owns buckets 2*tid, 2*tid+1
      const uint2 hb = reinterpret_cast<const uint2*>(HB)[tid];
      const uint32_t own = hb.x + hb.y;
      const uint32_t incl = wave_incl_scan_u32(own);
      // bit 31 of the wave's total: one of its buckets is too crowded to be ranked quadratically (TK2_BMAX)
      const uint32_t crowded = __ballot(hb.x > (uint32_t)TK2_BMAX || hb.y > (uint32_t)TK2_BMAX) != 0ull ? 0x80000000u : 0u;
      if (lane == 63) wcnt[16 + wave] = incl | crowded;
      __syncthreads();
      // cross-wave offsets: lane w reads the total of wave w, one wave scan gives every wave its base and the grand total
      const uint32_t wraw = lane < TK_WAVES ? wcnt[16 + lane] : 0u;
      const bool crowd = __ballot((wraw >> 31) != 0u) != 0ull;
      const uint32_t wt = wraw & 0x7fffffffu;
      const uint32_t wincl = wave_incl_scan_u32(wt);
      const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
      const uint32_t lower = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wt), wave);
      if (PKV_TRACE(p) && tid == 0 && row == 0) PKV_TRACE(p)[15] = C;
      ok2 = C <= (uint32_t)TK2_CMAX && C >= (uint32_t)k && !crowd;
      if (ok2) {
        const uint32_t bs = lower + incl - own;
        reinterpret_cast<uint2*>(HB)[tid] = make_uint2(bs, bs + hb.x);
        __syncthreads();
        PKV_STAMP(3);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < ni && comp[i] != 0u) {
            const uint32_t db = ((comp[i] >> 16) - xstar) >> SH;
            const uint32_t dst = atomicAdd(&HB[TK2_NB - 1 - (db < (uint32_t)TK2_NB - 1 ? db : (uint32_t)TK2_NB - 1)], 1u);
            tmpl[dst] = comp[i];
          }
        }
        if (tid < 4) tmpl[C + tid] = 0u;          // padding behind the list for the 16-byte reads of the ranking (inside X2's spare words)
        __syncthreads();
        PKV_STAMP(5);
        // HB[d] is now the END of bucket d; rank inside the bucket by comparing composites (unique).  Four list entries per
        // LDS read (round 5): the scan starts at the 16-byte boundary at or below the bucket's first entry - whatever lies
        // between that boundary and the bucket belongs to EARLIER buckets, i.e. larger composites, and is counted exactly as
        // the `st` entries in front of the bucket were; whatever follows the bucket's end is smaller (or the zero padding
        // behind the list) and never counted.  Real max-pooled scores put ~30 equal keys into a bucket: this loop was
        // 6800 of the 25 600 cycles at k = 512 and 1900 of 17 600 at k = 120 (profiles/r05/topk_k_probe_debug.json).
        for (int i = tid; i < (int)C; i += TK_THREADS) {
          const uint32_t mine = tmpl[i];
          const uint32_t db = ((mine >> 16) - xstar) >> SH;
          const uint32_t d = TK2_NB - 1 - (db < (uint32_t)TK2_NB - 1 ? db : (uint32_t)TK2_NB - 1);
          const uint32_t st = d ? HB[d - 1] : 0u, en = HB[d];
          const uint32_t a0 = st & ~3u;
          uint32_t rank = a0;
          for (uint32_t jj = a0; jj < en; jj += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(tmpl + jj);
            rank += (uint32_t)(v.x > mine) + (uint32_t)(v.y > mine) + (uint32_t)(v.z > mine) + (uint32_t)(v.w > mine);
          }
          if (rank < (uint32_t)k) emit((int)rank, mine);
        }
        PKV_STAMP(6);
        if (PKV_WGTRACE(p) && tid == 0) { PKV_WGTRACE(p)[2 * (131072 + row)] = t_start; PKV_WGTRACE(p)[2 * (131072 + row) + 1] = wall_clock64(); }
        if (ADA) ada_epilogue();
        return;
      }
    }
    // ---- Round 6 - rows that are mostly ZERO (fp16 at long context: most window probabilities underflow; fewer than k chunks
    //      hold anything above zero's histogram bin, and the k-th value is zero or a subnormal tied thousands of times).  x* then
    //      sits at or below zero's bin, every chunk passes the prefilter above and the attempt gives up - such rows used to
    //      take the full path (20 us instead of 7).  They are decided here with the same machinery and the EXACT prefilter
    //      "key > zero": the chunks holding a positive score (a few dozen) are staged, their positive keys counting-sorted by
    //      key bucket (32 codes wide: the positive scores of a whole row, not a band under the maximum) and ranked inside their
    //      bucket - rank == output position; all of them are selected when there are fewer than k - and the rest of the
    //      selection is the lowest-index zeros, which is what (value desc, index asc) takes: ONE wave at a time walks its chunks
    //      in index order (the first wave's first chunk usually holds them all; the other waves wait at a barrier, so its
    //      instructions cost 4 cycles, not 16).  Too many positive keys, or too few zeros (negative scores): the full path.
    //      This code sits BEHIND the ordinary attempt on purpose: in front of it, it cost ordinary rows 0.25-0.5 us. ----
    {
      const uint32_t zkey = order_key<T>((uint16_t)0);
      if (__builtin_expect(xstar <= zkey, 0)) {                      // cold: the register allocator keeps its spill code out of the attempt above
        // bit j: chunk (j, lane) holds a key above zero.  Not carried over from the attempt above (one more live register and two
        // more instructions per chunk there cost ordinary rows 0.2 us): the chunk maxima are read again where finalize_kernel
        // left them (L2-hot), or taken from the raw scores (callers without chunk maxima).
        uint32_t pos_bits = 0;
        if (use_cmax) {
          const uint16_t* cmp = reinterpret_cast<const uint16_t*>(p.cmax) + (int64_t)row * p.cmax_stride + (seg_off >> 3);
          uint16_t cz[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < niter) {
              const int c = wave * (Lw >> 3) + j * 64 + lane;
              cz[j] = cmp[c < nch ? c : 0];
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < niter) {
              const int c = wave * (Lw >> 3) + j * 64 + lane;
              pos_bits |= (c < nch && order_key<T>(cz[j]) > zkey ? 1u : 0u) << j;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < niter) {
              const int base = wave * Lw + j * 512 + lane * 8;
              uint32_t m = 0;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                uint32_t rw = raw[j].w[q];
                asm volatile("" : "+v"(rw));                       // recomputed, not kept: see the zero walk below
                const uint32_t kk = order_key_pk<T>(rw);
                const uint32_t lo = (base + 2 * q < L) ? (kk & 0xffffu) : 0u, hi = (base + 2 * q + 1 < L) ? (kk >> 16) : 0u;
                m = m > lo ? m : lo;
                m = m > hi ? m : hi;
              }
              pos_bits |= (m > zkey ? 1u : 0u) << j;
            }
          }
        }
        __syncthreads();                                           // the attempt above is done with miscu[4] and HB
        if (tid == 0) miscu[4] = 0;
        reinterpret_cast<uint2*>(HB)[tid] = make_uint2(0u, 0u);
        __syncthreads();
        {
          uint32_t npass = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < niter) npass += (uint32_t)__popcll(__ballot((pos_bits >> j) & 1u));
          uint32_t slot0 = 0;
          if (lane == 0 && npass) slot0 = atomicAdd(&miscu[4], npass);
          slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < niter) {
              const bool pass = (pos_bits >> j) & 1u;
              const uint64_t mk = __ballot(pass);
              if (mk != 0ull) {
                const int c = wave * (Lw >> 3) + j * 64 + lane;
                const uint32_t q = slot0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                if (pass && q < (uint32_t)pmax) { stage_raw[q] = raw[j].v; stage_id[q] = (uint16_t)c; }
                slot0 += (uint32_t)__popcll(mk);
              }
            }
          }
        }
        __syncthreads();
        const uint32_t Pz = miscu[4];
        if (Pz <= 512u && Pz <= (uint32_t)pmax) {                  // at most 4096 staged keys: four per thread
          constexpr int ZSH = 5;                                   // (65535 - zkey) >> 5 < 2048 buckets
          const uint32_t xs = zkey + 1u;
          const uint16_t* stage_h = reinterpret_cast<const uint16_t*>(stage_raw);
          const int nslots = (int)Pz * 8;
          uint32_t comp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            comp[i] = 0u;
            const int sl = tid + i * TK_THREADS;
            if (sl < nslots) {
              const uint32_t key = order_key<T>(stage_h[sl]);
              const uint32_t idx = (uint32_t)stage_id[sl >> 3] * 8u + (uint32_t)(sl & 7);
              if ((int)idx < L && key >= xs) {
                comp[i] = (key << 16) | (0xffffu - idx);
                atomicAdd(&HB[TK2_NB - 1 - (int)((key - xs) >> ZSH)], 1u);
              }
            }
          }
          __syncthreads();
          const uint2 hb = reinterpret_cast<const uint2*>(HB)[tid];
          const uint32_t own = hb.x + hb.y;
          const uint32_t incl = wave_incl_scan_u32(own);
          const uint32_t crowded = __ballot(hb.x > (uint32_t)TK2_BMAX || hb.y > (uint32_t)TK2_BMAX) != 0ull ? 0x80000000u : 0u;
          if (lane == 63) wcnt[16 + wave] = incl | crowded;
          __syncthreads();
          const uint32_t wraw = lane < TK_WAVES ? wcnt[16 + lane] : 0u;
          const bool crowd = __ballot((wraw >> 31) != 0u) != 0ull;
          const uint32_t wt = wraw & 0x7fffffffu;
          const uint32_t wincl = wave_incl_scan_u32(wt);
          const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
          const uint32_t lower = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wt), wave);
          if (C <= (uint32_t)TK2_CMAX && !crowd) {
            const uint32_t bs = lower + incl - own;
            reinterpret_cast<uint2*>(HB)[tid] = make_uint2(bs, bs + hb.x);
            if (tid == 0) miscu[6] = C < (uint32_t)k ? (uint32_t)k - C : 0u;     // zeros still to be found
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (comp[i] != 0u) {
                const uint32_t dst = atomicAdd(&HB[TK2_NB - 1 - (int)(((comp[i] >> 16) - xs) >> ZSH)], 1u);
                tmpl[dst] = comp[i];
              }
            }
            if (tid < 4) tmpl[C + tid] = 0u;
            __syncthreads();
            for (int i = tid; i < (int)C; i += TK_THREADS) {       // HB[d] is now the END of bucket d: rank inside the bucket
              const uint32_t mine = tmpl[i];
              const uint32_t d = TK2_NB - 1 - (((mine >> 16) - xs) >> ZSH);
              const uint32_t st = d ? HB[d - 1] : 0u, en = HB[d];
              const uint32_t a0 = st & ~3u;
              uint32_t rank = a0;
              for (uint32_t jj = a0; jj < en; jj += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(tmpl + jj);
                rank += (uint32_t)(v.x > mine) + (uint32_t)(v.y > mine) + (uint32_t)(v.z > mine) + (uint32_t)(v.w > mine);
              }
              if (rank < (uint32_t)k) emit((int)rank, mine);
            }
            // positions C .. k-1: the first k - C zeros in index order, one wave at a time
            const uint32_t need = C < (uint32_t)k ? (uint32_t)k - C : 0u;
            for (int w2 = 0; w2 < TK_WAVES; ++w2) {
              const uint32_t rem = miscu[6];                       // workgroup-uniform
              if (rem == 0u) break;
              if (wave == w2) {
                uint32_t found = need - rem;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  if (j < niter && found < need) {                 // wave-uniform
                    int base = wave * Lw + j * 512 + lane * 8;
                    asm volatile("" : "+v"(base));                 // nothing of this body is hoisted out of the wave loop (it was: 90 spilled registers)
                    uint32_t m = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                      uint32_t rw = raw[j].w[q];
                      asm volatile("" : "+v"(rw));
                      const uint32_t kk = order_key_pk<T>(rw);
                      m |= ((kk & 0xffffu) == zkey && base + 2 * q < L ? 1u : 0u) << (2 * q);
                      m |= ((kk >> 16) == zkey && base + 2 * q + 1 < L ? 1u : 0u) << (2 * q + 1);
                    }
                    const uint32_t cnt = (uint32_t)__popc(m);
                    const uint32_t inc2 = wave_incl_scan_u32(cnt);
                    uint32_t r = found + inc2 - cnt;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                      if ((m >> e) & 1u) {
                        if (r < need) emit((int)(C + r), (zkey << 16) | (0xffffu - (uint32_t)(base + e)));
                        ++r;
                      }
                    }
                    found += (uint32_t)__builtin_amdgcn_readlane((int)inc2, 63);
                  }
                }
                if (lane == 0) miscu[6] = found >= need ? 0u : need - found;
              }
              __syncthreads();
            }
            if (miscu[6] == 0u) {
              PKV_STAMP(6);
              if (PKV_WGTRACE(p) && tid == 0) { PKV_WGTRACE(p)[2 * (131072 + row)] = t_start; PKV_WGTRACE(p)[2 * (131072 + row) + 1] = wall_clock64(); }
              if (ADA) ada_epilogue();
              return;
            }
          }
        }
      }
    }
    // heavy ties: exact full path.  Its key table and counters start from scratch.
    __syncthreads();
    transform_all();
    for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) { X[i] = 0; X2[i] = 0; }
    __syncthreads();
  }

  // ---- fast path (small k): the k-th largest of the per-chunk maxima is a lower bound x* of the
  //      selection threshold (at least k keys are >= x*), and usually only a few hundred keys pass it.
  //      Select x* from 1024*niter chunk maxima (<= 8 LDS atomics per lane instead of 8*niter*... per
  //      key), compact the candidates in index order, and rank them by counting: rank < k <=> selected,
  //      and the rank IS the output position.  Falls back to the full radix select when too many
  //      keys tie at or above x*. ----
  if (fast_ok && !fast2) {
    uint32_t gm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < niter) {
        uint32_t m = 0;
        if (use_cmax) {
          const int c = wave * (Lw >> 3) + j * 64 + lane;
          m = c < nch ? order_key<T>(cm[j]) : 0u;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) m = m > kreg[j].h[e] ? m : (uint32_t)kreg[j].h[e];
        }
        gm[j] = m;
        atomicAdd(&X[(m >> 8) * 32 + cslot], inc);
      }
    }
    __syncthreads();
    select_bin(X, hist, (uint32_t)k, &misc[0], &misc[1], tid);
    __syncthreads();
    PKV_STAMP(1);
    const uint32_t fb1 = (uint32_t)misc[0];
    const int fabove = misc[1];
    uint32_t* XF = dual ? X2 : X;
    if (!dual) {
      for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) X[i] = 0;
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < niter && (gm[j] >> 8) == fb1) atomicAdd(&XF[(gm[j] & 255u) * 32 + cslot], inc);
    __syncthreads();
    select_bin(XF, hist, (uint32_t)(k - fabove), &misc[2], &misc[3], tid);
    __syncthreads();
    PKV_STAMP(2);
    const uint32_t xstar = (fb1 << 8) | (uint32_t)misc[2];
    if (use_cmax) {
      // only chunks whose maximum reaches x* can hold candidates (their keys arrived with the first round trip)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < niter) {
          const int base = wave * Lw + j * 512 + lane * 8;
          kreg[j].v = make_uint4(0, 0, 0, 0);
          if (gm[j] >= xstar) {
#pragma unroll
            for (int q = 0; q < 4; ++q) kreg[j].w[q] = order_key_pk<T>(raw[j].w[q]);
            if (base + 8 > L) {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (base + e >= L) kreg[j].h[e] = 0;
            }
          }
        }
      }
    }
    uint32_t cl = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < niter) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cl += kreg[j].h[e] >= xstar;
      }
    const uint32_t cw = wave_sum_u32(cl);
    if (lane == 0) wcnt[wave] = cw;
    __syncthreads();
    uint32_t C = 0, cbase = 0;
#pragma unroll
    for (int w2 = 0; w2 < TK_WAVES; ++w2) { const uint32_t c = wcnt[w2]; C += c; cbase += (w2 < wave) ? c : 0u; }
    PKV_STAMP(3);
    if (PKV_TRACE(p) && tid == 0 && row == 0) PKV_TRACE(p)[15] = C;
    // rank = number of list entries with a larger composite; entries with rank < k are the selection and
    // the rank is the output position.  All 1024 threads take part: an entry is shared by G <= 4 adjacent
    // lanes, each scanning every G-th block of 32 composites (8 independent 16-B broadcast reads per step).
    auto rank_store = [&](uint32_t* list, int n) {
      int Cp = 64;
      while (Cp < n) Cp <<= 1;
      const int G = (TK_THREADS / Cp) > 4 ? 4 : (TK_THREADS / Cp);
      const int blocks = (n + 31) >> 5;
      const int steps = (blocks + G - 1) / G;
      const int cpad = steps * G * 32;
      for (int i = n + tid; i < cpad; i += TK_THREADS) list[i] = 0;
      __syncthreads();
      const int ci = tid / G, g = tid - ci * G;
      const uint32_t mine = ci < n ? list[ci] : 0xffffffffu;
      int rank = 0;
      const uint4* c4 = reinterpret_cast<const uint4*>(list);
      for (int it = 0; it < steps; ++it) {
        const uint4* blk = c4 + (it * G + g) * 8;
        uint4 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = blk[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) rank += (a[q].x > mine) + (a[q].y > mine) + (a[q].z > mine) + (a[q].w > mine);
      }
      for (int o = 1; o < G; o <<= 1) rank += __shfl_xor(rank, o, 64);
      if (g == 0 && ci < n && rank < k) emit(rank, mine);
    };
    if (C <= TK_FAST_C || (C <= TK_MID_C && dual)) {
      uint32_t* cand = X;                       // the stage-1 counters in X are no longer needed
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < niter) {
          uint32_t cj = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) cj += kreg[j].h[e] >= xstar;
          if (__ballot(cj != 0) == 0ull) continue;
          const uint32_t incl = wave_incl_scan_u32(cj);
          uint32_t pos = cbase + incl - cj;
          const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t key = kreg[j].h[e];
            if (key >= xstar) cand[pos++] = (key << 16) | (0xffffu - (uint32_t)(base + e));
          }
          cbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
      }
      if (C <= TK_FAST_C) {
        PKV_STAMP(5);
        rank_store(cand, (int)C);
      } else {
        // 448 < C <= 4096 candidates, in index order: a STABLE sort by key alone is the canonical order, and
        // its first k entries are the selection - two LSD radix passes (match-any ballots, no atomics) replace
        // a two-level select inside the list + compaction + ranking (measured 13k -> see profiles, shader clocks)
        __syncthreads();                                        // candidate list complete
        uint32_t* table = X2;                                   // [16][256]
        uint32_t* tmp = X2 + TK_WAVES * 256;                    // [4096]
        const int ept = ((int)C + TK_THREADS - 1) / TK_THREADS;
        radix_pass(cand, tmp, table, hist, (int)C, ept, 0, tid);
        radix_pass(tmp, cand, table, hist, (int)C, ept, 1, tid);
        PKV_STAMP(5);
        for (int i = tid; i < k; i += TK_THREADS) emit(i, cand[i]);
      }
      PKV_STAMP(6);
      if (PKV_WGTRACE(p) && tid == 0) { PKV_WGTRACE(p)[2 * (131072 + row)] = t_start; PKV_WGTRACE(p)[2 * (131072 + row) + 1] = wall_clock64(); }
      if (ADA) ada_epilogue();
      return;
    }
    // too many keys at or above x* (heavy ties): full path.  Its counters must start from zero.
    if (use_cmax) transform_all();       // the row is already in registers
    __syncthreads();
    for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) { X[i] = 0; if (dual) X2[i] = 0; }
    __syncthreads();
  }

  // ---- full path, pass A: histogram of the high byte (one LDS atomic per key: ~16 cycles per
  //      wave-instruction whatever the number of active lanes) ----
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < niter) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t key = kreg[j].h[e];
        if (key != 0u) atomicAdd(&X[(key >> 8) * 32 + cslot], inc);
      }
    }
  }
  __syncthreads();
  PKV_STAMP(1);
  select_bin(X, hist, (uint32_t)k, &misc[0], &misc[1], tid);
  __syncthreads();
  PKV_STAMP(2);
  const uint32_t b1 = (uint32_t)misc[0];
  const int n_above1 = misc[1];

  // ---- pass B: histogram of the low byte inside bin b1 ----
  uint32_t* XB = dual ? X2 : X;
  if (!dual) {
    for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) X[i] = 0;
    __syncthreads();
  }
  for (int j = 0; j < niter; ++j) {
    const int base = wave * Lw + j * 512 + lane * 8;
    U4 kv;
    kv.v = *reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t key = kv.h[e];
      if (key != 0u && (key >> 8) == b1) atomicAdd(&XB[(key & 255u) * 32 + cslot], inc);
    }
  }
  __syncthreads();
  select_bin(XB, hist, (uint32_t)(k - n_above1), &misc[2], &misc[3], tid);
  __syncthreads();
  PKV_STAMP(3);
  const uint32_t Tkey = (b1 << 8) | (uint32_t)misc[2];
  const int n_gt = n_above1 + misc[3];          // keys strictly above the threshold
  const int n_eq_take = k - n_gt;               // lowest-index keys equal to the threshold

  // ---- pass C: per-wave (gt, eq) counts so that compaction keeps index order ----
  {
    uint32_t cg = 0, ce = 0;
    for (int j = 0; j < niter; ++j) {
      const int base = wave * Lw + j * 512 + lane * 8;
      U4 kv;
      kv.v = *reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t key = kv.h[e];
        cg += key > Tkey;
        ce += key == Tkey;
      }
    }
    cg = wave_sum_u32(cg);
    ce = wave_sum_u32(ce);
    if (lane == 0) { wcnt[wave] = cg; wcnt[16 + wave] = ce; }
  }
  __syncthreads();   // also: every read of the counters in X is done; X becomes `sel`
  PKV_STAMP(4);
  uint32_t run_g = 0, run_e = 0;
  for (int w2 = 0; w2 < wave; ++w2) { run_g += wcnt[w2]; run_e += wcnt[16 + w2]; }
  uint32_t* sel = X;

  // ---- pass D: compaction in index order.  composite = key<<16 | (0xffff - index): descending
  //      composite order == (value desc, index asc).  Chunks without any candidate are skipped. ----
  for (int j = 0; j < niter; ++j) {
    const int base = wave * Lw + j * 512 + lane * 8;
    U4 kv;
    kv.v = *reinterpret_cast<const uint4*>(keys + base);
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t key = kv.h[e];
      cg += key > Tkey;
      ce += key == Tkey;
    }
    if (__ballot((cg | ce) != 0) == 0ull) continue;            // wave-uniform
    const uint32_t packed = (ce << 16) | cg;                   // per-iteration totals <= 512 each
    const uint32_t incl = wave_incl_scan_u32(packed);
    const uint32_t excl = incl - packed;
    uint32_t og = run_g + (excl & 0xffffu);
    uint32_t oe = run_e + (excl >> 16);
    if (cg | ce) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t key = kv.h[e];
        const uint32_t comp = (key << 16) | (0xffffu - (uint32_t)(base + e));
        if (key > Tkey) {
          sel[og++] = comp;
        } else if (key == Tkey) {
          if ((int)oe < n_eq_take) sel[n_gt + oe] = comp;
          ++oe;
        }
      }
    }
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    run_g += tot & 0xffffu;
    run_e += tot >> 16;
  }
  int kpad;
  if (k <= TK_RADIX_MAX) kpad = (k + 15) & ~15;
  else { kpad = 1; while (kpad < k) kpad <<= 1; }
  for (int i = k + tid; i < kpad; i += TK_THREADS) sel[i] = 0;   // padding: composite 0 sorts last, never emitted
  __syncthreads();
  PKV_STAMP(5);

  if (k <= TK_RANK_MAX) {
    // rank by counting: composites are unique, so ranks form a permutation of 0..k-1
    if (tid < k) {
      const uint32_t mine = sel[tid];
      int rank = 0;
      const uint4* s4 = reinterpret_cast<const uint4*>(sel);
      for (int j = 0; j < (kpad >> 4); ++j) {   // kpad is a multiple of 16 here; same address for every lane: LDS broadcast
        const uint4 a0 = s4[4 * j], a1 = s4[4 * j + 1], a2 = s4[4 * j + 2], a3 = s4[4 * j + 3];
        rank += (a0.x > mine) + (a0.y > mine) + (a0.z > mine) + (a0.w > mine) + (a1.x > mine) + (a1.y > mine) +
                (a1.z > mine) + (a1.w > mine) + (a2.x > mine) + (a2.y > mine) + (a2.z > mine) + (a2.w > mine) +
                (a3.x > mine) + (a3.y > mine) + (a3.z > mine) + (a3.w > mine);
      }
      emit(rank, mine);
    }
  } else if (k <= TK_RADIX_MAX && dual) {
    // the compacted list is index-ordered within equal keys, so a STABLE sort by key alone gives the
    // canonical order: two LSD radix passes (low byte, high byte), ping-pong sel <-> X2[4096..]
    uint32_t* table = X2;                 // [16][256]
    uint32_t* sel2 = X2 + TK_WAVES * 256; // [4096]
    const int ept = (k + TK_THREADS - 1) / TK_THREADS;
    radix_pass(sel, sel2, table, hist, k, ept, 0, tid);
    radix_pass(sel2, sel, table, hist, k, ept, 1, tid);
    for (int i = tid; i < k; i += TK_THREADS) emit(i, sel[i]);
  } else {
    // bitonic network, descending (kpad is a power of two here)
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    for (int i = kpad + tid; i < kp2; i += TK_THREADS) sel[i] = 0;
    __syncthreads();
    for (int size = 2; size <= kp2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (kp2 >> 1); t += TK_THREADS) {
          const int a = 2 * t - (t & (stride - 1));
          const int b = a + stride;
          const bool desc = (a & size) == 0;
          const uint32_t va = sel[a], vb = sel[b];
          if ((va < vb) == desc) { sel[a] = vb; sel[b] = va; }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < k; i += TK_THREADS) emit(i, sel[i]);
  }
  PKV_STAMP(6);
  if (PKV_WGTRACE(p) && tid == 0) { PKV_WGTRACE(p)[2 * (131072 + row)] = t_start; PKV_WGTRACE(p)[2 * (131072 + row) + 1] = wall_clock64(); }
  if (ADA) ada_epilogue();
#undef PKV_STAMP
}

// ------------------------------------------------------------------------------------------------
// sort_rows_kernel: full stable descending sort of one score row (L <= 32768), one workgroup per row.
// Two stable LSD radix passes over the 16-bit keys (low byte, then high byte): the input is in index
// order, so the result is (value desc, index asc) - the reference's attn_score.sort(descending=True) with
// the tie order pinned.  Ranking inside a wave uses match-any ballots (no atomics); composites
// (key<<16 | index) ping-pong through the sorted_idx output buffer, which every thread re-reads into
// registers before anything is overwritten.
// ------------------------------------------------------------------------------------------------

// match-any: the set of valid lanes of the wave that hold the same 8-bit digit
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t bb = __ballot(valid && bit);
    peers &= bit ? bb : ~bb;
  }
  return peers;
}

// Stable descending sort of one row per workgroup (keys = 16-bit scores, ties by ascending index): two LSD radix passes
// (low byte, high byte) ranked with match-any ballots.  Everything between the passes stays in LDS: the row's raw 16-bit
// values (the digits are recomputed from them, and they are what the value output needs) and a 16-bit index permutation
// written by pass 0 and walked by pass 1; only the final (index, value) scatter goes to memory.
template <typename T>
__global__ __launch_bounds__(TK_THREADS) void sort_rows_kernel(SortParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* val = reinterpret_cast<uint16_t*>(smem);                    // [n] raw scores, natural order
  uint16_t* perm = val + p.n;                                           // [n] indices in pass-0 order
  uint32_t* table = reinterpret_cast<uint32_t*>(perm + p.n);            // [16][256]
  uint32_t* tot = table + TK_WAVES * 256;                               // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int L = p.L;
  const int epw = (L + TK_THREADS - 1) / TK_THREADS;         // 64-element steps per wave
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)row * p.scores_stride;
  int32_t* oi = p.sorted_idx + (int64_t)row * L;
  uint16_t* ov = p.sorted_val ? reinterpret_cast<uint16_t*>(p.sorted_val) + (int64_t)row * L : nullptr;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

#define PKV_SSTAMP(i) do { if (PKV_TRACE(p) && tid == 0 && row == 0) PKV_TRACE(p)[i] = (unsigned long long)clock64(); } while (0)
  PKV_SSTAMP(0);
  for (int i = tid; i < L; i += TK_THREADS) val[i] = src[i];
  __syncthreads();
  PKV_SSTAMP(1);
  for (int pass = 0; pass < 2; ++pass) {
    // list element j of this pass: pass 0 = natural order, pass 1 = pass 0's output order
    auto element = [&](int j, uint32_t& idx, uint32_t& d) {
      idx = pass == 0 ? (uint32_t)j : (uint32_t)perm[j];
      const uint32_t key = order_key<T>(val[idx]);
      d = 255u - ((key >> (8 * pass)) & 255u);                          // ascending d == descending key
    };
    for (int i = tid; i < TK_WAVES * 256; i += TK_THREADS) table[i] = 0;
    __syncthreads();
    // sweep A: per-wave digit counts (wave w owns the contiguous list segment [w*epw*64, (w+1)*epw*64))
    for (int e = 0; e < epw; ++e) {
      const int j = wave * epw * 64 + e * 64 + lane;
      const bool valid = j < L;
      uint32_t idx = 0, d = 0;
      if (valid) element(j, idx, d);
      const uint64_t peers = match_digit(d, valid);
      if (valid && (peers & lt) == 0ull) table[wave * 256 + d] += (uint32_t)__popcll(peers);   // group leader
    }
    __syncthreads();
    PKV_SSTAMP(2 + pass * 4);
    if (tid < 256) {                                          // per digit: exclusive prefix over the 16 waves
      uint32_t run = 0;
#pragma unroll
      for (int w2 = 0; w2 < TK_WAVES; ++w2) {
        const uint32_t c = table[w2 * 256 + tid];
        table[w2 * 256 + tid] = run;
        run += c;
      }
      tot[tid] = run;
    }
    __syncthreads();
    if (tid < 64) {                                           // exclusive prefix over the 256 digit totals
      const uint4 h = reinterpret_cast<const uint4*>(tot)[tid];
      const uint32_t own = h.x + h.y + h.z + h.w;
      const uint32_t excl = wave_incl_scan_u32(own) - own;
      reinterpret_cast<uint4*>(tot)[tid] = make_uint4(excl, excl + h.x, excl + h.x + h.y, excl + h.x + h.y + h.z);
    }
    __syncthreads();
    for (int i = tid; i < TK_WAVES * 256; i += TK_THREADS) table[i] += tot[i & 255];   // absolute base of (wave, digit)
    __syncthreads();
    PKV_SSTAMP(3 + pass * 4);
    // sweep B: same ranking against the absolute bases: output positions come out directly (stable)
    for (int e = 0; e < epw; ++e) {
      const int j = wave * epw * 64 + e * 64 + lane;
      const bool valid = j < L;
      uint32_t idx = 0, d = 0;
      if (valid) element(j, idx, d);
      const uint64_t peers = match_digit(d, valid);
      const uint32_t before = (uint32_t)__popcll(peers & lt);
      uint32_t prev = 0;
      if (valid) prev = table[wave * 256 + d];                // every peer reads the running base ...
      if (valid && before == 0) table[wave * 256 + d] = prev + (uint32_t)__popcll(peers);   // ... the leader advances it
      if (valid) {
        const uint32_t pos = prev + before;
        if (pass == 0) {
          perm[pos] = (uint16_t)idx;                          // stays in LDS
        } else {
          oi[pos] = (int32_t)idx;
          if (ov) ov[pos] = val[idx];
        }
      }
    }
    __syncthreads();
    PKV_SSTAMP(4 + pass * 4);
  }
#undef PKV_SSTAMP
}

hipError_t launch_topk(int dtype, int rows, const TopkParams& p, size_t lds, hipStream_t st) {
  const bool ada = p.list_out != nullptr;
  const int ni = p.Lw / 512;
  void (*fn)(TopkParams) = nullptr;
#define PKV_TK(TT, AD)                                                                                                         \
  fn = ni == 1 ? topk_kernel<TT, AD, 1> : ni == 2 ? topk_kernel<TT, AD, 2> : ni == 4 ? topk_kernel<TT, AD, 4> : topk_kernel<TT, AD, 0>
  if (dtype == 0) { if (ada) PKV_TK(BF16, true); else PKV_TK(BF16, false); }
  else            { if (ada) PKV_TK(F16, true); else PKV_TK(F16, false); }
#undef PKV_TK
  if (lds > 64 * 1024) {
    hipError_t e = dyn_lds(reinterpret_cast<const void*>(fn), lds);
    if (e != hipSuccess) return e;
  }
  PKV_KLAUNCH(fn, dim3(rows), dim3(TK_THREADS), lds, st, p);
  return hipGetLastError();
}

// ---- long rows (L beyond one workgroup's LDS): per-segment top-k, then a top-k over the segment winners ----
// cand_score[row][t] = score of candidate t = (segment t / k, rank t % k), the lowest value where that segment holds
// fewer than k keys (only the last one can: the filler sits at the end of the list and loses every tie by position).
template <typename T>
__global__ __launch_bounds__(256) void topk_merge_prep_kernel(int L, int k, int nseg, int seg_len, const uint16_t* scores, int64_t scores_stride,
                                                              const int32_t* cand_idx, uint16_t* cand_score, int64_t cand_stride) {
  const int row = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nseg * k) return;
  const int seg = t / k, j = t - seg * k;
  const int lseg = min(seg_len, L - seg * seg_len);
  uint16_t v = Elem<T>::neg_inf();
  if (j < min(k, lseg)) v = scores[(int64_t)row * scores_stride + cand_idx[(int64_t)row * nseg * k + t]];
  cand_score[(int64_t)row * cand_stride + t] = v;
}

__global__ __launch_bounds__(256) void topk_merge_finish_kernel(int k, const int32_t* cand_idx, int64_t cand_stride, const int32_t* pos,
                                                                int32_t* idx_out, int64_t idx_stride) {
  const int row = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < k) idx_out[(int64_t)row * idx_stride + j] = cand_idx[(int64_t)row * cand_stride + pos[(int64_t)row * k + j]];
}

hipError_t launch_topk_merge_prep(int dtype, int rows, int L, int k, int nseg, int seg_len, const void* scores, int64_t scores_stride,
                                  const int32_t* cand_idx, void* cand_score, int64_t cand_stride, hipStream_t st) {
  dim3 grid((nseg * k + 255) / 256, rows);
  if (dtype == 0) hipLaunchKernelGGL(topk_merge_prep_kernel<BF16>, grid, dim3(256), 0, st, L, k, nseg, seg_len, static_cast<const uint16_t*>(scores),
                                     scores_stride, cand_idx, static_cast<uint16_t*>(cand_score), cand_stride);
  else hipLaunchKernelGGL(topk_merge_prep_kernel<F16>, grid, dim3(256), 0, st, L, k, nseg, seg_len, static_cast<const uint16_t*>(scores),
                          scores_stride, cand_idx, static_cast<uint16_t*>(cand_score), cand_stride);
  return hipGetLastError();
}

hipError_t launch_topk_merge_finish(int rows, int k, const int32_t* cand_idx, int64_t cand_stride, const int32_t* pos, int32_t* idx_out,
                                    int64_t idx_stride, hipStream_t st) {
  hipLaunchKernelGGL(topk_merge_finish_kernel, dim3((k + 255) / 256, rows), dim3(256), 0, st, k, cand_idx, cand_stride, pos, idx_out, idx_stride);
  return hipGetLastError();
}

hipError_t launch_sort_rows(int dtype, int rows, const SortParams& p, hipStream_t st) {
  auto fn = dtype == 0 ? sort_rows_kernel<BF16> : sort_rows_kernel<F16>;
  const size_t lds = (size_t)p.n * 4 + (size_t)TK_WAVES * 256 * 4 + 256 * 4;     // 2 x 16-bit per element + tables
  if (lds > 64 * 1024) {
    hipError_t e = dyn_lds(reinterpret_cast<const void*>(fn), lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(rows), dim3(TK_THREADS), lds, st, p);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// PKV_TIE_ATEN_ROCM (round 4): the order PyTorch-ROCm's tensor.topk leaves EQUAL scores in for k <= 32.
// ATen gathers the winners in two passes - the scores above the k-th value in index order, then the ties of the k-th value
// in index order - and sorts that list (value descending) with a 32-element bitonic network of compare-exchanges that also
// move equal keys (ATen/native/hip/SortUtils.cuh: bitonicSort<32> / bitonicSwap; slices of 33..128 go through a stable warp
// merge sort, longer ones through a stable radix sort: those ARE the canonical order).  One wave per row replays exactly
// that on the canonical list topk_kernel wrote: rank every entry into the gather order, run the network, write the row back.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void aten_small_order_kernel(const uint16_t* scores, int64_t scores_stride, int32_t* idx, int64_t idx_stride, int k) {
  __shared__ float key[32];
  __shared__ int32_t val[32];
  __shared__ bool ok[32];
  const int row = blockIdx.x, lane = threadIdx.x;
  int32_t* ir = idx + (int64_t)row * idx_stride;
  const bool have = lane < k;
  const int my = have ? ir[lane] : 0;
  const float v = have ? Elem<T>::to_f32(scores[(int64_t)row * scores_stride + my]) : 0.f;
  auto gt = [](float a, float b) { return (a != a && b == b) || a > b; };          // GTOp with NaN ranked largest
  const float vk = __shfl(v, k - 1, 64);                                           // canonical order: entry k-1 holds the k-th value
  const bool me_first = gt(v, vk);
  int rank = 0;
  for (int j = 0; j < k; ++j) {                                                     // wave-uniform trip count
    const float vj = __shfl(v, j, 64);
    const int ij = __shfl(my, j, 64);
    const bool j_first = gt(vj, vk);
    if (me_first) rank += (j_first && ij < my) ? 1 : 0;
    else rank += (j_first || ij < my) ? 1 : 0;
  }
  if (lane < 32) { key[lane] = 0.f; val[lane] = 0; ok[lane] = false; }
  __syncthreads();
  if (have) { key[rank] = v; val[rank] = my; ok[rank] = true; }
  const unsigned t = (unsigned)lane;                                                // 16 workers, two entries each
  auto cswap = [&](unsigned a, unsigned b, bool dir) {
    const bool swap = (gt(key[a], key[b]) && ok[a]) || !ok[b];                      // invalid entries sort to the end
    if (swap == dir) {
      const float kf = key[a]; key[a] = key[b]; key[b] = kf;
      const int32_t vi = val[a]; val[a] = val[b]; val[b] = vi;
      const bool vb = ok[a]; ok[a] = ok[b]; ok[b] = vb;
    }
  };
  for (unsigned size = 2; size < 32; size *= 2) {
    const bool flag = (t & (size / 2)) != 0;
    for (unsigned stride = size / 2; stride > 0; stride /= 2) {
      __syncthreads();
      if (t < 16) { const unsigned pos = 2 * t - (t & (stride - 1)); cswap(pos, pos + stride, flag); }
    }
  }
  for (unsigned stride = 16; stride > 0; stride /= 2) {
    __syncthreads();
    if (t < 16) { const unsigned pos = 2 * t - (t & (stride - 1)); cswap(pos, pos + stride, false); }
  }
  __syncthreads();
  if (have) ir[lane] = val[lane];
}

hipError_t launch_aten_small_order(int dtype, int rows, int k, const void* scores, int64_t scores_stride, int32_t* idx, int64_t idx_stride, hipStream_t st) {
  if (k < 2 || k > 32) return hipSuccess;                                           // k = 1: nothing to order; k > 32: ATen's sorts are stable
  const uint16_t* sc = reinterpret_cast<const uint16_t*>(scores);
  if (dtype == 0) hipLaunchKernelGGL(aten_small_order_kernel<BF16>, dim3(rows), dim3(64), 0, st, sc, scores_stride, idx, idx_stride, k);
  else hipLaunchKernelGGL(aten_small_order_kernel<F16>, dim3(rows), dim3(64), 0, st, sc, scores_stride, idx, idx_stride, k);
  return hipGetLastError();
}

}  // namespace pkv
