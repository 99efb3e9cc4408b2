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
#include "pkv_radix.hpp"

namespace pkv {

struct AdaWs {           // layout of the workspace handed to pkv_ada_budget
  float* ratio;          // [H]
  int32_t* cum_hi;       // [H][256]  #entries of head h with adaptive key >= (b<<8)
  int32_t* cum_lo;       // [H][256]  #entries with key >= (b1<<8 | c)
  uint16_t* list;        // [H][Lpad] sorted values of every head as staged by the first kernel (null: re-stage from the inputs)
  int Lpad;
  const int32_t* above_hi;   // optional [H]: entries above every key that shares cum_hi's prefix (more than two radix levels: fp32 keys)
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

template <int NW>
__device__ __forceinline__ void ada_finish(const BudgetParams& p, int gt, int eq, float one_minus_floor, int window, int32_t* head_lens,
                                           int32_t* cu_klen, int32_t* cu_headlens, int64_t* s_red, int* s_scan);

__device__ __forceinline__ void ada_finish_wave(const BudgetParams& p, int gt, int eq, int lane);

// Final step, one workgroup: thread h owns head h.  gt_h = entries above the global threshold, eq_h = entries equal to it;
// the ties are handed out in flattened (head-major) order: head h takes min(eq_h, need - ties taken by the heads before it),
// an exclusive prefix sum over the heads.  Optionally writes the var-len metadata of :682-691 as well (one launch less).
__global__ __launch_bounds__(256) void ada_final_kernel(BudgetParams p, AdaWs ws, float one_minus_floor, int window,
                                                        int32_t* head_lens, int32_t* cu_klen, int32_t* cu_headlens) {
  __shared__ int64_t s_sum[256];
  __shared__ int s_b;
  __shared__ int64_t s_red[4];
  __shared__ int s_scan[4];
  const int tid = threadIdx.x;
  const int64_t total = (int64_t)p.H * p.base;
  const int b1 = find_level(ws.cum_hi, p.H, total, s_sum, &s_b, tid);
  const int b2 = find_level(ws.cum_lo, p.H, total, s_sum, &s_b, tid);
  int gt = 0, eq = 0;
  if (tid < p.H) {
    gt = b2 < 255 ? ws.cum_lo[tid * 256 + b2 + 1] : (b1 < 255 ? ws.cum_hi[tid * 256 + b1 + 1] : (ws.above_hi ? ws.above_hi[tid] : 0));
    eq = ws.cum_lo[tid * 256 + b2] - gt;
  }
  if (p.H <= 64) {                                   // one wave finishes without barriers (find_level above ended with one)
    if (tid < 64) {
      BudgetParams q = p;
      q.one_minus_floor = one_minus_floor; q.window = window; q.head_lens_out = head_lens; q.cu_klen_out = cu_klen; q.cu_headlens_out = cu_headlens;
      ada_finish_wave(q, gt, eq, tid);
    }
    return;
  }
  ada_finish<4>(p, gt, eq, one_minus_floor, window, head_lens, cu_klen, cu_headlens, s_red, s_scan);
}

// ------------------------------------------------------------------------------------------------
// ada_fused_kernel (round 5): budgets + metadata of pkv_ada_select in ONE single-workgroup launch.
// topk_kernel<T, true> left every head's ADAPTIVE list behind (the first L entries of :706's order, times the head's ratio of
// :710, rounded as in :711, as order-preserving 16-bit keys), H x L <= 45 056 entries: all of it fits the LDS of one
// 1024-thread workgroup, so the three launches
// of the list path (ada_stats -> ada_lo -> ada_final, 18.7 us at H = 32: three dependent ~5 us kernels that each wait for
// the previous one's tables in memory) become one launch without any table in memory and without a device-scope meeting
// point (the last-block variant of round 4 lost to exactly that fence):
//   wave w owns heads w, w + 16, ...: exact two-level (8+8 bit) radix select of the (H*base)-th largest adaptive key over
//   all heads (:712-713) = the global threshold T; per head the entries above / at T (:714-717); ties, capacities (:719),
//   var-len metadata (:682-691) and the host mirror as in ada_final_kernel (one wave, no barrier, for H <= 64).
// Same integers as the three-kernel path (tests compare both with the oracle).
// ------------------------------------------------------------------------------------------------
constexpr int ADA_FUSED_MAX_KEYS = 45056;       // 2 B each (88 KB) + two 32 KB counter arrays + ~6 KB of statics inside 160 KB

// tail shared by ada_final_kernel and ada_fused_kernel: thread h < H holds (gt, eq) of head h; every thread of the workgroup calls
template <int NW>
__device__ __forceinline__ void ada_finish(const BudgetParams& p, int gt, int eq, float one_minus_floor, int window, int32_t* head_lens,
                                           int32_t* cu_klen, int32_t* cu_headlens, int64_t* s_red, int* s_scan) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)p.H * p.base;
  // need = total - sum_h gt_h
  int64_t g = gt;
  for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
  if (lane == 0) s_red[wave] = g;
  // exclusive prefix of eq over the heads
  const uint32_t incl = wave_incl_scan_u32((uint32_t)eq);
  if (lane == 63) s_scan[wave] = (int)incl;
  __syncthreads();
  int64_t above = 0;
#pragma unroll
  for (int w2 = 0; w2 < NW; ++w2) above += s_red[w2];
  const int64_t need = total - above;
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
    // reference syncs for the same reason, :718).  They go straight into pinned host memory as SELF-VALIDATING 64-bit words
    // (round 5): word h = host_seq << 32 | ran_out << 31 | cap_h, one relaxed system-scope store per head.  A 64-bit store
    // arrives whole, so the host needs no ordering between words: it polls until every word carries the current sequence
    // number.  No fence and no separate flag word - round 4's capacities + system fence + release flag cost the kernel
    // two PCIe round trips before it could retire (and before the host saw anything).
    // Short lists (fewer than min(L, H*base) entries per head): the threshold and every count are exact as long as no list
    // is used up at the threshold (counts above it are below the list length, so nothing was cut off).  A head whose whole
    // list lies at or above the threshold may own more entries than the list shows: reported in bit 31 of every word, and
    // the caller repeats the call with the full length.
    const int ran_out = __syncthreads_or(p.short_list && tid < p.H && gt + eq >= p.L);
    if (tid < p.H) {
      const unsigned long long wv = ((unsigned long long)(uint32_t)p.host_seq << 32) | (ran_out ? 0x80000000ull : 0ull) | (unsigned long long)(uint32_t)cap;
      __hip_atomic_store(p.host_mirror + tid, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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

__global__ __launch_bounds__(TK_THREADS) void ada_fused_kernel(BudgetParams p, const uint16_t* list, int Lpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ada_smem[];
  __shared__ __attribute__((aligned(16))) uint32_t hist[256];
  __shared__ int misc[4];
  __shared__ int s_gt[256], s_eq[256];
  __shared__ int64_t s_red[TK_WAVES];
  __shared__ int s_scan[TK_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, M = p.L;
  const int nk = H * Lpad;                                         // Lpad is a multiple of 8
  uint16_t* keys = reinterpret_cast<uint16_t*>(ada_smem);
  uint32_t* X1 = reinterpret_cast<uint32_t*>(ada_smem + (((size_t)nk * 2 + 15) & ~(size_t)15));
  uint32_t* X2 = X1 + TK_CNT_WORDS;
  const uint32_t inc = lane < 32 ? 1u : 65536u;
  const int cslot = lane & 31;
  {
    const uint4* src = reinterpret_cast<const uint4*>(list);
    const int nv = nk >> 3;
    for (int c0 = 0; c0 < nv; c0 += 8 * TK_THREADS) {
      uint4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c = c0 + j * TK_THREADS + tid; t[j] = src[c < nv ? c : 0]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c = c0 + j * TK_THREADS + tid; if (c < nv) reinterpret_cast<uint4*>(keys)[c] = t[j]; }
    }
  }
  for (int i = tid; i < 2 * TK_CNT_WORDS; i += TK_THREADS) X1[i] = 0;     // X1 and X2 are adjacent
  __syncthreads();
  // ---- per head (one wave each): high-byte histogram of the adaptive keys (the lists arrive as keys) ----
#pragma unroll 1
  for (int h = wave; h < H; h += TK_WAVES) {
    const uint16_t* v = keys + (size_t)h * Lpad;
    for (int i = lane; i < M; i += 64) atomicAdd(&X1[((uint32_t)v[i] >> 8) * 32 + cslot], inc);
  }
  __syncthreads();
  const uint32_t total = (uint32_t)((int64_t)H * p.base);            // <= H * M <= 47 104
  select_bin(X1, hist, total, &misc[0], &misc[1], tid);
  __syncthreads();
  const uint32_t b1 = (uint32_t)misc[0];
  const int above1 = misc[1];
#pragma unroll 1
  for (int h = wave; h < H; h += TK_WAVES) {
    const uint16_t* v = keys + (size_t)h * Lpad;
    for (int i = lane; i < M; i += 64) {
      const uint32_t key = v[i];
      if ((key >> 8) == b1) atomicAdd(&X2[(key & 255u) * 32 + cslot], inc);
    }
  }
  __syncthreads();
  select_bin(X2, hist, total - (uint32_t)above1, &misc[2], &misc[3], tid);
  __syncthreads();
  const uint32_t Tk = (b1 << 8) | (uint32_t)misc[2];
  // ---- per head: entries above / at the threshold ----
#pragma unroll 1
  for (int h = wave; h < H; h += TK_WAVES) {
    const uint16_t* v = keys + (size_t)h * Lpad;
    uint32_t cg = 0, ce = 0;
    for (int i = lane; i < M; i += 64) {
      const uint32_t key = v[i];
      cg += key > Tk;
      ce += key == Tk;
    }
    const uint32_t packed = wave_sum_u32((ce << 16) | cg);             // <= 47 104 entries in total: no carry between the halves
    if (lane == 0) { s_gt[h] = (int)(packed & 0xffffu); s_eq[h] = (int)(packed >> 16); }
  }
  __syncthreads();
  const int gt = tid < H ? s_gt[tid] : 0, eq = tid < H ? s_eq[tid] : 0;
  if (H <= 64) {                                     // one wave finishes without barriers
    if (wave == 0) ada_finish_wave(p, gt, eq, lane);
    return;
  }
  ada_finish<TK_WAVES>(p, gt, eq, p.one_minus_floor, p.window, p.head_lens_out, p.cu_klen_out, p.cu_headlens_out, s_red, s_scan);
}

// The same step for the shapes the short candidate lists produce (H <= 16 * RH heads, lists of at most 64 * TM entries): every
// lane keeps its entries - entries 8*lane .. 8*lane + 7 of heads wave, wave + 16, ONE 16-byte load per head (2-byte loads: 16 K
// lane-loads through one CU's address unit, ~2 us) - in REGISTERS from the one global load on
// (no staging of the lists in LDS, no LDS round trip per entry and phase: a single workgroup runs 16 waves on one CU, where
// every dependent LDS access costs ~100 cycles of wall time and the three phases below walked the lists three times).
// LDS holds the two bank-spread counter arrays only.  12.7 -> see profiles/r05 (H = 32, lists of 512).
// ada_finish for H <= 64 on ONE wave (lane h = head h), no barrier: in the single-workgroup kernels 16 waves would otherwise walk
// through its ~100 instructions and four barriers for the sake of 32 lanes (4900 of the 25 000 cycles of the one-launch
// budget kernel, profiles/r05/budget_kernel_stamps.json).
__device__ __forceinline__ void ada_finish_wave(const BudgetParams& p, int gt, int eq, int lane) {
  const int64_t total = (int64_t)p.H * p.base;
  const int64_t need = total - (int64_t)wave_sum_u32((uint32_t)gt);              // sum of gt <= H * L < 2^31
  const uint32_t incl = wave_incl_scan_u32((uint32_t)eq);
  const int64_t left = need - ((int64_t)incl - eq);
  const int take = (int)(left <= 0 ? 0 : (left < eq ? left : eq));
  const float capf = __fadd_rn(__fmul_rn((float)(gt + take), p.one_minus_floor), (float)p.floor_capacity);   // :719, fp32
  const int cap = lane < p.H ? (int)rintf(capf) : 0;                                                          // torch.round: half to even
  if (lane < p.H) p.head_capacity[lane] = cap;
  if (p.host_mirror) {
    const bool ran_out = __any(p.short_list && lane < p.H && gt + eq >= p.L) != 0;
    if (lane < p.H) {
      const unsigned long long wv = ((unsigned long long)(uint32_t)p.host_seq << 32) | (ran_out ? 0x80000000ull : 0ull) | (unsigned long long)(uint32_t)cap;
      __hip_atomic_store(p.host_mirror + lane, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (p.head_lens_out && p.cu_klen_out) {            // :684, :689-691
    const int n = lane < p.H ? cap + p.window : 0;
    const uint32_t in2 = wave_incl_scan_u32((uint32_t)n);
    const int off = (int)in2 - n;
    if (lane < p.H) { p.head_lens_out[lane] = n; p.cu_klen_out[lane] = off; if (p.cu_headlens_out) p.cu_headlens_out[lane] = off + n; }
    if (lane == p.H - 1) p.cu_klen_out[p.H] = off + n;
  }
}

template <int RH, int TM>
__global__ __launch_bounds__(TK_THREADS) void ada_fused_reg_kernel(BudgetParams p, const uint16_t* list, int Lpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ada_smem[];
  __shared__ __attribute__((aligned(16))) uint32_t hist[256];
  __shared__ int misc[4];
  __shared__ int s_gt[256], s_eq[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, M = p.L;
#define PKV_BSTAMP(i) do { if (PKV_TRACE(p) && tid == 0) PKV_TRACE(p)[i] = (unsigned long long)clock64(); } while (0)
  PKV_BSTAMP(0);
  uint32_t* X1 = reinterpret_cast<uint32_t*>(ada_smem);
  uint32_t* X2 = X1 + TK_CNT_WORDS;
  const uint32_t inc = lane < 32 ? 1u : 65536u;
  const int cslot = lane & 31;
  // one round trip: this lane's list entries and (thread h: head h) the 16 partial row sums
  static_assert(TM == 8, "a lane holds ONE 16-byte piece (8 consecutive entries) of every list it owns");
  uint32_t key[RH][TM];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int h = wave + r * TK_WAVES;
    const uint16_t* v = list + (size_t)(h < H ? h : 0) * Lpad;
    U4 u;                                                     // entries 8*lane .. 8*lane + 7 (Lpad is a multiple of 8: the piece is inside the row or skipped)
    u.v = *reinterpret_cast<const uint4*>(v + (8 * lane < Lpad ? 8 * lane : 0));
#pragma unroll
    for (int t = 0; t < TM; ++t) key[r][t] = u.h[t];
  }
#pragma unroll
  for (int j = 0; j < 2 * TK_CNT_WORDS / (4 * TK_THREADS); ++j)           // X1 and X2 are adjacent: 4 x 16 B per thread
    reinterpret_cast<uint4*>(X1)[j * TK_THREADS + tid] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  PKV_BSTAMP(1);
  // ---- high-byte histogram of the adaptive keys (the lists arrive as keys: topk_kernel<T, true>'s epilogue) ----
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const bool hv = wave + r * TK_WAVES < H;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const uint32_t kk = (hv && 8 * lane + t < M) ? key[r][t] : 0u;                 // 0 = no entry (real keys are >= 1)
      key[r][t] = kk;
      if (kk) atomicAdd(&X1[(kk >> 8) * 32 + cslot], inc);
    }
  }
  __syncthreads();
  PKV_BSTAMP(2);
  const uint32_t total = (uint32_t)((int64_t)H * p.base);
  select_bin(X1, hist, total, &misc[0], &misc[1], tid);
  __syncthreads();
  PKV_BSTAMP(3);
  const uint32_t b1 = (uint32_t)misc[0];
  const int above1 = misc[1];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const uint32_t k = key[r][t];
      if (k && (k >> 8) == b1) atomicAdd(&X2[(k & 255u) * 32 + cslot], inc);
    }
  }
  __syncthreads();
  PKV_BSTAMP(4);
  select_bin(X2, hist, total - (uint32_t)above1, &misc[2], &misc[3], tid);
  __syncthreads();
  PKV_BSTAMP(5);
  const uint32_t Tk = (b1 << 8) | (uint32_t)misc[2];
  // ---- per head: entries above / at the threshold ----
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    uint32_t packed = 0;
#pragma unroll
    for (int t = 0; t < TM; ++t) { const uint32_t k = key[r][t]; packed += (k > Tk ? 1u : 0u) + (k == Tk ? 65536u : 0u); }
    packed = wave_sum_u32(packed);
    const int h = wave + r * TK_WAVES;
    if (lane == 0 && h < H) { s_gt[h] = (int)(packed & 0xffffu); s_eq[h] = (int)(packed >> 16); }
  }
  __syncthreads();
  PKV_BSTAMP(6);
  if (wave != 0) return;                             // H <= 32 here: one wave finishes (no barrier below)
  ada_finish_wave(p, lane < H ? s_gt[lane] : 0, lane < H ? s_eq[lane] : 0, lane);
  PKV_BSTAMP(7);
#undef PKV_BSTAMP
}

// ------------------------------------------------------------------------------------------------
// Un-sorted variants (BudgetParams::unsorted): the same cum_hi / cum_lo tables straight from the score rows, by counting.
// Used when H * base > 4096 (budget 2048: M = min(L, H * base) is the whole row and a top-M list is a full sort).
// What :706-719 need of the order is two counts per head - how many adaptive scores lie above the global threshold, how
// many equal it - and one sum, the `base` largest scores of the row (:710).  Both are selections, not sorts:
//   ada_stats_u  one 1024-thread workgroup per head, the row in registers (<= 64 keys per thread):
//                exact two-level (8+8 bit) radix select of the base-th largest RAW score -> sum of the top `base` (ties
//                contribute the threshold value) and the row total -> ratio (:710); then the histogram of the ADAPTIVE keys'
//                high byte, suffix-summed into cum_hi;
//   ada_lo_u     after the global high byte b1 is known: histogram of the low byte of the adaptive keys in bin b1 -> cum_lo.
// ada_final_kernel is shared.  Counting is over the WHOLE row (no clamp at M), i.e. the reference's definition itself.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float key_value(uint32_t key) {     // inverse of order_key for finite scores
  const uint32_t k = key - KeyBias<T>::v;
  const uint16_t h = (k & 0x8000u) ? (uint16_t)(k ^ 0x8000u) : (uint16_t)(~k);
  return Elem<T>::to_f32(h);
}

template <typename T, int NIT>
__device__ __forceinline__ void load_row_regs(const uint16_t* row, int L, int Lw, int tid, U4 (&raw)[NIT]) {
  const int wave = tid >> 6, lane = tid & 63;
  const bool vec = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int base = wave * Lw + j * 512 + lane * 8;
    if (j * 512 < Lw) {
      if (vec && base + 8 <= L) {                              // whole chunks only: a row may end at its last score (stride == L)
        raw[j].v = *reinterpret_cast<const uint4*>(row + (base < L ? base : 0));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[j].h[e] = row[base + e < L ? base + e : L - 1];
      }
    }
  }
}

// one histogram pass over the thread's keys: X[digit][32 spread slots] += 1 for every key that `take`s; zeroes X first
template <int NIT, typename F>
__device__ __forceinline__ void hist_pass(uint32_t* X, int Lw, int L, int tid, F&& digit_of) {
  const int wave = tid >> 6, lane = tid & 63;
  const uint32_t inc = lane < 32 ? 1u : 65536u;
  const int cslot = lane & 31;
  for (int i = tid; i < TK_CNT_WORDS; i += TK_THREADS) X[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    if (j * 512 < Lw) {
      const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (base + e < L) {
          const int d = digit_of(j, e);
          if (d >= 0) atomicAdd(&X[d * 32 + cslot], inc);
        }
      }
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(TK_THREADS) void ada_stats_u_kernel(BudgetParams p, AdaWs ws) {
  constexpr int NIT = 8;                                // <= 64 keys per thread: rows up to 65 536 scores
  __shared__ __attribute__((aligned(16))) uint32_t X[TK_CNT_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t hist[256];
  __shared__ int misc[4];
  __shared__ double red[2][TK_WAVES];
  __shared__ float s_ratio;
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = p.Lrow;
  int Lw = ((L + TK_WAVES - 1) / TK_WAVES + 511) / 512 * 512;
  const uint16_t* row = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)h * p.scores_stride;
  U4 raw[NIT];
  load_row_regs<T, NIT>(row, L, Lw, tid, raw);
  float ratio = 1.0f;
  if (p.normalize) {
    // base-th largest raw score: high byte, then low byte inside that bin
    hist_pass<NIT>(X, Lw, L, tid, [&](int j, int e) { return (int)(order_key<T>(raw[j].h[e]) >> 8); });
    select_bin(X, hist, (uint32_t)p.base, &misc[0], &misc[1], tid);
    __syncthreads();
    const int b1 = misc[0], above1 = misc[1];
    hist_pass<NIT>(X, Lw, L, tid, [&](int j, int e) { const uint32_t k = order_key<T>(raw[j].h[e]); return (int)(k >> 8) == b1 ? (int)(k & 255u) : -1; });
    select_bin(X, hist, (uint32_t)(p.base - above1), &misc[2], &misc[3], tid);
    __syncthreads();
    const uint32_t Tk = ((uint32_t)b1 << 8) | (uint32_t)misc[2];
    const int n_gt = above1 + misc[3];
    double st = 0.0, sa = 0.0;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (j * 512 < Lw) {
        const int base = wave * Lw + j * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (base + e < L) {
            const double x = (double)Elem<T>::to_f32(raw[j].h[e]);
            sa += x;
            if (order_key<T>(raw[j].h[e]) > Tk) st += x;
          }
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o, 64); sa += __shfl_xor(sa, o, 64); }
    if (lane == 0) { red[0][wave] = st; red[1][wave] = sa; }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0, a = 0.0;
      for (int w2 = 0; w2 < TK_WAVES; ++w2) { t += red[0][w2]; a += red[1][w2]; }
      t += (double)(p.base - n_gt) * (double)key_value<T>(Tk);          // the ties at the threshold that belong to the top `base`
      const float tq = Elem<T>::to_f32(Elem<T>::from_f32((float)t));    // .sum() result in model dtype (:710)
      const float aq = Elem<T>::to_f32(Elem<T>::from_f32((float)a));
      s_ratio = Elem<T>::to_f32(Elem<T>::from_f32(tq / aq));            // model-dtype division (:710)
      ws.ratio[h] = s_ratio;
    }
    __syncthreads();
    ratio = s_ratio;
  } else if (tid == 0) {
    ws.ratio[h] = 1.0f;
  }
  const int norm = p.normalize;
  hist_pass<NIT>(X, Lw, L, tid, [&](int j, int e) { return (int)(adaptive_key<T>(raw[j].h, e, ratio, norm) >> 8); });
  reduce_counters(X, hist, tid);
  __syncthreads();
  {                                                      // cum_hi[b] = sum of bins b..255: suffix over 4 waves of 64 bins
    const uint32_t c = tid < 256 ? hist[tid] : 0u;
    const uint32_t incl = wave_incl_scan_u32(c);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t suf = wtot - incl + c;                      // bins tid .. end of this wave's 64
    if (lane == 0 && wave < 4) misc[wave] = (int)wtot;
    __syncthreads();
    if (tid < 256) {
      for (int w2 = wave + 1; w2 < 4; ++w2) suf += (uint32_t)misc[w2];
      ws.cum_hi[h * 256 + tid] = (int32_t)suf;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(TK_THREADS) void ada_lo_u_kernel(BudgetParams p, AdaWs ws) {
  constexpr int NIT = 8;
  __shared__ __attribute__((aligned(16))) uint32_t X[TK_CNT_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t hist[256];
  __shared__ int64_t s_sum[256];
  __shared__ int s_b;
  __shared__ int misc[4];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = p.Lrow;
  int Lw = ((L + TK_WAVES - 1) / TK_WAVES + 511) / 512 * 512;
  const uint16_t* row = reinterpret_cast<const uint16_t*>(p.scores) + (int64_t)h * p.scores_stride;
  U4 raw[NIT];
  load_row_regs<T, NIT>(row, L, Lw, tid, raw);
  const int64_t total = (int64_t)p.H * p.base;
  // global high byte (the 256 sums over the heads; threads 0..255)
  if (tid < 256) {
    int64_t s = 0;
    for (int h0 = 0; h0 < p.H; h0 += 32) {
      int32_t v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = ws.cum_hi[(h0 + j < p.H ? h0 + j : p.H - 1) * 256 + tid];
#pragma unroll
      for (int j = 0; j < 32; ++j) s += (h0 + j < p.H) ? v[j] : 0;
    }
    s_sum[tid] = s;
  }
  __syncthreads();
  if (tid < 256 && s_sum[tid] >= total && (tid == 255 || s_sum[tid + 1] < total)) s_b = tid;
  __syncthreads();
  const int b1 = s_b;
  const float ratio = ws.ratio[h];
  const int norm = p.normalize;
  hist_pass<NIT>(X, Lw, L, tid, [&](int j, int e) { const uint32_t k = adaptive_key<T>(raw[j].h, e, ratio, norm); return (int)(k >> 8) == b1 ? (int)(k & 255u) : -1; });
  reduce_counters(X, hist, tid);
  __syncthreads();
  const int above = b1 < 255 ? ws.cum_hi[h * 256 + b1 + 1] : 0;       // adaptive keys of this head in higher bins
  {
    const uint32_t c = tid < 256 ? hist[tid] : 0u;
    const uint32_t incl = wave_incl_scan_u32(c);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t suf = wtot - incl + c;
    if (lane == 0 && wave < 4) misc[wave] = (int)wtot;
    __syncthreads();
    if (tid < 256) {
      for (int w2 = wave + 1; w2 < 4; ++w2) suf += (uint32_t)misc[w2];
      ws.cum_lo[h * 256 + tid] = (int32_t)suf + above;
    }
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

// the shared last step for callers that build their own count tables (pkv_f32.hip: four radix levels of 32-bit keys; the
// last two levels' tables play cum_hi / cum_lo)
hipError_t launch_ada_final(const BudgetParams& p, int32_t* cum_hi, int32_t* cum_lo, const int32_t* above_hi, hipStream_t st) {
  AdaWs ws;
  ws.ratio = nullptr; ws.cum_hi = cum_hi; ws.cum_lo = cum_lo; ws.list = nullptr; ws.Lpad = 0; ws.above_hi = above_hi;
  hipLaunchKernelGGL(ada_final_kernel, dim3(1), dim3(256), 0, st, p, ws, p.one_minus_floor, p.window, p.head_lens_out,
                     p.cu_klen_out, p.cu_headlens_out);
  return hipGetLastError();
}

hipError_t launch_budget(int dtype, const BudgetParams& p, hipStream_t st) {
  AdaWs ws;
  ws.above_hi = nullptr;
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
  if (p.unsorted) {               // counts straight from the un-sorted score rows (no top-M list, no sort)
    if (dtype == 0) {
      hipLaunchKernelGGL(ada_stats_u_kernel<BF16>, dim3(p.H), dim3(TK_THREADS), 0, st, p, ws);
      hipLaunchKernelGGL(ada_lo_u_kernel<BF16>, dim3(p.H), dim3(TK_THREADS), 0, st, p, ws);
    } else {
      hipLaunchKernelGGL(ada_stats_u_kernel<F16>, dim3(p.H), dim3(TK_THREADS), 0, st, p, ws);
      hipLaunchKernelGGL(ada_lo_u_kernel<F16>, dim3(p.H), dim3(TK_THREADS), 0, st, p, ws);
    }
    hipLaunchKernelGGL(ada_final_kernel, dim3(1), dim3(256), 0, st, p, ws, omf, p.window, p.head_lens_out, p.cu_klen_out,
                       p.cu_headlens_out);
    return hipGetLastError();
  }
  if (lds > 48 * 1024) {                // the list kernels stage a whole list in LDS (the un-sorted kernels above use static LDS only)
    hipError_t e = dyn_lds(reinterpret_cast<const void*>(k_stats), lds);
    if (e == hipSuccess) e = dyn_lds(reinterpret_cast<const void*>(k_lo), lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_stats, dim3(p.H), dim3(256), lds, st, p, ws);
  if (p.adaptive_out) return hipGetLastError();
  hipLaunchKernelGGL(k_lo, dim3(p.H), dim3(256), lds, st, p, ws);
  hipLaunchKernelGGL(ada_final_kernel, dim3(1), dim3(256), 0, st, p, ws, omf, p.window, p.head_lens_out, p.cu_klen_out,
                     p.cu_headlens_out);
  return hipGetLastError();
}

bool ada_fused_fits(int H, int M) {
  const int64_t lpad = ((int64_t)M + 7) & ~(int64_t)7;
  return H >= 1 && H <= 256 && M >= 1 && (int64_t)H * lpad <= ADA_FUSED_MAX_KEYS;
}

template <int RH>
static hipError_t launch_ada_fused_reg(const BudgetParams& p, const uint16_t* list, int Lpad, hipStream_t st) {
  auto fn = ada_fused_reg_kernel<RH, 8>;
  const size_t lds = (size_t)2 * TK_CNT_WORDS * 4;
  hipError_t e = dyn_lds(reinterpret_cast<const void*>(fn), lds);
  if (e != hipSuccess) return e;
  PKV_KLAUNCH(fn, dim3(1), dim3(TK_THREADS), lds, st, p, list, Lpad);
  return hipGetLastError();
}

hipError_t launch_ada_fused(const BudgetParams& p, const void* list, int Lpad, hipStream_t st) {
  if (p.L <= 512 && p.H <= 32) {          // the short-list shapes: entries in registers (at most 2 heads x 8 entries per lane)
    const uint16_t* l16 = static_cast<const uint16_t*>(list);
    return p.H <= 16 ? launch_ada_fused_reg<1>(p, l16, Lpad, st) : launch_ada_fused_reg<2>(p, l16, Lpad, st);
  }
  auto fn = ada_fused_kernel;
  const size_t lds = (((size_t)p.H * Lpad * 2 + 15) & ~(size_t)15) + (size_t)2 * TK_CNT_WORDS * 4;
  if (lds > 48 * 1024) {
    hipError_t e = dyn_lds(reinterpret_cast<const void*>(fn), lds);
    if (e != hipSuccess) return e;
  }
  PKV_KLAUNCH(fn, dim3(1), dim3(TK_THREADS), lds, st, p, static_cast<const uint16_t*>(list), Lpad);
  return hipGetLastError();
}

hipError_t launch_ada_metadata(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen, hipStream_t st, int32_t* cu_headlens) {
  hipLaunchKernelGGL(ada_metadata_kernel, dim3(1), dim3(64), 0, st, H, w, cap, head_lens, cu_klen, cu_headlens);
  return hipGetLastError();
}

}  // namespace pkv
