// pkv_select.hip — small budgets (k <= 512): the tail of update_kv in TWO launches instead of three (gfx950).
//
//   select_parts_kernel   reference pyramidkv_utils.py:326-334: fp32 softmax -> dtype, window-row sum / mean -> dtype,
//                         avg / max pool (the arithmetic of finalize_kernel, operation for operation), then the EXACT
//                         top-min(k, 4096) of every 4096-position part of the row, in canonical order
//   gather_merge_kernel   :334-346: the k best of a head's part lists (a rank = own position + binary searches in the
//                         other lists), then the gather-compaction of gather_kernel
//
// Why: at budget 128 the K scan is followed by three dependent latency-bound launches (finalize 10.3 us, top-k 7.4 us on
// one CU per head, gather 4.4 us: a third of the call).  Nothing in that chain is bandwidth: it is kernel boundaries, cold
// round trips and a 16-wave workgroup's barriers.  Here every part of a row finishes its scores and selects its own
// candidates while they are still in registers (no score / chunk-maximum round trip through memory, no second cold load,
// 8 workgroups per head instead of one), and the one cross-part step - the merge of 8 x k sorted candidates - rides on
// the kernel boundary the gather needs anyway: no inter-workgroup signalling anywhere.
//
// Candidate = composite  key << 16 | (0xffff - position): descending composite order == (value desc, index asc), the
// canonical order of topk_kernel; composites are unique, so ranks are a permutation and the result is bit-identical to the
// three-launch path whatever the tie structure.  Rows up to 65 536 past tokens; longer rows and k > 512 take the
// three-launch path.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

constexpr int SP_THREADS = 512;
constexpr int SP_KEYS = 8;                         // positions per thread
constexpr int SP_PART = SP_THREADS * SP_KEYS;      // 4096 positions per workgroup
constexpr int SP_BINS = 8192;                      // 13-bit histogram of the order keys

int select_part_len() { return SP_PART; }

// p = round(exp(x - M) * (1/Z)) for two packed logits, accumulated in fp32: the inner operation of finalize_kernel
template <typename T>
__device__ __forceinline__ void prob_accum2(uint32_t two, float M, float RZ, float wt, float& a0, float& a1) {
  const pkv_f32x2 mm = {M, M}, rz = {RZ, RZ}, ww = {wt, wt};
  const pkv_f32x2 x = {Elem<T>::to_f32((uint16_t)(two & 0xffffu)), Elem<T>::to_f32((uint16_t)(two >> 16))};
  const pkv_f32x2 pr = pkv_exp_pair(x - mm) * rz;                                   // fp32 softmax (:326)
  const uint32_t pk2 = round_pack2<T>(pr.x, pr.y);                                  // .to(dtype)
  const pkv_f32x2 pv = {Elem<T>::to_f32((uint16_t)(pk2 & 0xffffu)), Elem<T>::to_f32((uint16_t)(pk2 >> 16))};
  const pkv_f32x2 a2 = __builtin_elementwise_fma(ww, pv, pkv_f32x2{a0, a1});        // fp32 row accumulate (:327)
  a0 = a2.x;
  a1 = a2.y;
}

template <typename T>
__global__ __launch_bounds__(SP_THREADS) void select_parts_kernel(SelectParams p) {
  __shared__ __attribute__((aligned(16))) uint32_t hist[SP_BINS];
  __shared__ __attribute__((aligned(16))) uint32_t lbuf[SP_PART];      // first: pre-pool scores u16[SP_PART + 16]; then: candidates
  __shared__ float rowM[64];
  __shared__ float rowS[64];
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t misc[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = blockIdx.x, bh = blockIdx.y;
  const int w = p.w, L = p.S - w;
  const int64_t rowbase = (int64_t)bh * w;
  const int p0 = part * SP_PART;
  const int s0 = p0 + tid * SP_KEYS;
  const bool in_row = s0 < L;
  const int nvalid = min(SP_PART, L - p0);
  const int k_loc = min(p.k, nvalid);

  {   // zero the histogram while nothing else is ready
    uint4* h4 = reinterpret_cast<uint4*>(hist);
#pragma unroll
    for (int i = 0; i < SP_BINS / 4 / SP_THREADS; ++i) h4[tid + i * SP_THREADS] = make_uint4(0, 0, 0, 0);
  }
  // ---- row statistics from the K scan's partials (finalize_kernel's reduction; 32 lanes per row, 16 rows per pass) ----
  {
    const int sub = tid & 31;
    for (int r0 = 0; r0 < w; r0 += SP_THREADS / 32) {
      const int r = r0 + (tid >> 5);
      const bool live = r < w;
      const float2* pr = p.partial + (rowbase + (live ? r : 0)) * p.nT;
      float m = -INFINITY, z = 0.f;
      for (int c0 = 0; c0 < p.nT; c0 += 256) {
        float2 pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = c0 + sub + 32 * i;
          pv[i] = pr[t < p.nT ? t : p.nT - 1];
          if (t >= p.nT) pv[i] = make_float2(-INFINITY, 0.f);
        }
        float mc = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) mc = fmaxf(mc, pv[i].x);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mc = fmaxf(mc, __shfl_xor(mc, o, 64));
        const float mn = fmaxf(m, mc);
        float zc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float t = pv[i].y * pkv_exp(pv[i].x - mn);
          zc += (pv[i].x != -INFINITY) ? t : 0.f;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zc += __shfl_xor(zc, o, 64);
        z = (m == -INFINITY ? 0.f : z * pkv_exp(m - mn)) + zc;
        m = mn;
      }
      if (live && sub == 0) { rowM[r] = m; rowS[r] = 1.0f / z; }
    }
  }
  __syncthreads();

  // ---- pre-pool scores of this thread's 8 positions (and of one halo position for threads 0..15) ----
  const uint16_t pad = (p.pool_kind == 2) ? Elem<T>::neg_inf() : (uint16_t)0;
  uint16_t* sc = reinterpret_cast<uint16_t*>(lbuf);                   // index = position - p0 + 8
  const uint16_t* lg0 = reinterpret_cast<const uint16_t*>(p.logits) + rowbase * (int64_t)p.Sp;
  {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint16_t* lgp = lg0 + (in_row ? s0 : 0);
    for (int rb = 0; rb < w; rb += 8) {
      u32x4 u[8];
      float M[8], RZ[8], wt[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = rb + j < w ? rb + j : w - 1;
        u[j] = *reinterpret_cast<const u32x4*>(lgp + (int64_t)r * p.Sp);
        M[j] = rowM[r];
        RZ[j] = rowS[r];
        wt[j] = rb + j < w ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        prob_accum2<T>(u[j].x, M[j], RZ[j], wt[j], acc[0], acc[1]);
        prob_accum2<T>(u[j].y, M[j], RZ[j], wt[j], acc[2], acc[3]);
        prob_accum2<T>(u[j].z, M[j], RZ[j], wt[j], acc[4], acc[5]);
        prob_accum2<T>(u[j].w, M[j], RZ[j], wt[j], acc[6], acc[7]);
      }
    }
    uint16_t ov[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (p.reduce == 1) ? (acc[e] / (float)w) : acc[e];             // mean (:661) or sum (:327)
      ov[e] = (in_row && s0 + e < L) ? Elem<T>::from_f32(v) : pad;
    }
    uint4 pk;
    pk.x = (uint32_t)ov[0] | ((uint32_t)ov[1] << 16);
    pk.y = (uint32_t)ov[2] | ((uint32_t)ov[3] << 16);
    pk.z = (uint32_t)ov[4] | ((uint32_t)ov[5] << 16);
    pk.w = (uint32_t)ov[6] | ((uint32_t)ov[7] << 16);
    *reinterpret_cast<uint4*>(sc + 8 + tid * 8) = pk;
  }
  if (tid < 16) {                                                      // halo: 8 positions before the part, 8 after it
    const int pos = tid < 8 ? p0 - 8 + tid : p0 + SP_PART + (tid - 8);
    const bool ok = pos >= 0 && pos < L;
    float a0 = 0.f, a1 = 0.f;
    const uint16_t* lgp = lg0 + (ok ? pos : 0);
    for (int r = 0; r < w; ++r) {
      const uint32_t x = lgp[(int64_t)r * p.Sp];
      prob_accum2<T>(x | (x << 16), rowM[r], rowS[r], 1.0f, a0, a1);
    }
    const float v = (p.reduce == 1) ? (a0 / (float)w) : a0;
    sc[tid < 8 ? tid : SP_PART + tid] = ok ? Elem<T>::from_f32(v) : pad;
  }
  __syncthreads();

  // ---- pooling (finalize_kernel's, 8 outputs per thread) -> ordered 16-bit keys ----
  uint32_t key[8];
  {
    float v[24];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const uint4 t = *reinterpret_cast<const uint4*>(sc + tid * 8 + i * 8);
      const uint32_t ww[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[i * 8 + 2 * q] = Elem<T>::to_f32((uint16_t)(ww[q] & 0xffffu));
        v[i * 8 + 2 * q + 1] = Elem<T>::to_f32((uint16_t)(ww[q] >> 16));
      }
    }
    const int half = p.pool_kernel >> 1;
    const float ks = (float)p.pool_kernel;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t r16;
      if (p.pool_kind == 0) {
        r16 = Elem<T>::from_f32(v[8 + e]);                               // exact: v is a widened 16-bit value
      } else if (p.pool_kind == 2) {                                     // max_pool1d, -inf padding (:331)
        float m = -INFINITY;
#pragma unroll
        for (int j = -8; j <= 8; ++j)
          if (j >= -half && j <= half) m = fmaxf(m, v[8 + e + j]);
        r16 = Elem<T>::from_f32(m);
      } else {                                                           // avg_pool1d, zero padding, / kernel (:329)
        float sum = 0.f;
#pragma unroll
        for (int j = -8; j <= 8; ++j)
          if (j >= -half && j <= half) sum += v[8 + e + j];              // left-to-right fp32 sum
        r16 = Elem<T>::from_f32(sum / ks);
      }
      key[e] = (in_row && s0 + e < L) ? order_key<T>(r16) : 0u;          // real keys are >= 1
    }
  }
  // ---- exact threshold in two histogram rounds.  All 4096 keys of a part sit in a few dozen 13-bit bins (16-bit scores
  //      of one softmax row span 2-3 binades), and LDS atomics on a hot address serialise (~1.7 cycles per lane-op: 7k
  //      cycles for 4096 keys, measured in session 3).  Round 1 looks at ONE value per thread, the maximum of its 8-key
  //      chunk: the k_loc-th largest chunk maximum x_c is a lower bound of the threshold (k_loc chunks hold a key >= x_c).
  //      Round 2 histograms only the keys >= x_c (a few hundred), at FULL key resolution relative to x_c. ----
  auto find_bin = [&](uint32_t need) -> uint32_t {        // bin b: count(bins > b) < need <= count(bins >= b); block-uniform result
    const uint4* h4 = reinterpret_cast<const uint4*>(hist) + tid * 4;    // bins 16*tid .. 16*tid + 15
    const uint4 a = h4[0], b = h4[1], c = h4[2], d = h4[3];
    const uint32_t hv[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t s16 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s16 += hv[i];
    const uint32_t incl = wave_incl_scan_u32(s16);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) - incl;   // bins of higher lanes of this wave
#pragma unroll
    for (int w2 = 0; w2 < SP_THREADS / 64; ++w2) above += (w2 > wave) ? wtot[w2] : 0u;
    if (above < need && need <= above + s16) {                                    // exactly one thread
#pragma unroll
      for (int i = 15; i >= 0; --i) {
        if (above < need && need <= above + hv[i]) misc[0] = (uint32_t)(tid * 16 + i);
        above += hv[i];
      }
    }
    __syncthreads();
    return misc[0];
  };
  uint32_t cmx = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) cmx = cmx > key[e] ? cmx : key[e];
  const int nchunks = (nvalid + 7) >> 3;
  uint32_t xc = 0;                                                       // round-1 bound; 0 = every key is a candidate
  if (nchunks >= k_loc) {                                                // block-uniform
    if (cmx != 0u) atomicAdd(&hist[cmx >> 3], 1u);
    __syncthreads();                                                     // also: every read of `sc` is done, lbuf is free
    xc = find_bin((uint32_t)k_loc) << 3;
    if (cmx != 0u) atomicSub(&hist[cmx >> 3], 1u);                       // back to all-zero without a 32 KB clear
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (key[e] != 0u && key[e] >= xc) {
      const uint32_t d = key[e] - xc;
      atomicAdd(&hist[d < (uint32_t)SP_BINS - 1 ? d : (uint32_t)SP_BINS - 1], 1u);
    }
  }
  __syncthreads();
  const uint32_t xstar = xc + find_bin((uint32_t)k_loc);                 // at least k_loc keys are >= xstar
  // ---- candidates (key >= xstar) in position order -> composites in LDS ----
  uint32_t cj = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) cj += (key[e] != 0u && key[e] >= xstar) ? 1u : 0u;
  const uint32_t incl = wave_incl_scan_u32(cj);
  if (lane == 63) wtot[8 + wave] = incl;
  __syncthreads();
  uint32_t posn = incl - cj, C = 0;
#pragma unroll
  for (int w2 = 0; w2 < SP_THREADS / 64; ++w2) {
    const uint32_t t = wtot[8 + w2];
    posn += (w2 < wave) ? t : 0u;
    C += t;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (key[e] != 0u && key[e] >= xstar) lbuf[posn++] = (key[e] << 16) | (0xffffu - (uint32_t)(s0 + e));
  for (uint32_t i = C + tid; i < ((C + 3u) & ~3u); i += SP_THREADS) lbuf[i] = 0u;        // pad to whole 16-B groups
  __syncthreads();
  // ---- rank by counting (composites are unique): rank == position in the canonical order; the first k_loc leave ----
  uint32_t* out = p.cand + ((int64_t)bh * p.nparts + part) * p.k;
  const uint4* l4 = reinterpret_cast<const uint4*>(lbuf);
  const int n4 = (int)((C + 3u) >> 2);
  for (uint32_t ci = tid; ci < C; ci += SP_THREADS) {
    const uint32_t mine = lbuf[ci];
    uint32_t rank = 0;
    for (int q = 0; q < n4; ++q) {
      const uint4 a = l4[q];
      rank += (a.x > mine) + (a.y > mine) + (a.z > mine) + (a.w > mine);
    }
    if (rank < (uint32_t)k_loc) out[rank] = mine;
  }
  for (int i = k_loc + tid; i < p.k; i += SP_THREADS) out[i] = 0u;        // a short last part: nothing sorts below 0
}

hipError_t launch_select_parts(int dtype, const SelectParams& p, hipStream_t st) {
  dim3 grid(p.nparts, p.B * p.H);
  if (dtype == 0) PKV_KLAUNCH(select_parts_kernel<BF16>, grid, dim3(SP_THREADS), 0, st, p);
  else PKV_KLAUNCH(select_parts_kernel<F16>, grid, dim3(SP_THREADS), 0, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// gather_merge_kernel: one workgroup = 1024 threads = 1024 / CH row slots x {K,V} of one (b,h).
// Every workgroup of a head merges the head's part lists itself (a few KB in LDS; every list is sorted, so a
// candidate's global rank is its own position plus one binary search per other list) and then moves its rows exactly
// as gather_kernel does.  Block 0 of a head also writes the index list (pkv_compress's idx_out / the workspace).
// ------------------------------------------------------------------------------------------------
constexpr int GM_THREADS = 1024;
constexpr int GM_MAXPARTS = 16;                   // 65 536 positions / 4096 per part

template <int CH>
__global__ __launch_bounds__(GM_THREADS) void gather_merge_kernel(GatherParams p, const uint32_t* cand, int nparts, int32_t* idx_out,
                                                                 int64_t idx_stride) {
  extern __shared__ __attribute__((aligned(16))) uint32_t gm_smem[];
  constexpr int SLOTS = GM_THREADS / CH;
  constexpr int DH = CH * 8;
  const int tid = threadIdx.x;
  const int bh = blockIdx.x / p.nblk;
  const int blk = blockIdx.x - bh * p.nblk;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int L = p.S - p.w, k = p.nsel, nrows = k + p.w;
  uint32_t* lists = gm_smem;                       // [nparts][k]
  int32_t* sel = reinterpret_cast<int32_t*>(gm_smem + nparts * k);   // [k]
  const int n = nparts * k;
  const uint32_t* src = cand + (int64_t)bh * n;
  for (int i = tid; i < n; i += GM_THREADS) lists[i] = __builtin_nontemporal_load(src + i);
  __syncthreads();
  int kp2 = 1;
  while (kp2 <= k) kp2 <<= 1;                      // power of two > k
  // One binary search per other list, ALL lists of a candidate in lockstep: every step issues its (<= 16) LDS reads back
  // to back.  (List after list, the 7 x 8 dependent reads per candidate cost 6.8 us per gather workgroup - session 3.)
  for (int c = tid; c < n; c += GM_THREADS) {
    const uint32_t x = lists[c];
    if (x == 0u) continue;
    const int j = c / k;
    int pos[GM_MAXPARTS];
#pragma unroll
    for (int j2 = 0; j2 < GM_MAXPARTS; ++j2) pos[j2] = 0;
    for (int s = kp2 >> 1; s > 0; s >>= 1) {
#pragma unroll
      for (int j2 = 0; j2 < GM_MAXPARTS; ++j2) {
        if (j2 < nparts) {                         // entries of list j2 greater than x (lists are descending, 0-padded)
          const int q = pos[j2] + s;
          if (q <= k && lists[j2 * k + q - 1] > x) pos[j2] = q;
        }
      }
    }
    int rank = 0;
#pragma unroll
    for (int j2 = 0; j2 < GM_MAXPARTS; ++j2) rank += (j2 < nparts && j2 != j) ? pos[j2] : 0;
    rank += c - j * k;                             // own list: the candidate's position
    if (rank < k) sel[rank] = (int32_t)(0xffffu - (x & 0xffffu));
  }
  __syncthreads();
  if (blk == 0 && idx_out) {
    for (int i = tid; i < k; i += GM_THREADS) idx_out[(int64_t)bh * idx_stride + i] = sel[i];
  }
  const int chunk = tid % CH, slot = tid / CH;
  const int r = blk * SLOTS + slot;
  if (r >= nrows) return;
  const int g = r < k ? min(max(sel[r], 0), L - 1) : L + (r - k);
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h + chunk * 8;
  const uint16_t* vb = reinterpret_cast<const uint16_t*>(p.vptr) + (int64_t)b * p.vs_b + (int64_t)hk * p.vs_h + chunk * 8;
  const u32x4 kd = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (int64_t)g * p.ks_s));
  const u32x4 vd = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (int64_t)g * p.vs_s));
  const int64_t orow = (int64_t)bh * nrows + r;
  __builtin_nontemporal_store(kd, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.k_out) + orow * DH + chunk * 8));
  __builtin_nontemporal_store(vd, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.v_out) + orow * DH + chunk * 8));
}

hipError_t launch_gather_merge(const GatherParams& p0, const uint32_t* cand, int nparts, int32_t* idx_out, int64_t idx_stride,
                               hipStream_t st) {
  GatherParams p = p0;
  const int ch = p.D / 8;
  const int slots = GM_THREADS / ch;
  const int nrows = p.nsel + p.w;
  p.nblk = (nrows + slots - 1) / slots;
  const size_t lds = ((size_t)nparts * p.nsel + p.nsel) * 4;
  dim3 grid((unsigned)(p.nblk * p.B * p.H));
#define PKV_GM(C)                                                                                                       \
  do {                                                                                                                 \
    if (lds > 64 * 1024) {                                                                                             \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_merge_kernel<C>),                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
      if (e_ != hipSuccess) return e_;                                                                                \
    }                                                                                                                  \
    PKV_KLAUNCH((gather_merge_kernel<C>), grid, dim3(GM_THREADS), lds, st, p, cand, nparts, idx_out, idx_stride);      \
  } while (0)
  if (ch == 16) PKV_GM(16); else if (ch == 8) PKV_GM(8); else PKV_GM(32);
#undef PKV_GM
  return hipGetLastError();
}

}  // namespace pkv
