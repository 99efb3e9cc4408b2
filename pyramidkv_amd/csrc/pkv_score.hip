// pkv_score.hip — observation-window score kernels (gfx950).
//
//   logits_kernel    reference pyramidkv_utils.py:317-324  (Q[-w:] K^T / sqrt(D), corner mask)
//                    + per-tile softmax partials (row max, sum of exp) so K is read exactly once
//   finalize_kernel  reference :326-333  (fp32 softmax -> model dtype, window-row sum/mean -> model
//                    dtype, avg/max pool)
//
// Roofline: HBM.  Algorithmic bytes per (b,h): S*D*e for K (the dominant term; /kv_group when the
// caller hands over un-expanded GQA K) + w*D*e for Q.  Logits [w][S] (1/16 of K at w=8) make one
// round trip through L2/MALL between the two kernels.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"
#include "pkv_mfma.hpp"

namespace pkv {

// ------------------------------------------------------------------------------------------------
// logits_kernel: one workgroup = 4*KPW keys of one (batch, kv-head group); 4 waves x KPW keys.
// MFMA 16x16x32: A = 16 keys x 32 d (K tile, rows), B = 32 d x 16 columns (window queries).
// A column c of the group is (head h0 + c / w, window row c % w); C = kv_group * w columns.
// D element [key i][col j] sits in lane (j + 16*(i/4)), register i%4.
// KPW = 32 keeps the K fragments at 32 VGPRs so ~7 workgroups stay resident per CU: the HBM queue
// never drains while other workgroups are in their MFMA / LDS / store phases.  K is streamed once
// and never re-read: nontemporal loads (NT) keep it out of the way of the logits in L2.
// ------------------------------------------------------------------------------------------------
// KS = head_dim / 32 MFMA k-steps (2, 4, 8 for head sizes 64, 128, 256).  This kernel is the general one: logits2_kernel
// below is specialised to 256-byte rows (head size 128) and takes over whenever it applies.
template <typename T, int KPW, bool NT, int KS>
__global__ __launch_bounds__(256) void logits_kernel(LogitsParams p) {
  constexpr int TILE = 4 * KPW;            // keys per workgroup
  constexpr int LROW = TILE + 8;           // LDS row stride in elements (16-B aligned rows, <=2-way write conflicts)
  constexpr int NSUB = KPW / 16;           // 16-key MFMA subtiles per wave
  constexpr int CH = TILE / 8;             // 16-B chunks (lanes) per logits row of the tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t* tile = reinterpret_cast<uint16_t*>(smem_raw);  // [C][LROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_idx = blockIdx.x;                 // key tile
  const int grp = blockIdx.y;                   // b * (H/G) + kv head
  const int HG = p.H / p.G;
  const int b = grp / HG;
  const int hk = grp - b * HG;
  const int h0 = hk * p.G;
  const int w = p.w;
  const int C = p.G * w;
  const int S = p.S;
  const int L = S - w;

  const unsigned long long t_start = PKV_WGTRACE(p) ? wall_clock64() : 0ull;
  const int li = lane & 15;   // key within 16-subtile (A rows) / column within 16-tile (B cols)
  const int lg = lane >> 4;   // 8-element d-chunk within a 32-wide k-step

  // ---- issue all K loads of this wave: NSUB subtiles x KS k-steps x 16 B per lane ----
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int s_wave = t_idx * TILE + wave * KPW;
  u32x4 kf[NSUB][KS];
#pragma unroll
  for (int t = 0; t < NSUB; ++t) {
    int s = s_wave + t * 16 + li;
    s = s < S ? s : S - 1;  // clamp: stay in bounds; out-of-range keys are masked out of the stats below
    const uint16_t* row = kbase + (int64_t)s * p.ks_s + lg * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const u32x4* ptr = reinterpret_cast<const u32x4*>(row + kk * 32);
      kf[t][kk] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    }
  }

  if (PKV_ABLATE(p) == 3) {   // measurement aid: K loads only
    u32x4 a = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NSUB; ++t)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) a ^= kf[t][kk];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) p.partial[0] = make_float2(0.f, 0.f);
    return;
  }
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b;
  const float fmin_v = Elem<T>::finfo_min();
  const int ntile_c = (C + 15) >> 4;
  const bool corner_tile = (t_idx + 1) * TILE > L;      // workgroup-uniform: this tile touches keys >= S - w

  for (int n = 0; n < ntile_c; ++n) {
    // B fragments: column c = n*16 + li -> Q[b, h0 + c/w, S - w + c%w, kk*32 + lg*8 ..]
    const int c = n * 16 + li;
    u32x4 qf[KS];
    if (c < C) {
      const int hh = h0 + c / w;
      const int rr = c - (c / w) * w;
      const uint16_t* qrow = qb + (int64_t)hh * p.qs_h + (int64_t)(L + rr) * p.qs_s + lg * 8;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[kk] = *reinterpret_cast<const u32x4*>(qrow + kk * 32);
    } else {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[kk] = u32x4{0, 0, 0, 0};
    }
    const int rr_c = c % w;  // window row of this lane's column
#pragma unroll
    for (int t = 0; t < NSUB; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) acc = Mfma<T>::run(kf[t][kk], qf[kk], acc);
      if (PKV_ABLATE(p) == 2) {   // measurement aid: loads + MFMA only
        if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) p.partial[0] = make_float2(0.f, 0.f);
        continue;
      }
      // epilogue: three roundings to the model dtype, as the reference materialises them; values are
      // handled as packed pairs (one v_cvt_pk per two roundings), the corner mask only in the tiles
      // that reach the observation window (wave-uniform branch)
      const int key0 = wave * KPW + t * 16 + lg * 4;       // key within the tile of acc[0]
      const int sg0 = t_idx * TILE + key0;                 // global key index
      uint32_t p01 = round_pack2<T>(acc[0], acc[1]);                               // matmul output dtype (:317)
      uint32_t p23 = round_pack2<T>(acc[2], acc[3]);
      float x0 = Elem<T>::to_f32((uint16_t)(p01 & 0xffffu)), x1 = Elem<T>::to_f32((uint16_t)(p01 >> 16));
      float x2 = Elem<T>::to_f32((uint16_t)(p23 & 0xffffu)), x3 = Elem<T>::to_f32((uint16_t)(p23 >> 16));
      x0 = scale_logit<T>(x0, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);              // "/ math.sqrt(head_dim)" (:317)
      x1 = scale_logit<T>(x1, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
      x2 = scale_logit<T>(x2, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
      x3 = scale_logit<T>(x3, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
      p01 = round_pack2<T>(x0, x1);
      p23 = round_pack2<T>(x2, x3);
      if (corner_tile) {                                                           // strict upper corner (:318-324)
        uint16_t o[4] = {(uint16_t)(p01 & 0xffffu), (uint16_t)(p01 >> 16), (uint16_t)(p23 & 0xffffu), (uint16_t)(p23 >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = sg0 + r;
          if (s >= L && (s - L) > rr_c) o[r] = Elem<T>::from_f32(Elem<T>::to_f32(o[r]) + fmin_v);
        }
        p01 = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        p23 = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
      }
      if (c < C) *reinterpret_cast<uint2*>(tile + c * LROW + key0) = make_uint2(p01, p23);
    }
  }
  if (PKV_ABLATE(p) >= 1) return;   // measurement aid: no statistics / logits store
  __syncthreads();

  // ---- per-row tile statistics + coalesced 16-B store of the logits tile (CH lanes per row) ----
  const int64_t rowbase = ((int64_t)b * p.H + h0) * w;
  uint16_t* lg_out = reinterpret_cast<uint16_t*>(p.logits);
  const int items = C * CH;
  for (int it = tid; it < items; it += 256) {
    const int row = it / CH;
    const int chunk = it % CH;
    U4 u;
    u.v = *reinterpret_cast<const uint4*>(tile + row * LROW + chunk * 8);
    const int s0 = t_idx * TILE + chunk * 8;
    float xv[8];
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xv[e] = Elem<T>::to_f32(u.h[e]);
      if (s0 + e < S) m = fmaxf(m, xv[e]);
    }
#pragma unroll
    for (int o = CH / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));   // CH-lane group = one row
    float l = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = pkv_exp(xv[e] - m);
      l += (s0 + e < S && m != -INFINITY) ? t : 0.f;
    }
#pragma unroll
    for (int o = CH / 2; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if (chunk == 0) p.partial[(rowbase + row) * p.nT + t_idx] = make_float2(m, l);
    *reinterpret_cast<uint4*>(lg_out + (rowbase + row) * (int64_t)p.Sp + s0) = u.v;
  }
  if (PKV_WGTRACE(p) && tid == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    PKV_WGTRACE(p)[2 * wg] = t_start; PKV_WGTRACE(p)[2 * wg + 1] = wall_clock64();
  }
}

// ------------------------------------------------------------------------------------------------
// finalize_kernel: one workgroup = 1024 output positions of one (b,h); 256 threads x 4 consecutive keys.  Row loads are
// issued 8 rows at a time (one round trip per 8 rows).  The 8 + 8 halo positions the pooling window needs on either side
// are computed as 16 * w single-probability tasks spread over the threads (round 3).  Before, a workgroup computed 1024
// positions and wrote 1008: 33 workgroups per head at S = 32768, 1056 in all on 256 CUs - 32 CUs carried a fifth workgroup
// of a kernel that is bound by vector issue.  1024 outputs per workgroup give 32 / 16 / 8 / 4 workgroups per head at
// S = 32k / 16k / 8k / 4k: whole multiples of the CU count for the BASELINE shapes (measured: 8.2 against 8.4 us at S = 32k).
// ------------------------------------------------------------------------------------------------
constexpr int FN_HALO = 8;                 // positions on either side (pool kernel <= 17)

// W (round 6): the observation window as a compile-time constant (8 = the runners', 32 = the init_* default; 0 = read from p.w).
// Four 256-thread workgroups per CU = 16 waves: every instruction of a wave costs ~16 cycles of wall time, and a run-time
// window means row clamps, row weights and loop tests in every pass over the window rows.
// PPT (round 6): positions per thread, 4 or 8 (1024 or 2048 positions per workgroup).  A third of a workgroup's instructions do
// not depend on the number of positions (the partial merge, the halo, the pooling set-up): with 8 positions per thread a head needs
// half the workgroups and a third fewer instructions in all, and a CU holds 8 instead of 16 of its waves.
template <typename T, int W, int PPT>
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeParams p) {
  constexpr int FN_OUT = 256 * PPT;                                            // positions computed and written per workgroup
  constexpr int NW = PPT / 2;                                                  // 32-bit words (pairs of scores) per thread and row
  __shared__ __attribute__((aligned(16))) uint16_t sc[FN_OUT + 2 * FN_HALO];   // index = position - first position + 8
  __shared__ float rowM[128];           // window <= 128 (check_desc)
  __shared__ float rowS[128];
  __shared__ __attribute__((aligned(16))) float halo_p[2 * FN_HALO * 64];   // rounded probabilities of the halo tasks, [position][window row] (w <= 64 path)
  __shared__ double rs_red[4];

  const int tid = threadIdx.x;
  const int bh = blockIdx.y;
  const int w = W > 0 ? W : p.w;
  const int L = p.S - w;
  const int64_t rowbase = (int64_t)bh * w;
#define PKV_FSTAMP(i) do { if (PKV_TRACE(p) && tid == 0 && blockIdx.x == 1 && bh == 0) PKV_TRACE(p)[i] = (unsigned long long)clock64(); } while (0)
  PKV_FSTAMP(0);
  const unsigned long long t_start = PKV_WGTRACE(p) ? wall_clock64() : 0ull;

  const int p0 = blockIdx.x * FN_OUT;
  const int s0 = p0 + tid * PPT;
  const bool in_row = s0 < L;
  const uint16_t* lgp = reinterpret_cast<const uint16_t*>(p.logits) + rowbase * (int64_t)p.Sp + (in_row ? s0 : 0);
  // row statistics from the per-tile partials: M = max_t m_t, Z = sum_t l_t * exp(m_t - M).
  // 32 lanes per row, 8 rows per pass; every lane issues its (<= 8 per chunk) partial loads back to
  // back (clamped index, masked afterwards) so a pass costs ONE memory round trip, not nT/32.
  {
    const int sub = tid & 31;
    for (int r0 = 0; r0 < w; r0 += 8) {
      const int r = r0 + (tid >> 5);
      const bool live = r < w;
      const float2* pr = p.partial + (rowbase + (live ? r : 0)) * p.nT;
      float m = -INFINITY, z = 0.f;
      for (int c0 = 0; c0 < p.nT; c0 += 256) {          // 256 partials per chunk (S <= 32k at tile 128: one chunk)
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
          const float t = pv[i].y * pkv_exp_stat(pv[i].x - mn);       // exp(-inf - mn) == 0
          zc += (pv[i].x != -INFINITY) ? t : 0.f;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zc += __shfl_xor(zc, o, 64);
        z = (m == -INFINITY ? 0.f : z * pkv_exp_stat(m - mn)) + zc;
        m = mn;
      }
      if (live && sub == 0) { rowM[r] = m; rowS[r] = 1.0f / z; }   // ATen CPU softmax: x * (1 / sum)
    }
  }
  __syncthreads();
  PKV_FSTAMP(1);

  const uint16_t pad = (p.pool_kind == 2) ? Elem<T>::neg_inf() : (uint16_t)0;
  // halo tasks (see below): the first pass's logit is fetched HERE, ahead of the main row loads - fetched after the main
  // arithmetic it would put one more cold round trip on the critical path
  const uint16_t* lgh = reinterpret_cast<const uint16_t*>(p.logits) + rowbase * (int64_t)p.Sp;
  const bool halo_spread = (w & (w - 1)) == 0 && w <= 64;
  const int wsh = __builtin_ctz((unsigned)w);
  auto halo_pos = [&](int i) { return i < FN_HALO ? p0 - FN_HALO + i : p0 + FN_OUT + (i - FN_HALO); };
  uint32_t halo_x0 = 0;
  if (halo_spread) {
    const int i = tid >> wsh, r = tid & (w - 1);
    const int pos = halo_pos(i < 2 * FN_HALO ? i : 0);
    halo_x0 = lgh[(int64_t)r * p.Sp + ((pos >= 0 && pos < L) ? pos : 0)];
  }
  uint16_t ov[PPT];
  if (in_row) {
    float acc[PPT];
#pragma unroll
    for (int e = 0; e < PPT; ++e) acc[e] = 0.f;
    for (int rb = 0; rb < w; rb += 8) {
      // 8 rows per pass, all loads first.  Rows past w are clamped (re-read row w-1) and weighted 0:
      // no control flow between the loads and their uses, so they stay batched (one round trip).
      uint32_t u[8][NW];
      float M[8], RZ[8], wt[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = rb + j < w ? rb + j : w - 1;
        if constexpr (PPT == 8) {
          const u32x4 t4 = *reinterpret_cast<const u32x4*>(lgp + (int64_t)r * p.Sp);
          u[j][0] = t4.x; u[j][1] = t4.y; u[j][2] = t4.z; u[j][3] = t4.w;
        } else {
          const u32x2 t2 = *reinterpret_cast<const u32x2*>(lgp + (int64_t)r * p.Sp);
          u[j][0] = t2.x; u[j][1] = t2.y;
        }
        M[j] = rowM[r];
        RZ[j] = rowS[r];
        wt[j] = rb + j < w ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // two keys per packed instruction: x - M, exp, * (1/Z), round, accumulate (same operations as the scalar form)
        const uint32_t* uu = u[j];
        const pkv_f32x2 mm = {M[j], M[j]}, rz = {RZ[j], RZ[j]}, ww = {wt[j], wt[j]};
#pragma unroll
        for (int h2 = 0; h2 < NW; ++h2) {
          const pkv_f32x2 x = {Elem<T>::to_f32((uint16_t)(uu[h2] & 0xffffu)), Elem<T>::to_f32((uint16_t)(uu[h2] >> 16))};
          const pkv_f32x2 pr = pkv_exp_pair(x - mm) * rz;                           // fp32 softmax (:326)
          const uint32_t pk2 = round_pack2<T>(pr.x, pr.y);                          // .to(dtype)
          const pkv_f32x2 pv = {Elem<T>::to_f32((uint16_t)(pk2 & 0xffffu)), Elem<T>::to_f32((uint16_t)(pk2 >> 16))};
          const pkv_f32x2 a2 = __builtin_elementwise_fma(ww, pv, pkv_f32x2{acc[2 * h2], acc[2 * h2 + 1]});   // fp32 row accumulate (:327)
          acc[2 * h2] = a2.x;
          acc[2 * h2 + 1] = a2.y;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
      const float v = (p.reduce == 1) ? (acc[e] / (float)w) : acc[e];             // mean (:661) or sum (:327)
      ov[e] = (s0 + e < L) ? Elem<T>::from_f32(v) : pad;
    }
  } else {
#pragma unroll
    for (int e = 0; e < PPT; ++e) ov[e] = pad;
  }
  PKV_FSTAMP(2);
  if constexpr (PPT == 8) {
    uint4 pk;
    pk.x = (uint32_t)ov[0] | ((uint32_t)ov[1] << 16);
    pk.y = (uint32_t)ov[2] | ((uint32_t)ov[3] << 16);
    pk.z = (uint32_t)ov[4] | ((uint32_t)ov[5] << 16);
    pk.w = (uint32_t)ov[6] | ((uint32_t)ov[7] << 16);
    *reinterpret_cast<uint4*>(sc + FN_HALO + tid * 8) = pk;
  } else {
    uint2 pk;
    pk.x = (uint32_t)ov[0] | ((uint32_t)ov[1] << 16);
    pk.y = (uint32_t)ov[2] | ((uint32_t)ov[3] << 16);
    *reinterpret_cast<uint2*>(sc + FN_HALO + tid * 4) = pk;
  }
  // ---- halo: positions p0-8 .. p0-1 and p0+1024 .. p0+1031, the same arithmetic, one (position, window row) task per thread
  //      and pass.  The w probabilities of a position sit in w consecutive lanes of ONE wave (w a power of two <= 64), go
  //      through LDS and are summed by the group's first lane in row order - the fp32 order of the main path. ----
  {
    if (halo_spread) {
      const int ntask = 2 * FN_HALO * w;
      for (int t0 = 0; t0 < ntask; t0 += 256) {
        const int task = t0 + tid;
        const int i = task >> wsh, r = task & (w - 1);
        const int pos = halo_pos(i < 2 * FN_HALO ? i : 0);
        const bool ok = task < ntask && pos >= 0 && pos < L;
        float pr = 0.f;
        if (ok) {
          const uint32_t x = t0 == 0 ? halo_x0 : (uint32_t)lgh[(int64_t)r * p.Sp + pos];
          const pkv_f32x2 xx = {Elem<T>::to_f32((uint16_t)x), Elem<T>::to_f32((uint16_t)x)};
          const pkv_f32x2 e2 = pkv_exp_pair(xx - pkv_f32x2{rowM[r], rowM[r]}) * pkv_f32x2{rowS[r], rowS[r]};
          const uint32_t pk2 = round_pack2<T>(e2.x, e2.y);
          pr = Elem<T>::to_f32((uint16_t)(pk2 & 0xffffu));
        }
        if (task < ntask) halo_p[task] = pr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (task < ntask && r == 0) {
          float acc = 0.f;
          if (w == 8) {                                             // the runners' window: both reads in flight, then the 8 adds
            const float4 a = *reinterpret_cast<const float4*>(halo_p + task), b = *reinterpret_cast<const float4*>(halo_p + task + 4);
            const float hv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) acc = __builtin_fmaf(1.0f, hv[rr], acc);
          } else {
            for (int rr = 0; rr < w; ++rr) acc = __builtin_fmaf(1.0f, halo_p[task + rr], acc);
          }
          const float v = (p.reduce == 1) ? (acc / (float)w) : acc;
          sc[i < FN_HALO ? i : FN_OUT + i] = ok ? Elem<T>::from_f32(v) : pad;
        }
      }
    } else if (tid < 2 * FN_HALO) {                               // other windows: one thread per halo position
      const int pos = halo_pos(tid);
      const bool ok = pos >= 0 && pos < L;
      float acc = 0.f;
      for (int r = 0; r < w; ++r) {
        const uint32_t x = lgh[(int64_t)r * p.Sp + (ok ? pos : 0)];
        const pkv_f32x2 xx = {Elem<T>::to_f32((uint16_t)x), Elem<T>::to_f32((uint16_t)x)};
        const pkv_f32x2 e2 = pkv_exp_pair(xx - pkv_f32x2{rowM[r], rowM[r]}) * pkv_f32x2{rowS[r], rowS[r]};
        const uint32_t pk2 = round_pack2<T>(e2.x, e2.y);
        acc = __builtin_fmaf(1.0f, Elem<T>::to_f32((uint16_t)(pk2 & 0xffffu)), acc);
      }
      const float v = (p.reduce == 1) ? (acc / (float)w) : acc;
      sc[tid < FN_HALO ? tid : FN_OUT + tid] = ok ? Elem<T>::from_f32(v) : pad;
    }
  }
  __syncthreads();
  PKV_FSTAMP(3);

  if (PKV_WGTRACE(p) && tid == 0) {
    const size_t wg = 65536 + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    PKV_WGTRACE(p)[2 * wg] = t_start; PKV_WGTRACE(p)[2 * wg + 1] = wall_clock64();
  }
  const bool writer = s0 < L;                                 // positions past the row write nothing
  uint16_t res[PPT];
  const int half = p.pool_kernel >> 1;
  if (p.pool_kind == 0 || !writer) {
#pragma unroll
    for (int e = 0; e < PPT; ++e) res[e] = ov[e];
  } else {
    // the PPT + 16 staged scores around this thread's positions: 8-byte (16-byte) LDS reads issued together
    float v[PPT + 16];
    if constexpr (PPT == 8) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const uint4 t = *reinterpret_cast<const uint4*>(sc + tid * 8 + i * 8);
        const uint32_t tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[i * 8 + 2 * q] = Elem<T>::to_f32((uint16_t)(tw[q] & 0xffffu));
          v[i * 8 + 2 * q + 1] = Elem<T>::to_f32((uint16_t)(tw[q] >> 16));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const uint2 t = *reinterpret_cast<const uint2*>(sc + tid * 4 + i * 4);
        v[i * 4 + 0] = Elem<T>::to_f32((uint16_t)(t.x & 0xffffu));
        v[i * 4 + 1] = Elem<T>::to_f32((uint16_t)(t.x >> 16));
        v[i * 4 + 2] = Elem<T>::to_f32((uint16_t)(t.y & 0xffffu));
        v[i * 4 + 3] = Elem<T>::to_f32((uint16_t)(t.y >> 16));
      }
    }
    // the runners' kernel sizes (run_longbench.py: maxpool 7, avgpool 5) get compile-time windows: the generic loop is 17
    // predicated taps per output, ~70 instructions per position
    auto pool_at = [&](int e, int hw, bool is_max) -> uint16_t {
      if (is_max) {                                                              // max_pool1d, -inf padding (:331)
        float m = -INFINITY;
#pragma unroll
        for (int j = -8; j <= 8; ++j)
          if (j >= -hw && j <= hw) m = fmaxf(m, v[8 + e + j]);
        return Elem<T>::from_f32(m);
      }
      float sum = 0.f;                                                           // avg_pool1d, zero padding, / kernel (:329)
#pragma unroll
      for (int j = -8; j <= 8; ++j)
        if (j >= -hw && j <= hw) sum += v[8 + e + j];                            // left-to-right fp32 sum
      return Elem<T>::from_f32(sum / (float)(2 * hw + 1));
    };
    if (p.pool_kind == 2 && half == 3) {
#pragma unroll
      for (int e = 0; e < PPT; ++e) res[e] = pool_at(e, 3, true);
    } else if (p.pool_kind == 1 && half == 2) {
#pragma unroll
      for (int e = 0; e < PPT; ++e) res[e] = pool_at(e, 2, false);
    } else {
#pragma unroll
      for (int e = 0; e < PPT; ++e) res[e] = pool_at(e, half, p.pool_kind == 2);
    }
  }
  // per-chunk maxima (8 consecutive positions: an even/odd lane pair, or - 8 positions per thread - the thread's own) for the top-k prefilter
  if (p.cmax) {
    float m4 = -INFINITY;
    uint16_t b4 = Elem<T>::neg_inf();
    if (writer) {
#pragma unroll
      for (int e = 0; e < PPT; ++e) {
        const float x = Elem<T>::to_f32(res[e]);
        if (s0 + e < L && x > m4) { m4 = x; b4 = res[e]; }
      }
    }
    if constexpr (PPT == 8) {
      if (writer) reinterpret_cast<uint16_t*>(p.cmax)[(int64_t)bh * p.cmax_stride + (s0 >> 3)] = b4;
    } else {
      const float mo = __shfl_xor(m4, 1, 64);
      const uint32_t bo = __shfl_xor((uint32_t)b4, 1, 64);
      if (writer && !(tid & 1))
        reinterpret_cast<uint16_t*>(p.cmax)[(int64_t)bh * p.cmax_stride + (s0 >> 3)] = (mo > m4) ? (uint16_t)bo : b4;
    }
  }
  if (p.rowsum_part) {        // Ada-SnapKV: this workgroup's share of the sum over all scores of the row (:710), fp64
    double sa = 0.0;
    if (writer) {
#pragma unroll
      for (int e = 0; e < PPT; ++e) if (s0 + e < L) sa += (double)Elem<T>::to_f32(res[e]);
    }
    sa = wave_sum_f64(sa);
    if ((tid & 63) == 0) rs_red[tid >> 6] = sa;
    __syncthreads();
    if (tid == 0) p.rowsum_part[(int64_t)bh * p.rowsum_np + blockIdx.x] = ((rs_red[0] + rs_red[1]) + rs_red[2]) + rs_red[3];
  }
  if (!writer) return;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.scores) + (int64_t)bh * p.scores_stride + s0;
  if constexpr (PPT == 8) {
    uint4 ro;
    ro.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
    ro.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
    ro.z = (uint32_t)res[4] | ((uint32_t)res[5] << 16);
    ro.w = (uint32_t)res[6] | ((uint32_t)res[7] << 16);
    *reinterpret_cast<uint4*>(out) = ro;    // stride % 8 == 0, s0 % 8 == 0, stride >= roundup(L,8): aligned, in bounds
  } else {
    uint2 ro;
    ro.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
    ro.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
    *reinterpret_cast<uint2*>(out) = ro;    // stride % 8 == 0, s0 % 4 == 0, stride >= roundup(L,8): aligned, in bounds
  }
  if (PKV_TRACE(p) && tid == 2 && blockIdx.x == 1 && bh == 0) PKV_TRACE(p)[4] = (unsigned long long)clock64();
#undef PKV_FSTAMP
}

// (m, l) of two disjoint key sets -> (max, sum exp relative to it); -inf-safe
__device__ __forceinline__ void stat_merge(float& m, float& l, float m2, float l2) {
  const float M = fmaxf(m, m2);
  const float Ms = (M == -INFINITY) ? 0.f : M;
  l = l * pkv_exp(m - Ms) + l2 * pkv_exp(m2 - Ms);
  m = M;
}

// ------------------------------------------------------------------------------------------------
// logits2_kernel: the same result as logits_kernel, organised as a software pipeline.
//
// A workgroup owns `nst` consecutive 128-key stages of one (batch, kv-head group).  Per stage and wave:
// 8 nontemporal 1-KB row-major loads (4 whole K rows per instruction) land in registers while the previous
// stage is being processed, are dropped into a wave-private XOR-swizzled LDS area, and come back as MFMA
// fragments.  The rounded logits of a stage go through a double-buffered [C][128] LDS tile to 256-B row
// segments in memory; (max, sum exp) per column are carried in registers and merged once per workgroup, so
// finalize_kernel reads one partial per (workgroup, row).  Loads of stage h+1 are in flight during the MFMA,
// epilogue, statistics and stores of stage h: the K stream never waits for a store acknowledgement or a
// workgroup launch (measured: the one-tile-per-workgroup kernel spends 20-25 % of its time there).
// ------------------------------------------------------------------------------------------------
// WG (round 6): window and GQA group as ONE compile-time constant (window * 256 + group; 0 = read from p): the column count, the
// column -> (head, window row) map, the corner-mask row and the trip count of the store loop become constants.
template <typename T, int NCT, bool NT, int WG>
__global__ __launch_bounds__(256) void logits2_kernel(LogitsParams p) {
  constexpr int HT = 128, LROW = HT + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int w = WG ? (WG >> 8) : p.w, G = WG ? (WG & 255) : p.G;
  const int C = G * w, S = p.S, L = S - w;
  u32x4* kst_all = reinterpret_cast<u32x4*>(smem_raw);                              // [4 waves][32 rows][16 chunks]
  uint16_t* tile0 = reinterpret_cast<uint16_t*>(smem_raw + 32768);                  // [2][C][LROW]
  float2* wst = reinterpret_cast<float2*>(tile0 + 2 * C * LROW);                    // [4][C]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  u32x4* kst = kst_all + wave * 512;
  const int chunk_i = blockIdx.x;               // chunk of nst stages within the head
  const int grp = blockIdx.y;
  const int HG = p.H / G;
  const int b = grp / HG;
  const int hk = grp - b * HG;
  const int h0 = hk * G;
  const int start = chunk_i * p.nst * HT;
  const int nh = min(p.nst, (S - start + HT - 1) / HT);      // stages that hold at least one key (>= 1)
  const float fmin_v = Elem<T>::finfo_min();
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b;
  const int64_t rowbase = ((int64_t)b * p.H + h0) * w;
  uint16_t* lg_out = reinterpret_cast<uint16_t*>(p.logits);
  // raw buffer over the logits workspace (sc1 stores need the buffer form: cache-policy bits are an operand of the builtin)
  const __amdgpu_buffer_rsrc_t lg_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.logits, 0, (int)p.logits_bytes, 0x00020000);

  u32x4 pre[8];
  auto issue = [&](int h) {
    const int s_wave = start + h * HT + wave * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rl = 4 * j + lg;
      int s = s_wave + rl;
      s = s < S ? s : S - 1;            // clamp: stay in bounds; keys past S are masked out of the statistics
      const u32x4* ptr = reinterpret_cast<const u32x4*>(kbase + (int64_t)s * p.ks_s) + (li ^ (rl & 15));
      pre[j] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    }
  };
  issue(0);

  u32x4 qf[NCT][4];
#pragma unroll
  for (int n = 0; n < NCT; ++n) {
    const int c = n * 16 + li;
    if (c < C) {
      const int hh = h0 + c / w;
      const int rr = c - (c / w) * w;
      const uint16_t* qrow = qb + (int64_t)hh * p.qs_h + (int64_t)(L + rr) * p.qs_s + lg * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[n][kk] = *reinterpret_cast<const u32x4*>(qrow + kk * 32);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[n][kk] = u32x4{0, 0, 0, 0};
    }
  }
  float m_run[NCT], l_run[NCT];
#pragma unroll
  for (int n = 0; n < NCT; ++n) { m_run[n] = -INFINITY; l_run[n] = 0.f; }

  // logits of stage h: C rows x 256 B out of tile[h & 1], 16 B per item
  auto store_stage = [&](int h) {
    if (PKV_ABLATE(p) == 1 || PKV_ABLATE(p) == 2 || PKV_ABLATE(p) == 4) return;    // measurement aids: no logits store
    const uint16_t* tile = tile0 + (h & 1) * C * LROW;
    const int s_stage = start + h * HT;
    for (int it = tid; it < C * 16; it += 256) {
      const int row = it >> 4, ch = it & 15;
      const uint4 v = *reinterpret_cast<const uint4*>(tile + row * LROW + ch * 8);
      if (PKV_ABLATE(p) == 3) {              // measurement aid: the same stores into a per-workgroup 2 KB patch that stays in L2
        *reinterpret_cast<uint4*>(lg_out + ((int64_t)((blockIdx.y * gridDim.x + blockIdx.x) & 2047) * 128 + it) * 8) = v;
        continue;
      }
      uint16_t* dst = lg_out + (rowbase + row) * (int64_t)p.Sp + s_stage + ch * 8;
      if (p.st_mode == 2) {
        // write-through (sc1): the 16.8 MB of logits leave the XCD's L2 while the K stream is still running instead of
        // as one write-back burst at the kernel boundary (MI355X_MICROARCH.md: a boundary costs + dirty bytes / 6 TB/s)
        const uint32_t voff = (uint32_t)(reinterpret_cast<const char*>(dst) - reinterpret_cast<const char*>(lg_out));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), lg_rsrc, voff, 0, 16);
      } else if (p.st_mode == 1) {
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(dst));
      } else {
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  };

  for (int h = 0; h < nh; ++h) {
    // land stage h in this wave's staging area, then put stage h+1 in flight
#pragma unroll
    for (int j = 0; j < 8; ++j) kst[j * 64 + lane] = pre[j];
    if (h + 1 < nh) issue(h + 1);       // nothing else is outstanding here, so the branch costs no extra wait
    // The logits of stage h-1 leave HERE, one stage late: vmcnt counts loads and stores in one in-order queue, so the wait
    // for the K rows of stage h+1 (top of the next iteration) also waits for every store issued before it.  Stores issued
    // at the end of a stage were acknowledged at that wait - a full write-through round trip exposed per stage; issued
    // here they have the whole stage to complete.  (tile[(h-1) & 1] is complete since the barrier that closed stage h-1 and
    // is not written again before the barrier that closes this stage.)
    if (h > 0) store_stage(h - 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32x4 kf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[t][kk] = kst[(t * 16 + li) * 16 + ((kk * 4 + lg) ^ li)];

    uint16_t* tile = tile0 + (h & 1) * C * LROW;
    const int s_stage = start + h * HT;
    const bool edge = s_stage + HT > L;                       // stage touches the window corner and/or keys >= S
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
      const int c = n * 16 + li;
      const int rr_c = c % w;
      float xs[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma<T>::run(kf[t][kk], qf[n][kk], acc);
        const int key0 = wave * 32 + t * 16 + lg * 4;          // key within the stage of acc[0]
        uint32_t p01 = round_pack2<T>(acc[0], acc[1]);                               // matmul output dtype (:317)
        uint32_t p23 = round_pack2<T>(acc[2], acc[3]);
        float x0 = Elem<T>::to_f32((uint16_t)(p01 & 0xffffu)), x1 = Elem<T>::to_f32((uint16_t)(p01 >> 16));
        float x2 = Elem<T>::to_f32((uint16_t)(p23 & 0xffffu)), x3 = Elem<T>::to_f32((uint16_t)(p23 >> 16));
        x0 = scale_logit<T>(x0, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);              // "/ math.sqrt(head_dim)" (:317)
        x1 = scale_logit<T>(x1, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
        x2 = scale_logit<T>(x2, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
        x3 = scale_logit<T>(x3, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);
        p01 = round_pack2<T>(x0, x1);
        p23 = round_pack2<T>(x2, x3);
        uint16_t o[4] = {(uint16_t)(p01 & 0xffffu), (uint16_t)(p01 >> 16), (uint16_t)(p23 & 0xffffu), (uint16_t)(p23 >> 16)};
        bool dead[4] = {false, false, false, false};                                 // masked, or past the end of the row
        if (edge) {                                                                  // strict upper corner (:318-324)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s = s_stage + key0 + r;
            const bool masked = s >= L && (s - L) > rr_c;
            if (masked) o[r] = Elem<T>::from_f32(Elem<T>::to_f32(o[r]) + fmin_v);
            dead[r] = masked || s >= S;
          }
          p01 = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          p23 = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        }
        if (c < C) *reinterpret_cast<uint2*>(tile + c * LROW + key0) = make_uint2(p01, p23);
        // In the statistics a masked entry is -inf: exp(finfo.min - M) is exactly 0 for every real maximum M, and a lane
        // group that sees ONLY masked keys must not carry the finite finfo.min as its maximum (-max * log2e overflows)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[t * 4 + r] = dead[r] ? -INFINITY : Elem<T>::to_f32(o[r]);
      }
      float m_loc = xs[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m_loc = fmaxf(m_loc, xs[i]);
      const float m_new = fmaxf(m_run[n], m_loc);
      if (PKV_ABLATE(p) == 2) { m_run[n] = m_new; continue; }   // measurement aid: no exponentials
      const float ms = (m_new == -INFINITY) ? 0.f : m_new;
      float sum = 0.f;
      if (p.fexp) {
        // The partial statistics only feed Z = sum_t l_t exp(m_t - M), whose fp32 summation order already differs from
        // ATen's: the hardware exponential (v_exp_f32 on x*log2e - m*log2e, one FMA per logit, ~1e-6 relative per term,
        // signs random) moves Z by ~1e-7 relative - below the order noise.  The probabilities themselves
        // (finalize_kernel) keep the accurate exponential.
        const float L2E = 1.44269502162933349609375f;
        const float msl = ms * L2E;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += __builtin_amdgcn_exp2f(fmaf(xs[i], L2E, -msl));
        l_run[n] = l_run[n] * __builtin_amdgcn_exp2f(fmaf(m_run[n], L2E, -msl)) + sum;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += pkv_exp(xs[i] - ms);
        l_run[n] = l_run[n] * pkv_exp(m_run[n] - ms) + sum;
      }
      m_run[n] = m_new;
    }
    __syncthreads();
  }
  store_stage(nh - 1);
  // per column: merge the 4 key-group lanes, then the 4 waves, one partial per (workgroup, row)
#pragma unroll
  for (int n = 0; n < NCT; ++n) {
    float m = m_run[n], l = l_run[n];
    stat_merge(m, l, __shfl_xor(m, 16, 64), __shfl_xor(l, 16, 64));
    stat_merge(m, l, __shfl_xor(m, 32, 64), __shfl_xor(l, 32, 64));
    const int c = n * 16 + li;
    if (lg == 0 && c < C) wst[wave * C + c] = make_float2(m, l);
  }
  __syncthreads();
  if (tid < C) {
    float2 a = wst[tid];
#pragma unroll
    for (int wv = 1; wv < 4; ++wv) { const float2 o = wst[wv * C + tid]; stat_merge(a.x, a.y, o.x, o.y); }
    p.partial[(rowbase + tid) * p.nT + chunk_i] = a;
  }
}

size_t logits2_lds_bytes(int C) { return 32768 + (size_t)2 * C * 136 * 2 + (size_t)4 * C * 8; }

hipError_t launch_logits2(int dtype, const LogitsParams& p, hipStream_t st) {
  const int C = p.G * p.w;
  const int nct = (C + 15) / 16;
  dim3 grid(p.nT, p.B * (p.H / p.G));
  const size_t lds = logits2_lds_bytes(C);
  // window 8 on expanded K (the reference's contract, the headline) gets its own instantiation: 112 instead of 128 registers,
  // 45.65 -> 45.45 us.  The same for un-expanded GQA K (window 8, group 4) was measured and is NOT instantiated: 16.95 -> 17.95 us
  // (profiles/r06/ab/logits_compile_time_window_group_ab.txt) - that scan is sensitive to its schedule, not to its scalar work.
#define PKV_L2(TT, NCT, NTL, WGC) PKV_KLAUNCH((logits2_kernel<TT, NCT, NTL, WGC>), grid, dim3(256), lds, st, p)
#define PKV_L2D(TT, NTL)                                                                                   \
  do {                                                                                                     \
    if (p.w == 8 && p.G == 1) PKV_L2(TT, 1, NTL, 8 * 256 + 1);                                             \
    else if (nct == 1) PKV_L2(TT, 1, NTL, 0);                                                              \
    else PKV_L2(TT, 2, NTL, 0);                                                                            \
  } while (0)
  if (dtype == 0) { if (p.nt) PKV_L2D(BF16, true); else PKV_L2D(BF16, false); }
  else            { if (p.nt) PKV_L2D(F16, true); else PKV_L2D(F16, false); }
#undef PKV_L2D
#undef PKV_L2
  return hipGetLastError();
}

hipError_t launch_logits(int dtype, const LogitsParams& p, hipStream_t st) {
  const int C = p.G * p.w;
  dim3 grid(p.nT, p.B * (p.H / p.G));
  const int tile = p.tile;
  const size_t lds = (size_t)C * (tile + 8) * sizeof(uint16_t);
#define PKV_LAUNCH(TT, KPW, NT, KS)                                                                                     \
  do {                                                                                                                 \
    if (lds > 64 * 1024) {  /* wide GQA groups x wide windows: opt in to more than 64 KB of dynamic LDS */           \
      hipError_t e_ = dyn_lds(reinterpret_cast<const void*>(logits_kernel<TT, KPW, NT, KS>), lds);                      \
      if (e_ != hipSuccess) return e_;                                                                                \
    }                                                                                                                  \
    PKV_KLAUNCH((logits_kernel<TT, KPW, NT, KS>), grid, dim3(256), lds, st, p);                                       \
  } while (0)
#define PKV_LAUNCH_D(TT, KPW, NT)                                                                                      \
  do {                                                                                                                 \
    if (p.D == 64) PKV_LAUNCH(TT, KPW, NT, 2); else if (p.D == 256) PKV_LAUNCH(TT, KPW, NT, 8); else PKV_LAUNCH(TT, KPW, NT, 4); \
  } while (0)
  if (dtype == 0) {
    if (tile == 128) { if (p.nt) PKV_LAUNCH_D(BF16, 32, true); else PKV_LAUNCH_D(BF16, 32, false); }
    else             { if (p.nt) PKV_LAUNCH_D(BF16, 64, true); else PKV_LAUNCH_D(BF16, 64, false); }
  } else {
    if (tile == 128) { if (p.nt) PKV_LAUNCH_D(F16, 32, true); else PKV_LAUNCH_D(F16, 32, false); }
    else             { if (p.nt) PKV_LAUNCH_D(F16, 64, true); else PKV_LAUNCH_D(F16, 64, false); }
  }
#undef PKV_LAUNCH_D
#undef PKV_LAUNCH
  return hipGetLastError();
}

// positions per thread of finalize_kernel for a shape: 8 (2048 positions per workgroup) while that still leaves 512 workgroups
// (two per CU), else 4.  Measured (tools/probes/finalize_ppt_probe.py, H = 32, window 8): B = 8, S = 32k 50.0 -> 44.5 us; B = 4 21.9 ->
// 19.4; B = 8, S = 8k 11.9 -> 10.7; B = 1, S = 32k 7.2 -> 7.0; but B = 1, S = 8k 4.5 -> 4.9 and S = 4k 4.3 -> 4.5: a launch that
// does not fill the chip is one latency chain, and the longer workgroup only stretches it.
static int finalize_ppt(int S, int w, int BH) {
  static int forced = [] { const char* v = getenv("PKV_FINALIZE_PPT"); return v && *v ? atoi(v) : 0; }();
  if (forced == 4 || forced == 8) return forced;
  const int64_t wgs8 = (int64_t)BH * ((S - w + 2047) / 2048);
  return wgs8 >= 512 ? 8 : 4;
}
int finalize_blocks(int S, int w, int BH) { const int out = 256 * finalize_ppt(S, w, BH); return (S - w + out - 1) / out; }

hipError_t launch_finalize(int dtype, const FinalizeParams& p, hipStream_t st) {
  const int ppt = finalize_ppt(p.S, p.w, p.B * p.H);
  dim3 grid(finalize_blocks(p.S, p.w, p.B * p.H), p.B * p.H);
#define PKV_FIN2(TT, PP)                                                                                         \
  do {                                                                                                         \
    if (p.w == 8) PKV_KLAUNCH((finalize_kernel<TT, 8, PP>), grid, dim3(256), 0, st, p);                        \
    else if (p.w == 32) PKV_KLAUNCH((finalize_kernel<TT, 32, PP>), grid, dim3(256), 0, st, p);                 \
    else PKV_KLAUNCH((finalize_kernel<TT, 0, PP>), grid, dim3(256), 0, st, p);                                 \
  } while (0)
#define PKV_FIN(TT) do { if (ppt == 8) PKV_FIN2(TT, 8); else PKV_FIN2(TT, 4); } while (0)
  if (dtype == 0) PKV_FIN(BF16); else PKV_FIN(F16);
#undef PKV_FIN
#undef PKV_FIN2
  return hipGetLastError();
}

}  // namespace pkv
