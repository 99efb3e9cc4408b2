// pkv_f32.hip — the window-score / top-k path for fp32 tensors (gfx950).
//
// The reference is dtype-generic (pyramidkv_utils.py:317-346 on fp32 tensors: fp32 matmul, fp32 softmax, no intermediate
// rounding); real models hand over bf16 / fp16, so this path is built for completeness and correctness, HBM-bound like the
// 16-bit one but not tuned to the same degree:
//   logits_f32_kernel    :317-324  Q[-w:] K^T / sqrt(D) on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate),
//                                  causal corner mask, per-tile (max, sum exp) partials
//   finalize_f32_kernel  :326-331  softmax over all S keys, sum / mean of the window rows, avg / max pool
//   topk_f32_kernel      :334      k largest of a row in (value desc, index asc) order: 4 x 8-bit radix select on the
//                                  order-preserving 32-bit keys, ties by position, bitonic sort of the k winners in LDS
// The gather needs no fp32 kernel: a [.., D] fp32 tensor is a [.., 2D] 16-bit tensor with doubled strides (pkv_api.hip).
// fp32 sums whose order is not pinned by the reference (the D-long dot product, the softmax denominator) differ from ATen's
// CPU kernels in the last bits; tests compare scores with a relative tolerance and selections up to score ties within it.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

// ------------------------------------------------------------------------------------------------
// logits_f32_kernel: one workgroup = 64 keys of one (batch, kv-head group), 16 keys per wave; the C = G*w query rows of
// the group in tiles of 16 columns.  MFMA 16x16x4 f32: A = keys (row l%16, k-slot l/16), B = queries (column l%16,
// k-slot l/16).  Every lane loads float4 pieces (columns 16j + 4*(l/16) .. +3 of its row), so MFMA number 4j+i multiplies
// element 16j + 4*(l/16) + i of both operands: the dot product's terms are all there, in a permuted order.
// ------------------------------------------------------------------------------------------------
template <int KS>                                   // float4 pieces per lane and row: D / 16
__global__ __launch_bounds__(256) void logits_f32_kernel(LogitsParams p) {
  __shared__ float2 wst[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tile = blockIdx.x, grp = blockIdx.y;
  const int HG = p.H / p.G;
  const int b = grp / HG, hk = grp - b * HG, h0 = hk * p.G;
  const int w = p.w, C = p.G * w, S = p.S, L = S - w;
  const int s_wave = tile * 64 + wave * 16;
  const float* kb = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float* qb = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.qs_b;
  const int64_t rowbase = ((int64_t)b * p.H + h0) * w;
  float* lg_out = reinterpret_cast<float*>(p.logits);
  const float fmin_v = -3.4028234663852886e38f;    // torch.finfo(torch.float32).min

  f32x4 kf[KS];
  {
    const int s = s_wave + li;
    const float* kr = kb + (int64_t)(s < S ? s : S - 1) * p.ks_s + 4 * lg;       // clamp: keys past S are masked below
#pragma unroll
    for (int j = 0; j < KS; ++j) kf[j] = *reinterpret_cast<const f32x4*>(kr + 16 * j);
  }
  const int nct = (C + 15) / 16;
  for (int n = 0; n < nct; ++n) {
    const int c = n * 16 + li;
    const int cc = c < C ? c : C - 1;
    const int hh = h0 + cc / w, rr = cc - (cc / w) * w;
    const float* qr = qb + (int64_t)hh * p.qs_h + (int64_t)(L + rr) * p.qs_s + 4 * lg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + 16 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][i], qv[i], acc, 0, 0, 0);
    }
    // lane (li, lg): column c, keys s_wave + 4*lg + r
    float x[4];
    float m4 = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = s_wave + 4 * lg + r;
      float v = p.scale_mode == 0 ? div_const(acc[r], p.sqrt_d, p.rcp_sqrt_d) : acc[r] * p.rcp_sqrt_d;   // / math.sqrt(head_dim) (:317)
      if (s >= L && (s - L) > rr) v = v + fmin_v;                                                          // strict upper corner (:318-324)
      x[r] = v;
      if (s < S) m4 = fmaxf(m4, v);
    }
    if (c < C && s_wave + 4 * lg < S)                 // Sp is a multiple of 256: the 4 floats are in bounds
      *reinterpret_cast<f32x4*>(lg_out + (rowbase + c) * (int64_t)p.Sp + s_wave + 4 * lg) = f32x4{x[0], x[1], x[2], x[3]};
    float l4 = 0.f;
    const float ms = (m4 == -INFINITY) ? 0.f : m4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (s_wave + 4 * lg + r < S) l4 += pkv_exp(x[r] - ms);
    // merge the four key groups of the wave, then the four waves
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float mo = __shfl_xor(m4, o, 64), lo = __shfl_xor(l4, o, 64);
      const float M = fmaxf(m4, mo), Ms = (M == -INFINITY) ? 0.f : M;
      l4 = l4 * pkv_exp(m4 - Ms) + lo * pkv_exp(mo - Ms);
      m4 = M;
    }
    __syncthreads();                                   // wst of the previous column tile has been read
    if (lg == 0) wst[wave][li] = make_float2(m4, l4);
    __syncthreads();
    if (tid < 16 && n * 16 + tid < C) {
      float2 a = wst[0][tid];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) {
        const float2 o = wst[wv][tid];
        const float M = fmaxf(a.x, o.x), Ms = (M == -INFINITY) ? 0.f : M;
        a.y = a.y * pkv_exp(a.x - Ms) + o.y * pkv_exp(o.x - Ms);
        a.x = M;
      }
      p.partial[(rowbase + n * 16 + tid) * p.nT + tile] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// finalize_f32_kernel: one workgroup = 1008 output positions (+8 halo each side) of one (b,h), 256 threads x 4 positions.
// ------------------------------------------------------------------------------------------------
constexpr int F32_SPAN = 1024, F32_OUT = F32_SPAN - 16;

__global__ __launch_bounds__(256) void finalize_f32_kernel(FinalizeParams p) {
  __shared__ __attribute__((aligned(16))) float sc[F32_SPAN];
  __shared__ float rowM[128], rowS[128];
  const int tid = threadIdx.x, bh = blockIdx.y;
  const int w = p.w, L = p.S - w;
  const int64_t rowbase = (int64_t)bh * w;
  // row statistics from the per-tile partials: M = max_t m_t, Z = sum_t l_t * exp(m_t - M); 32 lanes per row
  {
    const int sub = tid & 31;
    for (int r0 = 0; r0 < w; r0 += 8) {
      const int r = r0 + (tid >> 5);
      const bool live = r < w;
      const float2* pr = p.partial + (rowbase + (live ? r : 0)) * p.nT;
      float m = -INFINITY, z = 0.f;
      for (int t = sub; t < p.nT; t += 32) {
        const float2 pv = pr[t];
        const float mn = fmaxf(m, pv.x);
        if (mn != -INFINITY) z = z * pkv_exp(m - mn) + pv.y * pkv_exp(pv.x - mn);
        m = mn;
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float mo = __shfl_xor(m, o, 64), zo = __shfl_xor(z, o, 64);
        const float mn = fmaxf(m, mo);
        if (mn != -INFINITY) z = z * pkv_exp(m - mn) + zo * pkv_exp(mo - mn);
        m = mn;
      }
      if (live && sub == 0) { rowM[r] = m; rowS[r] = 1.0f / z; }     // ATen CPU softmax: exp(x - max) * (1 / sum)
    }
  }
  __syncthreads();
  const int s0 = blockIdx.x * F32_OUT - 8 + tid * 4;
  const float pad = (p.pool_kind == 2) ? -INFINITY : 0.f;
  float ov[4] = {pad, pad, pad, pad};
  if (s0 >= 0 && s0 < L) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* lgp = reinterpret_cast<const float*>(p.logits) + rowbase * (int64_t)p.Sp + s0;
    for (int r = 0; r < w; ++r) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(lgp + (int64_t)r * p.Sp);
      const float M = rowM[r], RZ = rowS[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += pkv_exp(u[e] - M) * RZ;           // fp32 softmax (:326), rows added in order (:327)
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = (p.reduce == 1) ? (acc[e] / (float)w) : acc[e];         // mean (:661) or sum (:327)
      if (s0 + e < L) ov[e] = v;
    }
  }
  *reinterpret_cast<f32x4*>(sc + tid * 4) = f32x4{ov[0], ov[1], ov[2], ov[3]};
  __syncthreads();
  if (tid < 2 || tid >= 254 || s0 >= L) return;                              // halo threads / positions past the row
  float res[4];
  const int half = p.pool_kernel >> 1;
  if (p.pool_kind == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) res[e] = ov[e];
  } else {
    float v[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) v[i] = sc[tid * 4 - 8 + i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float m = -INFINITY, sum = 0.f;
#pragma unroll
      for (int j = -8; j <= 8; ++j)
        if (j >= -half && j <= half) { m = fmaxf(m, v[8 + e + j]); sum += v[8 + e + j]; }   // left-to-right fp32 sum
      res[e] = p.pool_kind == 2 ? m : sum / (float)p.pool_kernel;            // max_pool1d (:331) / avg_pool1d (:329)
    }
  }
  float* out = reinterpret_cast<float*>(p.scores) + (int64_t)bh * p.scores_stride + s0;
  *reinterpret_cast<f32x4*>(out) = f32x4{res[0], res[1], res[2], res[3]};     // stride % 4 == 0, s0 % 4 == 0, stride >= roundup(L, 8)
}

// ------------------------------------------------------------------------------------------------
// topk_f32_kernel: one workgroup (1024 threads) per row.
// ------------------------------------------------------------------------------------------------
constexpr int TKF_MAX = 4096;

__device__ __forceinline__ uint32_t f32_key(float x) {   // ascending unsigned order == ascending float order; -0 == +0
  if (x == 0.f) x = 0.f;
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// REG: rows of up to 32768 scores keep their keys in registers (32 per thread, loaded once with all loads in flight);
// longer rows re-read the scores from memory in every pass.
constexpr int TKF_NV = 32;

template <bool REG>
__global__ __launch_bounds__(1024) void topk_f32_kernel(TopkParams p) {
  constexpr int UNR = REG ? TKF_NV : 1;       // full unroll keeps the keys in registers; the streaming form stays a loop
  __shared__ unsigned long long comp[TKF_MAX];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_digit, sh_kk, sh_cnt, wave_cnt[16], sh_taken;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x, L = p.L;
  int k = p.k_per_row ? min(p.k_per_row[row], p.k) : p.k;      // a device-side capacity never exceeds the validated k (LDS list, output row)
  if (k > L) k = L;
  if (k <= 0) return;
  const float* sc = reinterpret_cast<const float*>(p.scores) + (int64_t)row * p.scores_stride;
  uint32_t ureg[REG ? TKF_NV : 1];
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < TKF_NV; ++j) {
      const int i = j * 1024 + tid;
      ureg[j] = f32_key(sc[i < L ? i : L - 1]);
    }
  }
  const int nv = REG ? TKF_NV : (L + 1023) / 1024;
  auto key_at = [&](int j, int i) -> uint32_t {
    if constexpr (REG) return ureg[j];
    else return i < L ? f32_key(sc[i]) : 0u;
  };
  uint32_t prefix = 0, mask = 0, kk = (uint32_t)k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll UNR
    for (int j = 0; j < nv; ++j) {
      const int i = j * 1024 + tid;
      const uint32_t u = key_at(j, i);
      bool act = i < L && (u & mask) == prefix;
      const uint32_t dg = (u >> shift) & 255u;
      // scores of one row share their sign and most exponent bits: the top digits put nearly every lane of a wave into
      // ONE bin, and same-address LDS atomics serialise per lane.  One aggregated round for the first active lane's
      // digit (one atomic for the whole group), plain atomics for whoever is left.  (More rounds, or none for the lower
      // digits, measured slower.)
      const unsigned long long m = __ballot(act);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t dl = (uint32_t)__shfl((int)dg, leader, 64);
        const unsigned long long same = __ballot(act && dg == dl);
        if (lane == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
        act = act && dg != dl;
      }
      if (act) atomicAdd(&hist[dg], 1u);
    }
    __syncthreads();
    if (wave == 0) {                                   // digits from the top: lane l holds 255-4l .. 252-4l
      uint32_t c[4], s = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c[j] = hist[255 - (4 * lane + j)]; s += c[j]; }
      const uint32_t incl = wave_incl_scan_u32(s), excl = incl - s;
      if (excl < kk && kk <= incl) {
        uint32_t run = excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (run < kk && kk <= run + c[j]) { sh_digit = 255 - (4 * lane + j); sh_kk = kk - run; }
          run += c[j];
        }
      }
    }
    __syncthreads();
    prefix |= sh_digit << shift;
    mask |= 0xffu << shift;
    kk = sh_kk;
    __syncthreads();
  }
  // prefix = key of the k-th largest score; kk of the scores equal to it are taken, the earliest positions first
  const uint32_t T = prefix, need = kk;
  if (tid == 0) { sh_cnt = 0; sh_taken = 0; }
  __syncthreads();
#pragma unroll UNR
  for (int j = 0; j < nv; ++j) {
    const int i = j * 1024 + tid;
    const uint32_t u = key_at(j, i);
    if (i < L && u > T) {
      const uint32_t pos = atomicAdd(&sh_cnt, 1u);
      comp[pos] = ((unsigned long long)u << 32) | (0xffffffffu - (uint32_t)i);
    }
  }
  __syncthreads();
  const uint32_t n_gt = (uint32_t)k - need;            // == sh_cnt
#pragma unroll UNR
  for (int j = 0; j < nv; ++j) {                       // the scores equal to the threshold, in position order
    const uint32_t taken = sh_taken;
    if (taken >= need) break;
    const int i = j * 1024 + tid;
    const bool eq = i < L && key_at(j, i) == T;
    const unsigned long long bal = __ballot(eq);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int wv = 0; wv < 16; ++wv) { const uint32_t cw = wave_cnt[wv]; if (wv < wave) before += cw; total += cw; }
    const uint32_t rank = taken + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (eq && rank < need) comp[n_gt + rank] = ((unsigned long long)T << 32) | (0xffffffffu - (uint32_t)i);
    __syncthreads();
    if (tid == 0) sh_taken = taken + total;
    __syncthreads();
  }
  // descending bitonic sort of the k winners: (value desc, index asc); padding 0 is below every real entry
  int n = 1;
  while (n < k) n <<= 1;
  for (int i = k + tid; i < n; i += 1024) comp[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += 1024) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = comp[lo], b2 = comp[hi];
        if ((a < b2) == desc) { comp[lo] = b2; comp[hi] = a; }
      }
      __syncthreads();
    }
  }
  int32_t* out = p.idx_out + (int64_t)row * p.idx_stride;
  for (int j = tid; j < k; j += 1024) out[j] = (int32_t)(0xffffffffu - (uint32_t)comp[j]);
}

// ------------------------------------------------------------------------------------------------
// Ada-SnapKV budgets from fp32 score rows (reference pyramidkv_utils.py:706-719 on fp32 tensors; round 3).
// The scheme of pkv_ada.hip's un-sorted kernels with 32-bit keys: four 8-bit radix levels instead of two, one launch per
// level (the global digit of a level is a sum over all heads), one 1024-thread workgroup per head, the row in registers
// (<= 32768 scores).  Level 0 also makes the head's ratio (:710): exact radix select of the base-th largest raw score, the
// two sums in double, rounded to fp32, divided in fp32.  ATen sums fp32 tensors in an order its vector width decides, so
// there is no bit-level target for that quotient: against the CPU oracle the budgets agree up to the rare entry that sits
// within an ulp of the global threshold (tests/test_gpu_f32.py states the bar).
//   cum[level][h][d]  = number of adaptive keys of head h that are >= ((prefix_level << 8 | d) << shift_level)
//   above[level][h]   = number of adaptive keys of head h above every key that shares prefix_level
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float f32_key_value(uint32_t key) {           // inverse of f32_key
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

__global__ __launch_bounds__(1024) void ada_f32_level_kernel(BudgetParams p, float* ratio_ws, int32_t* cum, int32_t* above, int level) {
  __shared__ uint32_t hist[256];
  __shared__ int64_t s_sum[256];
  __shared__ uint32_t sh_digit, sh_kk;
  __shared__ int s_b;
  __shared__ double red[2][16];
  __shared__ uint32_t wsum[16];
  __shared__ float s_ratio;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, L = p.Lrow, H = p.H;
  const float* row = reinterpret_cast<const float*>(p.scores) + (int64_t)h * p.scores_stride;
  float sv[TKF_NV];
#pragma unroll
  for (int j = 0; j < TKF_NV; ++j) {
    const int i = j * 1024 + tid;
    sv[j] = row[i < L ? i : L - 1];
  }
  // one radix-histogram round over this thread's keys: digit `shift` of the keys for which take(j) holds.  Scores of one row
  // share sign and most exponent bits: one aggregated LDS atomic for the first active lane's digit, plain atomics for the rest
  auto hist_round = [&](auto&& key_of, auto&& take, int shift) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TKF_NV; ++j) {
      const int i = j * 1024 + tid;
      bool act = i < L && take(j);
      const uint32_t dg = (key_of(j) >> shift) & 255u;
      const unsigned long long m = __ballot(act);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t dl = (uint32_t)__shfl((int)dg, leader, 64);
        const unsigned long long same = __ballot(act && dg == dl);
        if (lane == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
        act = act && dg != dl;
      }
      if (act) atomicAdd(&hist[dg], 1u);
    }
    __syncthreads();
  };
  float ratio = 1.0f;
  if (level == 0) {
    if (p.normalize) {
      // base-th largest raw score (32-bit key, four rounds), then the two sums of :710
      uint32_t prefix = 0, mask = 0, kk = (uint32_t)p.base;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist_round([&](int j) { return f32_key(sv[j]); }, [&](int j) { return (f32_key(sv[j]) & mask) == prefix; }, shift);
        if (wave == 0) {                                   // digits from the top: lane l holds 255-4l .. 252-4l
          uint32_t c[4], sm = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) { c[j] = hist[255 - (4 * lane + j)]; sm += c[j]; }
          const uint32_t incl = wave_incl_scan_u32(sm), excl = incl - sm;
          if (excl < kk && kk <= incl) {
            uint32_t run = excl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (run < kk && kk <= run + c[j]) { sh_digit = 255 - (4 * lane + j); sh_kk = kk - run; }
              run += c[j];
            }
          }
        }
        __syncthreads();
        prefix |= sh_digit << shift;
        mask |= 0xffu << shift;
        kk = sh_kk;
        __syncthreads();
      }
      const uint32_t T = prefix;                           // kk entries equal to T belong to the top `base`
      double st = 0.0, sa = 0.0;
#pragma unroll
      for (int j = 0; j < TKF_NV; ++j) {
        const int i = j * 1024 + tid;
        if (i < L) {
          const double x = (double)sv[j];
          sa += x;
          if (f32_key(sv[j]) > T) st += x;
        }
      }
      for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o, 64); sa += __shfl_xor(sa, o, 64); }
      if (lane == 0) { red[0][wave] = st; red[1][wave] = sa; }
      __syncthreads();
      if (tid == 0) {
        double t = 0.0, a = 0.0;
        for (int w2 = 0; w2 < 16; ++w2) { t += red[0][w2]; a += red[1][w2]; }
        t += (double)kk * (double)f32_key_value(T);
        s_ratio = (float)t / (float)a;                     // fp32 sums, fp32 division (:710)
        ratio_ws[h] = s_ratio;
      }
      __syncthreads();
      ratio = s_ratio;
    } else if (tid == 0) {
      ratio_ws[h] = 1.0f;
    }
  } else {
    ratio = ratio_ws[h];
  }
  const int norm = p.normalize;
  auto akey = [&](int j) { return f32_key(norm ? sv[j] * ratio : sv[j]); };      // adaptive_attn_score * ratio_weight (:711)
  // the digits the earlier levels settled: largest d with sum_h cum[lv][h][d] >= H * base
  const int64_t total = (int64_t)H * p.base;
  uint32_t gprefix = 0;
  for (int lv = 0; lv < level; ++lv) {
    if (tid < 256) {
      const int32_t* c = cum + (int64_t)lv * H * 256;
      int64_t sm = 0;
      for (int h0 = 0; h0 < H; h0 += 32) {
        int32_t v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = c[(h0 + j < H ? h0 + j : H - 1) * 256 + tid];
#pragma unroll
        for (int j = 0; j < 32; ++j) sm += (h0 + j < H) ? v[j] : 0;
      }
      s_sum[tid] = sm;
    }
    __syncthreads();
    if (tid < 256 && s_sum[tid] >= total && (tid == 255 || s_sum[tid + 1] < total)) s_b = tid;
    __syncthreads();
    gprefix = (gprefix << 8) | (uint32_t)s_b;
    __syncthreads();
  }
  const int shift = 24 - 8 * level;
  hist_round(akey, [&](int j) { return level == 0 || (akey(j) >> (shift + 8)) == gprefix; }, shift);
  uint32_t ab = 0;                                          // keys above everything that shares the prefix
  if (level > 0) {
#pragma unroll
    for (int j = 0; j < TKF_NV; ++j) {
      const int i = j * 1024 + tid;
      ab += (i < L && (akey(j) >> (shift + 8)) > gprefix) ? 1u : 0u;
    }
    ab = wave_sum_u32(ab);
    if (lane == 0) wsum[wave] = ab;
  }
  {                                                         // suffix sums of the 256 digit counts (4 waves x 64 digits)
    const uint32_t c = tid < 256 ? hist[tid] : 0u;
    const uint32_t incl = wave_incl_scan_u32(c);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t suf = wtot - incl + c;
    __syncthreads();                                        // hist reads done; wsum complete
    if (lane == 0 && wave < 4) hist[wave] = wtot;
    uint32_t abv = 0;
    if (level > 0) for (int w2 = 0; w2 < 16; ++w2) abv += wsum[w2];
    __syncthreads();
    if (tid < 256) {
      for (int w2 = wave + 1; w2 < 4; ++w2) suf += hist[w2];
      cum[((int64_t)level * H + h) * 256 + tid] = (int32_t)(suf + abv);
    }
    if (tid == 0) above[level * H + h] = (int32_t)abv;
  }
}

int budget_f32_max_row() { return TKF_NV * 1024; }

hipError_t launch_budget_f32(const BudgetParams& p, hipStream_t st) {
  char* base = reinterpret_cast<char*>(p.ws);
  float* ratio = reinterpret_cast<float*>(base);
  int32_t* cum = reinterpret_cast<int32_t*>(base + 1024);
  int32_t* above = cum + (size_t)4 * p.H * 256;
  for (int level = 0; level < 4; ++level)
    hipLaunchKernelGGL(ada_f32_level_kernel, dim3(p.H), dim3(1024), 0, st, p, ratio, cum, above, level);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_ada_final(p, cum + (size_t)2 * p.H * 256, cum + (size_t)3 * p.H * 256, above + 2 * p.H, st);
}

hipError_t launch_logits_f32(const LogitsParams& p, hipStream_t st) {
  dim3 grid(p.nT, p.B * (p.H / p.G));
  if (p.D == 64) PKV_KLAUNCH(logits_f32_kernel<4>, grid, dim3(256), 0, st, p);
  else if (p.D == 256) PKV_KLAUNCH(logits_f32_kernel<16>, grid, dim3(256), 0, st, p);     // round 5: the reference takes any head size in fp32 too
  else PKV_KLAUNCH(logits_f32_kernel<8>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_finalize_f32(const FinalizeParams& p, hipStream_t st) {
  const int L = p.S - p.w;
  dim3 grid((L + F32_OUT - 1) / F32_OUT, p.B * p.H);
  PKV_KLAUNCH(finalize_f32_kernel, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

int topk_f32_max_k() { return TKF_MAX; }

hipError_t launch_topk_f32(int rows, const TopkParams& p, hipStream_t st) {
  if (p.L <= TKF_NV * 1024) PKV_KLAUNCH(topk_f32_kernel<true>, dim3(rows), dim3(1024), 0, st, p);
  else PKV_KLAUNCH(topk_f32_kernel<false>, dim3(rows), dim3(1024), 0, st, p);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// H2O on fp32 tensors (reference pyramidkv_utils.py:544-554 is dtype-generic; round 4).  The same two passes as the 16-bit
// kernels (pkv_h2o.hip) without any rounding to a model dtype, on v_mfma_f32_16x16x4_f32 (1/16 of the bf16 matrix rate:
// built for completeness, S x S x D fp32 flops are what they are):
//   h2o_stats_f32_kernel   per query row: (max, 1 / sum exp) over all keys; one wave = 16 resident query rows, keys streamed
//   h2o_colsum_f32_kernel  per key column: sum over all query rows of exp(x - max) * (1 / sum); one wave = 16 resident keys
// Operand layout as in logits_f32_kernel: lane (l % 16 = row / column, l / 16 = which float4 of every 16 elements).
// The fp32 sums (dot product, softmax denominator, column sum) are not pinned to an order by the reference: tests compare
// the scores with a relative tolerance, selections up to score ties within it.
// ------------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void load_row_f32(f32x4 (&f)[KS], const float* base, int64_t row, int64_t stride, int lg) {
  const float* r = base + row * stride + 4 * lg;
#pragma unroll
  for (int j = 0; j < KS; ++j) f[j] = *reinterpret_cast<const f32x4*>(r + 16 * j);
}
template <int KS>
__device__ __forceinline__ f32x4 dot16_f32(const f32x4 (&a)[KS], const f32x4 (&b)[KS]) {     // D[a row][b column]
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], b[j][i], acc, 0, 0, 0);
  return acc;
}

template <int KS>
__global__ __launch_bounds__(256) void h2o_stats_f32_kernel(H2OParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const float* qb = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const float* kb = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int qi = blockIdx.x * 64 + wave * 16 + li;                         // this lane's query (column of D)
  f32x4 qf[KS];
  load_row_f32<KS>(qf, qb, qi < S ? qi : S - 1, p.qs_s, lg);
  const float fmin_v = -3.4028234663852886e38f;                            // torch.finfo(torch.float32).min
  float m = -INFINITY, z = 0.f;
  for (int s0 = 0; s0 < S; s0 += 16) {
    f32x4 kf[KS];
    load_row_f32<KS>(kf, kb, s0 + li < S ? s0 + li : S - 1, p.ks_s, lg);
    const f32x4 acc = dot16_f32<KS>(kf, qf);                               // D[key s0 + 4*lg + r][query qi]
    float x[4], mx = m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = s0 + 4 * lg + r;
      float v = p.scale_mode == 0 ? div_const(acc[r], p.sqrt_d, p.rcp_sqrt_d) : acc[r] * p.rcp_sqrt_d;   // / math.sqrt(head_dim) (:544)
      if (qi >= L && s >= L && (s - L) > (qi - L)) v = v + fmin_v;        // the last w x w corner (:545-551)
      x[r] = s < S ? v : -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    if (mx > m) { z = (m == -INFINITY) ? 0.f : z * pkv_exp(m - mx); m = mx; }
    if (m != -INFINITY) {
#pragma unroll
      for (int r = 0; r < 4; ++r) z += pkv_exp(x[r] - m);
    }
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {                                     // the four key groups of the lane's query
    const float mo = __shfl_xor(m, o, 64), zo = __shfl_xor(z, o, 64);
    const float M = fmaxf(m, mo), Ms = (M == -INFINITY) ? 0.f : M;
    z = z * pkv_exp(m - Ms) + zo * pkv_exp(mo - Ms);
    m = M;
  }
  if (lg == 0 && qi < S) reinterpret_cast<float2*>(p.rowstat)[(int64_t)bh * S + qi] = make_float2(m, 1.0f / z);   // ATen CPU softmax: exp(x - max) * (1 / sum)
}

template <int KS>
__global__ __launch_bounds__(256) void h2o_colsum_f32_kernel(H2OParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const float* qb = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const float* kb = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float2* rs = reinterpret_cast<const float2*>(p.rowstat) + (int64_t)bh * S;
  const int kj = blockIdx.x * 64 + wave * 16 + li;                         // this lane's key (column of D); columns >= L are not stored
  f32x4 kf[KS];
  load_row_f32<KS>(kf, kb, kj < S ? kj : S - 1, p.ks_s, lg);
  float sum = 0.f;
  for (int i0 = 0; i0 < S; i0 += 16) {
    f32x4 qf[KS];
    load_row_f32<KS>(qf, qb, i0 + li < S ? i0 + li : S - 1, p.qs_s, lg);
    const f32x4 acc = dot16_f32<KS>(qf, kf);                               // D[query i0 + 4*lg + r][key kj]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * lg + r;
      const float2 st = rs[i < S ? i : S - 1];
      const float v = p.scale_mode == 0 ? div_const(acc[r], p.sqrt_d, p.rcp_sqrt_d) : acc[r] * p.rcp_sqrt_d;
      const float pr = pkv_exp(v - st.x) * st.y;                           // fp32 softmax (:553); keys < L never touch the masked corner
      sum += i < S ? pr : 0.f;                                             // fp32 column sum (:554)
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  if (lg == 0 && kj < L) reinterpret_cast<float*>(p.scores)[(int64_t)bh * p.scores_stride + kj] = sum;
}

hipError_t launch_h2o_f32(const H2OParams& p, hipStream_t st) {
  dim3 g1((p.S + 63) / 64, p.B * p.H), g2((p.S - p.w + 63) / 64, p.B * p.H);
  if (p.D == 64) {
    hipLaunchKernelGGL(h2o_stats_f32_kernel<4>, g1, dim3(256), 0, st, p);
    hipLaunchKernelGGL(h2o_colsum_f32_kernel<4>, g2, dim3(256), 0, st, p);
  } else if (p.D == 256) {
    hipLaunchKernelGGL(h2o_stats_f32_kernel<16>, g1, dim3(256), 0, st, p);
    hipLaunchKernelGGL(h2o_colsum_f32_kernel<16>, g2, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(h2o_stats_f32_kernel<8>, g1, dim3(256), 0, st, p);
    hipLaunchKernelGGL(h2o_colsum_f32_kernel<8>, g2, dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

}  // namespace pkv
