// pkv_h2o.hip — H2O score kernels (gfx950): attention mass each key receives from ALL S query rows.
//
//   reference pyramidkv_utils.py:544-554:
//     A = (Q K^T)/sqrt(D) [S x S], only the last w x w corner causally masked (early rows DO see
//     future keys - a reference quirk that is reproduced), P = softmax_fp32(A).to(dtype),
//     score[j] = sum_i P[i][j] for j < S-w (fp32 accumulate, one rounding).
//
// S x S is never materialised (68.7 GB bf16 at S=32k).  Two MFMA passes over the S x S tile space:
//   h2o_stats_kernel   per query row: max and sum-of-exp over all keys (online softmax statistics)
//   h2o_colsum_kernel  per key column: sum over all query rows of round(exp(x-m)/Z)
// Both recompute the logits with mfma_f32_16x16x32 and apply the reference's three roundings.
// exp: every probability is ONE v_exp_f32 of fma(x, log2e, c_row) with c_row = -(m*log2e + log2 Z) (the
// 1/Z factor folded into the exponent), relative error ~|x*log2e| * 2^-24 (about 1e-6).  That is ~20x the
// error of the window path's exp, and deliberately so: a score here is a sum over S >= 1000s of rounded
// probabilities, so the rare rounding flips (1e-6 / 2^-9 per element) average out far below the model-dtype
// resolution of the sum (measured against the oracle in tests/test_gpu_parity.py::test_h2o_*), while the
// kernels are VALU-bound and the accurate exp costs 8 of ~18 vector instructions per S x S element.
// Roofline: compute.  2*2*S^2*D*H flops per call on the matrix cores, plus ~60 VALU ops per S x S
// element for the rounding chain / exp / division, which is what actually bounds it.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8_t;

template <typename T> struct Mfma2;
template <> struct Mfma2<BF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma2<F16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ float logit_chain(float acc, const H2OParams& p) {
  float x = Elem<T>::to_f32(Elem<T>::from_f32(acc));                       // matmul output dtype (:544)
  x = scale_logit<T>(x, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);             // / math.sqrt(head_dim)
  return Elem<T>::to_f32(Elem<T>::from_f32(x));
}

// ------------------------------------------------------------------------------------------------
// Tiling shared by both passes.  A workgroup (4 waves) keeps 256 "resident" rows in registers as MFMA
// B operands (64 per wave = 4 column tiles x 4 k-steps x 16 B per lane) and streams the other matrix
// in 64-row tiles through LDS, where all four waves read it (one L2 read per workgroup instead of one
// per wave: without this the kernels are L2-bandwidth bound).  Staging is global -> VGPR -> ds_write,
// double buffered, the next tile's global loads in flight while the current one is consumed.
// LDS tile = [64 rows][16 chunks of 16 B]; chunk c of row r is stored at chunk c ^ (r & 15), so the
// A-fragment read (lane (li, lg) reads row li, chunk kk*4+lg) is bank-conflict free.
// ------------------------------------------------------------------------------------------------
constexpr int HT = 64;                 // streamed rows per LDS tile
constexpr int HR = 64;                 // resident rows per wave
constexpr int HWG = 4 * HR;            // resident rows per workgroup

struct Stager {                        // one thread's share of a 64 x 256 B tile: 4 x 16 B
  u32x4 v[4];
};

__device__ __forceinline__ void stage_load(Stager& st, const uint16_t* base, int64_t stride, int row0, int nrows, int tid) {
  const int c = tid & 15, r0 = tid >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = row0 + r0 + 16 * i;
    r = r < nrows ? r : nrows - 1;                                            // clamp: masked by the consumer
    st.v[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)r * stride + c * 8);
  }
}
__device__ __forceinline__ void stage_store(const Stager& st, u32x4* tile, int tid) {
  const int c = tid & 15, r0 = tid >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 16 * i;
    tile[r * 16 + (c ^ (r & 15))] = st.v[i];
  }
}
__device__ __forceinline__ void read_frags(u32x4 (&f)[4], const u32x4* tile, int sub, int li, int lg) {
  const int r = sub * 16 + li;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = tile[r * 16 + ((kk * 4 + lg) ^ li)];
}
__device__ __forceinline__ void load_frags(u32x4 (&f)[4], const uint16_t* base, int64_t row, int64_t stride, int lg) {
  const uint16_t* r = base + row * stride + lg * 8;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const u32x4*>(r + kk * 32);
}

// Pass 1: per query row, max and sum of exp over all keys.  Resident = 256 query rows, streamed = K.
// Per-lane online statistics (lane's column = one query, 4 keys per 16-key subtile); the running maximum
// is only rescaled when some lane of the wave actually found a larger logit (wave-uniform branch).
template <typename T>
__global__ __launch_bounds__(256) void h2o_stats_kernel(H2OParams p) {
  __shared__ __attribute__((aligned(16))) u32x4 tiles[2][HT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int q0 = blockIdx.x * HWG + wave * HR;
  const float fmin_v = Elem<T>::finfo_min();
  const bool corner_wave = q0 + HR > L;          // this wave holds observation-window rows (wave-uniform)

  u32x4 qf[4][4];
  int qi[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    qi[n] = q0 + n * 16 + li;
    load_frags(qf[n], qb, qi[n] < S ? qi[n] : S - 1, p.qs_s, lg);
  }
  const float L2E = 1.44269504088896340736f;
  float m[4], mL[4], Z[4];          // running max, -max*log2e, running sum of exp
#pragma unroll
  for (int n = 0; n < 4; ++n) { m[n] = -INFINITY; mL[n] = 0.f; Z[n] = 0.f; }

  Stager stg;
  stage_load(stg, kb, p.ks_s, 0, S, tid);
  stage_store(stg, tiles[0], tid);
  __syncthreads();
  const int ntiles = (S + HT - 1) / HT;
  for (int t = 0; t < ntiles; ++t) {
    const int s_tile = t * HT;
    const u32x4* cur = tiles[t & 1];
    if (t + 1 < ntiles) stage_load(stg, kb, p.ks_s, s_tile + HT, S, tid);     // in flight during the compute below
    const bool edge = (s_tile + HT > S) || (corner_wave && s_tile + HT > L);  // wave-uniform: tail / masked corner
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      u32x4 kf[4];
      read_frags(kf, cur, sub, li, lg);
      const int s0 = s_tile + sub * 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma2<T>::run(kf[kk], qf[n][kk], acc);
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = logit_chain<T>(acc[r], p);
        if (edge) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s = s0 + lg * 4 + r;
            if (qi[n] >= L && s >= L && (s - L) > (qi[n] - L))                // corner mask (:545-551)
              x[r] = Elem<T>::to_f32(Elem<T>::from_f32(x[r] + fmin_v));
            if (s >= S) x[r] = -INFINITY;
          }
        }
        const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        if (__any(mx > m[n])) {                                               // rare once the maxima settle
          const float mn = fmaxf(m[n], mx);
          Z[n] = (m[n] == -INFINITY) ? 0.f : Z[n] * __builtin_amdgcn_exp2f((m[n] - mn) * L2E);
          m[n] = mn;
          mL[n] = (mn == -INFINITY) ? 0.f : -mn * L2E;
        }
        const float c = mL[n];
        Z[n] += (__builtin_amdgcn_exp2f(fmaf(x[0], L2E, c)) + __builtin_amdgcn_exp2f(fmaf(x[1], L2E, c))) +
                (__builtin_amdgcn_exp2f(fmaf(x[2], L2E, c)) + __builtin_amdgcn_exp2f(fmaf(x[3], L2E, c)));
      }
    }
    if (t + 1 < ntiles) stage_store(stg, tiles[(t + 1) & 1], tid);            // buffer last read in iteration t-1
    __syncthreads();
  }
  float2* rs = p.rowstat + (int64_t)bh * S;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float mm = m[n], zz = Z[n];
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float mo = __shfl_xor(mm, o, 64), zo = __shfl_xor(zz, o, 64);
      const float mn = fmaxf(mm, mo);
      const float za = (mm == -INFINITY) ? 0.f : zz * __builtin_amdgcn_exp2f((mm - mn) * L2E);
      const float zb = (mo == -INFINITY) ? 0.f : zo * __builtin_amdgcn_exp2f((mo - mn) * L2E);
      mm = mn; zz = za + zb;
    }
    // c_row = -(m*log2e + log2 Z): pass 2 evaluates exp(x - m) / Z as exp2(x*log2e + c_row)
    if (lg == 0 && qi[n] < S) rs[qi[n]] = make_float2(-(mm * L2E + __builtin_amdgcn_logf(zz)), 0.f);
  }
}

// Pass 2: per key column, sum over all query rows of round(exp(x - m) / Z).  Resident = 256 key columns,
// streamed = Q (+ the 64 row statistics of the tile).
template <typename T>
__global__ __launch_bounds__(256) void h2o_colsum_kernel(H2OParams p) {
  __shared__ __attribute__((aligned(16))) u32x4 tiles[2][HT * 16];
  __shared__ float2 stats[2][HT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float2* rs = p.rowstat + (int64_t)bh * S;
  const int k0 = blockIdx.x * HWG + wave * HR;

  u32x4 kf[4][4];
  int kj[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    kj[n] = k0 + n * 16 + li;
    load_frags(kf[n], kb, kj[n] < S ? kj[n] : S - 1, p.ks_s, lg);
  }
  float col[4] = {0.f, 0.f, 0.f, 0.f};
  const float L2E2 = 1.44269504088896340736f;

  Stager stg;
  float2 sreg = make_float2(0.f, 0.f);
  stage_load(stg, qb, p.qs_s, 0, S, tid);
  if (tid < HT) sreg = rs[tid < S ? tid : S - 1];
  stage_store(stg, tiles[0], tid);
  if (tid < HT) stats[0][tid] = sreg;
  __syncthreads();
  const int ntiles = (S + HT - 1) / HT;
  for (int t = 0; t < ntiles; ++t) {
    const int i_tile = t * HT;
    const u32x4* cur = tiles[t & 1];
    const float2* cst = stats[t & 1];
    if (t + 1 < ntiles) {
      stage_load(stg, qb, p.qs_s, i_tile + HT, S, tid);
      if (tid < HT) { const int i = i_tile + HT + tid; sreg = rs[i < S ? i : S - 1]; }
    }
    const bool tail = i_tile + HT > S;                                        // wave-uniform
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      u32x4 qf[4];
      read_frags(qf, cur, sub, li, lg);
      float2 st[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) st[r] = cst[sub * 16 + lg * 4 + r];
      const int i0 = i_tile + sub * 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = Mfma2<T>::run(qf[kk], kf[n][kk], acc);   // D[query][key]
        float pq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = logit_chain<T>(acc[r], p);           // keys < L never touch the masked corner
          const float pr = __builtin_amdgcn_exp2f(fmaf(x, L2E2, st[r].x));   // fp32 softmax (:553): exp(x - m) / Z
          pq[r] = Elem<T>::to_f32(Elem<T>::from_f32(pr));
        }
        if (tail) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (i0 + lg * 4 + r >= S) pq[r] = 0.f;
        }
        col[n] += (pq[0] + pq[1]) + (pq[2] + pq[3]);           // sum over all rows, fp32 (:554)
      }
    }
    if (t + 1 < ntiles) {
      stage_store(stg, tiles[(t + 1) & 1], tid);
      if (tid < HT) stats[(t + 1) & 1][tid] = sreg;
    }
    __syncthreads();
  }
  uint16_t* out = reinterpret_cast<uint16_t*>(p.scores) + (int64_t)bh * p.scores_stride;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float c = col[n];
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    if (lg == 0 && kj[n] < L) out[kj[n]] = Elem<T>::from_f32(c);
  }
}

template __global__ void h2o_stats_kernel<BF16>(H2OParams);
template __global__ void h2o_stats_kernel<F16>(H2OParams);
template __global__ void h2o_colsum_kernel<BF16>(H2OParams);
template __global__ void h2o_colsum_kernel<F16>(H2OParams);

hipError_t launch_h2o_stats(int dtype, const H2OParams& p, hipStream_t st) {
  dim3 grid((p.S + HWG - 1) / HWG, p.B * p.H);
  if (dtype == 0) hipLaunchKernelGGL(h2o_stats_kernel<BF16>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(h2o_stats_kernel<F16>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_h2o_colsum(int dtype, const H2OParams& p, hipStream_t st) {
  const int L = p.S - p.w;
  dim3 grid((L + HWG - 1) / HWG, p.B * p.H);
  if (dtype == 0) hipLaunchKernelGGL(h2o_colsum_kernel<BF16>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(h2o_colsum_kernel<F16>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

}  // namespace pkv
