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
// kernels are instruction-issue-bound and the accurate exp costs 8 vector instructions per S x S element.
// Roofline: compute.  2*2*S^2*D*H flops per call on the matrix cores; what actually bounds it is the SIMD issue port,
// shared by the MFMAs and the ~10 vector instructions per S x S element of the rounding chain / exp (a quarter-rate
// transcendental) / accumulate: ~190 issue cycles per 16x16 output block against 64 cycles of matrix-pipe time.
// Issuing the next subtile's MFMAs ahead of the current epilogue (software pipelining) changed nothing (12.45 ->
// 12.26 ms at 192 VGPRs): the port, not the MFMA latency, is the limit.
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

// Four logits of one MFMA accumulator through the reference's two roundings.  bf16: the values travel as packed
// pairs - one v_cvt_pk_bf16_f32 per two roundings, one shift / mask per unpack, the scale as a packed fp32
// multiply (these kernels are VALU-bound: 13 -> ~10 vector-instruction equivalents per S x S element).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void logits4(const f32x4& acc, const H2OParams& p, float (&x)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = logit_chain<T>(acc[r], p);
}
template <>
__device__ __forceinline__ void logits4<BF16>(const f32x4& acc, const H2OParams& p, float (&x)[4]) {
  uint32_t p01 = round_pack2<BF16>(acc[0], acc[1]);                        // matmul output dtype (:544)
  uint32_t p23 = round_pack2<BF16>(acc[2], acc[3]);
  f32x2 a = {__uint_as_float(p01 << 16), __uint_as_float(p01 & 0xffff0000u)};
  f32x2 b = {__uint_as_float(p23 << 16), __uint_as_float(p23 & 0xffff0000u)};
  const f32x2 rc = {p.rcp_sqrt_d, p.rcp_sqrt_d};                           // / math.sqrt(head_dim): exact for bf16, see scale_logit
  a = a * rc;
  b = b * rc;
  p01 = round_pack2<BF16>(a.x, a.y);
  p23 = round_pack2<BF16>(b.x, b.y);
  x[0] = __uint_as_float(p01 << 16); x[1] = __uint_as_float(p01 & 0xffff0000u);
  x[2] = __uint_as_float(p23 << 16); x[3] = __uint_as_float(p23 & 0xffff0000u);
}
// round four fp32 probabilities to the model dtype and return their fp32 sum ((p0 + p1) + (p2 + p3))
template <typename T>
__device__ __forceinline__ float round_sum4(const float (&e)[4]) {
  float pq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) pq[r] = Elem<T>::to_f32(Elem<T>::from_f32(e[r]));
  return (pq[0] + pq[1]) + (pq[2] + pq[3]);
}
template <>
__device__ __forceinline__ float round_sum4<BF16>(const float (&e)[4]) {
  const uint32_t p01 = round_pack2<BF16>(e[0], e[1]);
  const uint32_t p23 = round_pack2<BF16>(e[2], e[3]);
  const f32x2 lo = {__uint_as_float(p01 << 16), __uint_as_float(p23 << 16)};
  const f32x2 hi = {__uint_as_float(p01 & 0xffff0000u), __uint_as_float(p23 & 0xffff0000u)};
  const f32x2 s = lo + hi;                                                 // (p0 + p1), (p2 + p3)
  return s.x + s.y;
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
        logits4<T>(acc, p, x);
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
        const f32x2 c2 = {mL[n], mL[n]}, l2 = {L2E, L2E};
        const f32x2 y01 = __builtin_elementwise_fma(f32x2{x[0], x[1]}, l2, c2);
        const f32x2 y23 = __builtin_elementwise_fma(f32x2{x[2], x[3]}, l2, c2);
        Z[n] += (__builtin_amdgcn_exp2f(y01.x) + __builtin_amdgcn_exp2f(y01.y)) +
                (__builtin_amdgcn_exp2f(y23.x) + __builtin_amdgcn_exp2f(y23.y));
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
        float x[4], e[4];
        logits4<T>(acc, p, x);                                 // keys < L never touch the masked corner
        const f32x2 l2 = {L2E2, L2E2};
        const f32x2 y01 = __builtin_elementwise_fma(f32x2{x[0], x[1]}, l2, f32x2{st[0].x, st[1].x});
        const f32x2 y23 = __builtin_elementwise_fma(f32x2{x[2], x[3]}, l2, f32x2{st[2].x, st[3].x});
        e[0] = __builtin_amdgcn_exp2f(y01.x); e[1] = __builtin_amdgcn_exp2f(y01.y);   // fp32 softmax (:553): exp(x - m) / Z
        e[2] = __builtin_amdgcn_exp2f(y23.x); e[3] = __builtin_amdgcn_exp2f(y23.y);
        if (tail) {                                            // wave-uniform, last tile only: rows past S contribute 0
#pragma unroll
          for (int r = 0; r < 4; ++r) if (i0 + lg * 4 + r >= S) e[r] = 0.f;
        }
        col[n] += round_sum4<T>(e);                            // .to(dtype), sum over all rows in fp32 (:554)
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
