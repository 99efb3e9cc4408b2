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
  static __device__ __forceinline__ f32x4 run(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma2<F16> {
  static __device__ __forceinline__ f32x4 run(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ float logit_chain(float acc, const H2OParams& p) {
  float x = Elem<T>::to_f32(Elem<T>::from_f32(acc));                       // matmul output dtype (:544)
  x = (p.scale_mode == 0) ? div_const(x, p.sqrt_d, p.rcp_sqrt_d) : (x * p.rcp_sqrt_d);   // / math.sqrt(head_dim), exact
  return Elem<T>::to_f32(Elem<T>::from_f32(x));
}

__device__ __forceinline__ void load_frags(uint4 (&f)[4], const uint16_t* base, int64_t row, int64_t stride, int lg) {
  const uint16_t* r = base + row * stride + lg * 8;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const uint4*>(r + kk * 32);
}

// one workgroup = 128 query rows (4 waves x 32); loop over all S keys
template <typename T>
__global__ __launch_bounds__(256) void h2o_stats_kernel(H2OParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float fmin_v = Elem<T>::finfo_min();

  uint4 qf[2][4];
  int qi[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    qi[n] = q0 + n * 16 + li;
    load_frags(qf[n], qb, qi[n] < S ? qi[n] : S - 1, p.qs_s, lg);
  }
  float m[2] = {-INFINITY, -INFINITY}, Z[2] = {0.f, 0.f};

  for (int s0 = 0; s0 < S; s0 += 16) {
    uint4 kf[4];
    const int sr = s0 + li;
    load_frags(kf, kb, sr < S ? sr : S - 1, p.ks_s, lg);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = Mfma2<T>::run(kf[kk], qf[n][kk], acc);
      float x[4];
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = s0 + lg * 4 + r;
        float v = logit_chain<T>(acc[r], p);
        if (qi[n] >= L && s >= L && (s - L) > (qi[n] - L))                  // corner mask (:545-551)
          v = Elem<T>::to_f32(Elem<T>::from_f32(v + fmin_v));
        x[r] = (s < S) ? v : -INFINITY;
        mx = fmaxf(mx, x[r]);
      }
      const float mn = fmaxf(m[n], mx);
      if (mn != -INFINITY) {
        float z = Z[n] * pkv_exp(m[n] - mn);
#pragma unroll
        for (int r = 0; r < 4; ++r) z += pkv_exp(x[r] - mn);
        Z[n] = z;
        m[n] = mn;
      }
    }
  }
  float2* rs = p.rowstat + (int64_t)bh * S;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float mm = m[n], zz = Z[n];
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float mo = __shfl_xor(mm, o, 64), zo = __shfl_xor(zz, o, 64);
      const float mn = fmaxf(mm, mo);
      const float za = (mm == -INFINITY) ? 0.f : zz * pkv_exp(mm - mn);
      const float zb = (mo == -INFINITY) ? 0.f : zo * pkv_exp(mo - mn);
      mm = mn; zz = za + zb;
    }
    if (lg == 0 && qi[n] < S) rs[qi[n]] = make_float2(mm, 1.0f / zz);   // (row max, 1 / row sum): ATen CPU softmax multiplies
  }
}

// one workgroup = 128 key columns (4 waves x 32); loop over all S query rows
template <typename T>
__global__ __launch_bounds__(256) void h2o_colsum_kernel(H2OParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float2* rs = p.rowstat + (int64_t)bh * S;
  const int k0 = blockIdx.x * 128 + wave * 32;

  uint4 kf[2][4];
  int kj[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    kj[n] = k0 + n * 16 + li;
    load_frags(kf[n], kb, kj[n] < S ? kj[n] : S - 1, p.ks_s, lg);
  }
  float col[2] = {0.f, 0.f};

  for (int i0 = 0; i0 < S; i0 += 16) {
    uint4 qf[4];
    const int ir = i0 + li;
    load_frags(qf, qb, ir < S ? ir : S - 1, p.qs_s, lg);
    float2 st[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + lg * 4 + r;
      st[r] = rs[i < S ? i : S - 1];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = Mfma2<T>::run(qf[kk], kf[n][kk], acc);   // D[query][key]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + lg * 4 + r;
        const float x = logit_chain<T>(acc[r], p);       // keys < L never touch the masked corner
        const float pr = pkv_exp(x - st[r].x) * st[r].y;    // fp32 softmax (:553)
        const float pq = Elem<T>::to_f32(Elem<T>::from_f32(pr));
        if (i < S) col[n] += pq;                         // sum over all rows, fp32 (:554)
      }
    }
  }
  uint16_t* out = reinterpret_cast<uint16_t*>(p.scores) + (int64_t)bh * p.scores_stride;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
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
  dim3 grid((p.S + 127) / 128, p.B * p.H);
  if (dtype == 0) hipLaunchKernelGGL(h2o_stats_kernel<BF16>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(h2o_stats_kernel<F16>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_h2o_colsum(int dtype, const H2OParams& p, hipStream_t st) {
  const int L = p.S - p.w;
  dim3 grid((L + 127) / 128, p.B * p.H);
  if (dtype == 0) hipLaunchKernelGGL(h2o_colsum_kernel<BF16>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(h2o_colsum_kernel<F16>, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}

}  // namespace pkv
