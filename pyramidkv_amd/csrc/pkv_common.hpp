// pkv_common.hpp — device helpers shared by the libpkv kernels (gfx950 / CDNA4 only).
//
// Rounding points follow the reference's eager PyTorch pipeline (pyramidkv_utils.py:317-333):
// every intermediate that the reference materialises in the model dtype is rounded here with
// round-to-nearest-even, exactly once, at the same place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PKV_WAVE 64

namespace pkv {

struct BF16 {};  // tag types: element = 16-bit pattern in a uint16_t
struct F16 {};

template <typename T> struct Elem;

template <> struct Elem<BF16> {
  static __device__ __forceinline__ float to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) {  // v_cvt_pk_bf16_f32: hardware RNE (checked
    __bf16 b = (__bf16)f;                                         // bit-for-bit against torch in the GPU tests)
    return __builtin_bit_cast(uint16_t, b);
  }
  static __device__ __forceinline__ float finfo_min() { return __uint_as_float(0xff7f0000u); }  // -3.3895e38
  static __device__ __forceinline__ uint16_t neg_inf() { return 0xff80u; }
  static __device__ __forceinline__ bool is_nan(uint16_t h) { return (h & 0x7fffu) > 0x7f80u; }
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// two fp32 values -> two model-dtype values packed in one dword (lo = a, hi = b): one v_cvt_pk_* instruction
template <typename T> __device__ __forceinline__ uint32_t round_pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t round_pack2<BF16>(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <> __device__ __forceinline__ uint32_t round_pack2<F16>(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

template <> struct Elem<F16> {
  static __device__ __forceinline__ float to_f32(uint16_t h) {
    _Float16 x = __builtin_bit_cast(_Float16, h);
    return (float)x;
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {  // v_cvt_f16_f32: RNE, subnormals kept, overflow -> inf
    _Float16 x = (_Float16)f;
    return __builtin_bit_cast(uint16_t, x);
  }
  static __device__ __forceinline__ float finfo_min() { return -65504.0f; }
  static __device__ __forceinline__ uint16_t neg_inf() { return 0xfc00u; }
  static __device__ __forceinline__ bool is_nan(uint16_t h) { return (h & 0x7fffu) > 0x7c00u; }
};

// Order-preserving 16-bit key: larger key <=> larger value.  Monotone map of the IEEE bit pattern
// (negative: ~h, non-negative: h | 0x8000), -0 folded onto +0, then a saturating bias that collapses
// every (positive) NaN onto the top key 0xffff while keeping +inf just below it (torch sorts NaN as
// greatest).  Real keys are always >= 1; key 0 is reserved for padding.
template <typename T> struct KeyBias;
template <> struct KeyBias<BF16> { static constexpr uint32_t v = 0x007eu; };   // +inf 0xff80 -> 0xfffe
template <> struct KeyBias<F16> { static constexpr uint32_t v = 0x03feu; };    // +inf 0xfc00 -> 0xfffe

// key buckets of the small-k counting sort: bf16 has 128 keys per binade (one key per bucket), fp16 1024 (8 per bucket)
template <typename T> struct KeyShift;
template <> struct KeyShift<BF16> { typedef char tag; };
template <> struct KeyShift<F16> { typedef short tag; };

template <typename T> __device__ __forceinline__ uint32_t order_key(uint16_t h) {
  uint32_t k = (uint32_t)h ^ ((h & 0x8000u) ? 0xffffu : 0x8000u);
  if (k == 0x7fffu) k = 0x8000u;                      // -0 == +0
  k += KeyBias<T>::v;
  return k > 0xffffu ? 0xffffu : k;
}

// inverse of order_key for finite scores and +-inf (NaNs were collapsed onto the top key): the score's own bit pattern
template <typename T> __device__ __forceinline__ uint16_t key_to_raw(uint32_t key) {
  const uint32_t k = key - KeyBias<T>::v;
  return (k & 0x8000u) ? (uint16_t)(k ^ 0x8000u) : (uint16_t)(~k);
}

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));

// the same map on two packed 16-bit values (v_pk_* ops: ~3.5 instructions per key)
template <typename T> __device__ __forceinline__ uint32_t order_key_pk(uint32_t x) {
  const i16x2 xs = __builtin_bit_cast(i16x2, x);
  const u16x2 sm = __builtin_bit_cast(u16x2, (i16x2)(xs >> (int16_t)15));          // 0xffff where negative
  u16x2 k = __builtin_bit_cast(u16x2, x) ^ (sm | (u16x2)(0x8000));
  const u16x2 t = k ^ (u16x2)(0x7fff);                                             // 0 where the value was -0
  k += __builtin_elementwise_sub_sat((u16x2)(1), t);                               // -0 -> +0
  k = __builtin_elementwise_add_sat(k, (u16x2)((uint16_t)KeyBias<T>::v));
  return __builtin_bit_cast(uint32_t, k);
}

// exp(x) for x <= ~0 (softmax arguments), ~1.4 ulp: range reduction in the log2 domain with a two-term
// log2(e) (f = x*log2e - n is formed by two FMAs, exact to ~2^-48 before its single rounding), v_exp_f32 on
// |f| <= 0.5, ldexp.  Branch-free; -inf, NaN and x < -104 give +0 (the result underflows in ldexp).
// The model-dtype rounding that follows every use absorbs the last-ulp differences exactly as it absorbs
// those of ATen's own vectorised exp (tests bound the disagreement with the CPU oracle).
__device__ __forceinline__ float pkv_exp(float x) {
  const float xc = fmaxf(x, -104.0f);
  const float L2E_HI = 1.44269502162933349609375f;       // fp32(log2 e)
  const float L2E_LO = 1.92596299112661746e-8f;          // log2 e - L2E_HI
  const float n = rintf(xc * L2E_HI);
  float f = fmaf(xc, L2E_HI, -n);
  f = fmaf(xc, L2E_LO, f);
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// pkv_exp on a pair: the multiply and the two FMAs of the range reduction are packed fp32 instructions (same IEEE
// operations, same results as two scalar calls); the kernels that evaluate one exponential per logit are bound by the
// SIMD issue port, where every instruction counts.
typedef float pkv_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pkv_f32x2 pkv_exp_pair(pkv_f32x2 x) {
  const pkv_f32x2 xc = {fmaxf(x.x, -104.0f), fmaxf(x.y, -104.0f)};
  const pkv_f32x2 hi = {1.44269502162933349609375f, 1.44269502162933349609375f};       // fp32(log2 e)
  const pkv_f32x2 lo = {1.92596299112661746e-8f, 1.92596299112661746e-8f};             // log2 e - hi
  const pkv_f32x2 t = xc * hi;
  const pkv_f32x2 n = {rintf(t.x), rintf(t.y)};
  pkv_f32x2 f = __builtin_elementwise_fma(xc, hi, -n);
  f = __builtin_elementwise_fma(xc, lo, f);
  return pkv_f32x2{ldexpf(__builtin_amdgcn_exp2f(f.x), (int)n.x), ldexpf(__builtin_amdgcn_exp2f(f.y), (int)n.y)};
}

// exp(x) for the merge of softmax PARTIAL statistics (x = m_t - M <= 0, or -inf): hardware v_exp_f32 on x * log2(e), ~1e-6
// relative.  The merged Z = sum_t l_t exp(m_t - M) only enters the probabilities through 1/Z, and its fp32 summation order
// already differs from ATen's by as much; the probabilities themselves use pkv_exp.
__device__ __forceinline__ float pkv_exp_stat(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }

// correctly rounded x / c for a loop-invariant c (rc = RN(1/c)): one Newton-Markstein correction step.
__device__ __forceinline__ float div_const(float x, float c, float rc) {
  const float q = x * rc;
  const float r = fmaf(-q, c, x);
  return fmaf(r, rc, q);
}

// "attn / math.sqrt(head_dim)" (reference pyramidkv_utils.py:317) on a value already rounded to the model dtype, result
// about to be rounded to the model dtype again: ONE multiply by a host-chosen constant `rc` in both scale modes
// (pkv_api.hip scale_multiplier; exhaustive checks over every finite 16-bit input in tests/test_abi_and_host.py):
//   head_dim 64 / 256: the divisor is a power of two, rc = 1/8, 1/16 is the division;
//   head_dim 128, bf16: round(x / fp32(sqrt(128))) == round(x * fp32(1/sqrt(128))) for EVERY finite bf16 x;
//   head_dim 128, fp16: that constant differs from the division on 52 inputs, its fp32 NEIGHBOUR ABOVE on none (round 6; the
//     "div" mode used to take a correctly rounded division, 3 instructions per logit - +40 % on the fp16 H2O pair) - so "div"
//     multiplies by nextafter(fp32(1/sqrt(128)), 1) and "rcp" (what ATen's HIP kernels do) by fp32(1/sqrt(128)).
template <typename T> __device__ __forceinline__ float scale_logit(float x, int, float, float rc) { return x * rc; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix sum across the 64 lanes of a wave: DPP row shifts + row broadcasts (pure VALU, no LDS)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return v;
}

// wave-wide sum, broadcast to every lane: DPP scan + readlane (no LDS traffic)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

// wave-wide fp64 sum, broadcast to every lane: the same DPP row shifts / row broadcasts on the two 32-bit halves (12 DPP moves +
// 6 adds + 2 readlanes; a __shfl_xor butterfly on doubles is 12 ds_bpermute with their address arithmetic, ~85 instructions -
// half of what the one-launch budget kernel spent on its ratios).  Lanes without a source read +0.0.
__device__ __forceinline__ double wave_sum_f64(double v) {
#define PKV_DPP_F64_STEP(CTRL, ROWMASK)                                                                                   \
  do {                                                                                                                   \
    const unsigned long long b_ = __builtin_bit_cast(unsigned long long, v);                                             \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b_, CTRL, ROWMASK, 0xf, false);          \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b_ >> 32), CTRL, ROWMASK, 0xf, false);  \
    v += __builtin_bit_cast(double, ((unsigned long long)hi_ << 32) | lo_);                                              \
  } while (0)
  PKV_DPP_F64_STEP(0x111, 0xf);   // row_shr:1
  PKV_DPP_F64_STEP(0x112, 0xf);   // row_shr:2
  PKV_DPP_F64_STEP(0x114, 0xf);   // row_shr:4
  PKV_DPP_F64_STEP(0x118, 0xf);   // row_shr:8
  PKV_DPP_F64_STEP(0x142, 0xa);   // row_bcast:15 -> rows 1,3
  PKV_DPP_F64_STEP(0x143, 0xc);   // row_bcast:31 -> rows 2,3
#undef PKV_DPP_F64_STEP
  const unsigned long long t_ = __builtin_bit_cast(unsigned long long, v);
  const uint32_t tl_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t_, 63), th_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t_ >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)th_ << 32) | tl_);
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable with nontemporal builtins

union U4 {
  uint4 v;
  uint16_t h[8];
  uint32_t w[4];
};

}  // namespace pkv
