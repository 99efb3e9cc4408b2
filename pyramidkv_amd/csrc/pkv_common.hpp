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

template <typename T> __device__ __forceinline__ uint32_t order_key(uint16_t h) {
  uint32_t k = (uint32_t)h ^ ((h & 0x8000u) ? 0xffffu : 0x8000u);
  if (k == 0x7fffu) k = 0x8000u;                      // -0 == +0
  k += KeyBias<T>::v;
  return k > 0xffffu ? 0xffffu : k;
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

// exp(x) for x <= ~0 (softmax arguments): Cody-Waite reduction + v_exp_f32 on |t| <= 0.5, ~1.5 ulp.
// The model-dtype rounding that follows every use absorbs it exactly as it absorbs ATen's own
// vectorised exp (tests bound the disagreement with the CPU oracle).  -inf, x < -104 and NaN give +0.
__device__ __forceinline__ float pkv_exp(float x) {
  // branch-free: evaluate on a clamped argument, then select the underflow / NaN results
  const float xc = fmaxf(x, -104.0f);
  const float L2E = 1.44269504088896340736f;
  const float n = rintf(xc * L2E);
  float r = fmaf(n, -0.693145751953125f, xc);            // ln2_hi: 12 trailing zero bits, n*ln2_hi exact
  r = fmaf(n, -1.42860682030941723212e-6f, r);           // ln2_lo
  const float p = __builtin_amdgcn_exp2f(r * L2E);       // v_exp_f32, argument in [-0.5, 0.5]
  const float e = ldexpf(p, (int)n);
  return (x > -104.0f) ? e : 0.0f;                       // NaN arguments give 0 (inputs are assumed finite)
}

// correctly rounded x / c for a loop-invariant c (rc = RN(1/c)): one Newton-Markstein correction step.
__device__ __forceinline__ float div_const(float x, float c, float rc) {
  const float q = x * rc;
  const float r = fmaf(-q, c, x);
  return fmaf(r, rc, q);
}

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable with nontemporal builtins

union U4 {
  uint4 v;
  uint16_t h[8];
  uint32_t w[4];
};

}  // namespace pkv
