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
  static __device__ __forceinline__ uint16_t from_f32(float f) {  // RNE, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
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

// Order-preserving 16-bit key: larger key <=> larger value; all NaNs collapse to the top key
// (torch sorts NaN as greatest), -0 == +0.  Real (non-padding) keys are always >= 1.
template <typename T> __device__ __forceinline__ uint32_t order_key(uint16_t h) {
  if (Elem<T>::is_nan(h)) return 0xffffu;
  if (h == 0x8000u) h = 0;
  return (h & 0x8000u) ? (uint32_t)(uint16_t)(~h) : (uint32_t)(h | 0x8000u);
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
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable with nontemporal builtins

union U4 {
  uint4 v;
  uint16_t h[8];
  uint32_t w[4];
};

}  // namespace pkv
