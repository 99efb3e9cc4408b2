// pkv_mfma.hpp - the one MFMA shape libpkv uses (gfx950): v_mfma_f32_16x16x32_{bf16,f16}.
//   A = 16 rows x 32 k: lane (li = lane & 15 -> row, lg = lane >> 4 -> 8-element k-chunk of the 32-wide step)
//   B = 32 k x 16 columns: lane li -> column, lg -> k-chunk
//   D[row i][col j] sits in lane (j + 16 * (i / 4)), register i % 4
#pragma once
#include "pkv_common.hpp"

namespace pkv {

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8_t;

template <typename T> struct Mfma;
template <> struct Mfma<BF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<F16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

}  // namespace pkv
