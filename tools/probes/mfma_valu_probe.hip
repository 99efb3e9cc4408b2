// How many vector instructions hide beside one MFMA on a gfx950 SIMD?  Loop of [1 MFMA + NV vector instructions],
// four rotating accumulators (no dependent-MFMA stalls), one or two waves per SIMD.  Prints shader cycles per group.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int KIND> __device__ __forceinline__ void filler(float& a, float b) {
  if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(b));
  if constexpr (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0" : "+v"(a));
  if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  if constexpr (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
}

template <int MF, int NV, int KIND>
__global__ __launch_bounds__(512) void probe(float* out, uint64_t* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  f32x4 c4[4] = {};
  f32x16 c16[2] = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  const float bb = 1.0001f;
  typedef float f32x2p __attribute__((ext_vector_type(2)));
  f32x2p pk[4], pkb = {1.0001f, 0.9999f};
  for (int i = 0; i < 4; ++i) pk[i] = f32x2p{0.001f * lane, 0.002f * i};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (MF == 0) c4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[u & 3], 0, 0, 0);
      if constexpr (MF == 1) c16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[u & 1], 0, 0, 0);
      if constexpr (MF == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c4[u & 3]) : "v"(a), "v"(b));   // accumulators in AGPRs
      if constexpr (MF == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c16[u & 1]) : "v"(a), "v"(b));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pk[(u * NV + k) & 3]) : "v"(pkb));   // 2 fp32 FMAs per instruction
        else filler<KIND>(v[(u * NV + k) & 7], bb);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c4[i][j];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += c16[i][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += pk[i].x + pk[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MF, int NV, int KIND>
void run(int waves_per_simd, float* out, uint64_t* cyc) {
  const int iters = 2000;
  const int threads = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MF, NV, KIND><<<256, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MF, NV, KIND><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime ticks at 100 MHz on this part?  report both: wall ns per group per wave and counter per group
  const double groups = (double)iters * 8;
  printf("mf=%d kind=%d nv=%2d waves/simd=%d : %.1f ns/group/wave-slot  (%.2f counter ticks/group)  -> %.1f cyc@2.4GHz per group per SIMD\n", MF, KIND, NV,
         waves_per_simd, 1e6 * ms / groups, (double)c / groups, 1e6 * ms / groups * 2.4 / 1.0);
}

template <int MF, int KIND> void sweep(float* out, uint64_t* cyc) {
  for (int w = 1; w <= 2; ++w) {
    run<MF, 0, KIND>(w, out, cyc); run<MF, 1, KIND>(w, out, cyc); run<MF, 2, KIND>(w, out, cyc); run<MF, 3, KIND>(w, out, cyc);
    run<MF, 4, KIND>(w, out, cyc); run<MF, 5, KIND>(w, out, cyc); run<MF, 6, KIND>(w, out, cyc); run<MF, 8, KIND>(w, out, cyc);
    run<MF, 10, KIND>(w, out, cyc); run<MF, 12, KIND>(w, out, cyc);
  }
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  if (getenv("PROBE_NOMFMA")) {   // vector instructions alone (MF = 4: no MFMA in the group): plain fma, packed fma, exp2, integer multiply
    run<4, 8, 0>(1, out, cyc); run<4, 8, 0>(2, out, cyc); run<4, 8, 3>(1, out, cyc); run<4, 8, 3>(2, out, cyc);
    run<4, 8, 2>(1, out, cyc); run<4, 8, 2>(2, out, cyc); run<4, 8, 4>(1, out, cyc); run<4, 8, 4>(2, out, cyc);
    run<0, 4, 3>(2, out, cyc); run<1, 6, 3>(2, out, cyc);
    return 0;
  }
  if (getenv("PROBE_AGPR")) { sweep<2, 0>(out, cyc); sweep<3, 0>(out, cyc); return 0; }
  sweep<0, 0>(out, cyc); sweep<1, 0>(out, cyc);
  sweep<0, 1>(out, cyc); sweep<0, 2>(out, cyc);
  sweep<1, 2>(out, cyc);
  return 0;
}
