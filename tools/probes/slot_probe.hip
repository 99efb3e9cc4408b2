// What stops the H2O epilogue from hiding beside its MFMAs?  The pipeline slot of pkv_h2o.hip (one 32x32x16 MFMA + two
// element chains round/scale/round/fma/exp2 + one pack; every 4th slot a column-sum MFMA) with the memory traffic
// removed, in variants.  Prints ns per slot per wave (x clock = cycles).
//   hipcc --offload-arch=gfx950 -O3 -o slot_probe slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2v;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2v;
__device__ __forceinline__ uint32_t pack2(float a, float b) { f32x2v v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v)); }
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define PIN() __builtin_amdgcn_sched_barrier(0)

// VAR bits: 1 chains read a static register set instead of the previous phase's accumulators
//           2 no exp2 (fma instead)      4 no column-sum MFMA      8 no main MFMA (vector work alone)
//           16 no vector work (MFMAs alone)   32 chains placed BEFORE the MFMA in each slot   64 four chains interleaved over two slots
template <int VAR>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 kf[2][8], ring[4];
  for (int i = 0; i < 16; ++i) kf[i >> 3][i & 7] = u32x4{(uint32_t)lane * 3 + i, 0x3f803f80u, (uint32_t)i, 0x3c003c00u};
  for (int i = 0; i < 4; ++i) ring[i] = u32x4{0x3f803f80u + i, 0x3f003f00u, 0x40004000u, 0x3e803e80u};
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  f32x16 acc0[2], acc1[2], cacc[2], fixed[2];
  for (int i = 0; i < 16; ++i) { acc0[0][i] = in[lane + i]; acc0[1][i] = in[lane + 16 + i]; acc1[0][i] = 0; acc1[1][i] = 0; cacc[0][i] = 0; cacc[1][i] = 0; fixed[0][i] = in[i]; fixed[1][i] = in[i + 7]; }
  float st[16];
  for (int i = 0; i < 16; ++i) st[i] = in[64 + i];
  const float rc = in[100], L2E = 1.44269504f;
  auto chain = [&](float a, float c) {
    float x = __uint_as_float(pack2(0.f, a));
    x = x * rc;
    x = __uint_as_float(pack2(0.f, x));
    const float y = __builtin_fmaf(x, L2E, c);
    return (VAR & 2) ? __builtin_fmaf(y, y, x) : __builtin_amdgcn_exp2f(y);
  };
  auto phase = [&](const f32x16 (&ac)[2], f32x16 (&an)[2]) {
    uint32_t pk[4];
    const f32x16 (&src)[2] = (VAR & 1) ? fixed : ac;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kk = s >> 1, cb = s & 1, ce = s >> 3, j0 = 2 * (s & 7);
      float e[2];
      if ((VAR & 32) && !(VAR & 16)) { e[0] = chain(src[ce][j0], st[j0]); e[1] = chain(src[ce][j0 + 1], st[j0 + 1]); PIN(); }
      if (!(VAR & 8)) an[cb] = mfma(ring[kk & 3], kf[cb][kk], an[cb]);
      PIN();
      if (!(VAR & 16)) {
        if (!(VAR & 32)) { e[0] = chain(src[ce][j0], st[j0]); e[1] = chain(src[ce][j0 + 1], st[j0 + 1]); }
        pk[s & 3] = pack2(e[0], e[1]);
        if ((s & 3) == 3) {
          const u32x4 pb = {pk[0], pk[1], pk[2], pk[3]};
          if (VAR & 4) cacc[ce][0] += __uint_as_float(pb[0] ^ pb[1] ^ pb[2] ^ pb[3]);
          else cacc[ce] = mfma(ones, pb, cacc[ce]);
        }
      }
      PIN();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ring[i]));     // the ring changes every phase in the real kernel: no hoisting
  };
  for (int it = 0; it < iters; ++it) {
    phase(acc0, acc1);
    phase(acc1, acc0);
  }
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) sum += acc0[0][i] + acc0[1][i] + acc1[0][i] + acc1[1][i] + cacc[0][i] + cacc[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int VAR> void run(float* out, const float* in, int wgs_per_cu) {
  const int iters = 1500;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<VAR><<<256 * wgs_per_cu, 256>>>(out, in, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<VAR><<<256 * wgs_per_cu, 256>>>(out, in, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("var=%3d waves/simd=%d : %.1f ns per slot per wave -> %.1f cycles at 2.33 GHz per slot per SIMD\n", VAR, wgs_per_cu, 1e6 * ms / (iters * 32.0),
         1e6 * ms / (iters * 32.0) * 2.33 / wgs_per_cu);
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&in, 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f * (i % 13) - 2.f;
  h[100] = 0.0883883f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0>(out, in, w); run<1>(out, in, w); run<2>(out, in, w); run<4>(out, in, w); run<6>(out, in, w); run<8>(out, in, w); run<12>(out, in, w);
    run<16>(out, in, w); run<20>(out, in, w); run<32>(out, in, w); run<33>(out, in, w); run<7>(out, in, w);
  }
  return 0;
}
