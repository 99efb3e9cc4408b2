// pkv_h2o.hip — H2O score kernels (gfx950): attention mass each key receives from ALL S query rows.
//
//   reference pyramidkv_utils.py:544-554:
//     A = (Q K^T)/sqrt(D) [S x S], only the last w x w corner causally masked (early rows DO see
//     future keys - a reference quirk that is reproduced), P = softmax_fp32(A).to(dtype),
//     score[j] = sum_i P[i][j] for j < S-w (fp32 accumulate, one rounding).
//
// S x S is never materialised (68.7 GB bf16 at S=32k).  Two MFMA passes over the S x S tile space:
//   h2o_stats_kernel   per query row: c_row = -log2 sum_j exp(x_ij)  (softmax denominator, log domain)
//   h2o_colsum_kernel  per key column: sum over all query rows of round(exp2(x*log2e + c_row))
// Both big passes recompute the logits with mfma_f32_16x16x32 and apply the reference's three roundings.
// exp: every probability is ONE v_exp_f32 of fma(x, log2e, c_row) (the 1/Z factor folded into the exponent), relative
// error ~|x*log2e| * 2^-24 (about 1e-6).  That is ~20x the error of the window path's exp, and deliberately so: a score
// here is a sum over S >= 1000s of rounded probabilities, so the rare rounding flips (1e-6 / 2^-9 per element) average out
// far below the model-dtype resolution of the sum (measured against the oracle in tests/test_gpu_parity.py::test_h2o_*),
// while the kernels are instruction-issue-bound and the accurate exp costs 8 vector instructions per S x S element.
// Probabilities below 2^-126 are +0 (the hardware exp2 and the MFMA operands flush them): scores below ~1e-35 come out 0.
//
// What bounds them (round 4 account with counters, clocks and power: profiles/r04/h2o/h2o_account.md).  A SIMD issues ONE
// vector instruction per 4 cycles (wave64 on 16 lanes; v_exp_f32 takes two such slots), and while an MFMA executes the
// vector port is blocked for part of its duration: cycles ~ 4 x (vector instructions + exps) + ~0.4 x matrix-pipe cycles,
// whatever the order of the instructions.  On random data the chip additionally sits at its ~1300 W power cap and trades
// cycles for clock - which is why THIS form ships and not round 4's 32x32x16 software pipeline (tools/probes/
// h2o_wide_pipeline.hip: 14.5 % fewer cycles in pass 1, 7 % faster on zero-filled operands, but clocked 8 % lower on N(0,1)
// data and 3 % SLOWER there than the kernels below; its accumulators move twice the register bytes per flop).
// The levers that are in:
//   * no running maximum in pass 1 (round 4; reference point changed in round 5).  softmax needs SOME reference point M_i
//     with exp(x - M_i) in fp32 range, not the maximum: M_i = the lane's maximum over the FIRST 64-key tile (tracked there,
//     then frozen; a real prompt's attention sinks sit in that tile).  Z_i >= 1 and everything up to +88 above the sampled
//     maximum still sums in range: 6 vector instructions per element (round, scale, round, fma, exp2, add) instead of
//     6.75 + a wave-uniform branch per 4 elements.  A workgroup with a row whose Z overflowed repeats its rows with the
//     exact online maximum - same kernel, second instance of the loop.  c_row = -(m log2e + log2 Z) does not depend on
//     the reference point, so pass 2 does not care.  (Round 4's point was a norm bound |q_i| max_j|k_j| / sqrt(D) - 64 from
//     a key-norm scan: one large-norm key that no row attends to put every row outside its window - the advisor's finding;)
//   * bf16 rounding as v_cvt_pk_bf16_f32 v, 0, x: the rounded value lands in the HIGH half over a zero low half, which
//     IS its fp32 representation - 1 instruction per rounding instead of pack + shift/mask (1.5);
//   * no packed-fp32 arithmetic (v_pk_mul/fma_f32 cost several issue slots beside MFMAs): compiled with -fno-slp-vectorize;
//   * pass 2 sums the columns ON THE MATRIX PIPE: P rounded to the model dtype is exact as an MFMA operand, so
//     ones[16 x 32] x P[32 queries x 16 keys] adds 32 query rows per instruction into an fp32 accumulator (all 16 rows
//     of the result hold the same sums) instead of unpack + add per element;
//   * rows past S carry c_row = -inf (probability exactly 0) and tile loads are raw buffer loads whose addresses are
//     advanced on the scalar unit and which return 0 past the end: no tail code, no per-tile vector address arithmetic;
//   * pass 1 runs its end-of-row / masked-corner handling in a separate instance of the loop body.
// Pass 1: 6 vector instructions per element (2 roundings + scale 3, fma, exp2, add) + 1 MFMA, pass 2: 5.5 (3, fma, exp2,
// pack 0.5) + 1.125 MFMA.  Built, measured and removed over three rounds (h2o_account.md, LABNOTES.md): rotated loops with
// pinned MFMAs, 32x32x16 MFMAs with an explicit software pipeline, the scale by 1/sqrt(D) on the matrix pipe, forced
// occupancy, tile loads without a memory stream (the loop is not memory bound).
#include <type_traits>
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
template <typename T> struct Ones2;                   // two 1.0 of the model dtype
template <> struct Ones2<BF16> { static constexpr uint32_t v = 0x3f803f80u; };
template <> struct Ones2<F16> { static constexpr uint32_t v = 0x3c003c00u; };

// v_max3_f32 on raw registers: fmaxf() of a value that came through integer bit operations makes the compiler
// canonicalise it first (one extra v_max per operand)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// One logit through the reference's two roundings.
template <typename T>
__device__ __forceinline__ float logit_chain(float acc, const H2OParams& p) {
  float x = Elem<T>::to_f32(Elem<T>::from_f32(acc));                       // matmul output dtype (:544)
  x = scale_logit<T>(x, p.scale_mode, p.sqrt_d, p.rcp_sqrt_d);             // / math.sqrt(head_dim)
  return Elem<T>::to_f32(Elem<T>::from_f32(x));
}
template <>
__device__ __forceinline__ float logit_chain<BF16>(float acc, const H2OParams& p) {
  float x = __uint_as_float(round_pack2<BF16>(0.f, acc));                  // matmul output dtype (:544); low half = +0
  x = x * p.rcp_sqrt_d;                                                    // / math.sqrt(head_dim): exact for bf16, see scale_logit
  return __uint_as_float(round_pack2<BF16>(0.f, x));
}
template <typename T>
__device__ __forceinline__ void logits4(const f32x4& acc, const H2OParams& p, float (&x)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = logit_chain<T>(acc[r], p);
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

// KS = head_dim / 32 MFMA k-steps (2, 4, 8 for head sizes 64, 128, 256).  A row is CPR = 4 * KS chunks of 16 B.
template <int KS> struct Stager {      // one thread's share of a 64-row tile: 64 * CPR / 256 = KS chunks of 16 B
  u32x4 v[KS];
};

// The streamed matrix of one head as a raw buffer.  Per thread one byte offset (row tid/16, chunk tid%16), computed
// once; per tile the descriptor's base and extent move on the scalar unit; rows r0+16i come through the scalar offset.
// A load that starts past `extent` returns 0, so the last tile needs no clamping.
struct TileStream {
  const uint16_t* base;                // row 0 of the head
  int64_t stride_b;                    // bytes between rows
  int nrows;
  uint32_t voff;                       // this thread's byte offset inside a tile
};
template <int KS>
__device__ __forceinline__ TileStream make_stream(const uint16_t* base, int64_t stride, int nrows, int tid) {
  constexpr int CPR = 4 * KS;
  TileStream s;
  s.base = base; s.stride_b = stride * 2; s.nrows = nrows;
  s.voff = (uint32_t)(tid / CPR) * (uint32_t)s.stride_b + (uint32_t)(tid % CPR) * 16u;
  return s;
}
template <int KS>
__device__ __forceinline__ void stage_load(Stager<KS>& st, const TileStream& s, int row0) {
  constexpr int CPR = 4 * KS, RPP = 256 / CPR;                             // rows covered by one pass of the 256 threads
  const int left = s.nrows - row0;                                          // rows still inside the matrix (scalar)
  const int64_t ext = left > 0 ? (int64_t)(left - 1) * s.stride_b + CPR * 16 : 0;
  const uint32_t extent = ext > 0xffffffffll ? 0xffffffffu : (uint32_t)ext;
  const char* tile = reinterpret_cast<const char*>(s.base) + (left > 0 ? (int64_t)row0 * s.stride_b : 0);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tile), 0, extent, 0x00020000);
#pragma unroll
  for (int i = 0; i < KS; ++i)
    st.v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, s.voff, (uint32_t)(RPP * i) * (uint32_t)s.stride_b, 0));
}
// chunk c of row r sits at chunk c ^ (r & SW): the fragment read below (16 rows x one chunk per 4-lane group) is then
// bank-conflict free; SW = 15 for 16 and 32 chunks per row, 7 for 8
template <int KS>
__device__ __forceinline__ void stage_store(const Stager<KS>& st, u32x4* tile, int tid) {
  constexpr int CPR = 4 * KS, RPP = 256 / CPR, SW = CPR >= 16 ? 15 : CPR - 1;
  const int c = tid % CPR, r0 = tid / CPR;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int r = r0 + RPP * i;
    tile[r * CPR + (c ^ (r & SW))] = st.v[i];
  }
}
template <int KS>
__device__ __forceinline__ void read_frags(u32x4 (&f)[KS], const u32x4* tile, int sub, int li, int lg) {
  constexpr int CPR = 4 * KS, SW = CPR >= 16 ? 15 : CPR - 1;
  const int r = sub * 16 + li;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) f[kk] = tile[r * CPR + ((kk * 4 + lg) ^ (li & SW))];
}
template <int KS>
__device__ __forceinline__ void load_frags(u32x4 (&f)[KS], const uint16_t* base, int64_t row, int64_t stride, int lg) {
  const uint16_t* r = base + row * stride + lg * 8;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) f[kk] = *reinterpret_cast<const u32x4*>(r + kk * 32);
}
// the 16 MFMAs of one 16-row sub-tile (fragments f, the A operand) against the wave's 64 resident rows (B): four
// independent accumulators back to back, no dependent-MFMA stall.  D[streamed row][resident row].
// LOWPRIO: the wave drops to priority 0 while it issues its 16 MFMAs and runs its vector epilogue at priority 1, so the
// epilogue instructions of the other waves of the SIMD win the issue slots an MFMA leaves free.  Round 5 A/B (complete
// libraries, same session, profiles/r05/h2o_ab.txt): pass 2 -1.6 % / -1.3 % / -3.0 % at S = 32768 / 8192 / 4096, pass 1
// -0.4 % / +2.3 % / +3.7 % (its statistics chain is shorter than the MFMA group: nothing to win, and the priority switches cost);
// the opposite assignment (MFMA waves first) changes nothing.  So pass 2 uses it and pass 1 does not.
template <typename T, int KS, bool LOWPRIO = false>
__device__ __forceinline__ void mm16(f32x4 (&acc)[4], const u32x4 (&f)[KS], const u32x4 (&res)[4][KS]) {
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (LOWPRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = Mfma2<T>::run(f[kk], res[n][kk], acc[n]);
  if (LOWPRIO) __builtin_amdgcn_s_setprio(1);
}

// Pass 1: per query row, c_row = -log2 sum_j exp(x_ij).  Resident = 256 query rows, streamed = K.
// Per-lane statistics (lane's column = one query, 4 keys per 16-key subtile).  Frozen path: exponentials relative to the lane's
// maximum over the first tile; exact path (a row overflowed): the running maximum is only rescaled when some lane of the
// wave actually found a larger logit (wave-uniform branch).
template <typename T, int KS>
__global__ __launch_bounds__(256) void h2o_stats_kernel(H2OParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char h2o_smem[];               // 2 tiles (64 KB at head size 256)
  u32x4 (*tiles)[HT * 4 * KS] = reinterpret_cast<u32x4 (*)[HT * 4 * KS]>(h2o_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const int q0 = blockIdx.x * HWG + wave * HR;
  u32x4 qf[4][KS];
  int qi[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    qi[n] = q0 + n * 16 + li;
    load_frags<KS>(qf[n], qb, qi[n] < S ? qi[n] : S - 1, p.qs_s, lg);
  }
  const float L2E = 1.44269504088896340736f;
  float m[4], mL[4], Z[4];          // running max, -max*log2e, running sum of exp
  const TileStream ks = make_stream<KS>(kb, p.ks_s, S, tid);
  Stager<KS> stg;
  stage_load<KS>(stg, ks, 0);
  stage_store<KS>(stg, tiles[0], tid);
  __syncthreads();
  const int ntiles = (S + HT - 1) / HT;
  // tiles that touch the end of the row or the masked corner run the EDGE instance of the body; all others carry no
  // masking code at all
  auto tile_body = [&](const int t, auto edge_tag, auto track_tag) {
    constexpr bool edge = decltype(edge_tag)::value, track = decltype(track_tag)::value;
    const int s_tile = t * HT;
    const u32x4* cur = tiles[t & 1];
    stage_load<KS>(stg, ks, s_tile + HT);                                     // in flight during the compute below; past the end: zeros
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      u32x4 kf[KS];
      read_frags<KS>(kf, cur, sub, li, lg);
      const int s0 = s_tile + sub * 16;
      f32x4 accs[4];
      mm16<T, KS>(accs, kf, qf);                                              // D[key][query]
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float x[4];
        logits4<T>(accs[n], p, x);
        if (edge) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s = s0 + lg * 4 + r;
            // corner mask (:545-551): the reference adds finfo.min, whose exp(x - max) is exactly 0 next to any visible
            // key (every row sees at least one) - so is exp(-inf), and -inf keeps a lane that has seen ONLY masked keys
            // out of the statistics (its maximum would be -3.4e38 for bf16, and -max * log2e overflows)
            if (qi[n] >= L && s >= L && (s - L) > (qi[n] - L)) x[r] = -INFINITY;
            if (s >= S) x[r] = -INFINITY;
          }
        }
        const float mx = track ? max3_raw(max3_raw(x[0], x[1], x[2]), x[3], m[n]) : 0.f;
        if (track && __any(mx > m[n])) {                                      // rare once the maxima settle
          const float mn = fmaxf(m[n], mx);
          Z[n] = (m[n] == -INFINITY) ? 0.f : Z[n] * __builtin_amdgcn_exp2f((m[n] - mn) * L2E);
          m[n] = mn;
          mL[n] = (mn == -INFINITY) ? 0.f : -mn * L2E;
        }
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(x[r], L2E, mL[n]);
        Z[n] += (__builtin_amdgcn_exp2f(y[0]) + __builtin_amdgcn_exp2f(y[1])) +
                (__builtin_amdgcn_exp2f(y[2]) + __builtin_amdgcn_exp2f(y[3]));
      }
    }
    stage_store<KS>(stg, tiles[(t + 1) & 1], tid);                            // buffer last read in iteration t-1
    __syncthreads();
  };
  const int t_plain = (L < S ? L : S) / HT;                                   // tiles [0, t_plain) end at or before L
  float* rs = p.rowstat + (int64_t)bh * S;
  // Reference point of the exponentials (round 5): the lane's maximum over the FIRST tile of keys, then frozen.  softmax needs
  // some M_i with exp(x - M_i) inside the fp32 range, not the row maximum; tile 0 holds what a row of a real prompt attends to
  // first (the attention sinks) and is a sample of 64 keys otherwise, so the row maximum lies within a few units of it - and
  // anything up to +88 above it still sums without overflow.  Z >= 1 by construction (the sampled maximum itself contributes
  // 1), so only Z = inf / NaN sends a workgroup through the exact online-maximum loop again.  Round 4 took the point from a
  // norm bound |q_i| max_j |k_j| / sqrt(D) - 64 instead (a key-norm scan + 20 instructions per row): right for N(0,1) data,
  // but a single massive-activation key nobody attends to pushed every row out of its window and doubled the pass
  // (the advisor's finding); the sample costs the tracking body for one tile in 512 and no extra kernel.
#pragma unroll
  for (int n = 0; n < 4; ++n) { m[n] = -INFINITY; mL[n] = 0.f; Z[n] = 0.f; }
  auto finish = [&](bool check) -> bool {        // merge the four key groups of every query row; -> true when some row overflowed
    float cr[4];
    bool bad = false;
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
      bad |= !(zz <= 0x1p120f);                                              // inf or NaN (a row that sees no key at all does not exist)
      // c_row = -(m*log2e + log2 Z): pass 2 evaluates exp(x - m) / Z as exp2(x*log2e + c_row); independent of the reference point
      cr[n] = -(mm * L2E + __builtin_amdgcn_logf(zz));
    }
    if (check && __syncthreads_or(bad)) return true;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      if (lg == 0 && qi[n] < S) rs[qi[n]] = cr[n];
    return false;
  };
  // Round 6 (the advisor's finding on the round-5 form): a lane's reference point is the maximum of ITS 16 keys of tile 0, and
  // a row whose maximum lies more than ~88 above it overflows and sends the whole workgroup through the pass again - with
  // wide logit distributions (large-norm q / k: std >= ~35) that was nearly every workgroup, 2x the pass (19.3 vs 9.2 ms at
  // S = 32768 with q, k x 8).  A look at the first 16 keys now says how wide the rows are BEFORE the pass starts: 16 samples of
  // a row with standard deviation s span ~3.5 s, and the row's maximum over 32k keys sits ~2.5 s above the sample's; when
  // any row of the workgroup spans more than H2O_WIDE (40: s > ~11) the workgroup runs the pass with the TRACKED maximum (the
  // exact online form, +10 %) right away instead of freezing and repeating.  N(0,1)-like rows (span ~3.5-8, attention sinks
  // included) never take it; the overflow check and the repeat stay behind the frozen form as the safety net.  The look is 16
  // MFMAs per wave in its own scope: inside tile 0's body the same four minima cost 20 registers = one wave per SIMD.
  constexpr float H2O_WIDE = 40.f;
  bool wide = t_plain == 0;                                                   // the corner mask reaches tile 0: a prompt of a few keys
  if (!wide) {
    u32x4 kf[KS];
    read_frags<KS>(kf, tiles[0], 0, li, lg);
    f32x4 accs[4];
    mm16<T, KS>(accs, kf, qf);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float x[4];
      logits4<T>(accs[n], p, x);
      float hi = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), lw = fminf(fminf(x[0], x[1]), fminf(x[2], x[3]));
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) { hi = fmaxf(hi, __shfl_xor(hi, o, 64)); lw = fminf(lw, __shfl_xor(lw, o, 64)); }
      wide |= !(hi - lw <= H2O_WIDE);
    }
  }
  if (!__syncthreads_or(wide)) {
    tile_body(0, std::false_type{}, std::true_type{});
    for (int t = 1; t < t_plain; ++t) tile_body(t, std::false_type{}, std::false_type{});
    for (int t = t_plain > 1 ? t_plain : 1; t < ntiles; ++t) tile_body(t, std::true_type{}, std::false_type{});
    if (!finish(true)) return;
    // a frozen row overflowed after all: repeat with the exact online maximum, tile 0 staged again
    stage_load<KS>(stg, ks, 0);
    stage_store<KS>(stg, tiles[0], tid);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 4; ++n) { m[n] = -INFINITY; mL[n] = 0.f; Z[n] = 0.f; }
  }
  for (int t = 0; t < t_plain; ++t) tile_body(t, std::false_type{}, std::true_type{});
  for (int t = t_plain; t < ntiles; ++t) tile_body(t, std::true_type{}, std::true_type{});
  finish(false);
}

// Pass 2: per key column, sum over all query rows of round(exp(x - m) / Z).  Resident = 256 key columns,
// streamed = Q (+ the 64 row constants c_row of the tile).  No branch in the loop.
template <typename T, int KS>
__global__ __launch_bounds__(256) void h2o_colsum_kernel(H2OParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char h2o_smem[];               // 2 tiles + 2 x 64 row constants
  u32x4 (*tiles)[HT * 4 * KS] = reinterpret_cast<u32x4 (*)[HT * 4 * KS]>(h2o_smem);
  float (*stats)[HT] = reinterpret_cast<float (*)[HT]>(h2o_smem + (size_t)2 * HT * 4 * KS * 16);   // c_row of the 64 streamed query rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H, hk = h / p.G;
  const int S = p.S, L = S - p.w;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.qs_b + (int64_t)h * p.qs_h;
  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h;
  const float* rs = p.rowstat + (int64_t)bh * S;
  const int k0 = blockIdx.x * HWG + wave * HR;

  u32x4 kf[4][KS];
  int kj[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    kj[n] = k0 + n * 16 + li;
    load_frags<KS>(kf[n], kb, kj[n] < S ? kj[n] : S - 1, p.ks_s, lg);
  }
  // matrix-pipe column sums (see the header): every row of cacc[n] holds the sums of key columns kj[n]
  f32x4 cacc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) cacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = {Ones2<T>::v, Ones2<T>::v, Ones2<T>::v, Ones2<T>::v};
  uint32_t held[4][2];
  const float L2E2 = 1.44269504088896340736f;
  auto row_const = [&](int i) {                                // rows past S: exp2(x*log2e - inf) = 0 (clamped load + select: no branch)
    const float c = rs[i < S ? i : S - 1];
    return i < S ? c : -INFINITY;
  };

  const TileStream qs = make_stream<KS>(qb, p.qs_s, S, tid);
  Stager<KS> stg;
  stage_load<KS>(stg, qs, 0);
  float sreg = row_const(lane);                                // all four waves carry the same 64 constants: no divergent branch
  stage_store<KS>(stg, tiles[0], tid);
  stats[0][lane] = sreg;
  __syncthreads();
  const int ntiles = (S + HT - 1) / HT;
  for (int t = 0; t < ntiles; ++t) {
    const u32x4* cur = tiles[t & 1];
    const float* cst = stats[t & 1];
    stage_load<KS>(stg, qs, (t + 1) * HT);                     // past the end: zeros (and c_row = -inf)
    sreg = row_const((t + 1) * HT + lane);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      u32x4 qf[KS];
      read_frags<KS>(qf, cur, sub, li, lg);
      const f32x4 st = *reinterpret_cast<const f32x4*>(cst + sub * 16 + lg * 4);
      f32x4 accs[4];
      mm16<T, KS, true>(accs, qf, kf);                         // D[query][key]
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float x[4], e[4];
        logits4<T>(accs[n], p, x);                             // keys < L never touch the masked corner
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r], L2E2, st[r]));   // fp32 softmax (:553): exp(x - m) / Z
        const uint32_t p01 = round_pack2<T>(e[0], e[1]), p23 = round_pack2<T>(e[2], e[3]);              // .to(dtype)
        if ((sub & 1) == 0) {
          held[n][0] = p01; held[n][1] = p23;
        } else {                                               // this lane's 8 query rows of key column li; fp32 sum (:554)
          const u32x4 pb = {held[n][0], held[n][1], p01, p23};
          cacc[n] = Mfma2<T>::run(ones, pb, cacc[n]);
        }
      }
    }
    stage_store<KS>(stg, tiles[(t + 1) & 1], tid);
    stats[(t + 1) & 1][lane] = sreg;
    __syncthreads();
  }
  uint16_t* out = reinterpret_cast<uint16_t*>(p.scores) + (int64_t)bh * p.scores_stride;
#pragma unroll
  for (int n = 0; n < 4; ++n)                                  // the MFMA already summed the four lane groups
    if (lg == 0 && kj[n] < L) out[kj[n]] = Elem<T>::from_f32(cacc[n][0]);
}


// raw-buffer tile loads carry 32-bit offsets: 64 rows of the streamed matrix must span less than 4 GB
static bool strides_ok(const H2OParams& p) {
  return p.qs_s > 0 && p.ks_s > 0 && p.qs_s * 2 * 64 < (int64_t)0xffffffffll && p.ks_s * 2 * 64 < (int64_t)0xffffffffll;
}

hipError_t launch_h2o_stats(int dtype, const H2OParams& p, hipStream_t st) {
  if (!strides_ok(p)) return hipErrorInvalidValue;
  dim3 grid((p.S + HWG - 1) / HWG, p.B * p.H);
#define PKV_H2O_S(TT, KS)                                                                                             \
  do {                                                                                                                \
    const size_t lds_ = (size_t)2 * HT * 4 * KS * 16;                                                                 \
    if (lds_ >= 64 * 1024) {                                                                                          \
      hipError_t e_ = dyn_lds(reinterpret_cast<const void*>(h2o_stats_kernel<TT, KS>), lds_);                    \
      if (e_ != hipSuccess) return e_;                                                                                \
    }                                                                                                                 \
    PKV_KLAUNCH((h2o_stats_kernel<TT, KS>), grid, dim3(256), lds_, st, p);                                            \
  } while (0)
  if (dtype == 0) { if (p.D == 64) PKV_H2O_S(BF16, 2); else if (p.D == 256) PKV_H2O_S(BF16, 8); else PKV_H2O_S(BF16, 4); }
  else { if (p.D == 64) PKV_H2O_S(F16, 2); else if (p.D == 256) PKV_H2O_S(F16, 8); else PKV_H2O_S(F16, 4); }
#undef PKV_H2O_S
  return hipGetLastError();
}

hipError_t launch_h2o_colsum(int dtype, const H2OParams& p, hipStream_t st) {
  if (!strides_ok(p)) return hipErrorInvalidValue;
  const int L = p.S - p.w;
  dim3 grid((L + HWG - 1) / HWG, p.B * p.H);
#define PKV_H2O_C(TT, KS)                                                                                             \
  do {                                                                                                                \
    const size_t lds_ = (size_t)2 * HT * 4 * KS * 16 + 2 * HT * 4;                                                    \
    if (lds_ >= 64 * 1024) {                                                                                          \
      hipError_t e_ = dyn_lds(reinterpret_cast<const void*>(h2o_colsum_kernel<TT, KS>), lds_);                    \
      if (e_ != hipSuccess) return e_;                                                                                \
    }                                                                                                                 \
    PKV_KLAUNCH((h2o_colsum_kernel<TT, KS>), grid, dim3(256), lds_, st, p);                                           \
  } while (0)
  if (dtype == 0) { if (p.D == 64) PKV_H2O_C(BF16, 2); else if (p.D == 256) PKV_H2O_C(BF16, 8); else PKV_H2O_C(BF16, 4); }
  else { if (p.D == 64) PKV_H2O_C(F16, 2); else if (p.D == 256) PKV_H2O_C(F16, 8); else PKV_H2O_C(F16, 4); }
#undef PKV_H2O_C
  return hipGetLastError();
}

}  // namespace pkv
