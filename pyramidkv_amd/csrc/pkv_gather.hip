// pkv_gather.hip — K/V gather-compaction (gfx950).
//
//   gather_kernel   reference pyramidkv_utils.py:335,341-346 (dense [B,H,k+w,D]),
//                   :607-620 (StreamingLLM: identity indices), :733-757 (AdaKV/HeadKV flat var-len)
//   flatten_kernel  reference csrc/csrc/cuda_api.cu:11-53 (decode-time flat-cache append)
//
// Pure data movement, bit-exact.  Roofline: HBM.  Algorithmic bytes per call
//   4 * (k+w) * D * e * B * H     (K and V, read + write).
// A head-row is D*e = 256 B = 16 lanes x 16 B; a wavefront moves 4 rows per instruction, every lane
// keeps 8 independent 16-B loads in flight (4 rows x {K,V}); writes are fully sequential per head.
#include "pkv_common.hpp"
#include "pkv_kernels.hpp"

namespace pkv {

// Work decomposition: one workgroup = 16*RPT output rows x {K,V} of one (b,h); RPT rows per lane and tensor
// (RPT = 4 -> 64 rows, 8 independent 16-B loads per lane).  The grid is one-dimensional so that the block -> (head,
// row block) map can follow the hardware's round-robin block -> XCD placement (block b runs on XCD b % 8, observed, used
// for speed only): the indices of head bh were written by top-k workgroup bh, i.e. on XCD bh % 8, and the gather
// blocks of that head are placed on the same XCD, where the index lines are still in the L2 - the one dependent
// round trip before any row can move is then an L2 hit instead of a trip to the fabric.
// CH = 16-B chunks per row = head_dim / 8 (8, 16, 32 for head sizes 64, 128, 256; 64 = fp32 rows of head size 256): 256 / CH row slots per pass.
template <int RPT, int CH>
__global__ __launch_bounds__(256) void gather_kernel(GatherParams p) {
  constexpr int SLOTS = 256 / CH;
  constexpr int ROWS = SLOTS * RPT;
  constexpr int DH = CH * 8;           // head size in elements
  const int tid = threadIdx.x;
  const int chunk = tid % CH;          // 16-B chunk of the row
  const int slot = tid / CH;           // row slot
  const int BH = p.B * p.H;
  int bh, blk;
  if (p.xcd_map) {                     // BH % 8 == 0 (host-checked)
    const int id = blockIdx.x, x = id & 7, q = id >> 3, hpx = BH >> 3;
    blk = q / hpx;
    bh = (q - blk * hpx) * 8 + x;
  } else {
    bh = blockIdx.x / p.nblk;
    blk = blockIdx.x - bh * p.nblk;
  }
  const int b = bh / p.H;
  const int h = bh - b * p.H;
  const int hk = h / p.G;
  const int L = p.S - p.w;
  const int r_blk = blk * ROWS;
  const int32_t* ib = p.idx ? p.idx + (int64_t)bh * p.idx_stride : nullptr;

  // All loads are unconditional (invalid slots read row 0 and are simply not stored) so that the RPT
  // index loads and then the 2*RPT 16-B row loads are issued back to back: two dependent memory round
  // trips per workgroup instead of one per row.
  // Round 6: the index loads go out FIRST, before the head's own row count and output offset (flat var-len layout: head_k,
  // cu_rows) are even asked for - their addresses do not depend on those, so a flat gather pays two dependent round trips
  // like the dense one, not three.
  int src[RPT];
  bool ok[RPT];
  int gi[RPT];
  if (ib) {                                     // uniform branch; the loads inside are unconditional
    // never past the row of the index list: a capacity beyond the list length (a caller that sized the list by a guess and
    // checks the capacities afterwards) re-reads the row's last entry instead of running into the next row; entries at or
    // beyond the head's own count are loaded and never used
    const int last = (int)p.idx_stride - 1;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int r = r_blk + j * SLOTS + slot;
      gi[j] = __builtin_nontemporal_load(ib + (r < last ? r : last));
    }
  } else {
#pragma unroll
    for (int j = 0; j < RPT; ++j) gi[j] = r_blk + j * SLOTS + slot;
  }
  const int nsel = p.head_k ? p.head_k[bh] : p.nsel;
  const int nrows = nsel + p.w;
  if (r_blk >= nrows) return;
  const unsigned long long t_start = PKV_WGTRACE(p) ? wall_clock64() : 0ull;
  const int64_t out_row0 = p.cu_rows ? (int64_t)p.cu_rows[bh] : (int64_t)bh * nrows;

  const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.kptr) + (int64_t)b * p.ks_b + (int64_t)hk * p.ks_h + chunk * 8;
  const uint16_t* vb = reinterpret_cast<const uint16_t*>(p.vptr) + (int64_t)b * p.vs_b + (int64_t)hk * p.vs_h + chunk * 8;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = r_blk + j * SLOTS + slot;
    ok[j] = r < nrows && out_row0 + r < p.out_rows;          // out_rows: rows the output buffers hold (flat layout: a bound)
    // a selected index outside [0, L) (the reference's gather raises there) is clamped: never an out-of-bounds read
    const int g = min(max(gi[j], 0), L - 1);
    const int s = (r < nsel) ? g : L + (r - nsel);
    src[j] = ok[j] ? s : 0;
  }
  u32x4 kd[RPT], vd[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    kd[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (int64_t)src[j] * p.ks_s));
    vd[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (int64_t)src[j] * p.vs_s));
  }
  uint16_t* ko = reinterpret_cast<uint16_t*>(p.k_out) + chunk * 8;
  uint16_t* vo = reinterpret_cast<uint16_t*>(p.v_out) + chunk * 8;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (ok[j]) {
      const int64_t orow = out_row0 + r_blk + j * SLOTS + slot;
      __builtin_nontemporal_store(kd[j], reinterpret_cast<u32x4*>(ko + orow * DH));
      __builtin_nontemporal_store(vd[j], reinterpret_cast<u32x4*>(vo + orow * DH));
    }
  }
  if (PKV_WGTRACE(p) && tid == 0) {
    const size_t wg = 196608 + (size_t)blockIdx.x;
    if (wg < 262144) { PKV_WGTRACE(p)[2 * wg] = t_start; PKV_WGTRACE(p)[2 * wg + 1] = wall_clock64(); }
  }
}

hipError_t launch_gather(const GatherParams& p0, int max_rows, hipStream_t st) {
  GatherParams p = p0;
  // rows per lane: budgets of ~1000+ rows per head stream best with 8 (16 loads in flight per lane: 13.8 us against 14.6 us for
  // 67 MB at B = 1), small budgets are launch-latency-bound and finish earlier with many small workgroups (measured, profiles/)
  int rpt = p.rpt;
  if (rpt != 2 && rpt != 4 && rpt != 8 && rpt != 16) rpt = max_rows >= 1024 ? 8 : 2;
  const int ch = p.D / 8;
  if (ch != 16 && rpt == 16) rpt = 8;                       // the 16-row variant exists for 256-byte rows only
  const int rows = (256 / ch) * rpt;
  const int BH = p.B * p.H;
  p.nblk = (max_rows + rows - 1) / rows;
  if (BH % 8 != 0) p.xcd_map = 0;
  dim3 grid((unsigned)(p.nblk * BH));
#define PKV_G(R, C) PKV_KLAUNCH((gather_kernel<R, C>), grid, dim3(256), 0, st, p)
  if (ch == 16) {
    if (rpt == 2) PKV_G(2, 16); else if (rpt == 16) PKV_G(16, 16); else if (rpt == 8) PKV_G(8, 16); else PKV_G(4, 16);
  } else if (ch == 8) {
    if (rpt == 2) PKV_G(2, 8); else if (rpt == 8) PKV_G(8, 8); else PKV_G(4, 8);
  } else if (ch == 64) {                                    // 1024-byte rows: fp32 tensors at head size 256
    if (rpt == 2) PKV_G(2, 64); else if (rpt == 8) PKV_G(8, 64); else PKV_G(4, 64);
  } else {
    if (rpt == 2) PKV_G(2, 32); else if (rpt == 8) PKV_G(8, 32); else PKV_G(4, 32);
  }
#undef PKV_G
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Decode-time flat-cache append.  grid = (H, FL_SPLIT); 16-byte vector copies; one inserting block.
// ------------------------------------------------------------------------------------------------
constexpr int FL_SPLIT = 32;

__global__ __launch_bounds__(256) void flatten_kernel(FlattenParams p) {
  const int h = blockIdx.x;
  const int n = p.head_lens[h];
  const int64_t src_row = p.cu_klen[h];
  const int64_t dst_row = src_row + h;                 // cuda_api.cu:28
  const int64_t ins_row = (int64_t)p.cu_klen[h + 1] + h;  // cuda_api.cu:29
  const int64_t units = (int64_t)n * p.row_bytes / 16;
  const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.cache) + src_row * p.row_bytes);
  uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out) + dst_row * p.row_bytes);
  for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < units; i += (int64_t)FL_SPLIT * 256) d[i] = s[i];
  if (blockIdx.y == 0) {
    const int ru = p.row_bytes / 16;
    const uint4* st = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.state) + (int64_t)h * p.row_bytes);
    uint4* di = reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out) + ins_row * p.row_bytes);
    for (int i = threadIdx.x; i < ru; i += 256) di[i] = st[i];
  }
}

hipError_t launch_flatten(const FlattenParams& p, hipStream_t st) {
  hipLaunchKernelGGL(flatten_kernel, dim3(p.H, FL_SPLIT), dim3(256), 0, st, p);
  return hipGetLastError();
}

__global__ void debug_exp_kernel(const float* in, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = pkv_exp(in[i]);
}
hipError_t launch_debug_exp(const float* in, float* out, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(debug_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}

template <typename T> __global__ void debug_round_kernel(const float* in, uint16_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = Elem<T>::from_f32(in[i]);
}
hipError_t launch_debug_round(int dtype, const float* in, uint16_t* out, int64_t n, hipStream_t st) {
  if (dtype == 0) hipLaunchKernelGGL(debug_round_kernel<BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
  else hipLaunchKernelGGL(debug_round_kernel<F16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}

}  // namespace pkv
