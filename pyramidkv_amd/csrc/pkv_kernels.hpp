// pkv_kernels.hpp — kernel parameter blocks and launch prototypes (internal to libpkv).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// records the hipError_t pkv_last_hip_error() reports for this thread (defined in pkv_api.hip; internal to libpkv)
extern "C" __attribute__((visibility("hidden"))) void pkv_set_last_hip_error(int e);

namespace pkv {

// Per-kernel timing (pkv_prof_*): while a single-kernel profiling scope is open the API layer points g_kev at a pair of
// events and the launcher attaches them to the dispatch itself (hipExtLaunchKernelGGL): they carry the kernel's own begin
// and end timestamps, the same interval rocprofv3 reports, with no extra barrier packets in the stream.
struct KernelEvents { hipEvent_t start, stop; bool used; };
extern thread_local KernelEvents* g_kev;
#define PKV_KLAUNCH(kernel, grid, block, lds, st, ...)                                                              \
  do {                                                                                                              \
    if (pkv::g_kev) {                                                                                               \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, pkv::g_kev->start, pkv::g_kev->stop, 0, __VA_ARGS__);     \
      pkv::g_kev->used = true;                                                                                      \
    } else {                                                                                                        \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                                \
    }                                                                                                               \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a (kernel, device) pair needs MORE dynamic LDS than it was last
// granted - not on every launch (one runtime call per update_kv saved for top-k at S >= 16k; bench.py `host_us`).
// The grant belongs to the loaded code object: a host that calls hipDeviceReset() (or otherwise re-creates the context) must
// call pkv_runtime_reset() afterwards (include/pkv.h), which empties this table - the attribute is then set again on the next
// launch.  A table that is full (64 kernels x devices) stops caching: the attribute is then simply set on every launch.
struct DynLdsRec { const void* fn; int dev; size_t have; };
struct DynLdsTable { DynLdsRec recs[64]; int n; unsigned epoch; };
extern unsigned g_dyn_lds_epoch;                       // bumped by pkv_runtime_reset(): every thread's table is stale
inline hipError_t dyn_lds(const void* fn, size_t bytes) {
  static thread_local DynLdsTable t = {{}, 0, 0};
  if (t.epoch != g_dyn_lds_epoch) { t.n = 0; t.epoch = g_dyn_lds_epoch; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  DynLdsRec* hit = nullptr;
  for (int i = 0; i < t.n; ++i)
    if (t.recs[i].fn == fn && t.recs[i].dev == dev) { hit = &t.recs[i]; break; }
  if (hit && hit->have >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (hit) hit->have = bytes;
  else if (t.n < 64) t.recs[t.n++] = {fn, dev, bytes};
  return hipSuccess;
}

// Measurement hooks (phase stamps, per-workgroup wall clocks, ablations that produce WRONG results) exist only in the
// -DPKV_DEBUG build (make debug -> libpkv_debug.so, loaded by tools/ through PKV_LIB); in the release library the
// accessors below are compile-time constants, so the hot loops carry no trace/ablation branches at all.
#ifdef PKV_DEBUG
#define PKV_ABLATE(p) ((p).ablate)
#define PKV_WGTRACE(p) ((p).wgtrace)
#define PKV_TRACE(p) ((p).trace)
#else
#define PKV_ABLATE(p) 0
#define PKV_WGTRACE(p) (static_cast<unsigned long long*>(nullptr))
#define PKV_TRACE(p) (static_cast<unsigned long long*>(nullptr))
#endif

struct LogitsParams {
  unsigned long long* wgtrace;   // debug: per-workgroup (start,end) wall clock, may be null
  const void* q;
  const void* k;
  void* logits;      // [B*H*w][Sp] model dtype
  float2* partial;   // [B*H*w][nT] (tile row max, tile sum exp)
  int B, H, S, w, G; // G = kv_group
  int D;             // head size: 64, 128 or 256 (logits2_kernel: 128 only)
  int Sp, nT;
  int tile;          // keys per workgroup: 128 or 256 (nT = ceil(S / tile))
  int nt;            // nontemporal K loads
  int nst;           // logits2_kernel: 128-key stages per workgroup (nT = chunks per head)
  int ablate;        // 0 = full kernel; 1..3 = measurement-only ablations (PKV_LOGITS_ABLATE)
  int fexp;          // logits2_kernel: hardware exp2 for the partial (max, sum exp) statistics
  int st_mode;       // logits2_kernel: logits stores 0 = plain, 1 = nontemporal, 2 = write-through (sc1)
  uint32_t logits_bytes;   // extent of the logits buffer (raw-buffer stores)
  int64_t qs_b, qs_h, qs_s;
  int64_t ks_b, ks_h, ks_s;
  int scale_mode;
  float sqrt_d, rcp_sqrt_d;
};

struct FinalizeParams {
  const void* logits;
  const float2* partial;
  void* scores;          // [B*H][scores_stride]
  int64_t scores_stride;
  int B, H, S, w, Sp, nT;
  int pool_kind, pool_kernel, reduce;
  void* cmax;            // [B*H][cmax_stride] max pooled score of every 8-position chunk (may be null)
  int64_t cmax_stride;
  unsigned long long* trace;   // debug: phase timestamps of block (0,0) (may be null)
  unsigned long long* wgtrace;
  // Ada-SnapKV (round 5): the sum over ALL pooled scores of a row (:710 `attn_score.sum(dim=-1)`) as one fp64 partial per
  // workgroup, [B*H][rowsum_np] with rowsum_np = the grid's x dimension (1024 positions each); added in workgroup order by the
  // budget kernel (a fixed order: run-to-run identical).  May be null.
  double* rowsum_part;
  int rowsum_np;
};

struct TopkParams {
  const void* scores;    // [rows][scores_stride] model dtype
  int64_t scores_stride;
  int L, k;
  const int32_t* k_per_row;
  const void* cmax;      // optional: per-chunk (8 scores) maxima written by finalize_kernel, [rows][cmax_stride]
  int64_t cmax_stride;
  int32_t* idx_out;      // [rows][idx_stride]
  int64_t idx_stride;
  unsigned long long* trace;   // debug: phase timestamps of row 0 (may be null)
  unsigned long long* wgtrace;
  int Lw;                // keys per wave (multiple of 512)
  int kpad;              // words reserved for the selection list (see topk_lds_bytes)
  int dual;              // second 32 KB counter / radix scratch region present in LDS
  int algo;              // small-k fast path: 1 = one-level histogram + bucket counting sort, 0 = two-level select + radix ordering
  int nseg, seg_len;     // long rows: workgroup r handles segment r % nseg (seg_len keys) of row r / nseg and writes
                         // row-global indices to idx_out row r; L stays the full row length.  nseg <= 1: off
  // Ada-SnapKV (round 5, topk_kernel<T, true>): the selection also hands over every head's ADAPTIVE list (:706-711) as
  // order-preserving 16-bit keys in output order: the winners' raw scores, times the head's ratio sum(top base) / sum(all)
  // (:710; the row total comes from finalize_kernel's per-workgroup fp64 partials), rounded to the model dtype (:711).  The
  // budget step then only has to find the global threshold over those keys.
  void* list_out;            // [rows][list_stride] uint16 keys, or null
  int64_t list_stride;
  const double* rowsum_part; // [rows][rowsum_np] finalize_kernel's partial sums of the row (normalize only)
  int rowsum_np;
  int ada_base;              // base capacity: the ratio's numerator sums the first ada_base entries of the list
  int ada_normalize;
};

struct SortParams {
  const void* scores;
  int64_t scores_stride;
  int L, n;              // n = power of two >= L
  int32_t* sorted_idx;   // [rows][L]
  void* sorted_val;      // [rows][L] or null
  unsigned long long* trace;   // debug: phase stamps of row 0 (may be null)
};

struct GatherParams {
  unsigned long long* wgtrace;   // debug: per-workgroup (start,end) wall clock, may be null
  const void* kptr;
  const void* vptr;
  void* k_out;
  void* v_out;
  const int32_t* idx;        // [B*H][idx_stride] or null (identity: StreamingLLM)
  int64_t idx_stride;
  const int32_t* head_k;     // per-(b,h) selected count (flat layout) or null -> uniform k
  const int32_t* cu_rows;    // per-(b,h) first output row (flat layout) or null -> bh*(k+w)
  int B, H, S, w, nsel, G;   // nsel = uniform selected-row count k
  int D;                     // head size: 64, 128 or 256 (a row is D*2 bytes = D/8 lanes x 16 B)
  int64_t ks_b, ks_h, ks_s;
  int64_t vs_b, vs_h, vs_s;
  int rpt;                   // rows per lane and tensor: 2, 4 or 8 (16*rpt rows per workgroup)
  int xcd_map;               // place the blocks of head bh on XCD bh % 8 (where top-k wrote its indices)
  int nblk;                  // row blocks per head (set by launch_gather)
  int64_t out_rows;          // rows the output buffers hold: nothing is stored at or beyond it
};

struct BudgetParams {
  const void* sorted_val;    // [H][L] descending, or null when the list is given as (sorted_idx, scores)
  const int32_t* sorted_idx; // [H][idx_stride]: the first L entries of every head's descending order (top-L indices), or null
  int64_t idx_stride;
  const void* scores;        // [H][scores_stride] un-sorted score rows of length Lrow (with sorted_idx)
  int64_t scores_stride;
  int Lrow;
  int H, L, base;            // L = entries per head in the sorted list
  float one_minus_floor;     // (float)(1.0 - floor_ratio), the fp32 scalar ATen multiplies by
  int floor_capacity;        // int(base * floor_ratio)
  int normalize;
  int32_t* head_capacity;    // [H]
  int window;                // with head_lens_out / cu_klen_out: the var-len metadata of :682-691 from the same launch
  int32_t* head_lens_out;    // [H] or null
  int32_t* cu_klen_out;      // [H+1] or null
  int32_t* cu_headlens_out;  // [H] inclusive prefix (:687) or null
  void* adaptive_out;        // optional: dtype [H][L] - emit every head's adaptive list (:711) and stop (head-sharded exchange)
  unsigned long long* host_mirror;   // optional: device-visible PINNED HOST memory, uint64 [H]: word h = host_seq << 32 | ran_out << 31 | cap_h
  int32_t host_seq;
  int short_list;            // the lists are SHORTER than min(L, H*base): bit 31 of the mirror words reports a head whose list ran out
  int unsorted;              // 1: `scores` are the un-sorted rows [H][scores_stride] of length Lrow and no list is given
  void* list_ws;             // optional: H * roundup(L,8) * 2 bytes, 16-B aligned - the looked-up lists travel through it
  void* ws;                  // 1024 B ratios + 2 * H*256 int32
  unsigned long long* trace; // debug: phase timestamps of the one-launch budget kernel (may be null)
};

struct MergeParams {         // LOOK-M pivot merge (pkv_merge.hip)
  const void* kptr;
  const void* vptr;
  const int32_t* idx;        // [B*H][idx_stride] selected past tokens (top-k order)
  int64_t idx_stride;
  void* k_out;               // [B,H,k+w,D]  rows ordered [window, selected]
  void* v_out;               // [B,H,k+w,D]  rows ordered [selected, window]
  int B, H, S, w, k, G;
  int64_t ks_b, ks_h, ks_s;
  int64_t vs_b, vs_h, vs_s;
  uint8_t* mask;             // [S] 1 = selected by some (b,h)
  int32_t* ndrop;            // number of dropped positions
  int32_t* drop;             // [S] dropped positions, ascending
  int32_t* pivot;            // [B*H][S] kept-row number (KEY order) every dropped row merges into
  void* tn;                  // [B*H][ntp][D] unit-norm kept keys
  int ntp;
  int D;                     // head size: 64, 128 or 256
  int32_t* kept_bad;         // [B*H] != 0: a kept key of the head has no finite unit-norm form (zeroed with the mask)
  int32_t* bstart;           // [B*H][k+w+1] first entry of every kept row's group in blist (last = number of dropped rows)
  int32_t* blist;            // [B*H][S] dropped positions grouped by the kept row they merge into
};

struct FlattenParams {
  const void* cache;
  const void* state;
  const int32_t* head_lens;
  const int32_t* cu_klen;
  void* out;
  int H, row_bytes;
};

struct H2OParams {
  const void* q;
  const void* k;
  float* rowstat;     // [B*H][S] c_row = -log2 sum_j exp(x_ij) of every query row
  void* scores;
  int64_t scores_stride;
  int B, H, S, w, G;
  int D;              // head size: 64, 128 or 256
  int64_t qs_b, qs_h, qs_s;
  int64_t ks_b, ks_h, ks_s;
  int scale_mode;
  float sqrt_d, rcp_sqrt_d;
};

hipError_t launch_logits(int dtype, const LogitsParams& p, hipStream_t st);
hipError_t launch_logits2(int dtype, const LogitsParams& p, hipStream_t st);
hipError_t launch_finalize(int dtype, const FinalizeParams& p, hipStream_t st);
size_t topk_lds_bytes(int L, int k, int* Lw_out, int* kpad_out);
hipError_t launch_topk(int dtype, int rows, const TopkParams& p, size_t lds, hipStream_t st);
// long-row merge helpers: candidate scores of the per-segment winners, and the final index look-up
hipError_t launch_topk_merge_prep(int dtype, int rows, int L, int k, int nseg, int seg_len, const void* scores, int64_t scores_stride,
                                  const int32_t* cand_idx, void* cand_score, int64_t cand_stride, hipStream_t st);
hipError_t launch_topk_merge_finish(int rows, int k, const int32_t* cand_idx, int64_t cand_stride, const int32_t* pos, int32_t* idx_out,
                                    int64_t idx_stride, hipStream_t st);
// PKV_TIE_ATEN_ROCM: rows of k <= 32 canonical indices -> the order PyTorch-ROCm's topk gives them (16-bit scores)
hipError_t launch_aten_small_order(int dtype, int rows, int k, const void* scores, int64_t scores_stride, int32_t* idx, int64_t idx_stride,
                                   hipStream_t st);
hipError_t launch_sort_rows(int dtype, int rows, const SortParams& p, hipStream_t st);
hipError_t launch_gather(const GatherParams& p, int max_rows, hipStream_t st);
hipError_t launch_budget(int dtype, const BudgetParams& p, hipStream_t st);
// Ada-SnapKV budgets + metadata in ONE single-workgroup launch from the adaptive key lists topk_kernel<T, true> left behind
// (p.L entries per head at list[h * Lpad]); ada_fused_fits() = the lists of all heads fit one workgroup's LDS next to the counters
bool ada_fused_fits(int H, int M);
hipError_t launch_ada_fused(const BudgetParams& p, const void* list, int Lpad, hipStream_t st);
int finalize_blocks(int S, int w, int BH);     // workgroups per (b, h) row of finalize_kernel = row-sum partials per head (BH = B * H rows in the launch)
hipError_t launch_ada_final(const BudgetParams& p, int32_t* cum_hi, int32_t* cum_lo, const int32_t* above_hi, hipStream_t st);
hipError_t launch_budget_f32(const BudgetParams& p, hipStream_t st);      // fp32 score rows (pkv_f32.hip), ws: 1024 + 4*H*256*4 + 4*H*4 bytes
int budget_f32_max_row();
hipError_t launch_ada_metadata(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen, hipStream_t st,
                               int32_t* cu_headlens = nullptr);
hipError_t launch_flatten(const FlattenParams& p, hipStream_t st);
hipError_t launch_merge(int dtype, const MergeParams& p, hipStream_t st);
size_t merge_max_seq();      // longest sequence the merge takes (LDS bitmap of the scatter kernel)
hipError_t launch_debug_exp(const float* in, float* out, int64_t n, hipStream_t st);
hipError_t launch_debug_round(int dtype, const float* in, uint16_t* out, int64_t n, hipStream_t st);
// fp32 tensors (pkv_f32.hip)
hipError_t launch_logits_f32(const LogitsParams& p, hipStream_t st);
hipError_t launch_finalize_f32(const FinalizeParams& p, hipStream_t st);
hipError_t launch_topk_f32(int rows, const TopkParams& p, hipStream_t st);
int topk_f32_max_k();
hipError_t launch_h2o_f32(const H2OParams& p, hipStream_t st);            // fp32 tensors (pkv_f32.hip): rowstat = float2 (max, 1/sum) per row
hipError_t launch_h2o_stats(int dtype, const H2OParams& p, hipStream_t st);
hipError_t launch_h2o_colsum(int dtype, const H2OParams& p, hipStream_t st);

}  // namespace pkv
