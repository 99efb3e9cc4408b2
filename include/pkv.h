/* pkv.h — C ABI of libpkv: the MI355X (gfx950) prefill-time KV-cache eviction path.
 *
 * This is the drop-in boundary for the reference's
 *     pyramidkv/pyramidkv_utils.py  *KVCluster.update_kv        (Zefan-Cai/PyramidKV @ 2024-12-20)
 * The reference has no FFI for this path (it is eager PyTorch); each entry point below names the
 * reference lines it replaces.  Plain pointers and sizes only: no torch types, no pybind.
 * The only native symbol the reference does export is the decode-time flat-cache append
 *     csrc/csrc/cuda_api.cu:55-91  tiny_api_cuda.update_flatten_view
 * which pkv_update_flatten_view() replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the device that is current when the call is made;
 *   - every launch goes to `stream` (a hipStream_t passed as void*); no call synchronises the
 *     device or allocates user-visible memory; workspaces are caller-provided;
 *   - tensors are [B,H,S,D] with ELEMENT strides for b,h,s and D contiguous (stride 1);
 *     base pointers and strides must keep every row 16-byte aligned;
 *   - return value 0 = success, negative = pkv_status; pkv_strerror() names it.  Nothing aborts.
 */
#ifndef PKV_H
#define PKV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PKV_VERSION 201 /* 0.2.1: + pkv_runtime_reset, pkv_debug_scale_multiplier; pkv_ada_budget_rows validates its host
                           mirror (8-byte aligned uint64 [H], host_seq >= 0) like pkv_ada_select; no layout change.  0.2.0: pkv_desc starts with struct_size (the struct can grow without breaking hosts built against an
                           older header); pkv_ada_select's budget step is one launch; host mirrors are uint64 [H] of
                           self-validating words; fp32 at D = 256; 0.1.2: short Ada-SnapKV candidate lists */
/* libpkv.so is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table (plus nothing). */
#define PKV_API __attribute__((visibility("default")))

typedef void* pkv_stream_t; /* hipStream_t */

enum pkv_status {
  PKV_OK = 0,
  PKV_ERR_DTYPE = -1,       /* dtype not bf16 / fp16 / fp32 */
  PKV_ERR_SHAPE = -2,       /* D not in {64,128,256}, window/topk out of range, k > L ... */
  PKV_ERR_ALIGN = -3,       /* pointer or stride breaks 16-byte row alignment */
  PKV_ERR_WORKSPACE = -4,   /* workspace too small */
  PKV_ERR_UNSUPPORTED = -5, /* valid request outside the limits of this build */
  PKV_ERR_HIP = -6,         /* a HIP runtime call failed; see pkv_last_hip_error() */
  PKV_ERR_NULL = -7,
  PKV_ERR_COLLECTIVE = -8,  /* an RCCL call failed; see pkv_last_nccl_error() */
  PKV_ERR_ABI = -9          /* pkv_desc.struct_size is not a size this library knows (host built against another pkv.h) */
};

/* PKV_F32: pkv_score_window, pkv_score_h2o, pkv_topk(_ws), pkv_gather_compact, pkv_gather_streaming, pkv_gather_flat,
 * pkv_compress, pkv_compress_h2o, pkv_select, pkv_merge_compact, pkv_ada_budget_rows, pkv_ada_metadata and
 * pkv_update_flatten_view, D in {64,128,256} (256 since 0.2.0), topk <= 4096; every other entry point (pkv_sort_rows, pkv_ada_budget,
 * pkv_ada_budget_topm, pkv_ada_adaptive_lists, pkv_ada_select) answers PKV_ERR_DTYPE / PKV_ERR_UNSUPPORTED. */
enum pkv_dtype { PKV_BF16 = 0, PKV_F16 = 1, PKV_F32 = 2 };
enum pkv_pool { PKV_POOL_NONE = 0, PKV_POOL_AVG = 1, PKV_POOL_MAX = 2 };
enum pkv_reduce { PKV_REDUCE_SUM = 0, PKV_REDUCE_MEAN = 1 };
/* how A/sqrt(D) is evaluated (pyramidkv_utils.py:317): DIV = fp32 division (ATen CPU),
 * RCP = multiply by fp32 reciprocal (ATen GPU kernels for a host-scalar divisor). */
enum pkv_scale { PKV_SCALE_DIV = 0, PKV_SCALE_RCP = 1 };
/* Order of EQUAL scores in the selected rows (pyramidkv_utils.py:334: `tensor.topk` leaves it to the backend).
 * CANONICAL = (value descending, index ascending) = a stable descending sort - what PyTorch-ROCm's topk produces for k > 32.
 * ATEN_ROCM: for topk <= 32 additionally reproduce the order PyTorch-ROCm's topk leaves equal scores in (its k <= 32 results go
 * through an unstable 32-element bitonic network, ATen/native/hip/SortUtils.cuh bitonicSortKVInPlace): the cache rows of
 * PyramidKV's upper layers (k = 17..32) then sit in the order of a reference run on the same GPU.  Same token set either way. */
enum pkv_tie { PKV_TIE_CANONICAL = 0, PKV_TIE_ATEN_ROCM = 1 };

/* The descriptor GROWS at its end from version to version.  The caller stores sizeof(pkv_desc) of the header it was compiled
 * against in struct_size; the library reads exactly that many bytes - never more - and takes every field the caller's
 * struct does not have yet as 0 (each new field is defined so that 0 = the behaviour before it existed).  A size below
 * PKV_DESC_MIN_SIZE or above the library's own sizeof(pkv_desc) returns PKV_ERR_ABI from every entry point that takes a
 * descriptor (pkv_workspace_bytes / pkv_merge_workspace_bytes return 0). */
typedef struct pkv_desc {
  uint32_t struct_size; /* sizeof(pkv_desc) as the CALLER compiled it */
  int32_t dtype;        /* pkv_dtype of q,k,v and of every score buffer */
  int32_t B, H, S, D;   /* H = number of query heads; D = 64, 128 or 256 (all entry points) */
  int32_t kv_group;     /* 1: k,v have H heads (post-repeat_kv, the reference contract).
                           g>1: k,v have H/g heads (un-expanded GQA); head h reads kv head h/g */
  int32_t reserved0;    /* must be 0 (keeps the strides 8-byte aligned without implicit padding) */
  int64_t q_stride[3];  /* element strides of q for b,h,s */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int32_t window;       /* w = window_size, 1..128 (scoring entry points; kv_group * window <= 256) */
  int32_t pool_kind;    /* pkv_pool */
  int32_t pool_kernel;  /* odd, <= 17; padding = kernel/2, stride 1 (pyramidkv_utils.py:328-331) */
  int32_t reduce;       /* pkv_reduce: SUM for SnapKV/PyramidKV (:327), MEAN for AdaKV/HeadKV (:661) */
  int32_t scale_mode;   /* pkv_scale */
  int32_t topk;         /* k = past tokens kept per head, 1..S-window (host-resolved per layer) */
  int32_t tie_order;    /* a pkv_tie value, honoured by the selecting entry points pkv_compress, pkv_compress_h2o, pkv_select on
                           bf16 / fp16 scores; 0 = canonical; any other value than 0 / 1 returns PKV_ERR_SHAPE from every entry
                           point that validates the descriptor (since 0.1.1) */
  int32_t reserved1;    /* must be 0 (explicit tail padding: sizeof(pkv_desc) is a multiple of 8 without implicit padding) */
} pkv_desc;
#define PKV_DESC_MIN_SIZE 136u /* sizeof(pkv_desc) of 0.2.0, the first layout that carries struct_size */

PKV_API int pkv_version(void);
PKV_API const char* pkv_strerror(int status);
PKV_API int pkv_last_hip_error(void); /* hipError_t of the last PKV_ERR_HIP on this thread */
/* libpkv remembers, per (kernel, device), how much dynamic LDS it has asked the runtime for (hipFuncSetAttribute is then not
 * called on every launch).  That grant dies with the context: a host that calls hipDeviceReset() - or unloads and re-creates
 * the primary context in any other way - calls this afterwards; the attribute is then set again on the next launch. */
PKV_API int pkv_runtime_reset(void);

/* Bytes of scratch pkv_score_window / pkv_score_h2o / pkv_compress need for `d` (256-B aligned). */
PKV_API size_t pkv_workspace_bytes(const pkv_desc* d);

/* Observation-window score (pyramidkv_utils.py:317-333, AdaKV :649-672):
 *   logits = (Q[-w:] K^T)/sqrt(D) -> causal corner mask -> fp32 softmax over all S keys -> round
 *   -> sum/mean of the w rows over columns [0,S-w) -> round -> avg/max pool.
 * scores_out: [B*H rows][scores_stride] elements of d->dtype, columns [0, S-w) written. */
PKV_API int pkv_score_window(const pkv_desc* d, const void* q, const void* k, void* scores_out,
                     int64_t scores_stride, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* H2O score (pyramidkv_utils.py:544-554,561): all S query rows, SxS never materialised.
 * Same output layout as pkv_score_window; no pooling. */
PKV_API int pkv_score_h2o(const pkv_desc* d, const void* q, const void* k, void* scores_out,
                  int64_t scores_stride, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Top-k token selection (pyramidkv_utils.py:334, :238, :270, :562): for each of `rows` score rows
 * of length L pick the k largest, emitted in (value desc, index asc) order as int32.
 * scores: [rows][scores_stride] of dtype; idx_out: [rows][idx_stride] int32.
 * k_per_row (device int32[rows], may be NULL) overrides k per row (k = upper bound then). */
PKV_API int pkv_topk(int32_t dtype, int32_t rows, int32_t L, int32_t k, const void* scores,
             int64_t scores_stride, const int32_t* k_per_row, int32_t* idx_out, int64_t idx_stride,
             pkv_stream_t stream);

/* Rows longer than one workgroup's LDS (L > 57 344 keys; k <= 32 768): the same selection through per-segment
 * top-k + a top-k over the segment winners; needs pkv_topk_workspace_bytes() of 16-B aligned scratch (0 for rows
 * pkv_topk handles itself; k_per_row is not supported on long rows). */
PKV_API size_t pkv_topk_workspace_bytes(int32_t rows, int32_t L, int32_t k);
PKV_API int pkv_topk_ws(int32_t dtype, int32_t rows, int32_t L, int32_t k, const void* scores,
                int64_t scores_stride, const int32_t* k_per_row, int32_t* idx_out, int64_t idx_stride,
                void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Gather-compaction (pyramidkv_utils.py:335,341-346): K_out/V_out[b,h] = rows idx[b,h,0..k) of
 * K/V[b,h,:S-w] followed by the w window rows S-w..S-1.  Outputs are contiguous [B,H,k+w,D].
 * idx: int32 [B*H][idx_stride].  d->topk = k.  Uses d->k_stride/v_stride/kv_group. */
PKV_API int pkv_gather_compact(const pkv_desc* d, const void* k, const void* v, const int32_t* idx,
                       int64_t idx_stride, void* k_out, void* v_out, pkv_stream_t stream);

/* StreamingLLM (pyramidkv_utils.py:607-620): idx = 0..k-1 for every head; no scoring. */
PKV_API int pkv_gather_streaming(const pkv_desc* d, const void* k, const void* v, void* k_out, void* v_out,
                         pkv_stream_t stream);

/* Fused update_kv for SnapKV / PyramidKV (pyramidkv_utils.py:306-347, :197-283; k resolved by the
 * host per layer): score -> top-k -> gather on `stream`.  idx_out (int32 [B*H][d->topk]) may be NULL
 * (then it lives in ws). */
PKV_API int pkv_compress(const pkv_desc* d, const void* q, const void* k, const void* v, void* k_out,
                 void* v_out, int32_t* idx_out, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Fused update_kv for H2O (pyramidkv_utils.py:533-575). */
PKV_API int pkv_compress_h2o(const pkv_desc* d, const void* q, const void* k, const void* v, void* k_out,
                     void* v_out, int32_t* idx_out, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Selection only: the front half of pkv_compress / pkv_compress_h2o (score -> top-k), idx_out int32 [B*H][d->topk].
 * What a caller needs when something other than the plain gather follows (the merge below).  ws: pkv_workspace_bytes(d). */
PKV_API int pkv_select(const pkv_desc* d, const void* q, const void* k, int32_t h2o, int32_t* idx_out, void* ws, size_t ws_bytes,
               pkv_stream_t stream);

/* LOOK-M pivot merge (pyramidkv_utils.py:119-170 merge_kv(..., merge="pivot"), called from :242,:274,:338,:566,:611 when the
 * cluster's `merge` is set) instead of the plain gather: every position NO (batch, head) selected - the window positions
 * included (:128-133) - is merged into its most cosine-similar kept key (first maximum, :150-151) as (x + pivot)/2, and the
 * kept rows become the scatter-mean of what reached them (:158-162).  As in the reference k_out is ordered
 * [window, selected] (:146) and v_out [selected, window] (:148), and the value merge uses the key order's pivot numbers.
 * idx: int32 [B*H][idx_stride] (d->topk = k entries per row, from pkv_select / pkv_topk).  Outputs [B,H,k+w,D] contiguous.
 * ws: pkv_merge_workspace_bytes(d), 16-B aligned.  Rounding points: oracle/pkv_oracle.py merge_kv_explicit.
 * bf16 / fp16 / fp32 (since 0.1.2; D = 256 since 0.2.0) at D = 64 / 128 / 256, fp32 with topk <= 4096; S <= 393 216 and
 * k + window <= 65 535 (PKV_ERR_UNSUPPORTED beyond). */
PKV_API size_t pkv_merge_workspace_bytes(const pkv_desc* d);
PKV_API int pkv_merge_compact(const pkv_desc* d, const void* k, const void* v, const int32_t* idx, int64_t idx_stride,
                      void* k_out, void* v_out, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* ---- Ada-SnapKV / HeadKV (pyramidkv_utils.py:674-757, :808-878): flat var-len output ---- */

/* Per-row full descending sort (:706 attn_score.sort(descending=True)), ties index-ascending.
 * sorted_idx: int32 [rows][L]; sorted_val (may be NULL): dtype [rows][L].  L <= 32768. */
PKV_API int pkv_sort_rows(int32_t dtype, int32_t rows, int32_t L, const void* scores, int64_t scores_stride,
                  int32_t* sorted_idx, void* sorted_val, pkv_stream_t stream);

/* Head budgets (:709-719): optional normalisation, global top-(H*base) over the flattened sorted
 * scores, per-head counts, cap_h = round(count*(1-floor) + int(base*floor)).
 * sorted_val: dtype [H][L] (from pkv_sort_rows).  Writes int32 head_capacity[H] (device).
 * ws: >= 1024 + 2*H*256*4 bytes.  H <= 256.  B must be 1 (:724). */
PKV_API int pkv_ada_budget(int32_t dtype, int32_t H, int32_t L, const void* sorted_val, int32_t base_capacity,
                   double floor_ratio, int32_t normalize, int32_t* head_capacity, void* ws,
                   size_t ws_bytes, pkv_stream_t stream);

/* The same budgets WITHOUT a full sort: one head can receive at most H*base entries of the global top-(H*base), so the
 * first M = min(L, H*base) entries of every head's descending order decide everything.  top_idx: int32 [H][idx_stride],
 * row h = pkv_topk(scores row h, k = M) (canonical order); scores: dtype [H][scores_stride], the un-sorted rows of length
 * L (the sum over all scores of :710 is taken from them).  M >= min(L, H*base) is checked.  Same ws as pkv_ada_budget.
 * head_lens / cu_klen (both or neither; device int32 [H] / [H+1]): when given, the var-len metadata of pkv_ada_metadata
 * for `window` is written by the same launch. */
PKV_API int pkv_ada_budget_topm(int32_t dtype, int32_t H, int32_t L, int32_t M, const void* scores, int64_t scores_stride,
                        const int32_t* top_idx, int64_t idx_stride, int32_t base_capacity, double floor_ratio,
                        int32_t normalize, int32_t window, int32_t* head_capacity, int32_t* head_lens, int32_t* cu_klen,
                        void* ws, size_t ws_bytes, pkv_stream_t stream);

/* The same budgets from the UN-SORTED score rows alone (no sort, no top-M list): what :706-719 consume of the order is, per
 * head, the sum of its `base` largest scores (:710) and how many of its adaptive scores lie above / at the global threshold
 * (:712-717) - selections and counts, computed by histograms over the whole row.  For H*base > 4096 (budget 2048), where
 * min(L, H*base) is the whole row.  scores: dtype [H][scores_stride], rows of length L <= 65536; PKV_F32 rows (L <= 32768;
 * 32-bit keys, four radix levels; ws >= 1024 + 4*H*256*4 + 4*H*4 bytes) are accepted here and nowhere else among the budget
 * entry points - an fp32 `sum()` ratio has no bit-level target (ATen's order depends on the host), see DESIGN.md section 5.  cu_headlens (optional,
 * int32 [H]): inclusive prefix of head_lens (:687).  host_mirror (optional): device-visible PINNED HOST uint64 [H], 8-byte
 * aligned: word h = host_seq << 32 | ran_out << 31 | cap_h, one system-scope store per head (0.2.0; before: int32 [H+1] with a
 * fence and a flag word) - the host polls until every word carries host_seq instead of copy + stream synchronise.
 * The gather then needs, per head, its first cap_h entries of the canonical order: pkv_topk with k_per_row = head_capacity. */
PKV_API int pkv_ada_budget_rows(int32_t dtype, int32_t H, int32_t L, const void* scores, int64_t scores_stride, int32_t base_capacity,
                        double floor_ratio, int32_t normalize, int32_t window, int32_t* head_capacity, int32_t* head_lens,
                        int32_t* cu_klen, int32_t* cu_headlens, uint64_t* host_mirror, int32_t host_seq, void* ws, size_t ws_bytes,
                        pkv_stream_t stream);

/* Head-sharded Ada-SnapKV: the budget of :712-717 couples ALL heads, so the ranks exchange one
 * thing - every head's ADAPTIVE list (:709-711: sorted scores x sum(top base)/sum(all), model dtype), first M entries.
 * lists_out: dtype [H][M] for this rank's H heads; after the all-gather every rank calls pkv_ada_budget(sorted_val = the
 * gathered lists, L = M, normalize = 0).  M >= min(S-w, H_total*base) keeps that exact (see pkv_ada_budget_topm). */
PKV_API int pkv_ada_adaptive_lists(int32_t dtype, int32_t H, int32_t L, int32_t M, const void* scores, int64_t scores_stride,
                           const int32_t* top_idx, int64_t idx_stride, int32_t base_capacity, int32_t normalize,
                           void* lists_out, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Fused front half of AdaKVCluster.update_kv / HeadKVCluster.update_kv (:674-731 / :808-852) in ONE call: window score
 * (d->reduce = PKV_REDUCE_MEAN, :661) -> top-M indices of every head (d->topk = M, canonical order) -> head budgets +
 * var-len metadata.  Ada-SnapKV: given_capacity = NULL, M >= min(S-w, H*base); writes head_capacity, head_lens, cu_klen.
 * Short lists (0.1.2): with a host_mirror, base <= M < min(S-w, H*base) is accepted as well.  The budgets are then exact
 * unless some head's list runs out at the global threshold; the kernel reports that in bit 31 of every mirror word
 * (word h = host_seq << 32 | ran_out << 31 | cap_h; host_seq >= 0) and the caller repeats the call with the full M.
 * (A head of a real prompt takes a few base budgets, not H of them: max(4 x base, 512) entries per head instead of H x base
 * cut the selection from 21 to 14 us at S = 32768, H = 32, base = 120.)  With lists that fit one workgroup (H x M <= 45 056)
 * the budget step is ONE launch: the selection leaves every head's adaptive list behind as 16-bit keys and a single
 * workgroup selects the global threshold (5 us against 19 us for the three launches of 0.1.x).
 * HeadKV: given_capacity = device int32 [H] (host-derived, :855), M >= max capacity; writes head_lens, cu_klen only.
 * The host then reads the capacities back (klen_sum / max_seqlen_k are Python ints at the boundary, :685-686; the
 * reference has the same sync at :718) and calls pkv_gather_flat(top_idx, idx_stride = M, ...).  ws: pkv_workspace_bytes(d). */
PKV_API int pkv_ada_select(const pkv_desc* d, const void* q, const void* k, int32_t base_capacity, double floor_ratio,
                   int32_t normalize, const int32_t* given_capacity, int32_t* top_idx, int32_t* head_capacity,
                   int32_t* head_lens, int32_t* cu_klen, int32_t* cu_headlens /* [H] inclusive prefix (:687), may be NULL */,
                   uint64_t* host_mirror /* may be NULL: device-visible PINNED HOST uint64 [H]; the budget kernel stores
                                            host_seq << 32 | ran_out << 31 | cap_h in word h (whole 64-bit stores, no fence):
                                            the host polls until every word carries host_seq instead of copying +
                                            synchronising */,
                   int32_t host_seq, void* ws, size_t ws_bytes, pkv_stream_t stream);

/* Var-len metadata (:682-698) from head_capacity: head_lens[H] = cap_h + w, cu_klen[H+1]
 * (exclusive prefix + total).  All device int32. */
PKV_API int pkv_ada_metadata(int32_t H, int32_t window, const int32_t* head_capacity, int32_t* head_lens,
                     int32_t* cu_klen, pkv_stream_t stream);

/* Flat gather (:733-757): for head h rows sorted_idx[h][0..cap_h) then the window tail, written at
 * row cu_klen[h] of the flat [sum_h(cap_h+w), D] outputs.  B must be 1.  d->topk, if > 0, is an upper
 * bound on max_h cap_h (sizes the launch); 0 = unknown (S-w).  out_rows = rows k_out / v_out hold (> 0: nothing is
 * stored at or beyond it - the outputs may be sized by a bound before the capacities are known on the host; 0 = not
 * checked). */
PKV_API int pkv_gather_flat(const pkv_desc* d, const void* k, const void* v, const int32_t* sorted_idx,
                    int64_t idx_stride, const int32_t* head_capacity, const int32_t* cu_klen,
                    void* k_out, void* v_out, int64_t out_rows, pkv_stream_t stream);

/* Decode-time flat-cache append (csrc/csrc/cuda_api.cu:11-85 update_flatten_view): out has
 * origin_rows + H rows; head h: copy head_lens[h] rows from cache row cu_klen[h] to out row
 * cu_klen[h]+h, then state[h] at out row cu_klen[h+1]+h.  head_dim elements of dtype per row. */
PKV_API int pkv_update_flatten_view(int32_t dtype, int32_t H, int32_t head_dim, const void* cache,
                            const void* state, const int32_t* head_lens, const int32_t* cu_klen,
                            void* out, pkv_stream_t stream);

/* ---- multi-GPU: the one exchange step of the head-sharded path (the reference has no
 * collective: run_longbench.py:390 only places layers with device_map="auto") ----
 * Every rank holds idx_local int32 [B][H_local][k] (its heads, from pkv_compress); after the call every rank holds
 * idx_all int32 [B][nranks*H_local][k] (heads in rank order).  ONE ncclAllGather on `stream`; for B > 1 plus one small
 * regroup kernel (rank-major -> head-major), which needs ws >= nranks*B*H_local*k*4 bytes (ws may be NULL for B == 1).
 * nccl_comm is the caller's ncclComm_t (libpkv creates none and does not link RCCL: the entry points are taken from the
 * RCCL already loaded in the process - PyTorch's or the host's own; only a process without one gets the library named
 * by PKV_RCCL_LIB, else librccl.so.1: a process must hold ONE RCCL, two copies corrupt the heap at exit).
 * PyTorch does not expose its communicator, so the Python host issues the same collective through torch.distributed
 * (pyramidkv_amd/dist.py); a C/C++ host calls this. */
PKV_API int pkv_allgather_indices(void* nccl_comm, const int32_t* idx_local, int32_t* idx_all, int32_t B, int32_t H_local,
                          int32_t k, void* ws, size_t ws_bytes, pkv_stream_t stream);
PKV_API int pkv_last_nccl_error(void); /* ncclResult_t of the last PKV_ERR_COLLECTIVE on this thread */

/* ---- per-kernel device timing (hipEvent pairs on `stream`), used by bench.py ---- */
enum pkv_kernel_id {
  PKV_K_LOGITS = 0, PKV_K_FINALIZE = 1, PKV_K_TOPK = 2, PKV_K_GATHER = 3, PKV_K_H2O_STATS = 4,
  PKV_K_H2O_COLSUM = 5, PKV_K_SORT = 6, PKV_K_BUDGET = 7, PKV_K_COUNT = 8
};
PKV_API int pkv_prof_enable(int on);                                   /* returns previous state */
PKV_API int pkv_prof_read(double* ms_sum, int64_t* launches, int reset); /* arrays of PKV_K_COUNT; syncs events */

/* ---- debug / test hooks (not part of the drop-in surface) ----
 * The trace hooks and the PKV_LOGITS_ABLATE measurement knob exist only in the -DPKV_DEBUG build (libpkv_debug.so,
 * `make -C pyramidkv_amd/csrc debug`); the release library returns PKV_ERR_UNSUPPORTED for a non-NULL buffer and
 * never reads the environment variable. */
PKV_API int pkv_debug_build(void);                    /* 1 = this library was built with -DPKV_DEBUG */
PKV_API int pkv_debug_topk_trace(void* device_u64x8); /* NULL disables; row 0 of every later top-k launch stamps 7 phase clocks */
PKV_API int pkv_debug_wg_trace(void* device_u64);     /* NULL disables; 2*262144 u64: per-workgroup (start,end) wall clock, 100 MHz */
/* Test hooks of the RELEASE library (tests/test_gpu_parity.py): two element-wise kernels that expose device functions the
 * hot kernels inline - the accurate exp() of finalize_kernel and the fp32 -> dtype rounding of every rounding point - so
 * that the suite checks the shipped code, not a copy.  Nothing on the product path launches them (~2 KB of code). */
PKV_API int pkv_debug_exp(const float* in, float* out, int64_t n, pkv_stream_t stream);
PKV_API int pkv_debug_round(int32_t dtype, const float* in, void* out, int64_t n, pkv_stream_t stream);
/* Host-only: the fp32 constant the bf16 / fp16 kernels multiply a logit by in place of "/ math.sqrt(head_dim)"
 * (pyramidkv_utils.py:317) for this dtype, head size and pkv_scale mode; tests/test_abi_and_host.py checks, over every finite
 * 16-bit input, that the multiply reproduces ATen's division ("div") / reciprocal multiply ("rcp") bit for bit. */
PKV_API float pkv_debug_scale_multiplier(int32_t dtype, int32_t D, int32_t scale_mode);

#ifdef __cplusplus
}
#endif
#endif /* PKV_H */
