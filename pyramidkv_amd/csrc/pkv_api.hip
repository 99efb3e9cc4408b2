// pkv_api.hip — the C ABI of libpkv (see include/pkv.h): validation, workspace carving, launches,
// optional per-kernel hipEvent timing.  No torch, no pybind; nothing here allocates device memory or
// synchronises the device (pkv_prof_read waits on its own events only).
#include "../../include/pkv.h"
#include "pkv_kernels.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>

namespace pkv { thread_local KernelEvents* g_kev = nullptr; unsigned g_dyn_lds_epoch = 1; }
using namespace pkv;

namespace {

thread_local int g_last_hip = 0;
unsigned long long* g_topk_trace = nullptr;   // debug hook, see pkv_debug_topk_trace
unsigned long long* g_wg_trace = nullptr;     // debug hook, see pkv_debug_wg_trace (2*262144 u64)

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
// logits kernel shape: keys per workgroup (128 = 32 keys/wave, ~7 workgroups per CU; 256 = 64 keys/wave)
// and nontemporal K loads.  Defaults are the measured best (profiles/); env vars exist for A/B runs.
int logits_tile() { static int t = env_int("PKV_LOGITS_TILE", 256) == 128 ? 128 : 256; return t; }
int logits_nt() { static int t = env_int("PKV_LOGITS_NT", 0); return t; }
// pipelined logits kernel (logits2_kernel): 1 = use it when the column count allows, 0 = one-tile-per-workgroup kernel
int logits_v2() { static int t = env_int("PKV_LOGITS_V2", 1); return t; }
int logits_v2_nt() { static int t = env_int("PKV_LOGITS_NT", 1); return t; }
int logits_v2_wgs() { static int t = env_int("PKV_LOGITS_V2_WGS", 0); return t; }   // target workgroup count, 0 = by column count (below)
int logits_fexp() { static int t = env_int("PKV_LOGITS_FEXP", 1); return t; }       // hardware exp2 in the partial statistics
int logits_store() { static int t = env_int("PKV_LOGITS_ST", 2); return t; }        // 0 plain, 1 nontemporal, 2 write-through
#ifdef PKV_DEBUG
int logits_ablate() { static int t = env_int("PKV_LOGITS_ABLATE", 0); return t; }   // measurement only: wrong results (debug build)
#else
constexpr int logits_ablate() { return 0; }   // the release library never reads PKV_LOGITS_ABLATE
#endif

// small-k selection algorithm of topk_kernel (identical results): 1 = one-level histogram + bucket counting sort
int topk_algo() { static int t = env_int("PKV_TOPK_ALGO", 1); return t; }
int topk_cmax() { static int t = env_int("PKV_TOPK_CMAX", 1); return t; }     // chunk maxima from finalize_kernel feed the top-k prefilter
int ada_fused() { static int t = env_int("PKV_ADA_FUSED", 1); return t; }     // 0: the three-launch budget step of round 4 (A/B runs)
// gather shape (both settings give identical results; defaults are the measured best, env vars exist for A/B runs)
int gather_rpt() { static int t = env_int("PKV_GATHER_RPT", 0); return t; }   // 0 = by size (launch_gather)
int gather_xcd() { static int t = env_int("PKV_GATHER_XCD", 0); return t; }

inline int hip_fail(hipError_t e) { g_last_hip = (int)e; return PKV_ERR_HIP; }

// the constant the 16-bit kernels multiply a logit by instead of dividing it by fp32(sqrt(head_dim)) - pkv_common.hpp scale_logit
float scale_multiplier(int dtype, int D, int scale_mode) {
  const float c = (float)sqrt((double)D);                  // math.sqrt(head_dim), cast to the fp32 opmath type
  float rc = 1.0f / c;                                     // ATen GPU path: a * (1.0f / b)
  if (dtype == PKV_F16 && scale_mode == PKV_SCALE_DIV && D == 128) rc = nextafterf(rc, 1.0f);   // == the fp32 division for every finite fp16 input
  return rc;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

// ---- per-kernel timing -----------------------------------------------------------------------
struct ProfRec { int id; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_pending;
std::vector<hipEvent_t> g_free_events;
double g_ms[PKV_K_COUNT] = {0};
int64_t g_n[PKV_K_COUNT] = {0};

hipEvent_t get_event() {
  if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

// ext = the scope holds exactly one kernel launched through PKV_KLAUNCH: its events ride on the dispatch (kernel begin /
// end).  Otherwise (several launches per scope) a pair of hipEventRecords brackets the scope.
struct ProfScope {
  bool on, ext; int id; hipStream_t st; hipEvent_t a, b; KernelEvents kev;
  ProfScope(int id_, hipStream_t st_, bool ext_ = false) : on(g_prof_on), ext(ext_), id(id_), st(st_) {
    if (on) {
      { std::lock_guard<std::mutex> lk(g_prof_mu); a = get_event(); b = get_event(); }
      if (ext) { kev = {a, b, false}; g_kev = &kev; }
      else (void)hipEventRecord(a, st);
    }
  }
  ~ProfScope() {
    if (on) {
      bool ok = true;
      if (ext) { g_kev = nullptr; ok = kev.used; }
      else (void)hipEventRecord(b, st);
      std::lock_guard<std::mutex> lk(g_prof_mu);
      if (ok) g_pending.push_back({id, a, b});
      else { g_free_events.push_back(a); g_free_events.push_back(b); }
    }
  }
};

// ---- validation ------------------------------------------------------------------------------
static_assert(sizeof(pkv_desc) == 136 && sizeof(pkv_desc) >= PKV_DESC_MIN_SIZE, "pkv_desc layout changed: bump PKV_VERSION, keep PKV_DESC_MIN_SIZE");

// The caller's descriptor -> the library's own struct: exactly `struct_size` bytes are read (the size of the struct the HOST
// was compiled against), fields the host does not have yet are 0.  Every entry point works on the copy.
int load_desc(const pkv_desc* in, pkv_desc* out) {
  if (!in) return PKV_ERR_NULL;
  const uint32_t n = in->struct_size;
  if (n < PKV_DESC_MIN_SIZE || n > sizeof(pkv_desc) || (n & 3u)) return PKV_ERR_ABI;
  memset(out, 0, sizeof(pkv_desc));
  memcpy(out, in, n);
  out->struct_size = (uint32_t)sizeof(pkv_desc);
  if (out->reserved0 != 0 || out->reserved1 != 0) return PKV_ERR_ABI;
  return PKV_OK;
}
#define PKV_LOAD_DESC(d, on_fail)                                   \
  pkv_desc d##_own;                                                 \
  {                                                                 \
    const int rc_ld = load_desc(d, &d##_own);                       \
    if (rc_ld) return on_fail;                                      \
  }                                                                 \
  d = &d##_own
// f32_ok: the entry point has an fp32 path (window scores, top-k, dense / streaming gather); every other one answers
// PKV_ERR_UNSUPPORTED for fp32 tensors
int check_desc(const pkv_desc* d, bool need_topk, bool scoring = true, bool f32_ok = false) {
  if (!d) return PKV_ERR_NULL;
  if (d->dtype != PKV_BF16 && d->dtype != PKV_F16 && d->dtype != PKV_F32) return PKV_ERR_DTYPE;
  if (d->D != 64 && d->D != 128 && d->D != 256) return PKV_ERR_SHAPE;
  if (d->dtype == PKV_F32 && !f32_ok) return PKV_ERR_UNSUPPORTED;
  if (d->B < 1 || d->H < 1 || d->S < 2) return PKV_ERR_SHAPE;
  if (d->kv_group < 1 || d->H % d->kv_group) return PKV_ERR_SHAPE;
  if (d->window < 1 || d->window >= d->S) return PKV_ERR_SHAPE;
  if (scoring && (d->window > 128 || d->kv_group * d->window > 256)) return PKV_ERR_UNSUPPORTED;
  if (need_topk && (d->topk < 1 || d->topk > d->S - d->window)) return PKV_ERR_SHAPE;
  if (d->pool_kind < 0 || d->pool_kind > 2) return PKV_ERR_SHAPE;
  if (d->tie_order != PKV_TIE_CANONICAL && d->tie_order != PKV_TIE_ATEN_ROCM) return PKV_ERR_SHAPE;
  if (d->pool_kind != PKV_POOL_NONE) {
    if (d->pool_kernel < 1 || !(d->pool_kernel & 1)) return PKV_ERR_SHAPE;
    if (d->pool_kernel > 17) return PKV_ERR_UNSUPPORTED;
  }
  const int64_t amask = d->dtype == PKV_F32 ? 3 : 7;                       // rows start on 16-byte boundaries
  for (int i = 0; i < 3; ++i)
    if ((d->q_stride[i] & amask) || (d->k_stride[i] & amask) || (d->v_stride[i] & amask)) return PKV_ERR_ALIGN;
  if ((int64_t)d->B * d->H > 65535) return PKV_ERR_UNSUPPORTED;
  if (d->dtype == PKV_F32 && need_topk && d->topk > topk_f32_max_k()) return PKV_ERR_UNSUPPORTED;
  return PKV_OK;
}

size_t topk_tmp_bytes(int rows, int L, int k);

struct WsLayout {
  int Sp, nT, Lp;
  size_t off_logits, off_partial, off_scores, off_idx, off_cmax, off_tk, tk_bytes, off_ada, off_ada_list, off_rowstat, total;
};

WsLayout ws_layout(const pkv_desc* d) {
  WsLayout w;
  w.nT = (d->S + logits_tile() - 1) / logits_tile();
  w.Sp = (d->S + 255) / 256 * 256;
  w.Lp = (int)align_up((size_t)(d->S - d->window), 8);
  const size_t rows = (size_t)d->B * d->H * d->window;
  const size_t es = d->dtype == PKV_F32 ? 4 : 2;                            // bytes per logit / score
  if (d->dtype == PKV_F32) w.nT = (d->S + 63) / 64;                         // logits_f32_kernel: 64 keys per workgroup
  size_t o = 0;
  w.off_logits = o;  o = align_up(o + rows * w.Sp * es, 256);
  w.off_partial = o; o = align_up(o + rows * (size_t)std::max(w.nT, (d->S + 127) / 128) * sizeof(float2), 256);
  w.off_scores = o;  o = align_up(o + (size_t)d->B * d->H * w.Lp * es, 256);
  w.off_idx = o;     o = align_up(o + (size_t)d->B * d->H * (d->topk > 0 ? d->topk : 1) * 4, 256);
  w.off_cmax = o;    o = align_up(o + (size_t)d->B * d->H * (w.Lp / 8) * 2, 256);
  w.tk_bytes = d->topk > 0 ? topk_tmp_bytes(d->B * d->H, d->S - d->window, d->topk) : 0;      // long-row top-k scratch (0 up to 57 344 keys)
  w.off_tk = o;      o = align_up(o + w.tk_bytes, 256);
  w.off_ada = o;     o = align_up(o + 1024 + (size_t)2 * d->H * 256 * 4, 256);                 // Ada-SnapKV budget scratch (pkv_ada_select)
  w.off_ada_list = o; o = align_up(o + (size_t)d->H * align_up((size_t)(d->topk > 0 ? d->topk : 1), 8) * 2, 256);   // looked-up top-M lists
  w.off_rowstat = o; o = align_up(o + (size_t)d->B * d->H * d->S * sizeof(float2), 256);   // H2O only: c_row of every query row (fp32 tensors: (max, 1/sum) pairs)
  w.total = o;
  return w;
}

int do_score_window(const pkv_desc* d, const void* q, const void* k, void* scores, int64_t stride,
                    char* ws, const WsLayout& L, hipStream_t st, bool want_cmax = false, double* rowsum_part = nullptr) {
  LogitsParams lp;
  lp.q = q; lp.k = k;
  lp.logits = ws + L.off_logits;
  lp.partial = reinterpret_cast<float2*>(ws + L.off_partial);
  lp.B = d->B; lp.H = d->H; lp.S = d->S; lp.w = d->window; lp.G = d->kv_group; lp.D = d->D;
  lp.fexp = 0; lp.st_mode = 0; lp.logits_bytes = 0;
  lp.Sp = L.Sp; lp.nT = L.nT; lp.nst = 0; lp.tile = d->D == 256 ? 128 : logits_tile();   // 512-byte rows: 32 keys per wave (register budget)
  if (d->D == 256) lp.nT = (d->S + 127) / 128; lp.nt = logits_nt(); lp.ablate = logits_ablate(); lp.wgtrace = g_wg_trace;
  lp.qs_b = d->q_stride[0]; lp.qs_h = d->q_stride[1]; lp.qs_s = d->q_stride[2];
  lp.ks_b = d->k_stride[0]; lp.ks_h = d->k_stride[1]; lp.ks_s = d->k_stride[2];
  lp.scale_mode = d->scale_mode;
  lp.sqrt_d = (float)sqrt((double)d->D);   // math.sqrt(head_dim), cast to the fp32 opmath type
  lp.rcp_sqrt_d = d->dtype == PKV_F32 ? 1.0f / lp.sqrt_d : scale_multiplier(d->dtype, d->D, d->scale_mode);
  const int C = d->kv_group * d->window;
  int nT_used = L.nT;
  if (d->dtype == PKV_F32) {
    lp.nT = L.nT;
    {
      ProfScope ps(PKV_K_LOGITS, st, true);
      hipError_t e = launch_logits_f32(lp, st);
      if (e != hipSuccess) return hip_fail(e);
    }
    FinalizeParams fp;
    fp.logits = lp.logits; fp.partial = lp.partial;
    fp.scores = scores; fp.scores_stride = stride;
    fp.B = d->B; fp.H = d->H; fp.S = d->S; fp.w = d->window; fp.Sp = L.Sp; fp.nT = L.nT;
    fp.pool_kind = d->pool_kind;
    fp.pool_kernel = d->pool_kind == PKV_POOL_NONE ? 1 : d->pool_kernel;
    fp.reduce = d->reduce;
    fp.cmax = nullptr; fp.cmax_stride = 0; fp.trace = nullptr; fp.wgtrace = nullptr; fp.rowsum_part = nullptr; fp.rowsum_np = 0;
    ProfScope ps(PKV_K_FINALIZE, st, true);
    hipError_t e = launch_finalize_f32(fp, st);
    return e == hipSuccess ? PKV_OK : hip_fail(e);
  }
  if (logits_v2() && C <= 32 && d->D == 128 && !g_wg_trace) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    }
    const int sph = (d->S + 127) / 128;                                     // stages per head group
    const int64_t total = (int64_t)d->B * (d->H / d->kv_group) * sph;
    // stages per workgroup.  C <= 8 columns (expanded K/V, window 8): ONE stage per workgroup - measured in round 2 over
    // {1, 2, 3, 4, 8 stages}: 1 is the fastest at B = 1 (45.1-46.0 us against 45.7-48.6 us for 4 stages in the same
    // sessions; 2 and 3 stages are the slowest) and at B = 8 (350 vs 353 us): the in-workgroup pipeline only pays when few
    // workgroups share a CU.  Round 4 tried a chunk-WALKING grid (as many workgroups as the chip holds at once, each walking
    // chunks blockIdx.x, + gridDim.x, ... with the next chunk's rows in flight): the loop costs 21 registers (one workgroup
    // per CU less, or spills), C = 8: 44.7 -> 58.7 us at 4 per CU, C = 32: 19.95 -> 19.8 us; and TWO stages of K rows in flight
    // per workgroup (a second register set: 166 / 204 registers = one workgroup per CU less): C = 8: 44.6 -> 48.4 us, C = 32:
    // 17.1 -> 21.0 us (19.2 with 4 stages per workgroup).  Both removed (profiles/r04/ab/).
    // More columns per stage (un-expanded GQA K, C = 32): 4 workgroups per CU with 2..8 stages each
    // (one stage per workgroup costs +4 us there).  Round 3, after the logits stores moved one stage back (pkv_score.hip):
    // same ranking - C = 8: 44.3 us at 1 stage, 46.2 / 45.6 / 46.4 / 47.3 at 2 / 4 / 8 / 16; C = 32: 16.8 us at 2 stages
    // (1024 workgroups), 17.2 at 3, 17.3 at 4, 20.1 at 1 (profiles/r03/ab/defer_stores_ab.txt, one_stage_lds_direct_ab.txt).
    // Round 4 measured the review's "every finalize workgroup re-merges all partials of its head" (256 per row at one stage per
    // workgroup): 8 stages per workgroup at B = 8 (32 partials per row) left finalize_kernel at 44.5 us - it is bound by its
    // ~790 vector instructions per wave, not by those reads - and the K scan at 344 us.  One stage stays.
    // Round 6: a launch of at most TWO rounds of the chip's 4-per-CU residency (S = 8192 at B = 1: 2048 stages) runs as ONE round of
    // two-stage workgroups - 14.4 -> 13.5 us there (tools/probes/scan_stages_probe.py); one stage per workgroup stays the
    // fastest from three rounds up (S = 16384: 25.2 either way, S = 32768: 45.9 against 48-51 us).
    const int64_t target = logits_v2_wgs() > 0 ? logits_v2_wgs()
                         : (C > 8 ? (int64_t)4 * cus : (total > (int64_t)4 * cus && total <= (int64_t)8 * cus ? (int64_t)4 * cus : total));
    int nst = (int)((total + target - 1) / target);
    nst = std::max(1, std::min(nst, std::min(sph, 64)));
    lp.nst = nst;
    lp.nT = nT_used = (sph + nst - 1) / nst;
    lp.nt = logits_v2_nt();
    lp.fexp = logits_fexp();
    const size_t lbytes = (size_t)d->B * d->H * d->window * L.Sp * 2;
    lp.st_mode = lbytes < 0xffffffffull ? logits_store() : 0;                 // raw-buffer extent is 32-bit
    lp.logits_bytes = (uint32_t)std::min<size_t>(lbytes, 0xffffffffull);
    ProfScope ps(PKV_K_LOGITS, st, true);
    hipError_t e = launch_logits2(d->dtype, lp, st);
    if (e != hipSuccess) return hip_fail(e);
  } else {
    nT_used = lp.nT;
    ProfScope ps(PKV_K_LOGITS, st, true);
    hipError_t e = launch_logits(d->dtype, lp, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  FinalizeParams fp;
  fp.logits = lp.logits; fp.partial = lp.partial;
  fp.scores = scores; fp.scores_stride = stride;
  fp.B = d->B; fp.H = d->H; fp.S = d->S; fp.w = d->window; fp.Sp = L.Sp; fp.nT = nT_used;
  fp.pool_kind = d->pool_kind;
  fp.pool_kernel = d->pool_kind == PKV_POOL_NONE ? 1 : d->pool_kernel;
  fp.reduce = d->reduce;
  fp.cmax = want_cmax ? ws + L.off_cmax : nullptr; fp.cmax_stride = L.Lp / 8;
  fp.trace = g_topk_trace ? g_topk_trace + 8 : nullptr;
  fp.wgtrace = g_wg_trace;
  fp.rowsum_part = rowsum_part; fp.rowsum_np = finalize_blocks(d->S, d->window, d->B * d->H);
  {
    ProfScope ps(PKV_K_FINALIZE, st, true);
    hipError_t e = launch_finalize(d->dtype, fp, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  return PKV_OK;
}

int do_score_h2o(const pkv_desc* d, const void* q, const void* k, void* scores, int64_t stride,
                 char* ws, const WsLayout& L, hipStream_t st) {
  H2OParams hp;
  hp.q = q; hp.k = k;
  hp.rowstat = reinterpret_cast<float*>(ws + L.off_rowstat);
  hp.scores = scores; hp.scores_stride = stride;
  hp.B = d->B; hp.H = d->H; hp.S = d->S; hp.w = d->window; hp.G = d->kv_group; hp.D = d->D;
  hp.qs_b = d->q_stride[0]; hp.qs_h = d->q_stride[1]; hp.qs_s = d->q_stride[2];
  hp.ks_b = d->k_stride[0]; hp.ks_h = d->k_stride[1]; hp.ks_s = d->k_stride[2];
  hp.scale_mode = d->scale_mode;
  hp.sqrt_d = (float)sqrt((double)d->D);
  hp.rcp_sqrt_d = d->dtype == PKV_F32 ? 1.0f / hp.sqrt_d : scale_multiplier(d->dtype, d->D, d->scale_mode);
  if (d->dtype == PKV_F32) {                                                 // both passes of the fp32 form (pkv_f32.hip)
    ProfScope ps(PKV_K_H2O_COLSUM, st);
    hipError_t e = launch_h2o_f32(hp, st);
    return e == hipSuccess ? PKV_OK : hip_fail(e);
  }
  {
    ProfScope ps(PKV_K_H2O_STATS, st, true);                                 // statistics pass
    hipError_t e = launch_h2o_stats(d->dtype, hp, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  {
    ProfScope ps(PKV_K_H2O_COLSUM, st, true);
    hipError_t e = launch_h2o_colsum(d->dtype, hp, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  return PKV_OK;
}

constexpr int TK_SEG = 32768;     // segment length of the long-row path (one workgroup's LDS holds up to 57 344 keys)

bool topk_fits(int L, int k) {
  int Lw, kpad;
  const size_t lds = topk_lds_bytes(L, k, &Lw, &kpad);
  return lds <= 160 * 1024 && 16 * (size_t)Lw <= 65536;
}

// scratch of the long-row path: segment winners [rows][nseg*k] int32, their scores, positions of the final winners
size_t topk_tmp_bytes(int rows, int L, int k) {
  if (L < 1 || k < 1 || rows < 1 || topk_fits(L, k)) return 0;
  const size_t nseg = ((size_t)L + TK_SEG - 1) / TK_SEG;
  const size_t cand = nseg * (size_t)k, cstride = align_up(cand, 8);
  return align_up((size_t)rows * cand * 4, 256) + align_up((size_t)rows * cstride * 2, 256) + align_up((size_t)rows * k * 4, 256);
}

int do_topk(int dtype, int rows, int L, int k, const void* scores, int64_t stride, const int32_t* kpr,
            int32_t* idx, int64_t idx_stride, hipStream_t st, const void* cmax = nullptr, int64_t cmax_stride = 0,
            void* tmp = nullptr, size_t tmp_bytes = 0, void* list_out = nullptr, int64_t list_stride = 0,
            const double* rowsum_part = nullptr, int rowsum_np = 0, int ada_base = 0, int ada_normalize = 0) {
  if (L < 1 || k < 1 || k > L || rows < 1) return PKV_ERR_SHAPE;
  TopkParams tp;
  tp.list_out = list_out; tp.list_stride = list_stride;                               // Ada-SnapKV hand-over (single-workgroup rows only)
  tp.rowsum_part = rowsum_part; tp.rowsum_np = rowsum_np; tp.ada_base = ada_base; tp.ada_normalize = ada_normalize;
  if (list_out && (dtype == PKV_F32 || !topk_fits(L, k))) return PKV_ERR_UNSUPPORTED;
  tp.scores = scores; tp.scores_stride = stride; tp.L = L; tp.k = k; tp.k_per_row = kpr;
  tp.idx_out = idx; tp.idx_stride = idx_stride; tp.trace = g_topk_trace; tp.wgtrace = g_wg_trace; tp.cmax = cmax; tp.cmax_stride = cmax_stride;
  tp.nseg = 1; tp.seg_len = 0; tp.algo = topk_algo();
  if (dtype == PKV_F32) {                                  // radix select on 32-bit keys (pkv_f32.hip); k_per_row <= k
    if (k > topk_f32_max_k()) return PKV_ERR_UNSUPPORTED;
    ProfScope ps(PKV_K_TOPK, st, true);
    hipError_t e = launch_topk_f32(rows, tp, st);
    return e == hipSuccess ? PKV_OK : hip_fail(e);
  }
  auto launch = [&](int nrows, int Lwg) -> int {     // Lwg = keys one workgroup handles
    const size_t lds = topk_lds_bytes(Lwg, tp.k, &tp.Lw, &tp.kpad);
    if (lds > 160 * 1024 || 16 * (size_t)tp.Lw > 65536) return PKV_ERR_UNSUPPORTED;
    const size_t xw = (size_t)(tp.kpad > 8192 ? tp.kpad : 8192);
    tp.dual = lds >= (size_t)2 * 16 * tp.Lw + 4 * xw + 4 * 256 + 4 * 64 + 4 * 8192 ? 1 : 0;
    ProfScope ps(PKV_K_TOPK, st, true);
    hipError_t e = launch_topk(dtype, nrows, tp, lds, st);
    return e == hipSuccess ? PKV_OK : hip_fail(e);
  };
  if (topk_fits(L, k)) return launch(rows, L);
  // long rows: top-k of every 32k segment (row-global indices, canonical order inside the segment), then the top-k of
  // the nseg*k winners.  The candidate list is segment-major, so position order == index order among equal scores and
  // the second selection reproduces (value desc, index asc) of the whole row.
  const size_t need = topk_tmp_bytes(rows, L, k);
  if (!tmp || tmp_bytes < need || kpr || k > TK_SEG || (reinterpret_cast<uintptr_t>(tmp) & 15)) return PKV_ERR_UNSUPPORTED;
  const int nseg = (L + TK_SEG - 1) / TK_SEG;
  const int64_t cand = (int64_t)nseg * k, cstride = (int64_t)align_up((size_t)cand, 8);
  if (!topk_fits((int)cand, k) || (int64_t)rows * nseg > 0x7fffffff) return PKV_ERR_UNSUPPORTED;
  char* t8 = static_cast<char*>(tmp);
  int32_t* cand_idx = reinterpret_cast<int32_t*>(t8);
  void* cand_score = t8 + align_up((size_t)rows * cand * 4, 256);
  int32_t* pos = reinterpret_cast<int32_t*>(t8 + align_up((size_t)rows * cand * 4, 256) + align_up((size_t)rows * cstride * 2, 256));
  tp.nseg = nseg; tp.seg_len = TK_SEG; tp.idx_out = cand_idx; tp.idx_stride = k;
  int rc = launch(rows * nseg, TK_SEG);
  if (rc) return rc;
  hipError_t e = launch_topk_merge_prep(dtype, rows, L, k, nseg, TK_SEG, scores, stride, cand_idx, cand_score, cstride, st);
  if (e != hipSuccess) return hip_fail(e);
  rc = do_topk(dtype, rows, (int)cand, k, cand_score, cstride, nullptr, pos, k, st);
  if (rc) return rc;
  e = launch_topk_merge_finish(rows, k, cand_idx, cand, pos, idx, idx_stride, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

GatherParams make_gather(const pkv_desc* d, const void* k, const void* v, void* ko, void* vo) {
  GatherParams g;
  g.kptr = k; g.vptr = v; g.k_out = ko; g.v_out = vo;
  g.idx = nullptr; g.idx_stride = 0; g.head_k = nullptr; g.cu_rows = nullptr; g.wgtrace = g_wg_trace;
  g.rpt = gather_rpt(); g.xcd_map = gather_xcd(); g.nblk = 0;
  g.out_rows = (int64_t)d->B * d->H * (d->topk + d->window);          // dense layout; pkv_gather_flat overrides it
  g.B = d->B; g.H = d->H; g.S = d->S; g.w = d->window; g.nsel = d->topk; g.G = d->kv_group; g.D = d->D;
  g.ks_b = d->k_stride[0]; g.ks_h = d->k_stride[1]; g.ks_s = d->k_stride[2];
  g.vs_b = d->v_stride[0]; g.vs_h = d->v_stride[1]; g.vs_s = d->v_stride[2];
  if (d->dtype == PKV_F32) {            // the copy kernels move 16-byte pieces: [.., D] fp32 is [.., 2D] 16-bit with doubled strides
    g.D = 2 * d->D;
    g.ks_b *= 2; g.ks_h *= 2; g.ks_s *= 2; g.vs_b *= 2; g.vs_h *= 2; g.vs_s *= 2;
  }
  return g;
}

int do_gather(const GatherParams& g, int max_rows, hipStream_t st) {
  ProfScope ps(PKV_K_GATHER, st, true);
  hipError_t e = launch_gather(g, max_rows, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int compress_common(bool h2o, const pkv_desc* d, const void* q, const void* k, const void* v, void* k_out,
                    void* v_out, int32_t* idx_out, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true, true, true);
  if (rc) return rc;
  if (!q || !k || !v || !k_out || !v_out || !ws) return PKV_ERR_NULL;
  if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(k_out) || misaligned(v_out) || misaligned(ws))
    return PKV_ERR_ALIGN;
  WsLayout L = ws_layout(d);
  if (ws_bytes < (h2o ? L.total : L.off_rowstat)) return PKV_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  void* scores = w + L.off_scores;
  const bool cm = !h2o && d->dtype != PKV_F32 && topk_cmax() != 0;   // chunk maxima feed the top-k prefilter
  rc = h2o ? do_score_h2o(d, q, k, scores, L.Lp, w, L, st) : do_score_window(d, q, k, scores, L.Lp, w, L, st, cm);
  if (rc) return rc;
  int32_t* idx = idx_out ? idx_out : reinterpret_cast<int32_t*>(w + L.off_idx);
  rc = do_topk(d->dtype, d->B * d->H, d->S - d->window, d->topk, scores, L.Lp, nullptr, idx, d->topk, st,
               cm ? w + L.off_cmax : nullptr, L.Lp / 8,
               w + L.off_tk, L.tk_bytes);
  if (rc) return rc;
  if (d->tie_order == PKV_TIE_ATEN_ROCM && d->dtype != PKV_F32) {             // k <= 32: the order PyTorch-ROCm's topk leaves ties in
    hipError_t e = launch_aten_small_order(d->dtype, d->B * d->H, d->topk, scores, L.Lp, idx, d->topk, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  GatherParams g = make_gather(d, k, v, k_out, v_out);
  g.idx = idx; g.idx_stride = d->topk;
  return do_gather(g, d->topk + d->window, st);
}

}  // namespace

extern "C" {

int pkv_version(void) { return PKV_VERSION; }

const char* pkv_strerror(int s) {
  switch (s) {
    case PKV_OK: return "ok";
    case PKV_ERR_DTYPE: return "unsupported dtype (need bf16 or fp16)";
    case PKV_ERR_SHAPE: return "bad shape / parameter out of range";
    case PKV_ERR_ALIGN: return "pointer or stride breaks 16-byte row alignment";
    case PKV_ERR_WORKSPACE: return "workspace too small";
    case PKV_ERR_UNSUPPORTED: return "request outside the limits of this build";
    case PKV_ERR_HIP: return "HIP runtime error";
    case PKV_ERR_NULL: return "null pointer";
    case PKV_ERR_COLLECTIVE: return "RCCL call failed";
    case PKV_ERR_ABI: return "pkv_desc.struct_size does not match a layout this library knows (host built against another pkv.h)";
    default: return "unknown status";
  }
}

int pkv_last_hip_error(void) { return g_last_hip; }
__attribute__((visibility("hidden"))) void pkv_set_last_hip_error(int e) { g_last_hip = e; }   /* internal (not exported): other translation units of libpkv report through it */


size_t pkv_workspace_bytes(const pkv_desc* d) {
  PKV_LOAD_DESC(d, 0);
  if ( d->S < 2 || d->window < 1 || d->window >= d->S || d->B < 1 || d->H < 1) return 0;
  return ws_layout(d).total;
}

int pkv_score_window(const pkv_desc* d, const void* q, const void* k, void* scores_out,
                     int64_t scores_stride, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, false, true, true);
  if (rc) return rc;
  if (!q || !k || !scores_out || !ws) return PKV_ERR_NULL;
  if (misaligned(q) || misaligned(k) || misaligned(scores_out) || misaligned(ws)) return PKV_ERR_ALIGN;
  WsLayout L = ws_layout(d);
  if (ws_bytes < L.off_scores) return PKV_ERR_WORKSPACE;
  if ((scores_stride & 7) || scores_stride < L.Lp) return PKV_ERR_ALIGN;
  return do_score_window(d, q, k, scores_out, scores_stride, static_cast<char*>(ws), L, static_cast<hipStream_t>(stream));
}

int pkv_score_h2o(const pkv_desc* d, const void* q, const void* k, void* scores_out,
                  int64_t scores_stride, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, false, true, true);
  if (rc) return rc;
  if (!q || !k || !scores_out || !ws) return PKV_ERR_NULL;
  if (misaligned(q) || misaligned(k) || misaligned(ws)) return PKV_ERR_ALIGN;
  WsLayout L = ws_layout(d);
  if (ws_bytes < L.total) return PKV_ERR_WORKSPACE;
  if (scores_stride < d->S - d->window) return PKV_ERR_SHAPE;
  return do_score_h2o(d, q, k, scores_out, scores_stride, static_cast<char*>(ws), L, static_cast<hipStream_t>(stream));
}

int pkv_topk(int32_t dtype, int32_t rows, int32_t L, int32_t k, const void* scores, int64_t scores_stride,
             const int32_t* k_per_row, int32_t* idx_out, int64_t idx_stride, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16 && dtype != PKV_F32) return PKV_ERR_DTYPE;
  if (!scores || !idx_out) return PKV_ERR_NULL;
  if (scores_stride < L || idx_stride < k) return PKV_ERR_SHAPE;
  return do_topk(dtype, rows, L, k, scores, scores_stride, k_per_row, idx_out, idx_stride, static_cast<hipStream_t>(stream));
}

size_t pkv_topk_workspace_bytes(int32_t rows, int32_t L, int32_t k) { return topk_tmp_bytes(rows, L, k); }

int pkv_topk_ws(int32_t dtype, int32_t rows, int32_t L, int32_t k, const void* scores, int64_t scores_stride,
                const int32_t* k_per_row, int32_t* idx_out, int64_t idx_stride, void* ws, size_t ws_bytes,
                pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16 && dtype != PKV_F32) return PKV_ERR_DTYPE;
  if (!scores || !idx_out) return PKV_ERR_NULL;
  if (scores_stride < L || idx_stride < k) return PKV_ERR_SHAPE;
  if (dtype != PKV_F32 && topk_tmp_bytes(rows, L, k) > 0 && (!ws || ws_bytes < topk_tmp_bytes(rows, L, k))) return PKV_ERR_WORKSPACE;
  return do_topk(dtype, rows, L, k, scores, scores_stride, k_per_row, idx_out, idx_stride, static_cast<hipStream_t>(stream),
                 nullptr, 0, ws, ws_bytes);
}

int pkv_gather_compact(const pkv_desc* d, const void* k, const void* v, const int32_t* idx,
                       int64_t idx_stride, void* k_out, void* v_out, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true, false, true);
  if (rc) return rc;
  if (!k || !v || !idx || !k_out || !v_out) return PKV_ERR_NULL;
  if (misaligned(k) || misaligned(v) || misaligned(k_out) || misaligned(v_out)) return PKV_ERR_ALIGN;
  if (idx_stride < d->topk) return PKV_ERR_SHAPE;
  GatherParams g = make_gather(d, k, v, k_out, v_out);
  g.idx = idx; g.idx_stride = idx_stride;
  return do_gather(g, d->topk + d->window, static_cast<hipStream_t>(stream));
}

int pkv_gather_streaming(const pkv_desc* d, const void* k, const void* v, void* k_out, void* v_out,
                         pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true, false, true);
  if (rc) return rc;
  if (!k || !v || !k_out || !v_out) return PKV_ERR_NULL;
  if (misaligned(k) || misaligned(v) || misaligned(k_out) || misaligned(v_out)) return PKV_ERR_ALIGN;
  GatherParams g = make_gather(d, k, v, k_out, v_out);   // idx == nullptr: rows 0..k-1 (:607-608)
  return do_gather(g, d->topk + d->window, static_cast<hipStream_t>(stream));
}

int pkv_compress(const pkv_desc* d, const void* q, const void* k, const void* v, void* k_out, void* v_out,
                 int32_t* idx_out, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  return compress_common(false, d, q, k, v, k_out, v_out, idx_out, ws, ws_bytes, stream);
}

int pkv_compress_h2o(const pkv_desc* d, const void* q, const void* k, const void* v, void* k_out, void* v_out,
                     int32_t* idx_out, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  return compress_common(true, d, q, k, v, k_out, v_out, idx_out, ws, ws_bytes, stream);
}

// ---- selection only (score -> top-k), the front half of pkv_compress: what the merge path needs ----
int pkv_select(const pkv_desc* d, const void* q, const void* k, int32_t h2o, int32_t* idx_out, void* ws, size_t ws_bytes,
               pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true, true, true);
  if (rc) return rc;
  if (!q || !k || !idx_out || !ws) return PKV_ERR_NULL;
  if (misaligned(q) || misaligned(k) || misaligned(ws)) return PKV_ERR_ALIGN;
  WsLayout L = ws_layout(d);
  if (ws_bytes < (h2o ? L.total : L.off_rowstat)) return PKV_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  void* scores = w + L.off_scores;
  const bool cm = !h2o && d->dtype != PKV_F32 && topk_cmax() != 0;
  rc = h2o ? do_score_h2o(d, q, k, scores, L.Lp, w, L, st) : do_score_window(d, q, k, scores, L.Lp, w, L, st, cm);
  if (rc) return rc;
  rc = do_topk(d->dtype, d->B * d->H, d->S - d->window, d->topk, scores, L.Lp, nullptr, idx_out, d->topk, st,
               cm ? w + L.off_cmax : nullptr, L.Lp / 8, w + L.off_tk, L.tk_bytes);
  if (rc) return rc;
  if (d->tie_order == PKV_TIE_ATEN_ROCM && d->dtype != PKV_F32) {
    hipError_t e = launch_aten_small_order(d->dtype, d->B * d->H, d->topk, scores, L.Lp, idx_out, d->topk, st);
    if (e != hipSuccess) return hip_fail(e);
  }
  return PKV_OK;
}

namespace {
struct MergeWs { size_t off_mask, off_bad, off_n, off_drop, off_pivot, off_tn, off_start, off_list, total; int ntp; };
MergeWs merge_ws(const pkv_desc* d) {
  MergeWs m;
  size_t o = 0;
  m.ntp = (int)align_up((size_t)(d->topk + d->window), 8);
  m.off_mask = o;  o = align_up(o + (size_t)d->S, 256);
  m.off_bad = o;   o = align_up(o + (size_t)d->B * d->H * 4, 256);         // directly behind the mask: one memset clears both
  m.off_n = o;     o = align_up(o + 4, 256);
  m.off_drop = o;  o = align_up(o + (size_t)d->S * 4, 256);
  m.off_pivot = o; o = align_up(o + (size_t)d->B * d->H * d->S * 4, 256);
  m.off_tn = o;    o = align_up(o + (size_t)d->B * d->H * m.ntp * d->D * (d->dtype == PKV_F32 ? 4 : 2), 256);
  m.off_start = o; o = align_up(o + (size_t)d->B * d->H * (d->topk + d->window + 1) * 4, 256);
  m.off_list = o;  o = align_up(o + (size_t)d->B * d->H * d->S * 4, 256);
  m.total = o;
  return m;
}
}  // namespace

size_t pkv_merge_workspace_bytes(const pkv_desc* d) {
  PKV_LOAD_DESC(d, 0);
  if ( d->S < 2 || d->window < 1 || d->window >= d->S || d->B < 1 || d->H < 1 || d->topk < 1) return 0;
  return merge_ws(d).total;
}

int pkv_merge_compact(const pkv_desc* d, const void* k, const void* v, const int32_t* idx, int64_t idx_stride,
                      void* k_out, void* v_out, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true, false, true);
  if (rc) return rc;
  if (!k || !v || !idx || !k_out || !v_out || !ws) return PKV_ERR_NULL;
  if (misaligned(k) || misaligned(v) || misaligned(k_out) || misaligned(v_out) || misaligned(ws)) return PKV_ERR_ALIGN;
  if (idx_stride < d->topk) return PKV_ERR_SHAPE;
  if ((size_t)d->S > merge_max_seq() || d->topk + d->window > 65535) return PKV_ERR_UNSUPPORTED;   // LDS bitmap; 16-bit kept-row numbers in the pivot keys
  MergeWs m = merge_ws(d);
  if (ws_bytes < m.total) return PKV_ERR_WORKSPACE;
  char* w = static_cast<char*>(ws);
  MergeParams p;
  p.kptr = k; p.vptr = v; p.idx = idx; p.idx_stride = idx_stride; p.k_out = k_out; p.v_out = v_out;
  p.B = d->B; p.H = d->H; p.S = d->S; p.w = d->window; p.k = d->topk; p.G = d->kv_group;
  p.ks_b = d->k_stride[0]; p.ks_h = d->k_stride[1]; p.ks_s = d->k_stride[2];
  p.vs_b = d->v_stride[0]; p.vs_h = d->v_stride[1]; p.vs_s = d->v_stride[2];
  p.mask = reinterpret_cast<uint8_t*>(w + m.off_mask); p.kept_bad = reinterpret_cast<int32_t*>(w + m.off_bad); p.ndrop = reinterpret_cast<int32_t*>(w + m.off_n);
  p.drop = reinterpret_cast<int32_t*>(w + m.off_drop); p.pivot = reinterpret_cast<int32_t*>(w + m.off_pivot);
  p.tn = w + m.off_tn; p.ntp = m.ntp; p.D = d->D;
  p.bstart = reinterpret_cast<int32_t*>(w + m.off_start); p.blist = reinterpret_cast<int32_t*>(w + m.off_list);
  hipError_t e = launch_merge(d->dtype, p, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_sort_rows(int32_t dtype, int32_t rows, int32_t L, const void* scores, int64_t scores_stride,
                  int32_t* sorted_idx, void* sorted_val, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16) return PKV_ERR_DTYPE;
  if (!scores || !sorted_idx) return PKV_ERR_NULL;
  if (rows < 1 || L < 1 || scores_stride < L) return PKV_ERR_SHAPE;
  if (L > 32768) return PKV_ERR_UNSUPPORTED;
  SortParams sp;
  sp.scores = scores; sp.scores_stride = scores_stride; sp.L = L;
  sp.n = (L + 7) & ~7;   // raw values + index permutation kept in LDS (2 x 2 B each; keeps the tables 16-B aligned)
  sp.sorted_idx = sorted_idx; sp.sorted_val = sorted_val;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(PKV_K_SORT, st);
  sp.trace = g_topk_trace;
  hipError_t e = launch_sort_rows(dtype, rows, sp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_budget(int32_t dtype, int32_t H, int32_t L, const void* sorted_val, int32_t base_capacity,
                   double floor_ratio, int32_t normalize, int32_t* head_capacity, void* ws,
                   size_t ws_bytes, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16) return PKV_ERR_DTYPE;
  if (!sorted_val || !head_capacity || !ws) return PKV_ERR_NULL;
  if (H < 1 || H > 256 || L < 1 || base_capacity < 1 || base_capacity > L) return PKV_ERR_SHAPE;
  if (ws_bytes < 1024 + (size_t)2 * H * 256 * 4) return PKV_ERR_WORKSPACE;
  BudgetParams bp;
  bp.trace = nullptr;
  bp.sorted_idx = nullptr; bp.idx_stride = 0; bp.scores = nullptr; bp.scores_stride = 0; bp.Lrow = L;
  bp.sorted_val = sorted_val; bp.H = H; bp.L = L; bp.base = base_capacity;
  bp.one_minus_floor = (float)(1.0 - floor_ratio);                    // python double, then the fp32 scalar of :719
  bp.floor_capacity = (int)((double)base_capacity * floor_ratio);     // int(base_capacity * floor_ratio) (:632)
  bp.normalize = normalize; bp.head_capacity = head_capacity; bp.ws = ws; bp.list_ws = nullptr; bp.unsorted = 0;
  bp.window = 0; bp.head_lens_out = nullptr; bp.cu_klen_out = nullptr; bp.cu_headlens_out = nullptr;
  bp.host_mirror = nullptr; bp.host_seq = 0; bp.short_list = 0; bp.adaptive_out = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(PKV_K_BUDGET, st);
  hipError_t e = launch_budget(dtype, bp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_budget_topm(int32_t dtype, int32_t H, int32_t L, int32_t M, const void* scores, int64_t scores_stride,
                        const int32_t* top_idx, int64_t idx_stride, int32_t base_capacity, double floor_ratio,
                        int32_t normalize, int32_t window, int32_t* head_capacity, int32_t* head_lens, int32_t* cu_klen,
                        void* ws, size_t ws_bytes, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16) return PKV_ERR_DTYPE;
  if (!scores || !top_idx || !head_capacity || !ws) return PKV_ERR_NULL;
  if ((head_lens == nullptr) != (cu_klen == nullptr)) return PKV_ERR_NULL;
  if (H < 1 || H > 256 || L < 1 || base_capacity < 1 || base_capacity > L) return PKV_ERR_SHAPE;
  // M must cover everything one head can receive: min(L, H*base) (a shorter list could cut a head's share off)
  const int64_t need = std::min<int64_t>(L, (int64_t)H * base_capacity);
  if (M < need || M > L || scores_stride < L || idx_stride < M) return PKV_ERR_SHAPE;
  if (M > 65536) return PKV_ERR_UNSUPPORTED;          // the staged list must fit in LDS (2 bytes per entry)
  if (ws_bytes < 1024 + (size_t)2 * H * 256 * 4) return PKV_ERR_WORKSPACE;
  BudgetParams bp;
  bp.trace = nullptr;
  bp.sorted_val = nullptr; bp.sorted_idx = top_idx; bp.idx_stride = idx_stride; bp.scores = scores; bp.scores_stride = scores_stride;
  bp.Lrow = L; bp.H = H; bp.L = M; bp.base = base_capacity;
  bp.one_minus_floor = (float)(1.0 - floor_ratio);
  bp.floor_capacity = (int)((double)base_capacity * floor_ratio);
  bp.normalize = normalize; bp.head_capacity = head_capacity; bp.ws = ws; bp.list_ws = nullptr; bp.unsorted = 0;
  bp.window = window; bp.head_lens_out = head_lens; bp.cu_klen_out = cu_klen; bp.cu_headlens_out = nullptr;
  bp.host_mirror = nullptr; bp.host_seq = 0; bp.short_list = 0; bp.adaptive_out = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(PKV_K_BUDGET, st);
  hipError_t e = launch_budget(dtype, bp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_budget_rows(int32_t dtype, int32_t H, int32_t L, const void* scores, int64_t scores_stride, int32_t base_capacity,
                        double floor_ratio, int32_t normalize, int32_t window, int32_t* head_capacity, int32_t* head_lens,
                        int32_t* cu_klen, int32_t* cu_headlens, uint64_t* host_mirror, int32_t host_seq, void* ws, size_t ws_bytes,
                        pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16 && dtype != PKV_F32) return PKV_ERR_DTYPE;
  if (!scores || !head_capacity || !ws) return PKV_ERR_NULL;
  if ((head_lens == nullptr) != (cu_klen == nullptr)) return PKV_ERR_NULL;
  if (H < 1 || H > 256 || L < 1 || base_capacity < 1 || base_capacity > L || scores_stride < L) return PKV_ERR_SHAPE;
  if (reinterpret_cast<uintptr_t>(host_mirror) & 7) return PKV_ERR_ALIGN;          // 64-bit words since 0.2.0 (was int32 [H+1] before)
  if (host_mirror && host_seq < 0) return PKV_ERR_SHAPE;
  if (L > (dtype == PKV_F32 ? budget_f32_max_row() : 65536)) return PKV_ERR_UNSUPPORTED;   // a row lives in the registers of one 1024-thread workgroup
  if (ws_bytes < (dtype == PKV_F32 ? 1024 + (size_t)4 * H * 256 * 4 + (size_t)4 * H * 4 : 1024 + (size_t)2 * H * 256 * 4)) return PKV_ERR_WORKSPACE;
  BudgetParams bp;
  bp.trace = nullptr;
  bp.sorted_val = nullptr; bp.sorted_idx = nullptr; bp.idx_stride = 0; bp.scores = scores; bp.scores_stride = scores_stride;
  bp.Lrow = L; bp.H = H; bp.L = L; bp.base = base_capacity;
  bp.one_minus_floor = (float)(1.0 - floor_ratio);
  bp.floor_capacity = (int)((double)base_capacity * floor_ratio);
  bp.normalize = normalize; bp.head_capacity = head_capacity; bp.ws = ws; bp.list_ws = nullptr; bp.unsorted = 1;
  bp.window = window; bp.head_lens_out = head_lens; bp.cu_klen_out = cu_klen; bp.cu_headlens_out = cu_headlens;
  bp.host_mirror = reinterpret_cast<unsigned long long*>(host_mirror); bp.host_seq = host_seq; bp.short_list = 0; bp.adaptive_out = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(PKV_K_BUDGET, st);
  hipError_t e = dtype == PKV_F32 ? launch_budget_f32(bp, st) : launch_budget(dtype, bp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_adaptive_lists(int32_t dtype, int32_t H, int32_t L, int32_t M, const void* scores, int64_t scores_stride,
                           const int32_t* top_idx, int64_t idx_stride, int32_t base_capacity, int32_t normalize,
                           void* lists_out, void* ws, size_t ws_bytes, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16) return PKV_ERR_DTYPE;
  if (!scores || !top_idx || !lists_out || !ws) return PKV_ERR_NULL;
  if (H < 1 || L < 1 || M < 1 || M > L || base_capacity < 1 || base_capacity > M) return PKV_ERR_SHAPE;
  if (scores_stride < L || idx_stride < M) return PKV_ERR_SHAPE;
  if (M > 65536) return PKV_ERR_UNSUPPORTED;
  if (ws_bytes < 1024 + (size_t)2 * H * 256 * 4) return PKV_ERR_WORKSPACE;
  BudgetParams bp;
  bp.trace = nullptr;
  bp.sorted_val = nullptr; bp.sorted_idx = top_idx; bp.idx_stride = idx_stride; bp.scores = scores; bp.scores_stride = scores_stride;
  bp.Lrow = L; bp.H = H; bp.L = M; bp.base = base_capacity;
  bp.one_minus_floor = 1.0f; bp.floor_capacity = 0;
  bp.normalize = normalize; bp.head_capacity = nullptr; bp.ws = ws; bp.list_ws = nullptr; bp.unsorted = 0;
  bp.window = 0; bp.head_lens_out = nullptr; bp.cu_klen_out = nullptr; bp.cu_headlens_out = nullptr;
  bp.host_mirror = nullptr; bp.host_seq = 0; bp.short_list = 0; bp.adaptive_out = lists_out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(PKV_K_BUDGET, st);
  hipError_t e = launch_budget(dtype, bp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_select(const pkv_desc* d, const void* q, const void* k, int32_t base_capacity, double floor_ratio,
                   int32_t normalize, const int32_t* given_capacity, int32_t* top_idx, int32_t* head_capacity,
                   int32_t* head_lens, int32_t* cu_klen, int32_t* cu_headlens, uint64_t* host_mirror, int32_t host_seq,
                   void* ws, size_t ws_bytes, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, true);
  if (rc) return rc;
  if (d->B != 1) return PKV_ERR_SHAPE;                     // reference asserts bsz == 1 (:724)
  if (!q || !k || !top_idx || !head_lens || !cu_klen || !ws) return PKV_ERR_NULL;
  if (!given_capacity && !head_capacity) return PKV_ERR_NULL;
  if (misaligned(q) || misaligned(k) || misaligned(ws) || (reinterpret_cast<uintptr_t>(host_mirror) & 7)) return PKV_ERR_ALIGN;
  if (host_mirror && host_seq < 0) return PKV_ERR_SHAPE;
  const int L = d->S - d->window, M = d->topk, H = d->H;
  if (!given_capacity) {
    if (H > 256 || base_capacity < 1 || base_capacity > L) return PKV_ERR_SHAPE;
    // M >= min(L, H*base) decides everything (pkv_ada_budget_topm).  A SHORTER list (>= base) is accepted together with a host
    // mirror: the result is exact unless bit 31 of the mirror words says that some head's list ran out
    if (M < std::min<int64_t>(L, (int64_t)H * base_capacity) && (!host_mirror || M < base_capacity)) return PKV_ERR_SHAPE;
    if (M > 65536) return PKV_ERR_UNSUPPORTED;
  }
  WsLayout W = ws_layout(d);
  if (ws_bytes < W.off_rowstat) return PKV_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  void* scores = w + W.off_scores;
  const bool cm = topk_cmax() != 0;
  // Round 5: when the lists of all heads fit one workgroup's LDS the kernels in front hand the budget step what it needs -
  // finalize the row sums (one fp64 partial per workgroup), the selection every head's descending list of raw scores - and the
  // budgets are ONE single-workgroup launch
  const int Lpad = (M + 7) & ~7;
  const int np = finalize_blocks(d->S, d->window, d->B * d->H);
  const bool fused = !given_capacity && d->dtype != PKV_F32 && ada_fused_fits(H, M) && topk_fits(L, M) && np <= 128 && ada_fused() != 0;
  double* rowsum = reinterpret_cast<double*>(w + W.off_ada + 1024);           // [H][np] doubles inside the (then unused) count tables
  rc = do_score_window(d, q, k, scores, W.Lp, w, W, st, cm, fused ? rowsum : nullptr);
  if (rc) return rc;
  rc = do_topk(d->dtype, H, L, M, scores, W.Lp, nullptr, top_idx, M, st, cm ? w + W.off_cmax : nullptr, W.Lp / 8,
               w + W.off_tk, W.tk_bytes, fused ? w + W.off_ada_list : nullptr, Lpad, rowsum, np, base_capacity, normalize ? 1 : 0);
  if (rc) return rc;
  if (given_capacity) {                                    // HeadKV: capacities come from the host (:855); metadata only
    hipError_t e = launch_ada_metadata(H, d->window, given_capacity, head_lens, cu_klen, st, cu_headlens);
    return e == hipSuccess ? PKV_OK : hip_fail(e);
  }
  BudgetParams bp;
  bp.trace = nullptr;
  bp.sorted_val = nullptr; bp.sorted_idx = top_idx; bp.idx_stride = M; bp.scores = scores; bp.scores_stride = W.Lp;
  bp.Lrow = L; bp.H = H; bp.L = M; bp.base = base_capacity;
  bp.one_minus_floor = (float)(1.0 - floor_ratio);
  bp.floor_capacity = (int)((double)base_capacity * floor_ratio);
  bp.normalize = normalize; bp.head_capacity = head_capacity; bp.ws = w + W.off_ada; bp.list_ws = w + W.off_ada_list; bp.unsorted = 0;
  bp.window = d->window; bp.head_lens_out = head_lens; bp.cu_klen_out = cu_klen; bp.cu_headlens_out = cu_headlens;
  bp.host_mirror = reinterpret_cast<unsigned long long*>(host_mirror); bp.host_seq = host_seq; bp.adaptive_out = nullptr;
  bp.short_list = M < std::min<int64_t>(L, (int64_t)H * base_capacity) ? 1 : 0;
  bp.trace = g_topk_trace ? g_topk_trace + 16 : nullptr;      // debug build: the caller's buffer holds 32 stamps
  ProfScope ps(PKV_K_BUDGET, st, fused);            // one launch: its own begin / end (three launches: a bracket of event records)
  hipError_t e = fused ? launch_ada_fused(bp, w + W.off_ada_list, Lpad, st) : launch_budget(d->dtype, bp, st);
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_ada_metadata(int32_t H, int32_t window, const int32_t* head_capacity, int32_t* head_lens,
                     int32_t* cu_klen, pkv_stream_t stream) {
  if (!head_capacity || !head_lens || !cu_klen) return PKV_ERR_NULL;
  if (H < 1) return PKV_ERR_SHAPE;
  hipError_t e = launch_ada_metadata(H, window, head_capacity, head_lens, cu_klen, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_gather_flat(const pkv_desc* d, const void* k, const void* v, const int32_t* sorted_idx,
                    int64_t idx_stride, const int32_t* head_capacity, const int32_t* cu_klen,
                    void* k_out, void* v_out, int64_t out_rows, pkv_stream_t stream) {
  PKV_LOAD_DESC(d, rc_ld);
  int rc = check_desc(d, false, false, true);      // fp32 rows move as 2D 16-bit elements (make_gather)
  if (rc) return rc;
  if (d->B != 1) return PKV_ERR_SHAPE;   // reference asserts bsz == 1 (:724)
  if (!k || !v || !sorted_idx || !head_capacity || !cu_klen || !k_out || !v_out) return PKV_ERR_NULL;
  if (misaligned(k) || misaligned(v) || misaligned(k_out) || misaligned(v_out)) return PKV_ERR_ALIGN;
  GatherParams g = make_gather(d, k, v, k_out, v_out);
  g.idx = sorted_idx; g.idx_stride = idx_stride; g.head_k = head_capacity; g.cu_rows = cu_klen;
  g.out_rows = out_rows > 0 ? out_rows : INT64_MAX;
  // worst case rows per head: d->topk if the caller knows max(cap_h), else every past token
  const int max_sel = d->topk > 0 ? d->topk : d->S - d->window;
  return do_gather(g, max_sel + d->window, static_cast<hipStream_t>(stream));
}

int pkv_update_flatten_view(int32_t dtype, int32_t H, int32_t head_dim, const void* cache, const void* state,
                            const int32_t* head_lens, const int32_t* cu_klen, void* out, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16 && dtype != PKV_F32) return PKV_ERR_DTYPE;
  if (!cache || !state || !head_lens || !cu_klen || !out) return PKV_ERR_NULL;
  if (H < 1 || head_dim < 8 || (head_dim & 7)) return PKV_ERR_SHAPE;
  if (misaligned(cache) || misaligned(state) || misaligned(out)) return PKV_ERR_ALIGN;
  FlattenParams fp;
  fp.cache = cache; fp.state = state; fp.head_lens = head_lens; fp.cu_klen = cu_klen; fp.out = out;
  fp.H = H; fp.row_bytes = head_dim * (dtype == PKV_F32 ? 4 : 2);
  hipError_t e = launch_flatten(fp, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

/* ---- debug / test hooks (not part of the drop-in surface) ---- */
#ifdef PKV_DEBUG
int pkv_debug_topk_trace(void* device_u64x8) { g_topk_trace = static_cast<unsigned long long*>(device_u64x8); return PKV_OK; }
int pkv_debug_wg_trace(void* device_u64) { g_wg_trace = static_cast<unsigned long long*>(device_u64); return PKV_OK; }
int pkv_debug_build(void) { return 1; }
#else   // release library: the trace hooks do not exist in the kernels (see pkv_kernels.hpp)
int pkv_debug_topk_trace(void* device_u64x8) { return device_u64x8 ? PKV_ERR_UNSUPPORTED : PKV_OK; }
int pkv_debug_wg_trace(void* device_u64) { return device_u64 ? PKV_ERR_UNSUPPORTED : PKV_OK; }
int pkv_debug_build(void) { return 0; }
#endif

int pkv_runtime_reset(void) {
  ++pkv::g_dyn_lds_epoch;            // every thread's (kernel, device) -> granted-LDS table is dropped on its next launch
  return PKV_OK;
}

float pkv_debug_scale_multiplier(int32_t dtype, int32_t D, int32_t scale_mode) { return scale_multiplier(dtype, D, scale_mode); }

int pkv_debug_exp(const float* in, float* out, int64_t n, pkv_stream_t stream) {
  hipError_t e = launch_debug_exp(in, out, n, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_debug_round(int32_t dtype, const float* in, void* out, int64_t n, pkv_stream_t stream) {
  if (dtype != PKV_BF16 && dtype != PKV_F16) return PKV_ERR_DTYPE;
  hipError_t e = launch_debug_round(dtype, in, static_cast<uint16_t*>(out), n, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? PKV_OK : hip_fail(e);
}

int pkv_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int prev = g_prof_on ? 1 : 0;
  g_prof_on = on != 0;
  return prev;
}

int pkv_prof_read(double* ms_sum, int64_t* launches, int reset) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_pending) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e);
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e);
    g_ms[r.id] += ms;
    g_n[r.id] += 1;
    g_free_events.push_back(r.a);
    g_free_events.push_back(r.b);
  }
  g_pending.clear();
  for (int i = 0; i < PKV_K_COUNT; ++i) {
    if (ms_sum) ms_sum[i] = g_ms[i];
    if (launches) launches[i] = g_n[i];
    if (reset) { g_ms[i] = 0; g_n[i] = 0; }
  }
  return PKV_OK;
}

}  // extern "C"
