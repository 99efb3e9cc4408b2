/* pkv_oracle.c - plain C restatement of the INTEGER / BYTE stages of the reference's update_kv
 * (Zefan-Cai/PyramidKV @ 2024-12-20, pyramidkv/pyramidkv_utils.py).  TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this; the product never does.
 *
 * Stages restated here (bit-exact domain): per-layer pyramid budgets (:205-215, branches :218,:220,:252),
 * max/avg pooling on 16-bit scores (:328-331), canonical top-k (:334, tie rule value desc / index asc),
 * gather-compaction (:335,:341-346), AdaKV head budgets given sorted scores (:709-719), var-len
 * metadata (:682-691) and the decode-time flat append (csrc/csrc/cuda_api.cu:11-53).
 * The floating-point score stage (:317-327) lives in the torch restatement oracle/pkv_oracle.py, which
 * executes the very ATen ops the reference calls.  Pinned by tests/test_oracle_c.py against that
 * restatement and, through it, against tests/golden (outputs of the real reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { BF16 = 0, F16 = 1 };

static float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float f16_to_f32(uint16_t h) {
  uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, u;
  if (e == 0) {
    if (m == 0) u = s << 31;
    else { int sh = 0; while (!(m & 1024)) { m <<= 1; ++sh; } m &= 1023; u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13); }
  } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
  else u = (s << 31) | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static float to_f32(uint16_t h, int dt) { return dt == BF16 ? bf16_to_f32(h) : f16_to_f32(h); }
static uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static uint16_t f32_to_f16(float f) { /* round to nearest even, subnormals, overflow to inf */
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t s = (x >> 16) & 0x8000u, a = x & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u);            /* >= 65536 -> inf (65520 handled by rounding below) */
  if (a < 0x38800000u) {                                           /* subnormal half */
    if (a < 0x33000000u) return (uint16_t)s;                       /* < 2^-25 -> 0 (ties-to-even at exactly 2^-25 -> 0) */
    int e = (int)(a >> 23); uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = 126 - e;                                           /* 14..24 */
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    return (uint16_t)(s | r);
  }
  uint32_t r = a - 0x38000000u;                                    /* rebias */
  uint32_t rem = r & 0x1fffu; r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;           /* may carry into exponent / inf: correct */
  return (uint16_t)(s | r);
}
static uint16_t from_f32(float f, int dt) { return dt == BF16 ? f32_to_bf16(f) : f32_to_f16(f); }
static int is_nan16(uint16_t h, int dt) { return (h & 0x7fff) > (dt == BF16 ? 0x7f80 : 0x7c00); }

/* order-preserving key: NaN greatest (torch), -0 == +0 */
static uint32_t order_key(uint16_t h, int dt) {
  if (is_nan16(h, dt)) return 0xffffu;
  if (h == 0x8000u) h = 0;
  return (h & 0x8000u) ? (uint32_t)(uint16_t)~h : (uint32_t)(h | 0x8000u);
}

/* pyramidkv_utils.py:205-215 + branches.  branch: 0 passthrough (:218), 1 snap (:220), 2 pyramid (:252) */
void pkv_o_pyramid_budget(int cap, int w, int layers, int layer, int q_len, int beta, int* branch, int* k) {
  int min_num = (cap - w) / beta;                    /* python // on non-negative ints */
  int max_num = (cap - w) * 2 - min_num;
  if (max_num >= q_len - w) { max_num = q_len - w; min_num = (cap - w) * 2 - max_num; }
  int d = max_num - min_num, n = layers - 1;
  int steps = d / n; if ((d % n != 0) && ((d < 0) != (n < 0))) --steps;   /* floor division */
  int kl = max_num - layer * steps;
  if (q_len < cap) { *branch = 0; *k = 0; }
  else if (q_len < (cap - w) * 2) { *branch = 1; *k = cap - w; }
  else { *branch = 2; *k = kl; }
}

/* :328-331  pool_kind 1 avg (zero pad, divisor = ks, fp32 sequential sum, one rounding), 2 max (-inf pad) */
void pkv_o_pool(const uint16_t* in, int dt, int L, int pool_kind, int ks, uint16_t* out) {
  int half = ks / 2;
  for (int i = 0; i < L; ++i) {
    if (pool_kind == 2) {
      float m = -INFINITY; uint16_t mb = dt == BF16 ? 0xff80 : 0xfc00;
      for (int j = i - half; j <= i + half; ++j) {
        if (j < 0 || j >= L) continue;
        float v = to_f32(in[j], dt);
        if (v > m) { m = v; mb = in[j]; }
      }
      out[i] = from_f32(m, dt); (void)mb;
    } else {
      float s = 0.f;
      for (int j = i - half; j <= i + half; ++j) if (j >= 0 && j < L) s += to_f32(in[j], dt);
      out[i] = from_f32(s / (float)ks, dt);
    }
  }
}

typedef struct { uint32_t key; int32_t idx; } kv_t;
static int cmp_desc(const void* a, const void* b) {
  const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
/* :334 with the canonical tie rule; idx_out[k] */
int pkv_o_topk(const uint16_t* scores, int dt, int L, int k, int32_t* idx_out) {
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * (size_t)L);
  if (!a) return -1;
  for (int i = 0; i < L; ++i) { a[i].key = order_key(scores[i], dt); a[i].idx = i; }
  qsort(a, (size_t)L, sizeof(kv_t), cmp_desc);
  for (int i = 0; i < k; ++i) idx_out[i] = a[i].idx;
  free(a);
  return 0;
}

/* :335,:341-346 for one (b,h): src rows of row_bytes, L = S - w past rows then w window rows */
void pkv_o_gather(const uint8_t* src, int64_t row_stride_bytes, int row_bytes, int S, int w, const int32_t* idx, int k,
                  uint8_t* out) {
  for (int r = 0; r < k; ++r) memcpy(out + (size_t)r * row_bytes, src + (int64_t)idx[r] * row_stride_bytes, row_bytes);
  for (int r = 0; r < w; ++r) memcpy(out + (size_t)(k + r) * row_bytes, src + (int64_t)(S - w + r) * row_stride_bytes, row_bytes);
}

/* :709-719 given per-head DESCENDING scores sorted[H][L]; cap_out[H] */
int pkv_o_ada_capacity(const uint16_t* sorted, int dt, int H, int L, int base, double floor_ratio, int normalize,
                       int32_t* cap_out) {
  size_t n = (size_t)H * L;
  if (base > L) return -2;                                /* :700: the reference does not compress such a prompt */
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * n);
  if (!a) return -1;
  for (int h = 0; h < H; ++h) {
    const uint16_t* v = sorted + (size_t)h * L;
    float ratio = 1.f;
    if (normalize) {
      double st = 0, sa = 0;
      for (int i = 0; i < L; ++i) { double x = to_f32(v[i], dt); sa += x; if (i < base) st += x; }
      float tq = to_f32(from_f32((float)st, dt), dt), aq = to_f32(from_f32((float)sa, dt), dt);
      ratio = to_f32(from_f32(tq / aq, dt), dt);
    }
    for (int i = 0; i < L; ++i) {
      uint16_t x = v[i];
      if (normalize) x = from_f32(to_f32(x, dt) * ratio, dt);
      a[(size_t)h * L + i].key = order_key(x, dt);
      a[(size_t)h * L + i].idx = (int32_t)((size_t)h * L + i);
    }
  }
  qsort(a, n, sizeof(kv_t), cmp_desc);                    /* stable by construction: idx breaks ties */
  int64_t* cnt = (int64_t*)calloc((size_t)H, sizeof(int64_t));
  for (int64_t i = 0; i < (int64_t)H * base; ++i) cnt[a[i].idx / L] += 1;     /* :714-717 */
  float omf = (float)(1.0 - floor_ratio);
  int floor_cap = (int)((double)base * floor_ratio);                          /* :632 */
  for (int h = 0; h < H; ++h) {
    volatile float m = (float)cnt[h] * omf;                                   /* separate roundings, no fma */
    volatile float c = m + (float)floor_cap;
    cap_out[h] = (int32_t)rintf(c);                                           /* half to even */
  }
  free(cnt); free(a);
  return 0;
}

/* :682-691 */
void pkv_o_ada_metadata(int H, int w, const int32_t* cap, int32_t* head_lens, int32_t* cu_klen) {
  int run = 0;
  for (int h = 0; h < H; ++h) { head_lens[h] = cap[h] + w; cu_klen[h] = run; run += head_lens[h]; }
  cu_klen[H] = run;
}

/* csrc/csrc/cuda_api.cu:11-53 */
void pkv_o_update_flatten_view(const uint8_t* cache, const uint8_t* state, const int32_t* head_lens,
                               const int32_t* cu_klen, int H, int row_bytes, uint8_t* out) {
  for (int h = 0; h < H; ++h) {
    memcpy(out + ((size_t)cu_klen[h] + h) * row_bytes, cache + (size_t)cu_klen[h] * row_bytes, (size_t)head_lens[h] * row_bytes);
    memcpy(out + ((size_t)cu_klen[h + 1] + h) * row_bytes, state + (size_t)h * row_bytes, row_bytes);
  }
}

/* conversions exported for the tests */
float pkv_o_to_f32(uint16_t h, int dt) { return to_f32(h, dt); }
uint16_t pkv_o_from_f32(float f, int dt) { return from_f32(f, dt); }
