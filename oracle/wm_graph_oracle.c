/*
 * wm_graph_oracle.c — CPU restatement of the reference's neighbour sampling / append_unique / add_self_loop
 * (rapidsai/wholegraph 24.12, paths relative to /root/reference/cpp). TEST INFRASTRUCTURE ONLY (see wm_oracle.c).
 *
 * Sampling follows the reference's own HOST statement of its kernels
 *   tests/wholegraph_ops/graph_sampling_test_utils.cu:306-321 (index sampling: Q = iota(N); a[i] = Q[r[i]];
 *   Q[r[i]] = Q[N-1-i]), :323-440 (which random stream feeds which draw; the > 1024 reservoir with "largest
 *   candidate wins"), and src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:40-300,407-445.
 * The random generator is raft's PCGenerator (raft 24.12, NOT vendored): PCG-XSH-RR 64/32 restated from the published
 * algorithm. PARITY UNPINNED for the raft wrapping (seeding order, skip-ahead by the subsequence, sign-bit clearing):
 * the reference holds no fixed output of it. The PCG core itself is pinned by the pcg32 known-answer vector in
 * tests/test_graph_oracle.py.
 * append_unique: tests/graph_ops/append_unique_test_utils.cu:27-80 (targets first, then neighbours in first-seen
 * order — the reference kernel's tail order is hash-slot order, its test only compares after sorting).
 * add_self_loop: src/graph_ops/csr_add_self_loop_func.cuh:24-44.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint64_t state, inc;
} pcg_t;

static uint32_t pcg_next(pcg_t* g)
{
  uint64_t old = g->state;
  g->state     = old * 6364136223846793005ULL + g->inc;
  uint32_t xs  = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((32u - rot) & 31u));
}

/* jump `delta` steps of the LCG in O(log delta) (Brown 1994) */
static void pcg_skip(pcg_t* g, uint64_t delta)
{
  uint64_t acc_mult = 1, acc_plus = 0, cur_mult = 6364136223846793005ULL, cur_plus = g->inc;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
  g->state = acc_mult * g->state + acc_plus;
}

void wmo_pcg_init(pcg_t* g, uint64_t seed, uint64_t subsequence, uint64_t offset)
{
  g->state = 0;
  g->inc   = (subsequence << 1) | 1u;
  pcg_next(g);
  g->state += seed;
  pcg_next(g);
  pcg_skip(g, offset);
}

/* raft PCGenerator(DeviceState{seed, base_subsequence = 0}, subsequence) */
static void pcg_for_thread(pcg_t* g, uint64_t seed, uint64_t subsequence) { wmo_pcg_init(g, seed, subsequence, subsequence); }

static int32_t pcg_i32(pcg_t* g) { return (int32_t)(pcg_next(g) & 0x7fffffffu); }

/* raw 32-bit outputs of init(seed, subsequence, offset): the known-answer hook */
void wmo_pcg_raw(uint64_t seed, uint64_t subsequence, uint64_t offset, int64_t n, uint32_t* out)
{
  pcg_t g;
  wmo_pcg_init(&g, seed, subsequence, offset);
  for (int64_t i = 0; i < n; i++) out[i] = pcg_next(&g);
}

/* what generate_random_positive_int_cpu returns (raft_random_gen.cu:26-63) */
void wmo_random_positive_int(int64_t seed, int64_t subsequence, int64_t n, int is64, void* out)
{
  pcg_t g;
  pcg_for_thread(&g, (uint64_t)seed, (uint64_t)subsequence);
  for (int64_t i = 0; i < n; i++) {
    if (is64) {
      uint64_t lo = pcg_next(&g), hi = pcg_next(&g);
      ((int64_t*)out)[i] = (int64_t)((lo | (hi << 32)) & 0x7fffffffffffffffULL);
    } else {
      ((int32_t*)out)[i] = pcg_i32(&g);
    }
  }
}

static void geometry(int max_sample, int* threads, int* items)
{
  /* func_array / warp_count_array of unweighted_sample_without_replacement_func.cuh:407-445 */
  static const int warps[32] = {1, 1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
  static const int ipt[32]   = {1, 2, 3, 2, 3, 3, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};
  int f    = (max_sample - 1) / 32;
  *threads = warps[f] * 32;
  *items   = ipt[f];
}

/* counts[i] = min(deg, max_sample) (all when max_sample <= 0); offsets = exclusive scan, offsets[n] = total */
int64_t wmo_sample_offsets(const int64_t* row_ptr, const void* centers, int center_is64, int64_t n, int max_sample, int32_t* offsets)
{
  int64_t acc = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t nid = center_is64 ? ((const int64_t*)centers)[i] : ((const int32_t*)centers)[i];
    int64_t deg = row_ptr[nid + 1] - row_ptr[nid];
    if (max_sample > 0 && deg > max_sample) deg = max_sample;
    offsets[i] = (int32_t)acc;
    acc += deg;
  }
  offsets[n] = (int32_t)acc;
  return acc;
}

/* out_ids: int64 neighbour ids (caller narrows); out_lid int32; out_egid int64. Outputs sized offsets[n]. */
void wmo_sample_unweighted(const int64_t* row_ptr, const void* col, int col_is64, const void* centers, int center_is64,
                           int64_t n, int max_sample, uint64_t seed, const int32_t* offsets, int64_t* out_ids,
                           int32_t* out_lid, int64_t* out_egid)
{
  for (int64_t c = 0; c < n; c++) {
    int64_t nid   = center_is64 ? ((const int64_t*)centers)[c] : ((const int32_t*)centers)[c];
    int64_t start = row_ptr[nid], end = row_ptr[nid + 1];
    int N = (int)(end - start), M = max_sample, off = offsets[c];
    if (N <= 0) continue;
    int take  = (M <= 0 || N <= M) ? N : M;
    int* a    = (int*)malloc(sizeof(int) * (size_t)take);
    if (M <= 0 || N <= M) {
      for (int i = 0; i < N; i++) a[i] = i;
    } else if (M > 1024) {
      /* large_sample_kernel: 32 threads, thread t visits idx = M + t, M + t + 32, ...; slot <- max(slot, idx) */
      for (int i = 0; i < M; i++) a[i] = i;
      for (int t = 0; t < 32; t++) {
        pcg_t g;
        pcg_for_thread(&g, seed, (uint64_t)c * 32 + (uint64_t)t);
        for (int idx = M + t; idx < N; idx += 32) {
          int32_t r = pcg_i32(&g) % (idx + 1);
          if (r < M && a[r] < idx) a[r] = idx;
        }
      }
    } else {
      int T, items;
      geometry(M, &T, &items);
      int* r = (int*)malloc(sizeof(int) * (size_t)M);
      for (int t = 0; t < T; t++) {
        pcg_t g;
        pcg_for_thread(&g, seed, (uint64_t)c * (uint64_t)T + (uint64_t)t);
        for (int k = 0; k < items; k++) {
          int idx    = k * T + t;
          int32_t rn = pcg_i32(&g);
          if (idx < M) r[idx] = rn % (N - idx);
        }
      }
      int* Q = (int*)malloc(sizeof(int) * (size_t)N);
      for (int i = 0; i < N; i++) Q[i] = i;
      for (int i = 0; i < M; i++) {
        a[i]    = Q[r[i]];
        Q[r[i]] = Q[N - 1 - i];
      }
      free(Q);
      free(r);
    }
    for (int i = 0; i < take; i++) {
      int64_t e       = start + a[i];
      out_ids[off + i] = col_is64 ? ((const int64_t*)col)[e] : ((const int32_t*)col)[e];
      if (out_lid) out_lid[off + i] = (int32_t)c;
      if (out_egid) out_egid[off + i] = e;
    }
    free(a);
  }
}

/* ---- append_unique: open-addressing set keyed by id, values = unique index ---- */
typedef struct {
  int64_t* keys;
  int32_t* vals;
  uint64_t mask;
} idmap_t;

static void idmap_init(idmap_t* m, int64_t n)
{
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
  m->keys = (int64_t*)malloc(sizeof(int64_t) * cap);
  m->vals = (int32_t*)malloc(sizeof(int32_t) * cap);
  m->mask = cap - 1;
  for (uint64_t i = 0; i < cap; i++) m->vals[i] = -1;
}
static int32_t* idmap_slot(idmap_t* m, int64_t key)
{
  uint64_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ULL) >> 17;
  for (;; h++) {
    uint64_t s = h & m->mask;
    if (m->vals[s] < 0) {
      m->keys[s] = key;
      return &m->vals[s];
    }
    if (m->keys[s] == key) return &m->vals[s];
  }
}

/* returns the unique count; out_unique has room for nt + nn int64; mapping (nn) may be NULL.
 * Duplicate targets keep every copy in place (the reference inserts targets with their own index; lookups resolve to
 * one of them — here the first). */
int64_t wmo_append_unique(const int64_t* targets, int64_t nt, const int64_t* neighbors, int64_t nn, int64_t* out_unique,
                          int32_t* mapping)
{
  idmap_t m;
  idmap_init(&m, nt + nn);
  for (int64_t i = 0; i < nt; i++) {
    int32_t* v = idmap_slot(&m, targets[i]);
    if (*v < 0) *v = (int32_t)i;
    out_unique[i] = targets[i];
  }
  int64_t count = nt;
  for (int64_t i = 0; i < nn; i++) {
    int32_t* v = idmap_slot(&m, neighbors[i]);
    if (*v < 0) {
      *v                  = (int32_t)count;
      out_unique[count++] = neighbors[i];
    }
    if (mapping) mapping[i] = *v;
  }
  free(m.keys);
  free(m.vals);
  return count;
}

void wmo_csr_add_self_loop(const int32_t* row_ptr, const int32_t* col, int64_t n_rows, int32_t* out_row, int32_t* out_col)
{
  for (int64_t r = 0; r < n_rows; r++) {
    int32_t s = row_ptr[r], e = row_ptr[r + 1];
    out_row[r]          = s + (int32_t)r;
    out_col[s + r]      = (int32_t)r;
    for (int32_t k = s; k < e; k++) out_col[k + r + 1] = col[k];
  }
  out_row[n_rows] = row_ptr[n_rows] + (int32_t)n_rows;
}

/* ---- weighted sampling (A-Res), weighted_sample_without_replacement_func.cuh:44-63,183-300 and the reference's host
 * statement tests/wholegraph_ops/graph_sampling_test_utils.cu:548-660 (virtual block of 128 threads, 256 when
 * max_sample > 256; thread j keys neighbours j, j + block, ... from stream center * block + j; the max_sample largest
 * keys win).
 * The key needs log2(1 + x): the product computes it with a fixed sequence of +, *, / in double so that host and
 * device agree bit for bit (wholegraph_amd/csrc/pcg.hpp:det_log2_1p); the same sequence is restated here, and
 * tests/test_graph_oracle.py checks it against libm to 1e-15. Ties (never equal composite keys: the neighbour index is
 * part of the key) break towards the smaller index; output order is key-descending. */
double wmo_det_log2_1p(double x)
{
  const double inv_ln2 = 1.4426950408889634074, ln2 = 0.69314718055994530942;
  if (x > -7.450580596923828125e-9) return (x - x * x * 0.5) * inv_ln2;
  double U = 1.0 + x;
  if (!(U > 0.0)) return -__builtin_inf();
  uint64_t bits;
  memcpy(&bits, &U, 8);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  bits  = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double f;
  memcpy(&f, &bits, 8);
  if (f > 1.4142135623730951) {
    f *= 0.5;
    e += 1;
  }
  double s = (f - 1.0) / (f + 1.0), s2 = s * s, p = 2.0 / 27.0;
  for (int k = 25; k >= 3; k -= 2) p = p * s2 + 2.0 / (double)k;
  p = p * s2 + 2.0;
  return ((double)e * ln2 + p * s) * inv_ln2;
}

static float weighted_key(pcg_t* g, float weight)
{
  float u0 = (float)(pcg_next(g) >> 8) / (float)(1u << 24);
  float m  = (float)(-(0.5 + 0.5 * (double)u0));
  uint64_t stream;
  int redraws = -1;
  do {
    uint64_t lo = pcg_next(g), hi = pcg_next(g);
    stream      = lo | (hi << 32);
    redraws++;
  } while (stream == 0);
  int zeros = redraws * 64 + __builtin_clzll(stream);
  double scale = 1.0;
  for (int z = zeros; z > 0;) {
    int step = z > 30 ? 30 : z;
    scale *= 1.0 / (double)(1u << step);
    z -= step;
  }
  float log2u = (float)wmo_det_log2_1p((double)m * scale);
  return log2u * (1.0f / weight);
}

void wmo_weighted_keys(uint64_t seed, uint64_t subsequence, int64_t n, const float* weights, float* keys)
{
  pcg_t g;
  pcg_for_thread(&g, seed, subsequence);
  for (int64_t i = 0; i < n; i++) keys[i] = weighted_key(&g, weights[i]);
}

typedef struct {
  float key;
  int idx;
} wkey_t;

static int wkey_cmp(const void* a, const void* b)
{
  const wkey_t *x = (const wkey_t*)a, *y = (const wkey_t*)b;
  if (x->key > y->key) return -1;
  if (x->key < y->key) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

void wmo_sample_weighted(const int64_t* row_ptr, const void* col, int col_is64, const void* weights, int weight_is_double,
                         const void* centers, int center_is64, int64_t n, int max_sample, uint64_t seed,
                         const int32_t* offsets, int64_t* out_ids, int32_t* out_lid, int64_t* out_egid)
{
  int block = max_sample > 256 ? 256 : 128;
  for (int64_t c = 0; c < n; c++) {
    int64_t nid   = center_is64 ? ((const int64_t*)centers)[c] : ((const int32_t*)centers)[c];
    int64_t start = row_ptr[nid], end = row_ptr[nid + 1];
    int N = (int)(end - start), M = max_sample, off = offsets[c];
    if (N <= 0) continue;
    int take  = (M <= 0 || N <= M) ? N : M;
    wkey_t* k = (wkey_t*)malloc(sizeof(wkey_t) * (size_t)N);
    if (M <= 0 || N <= M) {
      for (int i = 0; i < N; i++) k[i].idx = i;
    } else {
      for (int t = 0; t < block; t++) {
        pcg_t g;
        pcg_for_thread(&g, seed, (uint64_t)c * (uint64_t)block + (uint64_t)t);
        for (int id = t; id < N; id += block) {
          float w  = weight_is_double ? (float)((const double*)weights)[start + id] : ((const float*)weights)[start + id];
          k[id].key = weighted_key(&g, w);
          k[id].idx = id;
        }
      }
      qsort(k, (size_t)N, sizeof(wkey_t), wkey_cmp);
    }
    for (int i = 0; i < take; i++) {
      int64_t e       = start + k[i].idx;
      out_ids[off + i] = col_is64 ? ((const int64_t*)col)[e] : ((const int32_t*)col)[e];
      if (out_lid) out_lid[off + i] = (int32_t)c;
      if (out_egid) out_egid[off + i] = e;
    }
    free(k);
  }
}
