/*
 * hx_oracle.c -- CPU restatement (plain C + OpenMP) of the hybrid-retrieval hot path.
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; the product (rag_application_amd) never touches it.
 *
 * PARITY STATUS: parity unpinned (see oracle/oracle.py header): the reference
 * delegates this arithmetic to un-pinned Qdrant / fastembed.  This file follows the
 * same contract as oracle/oracle.py, function by function, and tests/ check the two
 * against each other bit for bit:
 *   ho_spec_dot            oracle.spec_dot            (cosine of COSINE collections,
 *                                                      qdrant_handler.py:59-77)
 *   ho_cosine_preprocess   oracle.cosine_preprocess
 *   ho_quantize_i8         oracle.quantize_i8         (qdrant_handler.py:144-146, 300-302)
 *   ho_search_dense/_i8    OracleIndex.search_dense / search_i8 (:311-315, 327-329, 335-339)
 *   ho_search_sparse       OracleIndex.search_sparse  (:347-354)
 *   ho_rescore             OracleIndex.rescore        (:307-330, 333-344, 363-372)
 *   ho_rrf                 oracle.rrf                 (:357-360)
 *   ho_synth_*             oracle.synth_*             
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off: no FMA contraction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- total order -------------------------------------------------------------- */
static inline uint32_t f32_orderable(float s) {
  uint32_t u;
  memcpy(&u, &s, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float orderable_f32(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float s;
  memcpy(&s, &u, 4);
  return s;
}
static inline uint64_t make_key(float s, uint32_t id) {
  return ((uint64_t)f32_orderable(s) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}

/* bounded min-heap of the best L keys (root = worst kept) */
typedef struct {
  uint64_t* k;
  int n, cap;
} heap_t;
static void heap_push(heap_t* h, uint64_t key) {
  if (h->n < h->cap) {
    int i = h->n++;
    h->k[i] = key;
    while (i > 0) {
      int p = (i - 1) >> 1;
      if (h->k[p] <= h->k[i]) break;
      uint64_t t = h->k[p]; h->k[p] = h->k[i]; h->k[i] = t;
      i = p;
    }
  } else if (h->cap > 0 && key > h->k[0]) {
    int i = 0;
    h->k[0] = key;
    for (;;) {
      int l = 2 * i + 1, r = l + 1, m = i;
      if (l < h->n && h->k[l] < h->k[m]) m = l;
      if (r < h->n && h->k[r] < h->k[m]) m = r;
      if (m == i) break;
      uint64_t t = h->k[m]; h->k[m] = h->k[i]; h->k[i] = t;
      i = m;
    }
  }
}
static int cmp_desc_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}
static void emit_sorted(uint64_t* keys, int n, int L, float* out_s, int64_t* out_i, int* out_c) {
  qsort(keys, (size_t)n, 8, cmp_desc_u64);
  for (int r = 0; r < L; ++r) {
    if (r < n) {
      out_s[r] = orderable_f32((uint32_t)(keys[r] >> 32));
      out_i[r] = (int64_t)(0xFFFFFFFFu - (uint32_t)keys[r]);
    } else {
      out_s[r] = -INFINITY;
      out_i[r] = -1;
    }
  }
  *out_c = n < L ? n : L;
}

/* ---- spec arithmetic ------------------------------------------------------------ */
float ho_spec_dot(const float* x, const float* q, int dim) {
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.0f;
  int j = 0;
  for (; j + 64 <= dim; j += 64)
    for (int l = 0; l < 64; ++l) p[l] = p[l] + x[j + l] * q[j + l];
  for (int l = 0; j + l < dim; ++l) p[l] = p[l] + x[j + l] * q[j + l];
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; ++l) p[l] = p[l] + p[l + off];
  return p[0] + 0.0f;
}

/* out[r, 0..d) = normalised in[r, 0..d) ; in has row stride `stride` */
void ho_cosine_preprocess(const float* in, int64_t n, int stride, int d, float* out, int skip_if_unit) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* x = in + r * stride;
    float* o = out + r * d;
    const float len2 = ho_spec_dot(x, x, d);
    int keep = len2 < 1.1920929e-07f;
    if (skip_if_unit && fabsf(len2 - 1.0f) <= 1.0e-6f) keep = 1;
    if (keep) {
      memcpy(o, x, (size_t)d * 4);
    } else {
      const float ln = sqrtf(len2);
      for (int c = 0; c < d; ++c) o[c] = x[c] / ln;
    }
  }
}

static inline int8_t quant1(float x) {
  const double t = (double)x * 127.0;
  const int32_t v = (fabs(t) < 2147483648.0) ? (int32_t)t : 0;
  return (int8_t)(v & 0xFF);
}
void ho_quantize_i8(const float* in, int64_t n, int dim, int8_t* out, float* rinv) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    int64_t n2 = 0;
    for (int c = 0; c < dim; ++c) {
      const int8_t t = quant1(in[r * dim + c]);
      out[r * dim + c] = t;
      n2 += (int)t * (int)t;
    }
    if (rinv) rinv[r] = n2 > 0 ? (float)(1.0 / sqrt((double)n2)) : 0.0f;
  }
}

/* ---- whole-collection dense / int8 search ------------------------------------------ */
static int n_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
int ho_num_threads(void) { return n_threads(); }
/* the timed baseline runs on a stated number of host threads (bench.py: the box share of one GPU) whatever another
 * OpenMP user of the process (PyTorch) set before */
void ho_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* Xn [n x d] normalised rows, Qn [B x d] normalised queries */
void ho_search_dense(const float* Xn, int64_t n, int d, const float* Qn, int B, int L, int64_t id_base,
                     float* out_s, int64_t* out_i, int* out_c) {
  const int T = n_threads();
  /* per (thread, query) heaps, merged at the end: threads split the rows */
  uint64_t* store = (uint64_t*)malloc((size_t)T * B * (size_t)(L > 0 ? L : 1) * 8);
  heap_t* heaps = (heap_t*)malloc((size_t)T * B * sizeof(heap_t));
  for (int64_t i = 0; i < (int64_t)T * B; ++i) {
    heaps[i].k = store + i * (L > 0 ? L : 1);
    heaps[i].n = 0;
    heaps[i].cap = L;
  }
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < n; ++r) {
      const float* x = Xn + r * d;
      for (int b = 0; b < B; ++b) {
        const float s = ho_spec_dot(x, Qn + (int64_t)b * d, d);
        heap_push(&heaps[(int64_t)t * B + b], make_key(s, (uint32_t)(id_base + r)));
      }
    }
  }
  uint64_t* tmp = (uint64_t*)malloc((size_t)T * (size_t)(L > 0 ? L : 1) * 8);
  for (int b = 0; b < B; ++b) {
    int m = 0;
    for (int t = 0; t < T; ++t) {
      heap_t* h = &heaps[(int64_t)t * B + b];
      memcpy(tmp + m, h->k, (size_t)h->n * 8);
      m += h->n;
    }
    emit_sorted(tmp, m, L, out_s + (int64_t)b * L, out_i + (int64_t)b * L, out_c + b);
  }
  free(tmp);
  free(heaps);
  free(store);
}

void ho_search_i8(const int8_t* X8, const float* rinv_x, int64_t n, int dim, const int8_t* Q8,
                  const float* rinv_q, int B, int L, int64_t id_base, float* out_s, int64_t* out_i,
                  int* out_c) {
  const int T = n_threads();
  uint64_t* store = (uint64_t*)malloc((size_t)T * B * (size_t)(L > 0 ? L : 1) * 8);
  heap_t* heaps = (heap_t*)malloc((size_t)T * B * sizeof(heap_t));
  for (int64_t i = 0; i < (int64_t)T * B; ++i) {
    heaps[i].k = store + i * (L > 0 ? L : 1);
    heaps[i].n = 0;
    heaps[i].cap = L;
  }
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < n; ++r) {
      const int8_t* x = X8 + r * dim;
      for (int b = 0; b < B; ++b) {
        const int8_t* q = Q8 + (int64_t)b * dim;
        int32_t dot = 0;
        for (int c = 0; c < dim; ++c) dot += (int32_t)x[c] * (int32_t)q[c];
        const float s = ((float)dot * rinv_x[r]) * rinv_q[b];
        heap_push(&heaps[(int64_t)t * B + b], make_key(s, (uint32_t)(id_base + r)));
      }
    }
  }
  uint64_t* tmp = (uint64_t*)malloc((size_t)T * (size_t)(L > 0 ? L : 1) * 8);
  for (int b = 0; b < B; ++b) {
    int m = 0;
    for (int t = 0; t < T; ++t) {
      heap_t* h = &heaps[(int64_t)t * B + b];
      memcpy(tmp + m, h->k, (size_t)h->n * 8);
      m += h->n;
    }
    emit_sorted(tmp, m, L, out_s + (int64_t)b * L, out_i + (int64_t)b * L, out_c + b);
  }
  free(tmp);
  free(heaps);
  free(store);
}

/* re-score candidate ids (deduplicated; ids outside [id_base, id_base+n) skipped) */
void ho_rescore(const float* Xn, int64_t n, int d, const float* qn, const int64_t* cand, int ncand, int L,
                int64_t id_base, float* out_s, int64_t* out_i, int* out_c) {
  uint64_t* keys = (uint64_t*)malloc((size_t)(ncand > 0 ? ncand : 1) * 8);
  int m = 0;
  for (int i = 0; i < ncand; ++i) {
    const int64_t local = cand[i] - id_base;
    if (cand[i] < 0 || local < 0 || local >= n) continue;
    keys[m++] = make_key(ho_spec_dot(Xn + local * d, qn, d), (uint32_t)cand[i]);
  }
  qsort(keys, (size_t)m, 8, cmp_desc_u64);
  int u = 0;
  for (int i = 0; i < m; ++i)
    if (i == 0 || keys[i] != keys[i - 1]) keys[u++] = keys[i];
  emit_sorted(keys, u, L, out_s, out_i, out_c);
  free(keys);
}

/* ---- sparse: inverted index + term-at-a-time scoring --------------------------------- */
typedef struct {
  int64_t n_docs, nnz, n_terms;
  int32_t* terms;   /* ascending unique term ids */
  int64_t* off;     /* [n_terms + 1] */
  int32_t* doc;     /* [nnz] ascending inside a posting run */
  float* w;         /* [nnz] */
} ho_inv;

static void radix_pass(const uint64_t* kin, const uint64_t* pin, uint64_t* kout, uint64_t* pout, int64_t n,
                       int shift) {
  int64_t* cnt = (int64_t*)calloc(65537, 8);
  for (int64_t i = 0; i < n; ++i) cnt[((kin[i] >> shift) & 0xFFFF) + 1]++;
  for (int i = 0; i < 65536; ++i) cnt[i + 1] += cnt[i];
  for (int64_t i = 0; i < n; ++i) {
    const int64_t p = cnt[(kin[i] >> shift) & 0xFFFF]++;
    kout[p] = kin[i];
    pout[p] = pin[i];
  }
  free(cnt);
}

ho_inv* ho_inv_build(const int64_t* indptr, const int32_t* idx, const float* val, int64_t n_docs) {
  const int64_t nnz = indptr[n_docs];
  ho_inv* iv = (ho_inv*)calloc(1, sizeof(ho_inv));
  iv->n_docs = n_docs;
  iv->nnz = nnz;
  uint64_t* k0 = (uint64_t*)malloc((size_t)(nnz + 1) * 8), *k1 = (uint64_t*)malloc((size_t)(nnz + 1) * 8);
  uint64_t* p0 = (uint64_t*)malloc((size_t)(nnz + 1) * 8), *p1 = (uint64_t*)malloc((size_t)(nnz + 1) * 8);
  for (int64_t d = 0; d < n_docs; ++d)
    for (int64_t i = indptr[d]; i < indptr[d + 1]; ++i) {
      uint32_t wb;
      memcpy(&wb, &val[i], 4);
      k0[i] = (uint64_t)(uint32_t)idx[i];
      p0[i] = ((uint64_t)d << 32) | wb;
    }
  radix_pass(k0, p0, k1, p1, nnz, 0);    /* stable LSD: docs stay ascending */
  radix_pass(k1, p1, k0, p0, nnz, 16);
  iv->doc = (int32_t*)malloc((size_t)(nnz + 1) * 4);
  iv->w = (float*)malloc((size_t)(nnz + 1) * 4);
  int64_t nt = 0;
  for (int64_t i = 0; i < nnz; ++i)
    if (i == 0 || k0[i] != k0[i - 1]) ++nt;
  iv->n_terms = nt;
  iv->terms = (int32_t*)malloc((size_t)(nt + 1) * 4);
  iv->off = (int64_t*)malloc((size_t)(nt + 1) * 8);
  int64_t t = 0;
  for (int64_t i = 0; i < nnz; ++i) {
    if (i == 0 || k0[i] != k0[i - 1]) {
      iv->terms[t] = (int32_t)k0[i];
      iv->off[t] = i;
      ++t;
    }
    iv->doc[i] = (int32_t)(p0[i] >> 32);
    const uint32_t wb = (uint32_t)p0[i];
    memcpy(&iv->w[i], &wb, 4);
  }
  iv->off[nt] = nnz;
  free(k0); free(k1); free(p0); free(p1);
  return iv;
}
void ho_inv_free(ho_inv* iv) {
  if (!iv) return;
  free(iv->terms); free(iv->off); free(iv->doc); free(iv->w); free(iv);
}

static int64_t find_term(const ho_inv* iv, int32_t term) {
  int64_t lo = 0, hi = iv->n_terms;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (iv->terms[mid] < term) lo = mid + 1; else hi = mid;
  }
  return (lo < iv->n_terms && iv->terms[lo] == term) ? lo : -1;
}

/* Sparse arithmetic switch (oracle.py SPARSE_FIX_BITS): < 0 = fp32 running sum over the query's terms in
 * ascending term id (upstream's order, the default); k >= 0 = sum of rint(f64(q_t)*f64(d_t)*2^k) as int64,
 * then f32(sum) * 2^-k (order-independent; round 1's engine arithmetic, kept as a switch). */
static int g_fix_bits = -1;
void ho_set_sparse_fix_bits(int bits) { g_fix_bits = bits; }
int ho_get_sparse_fix_bits(void) { return g_fix_bits; }

/* the query's terms of [qip[b], qip[b+1]) in ascending term id (stable): order[] holds positions */
static void sorted_terms(const int32_t* qix, int64_t s, int64_t e, int64_t* order) {
  const int64_t n = e - s;
  for (int64_t i = 0; i < n; ++i) order[i] = s + i;
  for (int64_t i = 1; i < n; ++i) {          /* insertion sort: a query has a handful of terms */
    const int64_t v = order[i];
    int64_t j = i;
    while (j > 0 && qix[order[j - 1]] > qix[v]) { order[j] = order[j - 1]; --j; }
    order[j] = v;
  }
}

/* term-at-a-time over the inverted index (OracleIndex.sparse_scores) */
void ho_search_sparse(const ho_inv* iv, const int64_t* qip, const int32_t* qix, const float* qv, int B,
                      int L, int64_t id_base, float* out_s, int64_t* out_i, int* out_c) {
  const int fix = g_fix_bits;
  const double scale = fix >= 0 ? ldexp(1.0, fix) : 0.0;
  const float unscale = fix >= 0 ? (float)ldexp(1.0, -fix) : 0.0f;
  int64_t tmax = 1;
  for (int b = 0; b < B; ++b) if (qip[b + 1] - qip[b] > tmax) tmax = qip[b + 1] - qip[b];
#pragma omp parallel
  {
    int64_t* acc = (int64_t*)calloc((size_t)(iv->n_docs + 1), 8);
    float* accf = (float*)calloc((size_t)(iv->n_docs + 1), 4);
    uint8_t* seen = (uint8_t*)calloc((size_t)(iv->n_docs + 1), 1);
    int32_t* touched = (int32_t*)malloc((size_t)(iv->n_docs + 1) * 4);
    uint64_t* hk = (uint64_t*)malloc((size_t)(L > 0 ? L : 1) * 8);
    int64_t* order = (int64_t*)malloc((size_t)tmax * 8);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      int64_t nt = 0;
      sorted_terms(qix, qip[b], qip[b + 1], order);
      for (int64_t jj = 0; jj < qip[b + 1] - qip[b]; ++jj) {
        const int64_t j = order[jj];
        const int64_t t = find_term(iv, qix[j]);
        if (t < 0) continue;
        const float qw = qv[j];
        for (int64_t i = iv->off[t]; i < iv->off[t + 1]; ++i) {
          const int32_t d = iv->doc[i];
          if (fix >= 0) acc[d] += (int64_t)nearbyint(((double)qw * (double)iv->w[i]) * scale);
          else accf[d] = accf[d] + qw * iv->w[i];      /* -ffp-contract=off: fp32 mul, fp32 add */
          if (!seen[d]) {
            seen[d] = 1;
            touched[nt++] = d;
          }
        }
      }
      heap_t h = {hk, 0, L};
      for (int64_t i = 0; i < nt; ++i) {
        const int32_t d = touched[i];
        const float sc = fix >= 0 ? (float)acc[d] * unscale : accf[d];
        heap_push(&h, make_key(sc, (uint32_t)(id_base + d)));
        acc[d] = 0;
        accf[d] = 0.0f;
        seen[d] = 0;
      }
      emit_sorted(hk, h.n, L, out_s + (int64_t)b * L, out_i + (int64_t)b * L, out_c + b);
    }
    free(acc); free(accf); free(seen); free(touched); free(hk); free(order);
  }
}

/* The same scores document-at-a-time from the doc-major CSR, no inverted index: every document is
 * intersected with every query (an independent second route to the same lists; bench.py uses it to
 * brute-force a query sample over the whole corpus chunk by chunk).  Threads split the documents;
 * each keeps a heap per query, merged at the end. */
void ho_sparse_brute(const int64_t* indptr, const int32_t* idx, const float* val, int64_t n_docs,
                     const int64_t* qip, const int32_t* qix, const float* qv, int B, int L, int64_t id_base,
                     float* out_s, int64_t* out_i, int* out_c) {
  const int fix = g_fix_bits;
  const double scale = fix >= 0 ? ldexp(1.0, fix) : 0.0;
  const float unscale = fix >= 0 ? (float)ldexp(1.0, -fix) : 0.0f;
  const int nt = n_threads();
  const int Lh = L > 0 ? L : 1;
  uint64_t* heaps = (uint64_t*)calloc((size_t)nt * B * Lh, 8);
  int* hn = (int*)calloc((size_t)nt * B, 4);
  int64_t nq = qip[B];
  int64_t* order = (int64_t*)malloc((size_t)(nq + 1) * 8);
  for (int b = 0; b < B; ++b) sorted_terms(qix, qip[b], qip[b + 1], order + qip[b]);
  /* open-addressing set of every query term: most document terms are in no query */
  int64_t cap = 64;
  while (cap < 4 * (nq + 1)) cap <<= 1;
  int32_t* hset = (int32_t*)malloc((size_t)cap * 4);
  for (int64_t i = 0; i < cap; ++i) hset[i] = -1;
  for (int64_t j = 0; j < nq; ++j) {
    uint64_t p = ((uint64_t)(uint32_t)qix[j] * 0x9E3779B97F4A7C15ull) >> 20 & (uint64_t)(cap - 1);
    while (hset[p] != -1 && hset[p] != qix[j]) p = (p + 1) & (uint64_t)(cap - 1);
    hset[p] = qix[j];
  }
#pragma omp parallel
  {
    const int t = omp_get_thread_num();
    int32_t hit_i[4096];
    float hit_w[4096];
#pragma omp for schedule(dynamic, 4096)
    for (int64_t d = 0; d < n_docs; ++d) {
      int nh = 0;
      for (int64_t i = indptr[d]; i < indptr[d + 1] && nh < 4096; ++i) {
        uint64_t p = ((uint64_t)(uint32_t)idx[i] * 0x9E3779B97F4A7C15ull) >> 20 & (uint64_t)(cap - 1);
        while (hset[p] != -1 && hset[p] != idx[i]) p = (p + 1) & (uint64_t)(cap - 1);
        if (hset[p] == idx[i]) { hit_i[nh] = idx[i]; hit_w[nh] = val[i]; ++nh; }
      }
      if (!nh) continue;
      for (int b = 0; b < B; ++b) {
        int64_t ai = 0;
        float af = 0.0f;
        int any = 0;
        for (int64_t jj = qip[b]; jj < qip[b + 1]; ++jj) {      /* ascending term id */
          const int64_t j = order[jj];
          for (int k = 0; k < nh; ++k)
            if (hit_i[k] == qix[j]) {
              if (fix >= 0) ai += (int64_t)nearbyint(((double)qv[j] * (double)hit_w[k]) * scale);
              else af = af + qv[j] * hit_w[k];
              any = 1;
              break;
            }
        }
        if (!any) continue;
        heap_t h = {heaps + ((size_t)t * B + b) * Lh, hn[(size_t)t * B + b], L};
        heap_push(&h, make_key(fix >= 0 ? (float)ai * unscale : af, (uint32_t)(id_base + d)));
        hn[(size_t)t * B + b] = h.n;
      }
    }
  }
  uint64_t* all = (uint64_t*)malloc((size_t)nt * Lh * 8);
  for (int b = 0; b < B; ++b) {
    int n = 0;
    for (int t = 0; t < nt; ++t)
      for (int i = 0; i < hn[(size_t)t * B + b]; ++i) all[n++] = heaps[((size_t)t * B + b) * Lh + i];
    emit_sorted(all, n, L, out_s + (int64_t)b * L, out_i + (int64_t)b * L, out_c + b);
  }
  free(all); free(heaps); free(hn); free(order); free(hset);
}

/* ---- RRF ------------------------------------------------------------------------------ */
void ho_rrf(const int64_t* a, int na, const int64_t* b, int nb, float k, int rank_base, int limit,
            float* out_s, int64_t* out_i, int* out_c) {
  uint64_t* keys = (uint64_t*)malloc((size_t)(na + nb + 1) * 8);
  int m = 0;
  for (int i = 0; i < na; ++i) {
    float s = 0.0f + 1.0f / ((float)(i + rank_base) + k);
    for (int j = 0; j < nb; ++j)
      if (b[j] == a[i]) {
        s = s + 1.0f / ((float)(j + rank_base) + k);
        break;
      }
    keys[m++] = make_key(s, (uint32_t)a[i]);
  }
  for (int j = 0; j < nb; ++j) {
    int dup = 0;
    for (int i = 0; i < na; ++i)
      if (a[i] == b[j]) { dup = 1; break; }
    if (!dup) keys[m++] = make_key(0.0f + 1.0f / ((float)(j + rank_base) + k), (uint32_t)b[j]);
  }
  emit_sorted(keys, m, limit, out_s, out_i, out_c);
  free(keys);
}

/* ---- synthetic data --------------------------------------------------------------------- */
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
static inline uint32_t hash2(uint32_t seed, uint32_t a, uint32_t b) {
  uint32_t h = fmix32(seed + a * 0x9E3779B1u);
  return fmix32(h ^ (b * 0x85EBCA77u));
}
void ho_synth_dense(uint32_t seed, int64_t row0, int64_t n, int dim, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r)
    for (int c = 0; c < dim; ++c) {
      const int32_t h = (int32_t)hash2(seed, (uint32_t)(row0 + r), (uint32_t)c);
      out[r * dim + c] = (float)(h >> 8) * 1.1920928955078125e-07f;
    }
}

static int zipf_rank(const uint32_t* cdf, int V, uint32_t u) {
  int lo = 0, hi = V;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo < V - 1 ? lo : V - 1;
}
static float bm25_w(int tf, int L) {
  const double k = 1.2, b = 0.75, avg = 256.0;
  const double t1 = 1.0 - b;
  const double t2 = (b * (double)L) / avg;
  const double den = (double)tf + k * (t1 + t2);
  return (float)(((double)tf * (k + 1.0)) / den);
}
/* fill=0: nnz_per_doc[n] only; fill=1: idx/val at indptr */
void ho_synth_sparse_docs(uint32_t seed, int64_t doc0, int64_t n, const uint32_t* cdf, int V,
                          const uint16_t* len_tab, int fill, int64_t* nnz_per_doc, const int64_t* indptr,
                          int32_t* idx, float* val) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t d = (uint32_t)(doc0 + i);
    const int L = len_tab[hash2(seed, d, 0xFFFFFFFFu) & 255];
    int prev = -1, tf = 0;
    int64_t nn = 0, o = fill ? indptr[i] : 0;
    for (int t = 0; t <= L; ++t) {
      int rank = -2;
      if (t < L) {
        const uint64_t u = (((uint64_t)t << 32) + hash2(seed, d, (uint32_t)t)) / (uint64_t)L;
        rank = zipf_rank(cdf, V, (uint32_t)u);
      }
      if (rank != prev) {
        if (prev >= 0) {
          if (fill) {
            idx[o] = (int32_t)(((uint32_t)prev * 0x9E3779B1u) & 0x7FFFFFFFu);
            val[o] = bm25_w(tf, L);
            ++o;
          }
          ++nn;
        }
        prev = rank;
        tf = 0;
      }
      ++tf;
    }
    if (!fill) nnz_per_doc[i] = nn;
  }
}
