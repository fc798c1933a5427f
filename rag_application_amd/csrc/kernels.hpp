// Kernel argument blocks and host-side launch prototypes (libhx, gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hx {

enum { KIND_F16 = 0, KIND_I8 = 1, KIND_F32 = 2 };

// Rigorous bound on |fp16-scan score - spec score| for unit-norm rows and queries
// (DESIGN.md "certificate"): input rounding 2*2^-11 + 2^-22, fp32 accumulation
// (D+19)*2^-24 for D <= 4096, plus the subnormal-half terms; 1.25e-3 has > 10 %
// slack over the sum for D <= 4096.
constexpr float HX_EPS_F16 = 1.25e-3f;

// ---- scan.hip ----------------------------------------------------------------
struct ScanArgs {
  const uint8_t* A;        // corpus matrix (fp16 or int8 rows), stride row_bytes
  const uint8_t* Q;        // query matrix [nq_tiles*BN x row_bytes], zero-padded rows
  int64_t row_bytes;       // multiple of 128
  int64_t row_begin, row_end;  // local rows scanned by this launch
  int B;                   // valid queries
  int nq_tiles;
  const float* tau;        // [B] append threshold (score >= tau passes)
  uint64_t* cand;          // [B x cap] candidate keys
  int* cnt;                // [B] appended so far (may exceed cap)
  int* overflow;           // [B] set when an append was dropped
  int cap;
  int64_t id_base;
  const float* rinv_x;     // i8: 1/||x|| per local row (padded)
  const float* rinv_q;     // i8: 1/||q|| per query (padded)
  const float* rinv_tile_max;  // i8, scan8: max of rinv_x over every 256-row tile (index: local row / 256)
  // scan8 only: per-wave append logs.  A passing (key, query) is stored -- fire and forget -- at
  // hitlog[wave * logcap + i]; launch_scatter_log moves the logs into cand/cnt afterwards.
  // hitlog == NULL (or a launch expected to pass most rows) selects the atomic-append kernel.
  uint4* hitlog;           // [SCAN8_WAVES x logcap x SCAN8_ENTRY]: {query, first row lo, hi, 0} + 16 scores
  int* hitcnt;             // [SCAN8_WAVES] entries written (may exceed logcap: the rest set overflow[q])
  int logcap;
  // k_scan only: tau is -inf and rows_end - row_begin <= cap, so every row has its own slot
  // (row - row_begin, key 0 for a NaN score or a row past n_total) and no counter is touched: the caller presets cnt.
  int all_pass;
  // scan8 only: the 256-row x 128-query form (65..128 queries; Q holds 128-row query tiles)
  int half_q;
  int oversub;             // k_scan only: k > 1 = k x the resident grid (less 32 workgroups), each with 1 / k of the static share --
                           // for a scan that runs BESIDE another stream's kernels: its two workgroups per CU take the whole LDS, so
                           // a workgroup of the other stream in the way at launch leaves a scan workgroup waiting for a whole
                           // round of its own kernel (measured: 1.2 -> 2.2 ms); queued shares are placed as others retire
  // Scan order.  [row_begin, row_end) are LOGICAL rows: logical 256-row tile t is physical tile
  // (t * perm_mul) mod perm_n (perm_n = 0: identity).  The stride is about 0.618 of the tile count, so every chunk of
  // the geometric scan is an even sample of the whole matrix: a corpus ingested document by document is topically
  // clustered, and a cluster that sat in ONE chunk would push hundreds of rows past a threshold that was set before
  // the scan reached it (buffer overflow, retry, exact path).  Row ids and validity follow the PHYSICAL row:
  // id = id_base + physical row, valid iff physical row < n_total.
  int64_t n_total;
  uint32_t perm_mul, perm_n;
  double perm_inv;         // 1.0 / perm_n
};
// (t * mul) mod n for t, mul < n < 2^24 (row ids are below 2^32, a tile is 256 rows): the product is exact in a
// double, the quotient estimate is off by at most one -- a handful of instructions instead of a 64-bit division
// inside the scan's tile cursor
__host__ __device__ inline uint32_t scan_phys_tile(uint32_t t, uint32_t mul, uint32_t n, double inv_n) {
  if (!n) return t;
  const double x = (double)t * (double)mul, dn = (double)n;
  double r = x - __builtin_floor(x * inv_n) * dn;
  r = r < 0.0 ? r + dn : r;
  r = r >= dn ? r - dn : r;
  return (uint32_t)r;
}
constexpr int SCAN8_WAVES = 256 * 8;   // waves of the largest scan8 grid
constexpr int SCAN8_LOGCAP = 4096;      // most entries per wave log (expected: a few hundred per launch)
#ifndef HX_S8_TS
#define HX_S8_TS 16                     // MFMA tile side of scan8.hip: 16 (16x16x32) or 32 (32x32x16)
#endif
constexpr int SCAN8_ENTRY = 1 + (64 / HX_S8_TS) * (HX_S8_TS * HX_S8_TS / 64 / 4);   // 16-byte words per log entry
// after_kernel (optional): recorded on `st` right behind the scan kernel itself (before scan8's log scatter)
void launch_scan(const ScanArgs& a, int kind, int bn, hipStream_t st, hipEvent_t after_kernel = nullptr);
// scan8.hip: the 256 x 256 staggered-phase kernel behind launch_scan for large batches
bool scan8_usable(const ScanArgs& a, int bn);
void launch_scan8(const ScanArgs& a, int kind, hipStream_t st, hipEvent_t after_kernel = nullptr);   // includes the log scatter

// ---- select.hip --------------------------------------------------------------
// Sort each query's buffer (first min(cnt, stride) keys) best-first, optionally drop
// duplicate keys, keep `keep`; writes out_keys[b*out_stride + r] (may alias `keys`
// when out_stride == stride), out_cnt[b], and tau[b] = score of the keep-th key when
// the list is full, else -inf (tau may be NULL).  in_cnt NULL => all `stride` slots.
// tau_rank (1-based, default keep): rank whose score becomes tau.  kept_io (optional): per-list
// count kept by the previous compaction of the same buffer, updated; with chk_rank > 0 the
// list is flagged in `underflow` when chk_rank + (keys appended since) < keep.
void launch_compact(uint64_t* keys, int stride, const int* in_cnt, int B, int keep, int dedupe,
                    uint64_t* out_keys, int out_stride, int* out_cnt, float* tau,
                    int max_cnt_hint, hipStream_t st, int tau_rank = 0, int chk_rank = 0,
                    int* kept_io = nullptr, int* underflow = nullptr);

// Exact spec score of listed candidates: out_keys[b*stride + i] = key(spec_dot(row, q_b), id)
// for i < min(cnt[b], stride); ids outside [id_base, id_base+n) give key 0.
struct RescoreArgs {
  int kind;                // KIND_F32 (spec_dot over fp32 rows) or KIND_I8
  const void* M;           // matrix base
  int64_t row_stride;      // elements per row
  int dim_pad;             // padded dim (multiple of 64)
  const void* Q;           // prepared queries: f32 [B x dim_pad] or i8 [B x row_stride]
  int64_t q_stride;
  const float* rinv_x;     // i8 only
  const float* rinv_q;     // i8 only
  int64_t n_rows;
  int64_t id_base;
  const uint64_t* cand;    // [B x stride]
  const int* cnt;          // [B] (NULL => stride)
  int stride;
  int B;
  uint64_t* out;           // [B x stride]
  int max_cnt;             // upper bound of cnt[] (0: stride): sizes the grid; slots beyond it are NOT written
};
void launch_rescore_list(const RescoreArgs& a, hipStream_t st);
// as launch_rescore_list for a list that is mostly OTHER shards' rows: only this shard's slots are scored and written
void launch_rescore_own(const RescoreArgs& a, hipStream_t st);
// exact re-score of the (<= 512) candidates of every query + top-L + the certificate of launch_certify, one launch
// (r.out = scratch [B x stride]; done = [B] zeroed counters, left zero).  Returns false when the sizes do not fit
// (lprime or L above 512): use launch_rescore_list + launch_compact + launch_certify.
bool launch_dense_finish(const RescoreArgs& r, int lprime, int L, uint64_t* out_keys, int* out_cnt, const int* overflow,
                         float eps, const float* eps_q, int* fail, int* nfail, unsigned int* done, hipStream_t st);

// Exact scores of ALL rows [row_begin,row_end) for the listed queries (fallback path):
// out[qsel[f]*stride + slot0 + (row-row_begin)] = key.
struct RangeArgs {
  RescoreArgs r;           // cand/cnt/out/stride reused: out = buffer, stride = its stride
  const int* qsel;         // [nsel] query indices
  int nsel;
  int64_t row_begin, row_end;
  int slot0;
};
void launch_rescore_range(const RangeArgs& a, hipStream_t st);

// fail[b] = overflow[b] || (approx_cnt[b] >= lprime && !(approx_Lprime_score + eps < exact_L_score))
// eps_q (optional): a radius per query instead of the common `eps`
void launch_certify(const uint64_t* approx_keys, int approx_stride, const int* approx_cnt, int lprime,
                    const uint64_t* exact_keys, int exact_stride, const int* exact_cnt, int L,
                    const int* overflow, float eps, int B, int* fail, int* nfail, hipStream_t st,
                    const float* eps_q = nullptr);

void launch_rrf(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride,
                const int* b_cnt, int B, float k, int rank_base, int limit, uint64_t* out,
                int* out_cnt, hipStream_t st);
// fusion + top-`limit` in one launch when both lists together hold at most 256 keys (returns false otherwise: use
// launch_rrf + launch_compact); out [B x out_stride], out_cnt [B]
bool launch_rrf_top(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride, const int* b_cnt,
                    int B, float k, int rank_base, int limit, uint64_t* out, int out_stride, int* out_cnt, hipStream_t st);
void launch_unpack(const uint64_t* keys, int64_t n, float* scores, int64_t* ids, hipStream_t st);
// out[b*(sa+sb) ...] = a-list then b-list (empty slots 0)
void launch_concat(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride,
                   const int* b_cnt, int B, uint64_t* out, hipStream_t st);
void launch_regroup(const uint64_t* g, int world, int B, int dl, int sl, uint64_t* d, uint64_t* s, hipStream_t st);
void launch_scan_init(float* tau, int* cnt, int* ovf, int* kept, int B, int cnt0, hipStream_t st);
void launch_fill_f32(float* p, int64_t n, float v, hipStream_t st);
void launch_fill_i32(int* p, int64_t n, int v, hipStream_t st);
void launch_flag_row(const int* nfail, const int* spsum, uint64_t* row, int len, hipStream_t st);
void launch_flag_add(int* acc, const int* add2, hipStream_t st);   // *acc += add2[0] + add2[1]
// internal id (id_base + local row) <-> global insertion-order id through the index's block table (select.hip)
void launch_remap_ids(const uint64_t* in, uint64_t* out, int64_t n, const uint32_t* row0, const uint32_t* gid0, int nb,
                      uint32_t id_base, uint32_t n_rows, int to_global, hipStream_t st);

// ---- shardx.hip: row-sharded H1, candidates exchanged before the exact scores ----------------
void launch_h1x_pack(const uint64_t* cand, int cstride, const int* cnt, const int* ovf, const float* eps, int complete,
                     int k1, const uint64_t* list, int lstride, const int* lcnt, const int* sflag, const int* sfail, int k2,
                     int lout, float wmax, int B, uint64_t* nom, hipStream_t st);
void launch_h1x_union(const uint64_t* g, int world, int B, int k1, int k2, uint64_t* du, uint64_t* su, hipStream_t st);
void launch_h1x_cuts(const uint64_t* g, int world, int B, int k1, int k2, const uint64_t* G, const int* gc, int lp,
                     const uint64_t* ST, const int* sc, int L_s, const int64_t* q_indptr, uint64_t* meta, uint32_t* thr_out,
                     int* q_margin, int* q_flag, hipStream_t st);
void launch_h1x_counts(const uint64_t* priv_cnt, int B, int* out, hipStream_t st);
void launch_h1x_place(const uint64_t* T, const int* tc, const int* ncand, const int* sp_fail, int B, int k3, int world,
                      int rank, uint64_t* se, uint64_t* nc, uint64_t* meta, hipStream_t st);
void launch_h1x_certify(const uint64_t* red, int world, int B, int lp, int k3, const uint64_t* D, const int* Dc, int L,
                        const uint64_t* S, const int* Sc, int L_s, int* fail, int* nfail, hipStream_t st);

// ---- prep.hip ----------------------------------------------------------------
// Derive the stored vectors of rows [0,n) of `raw` (fp32 [n x dim]):
struct PrepRowsArgs {
  const float* raw;        // [n x dim]
  int dim, dim_pad;        // dim_pad = round_up(dim, 64)
  int64_t n;
  float* dense;            // [n x dim_pad] L2-normalised (zero padded)
  _Float16* dense_h;       // [n x dim_pad]
  int8_t* q8;              // [n x dim_pad8] trunc(127*x) of the RAW row
  int dim_pad8;            // round_up(dim, 128)
  float* q8_rinv;          // [n]
  int n_prefix;
  int psize[3];
  float* pre[3];           // [n x psize] normalised prefixes of the RAW row
  _Float16* pre_h0;        // fp16 copy of prefix 0 (may be NULL)
  // candidate-pass copy of the NORMALISED row (may be NULL): rint(x * 127 / max|x|), its scale max|x| / 127, and
  // the running maximum over all rows of the quantisation error ||x - scale * x8||_2 (fp32 bits, rounded up)
  int8_t* q8s;             // [n x dim_pad8]
  float* q8s_scale;        // [n]
  uint32_t* err_max;       // one device word
};
void launch_prep_rows(const PrepRowsArgs& a, hipStream_t st);
void launch_requant_rows(const float* dense, int dim_pad, int dim_pad8, int64_t n, int8_t* q8s, float* scale,
                         uint32_t* err_max, hipStream_t st);
// candidate-pass form of normalised queries qn [B x dpad]: int8 rows [Bpad x dpad8], scales [Bpad], certificate
// radius per query [B] from the index's largest row quantisation error (*err_max)
void launch_prep_queries_s8(const float* qn, int dpad, int B, int Bpad, int dpad8, int8_t* q8, float* sq, float* eps,
                            const uint32_t* err_max, hipStream_t st);
// out[t] = max of the non-negative floats p[256 t, 256 t + 256) (clipped to n)
void launch_tile_max(const float* p, int64_t n, float* out, hipStream_t st);
void launch_synth_dense(float* raw, int64_t row0_global, int64_t n, int dim, uint32_t seed, hipStream_t st);

// Prepare a query batch for one named vector: normalised fp32 [B x dpad] (+ fp16
// zero-padded to Bpad rows) from raw q[:, :d]; or the int8 copy + rinv.
void launch_prep_queries_f(const float* q_raw, int q_dim, int B, int Bpad, int d, int dpad,
                           float* qn, _Float16* qh, hipStream_t st);
void launch_prep_queries_i8(const float* q_raw, int q_dim, int B, int Bpad, int dpad8, int8_t* q8,
                            float* rinv_q, hipStream_t st);

// ---- sparse2.hip / sprescore.hip ------------------------------------------------
// K7 in two passes (DESIGN.md "sparse stage"):
//   select  -- term-at-a-time over the inverted index with a 16-bit integer accumulator per document
//              (two documents per LDS word): segments of 32768 / 65536 documents per visit.  The integer
//              score `a` brackets the real one (u = score * scale, a - 1.01 k <= u <= a + 0.01 k for k
//              matching terms), so every document that can reach the exact top-L has a > a_L - M
//              (a_L = L-th best integer score, M = the query's margin): the pass keeps exactly those;
//   rescore -- the exact score of every kept candidate from the document-major CSR in upstream's order
//              (query terms in ascending term id, fp32 mul, fp32 add), then the top-L by key.
// Queries the integer pass cannot serve (a non-positive weight, more than SP_TMAX terms, a candidate
// buffer that overflowed) are flagged and re-run document-at-a-time over all rows (k_sparse_range).
constexpr int SEG_DOCS_SMALL = 32768, SEG_DOCS_LARGE = 65536;
constexpr int SP_TMAX = 64;            // query terms the select pass takes (one lane each)
// Term-major inverted index over the documents [doc0, doc0 + n_docs) of the shard: the postings of live
// term i (ascending document) are post[ptr[i*(n_segments+1) + 0] .. ptr[i*(n_segments+1) + n_segments]);
// ptr[i*(S+1) + s] is the first posting of term i whose document lies in segment s or later.  A posting
// is {the document's place in the select pass's accumulator: byte offset of its 32-bit LDS word (document index
// inside the segment mod seg_docs/2, times 4) | shift of its 16-bit half (0 / 16) << 24, fp32 weight bits}.
struct SparseIndexView {
  const uint2* post;           // [nnz] (+ 16 bytes of padding)
  const uint32_t* ptr;         // [n_live x (n_segments + 1)]
  const uint32_t* uterms;      // [n_live] live term ids, ascending
  int n_live;
  int64_t n_docs;
  int n_segments;
  int seg_docs;                // SEG_DOCS_SMALL or SEG_DOCS_LARGE
  int64_t id_base;             // global id of the view's first document
};
// per-query preparation, one wave per query: validity, scale, margin, live-term index of every term
struct SparsePrepArgs {
  const int64_t* q_indptr;     // [B+1]
  const int32_t* q_idx;        // ascending within a query
  const float* q_val;
  int B;
  SparseIndexView ix[2];       // base index and (optional, n_live = 0) tail index
  float wmax;                  // largest document weight of the shard
  int index_nonpos;            // some document weight is <= 0: the integer pass cannot bracket scores
  int32_t* q_ti[2];            // [B x SP_TMAX] live-term index per view, -1 = absent
  float* q_qs;                 // [B x SP_TMAX] query weight * scale
  int* q_margin;               // [B]
  int* q_flag;                 // [B] 0 ok, 1 = needs the document-at-a-time path, 2 = invalid (non-finite weight),
                               //     3 = invalid (term ids not strictly ascending, or negative)
  unsigned long long* q_work;  // [B] postings of the query's terms (both views): launch order, profile
  unsigned long long* stat_postings;   // optional: += sum of q_work
};
void launch_sparse_prep(const SparsePrepArgs& a, hipStream_t st);
struct SparseSelectArgs {
  SparseIndexView ix;
  const int64_t* q_indptr;
  const int32_t* q_ti;         // [B x SP_TMAX] for THIS view
  const float* q_qs;
  const int* q_margin;
  const int* q_flag;
  int B;
  int parts;                   // most workgroups per query; query q is cut into q_parts[q] of them
  const int* q_parts;          // [B] in [1, parts] (NULL: all `parts`): each owns a contiguous segment range
  const int* items;            // launch plan (NULL: workgroup b = query b / parts, part b % parts): query << 8 | part
  const int* n_items;
  int limit;
  int lout;                    // keys kept per (query, part) at most
  uint64_t* out;               // [B x parts_total x lout] integer-score keys, best first, 0 = empty
  int* out_cnt;                // [B x parts_total]
  int parts_total, part0;      // this launch writes parts [part0, part0 + parts) of every query
  uint64_t* cand;              // [B x parts x (seg_docs + seg_docs/8)] workgroup-private candidate buffers
  int* q_fail;                 // [B] set when a buffer overflowed: the query takes the exact path
  int cut_step;                // keys the buffer grows by between cuts (0: max(1024, 4 * limit))
};
void launch_sparse_select(const SparseSelectArgs& a, hipStream_t st);   // dispatches on a.ix.seg_docs
void launch_sparse_summary(const int* flag, const int* fail, int B, int* out, hipStream_t st);
namespace v32k { void launch_sparse_select_variant(const SparseSelectArgs& a, hipStream_t st); }
namespace v64k { void launch_sparse_select_variant(const SparseSelectArgs& a, hipStream_t st); }
// q_parts[q] in [1, pt_max]: workgroups the query is cut into; items[0, *n_items): query << 8 | part, heaviest first
void launch_sparse_plan(const unsigned long long* q_work, int B, int pt_max, int slots, int* q_parts, int* items,
                        int* n_items, hipStream_t st);
// document-major CSR of the shard (what the ingest path appends)
struct SparseCsr {
  const int64_t* indptr;       // [n_docs + 1]
  const int32_t* idx;
  const float* val;
  int64_t n_docs;
  int64_t id_base;
};
// exact (upstream-order fp32) scores of the kept candidates of every query
struct SparseRescoreArgs {
  SparseCsr d;
  const int64_t* q_indptr;
  const int32_t* q_idx;
  const float* q_val;
  const uint64_t* cand;        // [B x stride] integer-score keys, best first
  const int* cnt;              // [B]
  int stride;
  int B;
  int limit;
  const int* q_margin;
  const int* q_flag;
  uint64_t* out;               // [B x stride] exact keys of the candidates: slots [0, out_cnt[b]) (0 = foreign id)
  int* out_cnt;                // [B] candidates = the prefix of the list within the margin of its L-th key
  int* q_fail;                 // set when the list was cut short of the margin
  int blocks;                  // workgroups (of 4 waves) per query; 0: 32 (a list of ~110 candidates, more when the L-th ties)
  const uint32_t* thr_in;      // optional [B]: the integer-score threshold to apply instead of the list's own
                               // (a shard of the candidates-first exchange applies the GLOBAL one: shardx.hip)
};
void launch_sparse_rescore(const SparseRescoreArgs& a, hipStream_t st);
// out[b] = the parts' lists of query b packed into one run (stride pt * lout), out_cnt[b] = its length
void launch_sparse_pack(const uint64_t* parts, const int* pcnt, int B, int pt, int lout, uint64_t* out, int* out_cnt,
                        hipStream_t st);
// exact scores of ALL documents [row_begin, row_end) for the listed queries (document-at-a-time path).
// tau == NULL: out[f*stride + slot0 + (row - row_begin)] = key, 0 when the document shares no term with the query.
// tau != NULL: keys with score >= tau[f] are appended at out[f*stride + atomicAdd(cnt[f], 1)]; ovf[f] = 1 when full.
struct SparseRangeArgs {
  SparseCsr d;
  const int64_t* q_indptr;
  const int32_t* q_idx;
  const float* q_val;
  const int* qsel;
  int nsel;
  int64_t row_begin, row_end;
  uint64_t* out;
  int stride, slot0;
  const float* tau;
  int* cnt;
  int* ovf;
};
void launch_sparse_range(const SparseRangeArgs& a, hipStream_t st);
// {min, max} of val[0, n) merged into mm[0], mm[1] (fp32, device); mm[2] counts non-finite values
void launch_minmax_f32(const float* val, int64_t n, float* mm, hipStream_t st);
// n_rows rows of a device CSR whose offsets start at indptr_rows: *bad |= 1 unless indptr_rows[0] = first, the offsets
// are monotone and indptr_rows[n_rows] = last; *bad |= 2 if an idx in [first, last) is negative
void launch_csr_check(const int64_t* indptr_rows, const int32_t* idx, int64_t n_rows, int64_t first, int64_t last, int* bad,
                      hipStream_t st);
// term ids unique within every row (what hx_add_sparse enforces and k_sparse_rescore's one-ballot-per-term relies
// on): *bad |= 4 on a duplicate.  Rows longer than CSR_UNIQUE_WAVE_MAX terms are not compared on the device: their
// indices go to long_rows[0, *n_long) (at most long_cap are listed; *n_long keeps counting) for the host to check.
constexpr int CSR_UNIQUE_WAVE_MAX = 2048;
void launch_csr_unique(const int64_t* indptr, const int32_t* idx, int64_t n_rows, int* bad, int64_t* long_rows,
                       int long_cap, int* n_long, hipStream_t st);

// ---- spbuild.hip -------------------------------------------------------------
struct SparseBuildOut {
  uint2* post;
  uint32_t* ptr;
  uint32_t* uterms;
  int64_t n_live;
  int64_t ptr_entries;
};
// Build the term-major inverted index of documents [doc0, doc0 + n_docs) from the doc-major CSR on the
// device (indptr is the shard's: indptr[doc0] is the first posting taken).
void build_sparse_index(const int64_t* indptr, const int32_t* idx, const float* val, int64_t doc0, int64_t n_docs,
                        int seg_docs, SparseBuildOut* out, hipStream_t st);
// Synthetic docs (oracle synth_sparse_docs): two passes.
void synth_sparse_count(int64_t doc0_global, int64_t n, uint32_t seed, const uint32_t* cdf, int V,
                        const uint16_t* len_tab, int64_t* nnz_per_doc, hipStream_t st);
void synth_sparse_fill(int64_t doc0_global, int64_t n, uint32_t seed, const uint32_t* cdf, int V,
                       const uint16_t* len_tab, const int64_t* indptr, int32_t* idx, float* val,
                       hipStream_t st);
void exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t n, hipStream_t st);  // out[n] = total

}  // namespace hx
