// Host-side engine: device storage for one collection, stage orchestration, C ABI.
//
// The stage functions restate the nested-Prefetch query of the reference
// (app/core/vector_store/qdrant/qdrant_handler.py:296-372) as launches on one HIP
// stream.  See include/hx.h for the boundary and DESIGN.md for the algorithms.
#include "../../include/hx.h"
#include "hx_common.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <cmath>
#include <numeric>
#include <vector>

namespace hx {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

struct Workspace {
  std::map<int, std::pair<void*, size_t>> slots;
  void* get(int id, size_t bytes) {
    auto& s = slots[id];
    if (s.second < bytes) {
      if (s.first) HX_HIP(hipFree(s.first));
      s.first = nullptr;
      s.second = 0;
      const size_t want = (bytes + 255) / 256 * 256;
      HX_HIP(hipMalloc(&s.first, want));
      s.second = want;
    }
    return s.first;
  }
  void release() {
    for (auto& kv : slots)
      if (kv.second.first) (void)hipFree(kv.second.first);
    slots.clear();
  }
};

// Scratch of the index-free entries (hx_rrf, hx_merge, hx_h1_fuse): one Workspace per (device, calling thread), so
// two threads fusing lists on one device never share buffers.  A thread that exits hands its workspaces to a pool
// the next new thread takes them from: executor threads come and go, the device memory they used does not pile up
// -- and nothing calls hipFree from a thread-exit (or process-exit) destructor, where the runtime may be gone.
struct WorkspacePool {
  std::mutex mu;
  std::map<int, std::vector<Workspace*>> idle;   // device -> workspaces without a thread
  Workspace* take(int device) {
    std::lock_guard<std::mutex> g(mu);
    auto& v = idle[device];
    if (v.empty()) return new Workspace();
    Workspace* w = v.back();
    v.pop_back();
    return w;
  }
  void give(int device, Workspace* w) {
    std::lock_guard<std::mutex> g(mu);
    idle[device].push_back(w);
  }
};
static WorkspacePool& ws_pool() {
  static WorkspacePool* p = new WorkspacePool();   // never destroyed: threads may exit after main returns
  return *p;
}
struct ThreadWorkspaces {
  std::map<int, Workspace*> by_device;
  Workspace& get(int device) {
    auto it = by_device.find(device);
    if (it == by_device.end()) it = by_device.emplace(device, ws_pool().take(device)).first;
    return *it->second;
  }
  ~ThreadWorkspaces() {
    for (auto& kv : by_device) ws_pool().give(kv.first, kv.second);
  }
};

enum WsSlot {
  WS_QN = 1, WS_QH, WS_Q8, WS_RINVQ, WS_CAND, WS_CAND2, WS_CNT, WS_CNT2, WS_OVF, WS_TAU, WS_FAIL,
  WS_NFAIL, WS_QSEL, WS_FB_KEYS, WS_FB_CNT, WS_FB_INCNT, WS_SP_PARTS, WS_SP_PCNT, WS_RAW, WS_RS_TMP,
  WS_T_A, WS_T_ACNT, WS_T_B, WS_T_BCNT, WS_T_C, WS_T_CCNT, WS_T_D, WS_T_DCNT, WS_T_E, WS_T_ECNT,
  WS_T_F, WS_T_FCNT, WS_T_G, WS_T_GCNT, WS_H_QD, WS_H_QIP, WS_H_QIX, WS_H_QV, WS_H_OUT, WS_H_OCNT,
  WS_H_SC, WS_H_ID, WS_SYN_NNZ, WS_RRF_TMP, WS_MISC, WS_SP_CAND, WS_SP_PARK, WS_SP_ORDER, WS_HITLOG, WS_HITCNT, WS_KEPT,
  WS_F_DALL, WS_F_SALL, WS_F_D, WS_F_S, WS_F_DC, WS_F_SC,
  WS_SP_TI0, WS_SP_TI1, WS_SP_QS, WS_SP_MARGIN, WS_SP_FLAG, WS_SP_WORK, WS_SP_FAIL, WS_SP_LIST, WS_SP_LCNT,
  WS_SP_EXACT, WS_SP_ECNT, WS_SP_MM, WS_SP_FTAU, WS_SP_FOVF, WS_SP_QPARTS, WS_SP_SUM,
  WS_ID_IN, WS_LONG_ROWS, WS_Q8S, WS_SQ, WS_EPSQ, WS_TREE_FLAG, WS_DONE
};

template <typename T>
static void grow_copy(T*& p, int64_t old_elems, int64_t new_elems, bool zero_tail) {
  T* np_ = nullptr;
  HX_HIP(hipMalloc((void**)&np_, (size_t)std::max<int64_t>(new_elems, 1) * sizeof(T)));
  if (zero_tail) HX_HIP(hipMemset(np_, 0, (size_t)std::max<int64_t>(new_elems, 1) * sizeof(T)));
  if (p && old_elems > 0) HX_HIP(hipMemcpy(np_, p, (size_t)old_elems * sizeof(T), hipMemcpyDeviceToDevice));
  if (p) HX_HIP(hipFree(p));
  p = np_;
}

}  // namespace hx

using namespace hx;

struct hx_index {
  int dim = 0, dim_pad = 0, dim_pad8 = 0, device = 0;
  int n_pre = 0;
  int psize[3] = {0, 0, 0};
  int64_t id_base = 0;
  int64_t n = 0, cap = 0;
  float* dense = nullptr;
  _Float16* dense_h = nullptr;
  int8_t* q8 = nullptr;
  float* q8_rinv = nullptr;
  float* pre[3] = {nullptr, nullptr, nullptr};
  _Float16* pre_h0 = nullptr;
  // Candidate-pass copy of the normalised rows (DESIGN.md "int8 candidate pass"): rint(x * 127 / max|x|), the
  // row's scale, and -- one device word -- the largest quantisation error ||x - scale * x8|| of any row so far
  // (fp32 bits, rounded up; a rollback leaves it as it is: an upper bound is all the certificate needs).
  int8_t* q8s = nullptr;
  float* q8s_scale = nullptr;
  uint32_t* s8_err = nullptr;
  int cand8 = 1;                      // 0: fp16 candidates only, no int8 copy is kept (HX_DENSE_CAND=f16)
  bool cand8_off = false;             // hx_set_dense_candidates(h, 0): the copy is kept but the fp16 scan nominates
  int64_t cand8_queries = 0, cand8_failed = 0;   // queries the int8 candidate pass took / could not certify
  int64_t tree_redone = 0;            // tree batches run again the synchronous way (a deferred flag was set)
  // Adaptive guards of the two speculative paths (hx_stats): windows of recent outcomes
  int64_t c8_win_q = 0, c8_win_f = 0; // queries / uncertified queries of the current window of the int8 candidate pass
  bool cand8_auto_off = false;        // ... switched off by the guard (hx_set_dense_candidates(h, 1) switches it on again)
  int tree_win_n = 0, tree_win_redone = 0;
  bool tree_spec_off = false;
  int done_zeroed[2] = {0, 0};        // entries of the finish kernel's per-query counters known to be zero (level 0 / retry level)
  // query-tile routing of the scans (read from the environment at hx_create: tests and diagnostics):
  //   B <= 32: k_scan 128 x 32; <= bn64_max: k_scan 128 x 64; <= bn128_max: 128 queries per tile -- the staggered kernel's
  //   256 x 128 form (scan8.hip, HQ) unless no_hq, then k_scan 128 x 128; above: the staggered 256 x 256 kernel
  int bn32_max = 32, bn64_max = 32, bn128_max = 128;
  // second stream of the one-call H1 step: the sparse stage runs beside the dense stage's tail (hybrid_query_dev)
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap_tail = true;           // HX_DEBUG_NO_OVERLAP (diagnostics): everything on the caller's stream
  bool beside = false;                // a stage of this call runs on the second stream while the dense scans run
  int scan_oversub = 4;               // ScanArgs.oversub of those scans (HX_DEBUG_SCAN_OVERSUB)
  int sp_host_late_min_b = 128;       // batches from here on: the host enqueues the dense scans before the sparse stage (HX_DEBUG_SP_HOST_LATE_MIN_B)
  int fork_early_max = 1 << 30;       // batches of at most this many queries start the sparse stage beside the dense SCAN
                                      // (HX_DEBUG_FORK_EARLY_MAX=0: beside the stage's tail only, as round 4 began)
  bool no_hq = false;
  // doc-major sparse staging (device)
  int64_t* sp_indptr = nullptr;  // [sp_rows_cap + 1]
  int32_t* sp_idx = nullptr;
  float* sp_val = nullptr;
  int64_t sp_rows = 0, sp_rows_cap = 0, nnz = 0, nnz_cap = 0;
  bool sparse_stale = false;
  // Inverted index = an immutable BASE over documents [0, base.n_docs) plus a TAIL over the documents added
  // since (rebuilt on every finalize from the tail's postings only); the base is rebuilt once the tail has
  // grown to a quarter of it.  Both are searched by the same select kernel; the exact scores come from the
  // document-major CSR, which covers every document.
  struct SparseIx {
    SparseBuildOut sp{};
    int64_t doc0 = 0, n_docs = 0;
    int n_segments = 0;
    int seg_docs = SEG_DOCS_SMALL;
  };
  SparseIx sp_base, sp_tail;
  int seg_docs_force = 0;             // HX_DEBUG_SEG_DOCS (tests): 32768 / 65536, 0 = by size
  int sp_cut_step = 0;                // HX_DEBUG_SP_CUTSTEP (diagnostics): keys between cuts of the select pass, 0 = default
  int64_t tail_min_force = -1;        // HX_DEBUG_TAIL_MIN (tests): documents a tail may hold before the base is rebuilt
  float sp_wmin = 0.f, sp_wmax = 0.f; // range of the document weights (all finite: checked at ingest)
  float sp_wmax_shared = 0.f;         // hx_set_sparse_wmax: the largest weight of any shard of the collection (0: unset)
  bool sp_have_w = false;
  int64_t sparse_fallbacks = 0;       // queries served by the document-at-a-time path
  Workspace ws;
  int64_t dense_fallbacks = 0, i8_fallbacks = 0, retries = 0;
  // max of a per-row scale over every 256-row tile: the int8 scans' column bound (scan8.hip)
  struct TileMax {
    float* v = nullptr;
    int64_t rows = -1, cap = 0;
  };
  TileMax tm_q8, tm_q8s;            // of q8_rinv (the "quantized" stage) and of q8s_scale (the candidate pass)
  int scan_logcap = SCAN8_LOGCAP;   // entries per wave log (HX_DEBUG_SCAN8_LOGCAP shrinks it: tests)
  // optional HIP-event profile of the scan / sparse kernels (hx_profile)
  struct ProfRec { hipEvent_t a, b; int what; double flops, bytes; };
  bool prof = false;
  std::vector<ProfRec> prof_recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  unsigned long long* sp_counter = nullptr;   // postings of the queries' terms (k_sparse_prep) while profiling
  // One host round trip per batch for all the stages' failure flags: the small counters travel into this pinned
  // buffer ([0] a dense stage's failure count, [1..2] the sparse stage's summary) behind ONE synchronisation.
  int* pin = nullptr;
  bool sp_sum_pending = false;        // a sparse_enqueue's summary has not been consumed by sparse_resolve yet
  bool sp_sum_fetched = false;        // ... and is already in sp_sum (a dense stage's flag read took it along)
  int sp_sum[2] = {0, 0};             // flagged-or-failed queries, any invalid query

  // Insertion-order ids under row sharding.  Inside the engine a row's id is id_base + local
  // row.  What crosses the ABI is the row's GLOBAL id: rows arrive in blocks, block k = local rows
  // [row0[k], row0[k + 1]) with global ids gid0[k], gid0[k] + 1, ... (hx_set_next_id names the first id of the next
  // block; without it a block continues the previous one).  Both columns ascend, so the map is monotone: the order
  // (score desc, id asc) of a list is the same in both id spaces, and the remap is one pass over the keys a stage
  // hands out (and over the candidate keys hx_rescore takes in).  One block starting at id_base = the identity.
  struct IdBlocks {
    std::vector<uint32_t> row0, gid0;
    uint32_t* dev = nullptr;          // row0[0, nb) then gid0[0, nb)
    int dev_cap = 0;
    bool dirty = false;
    int64_t next = -1;                // hx_set_next_id: global id of the next appended row (-1: continue)
    int64_t end = -1;                 // global id after the last row (-1: no row yet)
  } ids;

  void set_device() const { HX_HIP(hipSetDevice(device)); }
};

namespace hx {

void launch_sparse_select(const SparseSelectArgs& a, hipStream_t st) {
  if (a.ix.seg_docs == SEG_DOCS_LARGE) v64k::launch_sparse_select_variant(a, st);
  else v32k::launch_sparse_select_variant(a, st);
}

// ---------------------------------------------------------------------------------
// storage
// ---------------------------------------------------------------------------------
static void reserve_rows(hx_index* h, int64_t want) {
  if (want <= h->cap) return;
  int64_t nc = std::max<int64_t>(want, h->cap * 2);
  nc = round_up(nc, 256);
  const int64_t n = h->n;
  grow_copy(h->dense, n * h->dim_pad, nc * h->dim_pad, false);
  grow_copy(h->dense_h, n * h->dim_pad, nc * h->dim_pad, false);
  grow_copy(h->q8, n * h->dim_pad8, nc * h->dim_pad8, false);
  grow_copy(h->q8_rinv, n, nc + 256, true);
  if (h->cand8) {
    grow_copy(h->q8s, n * h->dim_pad8, nc * h->dim_pad8, false);
    grow_copy(h->q8s_scale, n, nc + 256, true);
    if (!h->s8_err) {
      HX_HIP(hipMalloc((void**)&h->s8_err, 4));
      HX_HIP(hipMemset(h->s8_err, 0, 4));
    }
  }
  for (int p = 0; p < h->n_pre; ++p) grow_copy(h->pre[p], n * h->psize[p], nc * h->psize[p], false);
  if (h->n_pre > 0) grow_copy(h->pre_h0, n * h->psize[0], nc * h->psize[0], false);
  h->cap = nc;
}

static void reserve_sparse(hx_index* h, int64_t rows, int64_t nnz) {
  if (rows > h->sp_rows_cap) {
    int64_t nc = std::max<int64_t>(rows, h->sp_rows_cap * 2);
    nc = round_up(nc, 256);
    grow_copy(h->sp_indptr, h->sp_rows + 1, nc + 2, true);
    h->sp_rows_cap = nc;
  }
  if (nnz > h->nnz_cap) {
    int64_t nc = std::max<int64_t>(nnz, h->nnz_cap * 2);
    nc = round_up(nc, 1024);
    grow_copy(h->sp_idx, h->nnz, nc, false);
    grow_copy(h->sp_val, h->nnz, nc, false);
    h->nnz_cap = nc;
  }
}

struct ProfScope {
  hx_index* h;
  hipStream_t st;
  hx_index::ProfRec rec{};
  bool on;
  ProfScope(hx_index* h_, hipStream_t st_, int what, double flops, double bytes, bool enable = true)
      : h(h_), st(st_), on(h_->prof && enable) {
    if (!on) return;
    if (h->prof_pool.empty()) {
      hipEvent_t a, b;
      HX_HIP(hipEventCreate(&a));
      HX_HIP(hipEventCreate(&b));
      h->prof_pool.emplace_back(a, b);
    }
    rec.a = h->prof_pool.back().first;
    rec.b = h->prof_pool.back().second;
    h->prof_pool.pop_back();
    rec.what = what;
    rec.flops = flops;
    rec.bytes = bytes;
    HX_HIP(hipEventRecord(rec.a, st));
  }
  bool ended = false;        // rec.b was recorded by the callee (launch_scan: right behind the scan kernel)
  hipEvent_t end_event() {
    if (!on) return nullptr;
    ended = true;
    return rec.b;
  }
  ~ProfScope() {
    if (!on) return;
    if (!ended) (void)hipEventRecord(rec.b, st);
    h->prof_recs.push_back(rec);
  }
};

static void prep_rows_device(hx_index* h, const float* raw_dev, int64_t n, hipStream_t st) {
  PrepRowsArgs a{};
  a.raw = raw_dev;
  a.dim = h->dim;
  a.dim_pad = h->dim_pad;
  a.n = n;
  a.dense = h->dense + h->n * h->dim_pad;
  a.dense_h = h->dense_h + h->n * h->dim_pad;
  a.q8 = h->q8 + h->n * h->dim_pad8;
  a.dim_pad8 = h->dim_pad8;
  a.q8_rinv = h->q8_rinv + h->n;
  a.n_prefix = h->n_pre;
  for (int p = 0; p < h->n_pre; ++p) {
    a.psize[p] = h->psize[p];
    a.pre[p] = h->pre[p] + h->n * h->psize[p];
  }
  a.pre_h0 = h->n_pre > 0 ? h->pre_h0 + h->n * h->psize[0] : nullptr;
  const bool c8 = h->cand8 && h->q8s && h->q8s_scale && h->s8_err;   // (never an offset from a null base)
  HX_CHECK(!h->cand8 || c8, "int8 candidate copy enabled but not allocated");
  a.q8s = c8 ? h->q8s + h->n * h->dim_pad8 : nullptr;
  a.q8s_scale = c8 ? h->q8s_scale + h->n : nullptr;
  a.err_max = c8 ? h->s8_err : nullptr;
  // K1/K2 of SURVEY 8(d): the raw row read once, every derived copy written once
  double per_row = (double)h->dim * 4 + (double)h->dim_pad * 6 + (double)h->dim_pad8 * (h->cand8 ? 2 : 1) + (h->cand8 ? 8 : 4);
  for (int p = 0; p < h->n_pre; ++p) per_row += (double)h->psize[p] * 4;
  if (h->n_pre > 0) per_row += (double)h->psize[0] * 2;
  ProfScope ps(h, st, 4, 0.0, per_row * (double)n);
  launch_prep_rows(a, st);
}

static void free_sparse_ix(hx_index::SparseIx& x) {
  if (x.sp.post) (void)hipFree(x.sp.post);
  if (x.sp.ptr) (void)hipFree(x.sp.ptr);
  if (x.sp.uterms) (void)hipFree(x.sp.uterms);
  x = hx_index::SparseIx{};
}
static void free_sparse_index(hx_index* h) {
  free_sparse_ix(h->sp_base);
  free_sparse_ix(h->sp_tail);
}

// merge the range of val[from, to) (device) into the index's weight range; non-finite values are refused
static void track_weights_dev(hx_index* h, const float* val, int64_t n, hipStream_t st) {
  if (n <= 0) return;
  uint32_t* mm = (uint32_t*)h->ws.get(WS_SP_MM, 16);
  const uint32_t init[3] = {0xFFFFFFFFu, 0u, 0u};
  HX_HIP(hipMemcpyAsync(mm, init, 12, hipMemcpyHostToDevice, st));
  launch_minmax_f32(val, n, (float*)mm, st);
  uint32_t got[3];
  HX_HIP(hipMemcpyAsync(got, mm, 12, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  HX_CHECK(got[2] == 0, "sparse values must be finite");
  if (got[0] > got[1]) return;
  const float lo = orderable_f32(got[0]), hi = orderable_f32(got[1]);
  h->sp_wmin = h->sp_have_w ? std::min(h->sp_wmin, lo) : lo;
  h->sp_wmax = h->sp_have_w ? std::max(h->sp_wmax, hi) : hi;
  h->sp_have_w = true;
}

static void build_ix(hx_index* h, hx_index::SparseIx& x, int64_t doc0, int64_t n_docs, int seg_docs, hipStream_t st) {
  free_sparse_ix(x);
  if (n_docs <= 0) return;
  build_sparse_index(h->sp_indptr, h->sp_idx, h->sp_val, doc0, n_docs, seg_docs, &x.sp, st);
  if (!x.sp.post) return;             // no posting in the range
  x.doc0 = doc0;
  x.n_docs = n_docs;
  x.seg_docs = seg_docs;
  x.n_segments = (int)((n_docs + seg_docs - 1) / seg_docs);
}

static void finalize(hx_index* h, hipStream_t st) {
  if (!h->sparse_stale) return;
  if (h->nnz > 0) {
    // rows added without a sparse vector are empty documents
    const int64_t rows = std::max(h->sp_rows, h->n);
    if (rows > h->sp_rows) {
      reserve_sparse(h, rows, h->nnz);
      std::vector<int64_t> tail((size_t)(rows - h->sp_rows), h->nnz);
      HX_HIP(hipMemcpy(h->sp_indptr + h->sp_rows + 1, tail.data(), tail.size() * 8, hipMemcpyHostToDevice));
      h->sp_rows = rows;
    }
    const int64_t built = h->sp_base.n_docs;
    const int64_t fresh = h->sp_rows - built;
    const int64_t tail_max = h->tail_min_force >= 0 ? h->tail_min_force : std::max<int64_t>(built / 4, 1 << 18);
    if (built == 0 || fresh > tail_max) {
      // fewer, larger visits pay once a workgroup has enough segments to walk (kernels.hpp)
      const int sd = h->seg_docs_force ? h->seg_docs_force : (h->sp_rows >= 3000000 ? SEG_DOCS_LARGE : SEG_DOCS_SMALL);
      free_sparse_ix(h->sp_tail);
      build_ix(h, h->sp_base, 0, h->sp_rows, sd, st);
      if (!h->sp_base.sp.post) h->sp_base.n_docs = 0;
    } else if (fresh > 0) {
      // incremental upsert (qdrant_handler.py:190-193 upserts per document): only the tail is sorted
      build_ix(h, h->sp_tail, built, fresh, h->seg_docs_force ? h->seg_docs_force : SEG_DOCS_SMALL, st);
    }
  } else {
    free_sparse_index(h);
  }
  h->sparse_stale = false;
}

// ---------------------------------------------------------------------------------
// global (insertion-order) ids: hx_index::IdBlocks
// ---------------------------------------------------------------------------------
static bool ids_identity(const hx_index* h) {
  const auto& b = h->ids;
  return b.row0.empty() || (b.row0.size() == 1 && (int64_t)b.gid0[0] == h->id_base);
}
// first global id the next n rows would get; refuses before anything is stored
static int64_t ids_next(const hx_index* h, int64_t n) {
  const auto& b = h->ids;
  const int64_t cont = b.end >= 0 ? b.end : h->id_base;
  const int64_t g = b.next >= 0 ? b.next : cont;
  HX_CHECK(g >= cont, "hx_set_next_id: the ids of a shard must ascend with its rows");
  HX_CHECK(g + n < 0xFFFFFFFFll, "row ids must stay below 2^32 - 1");
  return g;
}
static void ids_commit(hx_index* h, int64_t row_first, int64_t n) {
  auto& b = h->ids;
  if (n <= 0) return;
  const int64_t cont = b.end >= 0 ? b.end : h->id_base;
  const int64_t g = ids_next(h, n);
  if (b.row0.empty() || g != cont) {
    b.row0.push_back((uint32_t)row_first);
    b.gid0.push_back((uint32_t)g);
    b.dirty = true;
  }
  b.end = g + n;
  b.next = -1;
}
// the table on the device: a fresh allocation per change (hipFree waits for whatever still reads the old one)
static void ids_upload(hx_index* h) {
  auto& b = h->ids;
  if (!b.dirty) return;
  const int nb = (int)b.row0.size();
  uint32_t* nd = nullptr;
  const int cap = std::max(nb, 1);
  HX_HIP(hipMalloc((void**)&nd, (size_t)cap * 8));
  if (nb) {
    HX_HIP(hipMemcpy(nd, b.row0.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
    HX_HIP(hipMemcpy(nd + cap, b.gid0.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
  }
  if (b.dev) HX_HIP(hipFree(b.dev));
  b.dev = nd;
  b.dev_cap = cap;
  b.dirty = false;
}
// keys a stage hands out: internal ids -> global ids, in place
static void remap_out(hx_index* h, uint64_t* keys, int64_t n, hipStream_t st) {
  if (n <= 0 || ids_identity(h)) return;
  ids_upload(h);
  launch_remap_ids(keys, keys, n, h->ids.dev, h->ids.dev + h->ids.dev_cap, (int)h->ids.row0.size(),
                   (uint32_t)h->id_base, (uint32_t)h->n, 1, st);
}
// candidate keys a stage takes in: global ids -> internal ids (rows of other shards become empty slots)
static const uint64_t* remap_in(hx_index* h, const uint64_t* keys, int64_t n, hipStream_t st, int slot = WS_ID_IN) {
  if (n <= 0 || ids_identity(h)) return keys;
  ids_upload(h);
  uint64_t* tmp = (uint64_t*)h->ws.get(slot, (size_t)n * 8);
  launch_remap_ids(keys, tmp, n, h->ids.dev, h->ids.dev + h->ids.dev_cap, (int)h->ids.row0.size(),
                   (uint32_t)h->id_base, (uint32_t)h->n, 0, st);
  return tmp;
}
// a failed or finished add consumes the id named by hx_set_next_id
struct NextIdGuard {
  hx_index* h;
  ~NextIdGuard() { h->ids.next = -1; }
};

// ---------------------------------------------------------------------------------
// per-search candidate geometry
// ---------------------------------------------------------------------------------
struct Geometry {
  int Lp;     // candidates kept by the approximate pass
  int C;      // per-query buffer capacity (power of two <= CAND_CAP)
  int grow;   // chunk growth factor the plan aims at
  int grow_max;   // largest growth the buffer takes (overflow probability <= PREDICT_EPS); >= grow
  bool predictive;
};
// Threshold of a chunk.  Rows are scanned in geometrically growing chunks [n0, g*n0); a chunk
// appends every row whose score reaches tau and the buffer is then compacted to the best Lp.
//   classic    tau = Lp-th best so far: about (g-1)*Lp appended per query, always exact;
//   predictive tau = kq-th best so far, kq < Lp: the Lp-th best of g*n0 rows sits near the
//              (Lp/g)-th best of the first n0, so about kq*(g-1) are appended -- several times
//              fewer.  The kept list is still the exact top-Lp provided at least Lp rows reach
//              tau, i.e. kq + appended >= Lp; k_compact checks exactly that and flags the query
//              otherwise (it is then retried with the classic rule, as after an overflow).
// The count appended given tau is Poisson with mean (g-1)*G, G ~ Gamma(kq): a negative binomial
// NB(kq, 1/g).  kq is the smallest rank with P(appended < Lp - kq) <= 1e-7, and g the largest
// growth (<= 64) with P(appended > C - Lp) <= 1e-7 -- both from the exact distribution.
constexpr double PREDICT_EPS = 1e-7;
// P(X <= kmax), X ~ NB(r, p): failures before the r-th success
static double nb_cdf(int r, double p, int kmax) {
  if (kmax < 0) return 0.0;
  if (p >= 1.0) return 1.0;
  const double lq = std::log1p(-p);
  double lp = r * std::log(p), sum = 0.0;   // log pmf(0)
  for (int k = 0; k <= kmax; ++k) {
    sum += std::exp(lp);
    lp += std::log((double)(k + r) / (double)(k + 1)) + lq;
  }
  return sum;
}
static int predict_rank(int Lp, double g_eff) {
  if (g_eff <= 1.0) return Lp;
  static thread_local std::map<std::pair<int, int64_t>, int> cache;
  const auto key = std::make_pair(Lp, (int64_t)(g_eff * 1024.0));
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int lo = std::min(8, Lp), hi = Lp;   // smallest kq in [lo, Lp) whose underflow tail is small enough
  while (lo < hi) {
    const int kq = (lo + hi) / 2;
    if (nb_cdf(kq, 1.0 / g_eff, Lp - kq - 1) <= PREDICT_EPS) hi = kq;
    else lo = kq + 1;
  }
  return cache[key] = lo;
}
// cand8: the int8 candidate pass.  Its certificate radius (the row and query quantisation errors, ~8e-3 on unit
// vectors of 768 uniform components, against 1.25e-3 for fp16) asks for more candidates: the L'-th best int8
// score has to lie a radius below the exact L-th best.
static int cand8_lprime(int L) {
  // Measured on 10M x 768 (uniform components, radius 0.22 sigma of the score distribution), L = 100: L' = 320
  // leaves 3 % of the queries uncertified, 400 none of 12,288 (a margin of ~3.8 standard deviations of the gap
  // between the two order statistics), 450 is ~4.8: an uncertified query costs a whole fp16 scan of its own.
  static const int mul2 = getenv("HX_DEBUG_CAND8_MUL") ? std::max(2, 2 * atoi(getenv("HX_DEBUG_CAND8_MUL"))) : 9;   // halves
  static const int add = getenv("HX_DEBUG_CAND8_ADD") ? std::max(0, atoi(getenv("HX_DEBUG_CAND8_ADD"))) : 288;
  return std::min(std::max(mul2 * L / 2, L + add), std::max(L, CAND_CAP / 4));
}
// lp_force > 0 (with cand8): keep exactly that many candidates -- a shard of the candidates-first H1 exchange nominates
// its share of the global L', not L' of its own (hx_h1_nominate_async)
// few_queries (B <= 32): the scan is bandwidth-bound and what a launch costs is the compaction of ONE list per query
// behind it (latency, not volume), so the growth goes as high as the buffer takes: 10M rows in 3 launches instead of 4
static Geometry geometry(int L, bool approx, bool safe, bool cand8 = false, int lp_force = 0, bool few_queries = false) {
  static thread_local std::map<std::tuple<int, bool, bool, bool, int, bool>, Geometry> cache;
  const auto key = std::make_tuple(L, approx, safe, cand8, lp_force, few_queries);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  Geometry g;
  g.Lp = approx ? L + std::max(32, L / 2) : L;
  if (cand8) g.Lp = lp_force > 0 ? lp_force : cand8_lprime(L);
  if (safe) g.Lp = std::min(std::max(2 * g.Lp, g.Lp + 256), CAND_CAP / 4);
  int c = next_pow2(std::max(8 * g.Lp, 1024));
  if (cand8) {
    static const int cmin = getenv("HX_DEBUG_CAND8_C") ? atoi(getenv("HX_DEBUG_CAND8_C")) : 4096;
    static const int cmin_n = getenv("HX_DEBUG_NOM_C") ? atoi(getenv("HX_DEBUG_NOM_C")) : 1024;   // (8 x 1.25M rows: 1.585 ms per step at 1024, 1.605 at 2048, 1.81 at 4096)
    c = std::max(c, lp_force > 0 ? cmin_n : cmin);
  }
  g.C = safe ? CAND_CAP : std::min(c, CAND_CAP);
  HX_CHECK(g.Lp * 2 <= g.C && g.Lp >= L, "limit too large");
  g.predictive = false;
  g.grow = 2;
  g.grow_max = 2;
  if (!safe) {
    for (int gr = 64; gr > 2; --gr) {          // what the buffer can take
      const int kq = predict_rank(g.Lp, (double)gr);
      if (kq < g.Lp && 1.0 - nb_cdf(kq, 1.0 / gr, g.C - g.Lp) <= PREDICT_EPS) {
        g.grow_max = gr;
        break;
      }
    }
    // (int8 candidates, L' = 450: growth 63 / 24 / 16 / 12 / 8 -> 2 / 3 / 3 / 4 / 4 launches per 10M rows, 10.69 /
    // 10.53 / 10.48 / 10.47 / 10.48 ms per step: the appended volume per launch falls with the growth; again with the
    // cheaper log scatter of late round 3, dense search only: 50 / 24 / 16 / 12 -> 7.53 / 7.44 / 7.43 / 7.44 ms)
    static const int gmax8 = getenv("HX_DEBUG_GROW_MAX8") ? std::max(3, atoi(getenv("HX_DEBUG_GROW_MAX8"))) : 16;
    static const int gmax = getenv("HX_DEBUG_GROW_MAX") ? std::max(3, atoi(getenv("HX_DEBUG_GROW_MAX"))) : 64;
    static const bool few_off = getenv("HX_DEBUG_NO_FEW") != nullptr;
    for (int gr = std::min(64, (few_queries && !few_off) ? 64 : (cand8 ? gmax8 : gmax)); gr > 2; --gr) {
      const int kq = predict_rank(g.Lp, (double)gr);
      if (kq < g.Lp && 1.0 - nb_cdf(kq, 1.0 / gr, g.C - g.Lp) <= PREDICT_EPS) {
        g.grow = gr;
        g.predictive = true;
        break;
      }
    }
  }
  g.grow_max = std::max(g.grow_max, g.grow);
  if (getenv("HX_DEBUG_GEOMETRY"))
    fprintf(stderr, "[hx] geometry L=%d approx=%d safe=%d cand8=%d: Lp=%d C=%d grow=%d (max %d) kq=%d predictive=%d\n", L, (int)approx,
            (int)safe, (int)cand8, g.Lp, g.C, g.grow, g.grow_max, g.predictive ? predict_rank(g.Lp, (double)g.grow) : g.Lp, (int)g.predictive);
  return cache[key] = g;
}

// query-tile width of the scan kernel for a batch of B queries
// Query-tile width of the scan for a batch of B queries.  Measured on 10M x 768 int8 candidates, ms per pass / scan fraction
// of HBM peak (profiles/r04_mid_batch.txt): the 256 x 128 form of the staggered kernel 1.53-1.59 / 0.69-0.72 over B = 48..128
// (k_scan's 128 x 64 tile 1.71-1.77 / 0.60-0.62 at 48-64, its 128 x 128 tile 1.94-2.04 / 0.51-0.54 at 65-128, the 256-wide
// staggered kernel 2.03-2.13 / 0.59 at 96-128); two 128-query tiles for 129..256 lose to the 256-wide form (2.46-2.93 / 1.9-2.2).
static int scan_bn(const hx_index* h, int B) {
  return B <= h->bn32_max ? 32 : (B <= h->bn64_max ? 64 : (B <= h->bn128_max ? 128 : 256));
}

struct MatrixRef {
  const float* m32;
  const _Float16* m16;
  int d, dpad;
};
static MatrixRef pick_matrix(hx_index* h, int prefix) {
  if (prefix == 0)
    return MatrixRef{h->dense, h->dense_h, h->dim, h->dim_pad};
  for (int p = 0; p < h->n_pre; ++p)
    if (h->psize[p] == prefix) return MatrixRef{h->pre[p], p == 0 ? h->pre_h0 : nullptr, prefix, prefix};
  throw Error("unknown prefix size " + std::to_string(prefix));
}

static void zero_outputs(uint64_t* keys, int* cnt, int B, int L, hipStream_t st) {
  HX_HIP(hipMemsetAsync(keys, 0, (size_t)B * L * 8, st));
  HX_HIP(hipMemsetAsync(cnt, 0, (size_t)B * 4, st));
}

// scan all rows with geometric chunks; leaves the best `keep` keys (sorted) in cand
// Ends of the chunks of a scan over n_log rows (multiples of 256; the first chunk is the buffer's capacity: every
// row of it has its own slot).  Predictive geometries get BALANCED growth: the fewest launches the aimed-at growth
// allows -- one fewer when a growth within 1.25 x of it (and within what the buffer takes) does it -- and then the same
// ratio for every launch, instead of full-growth launches followed by a short last one (1.25M rows, int8 candidates:
// 4096 -> 65k -> 1.05M -> 1.25M became 4096 -> 72k -> 1.25M; a launch costs its log scatter and a compaction whatever
// its size).
static std::vector<int64_t> chunk_plan(int64_t n_log, const Geometry& g) {
  std::vector<int64_t> ends;
  int64_t r = std::min<int64_t>(n_log, g.C);
  ends.push_back(r);
  if (r >= n_log) return ends;
  double growth = (double)g.grow;
  if (g.predictive && !getenv("HX_DEBUG_NO_BALANCE")) {
    const double ratio = (double)n_log / (double)r;
    int k = std::max(1, (int)std::ceil(std::log(ratio) / std::log((double)g.grow) - 1e-9));
    if (k > 1) {
      const double g1 = std::pow(ratio, 1.0 / (k - 1));
      if (g1 <= 1.25 * g.grow && g1 <= (double)g.grow_max) --k;
    }
    growth = std::max(2.0, std::pow(ratio, 1.0 / k) * (1.0 + 1e-9));
  }
  while (r < n_log) {
    // rows per launch bounded so the kernel's 32-bit tile counters cannot wrap
    int64_t nx = std::min<int64_t>(round_up((int64_t)std::ceil((double)r * growth), 256), r + (1ll << 27));
    if (nx >= n_log || (double)n_log / (double)nx < 1.02) nx = n_log;     // (no sliver of a last launch)
    ends.push_back(nx);
    r = nx;
  }
  return ends;
}

// int8 scans: `rinv_x` = the per-row factor of the score (f32(dot) * rinv_x[row]) * rinv_q[query], `tm` its tile maxima;
// `prof_what` = the profile slot of the launches (hx_prof)
static void chunked_scan(hx_index* h, int kind, const uint8_t* A, const uint8_t* Q, int64_t row_bytes, int B,
                         int bn, const Geometry& g, uint64_t* cand, int* cnt, int* ovf, float* tau,
                         const float* rinv_q, hipStream_t st, const float* rinv_x = nullptr,
                         hx_index::TileMax* tm = nullptr, int prof_what = -1) {
  ScanArgs a{};
  a.A = A;
  a.Q = Q;
  a.row_bytes = row_bytes;
  a.B = B;
  a.nq_tiles = (int)(round_up(B, bn) / bn);
  a.tau = tau;
  a.cand = cand;
  a.cnt = cnt;
  a.overflow = ovf;
  a.cap = g.C;
  a.id_base = h->id_base;
  if (kind == KIND_I8 && !rinv_x) {
    rinv_x = h->q8_rinv;
    tm = &h->tm_q8;
  }
  a.rinv_x = rinv_x;
  a.rinv_q = rinv_q;
  if (kind == KIND_I8) {
    if (tm->rows != h->n) {   // rows were added since the last scan with these scales
      const int64_t tiles = (h->cap + 255) / 256;
      if (tiles > tm->cap) {
        if (tm->v) HX_HIP(hipFree(tm->v));
        tm->v = nullptr;
        HX_HIP(hipMalloc((void**)&tm->v, (size_t)tiles * 4));
        tm->cap = tiles;
      }
      launch_tile_max(rinv_x, h->n, tm->v, st);
      tm->rows = h->n;
    }
    a.rinv_tile_max = tm->v;
  }
  if (prof_what < 0) prof_what = kind;
  uint4* hitlog = nullptr;
  int* hitcnt = nullptr;
  int logcap = h->scan_logcap;
  // 65..128 queries: the staggered kernel in its 256-row x 128-query form (scan8.hip, HQ) -- the 128 x 128 tile of
  // k_scan streams at half of HBM peak there, the 256-wide form of scan8 spends half its MFMAs on padding columns
  const bool hq = bn == 128 && !h->no_hq;
  a.half_q = hq ? 1 : 0;
  static const int os_always = getenv("HX_DEBUG_SCAN_OVERSUB_ALWAYS") ? atoi(getenv("HX_DEBUG_SCAN_OVERSUB_ALWAYS")) : 0;   // diagnostics
  a.oversub = h->beside ? h->scan_oversub : os_always;
  if (bn == 256 || hq) {   // per-wave append logs of the staggered kernel (scan8.hip)
    // a wave logs about (appended per query) * B / SCAN8_WAVES entries per launch; a full log only
    // flags its queries for the retry
    // Planned per launch: the chunk [r0, r1) appends about rank * (r1 / r0 - 1) rows per query (rank = the rank whose
    // score is its threshold) and its items run on min(SCAN8_WAVES, 8 waves per 256 x 256 tile) waves -- a SMALL
    // collection has few tiles, so few waves share the same number of appends (15000 rows, B = 130: 43 tiles, and
    // a capacity planned for 2048 waves overflowed for half the queries).  Three times the mean, the largest launch.
    double want = 0.0;
    {
      const std::vector<int64_t> plan = chunk_plan((h->n + 255) / 256 * 256, g);
      for (size_t i = 1; i < plan.size(); ++i) {
        const int64_t p1 = plan[i - 1], nx = plan[i];
        const double growth = (double)nx / (double)p1;
        const double rank = g.predictive ? predict_rank(g.Lp, growth) : g.Lp;
        const double waves = std::min<double>(SCAN8_WAVES, 8.0 * (double)((nx - p1 + 255) / 256) * (double)(round_up(B, bn) / bn));
        want = std::max(want, 3.0 * rank * (growth - 1.0) * (double)B / waves + 64.0);
      }
    }
    logcap = std::min(h->scan_logcap, std::max(256, next_pow2((int)std::min(1e9, want) + 1)));
    hitlog = (uint4*)h->ws.get(WS_HITLOG, (size_t)SCAN8_WAVES * logcap * SCAN8_ENTRY * sizeof(uint4));
    hitcnt = (int*)h->ws.get(WS_HITCNT, (size_t)SCAN8_WAVES * 4);
  }
  a.hitcnt = hitcnt;
  a.logcap = logcap;
  int* kept = (int*)h->ws.get(WS_KEPT, (size_t)B * 4);
  // Scan order (kernels.hpp): logical 256-row tile t is physical tile (t * mul) mod tiles, mul ~ 0.618 * tiles and
  // coprime to it, so every chunk samples the whole matrix evenly -- a topically clustered corpus stays
  // exchangeable for the predictive thresholds.  The logical range is [0, tiles * 256): rows past n count nothing.
  const int64_t n_tiles = (h->n + 255) / 256;
  const int64_t n_log = n_tiles * 256;
  a.n_total = h->n;
  a.perm_n = 0;
  a.perm_mul = 1;
  if (n_tiles >= 8 && !getenv("HX_DEBUG_NO_PERM")) {
    uint64_t m = (uint64_t)((double)n_tiles * 0.6180339887498949) | 1ull;
    while (std::gcd<uint64_t, uint64_t>(m, (uint64_t)n_tiles) != 1) m += 2;
    a.perm_n = (uint32_t)n_tiles;
    a.perm_mul = (uint32_t)(m % (uint64_t)n_tiles);
    a.perm_inv = 1.0 / (double)n_tiles;
  }
  const std::vector<int64_t> plan = chunk_plan(n_log, g);
  size_t pi = 0;
  int64_t r0 = 0, r1 = plan[0];
  // threshold -inf, flags 0; the first chunk passes every row into slot (row - r0): its count is known
  launch_scan_init(tau, cnt, ovf, kept, B, (int)(r1 - r0), st);
  int chk_rank = 0;   // rank whose score is the threshold of the chunk being scanned (0: none)
  while (r0 < n_log) {
    a.row_begin = r0;
    a.row_end = r1;
    a.hitlog = r0 > 0 ? hitlog : nullptr;   // the first chunk passes every row: k_scan, one slot per row
    a.all_pass = (r0 == 0 && r1 - r0 <= g.C) ? 1 : 0;
    {
      const double rows = (double)(r1 - r0);
      const double elems = kind == KIND_F16 ? (double)row_bytes / 2.0 : (double)row_bytes;
      // the profile counts the k_scan8 launches only (bn == 256): the first chunk is a few thousand rows
      // through k_scan and would only blur the per-launch average the roofline is quoted on
      ProfScope ps(h, st, prof_what, 2.0 * B * rows * elems, rows * (double)row_bytes + (double)B * row_bytes,
                   !(a.all_pass && bn == 256));
      if (a.all_pass && bn == 256) {
        // a few thousand rows: 128 x 128 tiles give four times the workgroups of the 256 x 256 form
        ScanArgs f = a;
        f.nq_tiles = (int)(round_up(B, 256) / 128);
        launch_scan(f, kind, 128, st);
      } else {
        launch_scan(a, kind, bn, st, ps.end_event());
      }
    }
    ++pi;
    const int64_t next = pi < plan.size() ? plan[pi] : n_log;
    const int next_rank = (g.predictive && next > r1) ? predict_rank(g.Lp, (double)next / (double)r1) : g.Lp;
    launch_compact(cand, g.C, cnt, B, g.Lp, 0, cand, g.C, cnt, tau, g.C, st, next_rank, chk_rank, kept, ovf);
    chk_rank = next_rank < g.Lp ? next_rank : 0;
    r0 = r1;
    r1 = next;
  }
}

// exact fallback for selected queries: stream all rows through the spec arithmetic
static void exact_range_fallback(hx_index* h, int kind, const void* M, int64_t row_stride, int dim_pad,
                                 const void* Qp, int64_t q_stride, const float* rinv_q,
                                 const std::vector<int>& sel, int L, uint64_t* out_keys, int* out_cnt,
                                 hipStream_t st) {
  const int nsel = (int)sel.size();
  if (!nsel) return;
  int* qsel = (int*)h->ws.get(WS_QSEL, (size_t)nsel * 4);
  HX_HIP(hipMemcpyAsync(qsel, sel.data(), (size_t)nsel * 4, hipMemcpyHostToDevice, st));
  const int C = CAND_CAP;
  uint64_t* buf = (uint64_t*)h->ws.get(WS_FB_KEYS, (size_t)nsel * C * 8);
  int* cnt = (int*)h->ws.get(WS_FB_CNT, (size_t)nsel * 4);
  int* incnt = (int*)h->ws.get(WS_FB_INCNT, (size_t)nsel * 4);
  HX_HIP(hipMemsetAsync(buf, 0, (size_t)nsel * C * 8, st));
  RangeArgs ra{};
  ra.r.kind = kind;
  ra.r.M = M;
  ra.r.row_stride = row_stride;
  ra.r.dim_pad = dim_pad;
  ra.r.Q = Qp;
  ra.r.q_stride = q_stride;
  ra.r.rinv_x = h->q8_rinv;
  ra.r.rinv_q = rinv_q;
  ra.r.n_rows = h->n;
  ra.r.id_base = h->id_base;
  ra.r.out = buf;
  ra.r.stride = C;
  ra.qsel = qsel;
  ra.nsel = nsel;
  ra.slot0 = L;
  const int64_t CH = C - L;
  for (int64_t r0 = 0; r0 < h->n; r0 += CH) {
    const int64_t r1 = std::min<int64_t>(h->n, r0 + CH);
    ra.row_begin = r0;
    ra.row_end = r1;
    launch_rescore_range(ra, st);
    launch_fill_i32(incnt, nsel, (int)(L + (r1 - r0)), st);
    launch_compact(buf, C, incnt, nsel, L, 0, buf, C, cnt, nullptr, C, st);
  }
  // scatter rows back: out_keys[sel[f]] = buf[f][0..L)
  for (int f = 0; f < nsel; ++f) {
    HX_HIP(hipMemcpyAsync(out_keys + (int64_t)sel[f] * L, buf + (int64_t)f * C, (size_t)L * 8,
                          hipMemcpyDeviceToDevice, st));
    HX_HIP(hipMemcpyAsync(out_cnt + sel[f], cnt + f, 4, hipMemcpyDeviceToDevice, st));
  }
}

static int* host_pin(hx_index* h) {
  if (!h->pin) HX_HIP(hipHostMalloc((void**)&h->pin, 64, hipHostMallocDefault));
  return h->pin;
}

static std::vector<int> read_failures(hx_index* h, int* fail, int* nfail, int B, hipStream_t st) {
  int* pin = host_pin(h);
  HX_HIP(hipMemcpyAsync(pin, nfail, 4, hipMemcpyDeviceToHost, st));
  // a sparse stage enqueued before this point (search_dense's `between`): its summary rides along
  const bool take_sparse = h->sp_sum_pending && !h->sp_sum_fetched;
  if (take_sparse) HX_HIP(hipMemcpyAsync(pin + 1, h->ws.get(WS_SP_SUM, 8), 8, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  const int nf = pin[0];
  if (take_sparse) {
    h->sp_sum[0] = pin[1];
    h->sp_sum[1] = pin[2];
    h->sp_sum_fetched = true;
  }
  std::vector<int> sel;
  if (nf > 0) {
    std::vector<int> f((size_t)B);
    HX_HIP(hipMemcpy(f.data(), fail, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      if (f[b]) sel.push_back(b);
  }
  (void)h;
  return sel;
}

// ---------------------------------------------------------------------------------
// stages
// ---------------------------------------------------------------------------------
// Re-run the failing queries of a batch as a smaller batch at `level + 1` and put
// their rows back.  level 0 = normal geometry, level 1 = safe geometry, then exact.
template <typename F>
static void retry_subset(hx_index* h, const float* q_dev, const std::vector<int>& sel, int L,
                         uint64_t* out_keys, int* out_cnt, hipStream_t st, int level, F&& run) {
  const int ns = (int)sel.size();
  h->retries += ns;
  const int off = 1000 * (level + 1);
  float* qs = (float*)h->ws.get(WS_H_QD + off, (size_t)ns * h->dim * 4);
  uint64_t* ks = (uint64_t*)h->ws.get(WS_H_OUT + off, (size_t)ns * L * 8);
  int* cs = (int*)h->ws.get(WS_H_OCNT + off, (size_t)ns * 4);
  for (int f = 0; f < ns; ++f)
    HX_HIP(hipMemcpyAsync(qs + (int64_t)f * h->dim, q_dev + (int64_t)sel[f] * h->dim, (size_t)h->dim * 4,
                          hipMemcpyDeviceToDevice, st));
  run(qs, ns, ks, cs);
  for (int f = 0; f < ns; ++f) {
    HX_HIP(hipMemcpyAsync(out_keys + (int64_t)sel[f] * L, ks + (int64_t)f * L, (size_t)L * 8,
                          hipMemcpyDeviceToDevice, st));
    HX_HIP(hipMemcpyAsync(out_cnt + sel[f], cs + f, 4, hipMemcpyDeviceToDevice, st));
  }
}

// `between` (optional): independent work of the caller, enqueued after this stage's kernels and BEFORE
// the host reads the failure flags -- so the device has work while the host waits, and the stream is
// still full when it returns.  It may consume out_keys speculatively: the return value says whether a
// retry or the exact path rewrote rows of out_keys after `between` ran (then the caller redoes it).
// `defer`: enqueue only -- no flag read, no retry; the stage's failure count stays in WS_NFAIL (level 0) for
// the caller (hx_h1_local_async), who redoes the batch through the synchronous path if it is not zero.
// `flag_acc` (with `defer`): the device word the stage ADDS its failure count to instead of WS_NFAIL -- several
// deferred stages of one query tree share it and the caller reads it once (hx_*_async, the tree of hybrid_query_dev).
static bool search_dense(hx_index* h, const float* q_dev, int B, int prefix, int L, uint64_t* out_keys,
                         int* out_cnt, hipStream_t st, int level = 0,
                         const std::function<void()>& between = std::function<void()>(), bool defer = false,
                         int* flag_acc = nullptr, const std::function<void()>& after_scan = std::function<void()>()) {
  // `after_scan` (optional): called once, right behind the candidate scan's launches and before the stage's small tail
  // kernels (exact re-score, top-L, certificate) -- the caller forks work onto another stream there
  HX_CHECK(B > 0, "B must be positive");
  HX_CHECK(L >= 1 && L <= MAX_LIMIT, "limit out of range [1, 2048]");
  if (h->n == 0) {
    zero_outputs(out_keys, out_cnt, B, L, st);
    if (after_scan) after_scan();
    if (between) between();
    return false;
  }
  const int wo = 1000 * level;  // workspace slots of this level
  const MatrixRef m = pick_matrix(h, prefix);
  const int bn = scan_bn(h, B);
  const int Bpad = (int)round_up(B, bn);
  float* qn = (float*)h->ws.get(WS_QN + wo, (size_t)B * m.dpad * 4);
  _Float16* qh = (_Float16*)h->ws.get(WS_QH + wo, (size_t)Bpad * m.dpad * 2);
  launch_prep_queries_f(q_dev, h->dim, B, Bpad, m.d, m.dpad, qn, qh, st);
  int* fail = (int*)h->ws.get(WS_FAIL + wo, (size_t)B * 4);
  int* nfail = (int*)h->ws.get(WS_NFAIL + wo, 4);
  std::vector<int> sel;
  // The candidate pass of the full-vector stage runs on the int8 matrix pipe (half the bytes, twice the rate of
  // fp16) when the index holds the scaled int8 copy; a query it cannot certify is retried through the fp16 scan
  // (level 1) like any other flagged query.  Final scores are spec_dot on the fp32 rows either way.
  const bool use8 = h->cand8 && !h->cand8_off && h->q8s && prefix == 0 && level == 0;
  if (m.m16 || use8) {
    const Geometry g = geometry(L, true, level > 0, use8, 0, B <= 32);
    uint64_t* cand = (uint64_t*)h->ws.get(WS_CAND + wo, (size_t)B * g.C * 8);
    uint64_t* cand2 = (uint64_t*)h->ws.get(WS_CAND2 + wo, (size_t)B * g.C * 8);
    int* cnt = (int*)h->ws.get(WS_CNT + wo, (size_t)B * 4);
    int* ovf = (int*)h->ws.get(WS_OVF + wo, (size_t)B * 4);
    float* tau = (float*)h->ws.get(WS_TAU + wo, (size_t)B * 4);
    const float* eps_q = nullptr;
    if (use8) {
      int8_t* q8 = (int8_t*)h->ws.get(WS_Q8S + wo, (size_t)Bpad * h->dim_pad8);
      float* sq = (float*)h->ws.get(WS_SQ + wo, (size_t)Bpad * 4);
      float* eq = (float*)h->ws.get(WS_EPSQ + wo, (size_t)B * 4);
      launch_prep_queries_s8(qn, m.dpad, B, Bpad, h->dim_pad8, q8, sq, eq, h->s8_err, st);
      chunked_scan(h, KIND_I8, (const uint8_t*)h->q8s, (const uint8_t*)q8, h->dim_pad8, B, bn, g, cand, cnt, ovf, tau,
                   sq, st, h->q8s_scale, &h->tm_q8s, 3);
      eps_q = eq;
      h->cand8_queries += B;
    } else {
      chunked_scan(h, KIND_F16, (const uint8_t*)m.m16, (const uint8_t*)qh, (int64_t)m.dpad * 2, B, bn, g,
                   cand, cnt, ovf, tau, nullptr, st);
    }
    if (after_scan) after_scan();
    RescoreArgs r{};
    r.kind = KIND_F32;
    r.M = m.m32;
    r.row_stride = m.dpad;
    r.dim_pad = m.dpad;
    r.Q = qn;
    r.q_stride = m.dpad;
    r.n_rows = h->n;
    r.id_base = h->id_base;
    r.cand = cand;
    r.cnt = cnt;
    r.stride = g.C;
    r.max_cnt = g.Lp;       // chunked_scan's last compaction keeps at most L' keys
    r.B = B;
    r.out = cand2;
    if (defer && flag_acc) nfail = flag_acc;
    else HX_HIP(hipMemsetAsync(nfail, 0, 4, st));
    // re-score + top-L + certificate in one launch (select.hip: k_dense_finish) when the lists fit a wave's registers
    unsigned int* done = (unsigned int*)h->ws.get(WS_DONE + wo, (size_t)B * 4);
    if (h->done_zeroed[level > 0] < B) {              // the kernel leaves its counters at zero: cleared once per growth
      HX_HIP(hipMemsetAsync(done, 0, (size_t)B * 4, st));
      h->done_zeroed[level > 0] = B;
    }
    static const bool no_fuse = getenv("HX_DEBUG_NO_FINISH_FUSE") != nullptr;
    // (a small batch only: at B = 1024 the three kernels take 0.74 ms of a dense search against 0.83 through the fused one
    // -- a wave per candidate over 113 blocks per query beats the last block's fold, and three launches are nothing there)
    if (no_fuse || B > 64 || !launch_dense_finish(r, g.Lp, L, out_keys, out_cnt, ovf, HX_EPS_F16, eps_q, fail, nfail, done, st)) {
      launch_rescore_list(r, st);
      launch_compact(cand2, g.C, cnt, B, L, 0, out_keys, L, out_cnt, nullptr, g.Lp, st);
      launch_certify(cand, g.C, cnt, g.Lp, out_keys, L, out_cnt, L, ovf, HX_EPS_F16, B, fail, nfail, st, eps_q);
    }
    if (between) between();
    if (defer) return false;
    sel = read_failures(h, fail, nfail, B, st);
    if (use8) {
      h->cand8_failed += (int64_t)sel.size();
      // the guard: a collection whose rows the int8 grid resolves badly (one bad row widens the radius for every query)
      // would pay an fp16 scan on top of the int8 one for query after query
      h->c8_win_q += B;
      h->c8_win_f += (int64_t)sel.size();
      if (h->c8_win_q >= 4096) {
        if (h->c8_win_f * 20 > h->c8_win_q) {
          h->cand8_off = true;
          h->cand8_auto_off = true;
        }
        h->c8_win_q = h->c8_win_f = 0;
      }
    }
    if (!sel.empty() && level == 0) {
      retry_subset(h, q_dev, sel, L, out_keys, out_cnt, st, level,
                   [&](const float* qs, int ns, uint64_t* ks, int* cs) {
                     search_dense(h, qs, ns, prefix, L, ks, cs, st, 1);
                   });
      return true;
    }
  } else {
    sel.resize((size_t)B);
    std::iota(sel.begin(), sel.end(), 0);
    if (after_scan) after_scan();
    if (between) between();
    if (defer) {                         // no fp16 copy to scan: every query needs the exact path
      launch_fill_i32(flag_acc ? flag_acc : nfail, 1, B, st);
      return false;
    }
  }
  if (!sel.empty()) {
    h->dense_fallbacks += (int64_t)sel.size();
    exact_range_fallback(h, KIND_F32, m.m32, m.dpad, m.dpad, qn, m.dpad, nullptr, sel, L, out_keys,
                         out_cnt, st);
    return true;
  }
  return false;
}

static void search_i8(hx_index* h, const float* q_dev, int B, int L, uint64_t* out_keys, int* out_cnt,
                      hipStream_t st, int level = 0, int* flag_acc = nullptr) {   // flag_acc: deferred, as search_dense
  HX_CHECK(B > 0, "B must be positive");
  HX_CHECK(L >= 1 && L <= MAX_LIMIT, "limit out of range [1, 2048]");
  if (h->n == 0) return zero_outputs(out_keys, out_cnt, B, L, st);
  const int wo = 1000 * level;
  const int bn = scan_bn(h, B);
  const int Bpad = (int)round_up(B, bn);
  int8_t* q8 = (int8_t*)h->ws.get(WS_Q8 + wo, (size_t)Bpad * h->dim_pad8);
  float* rq = (float*)h->ws.get(WS_RINVQ + wo, (size_t)Bpad * 4);
  launch_prep_queries_i8(q_dev, h->dim, B, Bpad, h->dim_pad8, q8, rq, st);
  const Geometry g = geometry(L, false, level > 0, false, 0, B <= 32);
  uint64_t* cand = (uint64_t*)h->ws.get(WS_CAND + wo, (size_t)B * g.C * 8);
  int* cnt = (int*)h->ws.get(WS_CNT + wo, (size_t)B * 4);
  int* ovf = (int*)h->ws.get(WS_OVF + wo, (size_t)B * 4);
  float* tau = (float*)h->ws.get(WS_TAU + wo, (size_t)B * 4);
  int* fail = (int*)h->ws.get(WS_FAIL + wo, (size_t)B * 4);
  int* nfail = (int*)h->ws.get(WS_NFAIL + wo, 4);
  chunked_scan(h, KIND_I8, (const uint8_t*)h->q8, (const uint8_t*)q8, h->dim_pad8, B, bn, g, cand, cnt, ovf,
               tau, rq, st);
  // the scan's scores are already exact: the list is final unless a buffer overflowed
  launch_compact(cand, g.C, cnt, B, L, 0, out_keys, L, out_cnt, nullptr, g.Lp, st);
  if (flag_acc) nfail = flag_acc;
  else HX_HIP(hipMemsetAsync(nfail, 0, 4, st));
  launch_certify(cand, g.C, cnt, std::numeric_limits<int>::max(), out_keys, L, out_cnt, L, ovf, 0.f, B,
                 fail, nfail, st);
  if (flag_acc) return;
  std::vector<int> sel = read_failures(h, fail, nfail, B, st);
  if (sel.empty()) return;
  if (level == 0) {
    retry_subset(h, q_dev, sel, L, out_keys, out_cnt, st, level,
                 [&](const float* qs, int ns, uint64_t* ks, int* cs) { search_i8(h, qs, ns, L, ks, cs, st, 1); });
    return;
  }
  h->i8_fallbacks += (int64_t)sel.size();
  exact_range_fallback(h, KIND_I8, h->q8, h->dim_pad8, h->dim_pad, q8, h->dim_pad8, rq, sel, L, out_keys,
                       out_cnt, st);
}

// keys a (query, part) may hand to the exact pass: the top-L plus the documents within the margin of the L-th
// (2048 at least: with discrete BM25 weights hundreds of documents can tie at the L-th score of a 10M-row shard)
static int sparse_lout(int L) { return std::max(2048, next_pow2(L + L / 2 + 64)); }

static SparseIndexView view_of(const hx_index* h, const hx_index::SparseIx& x) {
  SparseIndexView v{};
  v.post = x.sp.post;
  v.ptr = x.sp.ptr;
  v.uterms = x.sp.uterms;
  v.n_live = (int)x.sp.n_live;
  v.n_docs = x.n_docs;
  v.n_segments = x.n_segments;
  v.seg_docs = x.seg_docs;
  v.id_base = h->id_base + x.doc0;
  return v;
}
static SparseCsr csr_of(const hx_index* h) {
  SparseCsr d{};
  d.indptr = h->sp_indptr;
  d.idx = h->sp_idx;
  d.val = h->sp_val;
  d.n_docs = h->sp_rows;
  d.id_base = h->id_base;
  return d;
}

// K7: select (integer pass over the inverted index, base and tail) -> exact scores of the kept candidates
// -> top-L.  Everything is enqueued; sparse_resolve() then reads the per-query flags and serves the flagged
// queries document-at-a-time.
// the select pass of a batch: per query one list of integer-score keys, best first (the top-L plus the documents within
// the margin of the L-th), stride `lout`
struct SparseLists {
  uint64_t* list = nullptr;
  int* lcnt = nullptr;
  int lout = 0;
  int* margin = nullptr;
  int* flag = nullptr;     // per query: k_sparse_prep's verdict (0 ok)
  int* fail = nullptr;     // per query: a buffer of the select pass overflowed
};
static SparseLists sparse_select_lists(hx_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                                       int B, int L, hipStream_t st) {
  HX_CHECK(B > 0, "B must be positive");
  HX_CHECK(L >= 1 && L <= MAX_LIMIT, "limit out of range [1, 2048]");
  finalize(h, st);
  Workspace& w = h->ws;
  SparseLists out;
  int* flag = (int*)w.get(WS_SP_FLAG, (size_t)B * 4);
  int* fail = (int*)w.get(WS_SP_FAIL, (size_t)B * 4);
  out.flag = flag;
  out.fail = fail;
  HX_HIP(hipMemsetAsync(flag, 0, (size_t)B * 4, st));
  HX_HIP(hipMemsetAsync(fail, 0, (size_t)B * 4, st));
  const hx_index::SparseIx* ixs[2] = {&h->sp_base, &h->sp_tail};
  if (h->sp_base.n_segments == 0 && h->sp_tail.n_segments == 0) return out;     // no posting: list stays NULL
  const int lout = sparse_lout(L);
  // ---- per-query preparation
  SparsePrepArgs pa{};
  pa.q_indptr = q_indptr;
  pa.q_idx = q_idx;
  pa.q_val = q_val;
  pa.B = B;
  for (int v = 0; v < 2; ++v) {
    pa.ix[v] = view_of(h, *ixs[v]);
    pa.q_ti[v] = (int32_t*)w.get(v == 0 ? WS_SP_TI0 : WS_SP_TI1, (size_t)B * SP_TMAX * 4);
  }
  pa.wmax = std::max(h->sp_wmax, h->sp_wmax_shared);   // (shared: the largest weight of ANY shard, hx_set_sparse_wmax)
  pa.index_nonpos = (h->sp_have_w && h->sp_wmin > 0.0f) ? 0 : 1;
  pa.q_qs = (float*)w.get(WS_SP_QS, (size_t)B * SP_TMAX * 4);
  pa.q_margin = (int*)w.get(WS_SP_MARGIN, (size_t)B * 4);
  pa.q_flag = flag;
  pa.q_work = (unsigned long long*)w.get(WS_SP_WORK, (size_t)B * 8);
  pa.stat_postings = nullptr;
  if (h->prof) {
    if (!h->sp_counter) {
      HX_HIP(hipMalloc((void**)&h->sp_counter, 8));
      HX_HIP(hipMemset(h->sp_counter, 0, 8));
    }
    pa.stat_postings = h->sp_counter;
  }
  launch_sparse_prep(pa, st);
  // ---- select: parts of the base, then parts of the tail, side by side in one list buffer.  A query is cut
  // into as many workgroups as its share of the batch's postings asks for (k_sparse_plan), up to parts[0].
  int parts[2] = {0, 0};
  const int pt_cap = std::max(1, CAND_CAP / lout);
  if (h->sp_tail.n_segments) parts[1] = 1;
  int slots = h->sp_base.seg_docs == SEG_DOCS_LARGE ? 256 : 512;   // workgroups resident at once
  if (h->sp_base.n_segments) {
    int p = std::min(pt_cap - parts[1], h->sp_base.n_segments);
    if (const char* e = getenv("HX_DEBUG_SP_PTMAX")) p = std::min(p, std::max(1, atoi(e)));   // diagnostics
    parts[0] = std::max(p, 1);
  }
  if (const char* e = getenv("HX_DEBUG_SP_SLOTS")) slots = std::max(1, atoi(e));                // diagnostics
  int* qparts = nullptr;
  int* items = nullptr;
  int* n_items = nullptr;
  // cutting queries into parts pays only while the batch alone cannot fill the chip twice over
  if (parts[0]) parts[0] = std::max(1, std::min(parts[0], (2 * slots + B - 1) / B));
  if (B <= 4096 && parts[0]) {      // (the plan ranks by counting: quadratic in the batch)
    qparts = (int*)w.get(WS_SP_QPARTS, (size_t)B * 4);
    items = (int*)w.get(WS_SP_ORDER, ((size_t)B * parts[0] + 1) * 4);
    n_items = items + (size_t)B * parts[0];
    launch_sparse_plan(pa.q_work, B, parts[0], slots, qparts, items, n_items, st);
  } else if (parts[0]) {
    parts[0] = 1;
  }
  const int pt = parts[0] + parts[1];
  HX_CHECK(pt >= 1 && (int64_t)pt * lout <= CAND_CAP, "sparse: limit too large");
  uint64_t* pk = (uint64_t*)w.get(WS_SP_PARTS, (size_t)B * pt * lout * 8);
  int* pc = (int*)w.get(WS_SP_PCNT, (size_t)B * pt * 4);
  HX_HIP(hipMemsetAsync(pc, 0, (size_t)B * pt * 4, st));   // parts a query does not use stay empty
  {
    ProfScope ps(h, st, 2, 0.0, 0.0);
    for (int v = 0; v < 2; ++v) {
      if (!parts[v]) continue;
      SparseSelectArgs a{};
      a.ix = pa.ix[v];
      a.q_indptr = q_indptr;
      a.q_ti = pa.q_ti[v];
      a.q_qs = pa.q_qs;
      a.q_margin = pa.q_margin;
      a.q_flag = flag;
      a.B = B;
      a.parts = parts[v];
      a.q_parts = v == 0 ? qparts : nullptr;
      a.items = v == 0 ? items : nullptr;
      a.n_items = v == 0 ? n_items : nullptr;
      a.limit = L;
      a.lout = lout;
      a.out = pk;
      a.out_cnt = pc;
      a.parts_total = pt;
      a.part0 = v == 0 ? 0 : parts[0];
      a.cand = (uint64_t*)w.get(v == 0 ? WS_SP_CAND : WS_SP_PARK, (size_t)B * parts[v] * (ixs[v]->seg_docs + ixs[v]->seg_docs / 8) * 8);
      a.q_fail = fail;
      a.cut_step = h->sp_cut_step;
      launch_sparse_select(a, st);
    }
  }
  // ---- one list per query (best integer scores first), its exact scores, the top-L
  uint64_t* list = pk;
  int* lcnt = pc;
  if (pt > 1) {
    uint64_t* packed = (uint64_t*)w.get(WS_SP_LIST, (size_t)B * pt * lout * 8);
    lcnt = (int*)w.get(WS_SP_LCNT, (size_t)B * 4);
    launch_sparse_pack(pk, pc, B, pt, lout, packed, lcnt, st);
    list = pk;     // the sorted union goes back to the (now free) parts buffer, stride lout
    launch_compact(packed, pt * lout, lcnt, B, lout, 0, list, lout, lcnt, nullptr, pt * lout, st);
  }
  out.list = list;
  out.lcnt = lcnt;
  out.lout = lout;
  out.margin = pa.q_margin;
  return out;
}

static void sparse_enqueue(hx_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                           int B, int L, uint64_t* out_keys, int* out_cnt, hipStream_t st) {
  const SparseLists sl = sparse_select_lists(h, q_indptr, q_idx, q_val, B, L, st);
  if (!sl.list) return zero_outputs(out_keys, out_cnt, B, L, st);
  Workspace& w = h->ws;
  const int lout = sl.lout;
  int* flag = sl.flag;
  int* fail = sl.fail;
  SparseRescoreArgs ra{};
  ra.d = csr_of(h);
  ra.q_indptr = q_indptr;
  ra.q_idx = q_idx;
  ra.q_val = q_val;
  ra.cand = sl.list;
  ra.cnt = sl.lcnt;
  ra.stride = lout;
  ra.B = B;
  ra.limit = L;
  ra.q_margin = sl.margin;
  ra.q_flag = flag;
  ra.out = (uint64_t*)w.get(WS_SP_EXACT, (size_t)B * lout * 8);
  ra.out_cnt = (int*)w.get(WS_SP_ECNT, (size_t)B * 4);
  ra.q_fail = fail;
  launch_sparse_rescore(ra, st);
  launch_compact(ra.out, lout, ra.out_cnt, B, std::min(L, lout), 0, out_keys, L, out_cnt, nullptr, lout, st);
  launch_sparse_summary(flag, fail, B, (int*)w.get(WS_SP_SUM, 8), st);
  h->sp_sum_pending = true;
  h->sp_sum_fetched = false;
}

// Document-at-a-time path for the listed queries: every row through the exact arithmetic.  Rows are taken in
// geometrically growing chunks; a chunk appends the keys that reach the query's threshold (the L-th best so far)
// and the buffer is compacted to the best L after it -- the classic rule of chunked_scan.  A query whose buffer
// overflows (more than 8192 - L documents tie at or beat its threshold inside one chunk) is redone with a slot
// per row, 8192 - L rows at a time.
static void sparse_exact_fallback(hx_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                                  const std::vector<int>& sel_in, int L, uint64_t* out_keys, int* out_cnt,
                                  hipStream_t st) {
  if (sel_in.empty()) return;
  const int C = CAND_CAP;
  SparseRangeArgs ra{};
  ra.d = csr_of(h);
  ra.q_indptr = q_indptr;
  ra.q_idx = q_idx;
  ra.q_val = q_val;
  ra.stride = C;
  std::vector<int> sel = sel_in;
  for (int pass = 0; pass < 2 && !sel.empty(); ++pass) {
    const int nsel = (int)sel.size();
    int* qsel = (int*)h->ws.get(WS_QSEL, (size_t)nsel * 4);
    HX_HIP(hipMemcpyAsync(qsel, sel.data(), (size_t)nsel * 4, hipMemcpyHostToDevice, st));
    uint64_t* buf = (uint64_t*)h->ws.get(WS_FB_KEYS, (size_t)nsel * C * 8);
    int* cnt = (int*)h->ws.get(WS_FB_CNT, (size_t)nsel * 4);
    int* incnt = (int*)h->ws.get(WS_FB_INCNT, (size_t)nsel * 4);
    float* tau = (float*)h->ws.get(WS_SP_FTAU, (size_t)nsel * 4);
    int* ovf = (int*)h->ws.get(WS_SP_FOVF, (size_t)nsel * 4);
    HX_HIP(hipMemsetAsync(buf, 0, (size_t)nsel * C * 8, st));
    HX_HIP(hipMemsetAsync(cnt, 0, (size_t)nsel * 4, st));
    HX_HIP(hipMemsetAsync(ovf, 0, (size_t)nsel * 4, st));
    ra.qsel = qsel;
    ra.nsel = nsel;
    ra.out = buf;
    if (pass == 0) {
      launch_fill_f32(tau, nsel, -std::numeric_limits<float>::infinity(), st);
      ra.tau = tau;
      ra.cnt = cnt;
      ra.ovf = ovf;
      int64_t r0 = 0, r1 = std::min<int64_t>(h->sp_rows, C - L);
      while (r0 < h->sp_rows) {
        ra.row_begin = r0;
        ra.row_end = r1;
        launch_sparse_range(ra, st);
        launch_compact(buf, C, cnt, nsel, L, 0, buf, C, cnt, tau, C, st);
        r0 = r1;
        r1 = std::min<int64_t>(h->sp_rows, r1 * 2);
      }
    } else {
      ra.tau = nullptr;
      ra.slot0 = L;
      const int64_t CH = C - L;
      for (int64_t r0 = 0; r0 < h->sp_rows; r0 += CH) {
        const int64_t r1 = std::min<int64_t>(h->sp_rows, r0 + CH);
        ra.row_begin = r0;
        ra.row_end = r1;
        launch_sparse_range(ra, st);
        launch_fill_i32(incnt, nsel, (int)(L + (r1 - r0)), st);
        launch_compact(buf, C, incnt, nsel, L, 0, buf, C, cnt, nullptr, C, st);
      }
    }
    std::vector<int> hovf((size_t)nsel, 0);
    if (pass == 0) {
      HX_HIP(hipMemcpyAsync(hovf.data(), ovf, (size_t)nsel * 4, hipMemcpyDeviceToHost, st));
      HX_HIP(hipStreamSynchronize(st));
    }
    std::vector<int> again;
    for (int f = 0; f < nsel; ++f) {
      if (hovf[(size_t)f]) {
        again.push_back(sel[(size_t)f]);
        continue;
      }
      HX_HIP(hipMemcpyAsync(out_keys + (int64_t)sel[f] * L, buf + (int64_t)f * C, (size_t)L * 8,
                            hipMemcpyDeviceToDevice, st));
      HX_HIP(hipMemcpyAsync(out_cnt + sel[f], cnt + f, 4, hipMemcpyDeviceToDevice, st));
    }
    sel.swap(again);
  }
}

// Reads the flags of the last sparse_enqueue on this index (one host round trip) and serves the flagged
// queries exactly.  Returns whether rows of out_keys were rewritten.
static bool sparse_resolve(hx_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val, int B,
                           int L, uint64_t* out_keys, int* out_cnt, hipStream_t st) {
  if (h->sp_base.n_segments == 0 && h->sp_tail.n_segments == 0) return false;
  if (!h->sp_sum_fetched) {             // no dense stage read it along: one round trip of its own
    int* pin = host_pin(h);
    HX_HIP(hipMemcpyAsync(pin + 1, h->ws.get(WS_SP_SUM, 8), 8, hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    h->sp_sum[0] = pin[1];
    h->sp_sum[1] = pin[2];
  }
  h->sp_sum_pending = false;
  h->sp_sum_fetched = false;
  if (h->sp_sum[0] == 0 && h->sp_sum[1] == 0) return false;      // the usual case: nothing flagged
  std::vector<int> flag((size_t)B), fail((size_t)B);
  HX_HIP(hipMemcpyAsync(flag.data(), h->ws.get(WS_SP_FLAG, (size_t)B * 4), (size_t)B * 4, hipMemcpyDeviceToHost, st));
  HX_HIP(hipMemcpyAsync(fail.data(), h->ws.get(WS_SP_FAIL, (size_t)B * 4), (size_t)B * 4, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  std::vector<int> sel;
  for (int b = 0; b < B; ++b) {
    HX_CHECK(flag[(size_t)b] != 2, "sparse query values must be finite");
    HX_CHECK(flag[(size_t)b] != 3, "sparse query term ids must be non-negative and strictly ascending within a query");
    if (flag[(size_t)b] || fail[(size_t)b]) sel.push_back(b);
  }
  if (sel.empty()) return false;
  if (getenv("HX_DEBUG_SP_VERBOSE")) {
    for (int b : sel) fprintf(stderr, "[hx] sparse query %d -> exact path (flag %d, fail %d)\n", b, flag[(size_t)b], fail[(size_t)b]);
  }
  h->sparse_fallbacks += (int64_t)sel.size();
  sparse_exact_fallback(h, q_indptr, q_idx, q_val, sel, L, out_keys, out_cnt, st);
  return true;
}

static void search_sparse(hx_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                          int B, int L, uint64_t* out_keys, int* out_cnt, hipStream_t st) {
  sparse_enqueue(h, q_indptr, q_idx, q_val, B, L, out_keys, out_cnt, st);
  sparse_resolve(h, q_indptr, q_idx, q_val, B, L, out_keys, out_cnt, st);
}

static void rescore(hx_index* h, const float* q_dev, int B, int prefix, const uint64_t* cand, int cstride,
                    const int* ccnt, int L, uint64_t* out_keys, int* out_cnt, hipStream_t st) {
  HX_CHECK(B > 0, "B must be positive");
  HX_CHECK(L >= 1 && L <= MAX_LIMIT, "limit out of range [1, 2048]");
  HX_CHECK(cstride >= 1 && cstride <= CAND_CAP, "candidate stride out of range");
  if (h->n == 0) return zero_outputs(out_keys, out_cnt, B, L, st);
  const MatrixRef m = pick_matrix(h, prefix);
  float* qn = (float*)h->ws.get(WS_QN, (size_t)B * m.dpad * 4);
  launch_prep_queries_f(q_dev, h->dim, B, B, m.d, m.dpad, qn, nullptr, st);
  uint64_t* tmp = (uint64_t*)h->ws.get(WS_RS_TMP, (size_t)B * cstride * 8);
  RescoreArgs r{};
  r.kind = KIND_F32;
  r.M = m.m32;
  r.row_stride = m.dpad;
  r.dim_pad = m.dpad;
  r.Q = qn;
  r.q_stride = m.dpad;
  r.n_rows = h->n;
  r.id_base = h->id_base;
  r.cand = cand;
  r.cnt = ccnt;
  r.stride = cstride;
  r.B = B;
  r.out = tmp;
  launch_rescore_list(r, st);
  launch_compact(tmp, cstride, ccnt, B, std::min(L, cstride), 1, out_keys, L, out_cnt, nullptr, cstride, st);
}

static void rrf(hx_index* h, const uint64_t* a, int as, const int* ac, const uint64_t* b, int bs,
                const int* bc, int B, float k, int base, int limit, uint64_t* out_keys, int* out_cnt,
                hipStream_t st, Workspace& ws) {
  (void)h;
  HX_CHECK(limit >= 1 && limit <= MAX_LIMIT, "rrf limit out of range");
  HX_CHECK(as + bs <= CAND_CAP, "rrf lists too long");
  if (launch_rrf_top(a, as, ac, b, bs, bc, B, k, base, std::min(limit, as + bs), out_keys, limit, out_cnt, st)) return;
  uint64_t* tmp = (uint64_t*)ws.get(WS_RRF_TMP, (size_t)B * (as + bs) * 8);
  int* tcnt = (int*)ws.get(WS_MISC, (size_t)B * 4);
  launch_rrf(a, as, ac, b, bs, bc, B, k, base, std::min(limit, as + bs), tmp, tcnt, st);
  // copy the first `limit` slots of each fused row
  launch_compact(tmp, as + bs, tcnt, B, std::min(limit, as + bs), 0, out_keys, limit, out_cnt, nullptr,
                 as + bs, st);
}

// The second stream of an index (the sparse stage beside the dense one) and its two events, made on first use.
// HX_DEBUG_ST2_PRIO=1: the stream gets the device's highest priority (its few, large workgroups are placed first).
static void ensure_side_stream(hx_index* h) {
  if (h->st2) return;
  static const bool hi = getenv("HX_DEBUG_ST2_PRIO") != nullptr && atoi(getenv("HX_DEBUG_ST2_PRIO")) != 0;
  if (hi) {
    int least = 0, greatest = 0;
    HX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HX_HIP(hipStreamCreateWithPriority(&h->st2, hipStreamNonBlocking, greatest));
  } else {
    HX_HIP(hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
  }
  HX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  HX_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
}

static void hybrid_query_dev(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix,
                             const float* qv, int B, const hx_params* p, uint64_t* out_keys, int* out_cnt,
                             hipStream_t st, bool tree_sync = false) {
  // A batch beyond 4096 queries goes through in equal slices: the scan's threshold table and the sparse launch
  // plan end there (measured on 10M x 768: B = 5000 in one piece 100 ms, in two slices 75 ms; slicing at 2048
  // instead changes nothing at 4096 and costs 3 % at 5000), and a slice reads the query CSR through the same offsets.
  constexpr int BATCH_MAX = 4096;
  if (B > BATCH_MAX) {
    const int parts = (B + BATCH_MAX - 1) / BATCH_MAX, per = (B + parts - 1) / parts;
    for (int b0 = 0; b0 < B; b0 += per) {
      const int nb = std::min(per, B - b0);
      hybrid_query_dev(h, qd + (int64_t)b0 * h->dim, qip + b0, qix, qv, nb, p, out_keys + (int64_t)b0 * p->final_limit,
                       out_cnt + b0, st, tree_sync);
    }
    return;
  }
  Workspace& w = h->ws;
  auto keys = [&](int slot, int L) { return (uint64_t*)w.get(slot, (size_t)B * L * 8); };
  auto cnts = [&](int slot) { return (int*)w.get(slot, (size_t)B * 4); };
  const float rk = p->rrf_k;
  if (p->mode == HX_MODE_H1) {
    uint64_t* D = keys(WS_T_A, p->dense_limit);
    int* Dc = cnts(WS_T_ACNT);
    uint64_t* S = keys(WS_T_B, p->sparse_limit);
    int* Sc = cnts(WS_T_BCNT);
    // the sparse stage and the fusion are enqueued before the host looks at the dense stage's
    // failure flags (no idle device while it does); the fusion is redone if a retry patched D
    auto fuse = [&]() {
      rrf(h, D, p->dense_limit, Dc, S, p->sparse_limit, Sc, B, rk, p->rrf_rank_base, p->final_limit, out_keys,
          out_cnt, st, w);
    };
    // Round 4: the sparse stage runs on a second stream BESIDE the dense stage.  First beside its tail only (behind the last
    // scan launch the dense stage is ~0.5 ms of small kernels -- log scatter, compactions, the exact re-score of 450 candidates
    // per query, the certificate -- none of which needs LDS): +0.8 %.  Then from the start of the call (below): the select pass
    // and the scans cannot share a CU (both want its whole LDS) but each fills the other's ragged ends, and the stage's own
    // small kernels hide behind the scans.  (Round 1's probe, scripts/overlap_probe.py, had said no: other kernels, and the scan
    // confined to a part of the chip.)
    bool forked = false, marked = false;
    auto mark = [&]() {                                  // the point of the caller's stream the second stream starts from
      if (!h->overlap_tail) return;
      ensure_side_stream(h);
      HX_HIP(hipEventRecord(h->ev_fork, st));
      HX_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
      marked = true;
    };
    auto fork = [&]() {
      if (!marked) mark();
      if (!marked) return;
      sparse_enqueue(h, qip, qix, qv, B, p->sparse_limit, S, Sc, h->st2);
      HX_HIP(hipEventRecord(h->ev_join, h->st2));
      forked = true;
    };
    // ... and, measured late in round 4 (scripts/h1_small_batch.py, profiles/r04_h1_small_batch.txt): started beside the
    // SCAN the step is shorter still at every batch size -- 2-32 queries 1.76-1.99 -> 1.56-1.77 ms, 1024 queries 9.58 -> 9.32
    // (the two kernels cannot share a CU, but each fills the other's ragged ends; round 2's probe predates both kernels)
    // From 128 queries on the HOST enqueues the dense scans first all the same (sp_host_late_min_b): the second stream is tied
    // to the START of the call (`mark`), its kernels are enqueued once the scan launches are in -- the first chunks of the scan
    // run before the select pass holds every CU (B = 1024: 9.33-9.40 -> 9.24-9.27 ms, B = 128: 1.99 -> 1.97; below that the
    // sparse stage is better off first: B = 8 / 32 1.57-1.77 against 1.59-1.80 ms).
    std::function<void()> after_scan = fork;
    if (B <= h->fork_early_max) {
      if (B >= h->sp_host_late_min_b) {
        mark();
      } else {
        fork();
        after_scan = std::function<void()>();
      }
    }
    struct Beside {                     // (reset on every way out of the stage, exceptions included)
      hx_index* h;
      ~Beside() { h->beside = false; }
    } beside_guard{h};
    h->beside = forked || marked;
    const bool patched = search_dense(h, qd, B, 0, p->dense_limit, D, Dc, st, 0, [&]() {
      if (forked) HX_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
      else sparse_enqueue(h, qip, qix, qv, B, p->sparse_limit, S, Sc, st);
      fuse();
    }, false, nullptr, after_scan);
    h->beside = false;
    const bool sp_patched = sparse_resolve(h, qip, qix, qv, B, p->sparse_limit, S, Sc, st);
    if (patched || sp_patched) fuse();
    return;
  }
  // The tree's three whole-collection stages (prefix scan, int8 scan, sparse) each end with a look at their failure
  // flags.  Speculatively they do not: every stage is enqueued with its flags deferred into ONE device word, the
  // host reads that word once behind the whole tree, and a batch with a flagged query (a retry or the exact path was
  // needed: rare) is run again the synchronous way.  Round 2 read the flags three times per batch, with the device
  // idle each time (146 us of a 12.2 ms step).
  static const bool no_spec = getenv("HX_DEBUG_TREE_SYNC") != nullptr;
  int* tflag = nullptr;
  if (!no_spec && !tree_sync && !h->tree_spec_off) {
    tflag = (int*)w.get(WS_TREE_FLAG, 4);
    HX_HIP(hipMemsetAsync(tflag, 0, 4, st));
  }
  bool sp_enqueued = false;
  std::function<void()> sparse_late;      // the sparse stage's launches, when the host enqueues them behind the first scan's
  auto dense_stage = [&](int prefix, int L, uint64_t* ok, int* oc) {
    if (tflag) {
      std::function<void()> hook = std::move(sparse_late);
      sparse_late = nullptr;
      search_dense(h, qd, B, prefix, L, ok, oc, st, 0, std::function<void()>(), true, tflag, hook);
      if (hook && !sp_enqueued) hook();   // (an empty index: no scan was launched)
    } else {
      search_dense(h, qd, B, prefix, L, ok, oc, st);
    }
  };
  // --- matryoshka cascade (qdrant_handler.py:305-330)
  const int lim[3] = {p->matryoshka_64_limit, p->matryoshka_128_limit, p->matryoshka_256_limit};
  uint64_t* A = keys(WS_T_A, p->dense_limit);
  int* Ac = cnts(WS_T_ACNT);
  // --- sparse (:347-354), speculative tree: enqueued FIRST, on the second stream, beside the two whole-collection dense
  // scans (as in H1 above); joined in front of the RRF
  uint64_t* S = keys(WS_T_B, p->sparse_limit);
  int* Sc = cnts(WS_T_BCNT);
  bool sp_forked = false;
  if (tflag && h->overlap_tail && B <= h->fork_early_max) {
    ensure_side_stream(h);
    HX_HIP(hipEventRecord(h->ev_fork, st));
    HX_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
    auto enqueue = [&, S, Sc]() {
      if (sp_enqueued) return;
      sp_enqueued = true;
      sparse_enqueue(h, qip, qix, qv, B, p->sparse_limit, S, Sc, h->st2);
      h->sp_sum_pending = false;        // its summary is read with the tree's flag word below
      h->sp_sum_fetched = false;
      HX_HIP(hipEventRecord(h->ev_join, h->st2));
    };
    if (B >= h->sp_host_late_min_b) sparse_late = enqueue;   // (as in H1: the host enqueues the first scan's launches first)
    else enqueue();
    sp_forked = true;
  }
  struct Beside {
    hx_index* h;
    ~Beside() { h->beside = false; }
  } beside_guard{h};
  h->beside = sp_forked;
  if (h->n_pre == 0) {
    dense_stage(0, p->dense_limit, A, Ac);
  } else {
    uint64_t* c0 = keys(WS_T_C, lim[0]);
    int* c0c = cnts(WS_T_CCNT);
    dense_stage(h->psize[0], lim[0], c0, c0c);
    uint64_t* cur = c0;
    int* curc = c0c;
    int curL = lim[0];
    const int slots[2][2] = {{WS_T_D, WS_T_DCNT}, {WS_T_E, WS_T_ECNT}};
    for (int s = 1; s < h->n_pre; ++s) {
      uint64_t* nx = keys(slots[s - 1][0], lim[s]);
      int* nxc = cnts(slots[s - 1][1]);
      rescore(h, qd, B, h->psize[s], cur, curL, curc, lim[s], nx, nxc, st);
      cur = nx;
      curc = nxc;
      curL = lim[s];
    }
    rescore(h, qd, B, 0, cur, curL, curc, p->dense_limit, A, Ac, st);
  }
  // --- quantized -> dense refinement (:333-344)
  uint64_t* Qc = keys(WS_T_C, p->quantized_limit);
  int* Qcc = cnts(WS_T_CCNT);
  search_i8(h, qd, B, p->quantized_limit, Qc, Qcc, st, 0, tflag);
  uint64_t* Dq = keys(WS_T_D, p->dense_limit);
  int* Dqc = cnts(WS_T_DCNT);
  rescore(h, qd, B, 0, Qc, p->quantized_limit, Qcc, p->dense_limit, Dq, Dqc, st);
  // --- sparse (:347-354)
  h->beside = false;
  if (sp_forked) {
    HX_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
  } else if (tflag) {
    sparse_enqueue(h, qip, qix, qv, B, p->sparse_limit, S, Sc, st);
    h->sp_sum_pending = false;          // its summary is read with the tree's flag word below
    h->sp_sum_fetched = false;
  } else {
    search_sparse(h, qip, qix, qv, B, p->sparse_limit, S, Sc, st);
  }
  // --- RRF (:357-360)
  uint64_t* R = keys(WS_T_E, p->rrf_limit);
  int* Rc = cnts(WS_T_ECNT);
  rrf(h, Dq, p->dense_limit, Dqc, S, p->sparse_limit, Sc, B, rk, p->rrf_rank_base, p->rrf_limit, R, Rc, st, w);
  // --- root: union re-scored by dense cosine (:363-372)
  const int us = p->dense_limit + p->rrf_limit;
  uint64_t* U = keys(WS_T_F, us);
  launch_concat(A, p->dense_limit, Ac, R, p->rrf_limit, Rc, B, U, st);
  rescore(h, qd, B, 0, U, us, nullptr, p->final_limit, out_keys, out_cnt, st);
  if (tflag) {                           // the one look at the flags
    int* pin = host_pin(h);
    pin[1] = pin[2] = 0;
    HX_HIP(hipMemcpyAsync(pin, tflag, 4, hipMemcpyDeviceToHost, st));
    if (h->sp_base.n_segments || h->sp_tail.n_segments)
      HX_HIP(hipMemcpyAsync(pin + 1, w.get(WS_SP_SUM, 8), 8, hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    const bool redo = (pin[0] | pin[1] | pin[2]) != 0;
    // the guard: 4 redone batches within a window of 16 and the tree stops speculating on this collection
    h->tree_win_n += 1;
    h->tree_win_redone += redo ? 1 : 0;
    if (h->tree_win_redone >= 4) h->tree_spec_off = true;
    if (h->tree_win_n >= 16) h->tree_win_n = h->tree_win_redone = 0;
    if (redo) {      // some list is not final: the batch again, every stage resolving its own flags
      h->tree_redone += 1;
      hybrid_query_dev(h, qd, qip, qix, qv, B, p, out_keys, out_cnt, st, true);
    }
  }
}

static void check_params(const hx_params* p) {
  HX_CHECK(p != nullptr, "params is NULL");
  HX_CHECK(p->mode == HX_MODE_TREE || p->mode == HX_MODE_H1, "unknown mode");
  auto ok = [](int v) { return v >= 1 && v <= MAX_LIMIT; };
  HX_CHECK(ok(p->dense_limit) && ok(p->sparse_limit) && ok(p->final_limit), "limit out of range [1, 2048]");
  if (p->mode == HX_MODE_TREE) {
    HX_CHECK(ok(p->matryoshka_64_limit) && ok(p->matryoshka_128_limit) && ok(p->matryoshka_256_limit) &&
                 ok(p->quantized_limit) && ok(p->rrf_limit),
             "limit out of range [1, 2048]");
  }
}

}  // namespace hx

// =================================================================================
// C ABI
// =================================================================================
#define HX_TRY try {
#define HX_CATCH                                  \
  }                                               \
  catch (const std::exception& e) {               \
    hx::set_last_error(e.what());                 \
    return 1;                                     \
  }                                               \
  catch (...) {                                   \
    hx::set_last_error("unknown error");          \
    return 1;                                     \
  }                                               \
  return 0;

extern "C" {

const char* hx_last_error(void) { return hx::g_last_error.c_str(); }
int hx_abi_version(void) { return HX_ABI_VERSION; }

int hx_create(int32_t dim, const int32_t* msizes, int32_t n_msizes, int32_t device, int64_t id_base,
              hx_index** out) {
  HX_TRY
  HX_CHECK(out != nullptr, "out is NULL");
  HX_CHECK(dim >= 1 && dim <= 4096, "dim out of range [1, 4096]");
  HX_CHECK(n_msizes >= 0 && n_msizes <= 3, "at most 3 matryoshka sizes");
  HX_CHECK(id_base >= 0 && id_base < 0xFFFFFFFFll, "id_base out of range");
  int ndev = 0;
  HX_HIP(hipGetDeviceCount(&ndev));
  HX_CHECK(ndev > 0, "no HIP device: libhx has no CPU path");
  HX_CHECK(device >= 0 && device < ndev, "bad device ordinal");
  auto* h = new hx_index();
  h->dim = dim;
  h->dim_pad = (int)round_up(dim, 64);
  h->dim_pad8 = (int)round_up(dim, 128);
  h->device = device;
  h->id_base = id_base;
  h->n_pre = n_msizes;
  for (int i = 0; i < n_msizes; ++i) {
    if (!(msizes[i] % 64 == 0 && msizes[i] >= 64 && msizes[i] <= dim && (i == 0 || msizes[i] > msizes[i - 1]))) {
      delete h;
      throw Error("matryoshka sizes must be ascending multiples of 64, <= dim");
    }
    h->psize[i] = msizes[i];
  }
  if (const char* e = getenv("HX_DENSE_CAND")) h->cand8 = strcmp(e, "f16") == 0 ? 0 : 1;   // candidate pass: i8 (default) | f16
  if (const char* e = getenv("HX_DEBUG_SP_CUTSTEP")) h->sp_cut_step = std::max(0, atoi(e));
  if (const char* e = getenv("HX_DEBUG_BN32_MAX")) h->bn32_max = std::max(0, atoi(e));       // tests / diagnostics: scan routing
  if (const char* e = getenv("HX_DEBUG_BN64_MAX")) h->bn64_max = std::max(0, atoi(e));
  if (const char* e = getenv("HX_DEBUG_BN128_MAX")) h->bn128_max = std::max(32, atoi(e));
  h->no_hq = getenv("HX_DEBUG_NO_HQ") != nullptr;
  h->overlap_tail = getenv("HX_DEBUG_NO_OVERLAP") == nullptr;
  if (const char* e = getenv("HX_DEBUG_FORK_EARLY_MAX")) h->fork_early_max = atoi(e);
  if (const char* e = getenv("HX_DEBUG_SCAN_OVERSUB")) h->scan_oversub = atoi(e);
  if (const char* e = getenv("HX_DEBUG_SP_HOST_LATE_MIN_B")) h->sp_host_late_min_b = atoi(e);
  if (const char* e = getenv("HX_DEBUG_SEG_DOCS")) {       // tests: force a segment size
    const int v = atoi(e);
    if (v == SEG_DOCS_SMALL || v == SEG_DOCS_LARGE) h->seg_docs_force = v;
  }
  if (const char* e = getenv("HX_DEBUG_TAIL_MIN")) h->tail_min_force = atoll(e);   // tests: force / forbid a tail index
  if (const char* e = getenv("HX_DEBUG_SCAN8_LOGCAP")) {   // tests: force the log-overflow path
    const int v = atoi(e);
    if (v >= 1 && v <= SCAN8_LOGCAP) h->scan_logcap = v;
  }
  h->set_device();
  *out = h;
  HX_CATCH
}

int hx_destroy(hx_index* h) {
  HX_TRY
  if (!h) return 0;
  h->set_device();
  (void)hipDeviceSynchronize();
  free_sparse_index(h);
  void* ptrs[] = {h->dense, h->dense_h, h->q8, h->q8_rinv, h->pre[0], h->pre[1], h->pre[2], h->pre_h0,
                  h->sp_indptr, h->sp_idx, h->sp_val, h->sp_counter, h->q8s, h->q8s_scale, h->s8_err,
                  h->tm_q8.v, h->tm_q8s.v};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->st2) (void)hipStreamDestroy(h->st2);
  if (h->ids.dev) (void)hipFree(h->ids.dev);
  h->ws.release();
  delete h;
  HX_CATCH
}

int hx_reserve(hx_index* h, int64_t n_rows, int64_t nnz) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  h->set_device();
  reserve_rows(h, n_rows);
  if (nnz > 0) reserve_sparse(h, n_rows, nnz);
  HX_CATCH
}

// ---- ingest (store_document_vectors -> upsert, qdrant_handler.py:120-198) -----------------------------
static void add_dense_host(hx_index* h, const float* rows_host, int64_t n) {
  HX_CHECK(n >= 0, "n < 0");
  if (n == 0) return;
  HX_CHECK(rows_host, "rows is NULL");
  HX_CHECK(h->id_base + h->n + n < 0xFFFFFFFFll, "row ids must stay below 2^32 - 1");
  (void)ids_next(h, n);
  h->set_device();
  reserve_rows(h, h->n + n);
  const int64_t CH = 65536;
  float* raw = (float*)h->ws.get(WS_RAW, (size_t)std::min(n, CH) * h->dim * 4);
  const int64_t n_before = h->n;
  try {
    for (int64_t r0 = 0; r0 < n; r0 += CH) {
      const int64_t m = std::min(CH, n - r0);
      HX_HIP(hipMemcpy(raw, rows_host + r0 * h->dim, (size_t)m * h->dim * 4, hipMemcpyHostToDevice));
      prep_rows_device(h, raw, m, nullptr);
      HX_HIP(hipStreamSynchronize(nullptr));
      h->n += m;
    }
  } catch (...) {
    h->n = n_before;   // all or nothing
    throw;
  }
  ids_commit(h, n_before, n);
}

// the same for rows that already lie on the device (an encoder's output): read in place, no staging copy
static void add_dense_dev(hx_index* h, const float* rows_dev, int64_t n, hipStream_t st) {
  HX_CHECK(n >= 0, "n < 0");
  if (n == 0) return;
  HX_CHECK(rows_dev, "rows is NULL");
  HX_CHECK(h->id_base + h->n + n < 0xFFFFFFFFll, "row ids must stay below 2^32 - 1");
  (void)ids_next(h, n);
  h->set_device();
  HX_HIP(hipStreamSynchronize(st));   // reserve_rows may move the stores; the caller's rows must be complete anyway
  reserve_rows(h, h->n + n);
  const int64_t CH = 65536;
  const int64_t n_before = h->n;
  try {
    for (int64_t r0 = 0; r0 < n; r0 += CH) {
      const int64_t m = std::min(CH, n - r0);
      prep_rows_device(h, rows_dev + r0 * h->dim, m, st);
      h->n += m;
    }
    HX_HIP(hipStreamSynchronize(st));
  } catch (...) {
    h->n = n_before;   // all or nothing
    throw;
  }
  ids_commit(h, n_before, n);
}

// Largest |value| a sparse vector may hold: products q_t * d_t then stay far inside fp32.
constexpr float SPARSE_ABS_MAX = 1.0e18f;

// Sparse vectors of the NEXT n rows (paired with dense rows by position).  Validates everything on the
// host before anything is committed: monotone indptr, term ids in [0, 2^31) and unique per vector, finite
// values with |v| <= SPARSE_ABS_MAX.  Rows that were added without a sparse vector before are padded as
// empty documents first, so a sparse vector can never attach to the wrong row.
static void add_sparse_host(hx_index* h, const int64_t* indptr, const int32_t* idx, const float* val, int64_t n) {
  HX_CHECK(n >= 0, "n < 0");
  if (n == 0) return;
  HX_CHECK(indptr, "indptr is NULL");
  HX_CHECK(indptr[0] == 0, "indptr[0] must be 0");
  const int64_t nnz = indptr[n];
  HX_CHECK(nnz >= 0, "negative nnz");
  HX_CHECK(nnz == 0 || (idx && val), "idx/val is NULL");
  HX_CHECK(h->sp_rows <= h->n, "sparse rows are ahead of the dense rows: add the dense rows of the previous batch first");
  for (int64_t r = 0; r < n; ++r) HX_CHECK(indptr[r + 1] >= indptr[r] && indptr[r + 1] <= nnz, "indptr not monotone");
  h->set_device();
  HX_CHECK(h->nnz + nnz < 0xFFFFFFFFll, "nnz per shard must stay below 2^32");
  const int64_t pad = h->n - h->sp_rows;      // dense-only rows so far: empty documents
  reserve_sparse(h, h->sp_rows + pad + n, h->nnz + nnz);
  std::vector<int64_t> ip((size_t)(pad + n));
  for (int64_t r = 0; r < pad; ++r) ip[(size_t)r] = h->nnz;
  for (int64_t r = 0; r < n; ++r) ip[(size_t)(pad + r)] = h->nnz + indptr[r + 1];
  if (h->sp_rows == 0) {
    const int64_t zero = 0;
    HX_HIP(hipMemcpy(h->sp_indptr, &zero, 8, hipMemcpyHostToDevice));
  }
  // The batch goes to the device BEHIND the committed rows and is checked there -- term ids in [0, 2^31) and unique
  // within a vector (Qdrant rejects duplicates), finite values with |v| <= SPARSE_ABS_MAX -- before the counters move:
  // a refused batch leaves nothing behind.  (The per-row sort this replaces was the ingest rate: 0.2 s of one host
  // core per 131072 chunks against 2 ms of kernels.)
  HX_HIP(hipMemcpy(h->sp_indptr + h->sp_rows + 1, ip.data(), ip.size() * 8, hipMemcpyHostToDevice));
  float lo = 0.f, hi = 0.f;
  if (nnz) {
    HX_HIP(hipMemcpy(h->sp_idx + h->nnz, idx, (size_t)nnz * 4, hipMemcpyHostToDevice));
    HX_HIP(hipMemcpy(h->sp_val + h->nnz, val, (size_t)nnz * 4, hipMemcpyHostToDevice));
    constexpr int LONG_CAP = 4096;
    int64_t* long_rows = (int64_t*)h->ws.get(WS_LONG_ROWS, (size_t)LONG_CAP * 8 + 32);
    int* flags = (int*)(long_rows + LONG_CAP);             // [0] bad bits, [1] rows too long for the wave compare
    uint32_t* mm = (uint32_t*)(flags + 2);                 // [0..2] min, max (orderable), non-finite count
    const uint32_t init[5] = {0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    HX_HIP(hipMemcpy(flags, init, 20, hipMemcpyHostToDevice));
    const int64_t* rows_ip = h->sp_indptr + h->sp_rows + pad;
    launch_csr_check(rows_ip, h->sp_idx, n, h->nnz, h->nnz + nnz, flags, nullptr);
    launch_csr_unique(rows_ip, h->sp_idx, n, flags, long_rows, LONG_CAP, flags + 1, nullptr);
    launch_minmax_f32(h->sp_val + h->nnz, nnz, (float*)mm, nullptr);
    uint32_t got[5];
    HX_HIP(hipMemcpy(got, flags, 20, hipMemcpyDeviceToHost));
    HX_CHECK((got[0] & 1) == 0, "indptr not monotone");
    HX_CHECK((got[0] & 2) == 0, "sparse index out of range [0, 2^31)");
    HX_CHECK((got[0] & 4) == 0, "sparse indices must be unique within a vector");
    if (got[1] > 0) {          // vectors of more than CSR_UNIQUE_WAVE_MAX terms: sorted here, from the caller's arrays
      std::vector<int32_t> tmp;
      for (int64_t r = 0; r < n; ++r) {
        if (indptr[r + 1] - indptr[r] <= CSR_UNIQUE_WAVE_MAX) continue;
        tmp.assign(idx + indptr[r], idx + indptr[r + 1]);
        std::sort(tmp.begin(), tmp.end());
        HX_CHECK(std::adjacent_find(tmp.begin(), tmp.end()) == tmp.end(), "sparse indices must be unique within a vector");
      }
    }
    HX_CHECK(got[4] == 0, "sparse values must be finite and at most 1e18 in magnitude");
    lo = orderable_f32(got[2]);
    hi = orderable_f32(got[3]);
    HX_CHECK(lo >= -SPARSE_ABS_MAX && hi <= SPARSE_ABS_MAX, "sparse values must be finite and at most 1e18 in magnitude");
    h->sp_wmin = h->sp_have_w ? std::min(h->sp_wmin, lo) : lo;
    h->sp_wmax = h->sp_have_w ? std::max(h->sp_wmax, hi) : hi;
    h->sp_have_w = true;
  }
  h->sp_rows += pad + n;
  h->nnz += nnz;
  h->sparse_stale = true;
}

int hx_add_dense(hx_index* h, const float* rows_host, int64_t n) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  NextIdGuard guard{h};
  add_dense_host(h, rows_host, n);
  HX_CATCH
}

int hx_add_dense_dev(hx_index* h, const float* rows_dev, int64_t n, void* stream) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  NextIdGuard guard{h};
  add_dense_dev(h, rows_dev, n, (hipStream_t)stream);
  HX_CATCH
}

int hx_set_next_id(hx_index* h, int64_t first_id) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  HX_CHECK(first_id >= 0 && first_id < 0xFFFFFFFFll, "id out of range [0, 2^32 - 1)");
  const int64_t cont = h->ids.end >= 0 ? h->ids.end : h->id_base;
  HX_CHECK(first_id >= cont, "hx_set_next_id: the ids of a shard must ascend with its rows");
  h->ids.next = first_id;
  HX_CATCH
}

int hx_add_sparse(hx_index* h, const int64_t* indptr, const int32_t* idx, const float* val, int64_t n) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  add_sparse_host(h, indptr, idx, val, n);
  HX_CATCH
}

// a chunk's dense and sparse vectors are committed together: a failure of either leaves the index as it was
static void add_rows_atomic(hx_index* h, const int64_t* indptr, const int32_t* idx, const float* val, int64_t n,
                            const std::function<void()>& add_dense) {
  HX_CHECK(n >= 0, "n < 0");
  if (n == 0) return;
  HX_CHECK(h->id_base + h->n + n < 0xFFFFFFFFll, "row ids must stay below 2^32 - 1");
  (void)ids_next(h, n);
  const int64_t sp_rows0 = h->sp_rows, nnz0 = h->nnz;
  const bool stale0 = h->sparse_stale, have0 = h->sp_have_w;
  const float lo0 = h->sp_wmin, hi0 = h->sp_wmax;
  if (indptr) add_sparse_host(h, indptr, idx, val, n);
  try {
    add_dense();
  } catch (...) {
    h->sp_rows = sp_rows0;
    h->nnz = nnz0;
    h->sparse_stale = stale0;
    h->sp_have_w = have0;
    h->sp_wmin = lo0;
    h->sp_wmax = hi0;
    throw;
  }
}

int hx_add_rows(hx_index* h, const float* rows_host, const int64_t* indptr, const int32_t* idx, const float* val,
                int64_t n) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  NextIdGuard guard{h};
  HX_CHECK(n <= 0 || rows_host, "rows is NULL");
  add_rows_atomic(h, indptr, idx, val, n, [&]() { add_dense_host(h, rows_host, n); });
  HX_CATCH
}

int hx_add_rows_dev(hx_index* h, const float* rows_dev, const int64_t* indptr, const int32_t* idx, const float* val,
                    int64_t n, void* stream) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  NextIdGuard guard{h};
  HX_CHECK(n <= 0 || rows_dev, "rows is NULL");
  add_rows_atomic(h, indptr, idx, val, n, [&]() { add_dense_dev(h, rows_dev, n, (hipStream_t)stream); });
  HX_CATCH
}

// Roll the collection back to its first n_rows rows (a batch that one shard of a sharded collection could not
// store is undone on the shards that did store it; qdrant_handler.py:190-193 upserts a batch as one request).
int hx_truncate(hx_index* h, int64_t n_rows) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  HX_CHECK(n_rows >= 0 && n_rows <= h->n, "hx_truncate: n_rows out of range [0, count]");
  h->set_device();
  HX_HIP(hipDeviceSynchronize());
  h->ids.next = -1;
  if (n_rows == h->n && h->sp_rows <= n_rows) return 0;
  h->n = n_rows;
  h->tm_q8.rows = h->tm_q8s.rows = -1;
  if (h->sp_rows > n_rows) {
    int64_t nnz = 0;
    if (n_rows > 0) HX_HIP(hipMemcpy(&nnz, h->sp_indptr + n_rows, 8, hipMemcpyDeviceToHost));
    h->sp_rows = n_rows;
    h->nnz = nnz;
  }
  // the inverted index: a base that reaches past the cut is rebuilt, the tail always is.  The weight range stays
  // as it was: an upper bound of the remaining weights is all the select pass needs.
  free_sparse_ix(h->sp_tail);
  if (h->sp_base.n_docs > h->sp_rows) free_sparse_ix(h->sp_base);
  h->sparse_stale = true;
  auto& b = h->ids;
  while (!b.row0.empty() && (int64_t)b.row0.back() >= n_rows) {
    b.row0.pop_back();
    b.gid0.pop_back();
  }
  b.end = b.row0.empty() ? -1 : (int64_t)b.gid0.back() + (n_rows - (int64_t)b.row0.back());
  b.dirty = true;
  HX_CATCH
}

int hx_finalize(hx_index* h) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  h->set_device();
  finalize(h, nullptr);
  HX_CATCH
}

// the inverted index again from the document-major CSR (what the first search after a bulk ingest, or hx_finalize, does
// once): a measurement aid -- the first build of a fresh process also pays for its 32 GB of temporary allocations
int hx_rebuild_sparse(hx_index* h) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  h->set_device();
  HX_HIP(hipDeviceSynchronize());
  free_sparse_index(h);
  h->sparse_stale = true;
  finalize(h, nullptr);
  HX_CATCH
}

int hx_count(hx_index* h, int64_t* n_rows) {
  HX_TRY
  HX_CHECK(h && n_rows, "NULL argument");
  *n_rows = h->n;
  HX_CATCH
}
int hx_nnz(hx_index* h, int64_t* nnz) {
  HX_TRY
  HX_CHECK(h && nnz, "NULL argument");
  *nnz = h->nnz;
  HX_CATCH
}

int hx_synth_fill(hx_index* h, int64_t n, uint32_t seed_dense, uint32_t seed_sparse, const uint32_t* cdf,
                  int32_t V, const uint16_t* len_tab, int32_t with_sparse) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  HX_CHECK(n >= 0, "n < 0");
  if (n == 0) return 0;
  HX_CHECK(h->id_base + h->n + n < 0xFFFFFFFFll, "row ids must stay below 2^32 - 1");
  NextIdGuard guard{h};
  (void)ids_next(h, n);
  h->set_device();
  hipStream_t st = nullptr;
  reserve_rows(h, h->n + n);
  const int64_t row0 = h->n;
  if (with_sparse) {
    HX_CHECK(cdf && len_tab && V > 1, "sparse tables missing");
    HX_CHECK(h->sp_rows == h->n, "synthetic sparse fill needs one sparse row per dense row so far");
    uint32_t* dcdf = (uint32_t*)h->ws.get(WS_MISC, (size_t)V * 4 + 512);
    uint16_t* dlen = (uint16_t*)((uint8_t*)dcdf + (size_t)V * 4);
    HX_HIP(hipMemcpy(dcdf, cdf, (size_t)V * 4, hipMemcpyHostToDevice));
    HX_HIP(hipMemcpy(dlen, len_tab, 512, hipMemcpyHostToDevice));
    int64_t* per = (int64_t*)h->ws.get(WS_SYN_NNZ, (size_t)(n + 1) * 8 * 2);
    int64_t* ip = per + (n + 1);
    HX_HIP(hipMemsetAsync(per, 0, (size_t)(n + 1) * 8, st));
    synth_sparse_count(h->id_base + row0, n, seed_sparse, dcdf, V, dlen, per, st);
    exclusive_scan_i64(per, ip, n, st);
    int64_t add = 0;
    HX_HIP(hipMemcpy(&add, ip + n, 8, hipMemcpyDeviceToHost));
    HX_CHECK(h->nnz + add < 0xFFFFFFFFll, "nnz per shard must stay below 2^32");
    reserve_sparse(h, h->sp_rows + n, h->nnz + add);
    synth_sparse_fill(h->id_base + row0, n, seed_sparse, dcdf, V, dlen, ip, h->sp_idx + h->nnz,
                      h->sp_val + h->nnz, st);
    // global indptr = base nnz + local
    std::vector<int64_t> hip_((size_t)(n + 1));
    HX_HIP(hipMemcpy(hip_.data(), ip, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
    for (auto& v : hip_) v += h->nnz;
    HX_HIP(hipMemcpy(h->sp_indptr + h->sp_rows, hip_.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
    track_weights_dev(h, h->sp_val + h->nnz, add, st);
    h->sp_rows += n;
    h->nnz += add;
    h->sparse_stale = true;
  }
  const int64_t CH = 65536;
  float* raw = (float*)h->ws.get(WS_RAW, (size_t)std::min(n, CH) * h->dim * 4);
  for (int64_t r0 = 0; r0 < n; r0 += CH) {
    const int64_t m = std::min(CH, n - r0);
    launch_synth_dense(raw, h->id_base + h->n, m, h->dim, seed_dense, st);
    prep_rows_device(h, raw, m, st);
    h->n += m;
  }
  HX_HIP(hipStreamSynchronize(st));
  ids_commit(h, row0, n);
  HX_CATCH
}

int hx_synth_queries_dense(int32_t dim, int64_t q0, int32_t B, uint32_t seed, float* q_dev, void* stream) {
  HX_TRY
  HX_CHECK(q_dev && B > 0 && dim > 0, "bad argument");
  launch_synth_dense(q_dev, q0, B, dim, seed, (hipStream_t)stream);
  HX_CATCH
}

int hx_search_dense(hx_index* h, const float* q_dev, int32_t B, int32_t prefix, int32_t limit,
                    uint64_t* keys_dev, int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_dev && keys_dev && counts_dev, "NULL argument");
  h->set_device();
  search_dense(h, q_dev, B, prefix, limit, keys_dev, counts_dev, (hipStream_t)stream);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

int hx_search_i8(hx_index* h, const float* q_dev, int32_t B, int32_t limit, uint64_t* keys_dev,
                 int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_dev && keys_dev && counts_dev, "NULL argument");
  h->set_device();
  search_i8(h, q_dev, B, limit, keys_dev, counts_dev, (hipStream_t)stream);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

int hx_search_sparse(hx_index* h, const int64_t* q_indptr_dev, const int32_t* q_idx_dev,
                     const float* q_val_dev, int32_t B, int32_t limit,
                     uint64_t* keys_dev, int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_indptr_dev && keys_dev && counts_dev, "NULL argument");
  h->set_device();
  search_sparse(h, q_indptr_dev, q_idx_dev, q_val_dev, B, limit, keys_dev, counts_dev, (hipStream_t)stream);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

// ---- the whole-collection stages without their host round trip (hx.h: hx_*_async) -------------------------------
int hx_search_dense_async(hx_index* h, const float* q_dev, int32_t B, int32_t prefix, int32_t limit, uint64_t* keys_dev,
                          int32_t* counts_dev, int32_t* flag_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_dev && keys_dev && counts_dev && flag_dev, "NULL argument");
  h->set_device();
  search_dense(h, q_dev, B, prefix, limit, keys_dev, counts_dev, (hipStream_t)stream, 0, std::function<void()>(), true,
               flag_dev);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

int hx_search_i8_async(hx_index* h, const float* q_dev, int32_t B, int32_t limit, uint64_t* keys_dev, int32_t* counts_dev,
                       int32_t* flag_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_dev && keys_dev && counts_dev && flag_dev, "NULL argument");
  h->set_device();
  search_i8(h, q_dev, B, limit, keys_dev, counts_dev, (hipStream_t)stream, 0, flag_dev);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

int hx_search_sparse_async(hx_index* h, const int64_t* q_indptr_dev, const int32_t* q_idx_dev, const float* q_val_dev,
                           int32_t B, int32_t limit, uint64_t* keys_dev, int32_t* counts_dev, int32_t* flag_dev,
                           void* stream) {
  HX_TRY
  HX_CHECK(h && q_indptr_dev && keys_dev && counts_dev && flag_dev, "NULL argument");
  h->set_device();
  hipStream_t st = (hipStream_t)stream;
  int* spsum = (int*)h->ws.get(WS_SP_SUM, 8);
  HX_HIP(hipMemsetAsync(spsum, 0, 8, st));            // (an index without postings enqueues nothing)
  sparse_enqueue(h, q_indptr_dev, q_idx_dev, q_val_dev, B, limit, keys_dev, counts_dev, st);
  h->sp_sum_pending = false;                          // nobody will call sparse_resolve for this batch
  h->sp_sum_fetched = false;
  launch_flag_add(flag_dev, spsum, st);
  remap_out(h, keys_dev, (int64_t)B * limit, st);
  HX_CATCH
}

int hx_rescore(hx_index* h, const float* q_dev, int32_t B, int32_t prefix, const uint64_t* cand_keys_dev,
               int32_t cand_stride, const int32_t* cand_counts_dev, int32_t limit, uint64_t* keys_dev,
               int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && q_dev && cand_keys_dev && keys_dev && counts_dev, "NULL argument");
  h->set_device();
  HX_CHECK(B > 0 && cand_stride >= 1 && cand_stride <= CAND_CAP, "candidate stride out of range");
  const uint64_t* cand = remap_in(h, cand_keys_dev, (int64_t)B * cand_stride, (hipStream_t)stream);
  rescore(h, q_dev, B, prefix, cand, cand_stride, cand_counts_dev, limit, keys_dev, counts_dev,
          (hipStream_t)stream);
  remap_out(h, keys_dev, (int64_t)B * limit, (hipStream_t)stream);
  HX_CATCH
}

// Workspace of the index-free entries (hx_rrf, hx_merge, hx_h1_fuse): one per (device, calling thread), so two
// threads fusing lists on one device never share scratch (buffers of one thread are reused in stream order); a
// thread that exits returns its workspaces to the pool (hx::WorkspacePool above).
static hx::Workspace& static_ws(int device) {
  static thread_local hx::ThreadWorkspaces w;
  return w.get(device);
}

int hx_rrf(int32_t device, const uint64_t* a, int32_t as, const int32_t* ac, const uint64_t* b, int32_t bs,
           const int32_t* bc, int32_t B, float rrf_k, int32_t rank_base, int32_t limit, uint64_t* keys_dev,
           int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(a && ac && b && bc && keys_dev && counts_dev && B > 0, "bad argument");
  HX_HIP(hipSetDevice(device));
  rrf(nullptr, a, as, ac, b, bs, bc, B, rrf_k, rank_base, limit, keys_dev, counts_dev, (hipStream_t)stream,
      static_ws(device));
  HX_CATCH
}

int hx_merge(int32_t device, const uint64_t* in_keys, int32_t stride, const int32_t* in_counts, int32_t B,
             int32_t limit, int32_t dedupe, uint64_t* keys_dev, int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(in_keys && keys_dev && counts_dev && B > 0, "bad argument");
  HX_CHECK(stride >= 1 && stride <= CAND_CAP, "merge stride out of range [1, 8192]");
  HX_CHECK(limit >= 1 && limit <= MAX_LIMIT, "limit out of range [1, 2048]");
  HX_HIP(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  // compact cannot work in place across distinct strides: overlapping buffers go through scratch
  const uint64_t* src = in_keys;
  if (in_keys < keys_dev + (size_t)B * limit && keys_dev < in_keys + (size_t)B * stride) {
    uint64_t* tmp = (uint64_t*)static_ws(device).get(WS_RS_TMP, (size_t)B * stride * 8);
    HX_HIP(hipMemcpyAsync(tmp, in_keys, (size_t)B * stride * 8, hipMemcpyDeviceToDevice, st));
    src = tmp;
  }
  launch_compact(const_cast<uint64_t*>(src), stride, in_counts, B, std::min(limit, stride), dedupe, keys_dev, limit,
                 counts_dev, nullptr, stride, st);
  HX_CATCH
}

int hx_h1_local(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix, const float* qv, int32_t B,
                int32_t dense_limit, int32_t sparse_limit, uint64_t* keys_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && qd && qip && keys_dev && B > 0, "bad argument");
  h->set_device();
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = h->ws;
  uint64_t* D = (uint64_t*)w.get(WS_T_A, (size_t)B * dense_limit * 8);
  int* Dc = (int*)w.get(WS_T_ACNT, (size_t)B * 4);
  uint64_t* S = (uint64_t*)w.get(WS_T_B, (size_t)B * sparse_limit * 8);
  int* Sc = (int*)w.get(WS_T_BCNT, (size_t)B * 4);
  auto pack = [&]() { launch_concat(D, dense_limit, Dc, S, sparse_limit, Sc, B, keys_dev, st); };
  // as in hybrid_query_dev: sparse + packing are enqueued before the host reads the dense flags
  const bool patched = search_dense(h, qd, B, 0, dense_limit, D, Dc, st, 0, [&]() {
    sparse_enqueue(h, qip, qix, qv, B, sparse_limit, S, Sc, st);
    pack();
  });
  const bool sp_patched = sparse_resolve(h, qip, qix, qv, B, sparse_limit, S, Sc, st);
  if (patched || sp_patched) pack();
  remap_out(h, keys_dev, (int64_t)B * (dense_limit + sparse_limit), st);
  HX_CATCH
}

int hx_h1_local_async(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix, const float* qv, int32_t B,
                      int32_t dense_limit, int32_t sparse_limit, uint64_t* keys_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && qd && qip && keys_dev && B > 0, "bad argument");
  h->set_device();
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = h->ws;
  const int L2 = dense_limit + sparse_limit;
  uint64_t* D = (uint64_t*)w.get(WS_T_A, (size_t)B * dense_limit * 8);
  int* Dc = (int*)w.get(WS_T_ACNT, (size_t)B * 4);
  uint64_t* S = (uint64_t*)w.get(WS_T_B, (size_t)B * sparse_limit * 8);
  int* Sc = (int*)w.get(WS_T_BCNT, (size_t)B * 4);
  int* nfail = (int*)w.get(WS_NFAIL, 4);
  int* spsum = (int*)w.get(WS_SP_SUM, 8);
  HX_HIP(hipMemsetAsync(nfail, 0, 4, st));         // (an empty shard enqueues neither stage)
  HX_HIP(hipMemsetAsync(spsum, 0, 8, st));
  search_dense(h, qd, B, 0, dense_limit, D, Dc, st, 0, [&]() {
    sparse_enqueue(h, qip, qix, qv, B, sparse_limit, S, Sc, st);
    launch_concat(D, dense_limit, Dc, S, sparse_limit, Sc, B, keys_dev, st);
  }, true);
  h->sp_sum_pending = false;                       // nobody will call sparse_resolve for this batch
  h->sp_sum_fetched = false;
  remap_out(h, keys_dev, (int64_t)B * L2, st);
  launch_flag_row(nfail, spsum, keys_dev + (size_t)B * L2, L2, st);
  HX_CATCH
}

int hx_h1_fuse(int32_t device, const uint64_t* gathered, int32_t world, int32_t B, int32_t dense_limit,
               int32_t sparse_limit, int32_t limit, float rrf_k, int32_t rank_base, uint64_t* keys_dev,
               int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(gathered && keys_dev && counts_dev && B > 0 && world >= 1, "bad argument");
  HX_CHECK(dense_limit >= 1 && sparse_limit >= 1 && (int64_t)world * std::max(dense_limit, sparse_limit) <= CAND_CAP,
           "h1_fuse: world x limit out of range [1, 8192]");
  HX_CHECK(limit >= 1 && limit <= MAX_LIMIT, "limit out of range [1, 2048]");
  HX_HIP(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = static_ws(device);
  const int ds = world * dense_limit, ss = world * sparse_limit;
  uint64_t* dall = (uint64_t*)w.get(WS_F_DALL, (size_t)B * ds * 8);
  uint64_t* sall = (uint64_t*)w.get(WS_F_SALL, (size_t)B * ss * 8);
  uint64_t* D = (uint64_t*)w.get(WS_F_D, (size_t)B * dense_limit * 8);
  uint64_t* S = (uint64_t*)w.get(WS_F_S, (size_t)B * sparse_limit * 8);
  int* Dc = (int*)w.get(WS_F_DC, (size_t)B * 4);
  int* Sc = (int*)w.get(WS_F_SC, (size_t)B * 4);
  launch_regroup(gathered, world, B, dense_limit, sparse_limit, dall, sall, st);
  launch_compact(dall, ds, nullptr, B, dense_limit, 0, D, dense_limit, Dc, nullptr, ds, st);
  launch_compact(sall, ss, nullptr, B, sparse_limit, 0, S, sparse_limit, Sc, nullptr, ss, st);
  rrf(nullptr, D, dense_limit, Dc, S, sparse_limit, Sc, B, rrf_k, rank_base, limit, keys_dev, counts_dev, st, w);
  HX_CATCH
}

// ---- row-sharded H1, candidates first (hx.h; shardx.hip) -----------------------------------------------------------------
// Workspace slots of the rescore step: it runs on the exchange stream BESIDE the next batch's nominate step on the same
// index, so it shares no buffer with any other entry.
enum { WSX = 3000 };

int hx_h1_plan(int32_t dense_limit, int32_t sparse_limit, int32_t world, int32_t* k1, int32_t* k2, int32_t* lp,
               int32_t* k3, int32_t* lout) {
  HX_TRY
  HX_CHECK(k1 && k2 && lp && k3 && lout, "NULL argument");
  HX_CHECK(dense_limit >= 1 && dense_limit <= MAX_LIMIT && sparse_limit >= 1 && sparse_limit <= MAX_LIMIT, "limit out of range [1, 2048]");
  HX_CHECK(world >= 1 && world <= 64, "world out of range [1, 64]");
  // a shard's share of a global list of n is Binomial(n, 1/world) on exchangeable rows: mean + 10 sigma, to a
  // multiple of 32; a topically clustered collection trips the completeness checks instead and the caller widens the
  // lists (distributed.H1Pipeline doubles them after a redone batch)
  auto share = [&](int n) {
    const double p = 1.0 / world, mean = n * p, sd = std::sqrt(n * p * (1.0 - p));
    return (int)std::min<int64_t>(round_up((int64_t)std::ceil(mean + 10.0 * sd), 32), round_up(n, 32));
  };
  const int cap = CAND_CAP / world / 32 * 32;       // world x k keys are merged in one 8192-key buffer
  HX_CHECK(cap >= 32, "world too large for the candidates-first exchange");
  const int Lp = cand8_lprime(dense_limit);
  HX_CHECK(Lp <= MAX_LIMIT, "dense_limit too large for the candidates-first exchange");
  *lp = Lp;
  *k1 = std::min(share(Lp), cap);
  *k2 = std::min(share(sparse_limit), cap);         // integer scores: enough to fix the value of the global L-th
  *k3 = std::min(share(sparse_limit), cap);         // exact sparse keys a shard returns
  *lout = sparse_lout(sparse_limit);
  HX_CATCH
}

int hx_sparse_wmax(hx_index* h, float* wmax, int32_t* nonpos) {
  HX_TRY
  HX_CHECK(h && wmax && nonpos, "NULL argument");
  *wmax = h->sp_have_w ? h->sp_wmax : 0.0f;
  *nonpos = (h->sp_have_w && !(h->sp_wmin > 0.0f)) ? 1 : 0;
  HX_CATCH
}

int hx_set_sparse_wmax(hx_index* h, float wmax) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  HX_CHECK(wmax >= 0.0f && wmax <= SPARSE_ABS_MAX, "wmax out of range");
  h->sp_wmax_shared = wmax;
  HX_CATCH
}

int hx_h1_nominate_async(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix, const float* qv, int32_t B,
                         int32_t dense_limit, int32_t sparse_limit, int32_t k1, int32_t k2, uint64_t* nom_dev,
                         void* stream) {
  HX_TRY
  HX_CHECK(h && qd && qip && nom_dev && B > 0, "bad argument");
  HX_CHECK(dense_limit >= 1 && dense_limit <= MAX_LIMIT && sparse_limit >= 1 && sparse_limit <= MAX_LIMIT, "limit out of range [1, 2048]");
  const int lout = sparse_lout(sparse_limit);
  HX_CHECK(k1 >= 1 && k1 <= CAND_CAP / 4 && k2 >= 1 && k2 <= lout, "k1 / k2 out of range");
  h->set_device();
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = h->ws;
  // ---- sparse: the shard's integer-score list (the select pass only; ids stay internal: only the scores travel, the
  // list itself is consumed by this rank's own rescore step) -- on the second stream, beside the dense scan
  bool forked = false;
  hipStream_t sst = st;
  // Only for LARGE shards: alone the call is shorter with it at every size (1.34 -> 1.27 ms at 1.25M rows, B = 1024), but in
  // H1Pipeline the re-score + exchange of the previous batch already run beside it on the pipeline's side stream, and a third
  // stream in the mix made the pipelined step LONGER at the 8- and 4-GPU shard sizes (1.49 -> 1.55 ms at 1.25M rows, 2.68 ->
  // 2.75 at 2.5M) and shorter only at the 2-GPU one (5.00 -> 4.93 ms at 5M rows; profiles/r04_h1_small_batch.txt)
  const char* nom_e = getenv("HX_DEBUG_NOM_FORK");          // (read per call: the tests switch it)
  const int nom_env = nom_e ? atoi(nom_e) : -1;
  const bool nom_fork = nom_env >= 0 ? nom_env != 0 : h->n >= 4000000;
  if (h->overlap_tail && nom_fork && h->n > 0) {
    ensure_side_stream(h);
    HX_HIP(hipEventRecord(h->ev_fork, st));
    HX_HIP(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
    sst = h->st2;
    forked = true;
  }
  struct Beside {
    hx_index* h;
    ~Beside() { h->beside = false; }
  } beside_guard{h};
  h->beside = forked;
  const SparseLists sl = sparse_select_lists(h, qip, qix, qv, B, sparse_limit, sst);
  h->sp_sum_pending = false;                         // nobody will call sparse_resolve for this batch
  h->sp_sum_fetched = false;
  if (forked) HX_HIP(hipEventRecord(h->ev_join, h->st2));
  // ---- dense: the shard's best k1 rows by the int8 candidate score (no exact score here)
  uint64_t* cand = nullptr;
  int *cnt = nullptr, *ovf = nullptr;
  float* eq = nullptr;
  int cstride = 0;
  if (h->n > 0) {
    HX_CHECK(h->cand8 && !h->cand8_off && h->q8s, "the candidates-first exchange needs the int8 candidate copy");
    const MatrixRef m = pick_matrix(h, 0);
    const int bn = scan_bn(h, B);
    const int Bpad = (int)round_up(B, bn);
    float* qn = (float*)w.get(WS_QN, (size_t)B * m.dpad * 4);
    launch_prep_queries_f(qd, h->dim, B, B, m.d, m.dpad, qn, nullptr, st);
    const Geometry g = geometry(dense_limit, true, false, true, k1);
    cand = (uint64_t*)w.get(WS_CAND, (size_t)B * g.C * 8);
    cnt = (int*)w.get(WS_CNT, (size_t)B * 4);
    ovf = (int*)w.get(WS_OVF, (size_t)B * 4);
    float* tau = (float*)w.get(WS_TAU, (size_t)B * 4);
    int8_t* q8 = (int8_t*)w.get(WS_Q8S, (size_t)Bpad * h->dim_pad8);
    float* sq = (float*)w.get(WS_SQ, (size_t)Bpad * 4);
    eq = (float*)w.get(WS_EPSQ, (size_t)B * 4);
    launch_prep_queries_s8(qn, m.dpad, B, Bpad, h->dim_pad8, q8, sq, eq, h->s8_err, st);
    chunked_scan(h, KIND_I8, (const uint8_t*)h->q8s, (const uint8_t*)q8, h->dim_pad8, B, bn, g, cand, cnt, ovf, tau, sq, st,
                 h->q8s_scale, &h->tm_q8s, 3);
    cstride = g.C;
    h->cand8_queries += B;
    remap_out(h, cand, (int64_t)B * g.C, st);        // (identity for a shard filled in one block: skipped)
  }
  h->beside = false;
  if (forked) HX_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
  HX_CHECK(!sl.list || sl.lout == lout, "sparse list stride");
  launch_h1x_pack(cand, cstride, cnt, ovf, eq, h->n <= k1 ? 1 : 0, k1, sl.list, sl.lout, sl.lcnt, sl.flag, sl.fail, k2,
                  lout, std::max(h->sp_wmax, h->sp_wmax_shared), B, nom_dev, st);
  HX_CATCH
}

int hx_h1_rescore_async(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix, const float* qv, int32_t B,
                        const uint64_t* nom_dev, const uint64_t* gathered_dev, int32_t world, int32_t rank,
                        int32_t dense_limit, int32_t sparse_limit, int32_t k1, int32_t k2, int32_t lp, int32_t k3,
                        uint64_t* res_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && qd && qip && nom_dev && gathered_dev && res_dev && B > 0 && world >= 1 && rank >= 0 && rank < world, "bad argument");
  HX_CHECK(dense_limit >= 1 && dense_limit <= lp && lp <= MAX_LIMIT && sparse_limit >= 1 && sparse_limit <= MAX_LIMIT, "limits out of range");
  HX_CHECK((int64_t)world * k1 <= CAND_CAP && (int64_t)world * k2 <= CAND_CAP && (int64_t)world * k3 <= CAND_CAP && k1 >= 1 &&
               k2 >= 1 && k3 >= 1 && k3 <= 256,
           "world x k out of range");
  h->set_device();
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = h->ws;
  const int lout = sparse_lout(sparse_limit);
  const int du_s = world * k1, su_s = world * k2, L_s = sparse_limit;
  uint64_t* du = (uint64_t*)w.get(WSX + 1, (size_t)B * du_s * 8);
  uint64_t* su = (uint64_t*)w.get(WSX + 2, (size_t)B * su_s * 8);
  uint64_t* G = (uint64_t*)w.get(WSX + 3, (size_t)B * lp * 8);
  int* gc = (int*)w.get(WSX + 4, (size_t)B * 4);
  uint64_t* ST = (uint64_t*)w.get(WSX + 5, (size_t)B * L_s * 8);
  int* sc = (int*)w.get(WSX + 6, (size_t)B * 4);
  int* margin = (int*)w.get(WSX + 7, (size_t)B * 4);
  int* qflag = (int*)w.get(WSX + 8, (size_t)B * 4);
  int* ncand = (int*)w.get(WSX + 9, (size_t)B * 4);
  int* sfail = (int*)w.get(WSX + 10, (size_t)B * 4);
  uint32_t* thr = (uint32_t*)w.get(WSX + 14, (size_t)B * 4);
  int* pcnt = (int*)w.get(WSX + 15, (size_t)B * 4);
  uint64_t* ex = (uint64_t*)w.get(WSX + 16, (size_t)B * lout * 8);
  uint64_t* T = (uint64_t*)w.get(WSX + 17, (size_t)B * k3 * 8);
  int* tc = (int*)w.get(WSX + 18, (size_t)B * 4);
  const size_t V = (size_t)lp + (size_t)world * k3 + world + 4;
  uint64_t* de = res_dev;                                          // [B x lp]
  uint64_t* se = res_dev + (size_t)B * lp;                         // [B x world x k3]
  uint64_t* nc = se + (size_t)B * world * k3;                      // [B x world]
  uint64_t* meta = nc + (size_t)B * world;                         // [B x 4]
  HX_HIP(hipMemsetAsync(res_dev, 0, (size_t)B * V * 8, st));
  HX_HIP(hipMemsetAsync(sfail, 0, (size_t)B * 4, st));
  HX_HIP(hipMemsetAsync(ncand, 0, (size_t)B * 4, st));
  HX_HIP(hipMemsetAsync(tc, 0, (size_t)B * 4, st));
  launch_h1x_union(gathered_dev, world, B, k1, k2, du, su, st);
  launch_compact(du, du_s, nullptr, B, std::min(lp, du_s), 0, G, lp, gc, nullptr, du_s, st);
  launch_compact(su, su_s, nullptr, B, std::min(L_s, su_s), 0, ST, L_s, sc, nullptr, su_s, st);
  launch_h1x_cuts(gathered_dev, world, B, k1, k2, G, gc, lp, ST, sc, L_s, qip, meta, thr, margin, qflag, st);
  // ---- this rank's rows among the global dense candidates: exact spec_dot at their positions in G
  if (h->n > 0) {
    const MatrixRef m = pick_matrix(h, 0);
    float* qn = (float*)w.get(WSX + 11, (size_t)B * m.dpad * 4);
    launch_prep_queries_f(qd, h->dim, B, B, m.d, m.dpad, qn, nullptr, st);
    RescoreArgs r{};
    r.kind = KIND_F32;
    r.M = m.m32;
    r.row_stride = m.dpad;
    r.dim_pad = m.dpad;
    r.Q = qn;
    r.q_stride = m.dpad;
    r.n_rows = h->n;
    r.id_base = h->id_base;
    r.cand = remap_in(h, G, (int64_t)B * lp, st, WSX + 12);   // rows of other shards become empty slots
    r.cnt = gc;
    r.stride = lp;
    r.max_cnt = lp;
    r.B = B;
    r.out = de;
    launch_rescore_own(r, st);                         // (most slots of G belong to other shards)
    remap_out(h, de, (int64_t)B * lp, st);
  }
  // ---- this rank's documents at or above the GLOBAL threshold, from its own list of the nominate step: exact
  // upstream-order scores, its best k3 into its slot
  if (h->sp_rows > 0 && h->nnz > 0) {
    const uint64_t* plist = nom_dev + (size_t)B * (k1 + k2 + 2);
    launch_h1x_counts(plist + (size_t)B * lout, B, pcnt, st);
    SparseRescoreArgs ra{};
    ra.d = csr_of(h);
    ra.q_indptr = qip;
    ra.q_idx = qix;
    ra.q_val = qv;
    ra.cand = plist;
    ra.cnt = pcnt;
    ra.stride = lout;
    ra.B = B;
    ra.limit = sparse_limit;
    ra.q_margin = margin;
    ra.q_flag = qflag;
    ra.out = ex;
    ra.out_cnt = ncand;
    ra.q_fail = sfail;
    ra.thr_in = thr;
    ra.blocks = 4;            // ~ (L + 10) / world candidates per query here, not L + 10
    launch_sparse_rescore(ra, st);
    launch_compact(ex, lout, ncand, B, std::min(k3, lout), 0, T, k3, tc, nullptr, lout, st);
    remap_out(h, T, (int64_t)B * k3, st);
  }
  launch_h1x_place(T, tc, ncand, sfail, B, k3, world, rank, se, nc, meta, st);
  HX_CATCH
}

int hx_h1_finish(int32_t device, const uint64_t* reduced_dev, int32_t world, int32_t B, int32_t lp, int32_t k3,
                 int32_t dense_limit, int32_t sparse_limit, int32_t limit, float rrf_k, int32_t rank_base,
                 uint64_t* keys_dev, int32_t* counts_dev, int32_t* nfail_dev, void* stream) {
  HX_TRY
  HX_CHECK(reduced_dev && keys_dev && counts_dev && nfail_dev && B > 0 && world >= 1, "bad argument");
  HX_CHECK(dense_limit >= 1 && dense_limit <= lp && lp <= MAX_LIMIT && sparse_limit >= 1 && sparse_limit <= MAX_LIMIT && k3 >= 1 &&
               (int64_t)world * k3 <= CAND_CAP,
           "limits out of range");
  HX_CHECK(limit >= 1 && limit <= MAX_LIMIT, "limit out of range [1, 2048]");
  HX_HIP(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  Workspace& w = static_ws(device);
  uint64_t* D = (uint64_t*)w.get(WS_F_D, (size_t)B * dense_limit * 8);
  uint64_t* S = (uint64_t*)w.get(WS_F_S, (size_t)B * sparse_limit * 8);
  int* Dc = (int*)w.get(WS_F_DC, (size_t)B * 4);
  int* Sc = (int*)w.get(WS_F_SC, (size_t)B * 4);
  int* fail = (int*)w.get(WS_FAIL, (size_t)B * 4);
  uint64_t* de = const_cast<uint64_t*>(reduced_dev);
  uint64_t* se = de + (size_t)B * lp;
  const int ss = world * k3;
  launch_compact(de, lp, nullptr, B, dense_limit, 0, D, dense_limit, Dc, nullptr, lp, st);
  launch_compact(se, ss, nullptr, B, std::min(sparse_limit, ss), 0, S, sparse_limit, Sc, nullptr, ss, st);
  launch_h1x_certify(reduced_dev, world, B, lp, k3, D, Dc, dense_limit, S, Sc, sparse_limit, fail, nfail_dev, st);
  rrf(nullptr, D, dense_limit, Dc, S, sparse_limit, Sc, B, rrf_k, rank_base, limit, keys_dev, counts_dev, st, w);
  HX_CATCH
}

int hx_unpack(int32_t device, const uint64_t* keys_dev, int64_t n, float* scores_dev, int64_t* ids_dev,
              void* stream) {
  HX_TRY
  HX_CHECK(keys_dev && scores_dev && ids_dev, "NULL argument");
  HX_HIP(hipSetDevice(device));
  launch_unpack(keys_dev, n, scores_dev, ids_dev, (hipStream_t)stream);
  HX_CATCH
}

int hx_hybrid_query_dev(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix,
                        const float* qv, int32_t B, const hx_params* p,
                        uint64_t* keys_dev, int32_t* counts_dev, void* stream) {
  HX_TRY
  HX_CHECK(h && qd && qip && keys_dev && counts_dev, "NULL argument");
  check_params(p);
  h->set_device();
  hybrid_query_dev(h, qd, qip, qix, qv, B, p, keys_dev, counts_dev, (hipStream_t)stream);
  remap_out(h, keys_dev, (int64_t)B * p->final_limit, (hipStream_t)stream);
  HX_CATCH
}

int hx_hybrid_query_host(hx_index* h, const float* qd, const int64_t* qip, const int32_t* qix,
                         const float* qv, int32_t B, const hx_params* p, float* scores, int64_t* ids,
                         int32_t* counts) {
  HX_TRY
  HX_CHECK(h && qd && qip && scores && ids && counts && B > 0, "bad argument");
  check_params(p);
  h->set_device();
  hipStream_t st = nullptr;
  const int64_t nnz = qip[B];
  HX_CHECK(qip[0] == 0 && nnz >= 0, "bad query indptr");
  // sort each query's terms by id (the spec's summation order), reject duplicates
  std::vector<int32_t> six((size_t)nnz);
  std::vector<float> sv((size_t)nnz);
  std::vector<int> ord;
  for (int b = 0; b < B; ++b) {
    const int64_t s = qip[b], e = qip[b + 1];
    HX_CHECK(e >= s && e <= nnz, "bad query indptr");
    ord.resize((size_t)(e - s));
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return qix[s + x] < qix[s + y]; });
    for (size_t i = 0; i < ord.size(); ++i) {
      six[(size_t)s + i] = qix[s + ord[i]];
      sv[(size_t)s + i] = qv[s + ord[i]];
      HX_CHECK(six[(size_t)s + i] >= 0, "sparse index out of range");
      HX_CHECK(i == 0 || six[(size_t)s + i] != six[(size_t)s + i - 1], "duplicate sparse index in query");
    }
  }
  const int L = p->final_limit;
  float* dq = (float*)h->ws.get(WS_H_QD, (size_t)B * h->dim * 4);
  int64_t* dip = (int64_t*)h->ws.get(WS_H_QIP, (size_t)(B + 1) * 8);
  int32_t* dix = (int32_t*)h->ws.get(WS_H_QIX, (size_t)std::max<int64_t>(nnz, 1) * 4);
  float* dv = (float*)h->ws.get(WS_H_QV, (size_t)std::max<int64_t>(nnz, 1) * 4);
  uint64_t* ok = (uint64_t*)h->ws.get(WS_H_OUT, (size_t)B * L * 8);
  int* oc = (int*)h->ws.get(WS_H_OCNT, (size_t)B * 4);
  float* osc = (float*)h->ws.get(WS_H_SC, (size_t)B * L * 4);
  int64_t* oid = (int64_t*)h->ws.get(WS_H_ID, (size_t)B * L * 8);
  HX_HIP(hipMemcpyAsync(dq, qd, (size_t)B * h->dim * 4, hipMemcpyHostToDevice, st));
  HX_HIP(hipMemcpyAsync(dip, qip, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, st));
  if (nnz) {
    HX_HIP(hipMemcpyAsync(dix, six.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    HX_HIP(hipMemcpyAsync(dv, sv.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, st));
  }
  hybrid_query_dev(h, dq, dip, dix, dv, B, p, ok, oc, st);
  remap_out(h, ok, (int64_t)B * L, st);
  launch_unpack(ok, (int64_t)B * L, osc, oid, st);
  HX_HIP(hipMemcpyAsync(scores, osc, (size_t)B * L * 4, hipMemcpyDeviceToHost, st));
  HX_HIP(hipMemcpyAsync(ids, oid, (size_t)B * L * 8, hipMemcpyDeviceToHost, st));
  HX_HIP(hipMemcpyAsync(counts, oc, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  HX_CATCH
}

int hx_get_stats(hx_index* h, hx_stats* out) {
  HX_TRY
  HX_CHECK(h && out, "NULL argument");
  std::memset(out, 0, sizeof(*out));
  out->n_rows = h->n;
  out->nnz = h->nnz;
  out->n_segments = h->sp_base.n_segments + h->sp_tail.n_segments;
  out->n_groups = h->sp_base.sp.n_live + h->sp_tail.sp.n_live;                  // live terms
  out->hash_capacity = h->sp_base.sp.ptr_entries + h->sp_tail.sp.ptr_entries;   // entries of the term x segment offset tables
  out->bytes_dense_f32 = h->n * h->dim_pad * 4;
  out->bytes_dense_f16 = h->n * h->dim_pad * 2;
  out->bytes_i8 = h->n * h->dim_pad8;
  int64_t bp = 0;
  for (int p = 0; p < h->n_pre; ++p) bp += h->n * h->psize[p] * 4;
  if (h->n_pre) bp += h->n * h->psize[0] * 2;
  out->bytes_prefix = bp;
  out->bytes_sparse = h->nnz * 8 + out->hash_capacity * 4 + out->n_groups * 4;
  out->dense_fallback_queries = h->dense_fallbacks;
  out->i8_fallback_queries = h->i8_fallbacks;
  out->retry_queries = h->retries;
  out->sparse_fallback_queries = h->sparse_fallbacks;
  out->bytes_i8_cand = h->q8s ? h->n * h->dim_pad8 + h->n * 4 : 0;
  out->cand8_queries = h->cand8_queries;
  out->cand8_uncertified_queries = h->cand8_failed;
  float emax = 0.f;
  if (h->s8_err) {
    h->set_device();
    HX_HIP(hipMemcpy(&emax, h->s8_err, 4, hipMemcpyDeviceToHost));
  }
  out->cand8_row_error_max = (double)emax;
  out->tree_batches_redone = h->tree_redone;
  out->cand8_switched_off = h->cand8_auto_off ? 1 : 0;
  out->tree_deferral_switched_off = h->tree_spec_off ? 1 : 0;
  HX_CATCH
}

int hx_set_stream_overlap(hx_index* h, int32_t on) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  h->overlap_tail = on != 0;
  HX_CATCH
}

int hx_set_dense_candidates(hx_index* h, int32_t kind) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  HX_CHECK(kind == 0 || kind == 1, "kind: 0 = fp16 candidates, 1 = int8 candidates");
  HX_CHECK(kind == 0 || h->q8s || h->n == 0, "this index holds no int8 candidate copy (created with HX_DENSE_CAND=f16)");
  if (kind == 1 && !h->q8s) {                 // empty index: the copy is made as rows arrive
    // ... into buffers that must exist for the capacity the index already has (hx_reserve, or hx_truncate(h, 0) on an
    // index created under HX_DENSE_CAND=f16): reserve_rows returns early while want <= cap and would never make them
    h->set_device();
    if (h->cap > 0) {
      grow_copy(h->q8s, 0, h->cap * h->dim_pad8, false);
      grow_copy(h->q8s_scale, 0, h->cap + 256, true);
    }
    if (!h->s8_err) {
      HX_HIP(hipMalloc((void**)&h->s8_err, 4));
      HX_HIP(hipMemset(h->s8_err, 0, 4));
    }
    h->tm_q8s.rows = -1;
    h->cand8 = 1;
  }
  h->cand8_off = kind == 0;
  if (kind == 1) {                  // an explicit request also resets the guard
    h->cand8_auto_off = false;
    h->c8_win_q = h->c8_win_f = 0;
  }
  HX_CATCH
}

int hx_profile(hx_index* h, int32_t enable) {
  HX_TRY
  HX_CHECK(h, "index is NULL");
  h->prof = enable != 0;
  HX_CATCH
}

int hx_profile_read(hx_index* h, hx_prof* out) {
  HX_TRY
  HX_CHECK(h && out, "NULL argument");
  h->set_device();
  std::memset(out, 0, sizeof(*out));
  for (auto& r : h->prof_recs) {
    HX_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    HX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    const int w = r.what;  // 0 f16 scan, 1 i8 scan ("quantized" stage), 2 sparse select, 3 int8 candidate scan of the dense stage, 4 K1/K2 ingest
    out->launches[w] += 1;
    out->ms[w] += ms;
    out->flops[w] += r.flops;
    out->bytes[w] += r.bytes;
    h->prof_pool.emplace_back(r.a, r.b);
  }
  h->prof_recs.clear();
  if (h->sp_counter) {
    unsigned long long np = 0;
    HX_HIP(hipMemcpy(&np, h->sp_counter, 8, hipMemcpyDeviceToHost));
    HX_HIP(hipMemset(h->sp_counter, 0, 8));
    out->bytes[2] = (double)np * 8.0;   // {doc in segment, weight}: 8 B per posting visited
  }
  HX_CATCH
}

int hx_debug_row(hx_index* h, int32_t which, int64_t row, void* out_host) {
  HX_TRY
  HX_CHECK(h && out_host, "NULL argument");
  HX_CHECK(row >= 0 && row < h->n, "row out of range");
  h->set_device();
  if (which == 0) {
    HX_HIP(hipMemcpy(out_host, h->dense + row * h->dim_pad, (size_t)h->dim * 4, hipMemcpyDeviceToHost));
  } else if (which >= 1 && which <= h->n_pre) {
    const int p = which - 1;
    HX_HIP(hipMemcpy(out_host, h->pre[p] + row * h->psize[p], (size_t)h->psize[p] * 4, hipMemcpyDeviceToHost));
  } else if (which == 4) {
    HX_HIP(hipMemcpy(out_host, h->q8 + row * h->dim_pad8, (size_t)h->dim, hipMemcpyDeviceToHost));
  } else if (which == 5 && h->q8s) {       // the candidate-pass copy of the row
    HX_HIP(hipMemcpy(out_host, h->q8s + row * h->dim_pad8, (size_t)h->dim, hipMemcpyDeviceToHost));
  } else if (which == 6 && h->q8s) {       // ... and its scale
    HX_HIP(hipMemcpy(out_host, h->q8s_scale + row, 4, hipMemcpyDeviceToHost));
  } else {
    throw Error("bad `which`");
  }
  HX_CATCH
}

// ---- persistence -------------------------------------------------
// The reference asks Qdrant for on-disk storage (qdrant_handler.py:47-55, 62: on_disk=True,
// memmap_threshold).  A collection is written as one file: header, then the device arrays as they
// stand (derived vectors are stored, not re-derived: loading reproduces the index bit for bit) and
// the doc-major sparse CSR; the inverted index is rebuilt on load (K9, ~1.4 s per 10^9 postings).
namespace {
struct HxFileHeader {
  char magic[8];             // "HXIDX\0\0\2" (version 1: no id-block table, n_blocks reads 0)
  int32_t dim, n_pre, psize[3], n_blocks;
  int64_t id_base, n, sp_rows, nnz;
};
const char HX_MAGIC[8] = {'H', 'X', 'I', 'D', 'X', 0, 0, 2};
constexpr size_t HX_IO_CHUNK = (size_t)64 << 20;

struct File {
  FILE* f;
  explicit File(const char* path, const char* mode) : f(fopen(path, mode)) {
    if (!f) throw Error(std::string("cannot open ") + path);
  }
  ~File() { if (f) fclose(f); }
};
void dev_to_file(FILE* f, const void* dev, size_t bytes, std::vector<char>& buf) {
  for (size_t o = 0; o < bytes; o += HX_IO_CHUNK) {
    const size_t m = std::min(HX_IO_CHUNK, bytes - o);
    HX_HIP(hipMemcpy(buf.data(), (const char*)dev + o, m, hipMemcpyDeviceToHost));
    HX_CHECK(fwrite(buf.data(), 1, m, f) == m, "short write");
  }
}
void file_to_dev(FILE* f, void* dev, size_t bytes, std::vector<char>& buf) {
  for (size_t o = 0; o < bytes; o += HX_IO_CHUNK) {
    const size_t m = std::min(HX_IO_CHUNK, bytes - o);
    HX_CHECK(fread(buf.data(), 1, m, f) == m, "short read: truncated index file");
    HX_HIP(hipMemcpy((char*)dev + o, buf.data(), m, hipMemcpyHostToDevice));
  }
}
}  // namespace

int hx_save(hx_index* h, const char* path) {
  HX_TRY
  HX_CHECK(h && path, "NULL argument");
  h->set_device();
  HX_HIP(hipDeviceSynchronize());
  File fl(path, "wb");
  HxFileHeader hd{};
  memcpy(hd.magic, HX_MAGIC, 8);
  hd.dim = h->dim;
  hd.n_pre = h->n_pre;
  for (int p = 0; p < 3; ++p) hd.psize[p] = h->psize[p];
  hd.id_base = h->id_base;
  hd.n = h->n;
  hd.sp_rows = h->sp_rows;
  hd.nnz = h->nnz;
  hd.n_blocks = (int32_t)h->ids.row0.size();
  HX_CHECK(fwrite(&hd, sizeof hd, 1, fl.f) == 1, "short write");
  if (hd.n_blocks) {         // the id-block table: row0 column, then gid0 column
    HX_CHECK(fwrite(h->ids.row0.data(), 4, (size_t)hd.n_blocks, fl.f) == (size_t)hd.n_blocks, "short write");
    HX_CHECK(fwrite(h->ids.gid0.data(), 4, (size_t)hd.n_blocks, fl.f) == (size_t)hd.n_blocks, "short write");
  }
  std::vector<char> buf(HX_IO_CHUNK);
  const size_t n = (size_t)h->n;
  dev_to_file(fl.f, h->dense, n * h->dim_pad * 4, buf);
  dev_to_file(fl.f, h->dense_h, n * h->dim_pad * 2, buf);
  dev_to_file(fl.f, h->q8, n * h->dim_pad8, buf);
  dev_to_file(fl.f, h->q8_rinv, n * 4, buf);
  for (int p = 0; p < h->n_pre; ++p) dev_to_file(fl.f, h->pre[p], n * h->psize[p] * 4, buf);
  if (h->n_pre > 0) dev_to_file(fl.f, h->pre_h0, n * h->psize[0] * 2, buf);
  if (h->sp_rows > 0) {
    dev_to_file(fl.f, h->sp_indptr, ((size_t)h->sp_rows + 1) * 8, buf);
    dev_to_file(fl.f, h->sp_idx, (size_t)h->nnz * 4, buf);
    dev_to_file(fl.f, h->sp_val, (size_t)h->nnz * 4, buf);
  }
  HX_CHECK(fflush(fl.f) == 0, "flush failed");
  HX_CATCH
}

int hx_load(const char* path, int32_t device, hx_index** out) {
  HX_TRY
  HX_CHECK(path && out, "NULL argument");
  File fl(path, "rb");
  HxFileHeader hd{};
  HX_CHECK(fread(&hd, sizeof hd, 1, fl.f) == 1, "short read: not an index file");
  HX_CHECK(memcmp(hd.magic, HX_MAGIC, 7) == 0 && (hd.magic[7] == 1 || hd.magic[7] == 2),
           "not an hx index file (bad magic / version)");
  if (hd.magic[7] == 1) hd.n_blocks = 0;     // version 1: ids are id_base + row
  // The header is not trusted: every size is checked against the limits of the add path and against the
  // length of the file BEFORE anything is allocated (a corrupt count must not become a huge hipMalloc).
  HX_CHECK(hd.n >= 0 && hd.sp_rows >= 0 && hd.nnz >= 0 && hd.n_pre >= 0 && hd.n_pre <= 3, "corrupt header");
  HX_CHECK(hd.dim >= 1 && hd.dim <= 4096, "corrupt header: dim");
  HX_CHECK(hd.id_base >= 0 && hd.id_base + hd.n < 0xFFFFFFFFll, "corrupt header: row ids");
  HX_CHECK(hd.nnz < 0xFFFFFFFFll && hd.sp_rows <= hd.n && (hd.sp_rows > 0 || hd.nnz == 0), "corrupt header: sparse counts");
  {
    const int64_t dp = round_up(hd.dim, 64), dp8 = round_up(hd.dim, 128);
    int64_t per_row = dp * 4 + dp * 2 + dp8 + 4;
    for (int p = 0; p < hd.n_pre; ++p) {
      HX_CHECK(hd.psize[p] >= 64 && hd.psize[p] <= hd.dim && hd.psize[p] % 64 == 0, "corrupt header: prefix sizes");
      per_row += (int64_t)hd.psize[p] * 4;
    }
    if (hd.n_pre > 0) per_row += (int64_t)hd.psize[0] * 2;
    HX_CHECK(hd.n_blocks >= 0 && hd.n_blocks <= hd.n, "corrupt header: id blocks");
    int64_t want = (int64_t)sizeof hd + (int64_t)hd.n_blocks * 8 + hd.n * per_row;
    if (hd.sp_rows > 0) want += (hd.sp_rows + 1) * 8 + hd.nnz * 8;
    HX_CHECK(fseek(fl.f, 0, SEEK_END) == 0, "cannot seek");
    const int64_t have = (int64_t)ftell(fl.f);
    HX_CHECK(fseek(fl.f, (long)sizeof hd, SEEK_SET) == 0, "cannot seek");
    HX_CHECK(have == want, "index file length does not match its header (truncated or corrupt)");
  }
  // the id-block table: both columns ascend strictly from row 0, a block's ids end before the next block's begin,
  // the last id stays below 2^32 - 1
  std::vector<uint32_t> row0((size_t)hd.n_blocks), gid0((size_t)hd.n_blocks);
  if (hd.n_blocks) {
    HX_CHECK(fread(row0.data(), 4, row0.size(), fl.f) == row0.size(), "short read: truncated index file");
    HX_CHECK(fread(gid0.data(), 4, gid0.size(), fl.f) == gid0.size(), "short read: truncated index file");
    HX_CHECK(row0[0] == 0, "corrupt index file: id blocks");
    for (size_t k = 0; k < row0.size(); ++k) {
      const int64_t len = (k + 1 < row0.size() ? (int64_t)row0[k + 1] : hd.n) - (int64_t)row0[k];
      HX_CHECK(len >= 1 && (int64_t)gid0[k] + len < 0xFFFFFFFFll, "corrupt index file: id blocks");
      HX_CHECK(k + 1 == row0.size() || (int64_t)gid0[k] + len <= (int64_t)gid0[k + 1], "corrupt index file: id blocks");
    }
  } else if (hd.n > 0) {
    row0.assign(1, 0u);
    gid0.assign(1, (uint32_t)hd.id_base);
  }
  hx_index* h = nullptr;
  const int rc = hx_create(hd.dim, hd.psize, hd.n_pre, device, hd.id_base, &h);
  if (rc != 0) return rc;
  try {
    h->set_device();
    if (!row0.empty()) {
      h->ids.end = (int64_t)gid0.back() + (hd.n - (int64_t)row0.back());
      h->ids.row0.swap(row0);
      h->ids.gid0.swap(gid0);
      h->ids.dirty = true;
    }
    std::vector<char> buf(HX_IO_CHUNK);
    const size_t n = (size_t)hd.n;
    reserve_rows(h, hd.n);
    file_to_dev(fl.f, h->dense, n * h->dim_pad * 4, buf);
    file_to_dev(fl.f, h->dense_h, n * h->dim_pad * 2, buf);
    file_to_dev(fl.f, h->q8, n * h->dim_pad8, buf);
    file_to_dev(fl.f, h->q8_rinv, n * 4, buf);
    for (int p = 0; p < h->n_pre; ++p) file_to_dev(fl.f, h->pre[p], n * h->psize[p] * 4, buf);
    if (h->n_pre > 0) file_to_dev(fl.f, h->pre_h0, n * h->psize[0] * 2, buf);
    h->n = hd.n;
    // the candidate copy is derived data (not in the file): one pass over the stored rows
    if (h->cand8) launch_requant_rows(h->dense, h->dim_pad, h->dim_pad8, hd.n, h->q8s, h->q8s_scale, h->s8_err, nullptr);
    if (hd.sp_rows > 0) {
      reserve_sparse(h, hd.sp_rows, hd.nnz);
      file_to_dev(fl.f, h->sp_indptr, ((size_t)hd.sp_rows + 1) * 8, buf);
      file_to_dev(fl.f, h->sp_idx, (size_t)hd.nnz * 4, buf);
      file_to_dev(fl.f, h->sp_val, (size_t)hd.nnz * 4, buf);
      // what the add path guarantees and the kernels rely on: monotone indptr from 0 to nnz, term ids in
      // [0, 2^31), finite values
      int* bad = (int*)h->ws.get(WS_MISC, 4);
      HX_HIP(hipMemset(bad, 0, 4));
      launch_csr_check(h->sp_indptr, h->sp_idx, hd.sp_rows, 0, hd.nnz, bad, nullptr);
      int hbad = 0;
      HX_HIP(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      HX_CHECK(hbad == 0, "corrupt index file: sparse CSR is inconsistent");
      // term ids unique within a row (hx_add_sparse enforces it; the exact pass finds a query term with ONE ballot)
      constexpr int LONG_CAP = 4096;
      int64_t* long_rows = (int64_t*)h->ws.get(WS_LONG_ROWS, (size_t)LONG_CAP * 8 + 8);
      int* n_long = (int*)(long_rows + LONG_CAP);
      HX_HIP(hipMemset(n_long, 0, 4));
      launch_csr_unique(h->sp_indptr, h->sp_idx, hd.sp_rows, bad, long_rows, LONG_CAP, n_long, nullptr);
      int hn_long = 0;
      HX_HIP(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      HX_HIP(hipMemcpy(&hn_long, n_long, 4, hipMemcpyDeviceToHost));
      HX_CHECK(hbad == 0, "corrupt index file: a sparse vector repeats a term id");
      if (hn_long > 0) {            // the rows too long for the wave compare: sorted on the host
        std::vector<int64_t> rows;
        if (hn_long <= LONG_CAP) {
          rows.resize((size_t)hn_long);
          HX_HIP(hipMemcpy(rows.data(), long_rows, rows.size() * 8, hipMemcpyDeviceToHost));
        } else {                    // more of them than the device list holds (hx_add_sparse accepts any number): find
                                    // them all from the offsets -- a valid file, not a corrupt one
          std::vector<int64_t> ip((size_t)hd.sp_rows + 1);
          HX_HIP(hipMemcpy(ip.data(), h->sp_indptr, ip.size() * 8, hipMemcpyDeviceToHost));
          for (int64_t r = 0; r < hd.sp_rows; ++r)
            if (ip[(size_t)r + 1] - ip[(size_t)r] > CSR_UNIQUE_WAVE_MAX) rows.push_back(r);
        }
        std::vector<int32_t> ids;
        for (int64_t r : rows) {
          int64_t be[2];
          HX_HIP(hipMemcpy(be, h->sp_indptr + r, 16, hipMemcpyDeviceToHost));
          ids.resize((size_t)(be[1] - be[0]));
          HX_HIP(hipMemcpy(ids.data(), h->sp_idx + be[0], ids.size() * 4, hipMemcpyDeviceToHost));
          std::sort(ids.begin(), ids.end());
          HX_CHECK(std::adjacent_find(ids.begin(), ids.end()) == ids.end(), "corrupt index file: a sparse vector repeats a term id");
        }
      }
      track_weights_dev(h, h->sp_val, hd.nnz, nullptr);
      h->sp_rows = hd.sp_rows;
      h->nnz = hd.nnz;
      h->sparse_stale = true;
    }
  } catch (...) {
    (void)hx_destroy(h);
    throw;
  }
  *out = h;
  HX_CATCH
}

}  // extern "C"
