// K7 -- sparse ("BM25") scoring over the on-device inverted index.
//
// Mirrors Prefetch(query=SparseVector, using="sparse", limit=sparse_limit)
// (app/core/vector_store/qdrant/qdrant_handler.py:347-354): score(d) = sum over the
// query's terms of q_t * d_t, IDF-free (the collection sets no sparse modifier, :80-86);
// only documents that share a term with the query are candidates.
//
// Arithmetic (= oracle.OracleIndex.sparse_scores, bit for bit): every product q_t*d_t is
// formed exactly in fp64, scaled by 2^40 and rounded (ties to even) to a 64-bit integer; the integers
// are summed (associative, so the order of the terms cannot matter) and the sum is
// converted once to fp32.  That restates upstream's fp32 running sum order-independently
// (it differs from it by at most a few fp32 ulps) and lets all postings of a segment be
// accumulated concurrently with LDS integer atomics.
//
// Index layout (spbuild.hip): documents are cut into segments of SEG_DOCS; inside a
// segment postings are sorted by (term, doc) as {u16 doc_local, f32 weight}; an
// open-addressing table maps (segment, term) -> (offset, length).
//
// One 512-thread workgroup owns (query, part): a contiguous range of segments, a 64 KiB
// LDS accumulator (one 64-bit word per document of the segment).  The
// fast path (<= SP_TMAX query terms) is software-pipelined across segments: while
// segment s is accumulated (LDS integer atomics) and harvested (a linear sweep of the
// accumulator), the postings of s+1 and the
// directory probes of s+2 are in flight (raw s_barrier + lgkmcnt waits only, so the
// vector-memory queue is never drained inside the loop), and LDS operations are issued
// in independent batches (no dependent chains per term).  Survivors (score >= the
// running threshold) are appended to a per-workgroup buffer in global memory that is
// sorted and cut to `limit`, through the (then all-zero) accumulator as LDS scratch,
// whenever it could overflow.
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

// Diagnostic build only (-DHX_SP_STAMP): wave 0 accumulates s_memtime deltas per phase
// into a debug buffer of its own (never read by the kernel, never in a timed build).
#ifdef HX_SP_STAMP
__device__ unsigned long long g_sp_stamps[8 * 4096];
#define SP_STAMP_DECL unsigned long long st_t0 = clock64(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SP_STAMP(i) { const unsigned long long st_t1 = clock64(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; }
#define SP_STAMP_FLUSH if (tid == 0 && blockIdx.x < 4096) for (int i_ = 0; i_ < 8; ++i_) g_sp_stamps[blockIdx.x * 8 + i_] = st_acc[i_];
#else
#define SP_STAMP_DECL
#define SP_STAMP(i)
#define SP_STAMP_FLUSH
#endif

constexpr int SP_THREADS = 512;
constexpr int SP_TCH = 64;           // generic path: query terms looked up per round
constexpr int SP_TMAX = 12;          // pipelined path: max query terms
constexpr int SP_K = 4;              // pipelined path: posting slots per thread (SP_K*512 per segment in registers)
constexpr double SP_FIX = 1099511627776.0;          // 2^40
constexpr float SP_UNFIX = 9.094947017729282e-13f;  // 2^-40 (exact in fp32)

__host__ __device__ inline uint64_t sp_hash(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// An accumulator word is sum(fx_t) + k * 2^52, k = number of postings added: non-zero
// exactly when the document was touched, whatever the signs of the products.  Decodes
// uniquely while |sum| < 2^51 (|score| < 2048) and k < 2048.
constexpr unsigned long long SP_MARK = 1ull << 52;
__device__ __forceinline__ unsigned long long sp_fix(float q, float w) {
  // exact product in fp64, scaled, rounded to the nearest integer (ties to even) by the
  // 1.5*2^52 trick: the low mantissa bits of y ARE the integer (|x| < 2^51); + touch marker
  const double c = 6755399441055744.0;   // 2^52 + 2^51
  const double y = ((double)q * (double)w) * SP_FIX + c;
  return (unsigned long long)(__double_as_longlong(y) - __double_as_longlong(c)) + SP_MARK;
}
__device__ __forceinline__ float sp_unfix(unsigned long long v) {
  const unsigned long long k = (v + (SP_MARK >> 1)) >> 52;
  return __fmul_rn((float)(long long)(v - (k << 52)), SP_UNFIX);
}

// LDS-only barrier: does not wait for outstanding global loads
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct SpShared {
  union {
    unsigned long long acc[SEG_DOCS];        // marked fixed-point score per document of the segment
    uint64_t sort[SEG_DOCS];                 // sort scratch while acc is all zero
  };
  uint32_t t_off[3][SP_TCH], t_len[3][SP_TCH];   // directory ring (pipelined path uses 3 slots)
  float t_w[SP_TCH];
  int cnt;
  float tau;
};

// Sort the workgroup's candidate buffer (global) through LDS, keep `limit`, raise tau.
// Precondition: acc is all zero and every wave is past its last acc access.
__device__ __forceinline__ void sp_sort_truncate(SpShared& S, uint64_t* cand, int limit, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends have left the core
  __syncthreads();                                   // ... and every other wave's; S.cnt settled
  const int n = S.cnt;
  for (int i = tid; i < SP_CAND; i += SP_THREADS)
    S.sort[i] = i < n ? __hip_atomic_load(cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  lds_barrier();
  for (int k = 2; k <= SP_CAND; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < SP_CAND; i += SP_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = S.sort[i], y = S.sort[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            S.sort[i] = y;
            S.sort[ixj] = x;
          }
        }
      }
      lds_barrier();
    }
  }
  const int keep = n < limit ? n : limit;
  for (int i = tid; i < keep; i += SP_THREADS)
    __hip_atomic_store(cand + i, S.sort[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0 && n >= limit) {
    S.tau = key_score(S.sort[limit - 1]);
    S.cnt = limit;
  }
  lds_barrier();
  for (int i = tid; i < SP_CAND; i += SP_THREADS) S.sort[i] = 0ull;   // acc back to zero
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ void sp_append(SpShared& S, uint64_t* cand, float tau, unsigned long long fx,
                                          int64_t gid) {
  const float s = sp_unfix(fx);
  if (s >= tau) {
    const int pos = atomicAdd(&S.cnt, 1);
    __hip_atomic_store(cand + pos, make_key(s, (uint32_t)gid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Harvest by a linear sweep of the accumulator: thread tid owns documents tid + 512*e, so a
// wave reads 512 contiguous bytes per instruction (conflict free).  e in [e0, e1).
template <int E0, int E1>
__device__ __forceinline__ void sp_harvest_sweep(SpShared& S, uint64_t* cand, float tau, int64_t gbase, int tid) {
  // pass 1: which of my entries are touched (reads issued back to back)
  uint32_t m = 0;
#pragma unroll
  for (int e = E0; e < E1; ++e) m |= (S.acc[tid + e * SP_THREADS] != 0ull ? 1u : 0u) << e;
  // pass 2: a wave loops max-popcount times (touched entries are sparse), not E times
  while (m) {
    const int e = __builtin_ctz(m);
    m &= m - 1;
    const int d = tid + e * SP_THREADS;
    const unsigned long long v = S.acc[d];
    S.acc[d] = 0ull;
    sp_append(S, cand, tau, v, gbase + d);
  }
}

// Harvest a whole segment.  Precondition (sp_make_room): cnt + SEG_DOCS/2 <= SP_CAND.
__device__ __forceinline__ void sp_harvest(SpShared& S, const SparseQueryArgs& a, uint64_t* cand,
                                           unsigned long long* park, int seg, uint32_t total_len, int tid) {
  constexpr int E = SEG_DOCS / SP_THREADS;
  const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
  const uint32_t bound = total_len < (uint32_t)SEG_DOCS ? total_len : (uint32_t)SEG_DOCS;
  if ((uint32_t)S.cnt + bound <= (uint32_t)SP_CAND) {   // block-uniform
    sp_harvest_sweep<0, E>(S, cand, S.tau, gbase, tid);
  } else {
    // a segment can yield up to SEG_DOCS survivors: harvest the lower half, cut the buffer
    // to `limit` (the upper half of the accumulator is parked in global memory while the
    // sort borrows the LDS), then harvest the upper half.
    constexpr int H = SEG_DOCS / 2;
    sp_harvest_sweep<0, E / 2>(S, cand, S.tau, gbase, tid);
    lds_barrier();
    for (int i = tid; i < H; i += SP_THREADS) {
      __hip_atomic_store(park + i, S.acc[H + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.acc[H + i] = 0ull;
    }
    lds_barrier();
    sp_sort_truncate(S, cand, a.limit, tid);
    for (int i = tid; i < H; i += SP_THREADS)
      S.acc[H + i] = __hip_atomic_load(park + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();
    sp_harvest_sweep<E / 2, E>(S, cand, S.tau, gbase, tid);
  }
  lds_barrier();
}

// before a segment is accumulated (acc all zero): make room for its survivors
__device__ __forceinline__ void sp_make_room(SpShared& S, const SparseQueryArgs& a, uint64_t* cand, uint32_t total_len,
                                             int tid) {
  const uint32_t bound = total_len < (uint32_t)SEG_DOCS ? total_len : (uint32_t)SEG_DOCS;
  if ((uint32_t)S.cnt + bound > (uint32_t)SP_CAND && S.cnt > a.limit) sp_sort_truncate(S, cand, a.limit, tid);
}

__device__ __forceinline__ void sp_finish(SpShared& S, const SparseQueryArgs& a, uint64_t* cand, int q, int part,
                                          int tid) {
  sp_sort_truncate(S, cand, a.limit, tid);
  const int n = S.cnt < a.limit ? S.cnt : a.limit;
  uint64_t* o = a.out + ((int64_t)q * a.parts + part) * a.limit;
  for (int i = tid; i < a.limit; i += SP_THREADS)
    o[i] = i < n ? __hip_atomic_load(cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  if (tid == 0) a.out_cnt[q * a.parts + part] = n;
}

// ---------------------------------------------------------------------------------
// directory probes
// ---------------------------------------------------------------------------------
struct Probe {
  SpHashEntry e0, e1;   // the two table slots fetched speculatively
};
__device__ __forceinline__ Probe sp_probe_issue(const SparseQueryArgs& a, int seg, uint32_t term) {
  Probe p;
  const uint64_t key = ((uint64_t)seg << 31) | term;
  const uint64_t slot = sp_hash(key) & a.ix.table_mask;
  p.e0 = a.ix.table[slot];
  p.e1 = a.ix.table[(slot + 1) & a.ix.table_mask];
  return p;
}
__device__ __forceinline__ void sp_probe_resolve(const SparseQueryArgs& a, const Probe& p, int seg, uint32_t term,
                                                 uint32_t& off, uint32_t& len) {
  const uint64_t key = ((uint64_t)seg << 31) | term;
  off = 0;
  len = 0;
  if (p.e0.key == key) {
    off = p.e0.off;
    len = p.e0.len;
  } else if (p.e0.key == ~0ull) {
  } else if (p.e1.key == key) {
    off = p.e1.off;
    len = p.e1.len;
  } else if (p.e1.key == ~0ull) {
  } else {
    uint64_t slot = ((sp_hash(key) & a.ix.table_mask) + 2) & a.ix.table_mask;   // rare: longer chain
    while (true) {
      const SpHashEntry e = a.ix.table[slot];
      if (e.key == key) {
        off = e.off;
        len = e.len;
        break;
      }
      if (e.key == ~0ull) break;
      slot = (slot + 1) & a.ix.table_mask;
    }
  }
}

// ---------------------------------------------------------------------------------
// pipelined path: T <= SP_TMAX
// ---------------------------------------------------------------------------------
// The postings a segment holds for the query are the concatenation of its <= SP_TMAX
// runs; thread tid takes flat positions tid, tid + 512, ... (SP_K register slots), so all
// waves carry the same load whatever the run lengths.  A slot keeps {doc, weight, q_t}.
struct SpDir {
  uint32_t len[SP_TMAX], off[SP_TMAX];
  float tw[SP_TMAX];
  uint32_t total;
};
__device__ __forceinline__ SpDir sp_dir_read(const SpShared& S, int slot) {
  SpDir D;
  D.total = 0;
#pragma unroll
  for (int t = 0; t < SP_TMAX; ++t) {
    D.len[t] = S.t_len[slot][t];
    D.off[t] = S.t_off[slot][t];
    D.tw[t] = S.t_w[t];
    D.total += D.len[t];
  }
  return D;
}
// flat position f (< D.total) -> posting index and query weight, by a select chain
__device__ __forceinline__ void sp_map(const SpDir& D, uint32_t f, uint32_t& idx, float& qw) {
  uint32_t o = D.off[0], p = 0, run = D.len[0];
  qw = D.tw[0];
#pragma unroll
  for (int t = 1; t < SP_TMAX; ++t) {
    const bool c = f >= run;        // run = start of run t
    o = c ? D.off[t] : o;
    p = c ? run : p;
    qw = c ? D.tw[t] : qw;
    run += D.len[t];
  }
  idx = o + (f - p);
}

__device__ __forceinline__ void sp_body_pipe(SpShared& S, const SparseQueryArgs& a, uint64_t* cand,
                                             unsigned long long* park, int q, int part, int s0, int s1, int64_t qb,
                                             int T, int tid) {
  uint32_t my_term = 0;
  if (tid < T) my_term = (uint32_t)a.q_idx[qb + tid];
  // ---- prologue: directory of s0 (resolved), probes of s0+1 (in flight), postings of s0
  Probe pr{};
  if (tid < SP_TMAX) {
    uint32_t off = 0, len = 0;
    float w = 0.f;
    if (tid < T) {
      w = a.q_val[qb + tid];
      Probe p0 = sp_probe_issue(a, s0, my_term);
      sp_probe_resolve(a, p0, s0, my_term, off, len);
      if (s0 + 1 < s1) pr = sp_probe_issue(a, s0 + 1, my_term);
    }
    S.t_w[tid] = w;
    S.t_off[0][tid] = off;     // terms >= T are empty runs: the loops below need no t < T test
    S.t_len[0][tid] = len;
    S.t_off[1][tid] = S.t_off[2][tid] = 0;
    S.t_len[1][tid] = S.t_len[2][tid] = 0;
  }
  lds_barrier();
  uint32_t pd[SP_K];      // doc_local, 0xFFFF = empty slot
  float pw[SP_K], pq[SP_K];
  uint32_t total;
  {
    const SpDir D = sp_dir_read(S, 0);
    total = D.total;
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      const uint32_t f = tid + k * SP_THREADS;
      const bool ok = f < D.total;
      uint32_t i;
      sp_map(D, ok ? f : 0, i, pq[k]);
      i = D.total ? i : 0;
      const uint32_t d = a.ix.doc_local[i];
      pw[k] = a.ix.w[i];
      pd[k] = ok ? d : 0xFFFFu;
    }
  }
  unsigned long long npost = 0;
  int cur = 0;
  SP_STAMP_DECL
  for (int seg = s0; seg < s1; ++seg) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    // (1) directory of seg+1 into ring slot `nxt`: resolve the probes issued one iteration
    //     ago.  Slot nxt was last read in iteration seg-2, which every wave left before it
    //     passed the barrier below in iteration seg-1.
    if (tid < T) {
      uint32_t off = 0, len = 0;
      if (seg + 1 < s1) {
        sp_probe_resolve(a, pr, seg + 1, my_term, off, len);
        if (seg + 2 < s1) pr = sp_probe_issue(a, seg + 2, my_term);   // (2) probes of seg+2
      }
      S.t_off[nxt][tid] = off;
      S.t_len[nxt][tid] = len;
    }
    lds_barrier();
    SP_STAMP(0)
    // (3) postings of seg+1 -> registers (in flight while seg is accumulated and harvested)
    uint32_t nd[SP_K];
    float nw[SP_K], nq[SP_K];
    uint32_t ntotal;
    {
      const SpDir D = sp_dir_read(S, nxt);
      ntotal = D.total;
#pragma unroll
      for (int k = 0; k < SP_K; ++k) {
        nd[k] = 0xFFFFu;
        nw[k] = 0.f;
        nq[k] = 0.f;
        if ((uint32_t)(k * SP_THREADS) < D.total) {   // block-uniform
          const uint32_t f = tid + k * SP_THREADS;
          const bool ok = f < D.total;
          uint32_t i;
          sp_map(D, ok ? f : 0, i, nq[k]);
          const uint32_t d = a.ix.doc_local[i];
          nw[k] = a.ix.w[i];
          nd[k] = ok ? d : 0xFFFFu;
        }
      }
    }
    SP_STAMP(1)
    if (total) {   // block-uniform
      npost += total;
      const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
      const bool tails = total > (uint32_t)(SP_K * SP_THREADS);   // block-uniform
      sp_make_room(S, a, cand, total, tid);
      // (4) accumulate: integer atomics, order-independent
#pragma unroll
      for (int k = 0; k < SP_K; ++k) {
        if (pd[k] != 0xFFFFu) atomicAdd(&S.acc[pd[k]], sp_fix(pq[k], pw[k]));
      }
      if (tails) {
        const SpDir D = sp_dir_read(S, cur);
        for (uint32_t f = tid + SP_K * SP_THREADS; f < total; f += SP_THREADS) {
          uint32_t i;
          float qw;
          sp_map(D, f, i, qw);
          atomicAdd(&S.acc[a.ix.doc_local[i]], sp_fix(qw, a.ix.w[i]));
        }
      }
      lds_barrier();
      SP_STAMP(2)
      // (5) harvest: sweep the accumulator
      sp_harvest(S, a, cand, park, seg, total, tid);
      SP_STAMP(3)
    }
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      pd[k] = nd[k];
      pw[k] = nw[k];
      pq[k] = nq[k];
    }
    total = ntotal;
    cur = nxt;
    SP_STAMP(4)
  }
  SP_STAMP_FLUSH
  lds_barrier();
  if (tid == 0 && a.stat_postings) atomicAdd(a.stat_postings, npost);
  sp_finish(S, a, cand, q, part, tid);
}

// ---------------------------------------------------------------------------------
// kernel: pipelined body for short queries, generic loop otherwise
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS, 4) void k_sparse_score(SparseQueryArgs a) {
  __shared__ SpShared S;
  const int tid = threadIdx.x;
  const int slot = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
  const int q = a.q_order ? a.q_order[slot] : slot;     // heaviest queries first
  const int nseg = a.ix.n_segments;
  const int s0 = (int)((int64_t)nseg * part / a.parts), s1 = (int)((int64_t)nseg * (part + 1) / a.parts);
  const int64_t qb = a.q_indptr[q];
  const int T = (int)(a.q_indptr[q + 1] - qb);
  uint64_t* cand = a.cand + (int64_t)blockIdx.x * SP_CAND;
  unsigned long long* park = a.park + (int64_t)blockIdx.x * (SEG_DOCS / 2);
  static_assert(SP_CAND == SEG_DOCS, "the accumulator doubles as the sort scratch");

  for (int i = tid; i < SEG_DOCS; i += SP_THREADS) S.acc[i] = 0;
  if (tid == 0) {
    S.cnt = 0;
    S.tau = -__builtin_inff();
  }
  __syncthreads();
  if (T <= SP_TMAX && T > 0 && s0 < s1) {   // block-uniform
    sp_body_pipe(S, a, cand, park, q, part, s0, s1, qb, T, tid);
    return;
  }
  unsigned long long npost = 0;
  for (int seg = s0; seg < s1; ++seg) {
    uint32_t total = 0;
    sp_make_room(S, a, cand, SEG_DOCS, tid);   // acc is all zero between segments
    for (int tc = 0; tc < T; tc += SP_TCH) {
      const int nt = (T - tc) < SP_TCH ? (T - tc) : SP_TCH;
      if (tid < nt) {
        const uint32_t term = (uint32_t)a.q_idx[qb + tc + tid];
        Probe p = sp_probe_issue(a, seg, term);
        uint32_t off, len;
        sp_probe_resolve(a, p, seg, term, off, len);
        S.t_off[0][tid] = off;
        S.t_len[0][tid] = len;
        S.t_w[tid] = a.q_val[qb + tc + tid];
      }
      __syncthreads();
      for (int t = 0; t < nt; ++t) {
        const uint32_t len = S.t_len[0][t], off = S.t_off[0][t];
        const float qw = S.t_w[t];
        total += len;
        for (uint32_t i = tid; i < len; i += SP_THREADS) {
          atomicAdd(&S.acc[a.ix.doc_local[off + i]], sp_fix(qw, a.ix.w[off + i]));
        }
      }
      __syncthreads();  // t_off/t_len are rewritten by the next round
    }
    if (total) {  // block-uniform
      npost += total;
      sp_harvest(S, a, cand, park, seg, total, tid);
      __syncthreads();
    }
  }
  if (tid == 0 && a.stat_postings) atomicAdd(a.stat_postings, npost);
  sp_finish(S, a, cand, q, part, tid);
}

// order queries by descending term count (longest-processing-time first): one block
__global__ void k_sparse_order(const int64_t* q_indptr, int B, int* q_order) {
  __shared__ int hist[SP_TCH + 2];
  const int tid = threadIdx.x;
  for (int i = tid; i < SP_TCH + 2; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) {
    int T = (int)(q_indptr[b + 1] - q_indptr[b]);
    T = T > SP_TCH ? SP_TCH : T;
    atomicAdd(&hist[SP_TCH - T], 1);   // bucket 0 = longest
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i <= SP_TCH; ++i) {
      const int c = hist[i];
      hist[i] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) {
    int T = (int)(q_indptr[b + 1] - q_indptr[b]);
    T = T > SP_TCH ? SP_TCH : T;
    q_order[atomicAdd(&hist[SP_TCH - T], 1)] = b;
  }
}

#ifdef HX_SP_STAMP
extern "C" int hx_debug_sp_stamps(unsigned long long* out_host, int n) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sp_stamps), (size_t)n * 8);
}
#endif

void launch_sparse_score(const SparseQueryArgs& a, hipStream_t st) {
  if (a.B <= 0) return;
  HX_CHECK(a.limit * 2 <= SP_CAND, "sparse: limit too large");
  if (a.q_order) {
    hipLaunchKernelGGL(k_sparse_order, dim3(1), dim3(1024), 0, st, a.q_indptr, a.B, a.q_order);
    HX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_sparse_score, dim3(a.B * a.parts), dim3(SP_THREADS), 0, st, a);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
