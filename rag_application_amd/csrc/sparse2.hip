// K7, pass 1 -- candidate selection for the sparse ("BM25") stage over the on-device inverted index.
//
// Mirrors Prefetch(query=SparseVector, using="sparse", limit=sparse_limit)
// (app/core/vector_store/qdrant/qdrant_handler.py:347-354): score(d) = sum over the query's terms of
// q_t * d_t, IDF-free (the collection sets no sparse modifier, :80-86); only documents that share a term
// with the query are candidates.  The exact score (upstream's order: terms in ascending id, fp32 mul,
// fp32 add) is computed by pass 2 (sprescore.hip) for the candidates this pass keeps.
//
// This pass is integer work.  Per query the host-side preparation (k_sparse_prep) fixes
//     scale = (65535 - T - 8) / (wmax * sum_t q_t)        qs_t = f32(q_t * scale)
// and a posting contributes v = trunc(f32(w * qs_t)) + 1 >= 1 to a 16-bit accumulator; with u = exact
// score * scale and k <= T matching terms, a - 1.0078 k <= u <= a + 0.0078 k, and a < 65536 always.
// If a_L is the L-th best accumulator value seen, a document with a <= a_L - M (M = T + T/16 + 4)
// scores at least three units below L documents -- beyond what fp32 rounding of the exact score can
// bridge (0.26 unit) -- so it cannot be in, or tie with, the exact top-L.  The pass keeps every
// document with a > a_L - M: a superset of the exact top-L whose size is L plus the few documents
// within M units of the L-th.
//
// Index layout (spbuild.hip): TERM-major postings sorted by (term, document) as {document index inside
// its segment, fp32 weight}; documents are cut into segments of SEG_DOCS and a dense table gives, for
// every live term and segment, the offset of the term's first posting in that segment or later.
//
// One workgroup owns (query, part): a contiguous range of segments and an LDS accumulator of one 16-bit
// half-word per document of a segment (document d: word d mod SEG_DOCS/2, half d div SEG_DOCS/2, so the
// neighbours of a posting run never share a word).  Lane t of every wave holds query term t (T <= 64):
// its table row and the run [p0, p1) of the current segment.  A run is cut into chunks of 128 postings
// (two per lane); every wave derives the same chunk list from a wave-wide DPP scan of the chunk counts
// and wave w takes chunks w, w + W, ...: the first SP_K of them are loaded one visit AHEAD into
// registers, the rest (unusually dense segments) straight from memory.
//     visit(s):  derive the chunks of s + 1, issue their posting loads and the table offsets of s + 3
//                accumulate(s): ds_add_u32 of v << (16 * half)              -- barrier X --
//                harvest(s): ds_and_rtn_b32 clears the half and returns the word: the lane that gets a
//                non-zero half back owns the document; a >= tau appends a key     -- barrier Y --
// Both barriers wait for LDS only, so the loads of the next visit stay in flight across them.
// Survivors go to a workgroup-private buffer in global memory that is sorted through the (then all-zero)
// accumulator and cut to {a > a_L - M} whenever it has grown enough; tau follows.
#include "hx_common.hpp"
#include "kernels.hpp"
#include "wsort.hpp"

// Compiled twice (rag_application_amd/build.py): -DHX_SP_VARIANT=v32k -DHX_SEG_DOCS=32768 -DHX_SP_THREADS=512
// (two workgroups per CU) and -DHX_SP_VARIANT=v64k -DHX_SEG_DOCS=65536 -DHX_SP_THREADS=1024 (one per CU).
#ifndef HX_SP_VARIANT
#define HX_SP_VARIANT v32k
#endif
#ifndef HX_SEG_DOCS
#define HX_SEG_DOCS 32768
#endif
#ifndef HX_SP_THREADS
#define HX_SP_THREADS 512
#endif

namespace hx {
namespace HX_SP_VARIANT {

constexpr int SEG_DOCS = HX_SEG_DOCS;
constexpr int SEG_WORDS = SEG_DOCS / 2;      // accumulator words (two documents each)
constexpr int SEG_WSHIFT = SEG_DOCS == 65536 ? 15 : 14;
static_assert((1 << SEG_WSHIFT) == SEG_WORDS, "segment size");
constexpr int SP_CAP = SEG_DOCS / 4;         // candidate keys the LDS can sort (the accumulator as scratch)
constexpr int SP_GCAP = SEG_DOCS + SP_CAP / 2;   // keys of the workgroup's buffer in global memory: a visit appends
                                             // at most one key per document of the segment
constexpr int SP_HSHIFT = SEG_DOCS == 65536 ? 1 : 2;   // histogram pre-filter: SEG_WORDS 32-bit bins of 2 / 4 scores
constexpr int SP_THREADS = HX_SP_THREADS;
constexpr int SP_WAVES = SP_THREADS / 64;
constexpr int SP_K = 4;                      // chunks per wave and visit held in registers
constexpr int SP_CH = 128;                   // postings per chunk: two per lane

struct SpShared {
  union {
    uint32_t acc[SEG_WORDS];                 // two 16-bit integer scores per word
    uint64_t sort[SP_CAP];                   // sort scratch while acc is all zero
  };
  int cnt;                                   // candidates in the workgroup's global buffer
  int trig;                                  // cnt at which the buffer is sorted and cut
  int ovf;                                   // an append found the buffer full
  int redo;                                  // the register cut kept all 256: use the general cut
  uint32_t tau;                              // append threshold (integer score)
  int n2;                                    // pre-filter: keys loaded into the sort scratch
  uint32_t pre;                              // pre-filter threshold
  uint32_t scan[HX_SP_THREADS];              // pre-filter: per-thread bin sums
};
// One object at namespace scope: every access is provably LDS (ds_* instructions).
__shared__ SpShared g_sp;
#define S g_sp

// LDS-only barrier: does not wait for outstanding global loads
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// the candidate buffer is written and re-read by the waves of one workgroup: agent-scope accesses
// (sc1) so that no wave reads a stale L1 line
__device__ __forceinline__ uint64_t sp_ld_key(const uint64_t* p) {
  return __hip_atomic_load((const __attribute__((address_space(1))) uint64_t*)p, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sp_st_key(uint64_t* p, uint64_t v) {
  __hip_atomic_store((__attribute__((address_space(1))) uint64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// key of an integer score: descending key order = (score desc, id asc)
__device__ __forceinline__ uint64_t sp_key(uint32_t a, uint32_t gid) {
  return ((uint64_t)a << 32) | (uint64_t)(0xFFFFFFFFu - gid);
}

// threshold of a sorted list whose L-th best integer score is aL: keep a > aL - M
__device__ __forceinline__ uint32_t sp_thr(uint32_t aL, int M) {
  const int t = (int)aL - M + 1;
  return t < 1 ? 1u : (uint32_t)t;
}

// Sort the workgroup's candidate buffer (global) through LDS, keep {a >= thr(a_L)}, raise tau.
// Precondition: acc is all zero and every wave is past its last acc access.
__device__ __forceinline__ void sp_cut(uint64_t* cand, int limit, int M, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends have left the core
  __syncthreads();                                   // ... and every other wave's; S.cnt settled
  int n = S.cnt;
  n = n < SP_GCAP ? n : SP_GCAP;
  bool general = !(limit <= 256 && n <= SP_WAVES * 256);   // block-uniform
  if (!general) {
    // every wave sorts 256 keys in registers (wsort.hpp), then log2(SP_WAVES) pairwise folds through the
    // sort scratch keep the best 256: one barrier per fold instead of one per bitonic stage
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = w * 256 + e * 64 + lane;
      v[e] = i < n ? sp_ld_key(cand + i) : 0ull;
    }
    if (n > w * 256) w_sort<256>(v, lane);
#pragma unroll
    for (int s = 0; (1 << s) < SP_WAVES; ++s) {
      const int m = (2 << s) - 1;
      if ((w & m) == (1 << s)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) S.sort[w * 256 + lane * 4 + e] = v[e];
      }
      lds_barrier();   // (every wave's loads of cand were consumed by its sort: they precede wave 0's stores)
      const int pw = w + (1 << s);
      if ((w & m) == 0 && n > pw * 256) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k64max(v[e], S.sort[pw * 256 + 255 - (lane * 4 + e)]);
        w_merge<256, 128>(v, lane);
      }
    }
    if (w == 0) {
      uint32_t thr = 1;
      if (n >= limit) {
        const int kr = limit - 1;
        const uint64_t mine = (kr & 3) == 0 ? v[0] : ((kr & 3) == 1 ? v[1] : ((kr & 3) == 2 ? v[2] : v[3]));
        const uint64_t kth = (uint64_t)__shfl((unsigned long long)mine, kr >> 2, 64);
        thr = sp_thr((uint32_t)(kth >> 32), M);
      }
      int nk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) nk += __popcll(__ballot((uint32_t)(v[e] >> 32) >= thr));   // empty slots score 0
      const bool cut_short = n > 256 && nk == 256;   // the 256 kept all pass: more may lie beyond
      if (!cut_short) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((uint32_t)(v[e] >> 32) >= thr) sp_st_key(cand + lane * 4 + e, v[e]);   // sorted: a prefix
        if (lane == 0) {
          S.tau = thr;
          S.cnt = nk;
        }
      }
      if (lane == 0) S.redo = cut_short ? 1 : 0;
    }
    lds_barrier();     // every fold has read its partner's slice; S.redo is set
    if (w != 0) {      // acc back to zero: each wave but 0 wrote its slice at most once
#pragma unroll
      for (int e = 0; e < 4; ++e) S.sort[w * 256 + lane * 4 + e] = 0ull;
    }
    general = S.redo != 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!general) return;
  }
  // ---- general cut
  uint32_t pre = 0;
  if (n > SP_CAP) {
    // More keys than the LDS can sort (the first visits of a dense query, before there is a threshold): the
    // scores are 16-bit integers, so a histogram over the (all-zero) accumulator finds a lower bound of the
    // L-th best score, and only the keys within the margin of THAT are sorted.
    constexpr int PER = SEG_WORDS / SP_THREADS;      // bins per thread
    for (int i = tid; i < n; i += SP_THREADS)
      __hip_atomic_fetch_add(&S.acc[(uint32_t)(sp_ld_key(cand + i) >> 32) >> SP_HSHIFT], 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_barrier();
    uint32_t mine = 0;
#pragma unroll 8
    for (int j = 0; j < PER; ++j) mine += S.acc[tid * PER + j];
    S.scan[tid] = mine;
    if (tid == 0) S.pre = 1u;
    lds_barrier();
    uint32_t above = 0;                              // keys in the bins of the threads above this one
    for (int t = tid + 1; t < SP_THREADS; ++t) above += S.scan[t];
    if (above < (uint32_t)limit && above + mine >= (uint32_t)limit) {   // the L-th best key lies in this thread's bins
      uint32_t run = above;
      for (int j = PER - 1; j >= 0; --j) {
        run += S.acc[tid * PER + j];
        if (run >= (uint32_t)limit) {
          S.pre = sp_thr((uint32_t)(tid * PER + j) << SP_HSHIFT, M);   // lower edge of the bin: <= the true a_L
          break;
        }
      }
    }
    lds_barrier();
    pre = S.pre;
    for (int i = tid; i < SEG_WORDS; i += SP_THREADS) S.acc[i] = 0u;
    if (tid == 0) S.n2 = 0;
    lds_barrier();
    for (int i = tid; i < n; i += SP_THREADS) {
      const uint64_t k = sp_ld_key(cand + i);
      if ((uint32_t)(k >> 32) >= pre) {
        const int pos = atomicAdd(&S.n2, 1);
        if (pos < SP_CAP) S.sort[pos] = k;
      }
    }
    lds_barrier();
    if (S.n2 > SP_CAP && tid == 0) S.ovf = 1;        // more keys within the margin than can be sorted: exact path
    n = S.n2 < SP_CAP ? S.n2 : SP_CAP;
  }
  int P = SP_THREADS;                                // sort size: next power of two >= n
  while (P < n) P <<= 1;
  if (!pre) {
    for (int i = tid; i < P; i += SP_THREADS) S.sort[i] = i < n ? sp_ld_key(cand + i) : 0ull;
  }
  lds_barrier();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += SP_THREADS) {
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
  if (tid == 0) {
    uint32_t thr = 1;
    if (n >= limit) thr = sp_thr((uint32_t)(S.sort[limit - 1] >> 32), M);
    int lo = 0, hi = n;                              // first index whose score is below thr (descending list)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)(S.sort[mid] >> 32) >= thr) lo = mid + 1; else hi = mid;
    }
    S.tau = thr;
    S.cnt = lo;
  }
  lds_barrier();
  const int keep = S.cnt;
  for (int i = tid; i < keep; i += SP_THREADS) sp_st_key(cand + i, S.sort[i]);
  lds_barrier();
  for (int i = tid; i < P; i += SP_THREADS) S.sort[i] = 0ull;   // acc back to zero
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ void sp_append(uint64_t* cand, uint32_t a, uint32_t gid) {
  const int pos = atomicAdd(&S.cnt, 1);
  if (pos < SP_GCAP) sp_st_key(cand + pos, sp_key(a, gid));
  else S.ovf = 1;
}

// ---------------------------------------------------------------------------------
// directory: per-lane run offsets -> wave-uniform chunks
// ---------------------------------------------------------------------------------
struct SpDir {            // lane = term slot
  uint32_t p0, len, incl;  // first posting / postings / inclusive chunk count up to this term
};
// inclusive scan over the 64 lanes: four DPP row shifts (zeros shifted in), then the row totals
// travel with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3)
__device__ __forceinline__ uint32_t sp_wavescan(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xF, 0xF, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xF, 0xF, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xF, 0xF, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, false);   // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, false);   // row_bcast:31
  return v;
}
__device__ __forceinline__ SpDir sp_dir(uint32_t p0, uint32_t p1, bool active, uint32_t& nch) {
  SpDir d;
  d.p0 = p0;
  d.len = active ? p1 - p0 : 0u;
  d.incl = sp_wavescan((d.len + (SP_CH - 1)) / SP_CH);
  nch = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, 63);
  return d;
}
// chunk c of the segment: first posting, postings in it (1..128), scaled query weight.  c < nch.
__device__ __forceinline__ void sp_chunk(const SpDir& d, float qs_lane, uint32_t c, uint32_t& off, uint32_t& cnt,
                                         float& qs) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(d.incl > c);
  const int t = __builtin_ctzll(m);                     // m != 0 because c < nch = incl[63]
  const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)d.len, t);
  const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, t) - ((len + (SP_CH - 1)) / SP_CH);
  const uint32_t j = (c - start) * SP_CH;
  off = (uint32_t)__builtin_amdgcn_readlane((int)d.p0, t) + j;
  cnt = len - j < (uint32_t)SP_CH ? len - j : (uint32_t)SP_CH;
  qs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qs_lane), t));
}

// the two postings of lane `lane` in a chunk: {doc0, w0, doc1, w1}; only the first min(cnt - 2 lane, 2) count.
// A run starts at any posting, so the 16-byte load is 8-byte aligned only (one global_load_dwordx4 all the same).
struct __attribute__((aligned(8))) SpPair { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 sp_load2(const uint2* post, uint32_t off, uint32_t cnt, int lane) {
  const uint32_t i = (uint32_t)(2 * lane) < cnt ? (uint32_t)(2 * lane) : 0u;   // (the array is padded by one posting)
  const SpPair p = *(const SpPair*)(post + off + i);
  return make_uint4(p.x, p.y, p.z, p.w);
}
__device__ __forceinline__ uint32_t sp_units(uint32_t wbits, float qs) {
  return (uint32_t)__fmul_rn(__builtin_bit_cast(float, wbits), qs) + 1u;       // trunc(w * qs) + 1
}
__device__ __forceinline__ void sp_add1(uint32_t doc, uint32_t v) {
  __hip_atomic_fetch_add(&S.acc[doc & (SEG_WORDS - 1)], v << ((doc >> SEG_WSHIFT) << 4), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sp_accumulate(const uint4& p, uint32_t cnt, float qs, int lane) {
  if ((uint32_t)(2 * lane) < cnt) sp_add1(p.x, sp_units(p.y, qs));
  if ((uint32_t)(2 * lane + 1) < cnt) sp_add1(p.z, sp_units(p.w, qs));
}
__device__ __forceinline__ uint32_t sp_take1(uint32_t doc) {
  const uint32_t sh = (doc >> SEG_WSHIFT) << 4;
  const uint32_t old = __hip_atomic_fetch_and(&S.acc[doc & (SEG_WORDS - 1)], ~(0xFFFFu << sh), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
  return (old >> sh) & 0xFFFFu;
}
__device__ __forceinline__ void sp_harvest(const uint4& p, uint32_t cnt, int lane, uint64_t* cand, uint32_t tau,
                                           uint32_t gbase) {
  uint32_t a0 = 0, a1 = 0;
  if ((uint32_t)(2 * lane) < cnt) a0 = sp_take1(p.x);
  if ((uint32_t)(2 * lane + 1) < cnt) a1 = sp_take1(p.z);
  if (a0 >= tau && a0 != 0) sp_append(cand, a0, gbase + p.x);
  if (a1 >= tau && a1 != 0) sp_append(cand, a1, gbase + p.z);
}

// ---------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS, 4) void k_sparse_select(SparseSelectArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
  const int q = a.q_order ? a.q_order[slot] : slot;     // heaviest queries first
  const int nseg = a.ix.n_segments;
  const int s0 = (int)((int64_t)nseg * part / a.parts), s1 = (int)((int64_t)nseg * (part + 1) / a.parts);
  const int64_t qb = a.q_indptr[q];
  const int T = (int)(a.q_indptr[q + 1] - qb);
  uint64_t* cand = a.cand + (int64_t)blockIdx.x * SP_GCAP;
  uint64_t* o = a.out + ((int64_t)q * a.parts_total + a.part0 + part) * a.lout;
  int* ocnt = a.out_cnt + (int64_t)q * a.parts_total + a.part0 + part;
  if (a.q_flag[q] != 0 || T <= 0 || s0 >= s1) {         // block-uniform: nothing for this pass to do
    for (int i = tid; i < a.lout; i += SP_THREADS) o[i] = 0ull;
    if (tid == 0) *ocnt = 0;
    return;
  }
  const int M = a.q_margin[q];
  for (int i = tid; i < SEG_WORDS; i += SP_THREADS) S.acc[i] = 0u;
  if (tid == 0) {
    S.cnt = 0;
    S.trig = 2 * a.limit < 256 ? 256 : 2 * a.limit;     // first cut early: it gives the first threshold
    S.ovf = 0;
    S.redo = 0;
    S.tau = 1u;
  }
  // term slot: table row and scaled query weight (absent terms and unused slots are inactive)
  int ti = -1;
  float qs_lane = 0.f;
  if (lane < T) {
    ti = a.q_ti[(int64_t)q * SP_TMAX + lane];
    qs_lane = a.q_qs[(int64_t)q * SP_TMAX + lane];
  }
  const bool active = ti >= 0;
  const uint32_t* row = a.ix.ptr + (int64_t)(active ? ti : 0) * (nseg + 1);
  const uint2* post = a.ix.post;
  auto clampi = [&](int x) { return x <= nseg ? x : nseg; };   // row[nseg] is the end of the term's postings
  __syncthreads();

  // offsets of the segment being visited, the next one and the one after (per lane); the postings of a
  // visit are in flight since the visit before
  uint32_t o0 = row[clampi(s0)], o1 = row[clampi(s0 + 1)], o2 = row[clampi(s0 + 2)];
  uint32_t nch;
  SpDir d = sp_dir(o0, o1, active, nch);
  uint4 cur[SP_K];
  uint32_t ccnt[SP_K];
  float cqs[SP_K];
#pragma unroll
  for (int k = 0; k < SP_K; ++k) {
    const uint32_t c = (uint32_t)(k * SP_WAVES + wave);
    ccnt[k] = 0;
    cqs[k] = 0.f;
    cur[k] = make_uint4(0, 0, 0, 0);
    if (c < nch) {
      uint32_t off;
      sp_chunk(d, qs_lane, c, off, ccnt[k], cqs[k]);
      cur[k] = sp_load2(post, off, ccnt[k], lane);
    }
  }
  for (int seg = s0; seg < s1; ++seg) {
    const uint32_t gbase = (uint32_t)(a.ix.id_base + (int64_t)seg * SEG_DOCS);
    // ---- the next visit: directory from the offsets that have landed, its posting loads, the offsets after
    uint32_t nch_n;
    const SpDir dn = sp_dir(o1, o2, active, nch_n);
    uint4 nxt[SP_K];
    uint32_t ncnt[SP_K];
    float nqs[SP_K];
    const bool more = seg + 1 < s1;                     // scalar
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      const uint32_t c = (uint32_t)(k * SP_WAVES + wave);
      ncnt[k] = 0;
      nqs[k] = 0.f;
      nxt[k] = make_uint4(0, 0, 0, 0);
      if (more && c < nch_n) {
        uint32_t off;
        sp_chunk(dn, qs_lane, c, off, ncnt[k], nqs[k]);
        nxt[k] = sp_load2(post, off, ncnt[k], lane);
      }
    }
    const uint32_t o3 = row[clampi(seg + 3)];
    // ---- this visit
    if (nch) {                                          // scalar: the segment holds postings of the query
      int cnt = __builtin_amdgcn_readfirstlane(S.cnt);
      const int trig = __builtin_amdgcn_readfirstlane(S.trig);
      const uint32_t bound = nch * SP_CH < (uint32_t)SEG_DOCS ? nch * SP_CH : (uint32_t)SEG_DOCS;
      if (cnt > a.limit && (cnt >= trig || (uint32_t)cnt + bound > (uint32_t)SP_GCAP)) {
        sp_cut(cand, a.limit, M, tid);
        cnt = __builtin_amdgcn_readfirstlane(S.cnt);
        if (tid == 0) {                                 // later cuts: when the buffer has grown by a few lists
          const int t = cnt + (4 * a.limit < 1024 ? 1024 : 4 * a.limit);
          S.trig = t < SP_CAP * 3 / 4 ? t : SP_CAP * 3 / 4;
        }
      }
      const uint32_t tau = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.tau);
#pragma unroll
      for (int k = 0; k < SP_K; ++k)
        if (ccnt[k]) sp_accumulate(cur[k], ccnt[k], cqs[k], lane);
      for (uint32_t c = (uint32_t)(SP_K * SP_WAVES + wave); c < nch; c += SP_WAVES) {   // dense segment: the rest
        uint32_t off, n;
        float qs;
        sp_chunk(d, qs_lane, c, off, n, qs);
        sp_accumulate(sp_load2(post, off, n, lane), n, qs, lane);
      }
      lds_barrier();                                    // ---- X: every posting of the segment is in
#pragma unroll
      for (int k = 0; k < SP_K; ++k)
        if (ccnt[k]) sp_harvest(cur[k], ccnt[k], lane, cand, tau, gbase);
      for (uint32_t c = (uint32_t)(SP_K * SP_WAVES + wave); c < nch; c += SP_WAVES) {
        uint32_t off, n;
        float qs;
        sp_chunk(d, qs_lane, c, off, n, qs);
        sp_harvest(sp_load2(post, off, n, lane), n, lane, cand, tau, gbase);
      }
      lds_barrier();                                    // ---- Y: acc is all zero again
    }
    // ---- rotate
    o0 = o1;
    o1 = o2;
    o2 = o3;
    d = dn;
    nch = nch_n;
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      cur[k] = nxt[k];
      ccnt[k] = ncnt[k];
      cqs[k] = nqs[k];
    }
  }
  (void)o0;
  // ---- the part's list: cut once more, then the kept keys (best first)
  sp_cut(cand, a.limit, M, tid);
  const int nk = S.cnt;
  const int n = nk < a.lout ? nk : a.lout;
  for (int i = tid; i < a.lout; i += SP_THREADS) o[i] = i < n ? sp_ld_key(cand + i) : 0ull;
  if (tid == 0) {
    *ocnt = n;
    if (nk > a.lout || S.ovf) a.q_fail[q] = 1;          // cut short: the query takes the exact path
  }
}
#undef S

void launch_sparse_select_variant(const SparseSelectArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.parts <= 0) return;
  HX_CHECK(a.ix.seg_docs == SEG_DOCS, "sparse: index built for another segment size");
  HX_CHECK(a.limit >= 1 && a.limit <= a.lout && a.lout <= SP_CAP / 2, "sparse: limit too large");
  hipLaunchKernelGGL(k_sparse_select, dim3(a.B * a.parts), dim3(SP_THREADS), 0, st, a);
  HX_HIP(hipGetLastError());
}

}  // namespace HX_SP_VARIANT
}  // namespace hx
