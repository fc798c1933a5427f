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
// Index layout (spbuild.hip): TERM-major postings sorted by (term, document) as {place of the document's
// accumulator inside its segment (sp_word below), fp32 weight}; documents are cut into segments of SEG_DOCS
// and a dense table gives, for every live term and segment, the offset of the term's first posting in that
// segment or later.
//
// One workgroup owns (query, part): a contiguous range of segments and an LDS accumulator of one 16-bit
// half-word per document of a segment (document d: word d mod SEG_DOCS/2, half d div SEG_DOCS/2, so the
// neighbours of a posting run never share a word).  Lane t of every wave holds query term t (T <= 64):
// its table row and the run [p0, p1) of the current segment.  A run is cut into chunks of 128 postings
// (two per lane); every wave derives the same chunk list from a wave-wide DPP scan of the chunk counts
// and wave w takes chunks w, w + W, ...: the first SP_K of them are loaded two visits AHEAD into
// registers, the rest (unusually dense segments) straight from memory.
//     visit(s):  derive the chunks of s + 2, issue their posting loads; read the table offsets of s + 3
//                accumulate(s): ds_add_u32 of v << (16 * half)              -- barrier X --
//                harvest(s): ds_and_rtn_b32 clears the half and returns the word: the lane that gets a
//                non-zero half back owns the document; a >= tau appends a key     -- barrier Y --
// Both barriers wait for LDS only, so the loads of the next visit stay in flight across them.
// Survivors go to a workgroup-private buffer in global memory that is cut to {a > a_L - M} whenever it has
// grown enough -- a histogram of the 16-bit scores over the (then all-zero) accumulator finds a_L, nothing is
// sorted before the last cut (sp_cut); tau follows.
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
#ifndef HX_SP_EXP
#define HX_SP_EXP 0      // timing experiments only (wrong results): 1-3 LDS atomics removed, 4 no posting loads, 5 aligned loads, 6 no LDS traffic, 7 posting loads from L2
#endif

namespace hx {
namespace HX_SP_VARIANT {

constexpr int SEG_DOCS = HX_SEG_DOCS;
constexpr int SEG_WORDS = SEG_DOCS / 2;      // accumulator words (two documents each)
constexpr int SEG_WSHIFT = SEG_DOCS == 65536 ? 15 : 14;
static_assert((1 << SEG_WSHIFT) == SEG_WORDS, "segment size");
constexpr int SP_CAP = SEG_DOCS / 4;         // candidate keys the LDS can stage at a cut (the accumulator as scratch)
constexpr int SP_GCAP = SEG_DOCS + SP_CAP / 2;   // keys of the workgroup's buffer in global memory: a visit appends
                                             // at most one key per document of the segment
constexpr int SP_HSHIFT = SEG_DOCS == 65536 ? 1 : 2;   // histogram pre-filter: SEG_WORDS 32-bit bins of 2 / 4 scores
constexpr int SP_THREADS = HX_SP_THREADS;
constexpr int SP_WAVES = SP_THREADS / 64;
#ifndef HX_SP_K
#define HX_SP_K 3
#endif
constexpr int SP_K = HX_SP_K;                // chunks per wave and visit held in registers
constexpr int SP_CH = 128;                   // postings per chunk: two per lane
constexpr int SP_KPT = 8;                    // keys per thread a cut holds in registers

// Diagnostic build only (-DHX_SP_STAMP): lane 0 of waves 0 and 5 accumulate s_memtime deltas per phase of a
// visit into a debug buffer of its own (never read by the kernel, never in a timed build).
#ifdef HX_SP_STAMP
__device__ unsigned long long g_sp_stamps[2 * 8 * 4096];
#define SP_STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SP_STAMP(i) { const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; }
#define SP_STAMP_CUT st_acc[6] += 1ull << 40;
#define SP_STAMP_FLUSH if ((tid == 0 || tid == 320) && blockIdx.x < 4096) for (int i_ = 0; i_ < 8; ++i_) g_sp_stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i_] = st_acc[i_];
#else
#define SP_STAMP_DECL
#define SP_STAMP(i)
#define SP_STAMP_CUT
#define SP_STAMP_FLUSH
#endif

struct SpShared {
  union {
    uint32_t acc[SEG_WORDS];                 // two 16-bit integer scores per word
    uint64_t sort[SP_CAP];                   // the kept keys of a cut (sorted at the last one) while acc is all zero
  };
  int cnt;                                   // candidates in the workgroup's global buffer
  int ovf;                                   // an append found the buffer full
  uint32_t tau;                              // append threshold (integer score)
  int n2;                                    // cut: keys staged in the sort scratch
  uint32_t pre;                              // cut: the new threshold
  uint32_t scan[HX_SP_THREADS / 64];         // cut: per-wave histogram sums
  uint32_t otab[HX_SEG_DOCS == 65536 ? 6144 : 2560];   // run offsets of the query's terms for a window of segments
};
constexpr int SP_OT = HX_SEG_DOCS == 65536 ? 6144 : 2560;
// One object at namespace scope: every access is provably LDS (ds_* instructions).
__shared__ SpShared g_sp;
#define S g_sp

// LDS-only barrier: does not wait for outstanding global loads
#ifndef HX_SP_ACC_BRANCHFREE
#define HX_SP_ACC_BRANCHFREE 0   // 1: measured 1.96-1.97 ms against 1.905-1.91 (the idle adds cost more than the branches)
#endif
#ifndef HX_SP_HARVEST_BATCH
#define HX_SP_HARVEST_BATCH 1
#endif
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
// Cut the workgroup's candidate buffer (global) to {a >= thr}, raise tau.  thr = (a lower bound of the L-th best
// integer score) - M + 1.  No sort: the scores are 16-bit integers, so a histogram over the (all-zero) accumulator
// -- bins of 2^SP_HSHIFT scores -- finds the bin of the L-th best, and the bin's lower edge stands for a_L (a lower
// bound, so the kept set can only be larger; edges grow with a_L, so tau never falls).  The kept keys are staged
// in LDS and written back as a prefix of the buffer.  final: they are also sorted best first -- the part's list
// is consumed in that order (sprescore.hip); intermediate cuts need no order at all.
// Precondition: acc is all zero and every wave is past its last acc access.
__device__ __forceinline__ void sp_cut(uint64_t* cand, int limit, int M, int tid, bool final) {
  // (opaque copy of the thread index: hipcc otherwise hoists the eight 64-bit key addresses of this cut out of the
  // visit loop -- 16 VGPRs held across every visit, which is what made the kernel spill)
  asm volatile("" : "+v"(tid));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends have left the core
  __syncthreads();                                   // ... and every other wave's; S.cnt settled
  int n = S.cnt;
  n = n < SP_GCAP ? n : SP_GCAP;
  if (n < limit && !final) return;                   // block-uniform: no L-th best yet, every key stays
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PER = SEG_WORDS / SP_THREADS;        // bins per thread
  // the keys of this thread: i = tid + j * SP_THREADS; the first SP_KPT of them stay in registers across the
  // passes below (a buffer of up to SP_KPT * SP_THREADS keys -- the usual case -- is read from memory once)
  uint64_t kr[SP_KPT];
#pragma unroll
  for (int j = 0; j < SP_KPT; ++j) {
    const int i = tid + j * SP_THREADS;
    kr[j] = i < n ? sp_ld_key(cand + i) : 0ull;
  }
  uint32_t thr = 1u;
  if (n >= limit) {                                  // block-uniform
#pragma unroll
    for (int j = 0; j < SP_KPT; ++j)
      if (tid + j * SP_THREADS < n)
        __hip_atomic_fetch_add(&S.acc[(uint32_t)(kr[j] >> 32) >> SP_HSHIFT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int i = tid + SP_KPT * SP_THREADS; i < n; i += SP_THREADS)
      __hip_atomic_fetch_add(&S.acc[(uint32_t)(sp_ld_key(cand + i) >> 32) >> SP_HSHIFT], 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_barrier();
    // thread t owns bins [t * PER, (t + 1) * PER) (read rotated: the lanes of a wave then hit different banks)
    uint32_t mine = 0;
#pragma unroll 8
    for (int j = 0; j < PER; ++j) mine += S.acc[tid * PER + ((j + tid) & (PER - 1))];
    const uint32_t incl = sp_wavescan(mine);
    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (lane == 0) S.scan[w] = wtot;
    if (tid == 0) S.pre = 1u;
    lds_barrier();
    uint32_t above = wtot - incl;                    // keys in the bins of the threads above this one
    for (int x = w + 1; x < SP_WAVES; ++x) above += S.scan[x];
    if (above < (uint32_t)limit && above + mine >= (uint32_t)limit) {   // the L-th best key lies in this thread's bins
      uint32_t bins[PER];                            // (one LDS round trip, not PER of them)
#pragma unroll
      for (int j = 0; j < PER; ++j) bins[j] = S.acc[tid * PER + j];
      uint32_t run = above, edge = 0;
      bool found = false;
#pragma unroll
      for (int j = PER - 1; j >= 0; --j) {
        run += bins[j];
        if (!found && run >= (uint32_t)limit) {
          found = true;
          edge = (uint32_t)(tid * PER + j) << SP_HSHIFT;   // lower edge of the bin: <= the true a_L
        }
      }
      S.pre = sp_thr(edge, M);
    }
    lds_barrier();
    thr = S.pre;
#pragma unroll
    for (int j = 0; j < SP_KPT; ++j)
      if (tid + j * SP_THREADS < n) S.acc[(uint32_t)(kr[j] >> 32) >> SP_HSHIFT] = 0u;
    for (int i = tid + SP_KPT * SP_THREADS; i < n; i += SP_THREADS)
      S.acc[(uint32_t)(sp_ld_key(cand + i) >> 32) >> SP_HSHIFT] = 0u;
  }
  if (tid == 0) S.n2 = 0;
  lds_barrier();                                     // acc is all zero again: its words now stage the kept keys
  auto stage = [&](bool p, uint64_t k) {             // called by whole waves
    const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
    if (m) {                                         // wave-uniform: one LDS atomic per wave
      int base = 0;
      if (lane == 0) base = atomicAdd(&S.n2, __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (p && pos < SP_CAP) S.sort[pos] = k;
    }
  };
#pragma unroll
  for (int j = 0; j < SP_KPT; ++j)
    if (j * SP_THREADS < n) stage(tid + j * SP_THREADS < n && (uint32_t)(kr[j] >> 32) >= thr, kr[j]);   // block-uniform guard
  for (int i0 = SP_KPT * SP_THREADS; i0 < n; i0 += SP_THREADS) {   // (block-uniform trip count)
    const int i = i0 + tid;
    const uint64_t k = i < n ? sp_ld_key(cand + i) : 0ull;
    stage(i < n && (uint32_t)(k >> 32) >= thr, k);
  }
  lds_barrier();                                     // (every read of cand above precedes the stores below)
  int keep = S.n2;
  if (keep > SP_CAP) {                               // more keys within the margin than the LDS can stage: exact path
    if (tid == 0) S.ovf = 1;
    keep = SP_CAP;
  }
  if (final && keep > 1) {
    if (keep <= 256) {                               // one wave, in registers (wsort.hpp): the usual case
      if (w == 0) {
        uint64_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = S.sort[lane * 4 + e];       // zero beyond `keep`
        w_sort<256>(v, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) S.sort[lane * 4 + e] = v[e];
      }
      lds_barrier();
    } else {
      int P = 512;                                   // sort size: next power of two >= keep (zeros beyond it)
      while (P < keep) P <<= 1;
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
    }
  }
  for (int i = tid; i < keep; i += SP_THREADS) {
    sp_st_key(cand + i, S.sort[i]);
    S.sort[i] = 0ull;                                // acc back to zero
  }
  if (tid == 0) {
    S.tau = thr;
    S.cnt = keep;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Append the passing scores of a wave's lanes (two per lane) to the workgroup's buffer: ONE LDS atomic per
// wave reserves the range, a lane's slot follows from the ballots (before there is a threshold every touched
// document passes: one atomic per key on one LDS word would serialise the whole workgroup).
__device__ __forceinline__ void sp_append2(uint64_t* cand, bool p0, uint32_t a0, uint32_t g0, bool p1, uint32_t a1,
                                           uint32_t g1) {
  const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0), m1 = __builtin_amdgcn_ballot_w64(p1);
  if (!(m0 | m1)) return;                                // wave-uniform; the usual case once there is a threshold
  const int n0 = __popcll(m0), n1 = __popcll(m1);
  int base = 0;
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) base = atomicAdd(&S.cnt, n0 + n1);
  base = __builtin_amdgcn_readfirstlane(base);
  const int i0 = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
  const int i1 = base + n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
  if (p0) {
    if (i0 < SP_GCAP) sp_st_key(cand + i0, sp_key(a0, g0));
    else S.ovf = 1;
  }
  if (p1) {
    if (i1 < SP_GCAP) sp_st_key(cand + i1, sp_key(a1, g1));
    else S.ovf = 1;
  }
}

// ---------------------------------------------------------------------------------
// directory: per-lane run offsets -> wave-uniform chunks
// ---------------------------------------------------------------------------------
struct SpDir {            // lane = term slot
  uint32_t p0, len, incl;  // first posting / postings / inclusive chunk count up to this term
};
__device__ __forceinline__ SpDir sp_dir(uint32_t p0, uint32_t p1, bool active, uint32_t& nch) {
  SpDir d;
  d.p0 = p0;
  d.len = active ? p1 - p0 : 0u;
  d.incl = sp_wavescan((d.len + (SP_CH - 1)) / SP_CH);
  nch = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, 63);
  return d;
}
// chunk c of the segment: first posting, postings in it (1..128), term slot (lane) of its run.  c < nch.
__device__ __forceinline__ void sp_chunk(const SpDir& d, uint32_t c, uint32_t& off, uint32_t& cnt, int& t) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(d.incl > c);
  t = __builtin_ctzll(m);                               // m != 0 because c < nch = incl[63]
  const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)d.len, t);
  const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, t) - ((len + (SP_CH - 1)) / SP_CH);
  const uint32_t j = (c - start) * SP_CH;
  off = (uint32_t)__builtin_amdgcn_readlane((int)d.p0, t) + j;
  cnt = len - j < (uint32_t)SP_CH ? len - j : (uint32_t)SP_CH;
}
__device__ __forceinline__ float sp_lane_f(float v, int t) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
}

// the two postings of lane `lane` in a chunk: {doc0, w0, doc1, w1}; only the first min(cnt - 2 lane, 2) count.
// A run starts at any posting, so the 16-byte load is 8-byte aligned only (one global_load_dwordx4 all the same).
struct __attribute__((aligned(8))) SpPair { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 sp_load2(const uint2* post, uint32_t off, uint32_t cnt, int lane) {
  const uint32_t i = (uint32_t)(2 * lane) < cnt ? (uint32_t)(2 * lane) : 0u;   // (the array is padded by one posting)
#if HX_SP_EXP == 4      // timing experiment: no posting loads at all
  return make_uint4(((off + i) & 0x7FFFu) << 2, 0x3f800000u, ((off + i + 1) & 0x7FFFu) << 2, 0x3f800000u);
#elif HX_SP_EXP == 5    // timing experiment: 16-byte aligned loads (reads the wrong pair for odd offsets)
  return *(const uint4*)(post + ((off + i) & ~1u));
#endif
#if HX_SP_EXP == 7    // timing experiment: every chunk reads from the first 512 KiB of the postings (L2 hits; wrong results)
  off &= 0xFFFFu;
#endif
  const SpPair p = *(const SpPair*)(post + off + i);
  return make_uint4(p.x, p.y, p.z, p.w);
}
__device__ __forceinline__ uint32_t sp_units(uint32_t wbits, float qs) {
  return (uint32_t)__fmul_rn(__builtin_bit_cast(float, wbits), qs) + 1u;       // trunc(w * qs) + 1
}
// A posting's first word (spbuild.hip: k_make_postings) is the document's place in the accumulator, ready to use:
// bits 0-23 the BYTE offset of its 32-bit word (document index mod SEG_WORDS, times 4), bits 24-28 the shift of
// its 16-bit half inside the word (0 or 16) -- no index arithmetic per posting in the kernel.
__device__ __forceinline__ uint32_t* sp_word(uint32_t e) { return (uint32_t*)((char*)S.acc + (e & 0xFFFFFFu)); }
__device__ __forceinline__ uint32_t sp_doc(uint32_t e) {          // document index inside the segment (appends only)
  return ((e & 0xFFFFFFu) >> 2) | ((e >> 28) << SEG_WSHIFT);
}
__device__ __forceinline__ void sp_add1(uint32_t e, uint32_t v) {
#if HX_SP_EXP == 6      // timing experiment: no LDS traffic at all
  asm volatile("" ::"v"(v << (e >> 24)), "v"(e & 0xFFFFFFu));
#elif HX_SP_EXP == 2 || HX_SP_EXP == 3
  *sp_word(e) = v << (e >> 24);
#else
  __hip_atomic_fetch_add(sp_word(e), v << (e >> 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
// m: bit 0 / bit 1 = the lane's first / second posting counts
__device__ __forceinline__ uint32_t sp_lanebits(uint32_t cnt, int lane) {
  return ((uint32_t)(2 * lane) < cnt ? 1u : 0u) | ((uint32_t)(2 * lane + 1) < cnt ? 2u : 0u);
}
// full: the chunk holds 128 postings (wave-uniform) -- every lane's two postings count, no predication
__device__ __forceinline__ void sp_accumulate(const uint4& p, uint32_t m, float qs, bool full) {
  if (full) {
    sp_add1(p.x, sp_units(p.y, qs));
    sp_add1(p.z, sp_units(p.w, qs));
  } else {
#if HX_SP_ACC_BRANCHFREE
    // no per-posting predicate (two exec-mask branches per slot): a posting that does not count adds 0 to the lane's own word
    uint32_t* const idle = S.acc + (threadIdx.x & 63);
    __hip_atomic_fetch_add((m & 1u) ? sp_word(p.x) : idle, (m & 1u) ? sp_units(p.y, qs) << (p.x >> 24) : 0u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((m & 2u) ? sp_word(p.z) : idle, (m & 2u) ? sp_units(p.w, qs) << (p.z >> 24) : 0u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    if (m & 1u) sp_add1(p.x, sp_units(p.y, qs));
    if (m & 2u) sp_add1(p.z, sp_units(p.w, qs));
#endif
  }
}
__device__ __forceinline__ uint32_t sp_take1(uint32_t e) {
  const uint32_t sh = e >> 24;
#if HX_SP_EXP == 6
  return (e >> 31) + (sh >> 8);
#endif
#if HX_SP_EXP == 1 || HX_SP_EXP == 3
  const uint32_t o = *sp_word(e);
  ((uint16_t*)sp_word(e))[sh >> 4] = 0;
  return (o >> sh) & 0xFFFFu;
#endif
  const uint32_t old = __hip_atomic_fetch_and(sp_word(e), ~(0xFFFFu << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return (old >> sh) & 0xFFFFu;
}
__device__ __forceinline__ void sp_harvest(const uint4& p, uint32_t m, uint64_t* cand, uint32_t tau, uint32_t gbase,
                                           bool full) {
  uint32_t a0 = 0, a1 = 0;
  if (full) {
    a0 = sp_take1(p.x);
    a1 = sp_take1(p.z);
  } else {
    if (m & 1u) a0 = sp_take1(p.x);
    if (m & 2u) a1 = sp_take1(p.z);
  }
  sp_append2(cand, a0 >= tau, a0, gbase + sp_doc(p.x), a1 >= tau, a1, gbase + sp_doc(p.z));   // tau >= 1: a cleared half never passes
}

// The 2 * SP_K takes of SP_K chunks, issued back to back and waited for ONCE (sp_harvest under a per-posting
// predicate put `s_waitcnt lgkmcnt(0)` behind every take: six LDS round trips per wave and round).  A posting that
// does not count ANDs, with all ones, the lane's own word of the accumulator -- 64 idle lanes on ONE word would
// serialise (measured: 2.06 -> 3.40 ms) -- and its result is dropped.
__device__ __forceinline__ void sp_harvest_batch(const uint4 (&p)[SP_K], const uint32_t (&m)[SP_K], int lane, uint64_t* cand,
                                                 uint32_t tau, uint32_t gbase) {
  uint32_t old[2 * SP_K];
  uint32_t* const idle = S.acc + lane;
#pragma unroll
  for (int k = 0; k < SP_K; ++k) {
    const uint32_t e0 = p[k].x, e1 = p[k].z;
    old[2 * k] = __hip_atomic_fetch_and((m[k] & 1u) ? sp_word(e0) : idle, (m[k] & 1u) ? ~(0xFFFFu << (e0 >> 24)) : ~0u,
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old[2 * k + 1] = __hip_atomic_fetch_and((m[k] & 2u) ? sp_word(e1) : idle, (m[k] & 2u) ? ~(0xFFFFu << (e1 >> 24)) : ~0u,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#pragma unroll
  for (int k = 0; k < SP_K; ++k) {
    const uint32_t e0 = p[k].x, e1 = p[k].z;
    const uint32_t a0 = (m[k] & 1u) ? (old[2 * k] >> (e0 >> 24)) & 0xFFFFu : 0u;
    const uint32_t a1 = (m[k] & 2u) ? (old[2 * k + 1] >> (e1 >> 24)) & 0xFFFFu : 0u;
    sp_append2(cand, a0 >= tau, a0, gbase + sp_doc(e0), a1 >= tau, a1, gbase + sp_doc(e1));   // tau >= 1: a cleared half never passes
  }
}

// ---------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS, SEG_DOCS == 65536 ? SP_THREADS / 256 : SP_THREADS / 128) void k_sparse_select(SparseSelectArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int q = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
  if (a.items) {                                        // the launch plan: heaviest items first
    if ((int)blockIdx.x >= *a.n_items) return;
    const int it = a.items[blockIdx.x];
    q = it >> 8;
    part = it & 255;
  }
  const int nseg = a.ix.n_segments;
  int qp = a.q_parts ? a.q_parts[q] : a.parts;          // parts this query is cut into
  qp = qp < nseg ? qp : nseg;
  qp = qp < 1 ? 1 : qp;
  const int s0 = part < qp ? (int)((int64_t)nseg * part / qp) : 0, s1 = part < qp ? (int)((int64_t)nseg * (part + 1) / qp) : 0;
  const int64_t qb = a.q_indptr[q];
  const int T = (int)(a.q_indptr[q + 1] - qb);
  uint64_t* cand = a.cand + (int64_t)blockIdx.x * SP_GCAP;   // (a workgroup's own scratch: any layout will do)
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
    S.ovf = 0;
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
  const uint2* post = a.ix.post;
  // The run offsets (table rows of the query's terms) live in LDS for a window of W1 consecutive segments: one
  // coalesced fill by the whole workgroup every W1 - 4 visits instead of two 64-line gathers per wave and visit
  // -- and no table load queued in front of the posting loads any more.  Entry (t, x) = first posting of term
  // t in segment x or later (x clamped to the end of the term's postings).
  const int Tn = T < SP_TMAX ? T : SP_TMAX;
  const int W1 = SP_OT / (Tn > 0 ? Tn : 1);
  int w0 = s0;                                                  // first segment of the window
  auto fill_window = [&](int from) {                            // workgroup-wide; ends with a barrier
    w0 = from;
    for (int i = tid; i < Tn * W1; i += SP_THREADS) {
      const int t = i / W1, x = from + (i - t * W1);
      const int r = a.q_ti[(int64_t)q * SP_TMAX + t];
      S.otab[i] = r >= 0 ? a.ix.ptr[(int64_t)r * (nseg + 1) + (x <= nseg ? x : nseg)] : 0u;
    }
    __syncthreads();
  };
  auto offs = [&](int x, uint32_t& lo, uint32_t& hi) {          // offsets of segment x: needs w0 <= x, x + 1 < w0 + W1
    lo = hi = 0;
    if (lane < Tn) {
      lo = S.otab[lane * W1 + (x - w0)];
      hi = S.otab[lane * W1 + (x - w0) + 1];
    }
  };
  __syncthreads();

  // Three stages in rotation, stage(x) = x mod 3 holds everything of visit x: the run offsets of the lane's
  // term (loaded three visits ahead), the directory and the posting registers (loaded two visits ahead).
  // At visit s: derive the directory of s + 2 from the offsets that landed and issue its postings into
  // stage(s + 2) (processed last visit), issue the offsets of s + 3 into stage(s) (its own were consumed two
  // visits ago), then move stage(s) -- landed by now -- to the working registers and process it.  A stage is
  // a fixed set of registers (the three-way branch below picks it): nothing in flight is ever copied.
  struct Stage {
    uint32_t nch;            // scalar
    uint4 p[SP_K];
    uint32_t mask;           // 2 bits per slot: which of the lane's two postings count
    uint32_t tpack;          // scalar: 6 bits per slot, the term slot of the chunk; bit 24 + k: slot k is a full chunk
  };
  // (olo, ohi): the lane's run in the segment, read from the window table a visit earlier
  auto issue = [&](Stage& st, uint32_t olo, uint32_t ohi, bool on) {   // directory of the segment, then the posting loads
    const SpDir sd = sp_dir(olo, ohi, active && on, st.nch);
    st.mask = 0;
    st.tpack = 0;
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      const uint32_t c = (uint32_t)(k * SP_WAVES + wave);
      uint32_t off = 0, cnt = 0;
      if (c < st.nch) {
        int t;
        sp_chunk(sd, c, off, cnt, t);
        st.tpack |= ((uint32_t)t << (6 * k)) | (cnt == (uint32_t)SP_CH ? 1u << (24 + k) : 0u);
        st.mask |= sp_lanebits(cnt, lane) << (2 * k);
      }
      // EVERY slot loads (an empty one reads posting 0 and counts nothing): hipcc must see the same number of
      // loads on every path, or its counted waits fall back to draining everything in flight
      st.p[k] = sp_load2(post, off, cnt, lane);
    }
  };
  // one visit on stage `cur`; `nx` = stage(seg + 2).  Returns the chunks of the NEXT visit's segment.
  SP_STAMP_DECL
  uint32_t tau_r = 1u;      // S.tau in a scalar register: it changes at cuts only
  uint32_t nlo = 0, nhi = 0;   // the lane's run offsets of segment seg + 2 at the top of visit(seg)
  auto visit = [&](Stage& cur, Stage& nx, const Stage& nx1, int seg, uint32_t& appended_max) {
    SP_STAMP(0)                                         // loop control, cut check
    issue(nx, nlo, nhi, seg + 2 < s1);                  // postings of seg + 2: two visits ahead
    offs(seg + 3, nlo, nhi);                            // (consumed by the next visit: no LDS round trip at its top)
    SP_STAMP(1)
    const uint32_t nch = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.nch);
    appended_max = nch * SP_CH < (uint32_t)SEG_DOCS ? nch * SP_CH : (uint32_t)SEG_DOCS;
    if (nch) {                                          // scalar: the segment holds postings of the query
      const uint32_t tpack = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.tpack);
      const uint32_t mask = cur.mask;
      const uint32_t gbase = (uint32_t)(a.ix.id_base + (int64_t)seg * SEG_DOCS);
      const uint32_t tau = tau_r;
      // a dense segment: the chunks beyond the prefetched ones, SP_K at a time (their loads issued together); the
      // directory is derived again from the table (a stage does not keep it: registers).  The loads of the first such
      // round are issued BEFORE the prefetched slots are accumulated: their latency (HBM) runs under those adds.
      SpDir d{};
      uint4 r[SP_K];
      uint32_t rm[SP_K];
      float rq[SP_K];
      auto load_round = [&](uint32_t c0) {
#pragma unroll
        for (int k = 0; k < SP_K; ++k) {
          const uint32_t c = c0 + (uint32_t)(k * SP_WAVES);
          rm[k] = 0;
          rq[k] = 0.f;
          r[k] = make_uint4(0, 0, 0, 0);
          if (c < nch) {
            uint32_t off, n;
            int t;
            sp_chunk(d, c, off, n, t);
            rq[k] = sp_lane_f(qs_lane, t);
            rm[k] = sp_lanebits(n, lane);
            r[k] = sp_load2(post, off, n, lane);
          }
        }
      };
      // The units of the prefetched slots FIRST, unconditionally, in straight-line code: these are the uses hipcc places
      // its counted waits in front of (vmcnt(3 SP_K - 1) .. vmcnt(2 SP_K): the loads of the next two visits stay in
      // flight).  Round 3's order -- the dense segment's first round of loads ahead of these uses -- left a different
      // number of loads in flight on the two paths, and hipcc waited with vmcnt(0) here in one visit of three and again
      // behind barrier X: the whole prefetch drained.
      uint32_t u0[SP_K], u1[SP_K];
#pragma unroll
      for (int k = 0; k < SP_K; ++k) {
        const float qk = sp_lane_f(qs_lane, (int)((tpack >> (6 * k)) & 63u));
        u0[k] = sp_units(cur.p[k].y, qk);
        u1[k] = sp_units(cur.p[k].w, qk);
      }
      uint32_t c0 = (uint32_t)(SP_K * SP_WAVES + wave);
      if (nch > (uint32_t)(SP_K * SP_WAVES)) {          // scalar: more chunks than the prefetched slots
        asm volatile("" ::: "memory");                  // (keeps hipcc from hoisting this onto the common path)
        uint32_t n2, olo, ohi;
        offs(seg, olo, ohi);
        d = sp_dir(olo, ohi, active, n2);
        if (c0 < nch) load_round(c0);
      }
#pragma unroll
      for (int k = 0; k < SP_K; ++k) {
        const uint32_t mk = (mask >> (2 * k)) & 3u;
        if ((tpack >> (24 + k)) & 1u) {                 // scalar: a full chunk, no predicate
          sp_add1(cur.p[k].x, u0[k]);
          sp_add1(cur.p[k].z, u1[k]);
        } else {
          if (mk & 1u) sp_add1(cur.p[k].x, u0[k]);
          if (mk & 2u) sp_add1(cur.p[k].z, u1[k]);
        }
      }
      if (c0 < nch) {
        while (c0 < nch) {
#pragma unroll
          for (int k = 0; k < SP_K; ++k) sp_accumulate(r[k], rm[k], rq[k], false);
          c0 += SP_K * SP_WAVES;
          if (c0 < nch) load_round(c0);
        }
        // a real s_waitcnt instruction (not inline asm): hipcc's wait insertion reads it and knows that no load of this
        // loop is in flight behind it
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      SP_STAMP(2)
      lds_barrier();                                    // ---- X: every posting of the segment is in
      SP_STAMP(3)
#if HX_SP_HARVEST_BATCH
      {
        // chunks beyond the prefetched slots (a dense segment): the postings of the first such round are re-loaded
        // (L2) BEFORE the takes of the prefetched slots, so their latency runs under those
        uint4 r[SP_K];
        uint32_t rm[SP_K];
        auto load_round = [&](uint32_t c0) {
#pragma unroll
          for (int k = 0; k < SP_K; ++k) {
            const uint32_t c = c0 + (uint32_t)(k * SP_WAVES);
            rm[k] = 0;
            r[k] = make_uint4(0, 0, 0, 0);
            if (c < nch) {
              uint32_t off, n;
              int t;
              sp_chunk(d, c, off, n, t);
              rm[k] = sp_lanebits(n, lane);
              r[k] = sp_load2(post, off, n, lane);
            }
          }
        };
        uint32_t c0 = (uint32_t)(SP_K * SP_WAVES + wave);
        if (c0 < nch) load_round(c0);
        uint32_t hm[SP_K];
#pragma unroll
        for (int k = 0; k < SP_K; ++k) hm[k] = (mask >> (2 * k)) & 3u;
        sp_harvest_batch(cur.p, hm, lane, cand, tau, gbase);
        if (c0 < nch) {
          while (c0 < nch) {
            sp_harvest_batch(r, rm, lane, cand, tau, gbase);
            c0 += SP_K * SP_WAVES;
            if (c0 < nch) load_round(c0);
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);          // (as behind the loop of the accumulate phase)
        }
      }
#else
#pragma unroll
      for (int k = 0; k < SP_K; ++k) sp_harvest(cur.p[k], (mask >> (2 * k)) & 3u, cand, tau, gbase, (tpack >> (24 + k)) & 1u);
      for (uint32_t c0 = (uint32_t)(SP_K * SP_WAVES + wave); c0 < nch; c0 += SP_K * SP_WAVES) {
        uint4 r[SP_K];
        uint32_t rm[SP_K];
#pragma unroll
        for (int k = 0; k < SP_K; ++k) {
          const uint32_t c = c0 + (uint32_t)(k * SP_WAVES);
          rm[k] = 0;
          r[k] = make_uint4(0, 0, 0, 0);
          if (c < nch) {
            uint32_t off, n;
            int t;
            sp_chunk(d, c, off, n, t);
            rm[k] = sp_lanebits(n, lane);
            r[k] = sp_load2(post, off, n, lane);
          }
        }
#pragma unroll
        for (int k = 0; k < SP_K; ++k) sp_harvest(r[k], rm[k], cand, tau, gbase, false);
      }
#endif
      SP_STAMP(4)
      lds_barrier();                                    // ---- Y: acc is all zero again
      SP_STAMP(5)
      // the stage's registers stay live to here: handed to the dense-segment loops as destinations of THEIR loads (they
      // were dead behind the takes), the next issue into this stage had to wait for loads of unknown age
#pragma unroll
      for (int k = 0; k < SP_K; ++k) asm volatile("" ::"v"(cur.p[k].x), "v"(cur.p[k].y), "v"(cur.p[k].z), "v"(cur.p[k].w));
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)nx1.nch);
  };
  // `ub` >= S.cnt (every posting of a visit could become a candidate), `trig_r`: the count at which the buffer is
  // cut; the count is read from LDS only when the bound says a cut may be due
  uint32_t ub = 0;
  int trig_r = 2 * a.limit < 256 ? 256 : 2 * a.limit;   // first cut early: it gives the first threshold
  // after a visit: does the buffer have to be cut before the next one?
  auto cut_due = [&](uint32_t appended_max, uint32_t nch_next) {
    const uint32_t bound = nch_next * SP_CH < (uint32_t)SEG_DOCS ? nch_next * SP_CH : (uint32_t)SEG_DOCS;
    ub += appended_max;
    if (ub >= (uint32_t)trig_r || ub + bound > (uint32_t)SP_GCAP) {               // scalar
      const int cnt = __builtin_amdgcn_readfirstlane(S.cnt);       // settled: every wave is past barrier Y
      ub = (uint32_t)cnt;
      return cnt > a.limit && (cnt >= trig_r || (uint32_t)cnt + bound > (uint32_t)SP_GCAP);
    }
    return false;
  };
  int seg = s0;
  fill_window(s0);
  for (;;) {
    // (re)start the pipeline at `seg`: stage 0 = seg, stage 1 = seg + 1
    Stage st0, st1, st2;
    st2.nch = 0;
    st2.mask = 0;
    st2.tpack = 0;
#pragma unroll
    for (int k = 0; k < SP_K; ++k) st2.p[k] = make_uint4(0, 0, 0, 0);
    {
      uint32_t a0, a1, b0, b1;
      offs(seg, a0, a1);
      offs(seg + 1, b0, b1);
      offs(seg + 2, nlo, nhi);
      issue(st0, a0, a1, seg < s1);
      issue(st1, b0, b1, seg + 1 < s1);
    }
    // Visits until the candidate buffer has to be cut, or the range ends.  The body is three visits in a fixed
    // rotation of the stages -- straight-line, so that a stage is the same registers on every trip and the
    // loads in flight are never copied (a stage picked by a branch made hipcc copy all three at the loop
    // header, behind a vmcnt(0): no prefetch at all).
    bool cut = false;
    const int wend = w0 + W1 - 5;                       // last segment whose visit finds seg + 3 (and its end) inside the window
    for (;;) {
      uint32_t am, nn;
      if (seg >= s1 || seg > wend) break;
      nn = visit(st0, st2, st1, seg, am);
      ++seg;
      if ((cut = cut_due(am, nn))) break;
      if (seg >= s1 || seg > wend) break;
      nn = visit(st1, st0, st2, seg, am);
      ++seg;
      if ((cut = cut_due(am, nn))) break;
      if (seg >= s1 || seg > wend) break;
      nn = visit(st2, st1, st0, seg, am);
      ++seg;
      if ((cut = cut_due(am, nn))) break;
    }
    if (!cut && seg < s1) {                             // the window is used up: the next one starts at `seg`
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (loads issued for visits that restart below)
      fill_window(seg);
      continue;
    }
    // cut (acc is all zero between visits; it drains the loads in flight); the last one gives the part's list
    SP_STAMP(0)
    sp_cut(cand, a.limit, M, tid, seg >= s1);
    tau_r = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.tau);
    SP_STAMP(6)
    SP_STAMP_CUT
    if (seg >= s1) break;
    {                                                   // later cuts: when the buffer has grown by a few lists
      const int c = __builtin_amdgcn_readfirstlane(S.cnt);
      const int t = c + (a.cut_step > 0 ? a.cut_step : (4 * a.limit < 1024 ? 1024 : 4 * a.limit));
      trig_r = t < SP_CAP * 3 / 4 ? t : SP_CAP * 3 / 4;
      ub = (uint32_t)c;
    }
  }
  // ---- the part's list: the kept keys (best first)
  const int nk = S.cnt;
  const int n = nk < a.lout ? nk : a.lout;
  for (int i = tid; i < a.lout; i += SP_THREADS) o[i] = i < n ? sp_ld_key(cand + i) : 0ull;
  SP_STAMP(7)
  SP_STAMP_FLUSH
  if (tid == 0) {
    *ocnt = n;
    if (nk > a.lout || S.ovf) a.q_fail[q] = 1;          // cut short: the query takes the exact path
  }
}
#undef S

#if defined(HX_SP_STAMP) && HX_SEG_DOCS == 65536
extern "C" int hx_debug_sp_stamps(unsigned long long* out_host, int n) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sp_stamps), (size_t)n * 8);
}
#endif

void launch_sparse_select_variant(const SparseSelectArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.parts <= 0) return;
  HX_CHECK(a.ix.seg_docs == SEG_DOCS, "sparse: index built for another segment size");
  HX_CHECK(a.limit >= 1 && a.limit <= a.lout && a.lout <= SP_CAP / 2, "sparse: limit too large");
  hipLaunchKernelGGL(k_sparse_select, dim3(a.B * a.parts), dim3(SP_THREADS), 0, st, a);
  HX_HIP(hipGetLastError());
}

}  // namespace HX_SP_VARIANT
}  // namespace hx
