// K7 -- sparse ("BM25") scoring over the on-device inverted index.
//
// Mirrors Prefetch(query=SparseVector, using="sparse", limit=sparse_limit)
// (app/core/vector_store/qdrant/qdrant_handler.py:347-354): score(d) = sum over the
// query's terms of q_t * d_t, IDF-free (the collection sets no sparse modifier, :80-86);
// only documents that share a term with the query are candidates.
//
// Arithmetic (= oracle.OracleIndex.sparse_scores, bit for bit): every product q_t*d_t is
// formed exactly in fp64, scaled by 2^40 and rounded (ties to even) to a 64-bit integer;
// the integers are summed (associative, so the order of the terms cannot matter) and the
// sum is converted once to fp32.  That restates upstream's fp32 running sum
// order-independently (it differs from it by at most a few fp32 ulps) and lets all
// postings of a segment be accumulated concurrently with LDS integer atomics.
//
// Index layout (spbuild.hip): TERM-major.  Postings are sorted by (term, document) as
// {document index inside its segment, fp32 weight}; documents are cut into segments of
// SEG_DOCS, and a dense table gives, for every live term and segment, the offset of the
// term's first posting in that segment or later -- the run of (term, segment) is
// [ptr[t][s], ptr[t][s+1]) and the runs of consecutive segments are adjacent in memory.
//
// One 512-thread workgroup owns (query, part): a contiguous range of segments and a
// 64 KiB LDS accumulator (one marked 64-bit word per document of the segment).  Lane t of
// every 16-lane row holds query term t: its table row, the offsets of the current and the
// next segment.  Per segment ("visit") every wave derives the same chunk list -- a run of
// length len is ceil(len/64) chunks of consecutive postings -- from a DPP row scan of the
// chunk counts, and wave w takes chunks w, w+8 (then w+16, ... for unusually long
// segments): a chunk is wave-uniform, found with one ballot and three v_readlane.
//     accumulate(seg) from registers (LDS integer atomics); derive the chunks of seg+1 from
//     the offsets loaded a visit ago, issue its posting loads and the offsets of seg+3
//                                                                     -- barrier X --
//     harvest(seg): one LDS exchange per posting, the thread that gets the (never zero)
//     marked sum back owns the document
//                                                                     -- barrier Y --
// Both barriers are LDS-only (no vmcnt wait), so the loads of the next visit stay in flight
// across them.  Survivors (score >= the running threshold) go to a per-workgroup buffer in
// global memory that is sorted through the (then all-zero) accumulator and cut to `limit`
// whenever it has grown by a few times `limit`.  Queries with more than 16 terms take the
// same steps per group of 16 terms without the prefetch.
#include <type_traits>
#include "hx_common.hpp"
#include "kernels.hpp"
#include "wsort.hpp"

// Compiled twice (rag_application_amd/build.py): -DHX_SP_VARIANT=v8k -DHX_SEG_DOCS=8192
// -DHX_SP_THREADS=512 and -DHX_SP_VARIANT=v16k -DHX_SEG_DOCS=16384 -DHX_SP_THREADS=1024.
#ifndef HX_SP_VARIANT
#define HX_SP_VARIANT v8k
#endif
#ifndef HX_SEG_DOCS
#define HX_SEG_DOCS 8192
#endif

namespace hx {
namespace HX_SP_VARIANT {

constexpr int SEG_DOCS = HX_SEG_DOCS;   // docs per index segment (LDS accumulator: 8 B per doc)
constexpr int SP_CAND = SEG_DOCS;       // per-workgroup candidate buffer (keys, global memory)

// Diagnostic build only (-DHX_SP_STAMP): lane 0 of waves 0 and 3 accumulate s_memtime deltas per
// phase into a debug buffer of its own (never read by the kernel, never in a timed build).
#ifdef HX_SP_STAMP
__device__ unsigned long long g_sp_stamps[2 * 8 * 4096];
#define SP_STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SP_STAMP(i) { const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; }
#define SP_STAMP_FLUSH if ((tid == 0 || tid == 192) && blockIdx.x < 4096) for (int i_ = 0; i_ < 8; ++i_) g_sp_stamps[(blockIdx.x * 2 + (tid != 0)) * 8 + i_] = st_acc[i_];
#else
#define SP_STAMP_DECL
#define SP_STAMP(i)
#define SP_STAMP_FLUSH
#endif

#ifndef HX_SP_THREADS
#define HX_SP_THREADS 512
#endif
constexpr int SP_THREADS = HX_SP_THREADS;
constexpr int SP_WAVES = SP_THREADS / 64;
constexpr int SP_TG = 16;            // query terms per group (one 16-lane DPP row)
#ifndef HX_SP_K
#define HX_SP_K 2
#endif
constexpr int SP_K = HX_SP_K;        // pipelined path: chunks per wave held in registers per segment
constexpr int SP_TCH = 64;           // k_sparse_order: term-count buckets
constexpr int SP_TCACHE = 256;       // query terms whose table row is resolved once per workgroup
constexpr double SP_FIX = 1099511627776.0;          // 2^40
constexpr float SP_UNFIX = 9.094947017729282e-13f;  // 2^-40 (exact in fp32)

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
    unsigned long long acc[SEG_DOCS];         // marked fixed-point score per document of the segment
    uint64_t sort[SEG_DOCS];                 // sort scratch while acc is all zero
  };
  int cnt;                                   // candidates in the workgroup's global buffer
  int trig;                                  // cnt at which the buffer is sorted and cut
  float tau;
  int ti[SP_TCACHE];                         // live-term index of the query's first SP_TCACHE terms
};
// One object at namespace scope: every access is provably LDS (ds_* instructions).  Passed
// around by reference, hipcc fell back to FLAT addressing for part of the accesses, and a
// flat access waits for vmcnt(0) too -- which drained the prefetch at every table read.
__shared__ SpShared g_sp;
#define S g_sp

__device__ __forceinline__ uint64_t sp_ld_key(const uint64_t* p) {
  return __hip_atomic_load((const __attribute__((address_space(1))) uint64_t*)p, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sp_st_key(uint64_t* p, uint64_t v) {
  __hip_atomic_store((__attribute__((address_space(1))) uint64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sort the workgroup's candidate buffer (global) through LDS, keep `limit`, raise tau.
// Precondition: acc is all zero and every wave is past its last acc access.
__device__ __forceinline__ void sp_sort_truncate(uint64_t* cand, int limit, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends have left the core
  __syncthreads();                                   // ... and every other wave's; S.cnt settled
  const int n = S.cnt;
  if (limit <= 256 && n <= SP_WAVES * 256) {         // block-uniform; the usual case
    // every wave sorts 256 keys in registers (wsort.hpp), then log2(SP_WAVES) pairwise folds through
    // the sort scratch keep the best 256: one barrier per fold instead of one per bitonic stage
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
      lds_barrier();   // (the first one also orders every wave's loads of cand before wave 0's stores)
      const int pw = w + (1 << s);
      if ((w & m) == 0 && n > pw * 256) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k64max(v[e], S.sort[pw * 256 + 255 - (lane * 4 + e)]);
        w_merge<256, 128>(v, lane);
      }
    }
    if (w == 0) {
      const int keep = n < limit ? n : limit;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (lane * 4 + e < keep) sp_st_key(cand + lane * 4 + e, v[e]);
      if (n >= limit) {
        const int kr = limit - 1;
        const uint64_t mine = (kr & 3) == 0 ? v[0] : ((kr & 3) == 1 ? v[1] : ((kr & 3) == 2 ? v[2] : v[3]));
        const uint64_t kth = (uint64_t)__shfl((unsigned long long)mine, kr >> 2, 64);
        if (lane == 0) {
          S.tau = key_score(kth);
          S.cnt = limit;
        }
      }
    }
    lds_barrier();     // every fold has read its partner's slice
    if (w != 0) {      // acc back to zero: each wave but 0 wrote its slice exactly once
#pragma unroll
      for (int e = 0; e < 4; ++e) S.sort[w * 256 + lane * 4 + e] = 0ull;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return;
  }
  int P = SP_THREADS;                                // sort size: next power of two >= n
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += SP_THREADS) S.sort[i] = i < n ? sp_ld_key(cand + i) : 0ull;
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
  const int keep = n < limit ? n : limit;
  for (int i = tid; i < keep; i += SP_THREADS) sp_st_key(cand + i, S.sort[i]);
  if (tid == 0 && n >= limit) {
    S.tau = key_score(S.sort[limit - 1]);
    S.cnt = limit;
  }
  lds_barrier();
  for (int i = tid; i < P; i += SP_THREADS) S.sort[i] = 0ull;   // acc back to zero
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ void sp_append(uint64_t* cand, float tau, unsigned long long fx, int64_t gid) {
  const float s = sp_unfix(fx);
  if (s >= tau) {
    const int pos = atomicAdd(&S.cnt, 1);
    sp_st_key(cand + pos, make_key(s, (uint32_t)gid));
  }
}

// Harvest by a linear sweep of the accumulator (cold paths): thread tid owns documents
// tid + 256*e, so a wave reads 512 contiguous bytes per instruction.  e in [E0, E1).
template <int E0, int E1>
__device__ __forceinline__ void sp_harvest_sweep(uint64_t* cand, float tau, int64_t gbase, int tid) {
  uint32_t m = 0;
#pragma unroll
  for (int e = E0; e < E1; ++e) m |= (S.acc[tid + e * SP_THREADS] != 0ull ? 1u : 0u) << e;
  while (m) {   // a wave loops max-popcount times (touched entries are sparse), not E times
    const int e = __builtin_ctz(m);
    m &= m - 1;
    const int d = tid + e * SP_THREADS;
    const unsigned long long v = S.acc[d];
    S.acc[d] = 0ull;
    sp_append(cand, tau, v, gbase + d);
  }
}

// Sweep-harvest a whole segment.  Precondition (sp_make_room): cnt + SEG_DOCS/2 <= SP_CAND.
__device__ __forceinline__ void sp_harvest(const SparseQueryArgs& a, uint64_t* cand, unsigned long long* park, int seg,
                                           uint32_t total_len, int tid) {
  constexpr int E = SEG_DOCS / SP_THREADS;
  const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
  const uint32_t bound = total_len < (uint32_t)SEG_DOCS ? total_len : (uint32_t)SEG_DOCS;
  if ((uint32_t)S.cnt + bound <= (uint32_t)SP_CAND) {   // block-uniform
    sp_harvest_sweep<0, E>(cand, S.tau, gbase, tid);
  } else {
    // a segment can yield up to SEG_DOCS survivors: harvest the lower half, cut the buffer
    // to `limit` (the upper half of the accumulator is parked in global memory while the
    // sort borrows the LDS), then harvest the upper half.
    constexpr int H = SEG_DOCS / 2;
    sp_harvest_sweep<0, E / 2>(cand, S.tau, gbase, tid);
    lds_barrier();
    for (int i = tid; i < H; i += SP_THREADS) {
      sp_st_key((uint64_t*)park + i, S.acc[H + i]);
      S.acc[H + i] = 0ull;
    }
    lds_barrier();
    sp_sort_truncate(cand, a.limit, tid);
    for (int i = tid; i < H; i += SP_THREADS) S.acc[H + i] = sp_ld_key((const uint64_t*)park + i);
    lds_barrier();
    sp_harvest_sweep<E / 2, E>(cand, S.tau, gbase, tid);
  }
  lds_barrier();
}

// Before a segment is accumulated (acc all zero): cut the candidate buffer when it has
// grown past the trigger (cheap, early sorts raise tau quickly) or could not take the
// segment's survivors.  `cnt` is wave-uniform (readfirstlane) so the branch is scalar.
__device__ __forceinline__ void sp_make_room(const SparseQueryArgs& a, uint64_t* cand, uint32_t total_len, int tid) {
  const uint32_t bound = total_len < (uint32_t)SEG_DOCS ? total_len : (uint32_t)SEG_DOCS;
  const int cnt = __builtin_amdgcn_readfirstlane(S.cnt);
  const int trig = __builtin_amdgcn_readfirstlane(S.trig);
  if (cnt > a.limit && (cnt >= trig || (uint32_t)cnt + bound > (uint32_t)SP_CAND)) {
    sp_sort_truncate(cand, a.limit, tid);
    if (tid == 0) {   // later cuts: when the buffer holds a few times what survives
      const int t = 4 * a.limit;
      S.trig = t < 512 ? 512 : t;
    }
  }
}

__device__ __forceinline__ void sp_finish(const SparseQueryArgs& a, uint64_t* cand, int q, int part, int tid) {
  sp_sort_truncate(cand, a.limit, tid);
  const int n = S.cnt < a.limit ? S.cnt : a.limit;
  uint64_t* o = a.out + ((int64_t)q * a.parts + part) * a.limit;
  for (int i = tid; i < a.limit; i += SP_THREADS) o[i] = i < n ? sp_ld_key(cand + i) : 0ull;
  if (tid == 0) a.out_cnt[q * a.parts + part] = n;
}

// ---------------------------------------------------------------------------------
// directory: per-lane run offsets -> wave-uniform chunks
// ---------------------------------------------------------------------------------
struct SpDir {            // lane (l & 15) = term slot of the group; all four rows hold the same values
  uint32_t p0, len, incl;  // first posting / postings / inclusive chunk count up to this term
};
// inclusive scan over the 16 lanes of a row (DPP row shifts, zeros shifted in)
__device__ __forceinline__ uint32_t sp_rowscan(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  return v;
}
// total: postings of the segment -- exact (EXACT) or the upper bound 64 * nch
template <bool EXACT = true>
__device__ __forceinline__ SpDir sp_dir(uint32_t p0, uint32_t p1, bool active, uint32_t& nch, uint32_t& total) {
  SpDir d;
  d.p0 = p0;
  d.len = active ? p1 - p0 : 0u;
  d.incl = sp_rowscan((d.len + 63u) >> 6);
  nch = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, 15);
  if (EXACT) total = (uint32_t)__builtin_amdgcn_readlane((int)sp_rowscan(d.len), 15);
  else total = nch << 6;
  return d;
}
// chunk c of the segment: first posting, postings in it (1..64), query weight.  c < nch.
__device__ __forceinline__ void sp_chunk(const SpDir& d, float qw_lane, uint32_t c, uint32_t& off, uint32_t& cnt,
                                         float& qw) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_ballot_w64(d.incl > c) & 0xFFFFu;
  const int t = __builtin_ctz(m);                       // m != 0 because c < nch = incl[15]
  const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)d.len, t);
  const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)d.incl, t) - ((len + 63u) >> 6);
  const uint32_t j = (c - start) << 6;
  off = (uint32_t)__builtin_amdgcn_readlane((int)d.p0, t) + j;
  cnt = len - j < 64u ? len - j : 64u;
  qw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qw_lane), t));
}

// live-term index of `term` (binary search in the ascending list), -1 if absent
__device__ __forceinline__ int sp_find_term(const SparseIndexView& ix, uint32_t term) {
  int lo = 0, hi = ix.n_live;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ix.uterms[mid] < term) lo = mid + 1; else hi = mid;
  }
  return (lo < ix.n_live && ix.uterms[lo] == term) ? lo : -1;
}

// The same lookup by one wave in 64-ary steps: ~log64(n_live) dependent loads instead of log2 (a
// workgroup resolves its query's terms before it can start; `term` is wave-uniform).
__device__ __forceinline__ int sp_find_term_wave(const SparseIndexView& ix, uint32_t term, int lane) {
  int lo = 0, hi = ix.n_live;   // the first index whose term is >= `term` lies in [lo, hi]
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + lane * step;
    const bool in = p < hi;
    const uint32_t v = in ? ix.uterms[p] : 0xFFFFFFFFu;
    const int c = __popcll(__ballot(in && v < term));   // ascending list: a prefix of the probes
    if (c == 0) {
      hi = lo + 1;
      break;
    }
    const int nhi = lo + c * step + 1;                  // probe c (if any) is >= term
    lo = lo + (c - 1) * step + 1;                       // probe c - 1 is < term
    hi = nhi < hi ? nhi : hi;
  }
  const int p = lo + lane;
  const bool in = p < hi;
  const uint32_t v = in ? ix.uterms[p] : 0xFFFFFFFFu;
  const unsigned long long eq = __ballot(in && v == term);
  return eq ? lo + (int)__builtin_ctzll(eq) : -1;
}

// chunks [c0, nch) of a segment by stride SP_WAVES, straight from memory (no prefetch):
// accumulate (HARVEST = false) or exchange-harvest (HARVEST = true)
template <bool HARVEST>
__device__ __forceinline__ void sp_chunks_direct(const SparseQueryArgs& a, const SpDir& d, float qw_lane, uint32_t c0,
                                                 uint32_t nch, int lane, uint64_t* cand, float tau, int64_t gbase) {
  for (uint32_t c = c0; c < nch; c += SP_WAVES) {
    uint32_t off, cnt;
    float qw;
    sp_chunk(d, qw_lane, c, off, cnt, qw);
    if ((uint32_t)lane < cnt) {
      const uint2 p = a.ix.post[off + lane];
      if (!HARVEST) {
        atomicAdd(&S.acc[p.x], sp_fix(qw, __builtin_bit_cast(float, p.y)));
      } else {
        const unsigned long long x = atomicExch(&S.acc[p.x], 0ull);
        if (x != 0ull) sp_append(cand, tau, x, gbase + p.x);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// pipelined path: T <= SP_TG
// ---------------------------------------------------------------------------------
// One pipeline stage = everything a wave holds for one future visit.
struct SpStage {
  float q[SP_K];            // query weight of the chunk's term
  uint32_t cnt[SP_K];       // scalar: postings in slot k (lane l holds one iff l < cnt[k])
  SpDir d;
  uint32_t nch, total;      // scalar: chunks / postings of the segment
};
constexpr int SP_D = 3;                         // visits a load is in flight
constexpr int SP_LPV = SP_K + 2;                // asm loads per visit per wave: 2 offsets + SP_K postings
// The loads in flight live in VGPRs hipcc does not allocate: the kernel is built with
// amdgpu_num_vgpr(SP_NVGPR) and stage j owns the next 2*SP_K + 2 registers -- named only inside asm
// strings (and their clobber lists, so the kernel's register count covers them).  hipcc
// cannot see a pending value, so it cannot copy one before it has landed (it did: the
// loop-carried copies it inserted for asm OUTPUT operands read registers ahead of the wait).
// sp_stage_issue: table offsets first, then the postings -- SP_LPV loads in a fixed order.
// sp_stage_collect: one counted wait (everything but the SP_D - 1 younger visits' loads has
// landed), then the values move to ordinary registers.
static_assert(SP_D == 3, "register map below");
#if HX_SP_K == 2
// default: 2 chunk slots per wave, 4 waves per SIMD (two 512-thread workgroups of 64 KiB per CU)
#define SP_NVGPR 104
#define SP_WPE 4
#define SP_STAGE_FUNCS(J, R0, R1, R2, R3, R4, R5)                                                          \
  __device__ __forceinline__ void sp_stage_issue##J(const uint32_t* pl, const uint32_t* ph, const uint2* const* pp) { \
    asm volatile("global_load_dword v" #R4 ", %0, off\n\t"                                                 \
                 "global_load_dword v" #R5 ", %1, off\n\t"                                                 \
                 "global_load_dwordx2 v[" #R0 ":" #R1 "], %2, off\n\t"                                     \
                 "global_load_dwordx2 v[" #R2 ":" #R3 "], %3, off"                                         \
                 :                                                                                         \
                 : "v"(pl), "v"(ph), "v"(pp[0]), "v"(pp[1])                                                \
                 : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3, "v" #R4, "v" #R5);                        \
  }                                                                                                        \
  __device__ __forceinline__ void sp_stage_collect##J(uint32_t (&doc)[SP_K], float (&w)[SP_K], uint32_t& o_lo, \
                                                      uint32_t& o_hi) {                                    \
    asm volatile("s_waitcnt vmcnt(%6)\n\t"                                                                 \
                 "v_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R2 "\n\t"          \
                 "v_mov_b32 %3, v" #R3 "\n\tv_mov_b32 %4, v" #R4 "\n\tv_mov_b32 %5, v" #R5                 \
                 : "=v"(doc[0]), "=v"(w[0]), "=v"(doc[1]), "=v"(w[1]), "=v"(o_lo), "=v"(o_hi)              \
                 : "n"(SP_LPV * (SP_D - 1))                                                                \
                 : "memory");                                                                              \
  }
SP_STAGE_FUNCS(0, 104, 105, 106, 107, 108, 109)
SP_STAGE_FUNCS(1, 110, 111, 112, 113, 114, 115)
SP_STAGE_FUNCS(2, 116, 117, 118, 119, 120, 121)
#undef SP_STAGE_FUNCS
#else
// experiment: 1 chunk slot per wave, 6 waves per SIMD (three 512-thread workgroups; needs
// HX_SEG_DOCS = 4096 so that three accumulators fit the LDS)
static_assert(HX_SP_K == 1, "HX_SP_K is 1 or 2");
#define SP_NVGPR 68
#define SP_WPE 6
#define SP_STAGE_FUNCS(J, R0, R1, R4, R5)                                                                  \
  __device__ __forceinline__ void sp_stage_issue##J(const uint32_t* pl, const uint32_t* ph, const uint2* const* pp) { \
    asm volatile("global_load_dword v" #R4 ", %0, off\n\t"                                                 \
                 "global_load_dword v" #R5 ", %1, off\n\t"                                                 \
                 "global_load_dwordx2 v[" #R0 ":" #R1 "], %2, off"                                         \
                 :                                                                                         \
                 : "v"(pl), "v"(ph), "v"(pp[0])                                                            \
                 : "memory", "v" #R0, "v" #R1, "v" #R4, "v" #R5);                                          \
  }                                                                                                        \
  __device__ __forceinline__ void sp_stage_collect##J(uint32_t (&doc)[SP_K], float (&w)[SP_K], uint32_t& o_lo, \
                                                      uint32_t& o_hi) {                                    \
    asm volatile("s_waitcnt vmcnt(%4)\n\t"                                                                 \
                 "v_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R4 "\n\t"          \
                 "v_mov_b32 %3, v" #R5                                                                     \
                 : "=v"(doc[0]), "=v"(w[0]), "=v"(o_lo), "=v"(o_hi)                                        \
                 : "n"(SP_LPV * (SP_D - 1))                                                                \
                 : "memory");                                                                              \
  }
SP_STAGE_FUNCS(0, 68, 69, 70, 71)
SP_STAGE_FUNCS(1, 72, 73, 74, 75)
SP_STAGE_FUNCS(2, 76, 77, 78, 79)
#undef SP_STAGE_FUNCS
#endif

__device__ __forceinline__ void sp_body_pipe(const SparseQueryArgs& a, uint64_t* cand, unsigned long long* park, int q,
                                             int part, int s0, int s1, int64_t qb, int T, int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ts = lane & 15;
  // term slot: table row and query weight (absent terms and unused slots are inactive)
  int ti = -1;
  float qw_lane = 0.f;
  if (ts < T) {
    ti = S.ti[ts];
    qw_lane = a.q_val[qb + ts];
  }
  const bool active = ti >= 0;
  const uint32_t* row = a.ix.ptr + (int64_t)(active ? ti : 0) * (a.ix.n_segments + 1);
  const int last = a.ix.n_segments;     // row[last] is the end of the term's postings
  auto clampi = [&](int x) { return x <= last ? x : last; };

  // issue the loads of stage J for segment `sx`: offsets of sx + SP_D and sx + SP_D + 1, then the
  // postings of sx's chunks (directory st.d must be set).  Exactly SP_LPV loads, whatever the
  // segment holds (a missing chunk loads posting 0).
  auto issue = [&](auto jc, SpStage& st, int sx) {
    constexpr int J = decltype(jc)::value;
    const uint2* pp[SP_K];
#pragma unroll
    for (int k = 0; k < SP_K; ++k) {
      const uint32_t c = (uint32_t)(k * SP_WAVES + wave);
      uint32_t off = 0, cnt = 0;
      st.q[k] = 0.f;
      if (c < st.nch) sp_chunk(st.d, qw_lane, c, off, cnt, st.q[k]);   // wave-uniform
      pp[k] = a.ix.post + off + ((uint32_t)lane < cnt ? lane : 0);
      st.cnt[k] = cnt;
    }
    const uint32_t* pl = row + clampi(sx + SP_D);
    const uint32_t* ph = row + clampi(sx + SP_D + 1);
    if constexpr (J == 0) sp_stage_issue0(pl, ph, pp);
    if constexpr (J == 1) sp_stage_issue1(pl, ph, pp);
    if constexpr (J == 2) sp_stage_issue2(pl, ph, pp);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;

  SpStage st0, st1, st2;
  {
    // prologue: directories of the first SP_D segments from plain loads (hipcc waits for them)
    const uint32_t o0 = row[clampi(s0)], o1 = row[clampi(s0 + 1)], o2 = row[clampi(s0 + 2)], o3 = row[clampi(s0 + 3)];
    st0.d = sp_dir<false>(o0, o1, active, st0.nch, st0.total);
    st1.d = sp_dir<false>(o1, o2, active, st1.nch, st1.total);
    st2.d = sp_dir<false>(o2, o3, active, st2.nch, st2.total);
    issue(J0{}, st0, s0);
    issue(J1{}, st1, s0 + 1);
    issue(J2{}, st2, s0 + 2);
  }
  // The candidate count and the threshold live in LDS (S.cnt, S.tau); a visit works from scalar
  // copies: `ub` >= S.cnt (every posting of a visit could become a candidate) and `tau_r` <=
  // S.tau (a stale threshold only lets more candidates through).  They are refreshed where the
  // buffer may have to be cut: the first 16 visits (early cuts give the first threshold), then every
  // 8th, and whenever `ub` says the visit might not fit.
  uint32_t ub = 0;
  float tau_r = -__builtin_inff();
  int nvis = 0, next_chk = 0;         // non-empty visits so far / the one that refreshes ub and tau_r next
  SP_STAMP_DECL

  auto visit = [&](auto jc, SpStage& st, int seg) {
    constexpr int J = decltype(jc)::value;
    uint32_t doc[SP_K], o_lo, o_hi;
    float w[SP_K];
    // this stage's loads (issued SP_D visits ago) have landed
    if constexpr (J == 0) sp_stage_collect0(doc, w, o_lo, o_hi);
    if constexpr (J == 1) sp_stage_collect1(doc, w, o_lo, o_hi);
    if constexpr (J == 2) sp_stage_collect2(doc, w, o_lo, o_hi);
    SP_STAMP(0)
    const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
    const uint32_t total = __builtin_amdgcn_readfirstlane(st.total), nch = __builtin_amdgcn_readfirstlane(st.nch);
    const SpDir d = st.d;
    uint32_t pc[SP_K];                                              // scalar; the refill below overwrites st
#pragma unroll
    for (int k = 0; k < SP_K; ++k) pc[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.cnt[k]);
    const bool tails = nch > (uint32_t)(SP_K * SP_WAVES);          // scalar
    const uint32_t bound = total < (uint32_t)SEG_DOCS ? total : (uint32_t)SEG_DOCS;
    bool fast = true;
    if (total) {   // scalar
      if (nvis >= next_chk || ub + bound > (uint32_t)SP_CAND) {   // scalar
        sp_make_room(a, cand, total, tid);
        ub = (uint32_t)__builtin_amdgcn_readfirstlane(S.cnt);
        tau_r = S.tau;
        next_chk = nvis + (nvis < 16 ? 1 : 8);
      }
      ++nvis;
      fast = ub + bound <= (uint32_t)SP_CAND;
      ub += bound;
#pragma unroll
      for (int k = 0; k < SP_K; ++k)
        if ((uint32_t)lane < pc[k]) atomicAdd(&S.acc[doc[k]], sp_fix(st.q[k], w[k]));
      if (tails) sp_chunks_direct<false>(a, d, qw_lane, SP_K * SP_WAVES + wave, nch, lane, cand, 0.f, gbase);
    }
    SP_STAMP(1)
    // refill this stage for seg + SP_D
    st.d = sp_dir<false>(o_lo, o_hi, active, st.nch, st.total);
    issue(jc, st, seg + SP_D);
    SP_STAMP(2)
    lds_barrier();                                   // ---- X
    SP_STAMP(3)
    if (total) {
      if (!fast) {                                              // scalar
        sp_harvest(a, cand, park, seg, total, tid);             // sweep; ends with a barrier
        ub = (uint32_t)__builtin_amdgcn_readfirstlane(S.cnt);
        tau_r = S.tau;
      } else {
        unsigned long long v[SP_K];
#pragma unroll
        for (int k = 0; k < SP_K; ++k) {
          v[k] = 0ull;
          if ((uint32_t)lane < pc[k]) v[k] = atomicExch(&S.acc[doc[k]], 0ull);
        }
#pragma unroll
        for (int k = 0; k < SP_K; ++k)
          if (v[k] != 0ull) sp_append(cand, tau_r, v[k], gbase + doc[k]);
        if (tails) sp_chunks_direct<true>(a, d, qw_lane, SP_K * SP_WAVES + wave, nch, lane, cand, tau_r, gbase);
        SP_STAMP(4)
        lds_barrier();                               // ---- Y
        SP_STAMP(5)
      }
    }
  };
  int seg = s0;
  for (;;) {
    if (seg >= s1) break;
    visit(J0{}, st0, seg);
    if (++seg >= s1) break;
    visit(J1{}, st1, seg);
    if (++seg >= s1) break;
    visit(J2{}, st2, seg);
    ++seg;
  }
  SP_STAMP_FLUSH
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid < 16 && a.stat_postings && active) {   // postings of this lane's term in [s0, s1): adjacent runs
    const uint32_t np = row[clampi(s1)] - row[clampi(s0)];
    if (np) atomicAdd(a.stat_postings, (unsigned long long)np);
  }
  sp_finish(a, cand, q, part, tid);
}

// ---------------------------------------------------------------------------------
// kernel: pipelined body for queries of up to 16 terms, grouped loop otherwise
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS, SP_WPE) __attribute__((amdgpu_num_vgpr(SP_NVGPR))) void k_sparse_score(SparseQueryArgs a) {
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
    S.trig = 2 * a.limit < 256 ? 256 : 2 * a.limit;   // first cut early: it gives the first threshold
    S.tau = -__builtin_inff();
  }
  for (int i = tid >> 6; i < T && i < SP_TCACHE; i += SP_WAVES) {   // one wave per term
    const int r = sp_find_term_wave(a.ix, (uint32_t)a.q_idx[qb + i], tid & 63);
    if ((tid & 63) == 0) S.ti[i] = r;
  }
  __syncthreads();
  if (T <= SP_TG && T > 0 && s0 < s1) {   // block-uniform
    sp_body_pipe(a, cand, park, q, part, s0, s1, qb, T, tid);
    return;
  }
  // grouped loop: 16 terms at a time, no prefetch
  const int lane = tid & 63, wave = tid >> 6, ts = lane & 15;
  unsigned long long npost = 0;
  for (int seg = s0; seg < s1 && T > 0; ++seg) {
    const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
    uint32_t seg_total = 0;
    sp_make_room(a, cand, SEG_DOCS, tid);   // acc is all zero between segments
    for (int pass = 0; pass < 2; ++pass) {  // 0: accumulate every group, 1: harvest every group
      if (pass == 1) {
        lds_barrier();
        if (seg_total == 0) break;
        if ((uint32_t)__builtin_amdgcn_readfirstlane(S.cnt) + (seg_total < (uint32_t)SEG_DOCS ? seg_total : (uint32_t)SEG_DOCS) >
            (uint32_t)SP_CAND) {
          sp_harvest(a, cand, park, seg, seg_total, tid);
          break;
        }
      }
      const float tau = S.tau;
      for (int g0 = 0; g0 < T; g0 += SP_TG) {
        int ti = -1;
        float qw_lane = 0.f;
        if (g0 + ts < T) {
          ti = g0 + ts < SP_TCACHE ? S.ti[g0 + ts] : sp_find_term(a.ix, (uint32_t)a.q_idx[qb + g0 + ts]);
          qw_lane = a.q_val[qb + g0 + ts];
        }
        const bool active = ti >= 0;
        const uint32_t* row = a.ix.ptr + (int64_t)(active ? ti : 0) * (nseg + 1);
        uint32_t nch, total;
        const SpDir d = sp_dir(row[seg], row[seg + 1], active, nch, total);
        if (pass == 0) {
          seg_total += total;
          sp_chunks_direct<false>(a, d, qw_lane, (uint32_t)wave, nch, lane, cand, 0.f, gbase);
        } else {
          sp_chunks_direct<true>(a, d, qw_lane, (uint32_t)wave, nch, lane, cand, tau, gbase);
        }
      }
      if (pass == 1) lds_barrier();
    }
    npost += seg_total;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0 && a.stat_postings) atomicAdd(a.stat_postings, npost);
  sp_finish(a, cand, q, part, tid);
}
#undef S

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

#if defined(HX_SP_STAMP) && HX_SEG_DOCS == 8192
extern "C" int hx_debug_sp_stamps(unsigned long long* out_host, int n) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sp_stamps), (size_t)n * 8);
}
#endif

void launch_sparse_score_variant(const SparseQueryArgs& a, hipStream_t st) {
  if (a.B <= 0) return;
  HX_CHECK(a.ix.seg_docs == SEG_DOCS, "sparse: index built for another segment size");
  HX_CHECK(a.limit * 2 <= SP_CAND, "sparse: limit too large");
  if (a.q_order) {
    hipLaunchKernelGGL(k_sparse_order, dim3(1), dim3(1024), 0, st, a.q_indptr, a.B, a.q_order);
    HX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_sparse_score, dim3(a.B * a.parts), dim3(SP_THREADS), 0, st, a);
  HX_HIP(hipGetLastError());
}

}  // namespace HX_SP_VARIANT
}  // namespace hx
