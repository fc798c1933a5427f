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
// Index layout (spbuild.hip): documents are cut into segments of SEG_DOCS; inside a
// segment postings are sorted by (term, doc) as {u16 doc_local, f32 weight}; an
// open-addressing table maps (segment, term) -> (offset, length).
//
// One 256-thread workgroup owns (query, part): a contiguous range of segments and a
// 64 KiB LDS accumulator (one marked 64-bit word per document of the segment).  The fast
// path (<= SP_TMAX query terms) is software-pipelined across segments with two barriers
// per segment; wave 0 runs the directory while waves 1-3 carry the postings:
//     accumulate(seg) from registers; wave 0 publishes the chunk table of seg+2 from
//     the directory probes it issued a visit ago and issues the probes of seg+3
//                                                                     -- barrier X --
//     issue the posting loads of seg+2; harvest(seg): one LDS exchange per posting,
//     the thread that gets the (never zero) marked sum back owns the document
//                                                                     -- barrier Y --
// Prefetch loads are issued from inline asm and retired by ONE counted wait per visit
// whose operands are the registers being consumed (see sp_wait_slots): hipcc does not
// track them, so it can neither pull a wait up to the load nor drain the queue early.
// Survivors (score >= the running threshold) go to a per-workgroup buffer in global
// memory that is sorted through the (then all-zero) accumulator and cut to `limit`
// whenever it has grown by a few times `limit`.
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

constexpr int SP_THREADS = 256;
constexpr int SP_WAVES = SP_THREADS / 64;
constexpr int SP_TCH = 64;           // generic path: query terms looked up per round
constexpr int SP_TMAX = 12;          // pipelined path: max query terms
constexpr int SP_PW = SP_WAVES - 1;   // pipelined path: posting waves (wave 0 runs the directory instead)
constexpr int SP_K = 8;              // pipelined path: 64-posting chunks per posting wave held in registers per segment
constexpr int SP_NCH = 256;          // pipelined path: chunk-table capacity per segment
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

struct SpChunk {
  uint32_t off;        // first posting of the chunk
  uint32_t rem_term;   // postings in the chunk (1..64) | query-term index << 8
};
struct SpShared {
  union {
    unsigned long long acc[SEG_DOCS];        // marked fixed-point score per document of the segment
    uint64_t sort[SEG_DOCS];                 // sort scratch while acc is all zero
  };
  uint32_t t_off[SP_TCH], t_len[SP_TCH];     // generic path: directory of the current segment
  float t_w[SP_TCH];
  SpChunk chunk[3][SP_NCH];                  // pipelined path: chunk tables, ring of 3 segments
  uint32_t nchunks[3], total[3];
  int cnt;                                   // candidates in the workgroup's global buffer
  int trig;                                  // cnt at which the buffer is sorted and cut
  float tau;
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
// directory probes (inline-asm loads: untracked by hipcc, retired by counted waits)
// ---------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Probe {
  u32x4 r0, r1;   // the two 16-byte table slots fetched speculatively: {key lo, key hi, off, len}
};
__device__ __forceinline__ Probe sp_probe_issue(const SparseQueryArgs& a, int seg, uint32_t term) {
  Probe p;
  const uint64_t key = ((uint64_t)seg << 31) | term;
  const uint64_t slot = sp_hash(key) & a.ix.table_mask;
  const SpHashEntry* p0 = a.ix.table + slot;
  const SpHashEntry* p1 = a.ix.table + ((slot + 1) & a.ix.table_mask);
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(p.r0) : "v"(p0) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(p.r1) : "v"(p1) : "memory");
  return p;
}
// wait for EVERYTHING this wave has in flight; the probe registers are operands of the wait
__device__ __forceinline__ void sp_probe_wait(Probe& p) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.r0), "+v"(p.r1)::"memory");
}
// the probe must have landed (sp_probe_wait or sp_wait_slots)
__device__ __forceinline__ void sp_probe_resolve(const SparseQueryArgs& a, const Probe& p, int seg, uint32_t term,
                                                 uint32_t& off, uint32_t& len) {
  const uint64_t key = ((uint64_t)seg << 31) | term;
  const uint64_t k0 = ((uint64_t)p.r0.y << 32) | p.r0.x, k1 = ((uint64_t)p.r1.y << 32) | p.r1.x;
  off = 0;
  len = 0;
  if (k0 == key) {
    off = p.r0.z;
    len = p.r0.w;
  } else if (k0 == ~0ull) {
  } else if (k1 == key) {
    off = p.r1.z;
    len = p.r1.w;
  } else if (k1 == ~0ull) {
  } else {
    uint64_t slot = ((sp_hash(key) & a.ix.table_mask) + 2) & a.ix.table_mask;   // rare: longer chain
    while (true) {
      u32x4 r;
      const SpHashEntry* pe = a.ix.table + slot;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(pe) : "memory");
      const uint64_t k = ((uint64_t)r.y << 32) | r.x;
      if (k == key) {
        off = r.z;
        len = r.w;
        break;
      }
      if (k == ~0ull) break;
      slot = (slot + 1) & a.ix.table_mask;
    }
  }
}

// ---------------------------------------------------------------------------------
// pipelined path: T <= SP_TMAX
// ---------------------------------------------------------------------------------
// The postings a segment holds for the query are cut into chunks of <= 64 consecutive
// postings of ONE run (a run of length len gives ceil(len/64) chunks).  Posting wave w takes
// chunks w-1, w-1 + 3, w-1 + 6, ...: a chunk is wave-uniform, so a lane's posting is simply off + lane
// and all waves carry the same load whatever the run lengths.
struct SpSlots {            // one segment's postings held by this thread (SP_K chunks of its wave)
  uint32_t d[SP_K];         // doc_local exactly as loaded: NOTHING may touch a loaded value before
  float w[SP_K];            //   the counted wait of the visit that consumes it
  float q[SP_K];            // query weight of the chunk's term (from LDS)
  uint32_t valid;           // bit k: slot k holds a posting (from the chunk table, not from the data)
  uint32_t total, nch;      // block-uniform: postings / chunks of the segment
};

// Every group is EXACTLY 2*SP_K loads per wave (missing chunks load posting 0), so
// `s_waitcnt vmcnt(2*SP_K)` retires everything older than the youngest group.
__device__ __forceinline__ void sp_load_slots(const SparseQueryArgs& a, int ring, int wave, int lane, SpSlots& R) {
  R.total = S.total[ring];
  R.nch = S.nchunks[ring];
  R.valid = 0;
  if (wave == 0) return;   // the directory wave holds no postings (and issues no posting loads)
#pragma unroll
  for (int k = 0; k < SP_K; ++k) {
    const uint32_t c = (uint32_t)(k * SP_PW + wave - 1);
    uint32_t i = 0;
    float q = 0.f;
    if (c < R.nch && c < (uint32_t)SP_NCH) {   // wave-uniform
      const SpChunk e = S.chunk[ring][c];
      const bool ok = (uint32_t)lane < (e.rem_term & 0xFFu);
      i = e.off + (ok ? lane : 0);
      q = S.t_w[e.rem_term >> 8];
      R.valid |= (ok ? 1u : 0u) << k;
    }
    R.q[k] = q;
    const uint16_t* pd = a.ix.doc_local + i;
    const float* pw = a.ix.w + i;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(R.d[k]) : "v"(pd) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(R.w[k]) : "v"(pw) : "memory");
  }
}
// Retire everything older than the youngest group.  The registers of the group being
// consumed are operands of the wait, so no use of them
// can be scheduled above it and they stay allocated while the loads are in flight.
__device__ __forceinline__ void sp_wait_slots(SpSlots& R) {
  static_assert(SP_K == 8, "operand list below");
  asm volatile("s_waitcnt vmcnt(%16)"
               : "+v"(R.d[0]), "+v"(R.d[1]), "+v"(R.d[2]), "+v"(R.d[3]), "+v"(R.d[4]), "+v"(R.d[5]), "+v"(R.d[6]),
                 "+v"(R.d[7]), "+v"(R.w[0]), "+v"(R.w[1]), "+v"(R.w[2]), "+v"(R.w[3]), "+v"(R.w[4]), "+v"(R.w[5]),
                 "+v"(R.w[6]), "+v"(R.w[7])
               : "n"(2 * SP_K)
               : "memory");
}

// all 64 lanes of wave 0: publish the chunk table of one segment into ring slot `ring`
__device__ __forceinline__ void sp_publish(int ring, int tid, int T, uint32_t off, uint32_t len) {
  const uint32_t n = (len + 63u) >> 6;
  // inclusive scan over lanes 0..15 (T <= SP_TMAX <= 16) with DPP row shifts: no LDS round trips
  uint32_t incl = n, tot = len;
  incl += __builtin_amdgcn_update_dpp(0u, incl, 0x111, 0xF, 0xF, true);   // row_shr:1 (0 shifted in)
  tot += __builtin_amdgcn_update_dpp(0u, tot, 0x111, 0xF, 0xF, true);
  incl += __builtin_amdgcn_update_dpp(0u, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
  tot += __builtin_amdgcn_update_dpp(0u, tot, 0x112, 0xF, 0xF, true);
  incl += __builtin_amdgcn_update_dpp(0u, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
  tot += __builtin_amdgcn_update_dpp(0u, tot, 0x114, 0xF, 0xF, true);
  incl += __builtin_amdgcn_update_dpp(0u, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
  tot += __builtin_amdgcn_update_dpp(0u, tot, 0x118, 0xF, 0xF, true);
  const uint32_t start = incl - n;
  for (uint32_t j = 0; j < n; ++j) {
    const uint32_t c = start + j;
    if (c < (uint32_t)SP_NCH) {
      const uint32_t rem = len - (j << 6);
      S.chunk[ring][c] = SpChunk{off + (j << 6), (rem < 64u ? rem : 64u) | ((uint32_t)tid << 8)};
    }
  }
  if (tid == T - 1) {
    S.nchunks[ring] = incl;      // may exceed SP_NCH: the visit then takes the overflow path
    S.total[ring] = tot;
  }
}

__device__ __forceinline__ void sp_tails_accumulate(const SparseQueryArgs& a, int ring, uint32_t nch, int wave,
                                                    int lane) {
  if (wave == 0) return;
  for (uint32_t c = SP_K * SP_PW + wave - 1; c < nch; c += SP_PW) {
    const SpChunk e = S.chunk[ring][c];
    if ((uint32_t)lane < (e.rem_term & 0xFFu))
      atomicAdd(&S.acc[a.ix.doc_local[e.off + lane]], sp_fix(S.t_w[e.rem_term >> 8], a.ix.w[e.off + lane]));
  }
}
__device__ __forceinline__ void sp_tails_harvest(const SparseQueryArgs& a, uint64_t* cand, int ring, uint32_t nch,
                                                 int wave, int lane, float tau, int64_t gbase) {
  if (wave == 0) return;
  for (uint32_t c = SP_K * SP_PW + wave - 1; c < nch; c += SP_PW) {
    const SpChunk e = S.chunk[ring][c];
    if ((uint32_t)lane < (e.rem_term & 0xFFu)) {
      const uint32_t d = a.ix.doc_local[e.off + lane];
      const unsigned long long x = atomicExch(&S.acc[d], 0ull);
      if (x != 0ull) sp_append(cand, tau, x, gbase + d);
    }
  }
}
// a segment with more chunks than the table holds: walk the runs directly
__device__ __forceinline__ void sp_overflow_accumulate(const SparseQueryArgs& a, int seg, int64_t qb, int T, int tid) {
  for (int t = 0; t < T; ++t) {
    const uint32_t term = (uint32_t)a.q_idx[qb + t];
    Probe p = sp_probe_issue(a, seg, term);
    uint32_t off, len;
    sp_probe_wait(p);
    sp_probe_resolve(a, p, seg, term, off, len);
    const float qw = a.q_val[qb + t];
    for (uint32_t i = tid; i < len; i += SP_THREADS)
      atomicAdd(&S.acc[a.ix.doc_local[off + i]], sp_fix(qw, a.ix.w[off + i]));
  }
}

__device__ __forceinline__ void sp_body_pipe(const SparseQueryArgs& a, uint64_t* cand, unsigned long long* park, int q,
                                             int part, int s0, int s1, int64_t qb, int T, int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t my_term = 0;
  Probe pr{};
  if (tid < 64) {   // wave 0 owns the directory
    uint32_t off0 = 0, len0 = 0, off1 = 0, len1 = 0;
    if (tid < T) {
      my_term = (uint32_t)a.q_idx[qb + tid];
      S.t_w[tid] = a.q_val[qb + tid];
      Probe p0 = sp_probe_issue(a, s0, my_term);
      Probe p1{};
      if (s0 + 1 < s1) p1 = sp_probe_issue(a, s0 + 1, my_term);
      sp_probe_wait(p0);
      sp_probe_wait(p1);
      sp_probe_resolve(a, p0, s0, my_term, off0, len0);
      if (s0 + 1 < s1) sp_probe_resolve(a, p1, s0 + 1, my_term, off1, len1);
      if (s0 + 2 < s1) pr = sp_probe_issue(a, s0 + 2, my_term);
    }
    sp_publish(0, tid, T, off0, len0);
    sp_publish(1, tid, T, off1, len1);
  }
  lds_barrier();
  // Two register sets alternate.  Visit(seg) consumes R = postings(seg), copies the doc ids
  // the harvest needs, and re-fills R with postings(seg+2): in flight for ~1.5 visits.
  SpSlots A, Bq;
  sp_load_slots(a, 0, wave, lane, A);
  sp_load_slots(a, 1, wave, lane, Bq);
  unsigned long long npost = 0;
  SP_STAMP_DECL

  auto visit = [&](SpSlots& R, int seg, const int ring, const int ring2) {
    if (wave != 0) sp_wait_slots(R);   // R has landed; one younger group may still be in flight
    const uint32_t total = __builtin_amdgcn_readfirstlane(R.total), nch = __builtin_amdgcn_readfirstlane(R.nch);
    const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
    const bool overflow = nch > (uint32_t)SP_NCH;               // scalar, absurdly long runs
    const bool tails = nch > (uint32_t)(SP_K * SP_PW);          // scalar
    SP_STAMP(0)
    if (total) {   // scalar
      npost += total;
      sp_make_room(a, cand, total, tid);
      SP_STAMP(1)
      if (!overflow) {
#pragma unroll
        for (int k = 0; k < SP_K; ++k)
          if ((R.valid >> k) & 1u) atomicAdd(&S.acc[R.d[k]], sp_fix(R.q[k], R.w[k]));
        if (tails) sp_tails_accumulate(a, ring, nch, wave, lane);
      }
    }
    SP_STAMP(2)
    if (tid < 64) {   // chunk table of seg+2 (ring slot last used by seg-1, which everybody has left)
      uint32_t off = 0, len = 0;
      if (tid < T && seg + 2 < s1) {
        sp_probe_wait(pr);          // issued a whole visit ago; this wave has nothing else in flight
        sp_probe_resolve(a, pr, seg + 2, my_term, off, len);
        if (seg + 3 < s1) pr = sp_probe_issue(a, seg + 3, my_term);
      }
      sp_publish(ring2, tid, T, off, len);
    }
    if (total && overflow) sp_overflow_accumulate(a, seg, qb, T, tid);
    SP_STAMP(3)
    lds_barrier();                                   // ---- X
    SP_STAMP(4)
    uint32_t hd[SP_K];                               // landed values: plain register copies
    const uint32_t hvalid = R.valid;
#pragma unroll
    for (int k = 0; k < SP_K; ++k) hd[k] = R.d[k];
    sp_load_slots(a, ring2, wave, lane, R);          // postings of seg+2
    SP_STAMP(5)
    if (total) {
      const uint32_t bound = total < (uint32_t)SEG_DOCS ? total : (uint32_t)SEG_DOCS;
      const uint32_t cnt_now = (uint32_t)__builtin_amdgcn_readfirstlane(S.cnt);
      if (overflow || cnt_now + bound > (uint32_t)SP_CAND) {   // scalar
        sp_harvest(a, cand, park, seg, total, tid);             // sweep; ends with a barrier
      } else {
        const float tau = S.tau;
        unsigned long long v[SP_K];
#pragma unroll
        for (int k = 0; k < SP_K; ++k) {
          v[k] = 0ull;
          if ((hvalid >> k) & 1u) v[k] = atomicExch(&S.acc[hd[k]], 0ull);
        }
#pragma unroll
        for (int k = 0; k < SP_K; ++k)
          if (v[k] != 0ull) sp_append(cand, tau, v[k], gbase + hd[k]);
        if (tails) sp_tails_harvest(a, cand, ring, nch, wave, lane, tau, gbase);
        lds_barrier();                               // ---- Y
      }
    }
    SP_STAMP(6)
  };

  int seg = s0;
  for (;;) {   // ring slot of seg = (seg - s0) % 3, register set = (seg - s0) % 2: period 6
    if (seg >= s1) break;
    visit(A, seg, 0, 2);
    if (++seg >= s1) break;
    visit(Bq, seg, 1, 0);
    if (++seg >= s1) break;
    visit(A, seg, 2, 1);
    if (++seg >= s1) break;
    visit(Bq, seg, 0, 2);
    if (++seg >= s1) break;
    visit(A, seg, 1, 0);
    if (++seg >= s1) break;
    visit(Bq, seg, 2, 1);
    ++seg;
  }
  SP_STAMP_FLUSH
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0 && a.stat_postings) atomicAdd(a.stat_postings, npost);
  sp_finish(a, cand, q, part, tid);
}

// ---------------------------------------------------------------------------------
// kernel: pipelined body for short queries, generic loop otherwise
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS, 2) void k_sparse_score(SparseQueryArgs a) {
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
  __syncthreads();
  if (T <= SP_TMAX && T > 0 && s0 < s1) {   // block-uniform
    sp_body_pipe(a, cand, park, q, part, s0, s1, qb, T, tid);
    return;
  }
  unsigned long long npost = 0;
  for (int seg = s0; seg < s1; ++seg) {
    uint32_t total = 0;
    sp_make_room(a, cand, SEG_DOCS, tid);   // acc is all zero between segments
    for (int tc = 0; tc < T; tc += SP_TCH) {
      const int nt = (T - tc) < SP_TCH ? (T - tc) : SP_TCH;
      if (tid < nt) {
        const uint32_t term = (uint32_t)a.q_idx[qb + tc + tid];
        Probe p = sp_probe_issue(a, seg, term);
        uint32_t off, len;
        sp_probe_wait(p);
        sp_probe_resolve(a, p, seg, term, off, len);
        S.t_off[tid] = off;
        S.t_len[tid] = len;
        S.t_w[tid] = a.q_val[qb + tc + tid];
      }
      __syncthreads();
      for (int t = 0; t < nt; ++t) {
        const uint32_t len = S.t_len[t], off = S.t_off[t];
        const float qw = S.t_w[t];
        total += len;
        for (uint32_t i = tid; i < len; i += SP_THREADS)
          atomicAdd(&S.acc[a.ix.doc_local[off + i]], sp_fix(qw, a.ix.w[off + i]));
      }
      __syncthreads();  // t_off/t_len are rewritten by the next round
    }
    if (total) {  // block-uniform
      npost += total;
      sp_harvest(a, cand, park, seg, total, tid);
      __syncthreads();
    }
  }
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
