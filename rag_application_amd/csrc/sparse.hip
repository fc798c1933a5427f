// K7 -- sparse ("BM25") scoring over the on-device inverted index.
//
// Mirrors Prefetch(query=SparseVector, using="sparse", limit=sparse_limit)
// (app/core/vector_store/qdrant/qdrant_handler.py:347-354): score(d) = sum over the
// query's terms (ascending term id) of q_t * d_t, fp32 mul then fp32 add, IDF-free
// (the collection sets no sparse modifier, :80-86); only documents that share a term
// are candidates.  Arithmetic = oracle.OracleIndex.sparse_scores bit for bit.
//
// Index layout (spbuild.hip): documents are cut into segments of SEG_DOCS; inside a
// segment postings are sorted by (term, doc) as {u16 doc_local, f32 weight}; an
// open-addressing table maps (segment, term) -> (offset, length).
//
// One 512-thread workgroup owns (query, part): a contiguous range of segments.  Per
// segment it accumulates the terms ONE AFTER ANOTHER into a 32 KiB LDS accumulator
// (docs are unique inside a posting run, so a term step is race-free and the sum
// order is the oracle's), marks touched documents in an LDS bitmap, then harvests the
// touched documents against the running threshold into an LDS candidate buffer that
// is bitonic-sorted and truncated to `limit` whenever it could overflow.
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

constexpr int SP_THREADS = 512;
constexpr int SP_CB = 4096;          // LDS candidate buffer (keys)
constexpr int SP_PIECE = 2048;       // docs harvested between capacity checks
constexpr int SP_TCH = 64;           // query terms looked up per round

__host__ __device__ inline uint64_t sp_hash(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ void sp_sort_truncate(uint64_t* cb, int* cnt, int limit, float* tau, int tid) {
  // bitonic sort (descending) of the whole buffer; unused slots are 0
  const int n = *cnt;
  for (int i = n + tid; i < SP_CB; i += SP_THREADS) cb[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= SP_CB; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < SP_CB; i += SP_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = cb[i], y = cb[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            cb[i] = y;
            cb[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    if (n >= limit) {
      *tau = key_score(cb[limit - 1]);
      *cnt = limit;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(SP_THREADS) void k_sparse_score(SparseQueryArgs a) {
  __shared__ float acc[SEG_DOCS];
  __shared__ uint32_t bitmap[SEG_DOCS / 32];
  __shared__ uint64_t cb[SP_CB];
  __shared__ uint32_t t_off[SP_TCH], t_len[SP_TCH];
  __shared__ float t_w[SP_TCH];
  __shared__ int s_cnt;
  __shared__ float s_tau;
  __shared__ int s_any;

  const int tid = threadIdx.x;
  const int q = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
  const int nseg = a.ix.n_segments;
  const int s0 = (int)((int64_t)nseg * part / a.parts), s1 = (int)((int64_t)nseg * (part + 1) / a.parts);
  const int64_t qb = a.q_indptr[q];
  const int T = (int)(a.q_indptr[q + 1] - qb);

  for (int i = tid; i < SEG_DOCS; i += SP_THREADS) acc[i] = 0.0f;
  for (int i = tid; i < SEG_DOCS / 32; i += SP_THREADS) bitmap[i] = 0u;
  if (tid == 0) {
    s_cnt = 0;
    s_tau = -__builtin_inff();
    s_any = 0;
  }
  __syncthreads();

  for (int seg = s0; seg < s1; ++seg) {
    for (int tc = 0; tc < T; tc += SP_TCH) {
      const int nt = (T - tc) < SP_TCH ? (T - tc) : SP_TCH;
      if (tid < nt) {
        const uint32_t term = (uint32_t)a.q_idx[qb + tc + tid];
        const uint64_t key = ((uint64_t)seg << 31) | term;
        uint64_t slot = sp_hash(key) & a.ix.table_mask;
        uint32_t off = 0, len = 0;
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
        t_off[tid] = off;
        t_len[tid] = len;
        t_w[tid] = a.q_val[qb + tc + tid];
        if (len) s_any = 1;
      }
      __syncthreads();
      for (int t = 0; t < nt; ++t) {
        const uint32_t len = t_len[t];
        if (len == 0) continue;  // block-uniform
        const uint32_t off = t_off[t];
        const float qw = t_w[t];
        for (uint32_t i = tid; i < len; i += SP_THREADS) {
          const uint32_t d = a.ix.doc_local[off + i];
          const float w = a.ix.w[off + i];
          acc[d] = __fadd_rn(acc[d], __fmul_rn(qw, w));
          atomicOr(&bitmap[d >> 5], 1u << (d & 31));
        }
        __syncthreads();
      }
      __syncthreads();  // t_off/t_len may be rewritten by the next round
    }
    if (s_any) {  // block-uniform (read after a barrier)
      const int64_t gbase = a.ix.id_base + (int64_t)seg * SEG_DOCS;
      for (int piece = 0; piece < SEG_DOCS / SP_PIECE; ++piece) {
        if (s_cnt + SP_PIECE > SP_CB) sp_sort_truncate(cb, &s_cnt, a.limit, &s_tau, tid);
        const float tau = s_tau;
        const int d0 = piece * SP_PIECE + tid * (SP_PIECE / SP_THREADS);
        const uint32_t word = bitmap[d0 >> 5];
        uint32_t bits = (word >> (d0 & 31)) & ((1u << (SP_PIECE / SP_THREADS)) - 1u);
        while (bits) {
          const int bpos = __builtin_ctz(bits);
          bits &= bits - 1;
          const int d = d0 + bpos;
          const float s = acc[d];
          acc[d] = 0.0f;
          if (s >= tau) {
            const int pos = atomicAdd(&s_cnt, 1);
            cb[pos] = make_key(s, (uint32_t)(gbase + d));
          }
        }
        __syncthreads();
      }
      for (int i = tid; i < SEG_DOCS / 32; i += SP_THREADS) bitmap[i] = 0u;
      if (tid == 0) s_any = 0;
      __syncthreads();
    }
  }
  sp_sort_truncate(cb, &s_cnt, a.limit, &s_tau, tid);
  const int n = s_cnt < a.limit ? s_cnt : a.limit;
  uint64_t* o = a.out + ((int64_t)q * a.parts + part) * a.limit;
  for (int i = tid; i < a.limit; i += SP_THREADS) o[i] = i < n ? cb[i] : 0ull;
  if (tid == 0) a.out_cnt[q * a.parts + part] = n;
}

void launch_sparse_score(const SparseQueryArgs& a, hipStream_t st) {
  if (a.B <= 0) return;
  HX_CHECK(a.limit + SP_PIECE <= SP_CB, "sparse: limit too large");
  hipLaunchKernelGGL(k_sparse_score, dim3(a.B * a.parts), dim3(SP_THREADS), 0, st, a);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
