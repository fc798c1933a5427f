// K7, the passes around the select kernel (sparse2.hip): per-query preparation, launch order, the exact
// score of the kept candidates, and the document-at-a-time path for queries the integer pass cannot serve.
//
// Exact score = upstream's order (oracle.OracleIndex.sparse_scores, SPARSE_FIX_BITS = None): the query's
// terms in ascending term id, acc = acc + q_t * d_t from +0 with an fp32 multiply and an fp32 add, over
// the terms the document holds (app/core/vector_store/qdrant/qdrant_handler.py:347-354 -> Qdrant's sparse
// search, IDF-free).  One wave per (query, document): the lanes hold the document's terms, a ballot finds
// the query term among them.
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

// live-term index of `term` by one wave in 64-ary steps: ~log64(n_live) dependent loads (`term` is
// wave-uniform); -1 if absent
__device__ __forceinline__ int sp_find_term_wave(const uint32_t* uterms, int n_live, uint32_t term, int lane) {
  int lo = 0, hi = n_live;   // the first index whose term is >= `term` lies in [lo, hi]
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + lane * step;
    const bool in = p < hi;
    const uint32_t v = in ? uterms[p] : 0xFFFFFFFFu;
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
  const uint32_t v = in ? uterms[p] : 0xFFFFFFFFu;
  const unsigned long long eq = __ballot(in && v == term);
  return eq ? lo + (int)__builtin_ctzll(eq) : -1;
}

// ---------------------------------------------------------------------------------
// preparation: one wave per query
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sparse_prep(SparsePrepArgs a) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= a.B) return;
  const int64_t qb = a.q_indptr[q];
  const int T = (int)(a.q_indptr[q + 1] - qb);
  int flag = 0;
  // weights: finite (else the query is invalid), positive (else the integer pass cannot bracket the score)
  // term ids: strictly ascending (the exact score is a running fp32 sum in ascending term id, upstream's order:
  // another order gives other score bits; a repeated id would be counted twice) and non-negative
  bool bad = false, nonpos = false, disorder = false;
  double sum = 0.0;
  for (int j = lane; j < T; j += 64) {
    const float v = a.q_val[qb + j];
    bad |= !(__builtin_fabsf(v) <= 3.0e38f);
    nonpos |= !(v > 0.0f);
    sum += (double)v;
    const int32_t id = a.q_idx[qb + j];
    disorder |= id < 0 || (j > 0 && !(a.q_idx[qb + j - 1] < id));
  }
  if (__ballot(bad)) flag = 2;
  else if (__ballot(disorder)) flag = 3;
  else if (__ballot(nonpos) || a.index_nonpos || T > SP_TMAX) flag = 1;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);   // same value in every lane
  // scale: the largest possible score maps below 65535 - T - 8 (sparse2.hip)
  const double smax = sum * (double)a.wmax;
  double scale = 0.0;
  if (T > 0 && flag == 0) {
    if (smax > 0.0 && smax < 1.0e300) scale = (double)(65535 - T - 8) / smax;
    if (!(scale > 0.0 && scale < 1.0e30)) flag = 1;
  }
  // live-term index of every term in both views; postings of the query
  unsigned long long work = 0;
  const int Tn = T < SP_TMAX ? T : SP_TMAX;
  for (int t = 0; t < Tn; ++t) {
    const uint32_t term = (uint32_t)a.q_idx[qb + t];
    for (int v = 0; v < 2; ++v) {
      int r = -1;
      if (a.ix[v].n_live > 0) r = sp_find_term_wave(a.ix[v].uterms, a.ix[v].n_live, term, lane);
      if (lane == 0) {
        a.q_ti[v][(int64_t)q * SP_TMAX + t] = r;
        if (r >= 0) {
          const uint32_t* row = a.ix[v].ptr + (int64_t)r * (a.ix[v].n_segments + 1);
          work += (unsigned long long)(row[a.ix[v].n_segments] - row[0]);
        }
      }
    }
  }
  if (lane < Tn) {
    const float qs = (float)((double)a.q_val[qb + lane] * scale);
    a.q_qs[(int64_t)q * SP_TMAX + lane] = qs;
  }
  if (lane == 0) {
    a.q_margin[q] = T + T / 16 + 4;
    a.q_flag[q] = flag;
    a.q_work[q] = work;
    if (a.stat_postings && work) atomicAdd(a.stat_postings, work);
  }
}
void launch_sparse_prep(const SparsePrepArgs& a, hipStream_t st) {
  if (a.B <= 0) return;
  hipLaunchKernelGGL(k_sparse_prep, dim3((a.B + 3) / 4), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

// Launch plan.  A query is cut into q_parts workgroups ("parts", each a contiguous range of segments) by its share
// of the batch's postings -- a query far above the average would otherwise finish long after the rest -- and the
// (query, part) items are listed by descending work per item (longest-processing-time first; rank by counting).
// items[k] = query << 8 | part for k < *n_items.  Consecutive workgroups land on different XCDs, so the list must
// be dense: empty workgroups in between would leave whole XCDs without work.
// Every block derives the batch's totals and parts again (B <= 4096 values: cheaper than a second launch) and
// ranks 64 of the queries, 16 lanes per query.
__global__ __launch_bounds__(1024) void k_sparse_plan(const unsigned long long* work, int B, int pt_max, int slots,
                                                      int* q_parts, int* items, int* n_items) {
  __shared__ unsigned long long part[1024];
  __shared__ unsigned int s_w[4096];     // work per part, scaled to 32 bits (the order needs no more)
  __shared__ unsigned char s_p[4096];
  const int tid = threadIdx.x;
  unsigned long long mine = 0;
  for (int i = tid; i < B; i += 1024) mine += work[i];
  part[tid] = mine;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (tid < off) part[tid] += part[tid + off];
    __syncthreads();
  }
  const unsigned long long total = part[0];
  __syncthreads();
  // two workgroups' worth of work per resident slot keeps the tail short
  const unsigned long long target = total / (unsigned long long)(2 * slots) + 1;
  int sh = 0;
  while ((total >> sh) > 0xFFFFFFFFull) ++sh;      // no query's work exceeds the batch total
  unsigned long long np = 0;
  for (int i = tid; i < B; i += 1024) {
    unsigned long long p = (work[i] + target - 1) / target;
    p = p < 1 ? 1 : (p > (unsigned long long)pt_max ? (unsigned long long)pt_max : p);
    s_w[i] = (unsigned int)((work[i] >> sh) / p);
    s_p[i] = (unsigned char)p;
    np += p;
    if (blockIdx.x == 0) q_parts[i] = (int)p;
  }
  part[tid] = np;
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int off = 512; off >= 1; off >>= 1) {
      if (tid < off) part[tid] += part[tid + off];
      __syncthreads();
    }
    if (tid == 0) *n_items = (int)part[0];
  }
  const int i = blockIdx.x * 64 + (tid >> 4), sl = tid & 15;
  int pos = 0;
  if (i < B) {
    const unsigned int w = s_w[i];
    for (int j = sl; j < B; j += 16) {
      const unsigned int x = s_w[j];
      pos += (x > w || (x == w && j < i)) ? (int)s_p[j] : 0;
    }
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) pos += __shfl_xor(pos, off, 64);
  if (i < B && sl == 0) {
    const int p = (int)s_p[i];
    for (int k = 0; k < p; ++k) items[pos + k] = (i << 8) | k;
  }
}
void launch_sparse_plan(const unsigned long long* q_work, int B, int pt_max, int slots, int* q_parts, int* items,
                        int* n_items, hipStream_t st) {
  if (B <= 0) return;
  HX_CHECK(B <= 4096 && pt_max <= 255, "sparse plan: batch too large");
  hipLaunchKernelGGL(k_sparse_plan, dim3((B + 63) / 64), dim3(1024), 0, st, q_work, B, pt_max, slots, q_parts, items, n_items);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// exact score of (query, document) on one wave
// ---------------------------------------------------------------------------------
// PRE: the query's terms and weights are held by the wave (lane t: term t, T <= 64) instead of being loaded term by term
template <bool PRE = false>
__device__ __forceinline__ float sp_exact_score(const SparseCsr& d, int64_t doc, const int32_t* q_idx,
                                                const float* q_val, int64_t qb, int T, int lane, bool& any,
                                                int32_t qi_lane = 0, float qw_lane = 0.0f) {
  const int64_t b = d.indptr[doc], e = d.indptr[doc + 1];
  // the document's first 128 terms live in registers; longer documents re-read the rest per query term
  const int32_t i0 = b + lane < e ? d.idx[b + lane] : -1;
  const int32_t i1 = b + 64 + lane < e ? d.idx[b + 64 + lane] : -1;
  // ... and so do their weights: a matched term's weight is a lane read, not another trip to memory behind the match
  const float w0 = b + lane < e ? d.val[b + lane] : 0.0f;
  const float w1 = b + 64 + lane < e ? d.val[b + 64 + lane] : 0.0f;
  float acc = 0.0f;
  any = false;
  for (int t = 0; t < T; ++t) {
    const int32_t term = PRE ? __builtin_amdgcn_readlane(qi_lane, t) : q_idx[qb + t];
    bool hit = false;
    float w = 0.0f;
    unsigned long long m = __ballot(i0 == term);
    if (m) {
      hit = true;
      w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), __builtin_ctzll(m)));
    } else {
      m = __ballot(i1 == term);
      if (m) {
        hit = true;
        w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w1), __builtin_ctzll(m)));
      } else
        for (int64_t c = b + 128; c < e; c += 64) {
          m = __ballot(c + lane < e && d.idx[c + lane] == term);
          if (m) {
            hit = true;
            w = d.val[c + __builtin_ctzll(m)];
            break;
          }
        }
    }
    if (hit) {                                       // wave-uniform
      const float qw = PRE ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qw_lane), t)) : q_val[qb + t];
      acc = __fadd_rn(acc, __fmul_rn(qw, w));
      any = true;
    }
  }
  return acc;
}

__device__ __forceinline__ uint32_t spr_thr(uint32_t aL, int M) {   // sparse2.hip sp_thr
  const int t = (int)aL - M + 1;
  return t < 1 ? 1u : (uint32_t)t;
}

// grid (SPR_BLOCKS, B): the waves of a query's blocks stride over its candidates (a list holds the top-L plus the
// documents within the margin of the L-th: about L + 10 keys, thousands when many documents tie at the L-th score)
constexpr int SPR_BLOCKS = 32;
__global__ __launch_bounds__(256) void k_sparse_rescore(SparseRescoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  uint64_t* out = a.out + (int64_t)b * a.stride;
  int n = 0;
  uint32_t thr = 1u;
  const uint64_t* list = a.cand + (int64_t)b * a.stride;
  if (a.q_flag[b] == 0) {
    n = a.cnt[b];
    n = n < a.stride ? n : a.stride;
    if (a.thr_in) thr = a.thr_in[b];
    else if (n >= a.limit) thr = spr_thr((uint32_t)(list[a.limit - 1] >> 32), a.q_margin[b]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the list is sorted by integer score: the candidates are a prefix
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)(list[mid] >> 32) >= thr) lo = mid + 1; else hi = mid;
    }
    a.out_cnt[b] = lo;
    if (lo == a.stride) a.q_fail[b] = 1;        // a full list that passes to its last key: cut short
  }
  const int64_t qb = a.q_indptr[b];
  const int T = (int)(a.q_indptr[b + 1] - qb);
  // the query's terms once per wave (one vector load each) instead of one dependent scalar load per term and candidate
  const bool pre = T <= 64;
  const int32_t qi_lane = (pre && lane < T) ? a.q_idx[qb + lane] : -1;
  const float qw_lane = (pre && lane < T) ? a.q_val[qb + lane] : 0.0f;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int)gridDim.x * 4) {   // wave-uniform
    const uint64_t ck = list[i];
    if ((uint32_t)(ck >> 32) < thr) break;      // sorted: nothing further passes
    uint64_t k = 0ull;
    const int64_t doc = (int64_t)(0xFFFFFFFFu - (uint32_t)ck) - a.d.id_base;
    if (doc >= 0 && doc < a.d.n_docs) {
      bool any;
      const float s = pre ? sp_exact_score<true>(a.d, doc, a.q_idx, a.q_val, qb, T, lane, any, qi_lane, qw_lane)
                          : sp_exact_score<false>(a.d, doc, a.q_idx, a.q_val, qb, T, lane, any);
      if (any) k = make_key(s, (uint32_t)(a.d.id_base + doc));
    }
    if (lane == 0) out[i] = k;
  }
}
void launch_sparse_rescore(const SparseRescoreArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.stride <= 0) return;
  hipLaunchKernelGGL(k_sparse_rescore, dim3(a.blocks > 0 ? a.blocks : SPR_BLOCKS, a.B), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

// what the host needs to know about a batch's sparse flags in 8 bytes: out[0] = queries flagged for (or failed
// into) the document-at-a-time path, out[1] = queries with a non-finite weight
__global__ __launch_bounds__(256) void k_sparse_summary(const int* flag, const int* fail, int B, int* out) {
  __shared__ int s_bad, s_inv;
  if (threadIdx.x == 0) {
    s_bad = 0;
    s_inv = 0;
  }
  __syncthreads();
  int bad = 0, inv = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    bad += (flag[b] != 0 || fail[b] != 0) ? 1 : 0;
    inv += flag[b] >= 2 ? 1 : 0;
  }
  if (bad) atomicAdd(&s_bad, bad);
  if (inv) atomicAdd(&s_inv, inv);
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = s_bad;
    out[1] = s_inv;
  }
}
void launch_sparse_summary(const int* flag, const int* fail, int B, int* out, hipStream_t st) {
  hipLaunchKernelGGL(k_sparse_summary, dim3(1), dim3(256), 0, st, flag, fail, B, out);
  HX_HIP(hipGetLastError());
}

// the parts' lists of a query (each best first, `lout` slots, zeros after its count) -> one packed run + count
__global__ __launch_bounds__(256) void k_sparse_pack(const uint64_t* parts, const int* pcnt, int pt, int lout,
                                                     uint64_t* out, int* out_cnt) {
  const int b = blockIdx.x;
  int base = 0;
  for (int p = 0; p < pt; ++p) {
    int n = pcnt[b * pt + p];
    n = n < lout ? n : lout;
    for (int i = threadIdx.x; i < n; i += 256) out[(int64_t)b * pt * lout + base + i] = parts[((int64_t)b * pt + p) * lout + i];
    base += n;
  }
  if (threadIdx.x == 0) out_cnt[b] = base;
}
void launch_sparse_pack(const uint64_t* parts, const int* pcnt, int B, int pt, int lout, uint64_t* out, int* out_cnt,
                        hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_sparse_pack, dim3(B), dim3(256), 0, st, parts, pcnt, pt, lout, out, out_cnt);
  HX_HIP(hipGetLastError());
}

// Document-at-a-time path.  Rows [row_begin, row_end) against the listed queries: a document that shares a term
// with the query and scores >= tau[f] appends its key to the query's buffer (slots [cnt0, cap)); a full
// buffer sets ovf[f].  tau NULL: every row has its own slot (slot0 + row - row_begin), 0 when no term is shared.
__global__ __launch_bounds__(256) void k_sparse_range(SparseRangeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = a.row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.row_end) return;
  const int f = blockIdx.y;
  const int b = a.qsel[f];
  const int64_t qb = a.q_indptr[b];
  bool any;
  const float s = sp_exact_score(a.d, row, a.q_idx, a.q_val, qb, (int)(a.q_indptr[b + 1] - qb), lane, any);
  if (lane != 0) return;
  uint64_t* o = a.out + (int64_t)f * a.stride;
  if (!a.tau) {
    o[a.slot0 + (row - a.row_begin)] = any ? make_key(s, (uint32_t)(a.d.id_base + row)) : 0ull;
  } else if (any && s >= a.tau[f]) {
    const int pos = atomicAdd(a.cnt + f, 1);
    if (pos < a.stride) o[pos] = make_key(s, (uint32_t)(a.d.id_base + row));
    else a.ovf[f] = 1;
  }
}
void launch_sparse_range(const SparseRangeArgs& a, hipStream_t st) {
  const int64_t n = a.row_end - a.row_begin;
  if (n <= 0 || a.nsel <= 0) return;
  hipLaunchKernelGGL(k_sparse_range, dim3((unsigned)((n + 3) / 4), a.nsel), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// {min, max} of the document weights (as orderable u32), count of non-finite ones
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_f32(const float* __restrict__ val, int64_t n, uint32_t* mm) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0u, bad = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = val[i];
    if (!(__builtin_fabsf(v) <= 3.0e38f)) {
      ++bad;
      continue;
    }
    const uint32_t u = f32_orderable(v);
    lo = u < lo ? u : lo;
    hi = u > hi ? u : hi;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
    bad += __shfl_xor(bad, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(mm + 0, lo);
    atomicMax(mm + 1, hi);
    if (bad) atomicAdd(mm + 2, bad);
  }
}
void launch_minmax_f32(const float* val, int64_t n, float* mm, hipStream_t st) {
  if (n <= 0) return;
  const int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(k_minmax_f32, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, val, n,
                     (uint32_t*)mm);
  HX_HIP(hipGetLastError());
}

// rows of a CSR on the device (hx_load: all of them; hx_add_sparse: the batch being added, before it is committed):
// indptr_rows[0] = first, monotone, indptr_rows[n_rows] = last; idx[first .. last) >= 0.  *bad |= 1 (offsets) | 2 (ids)
__global__ void k_csr_check(const int64_t* indptr_rows, const int32_t* idx, int64_t n_rows, int64_t first, int64_t last,
                            int* bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b = 0;
  if (i < n_rows) b |= (indptr_rows[i] > indptr_rows[i + 1] || indptr_rows[i] < first || indptr_rows[i + 1] > last) ? 1 : 0;
  if (i == 0) b |= (indptr_rows[0] != first || indptr_rows[n_rows] != last) ? 1 : 0;
  if (i < last - first) b |= idx[first + i] < 0 ? 2 : 0;
  if (b) atomicOr(bad, b);
}
void launch_csr_check(const int64_t* indptr_rows, const int32_t* idx, int64_t n_rows, int64_t first, int64_t last, int* bad,
                      hipStream_t st) {
  const int64_t m = n_rows > last - first ? n_rows : last - first;
  if (m <= 0) return;
  hipLaunchKernelGGL(k_csr_check, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, indptr_rows, idx, n_rows, first, last, bad);
  HX_HIP(hipGetLastError());
}

// term ids unique within a row, one wave per row: the row's chunks of 64 ids against each other (len^2 / 64 lane
// reads: rows up to CSR_UNIQUE_WAVE_MAX ids; longer ones are listed for the host)
__global__ __launch_bounds__(256) void k_csr_unique(const int64_t* indptr, const int32_t* idx, int64_t n_rows, int* bad,
                                                    int64_t* long_rows, int long_cap, int* n_long) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int64_t b = indptr[row], e = indptr[row + 1];
  const int64_t len = e - b;
  if (len < 2) return;
  if (len > CSR_UNIQUE_WAVE_MAX) {
    if (lane == 0) {
      const int p = atomicAdd(n_long, 1);
      if (p < long_cap) long_rows[p] = row;
    }
    return;
  }
  bool dup = false;
  for (int64_t c = b; c < e; c += 64) {
    const int32_t mine = c + lane < e ? idx[c + lane] : -1 - lane;          // padding lanes: distinct negatives
    for (int64_t c2 = c; c2 < e; c2 += 64) {
      const int32_t other = c2 + lane < e ? idx[c2 + lane] : -100 - lane;
      const int m = (int)((e - c2) < 64 ? (e - c2) : 64);
      for (int j = 0; j < m; ++j) {
        const int32_t v = __builtin_amdgcn_readlane(other, j);
        dup |= (mine == v) && !(c2 == c && j == lane);
      }
    }
  }
  if (__ballot(dup) && lane == 0) atomicOr(bad, 4);
}
void launch_csr_unique(const int64_t* indptr, const int32_t* idx, int64_t n_rows, int* bad, int64_t* long_rows,
                       int long_cap, int* n_long, hipStream_t st) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(k_csr_unique, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, indptr, idx, n_rows, bad,
                     long_rows, long_cap, n_long);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
