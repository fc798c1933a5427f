// K9 -- on-device inverted-index build (rocPRIM radix sort + two small kernels), plus the synthetic sparse
// corpus generator.
//
// Mirrors Qdrant's sparse index build on upsert of the "sparse" named vector
// (app/core/vector_store/qdrant/qdrant_handler.py:80-86, 163, 190-193).  Input is the
// doc-major CSR the ingest path appends; output is the TERM-major posting store of
// sparse2.hip: postings sorted by (term, document), the ascending list of live terms, and a
// dense [live term x segment] table of posting offsets (so a workgroup that walks the
// segments in order finds every run by indexing -- no hashing, no probing -- and a term's
// runs of consecutive segments are adjacent in memory).
// The sort is rocPRIM's LSD radix sort on the 31-bit term id alone: it is stable and the
// input is document-major, so documents stay ascending inside a term.
#include "hx_common.hpp"
#include "kernels.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

namespace hx {

struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t bytes) { HX_HIP(hipMalloc(&p, bytes ? bytes : 16)); }
  ~DevBuf() { (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  template <typename T>
  T* as() { return (T*)p; }
};

// one wave per document of [doc0, doc0 + n_docs): key = term id, payload = (document index relative to doc0,
// weight bits); pair i of the output is posting indptr[doc0] + i of the CSR
__global__ void k_make_pairs(const int64_t* indptr, const int32_t* idx, const float* val, int64_t doc0, int64_t n_docs,
                             uint32_t* keys, uint64_t* pay) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (d >= n_docs) return;
  const int64_t base = indptr[doc0];
  const int64_t b = indptr[doc0 + d], e = indptr[doc0 + d + 1];
  for (int64_t i = b + lane; i < e; i += 64) {
    keys[i - base] = (uint32_t)idx[i];
    uint32_t wb;
    const float w = val[i];
    __builtin_memcpy(&wb, &w, 4);
    pay[i - base] = ((uint64_t)d << 32) | wb;
  }
}

// ---- bank-spread order of a (term, segment) run ---------------------------------------------------------------------
// The select pass (sparse2.hip) takes a run in chunks of 128 postings, lane l of a wave the postings 2l and 2l + 1 of
// the chunk, and adds them to LDS with two ds_add_u32 (and harvests them with two ds_and_rtn_b32).  A 32-bit LDS
// instruction is served in two groups of 32 lanes, bank = word mod 32 = document mod 32: so positions {64 b + 2 i} and
// {64 b + 2 i + 1}, i < 32, of a run are one group each, and every further posting of a group on a bank already taken
// costs an LDS cycle (measured in round 2: 60 % of the LDS cycles of the pass were such conflicts -- 32 random banks
// out of 32 put 3.4 on the fullest).  The integer pass does not care about the order inside a run (sums commute, a
// document appears once per run), the exact pass reads the document-major CSR.  So the build deals a run's postings
// into groups by bank: posting number k (in document order) of bank b goes to "tier" k; tiers in order, banks
// ascending inside a tier, is a sequence whose every aligned 32 hold distinct banks as long as the tiers are full
// (k < the rarest bank's count) and at most two per bank after that.  Group g of the sequence takes the positions
// of group g: the even, then the odd positions of block g / 2; the last, partial block is split in halves.
__device__ __forceinline__ uint2 sp_encode(uint64_t p, uint32_t seg_docs) {
  // first word: where the document's 16-bit accumulator lives in the select pass's LDS (sparse2.hip: sp_word) --
  // byte offset of its 32-bit word (document mod seg_docs/2, times 4) | shift of its half (0 / 16) << 24
  const uint32_t d = (uint32_t)(p >> 32) % seg_docs, words = seg_docs >> 1;
  return make_uint2(((d & (words - 1)) << 2) | (d >= words ? 16u << 24 : 0u), (uint32_t)p);
}
// sorted payloads -> postings {accumulator place of the document inside its segment, weight bits}
__global__ void k_make_postings(const uint64_t* pay, int64_t nnz, uint32_t seg_docs, uint2* post) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  post[i] = sp_encode(pay[i], seg_docs);
}

// one wave: run [b, e) of `pay` (document order) -> post[b .. e) in bank-spread order
__device__ __forceinline__ void sp_spread_run(const uint64_t* pay, uint2* post, uint64_t b, uint64_t e, uint32_t seg_docs,
                                              int lane) {
  const uint32_t len = (uint32_t)(e - b);
  if (len <= 2) {                                   // one posting per group at most
    if ((uint32_t)lane < len) post[b + lane] = sp_encode(pay[b + lane], seg_docs);
    return;
  }
  // pass A: postings per bank (wave-uniform counters: popcounts of ballots)
  uint32_t cnt[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) cnt[k] = 0;
  for (uint32_t c = 0; c < len; c += 64) {
    const bool in = c + lane < len;
    const uint32_t bank = in ? (uint32_t)(pay[b + c + lane] >> 32) & 31u : 0xFFu;
#pragma unroll
    for (int k = 0; k < 32; ++k) cnt[k] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(bank == (uint32_t)k));
  }
  // pass B: number of a posting inside its bank -> place in the tier sequence -> position of its group
  const uint32_t full = len >> 6, tail = len & 63u, half = (tail + 1) >> 1;
  uint32_t seen[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) seen[k] = 0;
  const uint64_t below = (1ull << lane) - 1ull;
  for (uint32_t c = 0; c < len; c += 64) {
    const bool in = c + lane < len;
    const uint64_t p = in ? pay[b + c + lane] : 0ull;
    const uint32_t bank = in ? (uint32_t)(p >> 32) & 31u : 0xFFu;
    uint32_t occ = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const uint64_t m = __builtin_amdgcn_ballot_w64(bank == (uint32_t)k);
      if (bank == (uint32_t)k) occ = seen[k] + (uint32_t)__builtin_popcountll(m & below);
      seen[k] += (uint32_t)__builtin_popcountll(m);
    }
    uint32_t i = 0;                                 // place in the sequence: the tiers before, the lower banks of this tier
#pragma unroll
    for (int k = 0; k < 32; ++k) i += (cnt[k] < occ ? cnt[k] : occ) + (((uint32_t)k < bank && cnt[k] > occ) ? 1u : 0u);
    const uint32_t blk = i >> 6, r = i & 63u;
    uint32_t pos;
    if (blk < full) pos = (blk << 6) + ((r & 31u) << 1) + (r >> 5);
    else pos = (blk << 6) + (r < half ? (r << 1) : (((r - half) << 1) + 1u));
    if (in) post[b + pos] = sp_encode(p, seg_docs);
  }
}

// terms of at most SPREAD_HEAVY postings: one wave per term, run by run (a run's end comes from the offset table);
// the others are listed for k_spread_heavy
constexpr uint32_t SPREAD_HEAVY = 32768;
__global__ __launch_bounds__(256) void k_spread_light(const uint64_t* pay, const uint64_t* run_off, const uint32_t* run_len,
                                                      const uint32_t* ptr, int64_t n_live, int nseg, uint32_t seg_docs,
                                                      uint2* post, uint32_t* heavy, uint32_t* n_heavy) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_live) return;
  const uint32_t cnt = run_len[t];
  if (cnt > SPREAD_HEAVY) {
    if (lane == 0) heavy[atomicAdd(n_heavy, 1u)] = (uint32_t)t;
    return;
  }
  const uint64_t b0 = run_off[t], e0 = b0 + cnt;
  const uint32_t* row = ptr + t * (int64_t)(nseg + 1);
  uint64_t cur = b0;
  while (cur < e0) {                                // wave-uniform
    const uint32_t seg = (uint32_t)(pay[cur] >> 32) / seg_docs;
    const uint64_t end = row[seg + 1];
    sp_spread_run(pay, post, cur, end, seg_docs, lane);
    cur = end;
  }
}
// one workgroup per (heavy term, 16 segments): a wave per run
__global__ __launch_bounds__(256) void k_spread_heavy(const uint64_t* pay, const uint32_t* ptr, const uint32_t* heavy,
                                                      int nseg, uint32_t seg_docs, uint2* post) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t t = heavy[blockIdx.x];
  const uint32_t* row = ptr + t * (int64_t)(nseg + 1);
  for (int s = blockIdx.y * 16 + w; s < nseg && s < (int)blockIdx.y * 16 + 16; s += 4) {
    const uint64_t b = row[s], e = row[s + 1];
    if (e > b) sp_spread_run(pay, post, b, e, seg_docs, lane);
  }
}

// one workgroup per live term: ptr[t][s] = first posting of the term's run [b, e) whose document
// is >= s * seg_docs (binary search over the sorted payloads), s = 0 .. nseg
__global__ __launch_bounds__(256) void k_fill_ptr(const uint64_t* pay, const uint64_t* run_off,
                                                  const uint32_t* run_len, int nseg, int seg_docs, uint32_t* ptr) {
  const int64_t t = blockIdx.x;
  const uint64_t b = run_off[t], e = b + run_len[t];
  uint32_t* row = ptr + t * (int64_t)(nseg + 1);
  for (int s = threadIdx.x; s <= nseg; s += 256) {
    const uint64_t first_doc = (uint64_t)s * (uint64_t)seg_docs;
    uint64_t lo = b, hi = e;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if ((pay[mid] >> 32) < first_doc) lo = mid + 1; else hi = mid;
    }
    row[s] = (uint32_t)lo;
  }
}

void build_sparse_index(const int64_t* indptr, const int32_t* idx, const float* val, int64_t doc0, int64_t n_docs,
                        int seg_docs, SparseBuildOut* out, hipStream_t st) {
  *out = SparseBuildOut{};
  if (n_docs <= 0) return;
  int64_t ends[2] = {0, 0};
  HX_HIP(hipMemcpyAsync(&ends[0], indptr + doc0, 8, hipMemcpyDeviceToHost, st));
  HX_HIP(hipMemcpyAsync(&ends[1], indptr + doc0 + n_docs, 8, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  const int64_t nnz = ends[1] - ends[0];
  if (nnz <= 0) return;
  HX_CHECK(nnz < (int64_t)0xFFFFFFFFll, "sparse index: nnz per shard must be < 2^32");
  HX_CHECK(n_docs < (int64_t)0xFFFFFFFFll, "sparse index: documents per shard must be < 2^32");
  HX_CHECK(seg_docs == SEG_DOCS_SMALL || seg_docs == SEG_DOCS_LARGE, "bad segment size");
  const int64_t nseg = (n_docs + seg_docs - 1) / seg_docs;

  DevBuf k_in(nnz * 4), k_out(nnz * 4), p_in(nnz * 8), p_out(nnz * 8);
  hipLaunchKernelGGL(k_make_pairs, dim3((unsigned)((n_docs + 3) / 4)), dim3(256), 0, st, indptr, idx, val, doc0,
                     n_docs, k_in.as<uint32_t>(), p_in.as<uint64_t>());
  HX_HIP(hipGetLastError());

  size_t tmp_bytes = 0;
  HX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in.as<uint32_t>(), k_out.as<uint32_t>(),
                                   p_in.as<uint64_t>(), p_out.as<uint64_t>(), (size_t)nnz, 0u, 31u, st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k_in.as<uint32_t>(), k_out.as<uint32_t>(),
                                     p_in.as<uint64_t>(), p_out.as<uint64_t>(), (size_t)nnz, 0u, 31u, st));
    HX_HIP(hipStreamSynchronize(st));
  }

  // run-length encode the sorted term ids -> live terms and their posting counts
  DevBuf uterms(nnz * 4), counts(nnz * 4), nruns(8);
  HX_HIP(rocprim::run_length_encode(nullptr, tmp_bytes, k_out.as<uint32_t>(), (unsigned int)nnz,
                                    uterms.as<uint32_t>(), counts.as<uint32_t>(), nruns.as<uint64_t>(), st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::run_length_encode(tmp.p, tmp_bytes, k_out.as<uint32_t>(), (unsigned int)nnz,
                                      uterms.as<uint32_t>(), counts.as<uint32_t>(), nruns.as<uint64_t>(),
                                      st));
    HX_HIP(hipStreamSynchronize(st));
  }
  uint64_t n_live = 0;
  HX_HIP(hipMemcpy(&n_live, nruns.p, 8, hipMemcpyDeviceToHost));

  DevBuf offs(n_live * 8);
  HX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, counts.as<uint32_t>(), offs.as<uint64_t>(),
                                 (uint64_t)0, (size_t)n_live, rocprim::plus<uint64_t>(), st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, counts.as<uint32_t>(), offs.as<uint64_t>(),
                                   (uint64_t)0, (size_t)n_live, rocprim::plus<uint64_t>(), st));
    HX_HIP(hipStreamSynchronize(st));
  }

  const int64_t ptr_entries = (int64_t)n_live * (nseg + 1);
  HX_HIP(hipMalloc((void**)&out->ptr, (size_t)ptr_entries * 4));
  hipLaunchKernelGGL(k_fill_ptr, dim3((unsigned)n_live), dim3(256), 0, st, p_out.as<uint64_t>(), offs.as<uint64_t>(),
                     counts.as<uint32_t>(), (int)nseg, seg_docs, out->ptr);
  HX_HIP(hipGetLastError());
  HX_HIP(hipMalloc((void**)&out->uterms, (size_t)n_live * 4));
  HX_HIP(hipMemcpyAsync(out->uterms, uterms.p, (size_t)n_live * 4, hipMemcpyDeviceToDevice, st));
  HX_HIP(hipMalloc((void**)&out->post, (size_t)(nnz + 2) * sizeof(uint2)));   // + padding: sparse2.hip loads pairs
  HX_HIP(hipMemsetAsync(out->post + nnz, 0, 2 * sizeof(uint2), st));
  // Measured (round 3, 10M x 1e9 postings, B = 1024; profiles/r03_pmc_sparse_select.json): the bank-spread order takes
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the select pass from 0.60 to 0.33 -- and its time from 2.040 to 2.035 ms:
  // the LDS pipe is active for 11 % of the kernel's cycles either way, the conflicts were never what the waves wait
  // for.  It costs 90 ms per 1e9 postings at build time, so it is OFF unless asked for (HX_SP_SPREAD=1).
  const char* sp_env = getenv("HX_SP_SPREAD");
  const bool spread = sp_env && atoi(sp_env) != 0;
  if (!spread) {
    hipLaunchKernelGGL(k_make_postings, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p_out.as<uint64_t>(),
                       nnz, (uint32_t)seg_docs, out->post);
    HX_HIP(hipGetLastError());
  } else {
    // every posting belongs to exactly one (term, segment) run: the two kernels write all of `post`
    DevBuf heavy((size_t)(nnz / SPREAD_HEAVY + 2) * 4);                      // a heavy term holds > SPREAD_HEAVY postings
    uint32_t* n_heavy = heavy.as<uint32_t>();                                 // [0] the count, then the list
    HX_HIP(hipMemsetAsync(n_heavy, 0, 4, st));
    hipLaunchKernelGGL(k_spread_light, dim3((unsigned)((n_live + 3) / 4)), dim3(256), 0, st, p_out.as<uint64_t>(),
                       offs.as<uint64_t>(), counts.as<uint32_t>(), out->ptr, (int64_t)n_live, (int)nseg, (uint32_t)seg_docs,
                       out->post, n_heavy + 1, n_heavy);
    HX_HIP(hipGetLastError());
    uint32_t nh = 0;
    HX_HIP(hipMemcpyAsync(&nh, n_heavy, 4, hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    if (nh) {
      hipLaunchKernelGGL(k_spread_heavy, dim3(nh, (unsigned)((nseg + 15) / 16)), dim3(256), 0, st, p_out.as<uint64_t>(),
                         out->ptr, n_heavy + 1, (int)nseg, (uint32_t)seg_docs, out->post);
      HX_HIP(hipGetLastError());
    }
  }
  HX_HIP(hipStreamSynchronize(st));
  out->n_live = (int64_t)n_live;
  out->ptr_entries = ptr_entries;
}

// ---- synthetic docs (oracle.synth_sparse_docs) ------------------------------------
__device__ __forceinline__ int zipf_rank(const uint32_t* cdf, int V, uint32_t u) {
  // #{r : cdf[r] <= u}  (searchsorted side="right"), clipped to V-1
  int lo = 0, hi = V;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo < V - 1 ? lo : V - 1;
}

__device__ __forceinline__ float bm25_w(int tf, int L) {
  const double k = 1.2, b = 0.75, avg = 256.0;
  const double t1 = 1.0 - b;
  const double t2 = (b * (double)L) / avg;
  const double den = (double)tf + k * (t1 + t2);
  return (float)(((double)tf * (k + 1.0)) / den);
}

template <bool FILL>
__global__ void k_synth_sparse(int64_t doc0, int64_t n, uint32_t seed, const uint32_t* cdf, int V,
                               const uint16_t* len_tab, int64_t* nnz_per_doc, const int64_t* indptr,
                               int32_t* idx, float* val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t d = (uint32_t)(doc0 + i);
  const int L = len_tab[hash2(seed, d, 0xFFFFFFFFu) & 255];
  int prev = -1, tf = 0;
  int64_t nn = 0;
  int64_t o = FILL ? indptr[i] : 0;
  for (int t = 0; t < L; ++t) {
    const uint64_t u = (((uint64_t)t << 32) + hash2(seed, d, (uint32_t)t)) / (uint64_t)L;
    const int rank = zipf_rank(cdf, V, (uint32_t)u);
    if (rank != prev) {
      if (prev >= 0) {
        if (FILL) {
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
  if (prev >= 0) {
    if (FILL) {
      idx[o] = (int32_t)(((uint32_t)prev * 0x9E3779B1u) & 0x7FFFFFFFu);
      val[o] = bm25_w(tf, L);
    }
    ++nn;
  }
  if (!FILL) nnz_per_doc[i] = nn;
}

void synth_sparse_count(int64_t doc0, int64_t n, uint32_t seed, const uint32_t* cdf, int V,
                        const uint16_t* len_tab, int64_t* nnz_per_doc, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_synth_sparse<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, doc0, n,
                     seed, cdf, V, len_tab, nnz_per_doc, nullptr, nullptr, nullptr);
  HX_HIP(hipGetLastError());
}
void synth_sparse_fill(int64_t doc0, int64_t n, uint32_t seed, const uint32_t* cdf, int V,
                       const uint16_t* len_tab, const int64_t* indptr, int32_t* idx, float* val,
                       hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_synth_sparse<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, doc0, n,
                     seed, cdf, V, len_tab, nullptr, indptr, idx, val);
  HX_HIP(hipGetLastError());
}

void exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t n, hipStream_t st) {
  // out has n+1 slots; out[n] = total.  Scan n+1 elements with a trailing zero input.
  size_t tmp_bytes = 0;
  HX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (int64_t)0, (size_t)(n + 1),
                                 rocprim::plus<int64_t>(), st));
  DevBuf tmp(tmp_bytes);
  HX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, (int64_t)0, (size_t)(n + 1),
                                 rocprim::plus<int64_t>(), st));
  HX_HIP(hipStreamSynchronize(st));
}

}  // namespace hx
