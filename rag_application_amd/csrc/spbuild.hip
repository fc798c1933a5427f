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

// sorted payloads -> postings {accumulator place of the document inside its segment, weight bits}
__global__ void k_make_postings(const uint64_t* pay, int64_t nnz, uint32_t seg_docs, uint2* post) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const uint64_t p = pay[i];
  // first word: where the document's 16-bit accumulator lives in the select pass's LDS (sparse2.hip: sp_word) --
  // byte offset of its 32-bit word (document mod seg_docs/2, times 4) | shift of its half (0 / 16) << 24
  const uint32_t d = (uint32_t)(p >> 32) % seg_docs, words = seg_docs >> 1;
  post[i] = make_uint2(((d & (words - 1)) << 2) | (d >= words ? 16u << 24 : 0u), (uint32_t)p);
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
  hipLaunchKernelGGL(k_make_postings, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p_out.as<uint64_t>(),
                     nnz, (uint32_t)seg_docs, out->post);
  HX_HIP(hipGetLastError());
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
