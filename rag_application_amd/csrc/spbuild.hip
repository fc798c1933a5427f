// K9 -- on-device inverted-index build, plus the synthetic sparse corpus generator.
//
// Mirrors Qdrant's sparse index build on upsert of the "sparse" named vector
// (app/core/vector_store/qdrant/qdrant_handler.py:80-86, 163, 190-193).  Input is the
// doc-major CSR the ingest path appends; output is the segment-major, term-sorted
// posting store and the (segment, term) -> (offset, length) table of sparse.hip.
// The sort itself is rocPRIM's LSD radix sort (stable, so documents stay ascending
// inside a posting run); everything around it is hand-written.
#include "hx_common.hpp"
#include "kernels.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

namespace hx {

__host__ __device__ inline uint64_t sp_hash_b(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t bytes) { HX_HIP(hipMalloc(&p, bytes ? bytes : 16)); }
  ~DevBuf() { (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  template <typename T>
  T* as() { return (T*)p; }
};

__global__ void k_make_pairs(const int64_t* indptr, const int32_t* idx, const float* val, int64_t n_docs,
                             uint64_t* keys, uint64_t* pay) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (d >= n_docs) return;
  const int64_t b = indptr[d], e = indptr[d + 1];
  const uint64_t seg = (uint64_t)(d / SEG_DOCS), dl = (uint64_t)(d % SEG_DOCS);
  for (int64_t i = b + lane; i < e; i += 64) {
    keys[i] = (seg << 31) | (uint64_t)(uint32_t)idx[i];
    uint32_t wb;
    const float w = val[i];
    __builtin_memcpy(&wb, &w, 4);
    pay[i] = (dl << 32) | wb;
  }
}

__global__ void k_split(const uint64_t* pay, int64_t nnz, uint16_t* doc_local, float* w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const uint64_t p = pay[i];
  doc_local[i] = (uint16_t)(p >> 32);
  const uint32_t wb = (uint32_t)p;
  float f;
  __builtin_memcpy(&f, &wb, 4);
  w[i] = f;
}

__global__ void k_table_clear(SpHashEntry* t, uint64_t cap) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    t[i].key = ~0ull;
    t[i].off = 0;
    t[i].len = 0;
  }
}

__global__ void k_table_insert(const uint64_t* ukeys, const uint32_t* counts, const uint64_t* offs,
                               int64_t n_groups, SpHashEntry* t, uint64_t mask) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint64_t key = ukeys[g];
  uint64_t slot = sp_hash_b(key) & mask;
  while (true) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&t[slot].key, ~0ull, (unsigned long long)key);
    if (prev == ~0ull) break;  // keys are unique: nobody else inserts `key`
    slot = (slot + 1) & mask;
  }
  t[slot].off = (uint32_t)offs[g];
  t[slot].len = counts[g];
}

void build_sparse_index(const int64_t* indptr, const int32_t* idx, const float* val, int64_t n_docs,
                        int64_t nnz, SparseBuildOut* out, hipStream_t st) {
  out->doc_local = nullptr;
  out->w = nullptr;
  out->table = nullptr;
  out->table_cap = 0;
  out->n_groups = 0;
  if (nnz <= 0 || n_docs <= 0) return;
  HX_CHECK(nnz < (int64_t)0xFFFFFFFFll, "sparse index: nnz per shard must be < 2^32");
  const int64_t nseg = (n_docs + SEG_DOCS - 1) / SEG_DOCS;
  int seg_bits = 1;
  while ((1ll << seg_bits) < nseg) ++seg_bits;

  DevBuf k_in(nnz * 8), k_out(nnz * 8), p_in(nnz * 8), p_out(nnz * 8);
  hipLaunchKernelGGL(k_make_pairs, dim3((unsigned)((n_docs + 3) / 4)), dim3(256), 0, st, indptr, idx, val,
                     n_docs, k_in.as<uint64_t>(), p_in.as<uint64_t>());
  HX_HIP(hipGetLastError());

  size_t tmp_bytes = 0;
  HX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(),
                                   p_in.as<uint64_t>(), p_out.as<uint64_t>(), (size_t)nnz, 0u,
                                   (unsigned)(31 + seg_bits), st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k_in.as<uint64_t>(), k_out.as<uint64_t>(),
                                     p_in.as<uint64_t>(), p_out.as<uint64_t>(), (size_t)nnz, 0u,
                                     (unsigned)(31 + seg_bits), st));
    HX_HIP(hipStreamSynchronize(st));
  }

  // run-length encode the sorted keys -> groups
  DevBuf ukeys(nnz * 8), counts(nnz * 4), nruns(8);
  HX_HIP(rocprim::run_length_encode(nullptr, tmp_bytes, k_out.as<uint64_t>(), (unsigned int)nnz,
                                    ukeys.as<uint64_t>(), counts.as<uint32_t>(), nruns.as<uint64_t>(), st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::run_length_encode(tmp.p, tmp_bytes, k_out.as<uint64_t>(), (unsigned int)nnz,
                                      ukeys.as<uint64_t>(), counts.as<uint32_t>(), nruns.as<uint64_t>(),
                                      st));
    HX_HIP(hipStreamSynchronize(st));
  }
  uint64_t n_groups = 0;
  HX_HIP(hipMemcpy(&n_groups, nruns.p, 8, hipMemcpyDeviceToHost));

  DevBuf offs(n_groups * 8);
  HX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, counts.as<uint32_t>(), offs.as<uint64_t>(),
                                 (uint64_t)0, (size_t)n_groups, rocprim::plus<uint64_t>(), st));
  {
    DevBuf tmp(tmp_bytes);
    HX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, counts.as<uint32_t>(), offs.as<uint64_t>(),
                                   (uint64_t)0, (size_t)n_groups, rocprim::plus<uint64_t>(), st));
    HX_HIP(hipStreamSynchronize(st));
  }

  uint64_t cap = 1024;
  while (cap < 2 * n_groups) cap <<= 1;
  HX_HIP(hipMalloc((void**)&out->table, cap * sizeof(SpHashEntry)));
  hipLaunchKernelGGL(k_table_clear, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, out->table, cap);
  hipLaunchKernelGGL(k_table_insert, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st,
                     ukeys.as<uint64_t>(), counts.as<uint32_t>(), offs.as<uint64_t>(), (int64_t)n_groups,
                     out->table, cap - 1);
  HX_HIP(hipGetLastError());

  HX_HIP(hipMalloc((void**)&out->doc_local, nnz * sizeof(uint16_t)));
  HX_HIP(hipMalloc((void**)&out->w, nnz * sizeof(float)));
  hipLaunchKernelGGL(k_split, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p_out.as<uint64_t>(), nnz,
                     out->doc_local, out->w);
  HX_HIP(hipGetLastError());
  HX_HIP(hipStreamSynchronize(st));
  out->table_cap = cap;
  out->n_groups = (int64_t)n_groups;
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
