// In-register bitonic sorting network of one wave: 64 E 64-bit keys, E per lane (E = 4: 256 keys, E = 8: 512),
// index i = lane * E + e, descending.  Strides below E are register swaps, the rest lane exchanges (no LDS traffic
// of its own, no barrier).  Used by k_compact_top (select.hip) and the candidate cut of k_sparse_select
// (sparse2.hip).  E is deduced from the register array.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hx {

__device__ __forceinline__ uint64_t k64max(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t k64min(uint64_t a, uint64_t b) { return a > b ? b : a; }
// compare-exchange stage (k, j) of the descending bitonic network over i = lane * E + e
template <int K, int J, int E>
__device__ __forceinline__ void w_cx(uint64_t (&v)[E], int lane) {
  if constexpr (J < E) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if ((e & J) == 0) {
        const bool desc = K < E ? ((e & K) == 0) : (((lane * E) & K) == 0);
        const uint64_t mx = k64max(v[e], v[e ^ J]), mn = k64min(v[e], v[e ^ J]);
        v[e] = desc ? mx : mn;
        v[e ^ J] = desc ? mn : mx;
      }
    }
  } else {
    constexpr int LM = J / E;
    const bool lower = (lane & LM) == 0;
    const bool desc = ((lane * E) & K) == 0;
    const bool take_max = lower == desc;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint64_t y = (uint64_t)__shfl_xor((unsigned long long)v[e], LM, 64);
      v[e] = take_max ? k64max(v[e], y) : k64min(v[e], y);
    }
  }
}
template <int K, int J, int E>
__device__ __forceinline__ void w_merge(uint64_t (&v)[E], int lane) {
  w_cx<K, J>(v, lane);
  if constexpr (J > 1) w_merge<K, J / 2>(v, lane);
}
template <int K, int E>
__device__ __forceinline__ void w_sort(uint64_t (&v)[E], int lane) {
  if constexpr (K > 2) w_sort<K / 2>(v, lane);
  w_merge<K, K / 2>(v, lane);
}

}  // namespace hx
