// Candidate-list kernels: exact re-scoring (K6), top-L compaction, certificate,
// reciprocal-rank fusion (K8), key packing.  All rankings use the 64-bit key whose
// descending order is (score desc, id asc) -- oracle/oracle.py order_key.
#include <cstdlib>
#include "hx_common.hpp"
#include "kernels.hpp"
#include "wsort.hpp"

#include <mutex>

namespace hx {

// ---------------------------------------------------------------------------------
// compaction: bitonic sort (descending) of <= 8192 keys in LDS, optional dedupe
// ---------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_compact(const uint64_t* __restrict__ keys, int stride,
                                                 const int* __restrict__ in_cnt, int Pmax, int keep,
                                                 int dedupe, uint64_t* out_keys, int out_stride,
                                                 int* out_cnt, float* tau, int tau_rank, int chk_rank,
                                                 int* kept_io, int* underflow) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint64_t* sk = (uint64_t*)smem;
  __shared__ int part[NT];
  __shared__ uint64_t s_kth;
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = in_cnt ? in_cnt[b] : stride;
  const int n_raw = n;
  n = n < stride ? n : stride;
  n = n < Pmax ? n : Pmax;
  int P = 256;                       // sort size of THIS list: next power of two >= n (<= Pmax)
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += NT) sk[i] = i < n ? keys[(int64_t)b * stride + i] : 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += NT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = sk[i], y = sk[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            sk[i] = y;
            sk[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  // flags + block scan (thread t owns `per` consecutive slots from t*per; threads past P idle)
  const int per = P >= NT ? P / NT : 1;
  const int base = tid * per;
  const int mine = base < P ? per : 0;
  int c = 0;
  for (int e = 0; e < mine; ++e) {
    const int i = base + e;
    const uint64_t x = sk[i];
    const bool f = x != 0ull && (!dedupe || i == 0 || x != sk[i - 1]);
    c += f;
  }
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < NT; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int total = part[NT - 1];
  int pos = part[tid] - c;
  uint64_t* o = out_keys + (int64_t)b * out_stride;
  // read everything we need before any (possibly aliasing) write: it is all in LDS.
  if (tid == 0) s_kth = 0ull;
  __syncthreads();
  for (int e = 0; e < mine; ++e) {
    const int i = base + e;
    const uint64_t x = sk[i];
    const bool f = x != 0ull && (!dedupe || i == 0 || x != sk[i - 1]);
    if (f) {
      if (pos < keep) o[pos] = x;
      if (pos == tau_rank - 1) s_kth = x;
      ++pos;
    }
  }
  const int kept = total < keep ? total : keep;
  for (int i = kept + tid; i < out_stride; i += NT) o[i] = 0ull;  // empty slots are 0
  __syncthreads();
  if (tid == 0) {
    out_cnt[b] = kept;
    if (tau) tau[b] = (total >= keep && keep > 0) ? key_score(s_kth) : -__builtin_inff();
    if (kept_io) {
      // The keys appended since the previous compaction passed the score of its rank chk_rank.  The
      // kept list is the exact top-`keep` only if at least `keep` rows reach that score: chk_rank of
      // the old list do, plus every appended one (engine.hip, chunked_scan).
      const int prev = kept_io[b];
      if (chk_rank > 0 && prev >= keep && chk_rank + (n_raw - prev) < keep) underflow[b] = 1;
      kept_io[b] = kept;
    }
  }
}

// ---------------------------------------------------------------------------------
// compaction, short form: keep <= 64 E of <= NW * 64 E keys, no dedupe (E = 4: 256 of <= 2048 -- every
// compaction of the fp16-candidate step; E = 8: 512 of <= 8192 -- the int8 candidate pass keeps more
// candidates).  Same results as k_compact, but no block-wide sort: every wave sorts 64 E keys in registers
// (E per lane; strides below E are register swaps, the rest lane exchanges -- no LDS traffic of its own, no
// barrier), then log2(NW) rounds fold the waves pairwise: max(A[i], B[64 E - 1 - i]) of two descending runs
// is a bitonic run holding the best 64 E of both, sorted again by the last stages.  One barrier per round
// instead of one per stage (66 at P = 2048).
// ---------------------------------------------------------------------------------
// The fold itself, for a block of exactly NW waves: src[0, n) (n <= NW * 64 E) -> wave 0 holds the best 64 E keys in v,
// sorted (index lane * E + e); returns the number of non-empty keys.  Every thread of the block must call it.
template <int NW, int E, typename LOAD>
__device__ __forceinline__ int compact_top_core(LOAD&& load, int n, uint64_t (&v)[E], int lane, int w) {
  constexpr int R = 64 * E;          // keys per wave
  // a wave hands its run to its partner through slot w / 2: the only earlier reader of that slot is the wave itself
  // (as the partner of wave w + 1 in the first round), so NW / 2 slots serve every round
  __shared__ uint64_t buf[NW > 1 ? (NW / 2) * R : 1];
  __shared__ int s_tot[NW];
  int tot = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {   // the order inside an unsorted run is free: coalesced loads
    const int i = w * R + e * 64 + lane;
    v[e] = i < n ? load(i) : 0ull;
    tot += __popcll(__ballot(v[e] != 0ull));
  }
  if (n > w * R) w_sort<R>(v, lane);
  if (NW > 1) {
    if (lane == 0) s_tot[w] = tot;
#pragma unroll
    for (int s = 0; (1 << s) < NW; ++s) {
      const int m = (2 << s) - 1;
      if ((w & m) == (1 << s)) {
#pragma unroll
        for (int e = 0; e < E; ++e) buf[(w >> 1) * R + lane * E + e] = v[e];
      }
      __syncthreads();   // also: every wave's loads have landed before any store below (in-place use)
      const int pw = w + (1 << s);
      if ((w & m) == 0 && n > pw * R) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = k64max(v[e], buf[(pw >> 1) * R + (R - 1) - (lane * E + e)]);
        w_merge<R, R / 2>(v, lane);
      }
    }
    tot = 0;
#pragma unroll
    for (int x = 0; x < NW; ++x) tot += s_tot[x];
  }
  return tot;
}

template <int NW, int E>
__global__ __launch_bounds__(NW * 64) void k_compact_top(const uint64_t* __restrict__ keys, int stride,
                                                         const int* __restrict__ in_cnt, int keep,
                                                         uint64_t* out_keys, int out_stride, int* out_cnt,
                                                         float* tau, int tau_rank, int chk_rank, int* kept_io,
                                                         int* underflow) {
  constexpr int R = 64 * E;          // keys per wave
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = in_cnt ? in_cnt[b] : stride;
  const int n_raw = n;
  n = n < stride ? n : stride;
  n = n < NW * R ? n : NW * R;
  const uint64_t* src = keys + (int64_t)b * stride;
  uint64_t v[E];
  const int tot = compact_top_core<NW, E>([&](int i) { return src[i]; }, n, v, lane, w);
  uint64_t* o = out_keys + (int64_t)b * out_stride;
  const int kept = tot < keep ? tot : keep;
  for (int i = R + tid; i < out_stride; i += NW * 64) o[i] = 0ull;
  if (w == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = lane * E + e;
      if (i < out_stride) o[i] = i < kept ? v[e] : 0ull;   // empty slots are 0
    }
    const int kr = tau_rank - 1;
    uint64_t mine = v[0];
#pragma unroll
    for (int e = 1; e < E; ++e) mine = (kr & (E - 1)) == e ? v[e] : mine;
    const uint64_t kth = (uint64_t)__shfl((unsigned long long)mine, kr / E, 64);
    if (lane == 0) {
      out_cnt[b] = kept;
      if (tau) tau[b] = (tot >= keep && keep > 0) ? key_score(kth) : -__builtin_inff();
      if (kept_io) {   // see k_compact
        const int prev = kept_io[b];
        if (chk_rank > 0 && prev >= keep && chk_rank + (n_raw - prev) < keep) underflow[b] = 1;
        kept_io[b] = kept;
      }
    }
  }
}

void launch_compact(uint64_t* keys, int stride, const int* in_cnt, int B, int keep, int dedupe,
                    uint64_t* out_keys, int out_stride, int* out_cnt, float* tau, int max_cnt_hint,
                    hipStream_t st, int tau_rank, int chk_rank, int* kept_io, int* underflow) {
  if (B <= 0) return;
  int m = max_cnt_hint < stride ? max_cnt_hint : stride;
  int P = next_pow2(m < 256 ? 256 : m);
  HX_CHECK(P <= CAND_CAP, "compact: list longer than CAND_CAP");
  HX_CHECK(keep <= out_stride, "compact: keep > out_stride");
#define HX_TOP(NW, E)                                                                                            \
  hipLaunchKernelGGL((k_compact_top<NW, E>), dim3(B), dim3(NW * 64), 0, st, keys, stride, in_cnt, keep, out_keys, \
                     out_stride, out_cnt, tau, tau_rank, chk_rank, kept_io, underflow)
  if (!dedupe && keep >= 1 && keep <= 256 && P <= 2048) {   // short form; the tau rank is <= keep
    if (tau_rank <= 0 || tau_rank > keep) tau_rank = keep;
    HX_CHECK(!kept_io || underflow, "compact: kept_io without an underflow flag array");
    if (P <= 256) HX_TOP(1, 4);
    else if (P <= 512) HX_TOP(2, 4);
    else if (P <= 1024) HX_TOP(4, 4);
    else HX_TOP(8, 4);
    HX_HIP(hipGetLastError());
    return;
  }
  if (!dedupe && keep > 256 && keep <= 512 && P <= 8192) {  // ... with 8 keys per lane: the int8 candidate pass
    if (tau_rank <= 0 || tau_rank > keep) tau_rank = keep;
    HX_CHECK(!kept_io || underflow, "compact: kept_io without an underflow flag array");
    if (P <= 512) HX_TOP(1, 8);
    else if (P <= 1024) HX_TOP(2, 8);
    else if (P <= 2048) HX_TOP(4, 8);
    else if (P <= 4096) HX_TOP(8, 8);
    else HX_TOP(16, 8);
    HX_HIP(hipGetLastError());
    return;
  }
#undef HX_TOP
  {   // the dynamic-LDS opt-in is a property of the function ON A DEVICE: once per device, under a lock
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    HX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      HX_HIP(hipFuncSetAttribute((const void*)k_compact<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 CAND_CAP * 8));
      HX_HIP(hipFuncSetAttribute((const void*)k_compact<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 CAND_CAP * 8));
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  if (tau_rank <= 0 || tau_rank > keep) tau_rank = keep;
  HX_CHECK(!kept_io || underflow, "compact: kept_io without an underflow flag array");
  if (P >= 1024)   // long lists: 1024 threads per list (16 of the 66 sort stages of P = 2048 per barrier)
    hipLaunchKernelGGL(k_compact<1024>, dim3(B), dim3(1024), (size_t)P * 8, st, keys, stride, in_cnt, P, keep,
                       dedupe, out_keys, out_stride, out_cnt, tau, tau_rank, chk_rank, kept_io, underflow);
  else
    hipLaunchKernelGGL(k_compact<256>, dim3(B), dim3(256), (size_t)P * 8, st, keys, stride, in_cnt, P, keep,
                       dedupe, out_keys, out_stride, out_cnt, tau, tau_rank, chk_rank, kept_io, underflow);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// spec arithmetic on one wave (oracle.spec_dot / i8_scores)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_spec_dot(const float* __restrict__ x, const float* __restrict__ q,
                                               int dim_pad, int lane) {
  float p = 0.0f;
  if (dim_pad <= 1024) {
    // all loads of the row first (a row is one dependent round trip, not dim_pad / 64 of them); the
    // sum keeps the order of the spec
    float xv[16], qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool in = i * 64 < dim_pad;
      xv[i] = in ? x[i * 64 + lane] : 0.0f;
      qv[i] = in ? q[i * 64 + lane] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i * 64 < dim_pad) p = __fadd_rn(p, __fmul_rn(xv[i], qv[i]));
  } else {
    for (int j = 0; j < dim_pad; j += 64) p = __fadd_rn(p, __fmul_rn(x[j + lane], q[j + lane]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_down(p, off, 64));
  return __fadd_rn(p, 0.0f);  // lane 0 holds the result
}

__device__ __forceinline__ int wave_i8_dot(const int8_t* __restrict__ x, const int8_t* __restrict__ q,
                                           int dim_pad8, int lane) {
  int acc = 0;
  for (int j = lane * 4; j < dim_pad8; j += 256) {
    const int xa = *(const int*)(x + j), qa = *(const int*)(q + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += (int)(int8_t)(xa >> (8 * e)) * (int)(int8_t)(qa >> (8 * e));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
  return acc;
}

__device__ __forceinline__ uint64_t exact_key(const RescoreArgs& a, int b, int64_t local, int lane) {
  float s;
  if (a.kind == KIND_F32) {
    const float* x = (const float*)a.M + local * a.row_stride;
    const float* q = (const float*)a.Q + (int64_t)b * a.q_stride;
    s = wave_spec_dot(x, q, a.dim_pad, lane);
  } else {
    const int8_t* x = (const int8_t*)a.M + local * a.row_stride;
    const int8_t* q = (const int8_t*)a.Q + (int64_t)b * a.q_stride;
    const int d = wave_i8_dot(x, q, (int)a.row_stride, lane);
    s = __fmul_rn(__fmul_rn((float)d, a.rinv_x[local]), a.rinv_q[b]);
  }
  return make_key(s, (uint32_t)(a.id_base + local));
}

__global__ __launch_bounds__(256) void k_rescore_list(RescoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (i >= a.stride) return;
  int n = a.cnt ? a.cnt[b] : a.stride;
  n = n < a.stride ? n : a.stride;
  uint64_t out = 0ull;
  if (i < n) {
    const uint64_t ck = a.cand[(int64_t)b * a.stride + i];
    if (ck != 0ull) {
      const int64_t local = (int64_t)key_id(ck) - a.id_base;
      if (local >= 0 && local < a.n_rows) out = exact_key(a, b, local, lane);
    }
  }
  if (lane == 0) a.out[(int64_t)b * a.stride + i] = out;
}

// The same for a list most of whose slots belong to OTHER shards (the global candidate list of the candidates-first
// exchange: one slot in `world` is this shard's).  A wave takes 64 slots, finds its own by one ballot and scores only
// those -- k_rescore_list starts a wave per slot, 7 of 8 of which find a foreign id and leave.  Slots that are not this
// shard's are NOT written (the caller zeroed `out`).
__global__ __launch_bounds__(256) void k_rescore_own(RescoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const int b = blockIdx.y;
  int n = a.cnt ? a.cnt[b] : a.stride;
  n = n < a.stride ? n : a.stride;
  if (c0 >= n) return;
  const int i = c0 + lane;
  const uint64_t ck = i < n ? a.cand[(int64_t)b * a.stride + i] : 0ull;
  const int64_t local = (int64_t)key_id(ck) - a.id_base;
  const bool own = ck != 0ull && local >= 0 && local < a.n_rows;
  unsigned long long m = __ballot(own);
  while (m) {                                          // wave-uniform
    const int l = __builtin_ctzll(m);
    m &= m - 1ull;
    const int64_t row = __shfl((long long)local, l, 64);
    const uint64_t k = exact_key(a, b, row, lane);
    if (lane == 0) a.out[(int64_t)b * a.stride + c0 + l] = k;
  }
}
void launch_rescore_own(const RescoreArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.stride <= 0) return;
  const int m = (a.max_cnt > 0 && a.max_cnt < a.stride) ? a.max_cnt : a.stride;
  hipLaunchKernelGGL(k_rescore_own, dim3((m + 255) / 256, a.B), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

void launch_rescore_list(const RescoreArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.stride <= 0) return;
  const int m = (a.max_cnt > 0 && a.max_cnt < a.stride) ? a.max_cnt : a.stride;
  hipLaunchKernelGGL(k_rescore_list, dim3((m + 3) / 4, a.B), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// the dense stage's finish in ONE launch: exact re-score of the candidates, top-L, certificate
// ---------------------------------------------------------------------------------
// grid (NB, B).  Every block scores its share of query b's candidates (one wave per row, as k_rescore_list) into `tmp`;
// the block that finishes LAST for the query (a counter per query, device scope) sorts the <= 512 exact keys in one
// wave's registers (wsort.hpp, 8 per lane), writes the top L and applies the certificate of k_certify in place.  Before:
// k_rescore_list + k_compact_top + k_certify, three launches behind every scan -- at B <= 32 their launch gaps were a
// tenth of the pass.
struct FinishArgs {
  RescoreArgs r;           // cand = the candidate keys (approximate scores, best first), cnt, stride; out = tmp [B x stride]
  int lprime;              // candidates the approximate pass keeps when its list is full
  int L;
  uint64_t* out_keys;      // [B x L]
  int* out_cnt;            // [B]
  const int* overflow;     // [B]
  float eps;
  const float* eps_q;      // optional [B]
  int* fail;               // [B]
  int* nfail;              // += failed queries
  unsigned int* done;      // [B] zero before the launch; left zero
};
template <int E>
__global__ __launch_bounds__(256) void k_dense_finish(FinishArgs f) {
  const RescoreArgs& a = f.r;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  int n = a.cnt ? a.cnt[b] : a.stride;
  n = n < a.stride ? n : a.stride;
  n = n < 512 ? n : 512;
  uint64_t* tmp = a.out + (int64_t)b * a.stride;
  for (int i = blockIdx.x * 4 + w; i < n; i += (int)gridDim.x * 4) {      // wave-uniform
    const uint64_t ck = a.cand[(int64_t)b * a.stride + i];
    uint64_t k = 0ull;
    if (ck != 0ull) {
      const int64_t local = (int64_t)key_id(ck) - a.id_base;
      if (local >= 0 && local < a.n_rows) k = exact_key(a, b, local, lane);
    }
    if (lane == 0) __hip_atomic_store(tmp + i, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ int s_last;
  // The keys were stored with agent scope (sc1: written through, no dirty line stays in this XCD's L2), so "visible
  // device-wide" is "acknowledged": s_waitcnt vmcnt(0).  (A __threadfence() here is a write-back of the whole L2 per
  // block -- 2400 of them at B = 32 cost 0.15 ms, more than the three launches this kernel replaces.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(f.done + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1 : 0;
  __syncthreads();
  if (!s_last) return;                               // (block-uniform)
  // the last block of the query: top-L of the <= 512 exact keys, its four waves folding as k_compact_top<4, E>
  uint64_t v[E];
  const int tot = compact_top_core<4, E>(
      [&](int i) { return __hip_atomic_load(tmp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, n, v, lane, w);
  const int kept = tot < f.L ? tot : f.L;
  uint64_t* o = f.out_keys + (int64_t)b * f.L;
  for (int i = 64 * E + (int)threadIdx.x; i < f.L; i += 256) o[i] = 0ull;
  if (w != 0) return;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = lane * E + e;
    if (i < f.L) o[i] = i < kept ? v[e] : 0ull;
  }
  // the L-th exact key
  const int kr = f.L - 1;
  uint64_t mine = v[0];
#pragma unroll
  for (int e = 1; e < E; ++e) mine = (kr & (E - 1)) == e ? v[e] : mine;
  const uint64_t kL = (uint64_t)__shfl((unsigned long long)mine, (kr / E) & 63, 64);
  if (lane == 0) {
    f.out_cnt[b] = kept;
    float eps = f.eps_q ? f.eps_q[b] : f.eps;
    bool bad = f.overflow[b] != 0;
    const int ac = a.cnt ? a.cnt[b] : a.stride;
    if (!bad && ac >= f.lprime) {                    // the candidate list is full: rows outside it score <= m
      const float m = key_score(a.cand[(int64_t)b * a.stride + f.lprime - 1]);
      if (tot < f.L) bad = true;                     // cannot happen (lprime > L distinct rows), be safe
      else bad = !(__fadd_rn(m, eps) < key_score(kL));
    }
    f.fail[b] = bad ? 1 : 0;
    if (bad) atomicAdd(f.nfail, 1);
    f.done[b] = 0u;                                  // for the next launch
  }
}
bool launch_dense_finish(const RescoreArgs& r, int lprime, int L, uint64_t* out_keys, int* out_cnt, const int* overflow,
                         float eps, const float* eps_q, int* fail, int* nfail, unsigned int* done, hipStream_t st) {
  if (lprime > 512 || L > 512 || L < 1 || r.stride < lprime) return false;
  if (r.B <= 0) return true;
  FinishArgs f{};
  f.r = r;
  f.lprime = lprime;
  f.L = L;
  f.out_keys = out_keys;
  f.out_cnt = out_cnt;
  f.overflow = overflow;
  f.eps = eps;
  f.eps_q = eps_q;
  f.fail = fail;
  f.nfail = nfail;
  f.done = done;
  // a small batch gets a wave per candidate (the chip is otherwise idle), a large one has blocks enough
  static const int nb_env = getenv("HX_DEBUG_FINISH_NB") ? atoi(getenv("HX_DEBUG_FINISH_NB")) : 0;
  int nb = r.B <= 64 ? (lprime + 3) / 4 : 16;
  if (nb_env > 0) nb = nb_env;
  // E keys per lane: the top 64 E end in wave 0 (L <= 64 E), 4 x 64 E keys fit the fold (>= 512 from E = 2)
  if (L <= 128) hipLaunchKernelGGL(k_dense_finish<2>, dim3(nb, r.B), dim3(256), 0, st, f);
  else if (L <= 256) hipLaunchKernelGGL(k_dense_finish<4>, dim3(nb, r.B), dim3(256), 0, st, f);
  else hipLaunchKernelGGL(k_dense_finish<8>, dim3(nb, r.B), dim3(256), 0, st, f);
  HX_HIP(hipGetLastError());
  return true;
}

__global__ __launch_bounds__(256) void k_rescore_range(RangeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = a.row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.row_end) return;
  const int f = blockIdx.y;          // slot of this query in the fallback buffers
  const int b = a.qsel[f];           // its index in the query batch
  const uint64_t k = exact_key(a.r, b, row, lane);
  if (lane == 0) a.r.out[(int64_t)f * a.r.stride + a.slot0 + (row - a.row_begin)] = k;
}

void launch_rescore_range(const RangeArgs& a, hipStream_t st) {
  const int64_t n = a.row_end - a.row_begin;
  if (n <= 0 || a.nsel <= 0) return;
  hipLaunchKernelGGL(k_rescore_range, dim3((unsigned)((n + 3) / 4), a.nsel), dim3(256), 0, st, a);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// certificate: nothing outside the candidate set can reach the exact L-th score
// ---------------------------------------------------------------------------------
__global__ void k_certify(const uint64_t* approx_keys, int approx_stride, const int* approx_cnt,
                          int lprime, const uint64_t* exact_keys, int exact_stride,
                          const int* exact_cnt, int L, const int* overflow, float eps, int B, int* fail,
                          int* nfail, const float* eps_q) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (eps_q) eps = eps_q[b];
  bool bad = overflow[b] != 0;
  if (!bad && approx_cnt[b] >= lprime) {
    // candidate list is full: rows outside it have approx score <= m
    const float m = key_score(approx_keys[(int64_t)b * approx_stride + lprime - 1]);
    const int ec = exact_cnt[b];
    if (ec < L) {
      bad = true;  // cannot happen (lprime > L distinct rows), be safe
    } else {
      const float eL = key_score(exact_keys[(int64_t)b * exact_stride + L - 1]);
      bad = !(__fadd_rn(m, eps) < eL);
    }
  }
  fail[b] = bad ? 1 : 0;
  if (bad) atomicAdd(nfail, 1);
}

void launch_certify(const uint64_t* approx_keys, int approx_stride, const int* approx_cnt, int lprime,
                    const uint64_t* exact_keys, int exact_stride, const int* exact_cnt, int L,
                    const int* overflow, float eps, int B, int* fail, int* nfail, hipStream_t st,
                    const float* eps_q) {
  hipLaunchKernelGGL(k_certify, dim3((B + 255) / 256), dim3(256), 0, st, approx_keys, approx_stride,
                     approx_cnt, lprime, exact_keys, exact_stride, exact_cnt, L, overflow, eps, B, fail,
                     nfail, eps_q);
  HX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------
// reciprocal-rank fusion of two ranked lists -> unsorted fused keys (then compact)
// ---------------------------------------------------------------------------------
// IEEE fp32 1/x through fp64 (innocuous double rounding)
__device__ __forceinline__ float rcp_f32_rn(float x) { return (float)(1.0 / (double)x); }

__global__ __launch_bounds__(256) void k_rrf(const uint64_t* a, int a_stride, const int* a_cnt,
                                             const uint64_t* b, int b_stride, const int* b_cnt, float k,
                                             int rank_base, uint64_t* out) {
  const int q = blockIdx.x, tid = threadIdx.x;
  int na = a_cnt[q], nb = b_cnt[q];
  na = na < a_stride ? na : a_stride;
  nb = nb < b_stride ? nb : b_stride;
  const uint64_t* la = a + (int64_t)q * a_stride;
  const uint64_t* lb = b + (int64_t)q * b_stride;
  uint64_t* o = out + (int64_t)q * (a_stride + b_stride);
  // Lists of up to RRF_LDS entries: the ids of both go to LDS first (coalesced loads); a thread's search of the OTHER list
  // was a chain of dependent global loads, one per entry until the match (29 us per 1024-query batch of two 100-entry lists).
  constexpr int RRF_LDS = 1024;
  __shared__ uint32_t ida[RRF_LDS], idb[RRF_LDS];
  if (na <= RRF_LDS && nb <= RRF_LDS) {
    for (int i = tid; i < na; i += 256) ida[i] = key_id(la[i]);
    for (int j = tid; j < nb; j += 256) idb[j] = key_id(lb[j]);
    __syncthreads();
    for (int i = tid; i < a_stride; i += 256) {
      uint64_t r = 0ull;
      if (i < na) {
        const uint32_t id = ida[i];
        float s = __fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(i + rank_base), k)));
        for (int j = 0; j < nb; ++j)
          if (idb[j] == id) {
            s = __fadd_rn(s, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k)));
            break;
          }
        r = make_key(s, id);
      }
      o[i] = r;
    }
    for (int j = tid; j < b_stride; j += 256) {
      uint64_t r = 0ull;
      if (j < nb) {
        const uint32_t id = idb[j];
        bool dup = false;
        for (int i = 0; i < na; ++i)
          if (ida[i] == id) {
            dup = true;
            break;
          }
        if (!dup) r = make_key(__fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k))), id);
      }
      o[a_stride + j] = r;
    }
    return;
  }
  for (int i = tid; i < a_stride; i += 256) {
    uint64_t r = 0ull;
    if (i < na) {
      const uint32_t id = key_id(la[i]);
      float s = __fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(i + rank_base), k)));
      for (int j = 0; j < nb; ++j)
        if (key_id(lb[j]) == id) {
          s = __fadd_rn(s, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k)));
          break;
        }
      r = make_key(s, id);
    }
    o[i] = r;
  }
  for (int j = tid; j < b_stride; j += 256) {
    uint64_t r = 0ull;
    if (j < nb) {
      const uint32_t id = key_id(lb[j]);
      bool dup = false;
      for (int i = 0; i < na; ++i)
        if (key_id(la[i]) == id) {
          dup = true;
          break;
        }
      if (!dup) r = make_key(__fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k))), id);
    }
    o[a_stride + j] = r;
  }
}

// The two usual lists (<= 256 keys together): fusion AND the top-`limit` in one kernel -- the fused keys go to LDS, wave 0
// sorts them in registers (wsort.hpp) and writes the first `limit`.  (k_rrf + a one-wave k_compact_top behind it were two
// launches and a round trip of the fused keys through memory: 24 + 25 us per 1024-query batch, 50 + 48 beside a busy
// second stream.)  Same arithmetic and order as k_rrf: a-list entries first, b-only entries after, (score desc, id asc).
__global__ __launch_bounds__(256) void k_rrf_top(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b,
                                                 int b_stride, const int* b_cnt, float k, int rank_base, int limit,
                                                 uint64_t* out, int out_stride, int* out_cnt) {
  const int q = blockIdx.x, tid = threadIdx.x;
  int na = a_cnt[q], nb = b_cnt[q];
  na = na < a_stride ? na : a_stride;
  nb = nb < b_stride ? nb : b_stride;
  const uint64_t* la = a + (int64_t)q * a_stride;
  const uint64_t* lb = b + (int64_t)q * b_stride;
  __shared__ uint32_t ida[256], idb[256];
  __shared__ uint64_t fk[256];
  if (tid < na) ida[tid] = key_id(la[tid]);
  if (tid < nb) idb[tid] = key_id(lb[tid]);
  fk[tid] = 0ull;
  __syncthreads();
  if (tid < na) {
    const uint32_t id = ida[tid];
    float s = __fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(tid + rank_base), k)));
    for (int j = 0; j < nb; ++j)
      if (idb[j] == id) {
        s = __fadd_rn(s, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k)));
        break;
      }
    fk[tid] = make_key(s, id);
  } else if (tid >= a_stride && tid - a_stride < nb) {
    const int j = tid - a_stride;
    const uint32_t id = idb[j];
    bool dup = false;
    for (int i = 0; i < na; ++i)
      if (ida[i] == id) {
        dup = true;
        break;
      }
    if (!dup) fk[tid] = make_key(__fadd_rn(0.0f, rcp_f32_rn(__fadd_rn((float)(j + rank_base), k))), id);
  }
  __syncthreads();
  if (tid >= 64) return;
  uint64_t v[4];
  int tot = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = fk[tid * 4 + e];
    tot += __popcll(__ballot(v[e] != 0ull));
  }
  w_sort<256>(v, tid);
  uint64_t* o = out + (int64_t)q * out_stride;
  const int kept = tot < limit ? tot : limit;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = tid * 4 + e;
    if (i < out_stride) o[i] = i < kept ? v[e] : 0ull;
  }
  for (int i = 256 + tid; i < out_stride; i += 64) o[i] = 0ull;
  if (tid == 0) out_cnt[q] = kept;
}
bool launch_rrf_top(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride, const int* b_cnt,
                    int B, float k, int rank_base, int limit, uint64_t* out, int out_stride, int* out_cnt, hipStream_t st) {
  if (a_stride + b_stride > 256 || a_stride < 1 || b_stride < 1 || limit > out_stride) return false;
  if (B <= 0) return true;
  hipLaunchKernelGGL(k_rrf_top, dim3(B), dim3(256), 0, st, a, a_stride, a_cnt, b, b_stride, b_cnt, k, rank_base, limit, out,
                     out_stride, out_cnt);
  HX_HIP(hipGetLastError());
  return true;
}

void launch_rrf(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride,
                const int* b_cnt, int B, float k, int rank_base, int limit, uint64_t* out, int* out_cnt,
                hipStream_t st) {
  // `out` doubles as the fused scratch: caller provides B x (a_stride + b_stride) keys.
  if (B <= 0) return;
  hipLaunchKernelGGL(k_rrf, dim3(B), dim3(256), 0, st, a, a_stride, a_cnt, b, b_stride, b_cnt, k,
                     rank_base, out);
  HX_HIP(hipGetLastError());
  const int stride = a_stride + b_stride;
  launch_compact(out, stride, nullptr, B, limit < stride ? limit : stride, 0, out, stride, out_cnt,
                 nullptr, stride, st);
}

__global__ void k_unpack(const uint64_t* keys, int64_t n, float* scores, int64_t* ids) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  scores[i] = k ? key_score(k) : -__builtin_inff();
  ids[i] = k ? (int64_t)key_id(k) : -1;
}
void launch_unpack(const uint64_t* keys, int64_t n, float* scores, int64_t* ids, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, n, scores, ids);
  HX_HIP(hipGetLastError());
}

__global__ void k_concat(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b,
                         int b_stride, const int* b_cnt, uint64_t* out) {
  const int q = blockIdx.x;
  const int na = a_cnt ? (a_cnt[q] < a_stride ? a_cnt[q] : a_stride) : a_stride;
  const int nb = b_cnt ? (b_cnt[q] < b_stride ? b_cnt[q] : b_stride) : b_stride;
  uint64_t* o = out + (int64_t)q * (a_stride + b_stride);
  for (int i = threadIdx.x; i < a_stride; i += blockDim.x) o[i] = i < na ? a[(int64_t)q * a_stride + i] : 0ull;
  for (int i = threadIdx.x; i < b_stride; i += blockDim.x)
    o[a_stride + i] = i < nb ? b[(int64_t)q * b_stride + i] : 0ull;
}
void launch_concat(const uint64_t* a, int a_stride, const int* a_cnt, const uint64_t* b, int b_stride,
                   const int* b_cnt, int B, uint64_t* out, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_concat, dim3(B), dim3(256), 0, st, a, a_stride, a_cnt, b, b_stride, b_cnt, out);
  HX_HIP(hipGetLastError());
}

// gathered [world x B x (dl + sl)] (rank-major) -> dense [B x world*dl], sparse [B x world*sl]
__global__ void k_regroup(const uint64_t* __restrict__ g, int world, int B, int dl, int sl,
                          uint64_t* __restrict__ d, uint64_t* __restrict__ s) {
  const int b = blockIdx.x, L = dl + sl;
  for (int i = threadIdx.x; i < world * L; i += blockDim.x) {
    const int r = i / L, j = i - r * L;
    const uint64_t k = g[((int64_t)r * B + b) * L + j];
    if (j < dl) d[(int64_t)b * world * dl + r * dl + j] = k;
    else s[(int64_t)b * world * sl + r * sl + (j - dl)] = k;
  }
}
void launch_regroup(const uint64_t* g, int world, int B, int dl, int sl, uint64_t* d, uint64_t* s, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_regroup, dim3(B), dim3(256), 0, st, g, world, B, dl, sl, d, s);
  HX_HIP(hipGetLastError());
}

template <typename T>
__global__ void k_fill(T* p, int64_t n, T v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// per-query state of a chunked scan in one launch: threshold -inf, counts, flags
__global__ void k_scan_init(float* tau, int* cnt, int* ovf, int* kept, int B, int cnt0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  tau[b] = -__builtin_inff();
  cnt[b] = cnt0;
  ovf[b] = 0;
  kept[b] = 0;
}
void launch_scan_init(float* tau, int* cnt, int* ovf, int* kept, int B, int cnt0, hipStream_t st) {
  hipLaunchKernelGGL(k_scan_init, dim3((B + 255) / 256), dim3(256), 0, st, tau, cnt, ovf, kept, B, cnt0);
  HX_HIP(hipGetLastError());
}

void launch_fill_f32(float* p, int64_t n, float v, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  HX_HIP(hipGetLastError());
}
// hx_h1_local_async: the row after a batch's lists -- element 0 = queries whose lists are not final, zeros after it
__global__ void k_flag_row(const int* nfail, const int* spsum, uint64_t* row, int len) {
  for (int i = threadIdx.x; i < len; i += blockDim.x)
    row[i] = i == 0 ? (uint64_t)(uint32_t)(nfail[0] + spsum[0] + spsum[1]) : 0ull;
}
void launch_flag_row(const int* nfail, const int* spsum, uint64_t* row, int len, hipStream_t st) {
  hipLaunchKernelGGL(k_flag_row, dim3(1), dim3(256), 0, st, nfail, spsum, row, len);
  HX_HIP(hipGetLastError());
}

// Insertion-order ids under row sharding (engine.hip IdBlocks): a shard's rows arrive in blocks, block k = local
// rows [row0[k], row0[k + 1]) with global ids gid0[k], gid0[k] + 1, ...; both columns ascend, so the map between the
// engine's internal id (id_base + local row) and the global id is monotone and a sorted list stays sorted.
//   to_global = 1: keys carry internal ids -> global ids, in place (out == keys allowed);
//   to_global = 0: keys carry global ids -> internal ids; ids of other shards (and empty slots) give key 0.
__device__ __forceinline__ int last_le(const uint32_t* col, int nb, uint32_t v) {   // last k with col[k] <= v, -1 if none
  int lo = 0, hi = nb;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}
__global__ void k_remap_ids(const uint64_t* in, uint64_t* out, int64_t n, const uint32_t* row0, const uint32_t* gid0,
                            int nb, uint32_t id_base, uint32_t n_rows, int to_global) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = in[i];
  uint64_t r = 0ull;
  if (k != 0ull) {
    const uint32_t id = key_id(k);
    if (to_global) {
      const uint32_t row = id - id_base;
      const int b = last_le(row0, nb, row);
      if (b >= 0 && id >= id_base && row < n_rows) r = (k & 0xFFFFFFFF00000000ull) | (uint64_t)(0xFFFFFFFFu - (gid0[b] + (row - row0[b])));
    } else {
      const int b = last_le(gid0, nb, id);
      if (b >= 0) {
        const uint32_t row = row0[b] + (id - gid0[b]);
        const uint32_t end = b + 1 < nb ? row0[b + 1] : n_rows;
        if (row < end) r = (k & 0xFFFFFFFF00000000ull) | (uint64_t)(0xFFFFFFFFu - (id_base + row));
      }
    }
  }
  out[i] = r;
}
void launch_remap_ids(const uint64_t* in, uint64_t* out, int64_t n, const uint32_t* row0, const uint32_t* gid0, int nb,
                      uint32_t id_base, uint32_t n_rows, int to_global, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_remap_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n, row0, gid0, nb,
                     id_base, n_rows, to_global);
  HX_HIP(hipGetLastError());
}

// *acc += add[0] + add[1] (a deferred stage folds its summary into the caller's flag word)
__global__ void k_flag_add(int* acc, const int* add) {
  if (threadIdx.x == 0) {
    const int v = add[0] + add[1];
    if (v) atomicAdd(acc, v);
  }
}
void launch_flag_add(int* acc, const int* add, hipStream_t st) {
  hipLaunchKernelGGL(k_flag_add, dim3(1), dim3(64), 0, st, acc, add);
  HX_HIP(hipGetLastError());
}

void launch_fill_i32(int* p, int64_t n, int v, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill<int>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
