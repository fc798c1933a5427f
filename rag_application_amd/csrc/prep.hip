// K1/K2 -- derive the stored named vectors from raw rows, and prepare query batches.
//
// Mirrors what the reference hands to Qdrant per chunk
// (app/core/vector_store/qdrant/qdrant_handler.py:144-163): "dense" (L2-normalised by
// the COSINE collection), "matryoshka_{64,128,256}" = normalised prefixes of the RAW
// embedding (:148-150), "quantized" = clip((x*127).astype(int8)) of the RAW embedding
// (:144-146).  One wave per row; arithmetic = oracle.cosine_preprocess / quantize_i8.
#include <algorithm>
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

// numpy (x86) float64 -> int8 cast of x*127: truncate, wrap through int32; 0 when
// |t| >= 2^31 or t is NaN (tests/golden/i8_kat.json pins this).
__device__ __forceinline__ int8_t quant_i8(float x) {
  const double t = (double)x * 127.0;
  const int32_t v = (fabs(t) < 2147483648.0) ? (int32_t)t : 0;
  return (int8_t)(v & 0xFF);
}

// IEEE fp32 sqrt / divide through fp64 (53 >= 2*24+2 bits: the double rounding is
// innocuous), independent of how hipcc lowers the fp32 forms.
__device__ __forceinline__ float sqrt_f32_rn(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float div_f32_rn(float a, float b) { return (float)((double)a / (double)b); }

__device__ __forceinline__ bool keep_unnormalised(float len2) {
  return len2 < 1.1920929e-07f || fabsf(__fsub_rn(len2, 1.0f)) <= 1.0e-6f;
}

constexpr int MAXCH = 64;  // dim <= 4096
// the derived copies of a row are written once and read by later kernels only: streaming stores (HX_PREP_NT=0: plain)
#ifndef HX_PREP_NT
#define HX_PREP_NT 1
#endif
#if HX_PREP_NT
#define HX_NT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define HX_NT_STORE(ptr, val) (*(ptr) = (val))
#endif

// Candidate-pass copy (DESIGN.md "int8 candidate pass"): the NORMALISED row scaled to its own range,
// x8 = rint(x * 127 / max|x|) with sx = max|x| / 127 stored beside it, and the quantisation error
// ||x - sx * x8||_2 evaluated in fp64 -- the certificate of the dense stage is built from the largest one.
__device__ __forceinline__ int quant_s8(float x, float inv_sx) {
  float t = __fmul_rn(x, inv_sx);
  t = t == t ? t : 0.0f;                                 // NaN -> 0 (such a row scores 0 in the candidate pass)
  t = t > 127.0f ? 127.0f : (t < -127.0f ? -127.0f : t);
  return (int)__builtin_rintf(t);
}
__device__ __forceinline__ float f32_round_up(double x) {   // smallest fp32 >= x (x >= 0, finite)
  float f = (float)x;
  if ((double)f < x) f = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) + 1u);   // next fp32 up (f >= 0, finite)
  return f;
}

// One wave per row, four rows per workgroup, no per-thread arrays (the previous form indexed float v[64] by a
// run-time chunk count and ran out of scratch memory).  Pass 1 reads the raw row in the arithmetic contract's
// layout -- lane l owns elements 64 j + l and accumulates their squares in ascending j, the prefix sums are
// snapshots of the same accumulator -- and parks it in the wave's slice of LDS.  Pass 2 re-reads it four
// consecutive elements per lane, so every derived copy leaves as 16- (fp32), 8- (fp16) and 4-byte (int8) stores.
__global__ __launch_bounds__(256) void k_prep_rows(PrepRowsArgs a) {
  extern __shared__ float lds_rows[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= a.n) return;                                // (no workgroup barrier below: the LDS slice is the wave's own)
  float* lv = lds_rows + (size_t)w * a.dim_pad8;
  const int nch = a.dim_pad >> 6;
  const float* src = a.raw + row * a.dim;
  // ---- pass 1
  float p = 0.0f, vmax = 0.0f, psum[3] = {0.0f, 0.0f, 0.0f};
  {
    // The row's loads are issued PB at a time and consumed afterwards (one load, one `s_waitcnt vmcnt(0)`, one add per
    // trip made a row twelve memory latencies long -- 21 us per row and wave at 0.53 of HBM peak).  Fully unrolled: the
    // batch is registers, never indexed by a run-time value.
    constexpr int PB = 16;
    int pi = 0;
    for (int j0 = 0; j0 < nch; j0 += PB) {
      float t[PB];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int c = ((j0 + u) << 6) + lane;
        t[u] = (j0 + u < nch && c < a.dim) ? src[c] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int j = j0 + u;
        if (j < nch) {                                               // wave-uniform
          const float v = t[u];
          lv[(j << 6) + lane] = v;
          p = __fadd_rn(p, __fmul_rn(v, v));
          vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v));
          if (pi < a.n_prefix && ((j + 1) << 6) == a.psize[pi]) {    // wave-uniform
            psum[pi] = p;
            ++pi;
          }
        }
      }
    }
    for (int c = a.dim_pad + lane; c < a.dim_pad8; c += 64) lv[c] = 0.0f;
  }
  float len2[4] = {p, psum[0], psum[1], psum[2]};        // [0] the full row, [1 + k] prefix k
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = len2[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t = __fadd_rn(t, __shfl_down(t, off, 64));
    t = __fadd_rn(t, 0.0f);
    len2[k] = __shfl(t, 0, 64);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, off, 64));
  bool keep[4];
  float ln[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    keep[k] = keep_unnormalised(len2[k]);
    ln[k] = sqrt_f32_rn(len2[k]);
  }
  // scale of the candidate copy: division by ln > 0 is monotone, so max |x| of the stored row is |vmax| through
  // the same rounding; a row with a non-finite element gets scale 0 (it scores 0 in the candidate pass)
  float xmax = keep[0] ? vmax : div_f32_rn(vmax, ln[0]);
  if (!(xmax <= 3.0e38f)) xmax = 0.0f;
  const float sx = __fdiv_rn(xmax, 127.0f);
  const float inv_sx = xmax > 0.0f ? __fdiv_rn(127.0f, xmax) : 0.0f;
  // ---- pass 2: four consecutive elements per lane
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  float* d32 = a.dense + row * a.dim_pad;
  _Float16* d16 = a.dense_h + row * a.dim_pad;
  int8_t* d8 = a.q8 + row * a.dim_pad8;
  int8_t* ds8 = a.q8s ? a.q8s + row * a.dim_pad8 : nullptr;
  int n2 = 0;
  double err2 = 0.0;
  for (int g = lane; g < (a.dim_pad8 >> 2); g += 64) {
    const int c = g << 2;
    const f4 v = *(const f4*)(lv + c);
    if (c < a.dim_pad) {
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = keep[0] ? v[e] : div_f32_rn(v[e], ln[0]);
      HX_NT_STORE((f4*)(d32 + c), o);
      HX_NT_STORE((h4*)(d16 + c), (h4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]}));
      if (ds8) {
        uint32_t pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = quant_s8(o[e], inv_sx);
          pk |= (uint32_t)(t & 0xFF) << (8 * e);
          const double de = (double)o[e] - (double)sx * (double)t;
          err2 += de * de;
        }
        HX_NT_STORE((uint32_t*)(ds8 + c), pk);
      }
      for (int k = 0; k < a.n_prefix; ++k) {
        if (c >= a.psize[k]) continue;
        f4 po;
#pragma unroll
        for (int e = 0; e < 4; ++e) po[e] = keep[1 + k] ? v[e] : div_f32_rn(v[e], ln[1 + k]);
        HX_NT_STORE((f4*)(a.pre[k] + row * a.psize[k] + c), po);
        if (k == 0 && a.pre_h0)
          HX_NT_STORE((h4*)(a.pre_h0 + row * a.psize[0] + c), (h4{(_Float16)po[0], (_Float16)po[1], (_Float16)po[2], (_Float16)po[3]}));
      }
    } else if (ds8) {
      *(uint32_t*)(ds8 + c) = 0u;
    }
    uint32_t pk = 0;                                     // the reference's int8 copy of the RAW row
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int8_t t = quant_i8(v[e]);
      pk |= (uint32_t)(uint8_t)t << (8 * e);
      n2 += (int)t * (int)t;
    }
    HX_NT_STORE((uint32_t*)(d8 + c), pk);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    n2 += __shfl_down(n2, off, 64);
    err2 += __shfl_down(err2, off, 64);
  }
  if (lane == 0) {
    a.q8_rinv[row] = n2 > 0 ? (float)(1.0 / sqrt((double)n2)) : 0.0f;
    if (ds8) {
      a.q8s_scale[row] = sx;
      // (65536 atomics on ONE address per launch serialise at the memory side: only a row that raises the maximum
      // -- a handful per collection -- issues one)
      const double e = sqrt(err2);
      if (e <= 3.0e38) {
        const uint32_t eb = __builtin_bit_cast(uint32_t, f32_round_up(e));   // >= 0: bits order as values
        if (eb > __builtin_nontemporal_load(a.err_max)) atomicMax(a.err_max, eb);
      }
    }
  }
}

void launch_prep_rows(const PrepRowsArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  HX_CHECK(a.dim_pad <= MAXCH * 64, "dim > 4096 unsupported");
  const size_t lds = (size_t)4 * a.dim_pad8 * sizeof(float);
  hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)((a.n + 3) / 4)), dim3(256), lds, st, a);
  HX_HIP(hipGetLastError());
}

// The candidate copy of rows that are already stored (hx_load: the copy is derived data and is not in the file).
__global__ __launch_bounds__(256) void k_requant_rows(const float* dense, int dim_pad, int dim_pad8, int64_t n,
                                                      int8_t* q8s, float* scale, uint32_t* err_max) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const float* x = dense + row * dim_pad;
  float xmax = 0.0f;
  for (int g = lane; g < (dim_pad >> 2); g += 64) {
    const f4 v = *(const f4*)(x + (g << 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) xmax = __builtin_fmaxf(xmax, __builtin_fabsf(v[e]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) xmax = __builtin_fmaxf(xmax, __shfl_xor(xmax, off, 64));
  if (!(xmax <= 3.0e38f)) xmax = 0.0f;
  const float sx = __fdiv_rn(xmax, 127.0f);
  const float inv_sx = xmax > 0.0f ? __fdiv_rn(127.0f, xmax) : 0.0f;
  double err2 = 0.0;
  for (int g = lane; g < (dim_pad8 >> 2); g += 64) {
    const int c = g << 2;
    uint32_t pk = 0;
    if (c < dim_pad) {
      const f4 v = *(const f4*)(x + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = quant_s8(v[e], inv_sx);
        pk |= (uint32_t)(t & 0xFF) << (8 * e);
        const double de = (double)v[e] - (double)sx * (double)t;
        err2 += de * de;
      }
    }
    *(uint32_t*)(q8s + row * dim_pad8 + c) = pk;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) err2 += __shfl_down(err2, off, 64);
  if (lane == 0) {
    scale[row] = sx;
    const double e = sqrt(err2);
    if (e <= 3.0e38) {
      const uint32_t eb = __builtin_bit_cast(uint32_t, f32_round_up(e));
      if (eb > __builtin_nontemporal_load(err_max)) atomicMax(err_max, eb);
    }
  }
}
void launch_requant_rows(const float* dense, int dim_pad, int dim_pad8, int64_t n, int8_t* q8s, float* scale,
                         uint32_t* err_max, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_requant_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, dense, dim_pad, dim_pad8, n, q8s,
                     scale, err_max);
  HX_HIP(hipGetLastError());
}

__global__ void k_synth_dense(float* raw, int64_t row0_global, int64_t n, int dim, uint32_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const int64_t r = i / dim;
  const int c = (int)(i - r * dim);
  raw[i] = synth_value(seed, (uint32_t)(row0_global + r), (uint32_t)c);
}
void launch_synth_dense(float* raw, int64_t row0_global, int64_t n, int dim, uint32_t seed, hipStream_t st) {
  const int64_t tot = n * dim;
  if (tot <= 0) return;
  hipLaunchKernelGGL(k_synth_dense, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, raw,
                     row0_global, n, dim, seed);
  HX_HIP(hipGetLastError());
}

// queries: normalised first d elements of each raw query -> fp32 [B x dpad], fp16 [Bpad x dpad]
__global__ __launch_bounds__(256) void k_prep_queries_f(const float* q_raw, int q_dim, int B, int Bpad,
                                                        int d, int dpad, float* qn, _Float16* qh) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= Bpad) return;
  const int nch = dpad >> 6;
  if (b >= B) {
    if (qh)
      for (int j = 0; j < nch; ++j) qh[(int64_t)b * dpad + (j << 6) + lane] = (_Float16)0.0f;
    return;
  }
  // Two passes over the raw query (768 floats: L1 after the first) instead of a per-thread array indexed by a
  // run-time chunk count, which hipcc kept in scratch (272 B per lane; the same pattern k_prep_rows lost in round 3).
  // Lane l accumulates the squares of elements 64 j + l for ascending j: spec_dot's order.
  const float* src = q_raw + (int64_t)b * q_dim;
  float p = 0.0f;
  for (int j = 0; j < nch; ++j) {
    const int c = (j << 6) + lane;
    const float x = c < d ? src[c] : 0.0f;
    p = __fadd_rn(p, __fmul_rn(x, x));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_down(p, off, 64));
  p = __fadd_rn(p, 0.0f);
  const float len2 = __shfl(p, 0, 64);
  const bool keep = keep_unnormalised(len2);
  const float ln = sqrt_f32_rn(len2);
  for (int j = 0; j < nch; ++j) {
    const int c = (j << 6) + lane;
    const float x = c < d ? src[c] : 0.0f;
    const float o = keep ? x : div_f32_rn(x, ln);
    qn[(int64_t)b * dpad + c] = o;
    if (qh) qh[(int64_t)b * dpad + c] = (_Float16)o;
  }
}
void launch_prep_queries_f(const float* q_raw, int q_dim, int B, int Bpad, int d, int dpad, float* qn,
                           _Float16* qh, hipStream_t st) {
  if (Bpad <= 0) return;
  HX_CHECK(dpad <= MAXCH * 64, "dim > 4096 unsupported");
  hipLaunchKernelGGL(k_prep_queries_f, dim3((Bpad + 3) / 4), dim3(256), 0, st, q_raw, q_dim, B, Bpad, d,
                     dpad, qn, qh);
  HX_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_prep_queries_i8(const float* q_raw, int q_dim, int B, int Bpad,
                                                         int dpad8, int8_t* q8, float* rinv_q) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= Bpad) return;
  int n2 = 0;
  for (int c = lane; c < dpad8; c += 64) {
    const int8_t t = (b < B && c < q_dim) ? quant_i8(q_raw[(int64_t)b * q_dim + c]) : (int8_t)0;
    q8[(int64_t)b * dpad8 + c] = t;
    n2 += (int)t * (int)t;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_down(n2, off, 64);
  if (lane == 0) rinv_q[b] = n2 > 0 ? (float)(1.0 / sqrt((double)n2)) : 0.0f;
}
void launch_prep_queries_i8(const float* q_raw, int q_dim, int B, int Bpad, int dpad8, int8_t* q8,
                            float* rinv_q, hipStream_t st) {
  if (Bpad <= 0) return;
  hipLaunchKernelGGL(k_prep_queries_i8, dim3((Bpad + 3) / 4), dim3(256), 0, st, q_raw, q_dim, B, Bpad,
                     dpad8, q8, rinv_q);
  HX_HIP(hipGetLastError());
}

// Candidate-pass form of a prepared (normalised) query batch: q8 = rint(q * 127 / max|q|), its scale sq, and the
// query's certificate radius
//     eps_q = (XMAX + E_X) * E_q + E_X * ||q|| + SLOP
// -- |spec_dot(x, q) - sx sq <x8, q8>| <= eps_q for every stored row x (||x|| <= XMAX, quantisation error <= E_X;
// E_q = ||q - sq q8||, all norms in fp64 and rounded up).  SLOP covers what is rounded in fp32: spec_dot itself
// (dpad / 64 chunk additions + 6 tree additions + the products, each 2^-24 of a sum of |x_i q_i| <= XMAX * ||q||),
// the scan's score (int -> float, two multiplications) and the comparison m + eps < e_L: (dpad / 64 + 16) * 2^-24
// * XMAX * ||q||.  DESIGN.md "int8 candidate pass".
// XMAX: a stored row is v / fl(sqrt(fl(|v|^2))) or, by the keep-if-unit rule, a row with |fl(|v|^2) - 1| <= 1e-6;
// fl(|v|^2) is spec_dot(v, v), off by at most (64 + 7) * 2^-24 = 4.3e-6 relative at dim 4096: ||x|| <= 1 + 4e-6.
constexpr double S8_XMAX = 1.00001;
__global__ __launch_bounds__(256) void k_prep_queries_s8(const float* qn, int dpad, int B, int Bpad, int dpad8,
                                                         int8_t* q8, float* sq_out, float* eps_out,
                                                         const uint32_t* err_max) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= Bpad) return;
  if (b >= B) {
    for (int c = lane * 4; c < dpad8; c += 256) *(uint32_t*)(q8 + (int64_t)b * dpad8 + c) = 0u;
    if (lane == 0) sq_out[b] = 0.0f;
    return;
  }
  const float* q = qn + (int64_t)b * dpad;
  float qmax = 0.0f;
  for (int c = lane; c < dpad; c += 64) qmax = __builtin_fmaxf(qmax, __builtin_fabsf(q[c]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) qmax = __builtin_fmaxf(qmax, __shfl_xor(qmax, off, 64));
  const bool finite = qmax <= 3.0e38f;
  if (!finite) qmax = 0.0f;
  const float sq = __fdiv_rn(qmax, 127.0f);
  const float inv = qmax > 0.0f ? __fdiv_rn(127.0f, qmax) : 0.0f;
  double err2 = 0.0, nrm2 = 0.0;
  for (int c = lane * 4; c < dpad8; c += 256) {
    uint32_t pk = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = c + e < dpad ? q[c + e] : 0.0f;
      const int t = quant_s8(v, inv);
      pk |= (uint32_t)(t & 0xFF) << (8 * e);
      const double de = (double)v - (double)sq * (double)t;
      err2 += de * de;
      nrm2 += (double)v * (double)v;
    }
    *(uint32_t*)(q8 + (int64_t)b * dpad8 + c) = pk;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    err2 += __shfl_down(err2, off, 64);
    nrm2 += __shfl_down(nrm2, off, 64);
  }
  if (lane == 0) {
    sq_out[b] = sq;
    const double EX = (double)__builtin_bit_cast(float, *err_max);
    const double qn2 = sqrt(nrm2);
    const double slop = (double)((dpad >> 6) + 16) * 5.9604644775390625e-08 * S8_XMAX * qn2;
    const double eps = (S8_XMAX + EX) * sqrt(err2) + EX * qn2 + slop;
    // a query the bound does not cover (non-finite) gets an infinite radius: its certificate fails, it is re-run
    eps_out[b] = (finite && eps < 1.0e30) ? f32_round_up(eps * (1.0 + 1e-9)) : __builtin_inff();
  }
}
void launch_prep_queries_s8(const float* qn, int dpad, int B, int Bpad, int dpad8, int8_t* q8, float* sq, float* eps,
                            const uint32_t* err_max, hipStream_t st) {
  if (Bpad <= 0) return;
  hipLaunchKernelGGL(k_prep_queries_s8, dim3((Bpad + 3) / 4), dim3(256), 0, st, qn, dpad, B, Bpad, dpad8, q8, sq, eps,
                     err_max);
  HX_HIP(hipGetLastError());
}

// out[t] = max of the non-negative floats p[256 t .. 256 t + 255] (clipped to n): one wave per tile
__global__ __launch_bounds__(256) void k_tile_max(const float* p, int64_t n, float* out) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t * 256 >= n) return;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = t * 256 + k * 64 + lane;
    const float v = i < n ? p[i] : 0.f;
    m = v > m ? v : m;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
  }
  if (lane == 0) out[t] = m;
}
void launch_tile_max(const float* p, int64_t n, float* out, hipStream_t st) {
  if (n <= 0) return;
  const int64_t tiles = (n + 255) / 256;
  hipLaunchKernelGGL(k_tile_max, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, p, n, out);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
