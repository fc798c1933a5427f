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

__device__ __forceinline__ float wave_bcast_sum_sq(const float* v, int nchunks, int lane) {
  float p = 0.0f;
  for (int j = 0; j < nchunks; ++j) p = __fadd_rn(p, __fmul_rn(v[j], v[j]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) p = __fadd_rn(p, __shfl_down(p, off, 64));
  p = __fadd_rn(p, 0.0f);
  return __shfl(p, 0, 64);
}

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

__global__ __launch_bounds__(256) void k_prep_rows(PrepRowsArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  const int nch = a.dim_pad >> 6;
  const float* src = a.raw + row * a.dim;
  float v[MAXCH];
#pragma unroll 4
  for (int j = 0; j < nch; ++j) {
    const int c = (j << 6) + lane;
    v[j] = c < a.dim ? src[c] : 0.0f;
  }
  // dense
  {
    const float len2 = wave_bcast_sum_sq(v, nch, lane);
    const bool keep = keep_unnormalised(len2);
    const float ln = sqrt_f32_rn(len2);
    float* d = a.dense + row * a.dim_pad;
    _Float16* dh = a.dense_h + row * a.dim_pad;
    for (int j = 0; j < nch; ++j) {
      const float o = keep ? v[j] : div_f32_rn(v[j], ln);
      d[(j << 6) + lane] = o;
      dh[(j << 6) + lane] = (_Float16)o;
    }
  }
  // prefixes of the RAW row
  for (int p = 0; p < a.n_prefix; ++p) {
    const int pch = a.psize[p] >> 6;
    const float len2 = wave_bcast_sum_sq(v, pch, lane);
    const bool keep = keep_unnormalised(len2);
    const float ln = sqrt_f32_rn(len2);
    float* d = a.pre[p] + row * a.psize[p];
    for (int j = 0; j < pch; ++j) {
      const float o = keep ? v[j] : div_f32_rn(v[j], ln);
      d[(j << 6) + lane] = o;
      if (p == 0 && a.pre_h0) a.pre_h0[row * a.psize[0] + (j << 6) + lane] = (_Float16)o;
    }
  }
  // int8 copy of the RAW row + 1/||.||
  {
    int8_t* d = a.q8 + row * a.dim_pad8;
    int n2 = 0;
    for (int j = 0; j < nch; ++j) {
      const int8_t t = quant_i8(v[j]);
      d[(j << 6) + lane] = t;
      n2 += (int)t * (int)t;
    }
    for (int c = a.dim_pad + lane; c < a.dim_pad8; c += 64) d[c] = 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_down(n2, off, 64);
    if (lane == 0) a.q8_rinv[row] = n2 > 0 ? (float)(1.0 / sqrt((double)n2)) : 0.0f;
  }
}

void launch_prep_rows(const PrepRowsArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  HX_CHECK(a.dim_pad <= MAXCH * 64, "dim > 4096 unsupported");
  hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, st, a);
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
  float v[MAXCH];
#pragma unroll 4
  for (int j = 0; j < nch; ++j) {
    const int c = (j << 6) + lane;
    v[j] = c < d ? q_raw[(int64_t)b * q_dim + c] : 0.0f;
  }
  const float len2 = wave_bcast_sum_sq(v, nch, lane);
  const bool keep = keep_unnormalised(len2);
  const float ln = sqrt_f32_rn(len2);
  for (int j = 0; j < nch; ++j) {
    const float o = keep ? v[j] : div_f32_rn(v[j], ln);
    qn[(int64_t)b * dpad + (j << 6) + lane] = o;
    if (qh) qh[(int64_t)b * dpad + (j << 6) + lane] = (_Float16)o;
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
