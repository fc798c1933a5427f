// Row-sharded H1 with the exchange BEFORE the exact scores ("candidates first").
//
// The reference has no multi-device path; the query these kernels serve is the H1 configuration
// of app/core/vector_store/qdrant/qdrant_handler.py:327-360 (dense Prefetch (+) sparse Prefetch -> Fusion.RRF).
//
// What every shard repeated per QUERY, whatever its row count, was the exact re-score of L' = 450 dense candidates, the
// compaction of 4096-key buffers and the exact re-score of ~110 sparse candidates.  Here a shard only NOMINATES:
//   nominate   its best k1 rows by the int8 candidate score s8 and its best k2 documents by the integer BM25 score
//              (k1 ~ L'/world + 10 sigma, k2 likewise), plus two words per query (certificate radius, flags, list length);
//   all-gather of the nominations;
//   rescore    every rank merges them into the GLOBAL candidate lists -- top-L' by s8, the documents within the margin of
//              the global L-th integer score -- checks that no shard's list was cut above the global cut (else the
//              batch is flagged and redone through the per-shard path), and computes the exact scores of ITS OWN rows
//              among them: ~L'/world + ~(L + 10)/world per query;
//   all-reduce (integer sum) of the exact keys: a slot is written by exactly one rank, the others hold 0 (the sum of
//              one key and zeros is the key; max would need a sign-safe key form, keys use all 64 bits);
//   finish     top-L of both lists, the certificate m + eps < e_L evaluated once on the global list, RRF.
// These are the small kernels between those steps; the heavy ones are the engine's own (scan8.hip, sparse2.hip,
// select.hip, sprescore.hip).
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

// ---- nominate: pack the shard's lists and its two meta words per query -------------------------------------------------
// nom = [B x k1] dense s8 keys | [B x k2] sparse integer keys | [B x 2] meta:
//   meta0 = eps_q bits | dflag << 32   (dflag bit 0: the dense list is not trustworthy -- a buffer or a log overflowed;
//                                       bit 1: the list holds every row the shard has)
//   meta1 = length of the sparse list before the cut to k2 (31 bits) | sflag << 31 (the integer pass flagged or failed the
//           query in this shard) | bits of the document-weight bound the shard's integer scores are scaled by << 32
//           (integer scores of two shards compare only under ONE scale: hx_set_sparse_wmax)
__global__ __launch_bounds__(256) void k_h1x_pack(const uint64_t* cand, int cstride, const int* cnt, const int* ovf,
                                                  const float* eps, int complete, int k1, const uint64_t* list,
                                                  int lstride, const int* lcnt, const int* sflag, const int* sfail, int k2,
                                                  float wmax, int B, uint64_t* nom) {
  const int b = blockIdx.x;
  uint64_t* d = nom + (int64_t)b * k1;
  uint64_t* s = nom + (int64_t)B * k1 + (int64_t)b * k2;
  uint64_t* m = nom + (int64_t)B * (k1 + k2) + 2 * b;
  int nd = cand ? cnt[b] : 0;
  nd = nd < k1 ? nd : k1;
  for (int j = threadIdx.x; j < k1; j += 256) d[j] = j < nd ? cand[(int64_t)b * cstride + j] : 0ull;
  int ns = list ? lcnt[b] : 0;
  const int nsk = ns < k2 ? ns : k2;
  for (int j = threadIdx.x; j < k2; j += 256) s[j] = j < nsk ? list[(int64_t)b * lstride + j] : 0ull;
  if (threadIdx.x == 0) {
    const uint32_t df = ((cand && ovf[b]) ? 1u : 0u) | (complete ? 2u : 0u);
    m[0] = (uint64_t)(cand ? __builtin_bit_cast(uint32_t, eps[b]) : 0u) | ((uint64_t)df << 32);
    const uint32_t sf = (list && (sflag[b] != 0 || sfail[b] != 0)) ? 1u : 0u;
    m[1] = (uint64_t)((uint32_t)ns & 0x7FFFFFFFu) | ((uint64_t)sf << 31) | ((uint64_t)__builtin_bit_cast(uint32_t, wmax) << 32);
  }
}
void launch_h1x_pack(const uint64_t* cand, int cstride, const int* cnt, const int* ovf, const float* eps, int complete,
                     int k1, const uint64_t* list, int lstride, const int* lcnt, const int* sflag, const int* sfail, int k2,
                     float wmax, int B, uint64_t* nom, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_pack, dim3(B), dim3(256), 0, st, cand, cstride, cnt, ovf, eps, complete, k1, list, lstride,
                     lcnt, sflag, sfail, k2, wmax, B, nom);
  HX_HIP(hipGetLastError());
}

// ---- rescore, step 1: the shards' nominations side by side per query ----------------------------------------------------
// g = [world][B * (k1 + k2 + 2)] (what the all-gather leaves) -> du [B x world * k1], su [B x world * k2]
__global__ __launch_bounds__(256) void k_h1x_union(const uint64_t* g, int world, int B, int k1, int k2, uint64_t* du,
                                                   uint64_t* su) {
  const int b = blockIdx.x;
  const int64_t W = (int64_t)B * (k1 + k2 + 2);
  for (int i = threadIdx.x; i < world * k1; i += 256) {
    const int r = i / k1, j = i - r * k1;
    du[(int64_t)b * world * k1 + i] = g[r * W + (int64_t)b * k1 + j];
  }
  for (int i = threadIdx.x; i < world * k2; i += 256) {
    const int r = i / k2, j = i - r * k2;
    su[(int64_t)b * world * k2 + i] = g[r * W + (int64_t)B * k1 + (int64_t)b * k2 + j];
  }
}
void launch_h1x_union(const uint64_t* g, int world, int B, int k1, int k2, uint64_t* du, uint64_t* su, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_union, dim3(B), dim3(256), 0, st, g, world, B, k1, k2, du, su);
  HX_HIP(hipGetLastError());
}

__device__ __forceinline__ uint32_t h1x_thr(uint32_t aL, int M) {   // sparse2.hip sp_thr
  const int t = (int)aL - M + 1;
  return t < 1 ? 1u : (uint32_t)t;
}

// ---- rescore, step 2: the global cuts and the completeness of every shard's list ------------------------------------------
// G [B x lp] = top-lp of the union by s8 (sorted, gc[b] of them), SL [B x ks] = top-ks of the union by integer score.
// One thread per query writes
//   meta[5 b + 0] = orderable(m): the lp-th best s8, the bound of every row outside G; 0 when G holds every nominated row
//                   (fewer than lp in all: nothing is outside)
//   meta[5 b + 1] = the largest certificate radius eps_q any shard computed (bits of a non-negative float)
//   meta[5 b + 2] = flags != 0: the batch must be redone per shard.  A shard's dense list is cut at k1: the rows it did
//                   not send score at most its k1-th key, so the list is complete enough iff that key does not beat the
//                   global cut (or the shard sent every row it has).  Likewise a sparse list cut at k2 must reach
//                   below the global threshold a_L - M + 1.
//   meta[5 b + 3] = gc[b]: every key of G must come back from exactly one rank (hx_h1_finish counts them)
//   meta[5 b + 4] = (filled after the sparse re-score) the sparse candidates that must come back
// and q_margin[b] (M = T + T/16 + 4, sprescore.hip: k_sparse_prep) / q_flag[b] = 0 for k_sparse_rescore.
__global__ __launch_bounds__(256) void k_h1x_cuts(const uint64_t* g, int world, int B, int k1, int k2, const uint64_t* G,
                                                  const int* gc, int lp, const uint64_t* SL, const int* sc, int ks, int L_s,
                                                  const int64_t* q_indptr, uint64_t* meta, int* q_margin, int* q_flag) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int64_t W = (int64_t)B * (k1 + k2 + 2);
  const int T = (int)(q_indptr[b + 1] - q_indptr[b]);
  const int M = T + T / 16 + 4;
  q_margin[b] = M;
  q_flag[b] = 0;
  const int ng = gc[b];
  const uint64_t gcut = ng >= lp ? G[(int64_t)b * lp + lp - 1] : 0ull;       // 0: no cut, every nominated row is in G
  const int ns = sc[b] < ks ? sc[b] : ks;
  const uint32_t thr = ns >= L_s ? h1x_thr((uint32_t)(SL[(int64_t)b * ks + L_s - 1] >> 32), M) : 1u;
  uint32_t flags = 0, epsb = 0;
  const uint32_t wbits0 = (uint32_t)(g[(int64_t)B * (k1 + k2) + 2 * b + 1] >> 32);
  for (int r = 0; r < world; ++r) {
    const uint64_t* mr = g + r * W + (int64_t)B * (k1 + k2) + 2 * b;
    const uint64_t m0 = mr[0], m1 = mr[1];
    const uint32_t df = (uint32_t)(m0 >> 32), sf = (uint32_t)(m1 >> 31) & 1u;
    const int slen = (int)((uint32_t)m1 & 0x7FFFFFFFu);
    if ((uint32_t)(m1 >> 32) != wbits0) flags |= 32u;                        // the shards scaled their integer scores differently
    const uint32_t e = (uint32_t)m0;
    epsb = e > epsb ? e : epsb;                                               // (non-negative floats: bit order = value order)
    if (df & 1u) flags |= 1u;
    if (!(df & 2u)) {                                                         // the shard has more rows than it sent
      const uint64_t last = g[r * W + (int64_t)b * k1 + k1 - 1];
      if (last != 0ull && (gcut == 0ull || last > gcut)) flags |= 2u;          // its cut lies above the global one
    }
    if (sf & 1u) flags |= 4u;
    if (slen > k2) {                                                          // sparse list cut at k2
      const uint64_t last = g[r * W + (int64_t)B * k1 + (int64_t)b * k2 + k2 - 1];
      if ((uint32_t)(last >> 32) >= thr) flags |= 8u;
    }
  }
  meta[5 * b + 0] = gcut ? (gcut >> 32) : 0ull;
  meta[5 * b + 1] = (uint64_t)epsb;
  meta[5 * b + 2] = (uint64_t)flags;
  meta[5 * b + 3] = (uint64_t)(uint32_t)ng;
  meta[5 * b + 4] = 0ull;
}
void launch_h1x_cuts(const uint64_t* g, int world, int B, int k1, int k2, const uint64_t* G, const int* gc, int lp,
                     const uint64_t* SL, const int* sc, int ks, int L_s, const int64_t* q_indptr, uint64_t* meta,
                     int* q_margin, int* q_flag, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_cuts, dim3((B + 255) / 256), dim3(256), 0, st, g, world, B, k1, k2, G, gc, lp, SL, sc, ks, L_s,
                     q_indptr, meta, q_margin, q_flag);
  HX_HIP(hipGetLastError());
}

// after k_sparse_rescore over the global list: its prefix length (the candidates within the margin) and its failure flag
// (the margin set did not fit ks keys) join the meta words
__global__ void k_h1x_fold(const int* sp_pref, const int* sp_fail, int B, uint64_t* meta) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  meta[5 * b + 4] = (uint64_t)(uint32_t)sp_pref[b];
  if (sp_fail[b]) meta[5 * b + 2] |= 16ull;
}
void launch_h1x_fold(const int* sp_pref, const int* sp_fail, int B, uint64_t* meta, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_fold, dim3((B + 255) / 256), dim3(256), 0, st, sp_pref, sp_fail, B, meta);
  HX_HIP(hipGetLastError());
}

// ---- finish: the certificate, once, on the global list -------------------------------------------------------------------
// red = the all-reduced (sum) [B x lp] dense exact keys | [B x ks] sparse exact keys | [B x 5] meta (the same on every rank,
// so it comes back multiplied by `world`); D [B x L] = the exact top-L (sorted), Dc its counts.  One wave per query
// counts the keys that came back and applies
//     fail = flags || keys missing || (a cut exists && !(m + eps < e_L))            (select.hip: k_certify)
// nfail += failed queries (the word the pipeline reads two submits later).
__global__ __launch_bounds__(256) void k_h1x_certify(const uint64_t* red, int world, int B, int lp, int ks, const uint64_t* D,
                                                     const int* Dc, int L, int* fail, int* nfail) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const uint64_t* de = red + (int64_t)b * lp;
  const uint64_t* se = red + (int64_t)B * lp + (int64_t)b * ks;
  const uint64_t* m = red + (int64_t)B * (lp + ks) + 5 * b;
  int nd = 0, ns = 0;
  for (int j = lane; j < lp; j += 64) nd += de[j] != 0ull ? 1 : 0;
  for (int j = lane; j < ks; j += 64) ns += se[j] != 0ull ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    nd += __shfl_xor(nd, off, 64);
    ns += __shfl_xor(ns, off, 64);
  }
  if (lane != 0) return;
  const uint64_t wd = (uint64_t)world;
  bool bad = m[2] != 0ull || (m[0] % wd) != 0ull || (m[1] % wd) != 0ull || (uint64_t)nd * wd != m[3] || (uint64_t)ns * wd != m[4];
  if (!bad && m[0] != 0ull) {
    const float cut = orderable_f32((uint32_t)(m[0] / wd));
    const float eps = __builtin_bit_cast(float, (uint32_t)(m[1] / wd));
    if (Dc[b] < L) bad = true;
    else bad = !(__fadd_rn(cut, eps) < key_score(D[(int64_t)b * L + L - 1]));
  }
  fail[b] = bad ? 1 : 0;
  if (bad) atomicAdd(nfail, 1);
}
void launch_h1x_certify(const uint64_t* red, int world, int B, int lp, int ks, const uint64_t* D, const int* Dc, int L,
                        int* fail, int* nfail, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_certify, dim3((B + 3) / 4), dim3(256), 0, st, red, world, B, lp, ks, D, Dc, L, fail, nfail);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
