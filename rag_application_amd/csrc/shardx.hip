// Row-sharded H1 with the exchange BEFORE the exact scores ("candidates first").
//
// The reference has no multi-device path; the query these kernels serve is the H1 configuration
// of app/core/vector_store/qdrant/qdrant_handler.py:327-360 (dense Prefetch (+) sparse Prefetch -> Fusion.RRF).
//
// What every shard repeated per QUERY, whatever its row count, was the exact re-score of L' = 450 dense candidates, the
// compaction of 4096-key buffers and the exact re-score of ~110 sparse candidates.  Here a shard only NOMINATES:
//   nominate   its best k1 rows by the int8 candidate score s8 and the k2 best integer BM25 scores of its documents
//              (k1 ~ L'/world + 10 sigma, k2 likewise), plus two words per query (certificate radius, flags, list length,
//              the scale of its integer scores); its full integer-score list stays with the batch, on this rank;
//   all-gather of the nominations;
//   rescore    every rank derives the GLOBAL cuts -- the top-L' by s8; the L-th best integer score, hence the threshold
//              a_L - M + 1 of the margin set -- checks that no shard's list was cut above them (else the batch is flagged
//              and redone through the per-shard path), and computes the exact scores of ITS OWN rows among the global
//              dense candidates (~L'/world per query, at their positions in the global list) and of its own documents
//              at or above the global threshold (~(L + 10)/world; its best k3 of them go into its own slot);
//   all-reduce (integer sum) of that buffer: every key slot has one owner, the others hold 0 (the sum of one key and
//              zeros is the key; max would need a sign-safe key form, keys use all 64 bits);
//   finish     top-L of both lists, the certificate m + eps < e_L evaluated once on the global list, RRF.
// These are the small kernels between those steps; the heavy ones are the engine's own (scan8.hip, sparse2.hip,
// select.hip, sprescore.hip).
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

// ---- nominate: pack the shard's lists and its two meta words per query -------------------------------------------------
// nom = [B x k1] dense s8 keys | [B x k2] sparse integer keys | [B x 2] meta   (gathered)
//       | [B x lout] the shard's whole integer-score list | [B] its length     (private: stays on this rank, for rescore)
//   meta0 = eps_q bits | dflag << 32   (dflag bit 0: the dense list is not trustworthy -- a buffer or a log overflowed;
//                                       bit 1: the list holds every row the shard has)
//   meta1 = length of the sparse list (31 bits) | sflag << 31 (the integer pass flagged or failed the query in this
//           shard) | bits of the document-weight bound the shard's integer scores are scaled by << 32
//           (integer scores of two shards compare only under ONE scale: hx_set_sparse_wmax)
__global__ __launch_bounds__(256) void k_h1x_pack(const uint64_t* cand, int cstride, const int* cnt, const int* ovf,
                                                  const float* eps, int complete, int k1, const uint64_t* list,
                                                  int lstride, const int* lcnt, const int* sflag, const int* sfail, int k2,
                                                  int lout, float wmax, int B, uint64_t* nom) {
  const int b = blockIdx.x;
  uint64_t* d = nom + (int64_t)b * k1;
  uint64_t* s = nom + (int64_t)B * k1 + (int64_t)b * k2;
  uint64_t* m = nom + (int64_t)B * (k1 + k2) + 2 * b;
  uint64_t* pl = nom + (int64_t)B * (k1 + k2 + 2) + (int64_t)b * lout;
  uint64_t* pc = nom + (int64_t)B * (k1 + k2 + 2) + (int64_t)B * lout + b;
  int nd = cand ? cnt[b] : 0;
  nd = nd < k1 ? nd : k1;
  for (int j = threadIdx.x; j < k1; j += 256) d[j] = j < nd ? cand[(int64_t)b * cstride + j] : 0ull;
  int ns = list ? lcnt[b] : 0;
  ns = ns < lstride ? ns : lstride;
  const int nsk = ns < k2 ? ns : k2;
  for (int j = threadIdx.x; j < k2; j += 256) s[j] = j < nsk ? list[(int64_t)b * lstride + j] : 0ull;
  const int npl = ns < lout ? ns : lout;
  for (int j = threadIdx.x; j < npl; j += 256) pl[j] = list[(int64_t)b * lstride + j];
  if (threadIdx.x == 0) {
    const uint32_t df = ((cand && ovf[b]) ? 1u : 0u) | (complete ? 2u : 0u);
    m[0] = (uint64_t)(cand ? __builtin_bit_cast(uint32_t, eps[b]) : 0u) | ((uint64_t)df << 32);
    const uint32_t sf = (list && (sflag[b] != 0 || sfail[b] != 0 || ns > lout)) ? 1u : 0u;
    m[1] = (uint64_t)((uint32_t)ns & 0x7FFFFFFFu) | ((uint64_t)sf << 31) | ((uint64_t)__builtin_bit_cast(uint32_t, wmax) << 32);
    *pc = (uint64_t)(uint32_t)npl;
  }
}
void launch_h1x_pack(const uint64_t* cand, int cstride, const int* cnt, const int* ovf, const float* eps, int complete,
                     int k1, const uint64_t* list, int lstride, const int* lcnt, const int* sflag, const int* sfail, int k2,
                     int lout, float wmax, int B, uint64_t* nom, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_pack, dim3(B), dim3(256), 0, st, cand, cstride, cnt, ovf, eps, complete, k1, list, lstride,
                     lcnt, sflag, sfail, k2, lout, wmax, B, nom);
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
// G [B x lp] = top-lp of the union by s8 (sorted, gc[b] of them); ST [B x L_s] = the L_s best integer-score keys of the
// union (sc[b] of them).  One thread per query writes
//   meta[4 b + 0] = orderable(m): the lp-th best s8, the bound of every row outside G; 0 when G holds every nominated row
//                   (fewer than lp in all: nothing is outside)
//   meta[4 b + 1] = the largest certificate radius eps_q any shard computed (bits of a non-negative float)
//   meta[4 b + 2] = flags != 0: the batch must be redone per shard.  A shard's dense list is cut at k1: the rows it did
//                   not send score at most its k1-th key, so the list is complete enough iff that key does not beat the
//                   global cut (or the shard sent every row it has).  The sparse nominations only have to fix the VALUE
//                   of the global L-th integer score a_L: a list cut at k2 is enough iff its last score does not exceed
//                   a_L (what it did not send scores no higher, and equal scores do not move the L-th value).
//   meta[4 b + 3] = gc[b]: every key of G must come back from exactly one rank (hx_h1_finish counts them)
// and thr[b] = a_L - M + 1 (M = T + T/16 + 4, sprescore.hip: k_sparse_prep): the GLOBAL threshold of the margin set every
// shard applies to its own list; 1 when the collection holds fewer than L_s candidates.
__global__ __launch_bounds__(256) void k_h1x_cuts(const uint64_t* g, int world, int B, int k1, int k2, const uint64_t* G,
                                                  const int* gc, int lp, const uint64_t* ST, const int* sc, int L_s,
                                                  const int64_t* q_indptr, uint64_t* meta, uint32_t* thr_out,
                                                  int* q_margin, int* q_flag) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int64_t W = (int64_t)B * (k1 + k2 + 2);
  const int T = (int)(q_indptr[b + 1] - q_indptr[b]);
  const int M = T + T / 16 + 4;
  q_margin[b] = M;
  q_flag[b] = 0;
  const int ng = gc[b];
  const uint64_t gcut = ng >= lp ? G[(int64_t)b * lp + lp - 1] : 0ull;       // 0: no cut, every nominated row is in G
  const bool have_L = sc[b] >= L_s;
  const uint32_t aL = have_L ? (uint32_t)(ST[(int64_t)b * L_s + L_s - 1] >> 32) : 0u;
  thr_out[b] = have_L ? h1x_thr(aL, M) : 1u;
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
    if (slen > k2) {                                                          // it holds more scores than it sent
      const uint64_t last = g[r * W + (int64_t)B * k1 + (int64_t)b * k2 + k2 - 1];
      if (!have_L || (uint32_t)(last >> 32) > aL) flags |= 8u;
    }
  }
  meta[4 * b + 0] = gcut ? (gcut >> 32) : 0ull;
  meta[4 * b + 1] = (uint64_t)epsb;
  meta[4 * b + 2] = (uint64_t)flags;
  meta[4 * b + 3] = (uint64_t)(uint32_t)ng;
}
void launch_h1x_cuts(const uint64_t* g, int world, int B, int k1, int k2, const uint64_t* G, const int* gc, int lp,
                     const uint64_t* ST, const int* sc, int L_s, const int64_t* q_indptr, uint64_t* meta, uint32_t* thr_out,
                     int* q_margin, int* q_flag, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_cuts, dim3((B + 255) / 256), dim3(256), 0, st, g, world, B, k1, k2, G, gc, lp, ST, sc, L_s,
                     q_indptr, meta, thr_out, q_margin, q_flag);
  HX_HIP(hipGetLastError());
}

// the private list of the nominate step as k_sparse_rescore wants it: counts as int32
__global__ void k_h1x_counts(const uint64_t* priv_cnt, int B, int* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = (int)(uint32_t)priv_cnt[b];
}
void launch_h1x_counts(const uint64_t* priv_cnt, int B, int* out, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_counts, dim3((B + 255) / 256), dim3(256), 0, st, priv_cnt, B, out);
  HX_HIP(hipGetLastError());
}

// this rank's best k3 exact sparse keys T [B x k3] (tc[b] of them) into ITS slot of the summed buffer, the number of its
// documents at or above the threshold beside them; a list the select pass cut short joins the flags
__global__ __launch_bounds__(64) void k_h1x_place(const uint64_t* T, const int* tc, const int* ncand, const int* sp_fail,
                                                  int B, int k3, int world, int rank, uint64_t* se, uint64_t* nc,
                                                  uint64_t* meta) {
  const int b = blockIdx.x;
  const int n = tc[b] < k3 ? tc[b] : k3;
  uint64_t* o = se + ((int64_t)b * world + rank) * k3;
  for (int j = threadIdx.x; j < k3; j += 64) o[j] = j < n ? T[(int64_t)b * k3 + j] : 0ull;
  if (threadIdx.x == 0) {
    nc[(int64_t)b * world + rank] = (uint64_t)(uint32_t)ncand[b];
    if (sp_fail[b]) meta[4 * b + 2] |= 16ull;
  }
}
void launch_h1x_place(const uint64_t* T, const int* tc, const int* ncand, const int* sp_fail, int B, int k3, int world,
                      int rank, uint64_t* se, uint64_t* nc, uint64_t* meta, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_place, dim3(B), dim3(64), 0, st, T, tc, ncand, sp_fail, B, k3, world, rank, se, nc, meta);
  HX_HIP(hipGetLastError());
}

// ---- finish: the certificate, once, on the global list -------------------------------------------------------------------
// red = the all-reduced (sum) [B x lp] dense exact keys | [B x world x k3] sparse exact keys | [B x world] candidates per
// rank | [B x 4] meta (the same on every rank, so it comes back multiplied by `world`); D [B x L] / S [B x L_s] = the
// exact top lists (sorted), Dc / Sc their counts.  One wave per query counts the dense keys that came back and applies
//     fail = flags || keys missing || (a cut exists && !(m + eps < e_L))            (select.hip: k_certify)
//            || a rank held more sparse candidates than its k3 slots and its last slot beats the global L_s-th key
// nfail += failed queries (the word the pipeline reads two submits later).
__global__ __launch_bounds__(256) void k_h1x_certify(const uint64_t* red, int world, int B, int lp, int k3, const uint64_t* D,
                                                     const int* Dc, int L, const uint64_t* S, const int* Sc, int L_s,
                                                     int* fail, int* nfail) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const uint64_t* de = red + (int64_t)b * lp;
  const uint64_t* se = red + (int64_t)B * lp + (int64_t)b * world * k3;
  const uint64_t* nc = red + (int64_t)B * (lp + world * k3) + (int64_t)b * world;
  const uint64_t* m = red + (int64_t)B * (lp + world * k3 + world) + 4 * b;
  int nd = 0;
  for (int j = lane; j < lp; j += 64) nd += de[j] != 0ull ? 1 : 0;
  bool cutshort = false;
  const uint64_t sL = Sc[b] >= L_s ? S[(int64_t)b * L_s + L_s - 1] : 0ull;
  for (int r = lane; r < world; r += 64)
    if ((int64_t)nc[r] > (int64_t)k3 && (sL == 0ull || se[(int64_t)r * k3 + k3 - 1] > sL)) cutshort = true;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nd += __shfl_xor(nd, off, 64);
  const bool anycut = __ballot(cutshort) != 0ull;
  if (lane != 0) return;
  const uint64_t wd = (uint64_t)world;
  bool bad = anycut || m[2] != 0ull || (m[0] % wd) != 0ull || (m[1] % wd) != 0ull || (uint64_t)nd * wd != m[3];
  if (!bad && m[0] != 0ull) {
    const float cut = orderable_f32((uint32_t)(m[0] / wd));
    const float eps = __builtin_bit_cast(float, (uint32_t)(m[1] / wd));
    if (Dc[b] < L) bad = true;
    else bad = !(__fadd_rn(cut, eps) < key_score(D[(int64_t)b * L + L - 1]));
  }
  fail[b] = bad ? 1 : 0;
  if (bad) atomicAdd(nfail, 1);
}
void launch_h1x_certify(const uint64_t* red, int world, int B, int lp, int k3, const uint64_t* D, const int* Dc, int L,
                        const uint64_t* S, const int* Sc, int L_s, int* fail, int* nfail, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_h1x_certify, dim3((B + 3) / 4), dim3(256), 0, st, red, world, B, lp, k3, D, Dc, L, S, Sc, L_s, fail,
                     nfail);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
