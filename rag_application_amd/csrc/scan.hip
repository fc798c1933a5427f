// K3/K4/K5 -- whole-collection scan: S = X . Q^T on the matrix cores, fused with a
// threshold filter that appends (score, id) keys to per-query candidate buffers.
//
// Mirrors Prefetch(query=..., using="dense"|"matryoshka_*"|"quantized", limit=...)
// of the reference (app/core/vector_store/qdrant/qdrant_handler.py:311-315,
// 327-329, 335-339): "rank the whole collection by cosine".  Two element kinds
// share one byte geometry (128 B of a row per k-step):
//   KIND_F16: fp16 copies of the L2-normalised rows, v_mfma_f32_32x32x16_f16.
//             Scores are APPROXIMATE (|err| <= HX_EPS_F16); the caller re-scores
//             the survivors exactly and certifies the result (engine.cpp).
//   KIND_I8 : the reference's trunc(127*x) copy, v_mfma_i32_32x32x32_i8; the
//             integer dot is exact and the score (f32(dot)*rinv_x)*rinv_q is the
//             oracle's arithmetic bit for bit.
//
// Layout: corpus rows are the MFMA A operand (M), queries the B operand (N), so a
// lane of the 32x32 accumulator owns ONE query (col = lane & 31) and 16 corpus
// rows: the per-query threshold is one register per tile.
//
// Staging: global_load_lds (16 B/lane, 1 KiB per wave-instruction = 8 rows x
// 128 B) into an NSTAGE ring; the LDS image is lane-linear, the 16-B slot XOR
// swizzle ((row>>1)&7) is applied on the SOURCE address and again on the
// ds_read_b128 address, so every 16-lane read group hits 16 distinct bank slots.
#include <stdlib.h>
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#ifndef HX_SCAN_NT
#define HX_SCAN_NT 1
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile shapes: 256 x 256 with 8 waves (each 128 x 64: 4 x 2 MFMA tiles, 0.75 LDS fragment
// reads per MFMA, 128 KiB of LDS, one workgroup per CU) for large batches; 128 x {128,64,32}
// with 4 waves (two workgroups per CU) for small ones.
// QRES (one query tile whose rows are at most QRES_KT * 128 bytes): the query tile is loaded ONCE into its own LDS area
// and stays there; the ring carries corpus rows only -- a fifth fewer LDS-DMA pieces per k-step at 32 queries.
constexpr int QRES_KT = 6;
template <int KIND, int BM, int BN, int NSTAGE, bool QRES = false>
__global__ __launch_bounds__(BM * 2, 2) void k_scan(ScanArgs a) {
  constexpr int NW = BM / 32;                       // waves: 8 or 4
  constexpr int WN = BN >= 256 ? 4 : (BN >= 64 ? 2 : 1);
  constexpr int WM = NW / WN;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + (QRES ? 0 : B_BYTES);
  constexpr int A_LPW = BM / 8 / NW;  // 1-KiB pieces per wave
  constexpr int B_LPW = BN / 8 / NW;
  constexpr int LPW = A_LPW + (QRES ? 0 : B_LPW);
  static_assert(B_LPW >= 1, "BN >= 8 * waves");

  // Per-query constants of the epilogue (threshold, int8 query scale) and the appends of this workgroup are kept in LDS:
  // a VGPR-destination global load or a returning global atomic inside the stream is waited for with vmcnt, which
  // counts in issue order -- the wait would drain the LDS-DMA ring ahead of it (round 3: the epilogue of every tile
  // did, for the threshold; every append did, for its slot).
  constexpr int TABQ = BN <= 128 ? BN : 512;        // queries the tables hold (nq_tiles * BN above that: global loads)
  constexpr int LCAP = (BN == 64 || QRES) ? 512 : 1024;   // staged appends per workgroup (beyond: the global atomic, as before)
  __shared__ __attribute__((aligned(1024))) uint8_t lds[NSTAGE * STAGE + (QRES ? QRES_KT * B_BYTES : 0)];
  uint8_t* const qres = lds + NSTAGE * STAGE;
  __shared__ float lds_tau[TABQ], lds_rq[TABQ];
  __shared__ uint64_t lds_key[LCAP];
  __shared__ int lds_kq[LCAP];
  __shared__ int lds_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // kernel-argument pointers inside a struct are generic to hipcc: make them provably global, or the
  // epilogue's accesses become FLAT instructions whose waits also drain the LDS-DMA queue
  typedef __attribute__((address_space(1))) const float GF;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) const f32x4 GF4;
  GF* g_tau = (GF*)a.tau;
  GF* g_rinv_q = (GF*)a.rinv_q;
  GF* g_rinv_x = (GF*)a.rinv_x;
  auto* g_cnt = (__attribute__((address_space(1))) int*)a.cnt;
  auto* g_ovf = (__attribute__((address_space(1))) int*)a.overflow;
  auto* g_cand = (__attribute__((address_space(1))) uint64_t*)a.cand;

  const int KT = (int)(a.row_bytes >> 7);
  const int64_t n_rows = a.row_end - a.row_begin;
  const int n_row_tiles = (int)((n_rows + BM - 1) / BM);
  const int nq = a.nq_tiles;

  // XCD-aware tile walk: blocks with equal (blockIdx % 8) share an XCD (L2); XCD x
  // owns row tiles == x (mod 8) and walks them with the query tile fastest, so the
  // nq blocks that need one corpus tile run together on one L2.
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7;
  const int per_xcd = G >> 3;
  const int my_rt = (n_row_tiles - xcd + 7) >> 3;         // row tiles owned by this XCD
  const int items_x = my_rt * nq;  // < 2^31: launch_scan bounds rows per launch
  const int i0 = blockIdx.x >> 3;
  if (i0 >= items_x) return;
  const int my_items = (items_x - i0 + per_xcd - 1) / per_xcd;
  const int64_t total_steps = (int64_t)my_items * KT;
  const bool tabs = nq * BN <= TABQ;
  if (tabs)
    for (int q = tid; q < nq * BN; q += BM * 2) {
      lds_tau[q] = q < a.B ? g_tau[q] : __builtin_inff();
      lds_rq[q] = (KIND == KIND_I8 && q < a.B) ? g_rinv_q[q] : 0.f;
    }
  if (tid == 0) lds_n = 0;
  __syncthreads();

  // ---- load cursor -----------------------------------------------------------
  // Per tile, each lane keeps the byte offset of its 16-byte slot in every 1-KiB piece it
  // loads (rows past the end of the scan range are clamped to the last row); per k-step
  // the source is tile base + lane offset + 128*kt: two 64-bit adds per load.
  int lj = i0;
  int l_kt = 0;
  int64_t l_step = 0;
  const int rin = lane >> 3, pslot = lane & 7;
  const uint8_t* a_tile = nullptr;
  const uint8_t* q_tile = nullptr;
  int64_t a_off[A_LPW], b_off[B_LPW];
  // first PHYSICAL row of the rt-th tile of this launch (kernels.hpp: scan order)
  auto phys_row0 = [&](int rt) {
    const int64_t lrow = a.row_begin + (int64_t)rt * BM;            // logical
    const uint32_t pt = scan_phys_tile((uint32_t)(lrow >> 8), a.perm_mul, a.perm_n, a.perm_inv);
    return (int64_t)pt * 256 + (lrow & 255);
  };
  auto set_tile = [&]() {
    const int rt = (lj / nq) * 8 + xcd, qt = lj % nq;
    const int64_t row0 = phys_row0(rt);
    const int64_t left = a.n_total - row0;             // valid rows of the tile (<= 0: a tile past the end)
    a_tile = a.A + row0 * a.row_bytes;
    q_tile = a.Q + (int64_t)qt * BN * a.row_bytes;
#pragma unroll
    for (int c = 0; c < A_LPW; ++c) {
      const int trow = (wave + NW * c) * 8 + rin;
      const int64_t erow = trow < left ? trow : (left >= 1 ? left - 1 : -row0);   // clamp to a row that exists
      a_off[c] = erow * a.row_bytes + ((pslot ^ ((trow >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int c = 0; c < B_LPW; ++c) {
      const int trow = (wave + NW * c) * 8 + rin;
      b_off[c] = (int64_t)trow * a.row_bytes + ((pslot ^ ((trow >> 1) & 7)) << 4);
    }
  };
  set_tile();

  auto issue_load = [&]() {
    const int st = (int)(l_step % NSTAGE);
    uint8_t* sbase = lds + st * STAGE;
    const int64_t koff = (int64_t)l_kt << 7;
#pragma unroll
    for (int c = 0; c < A_LPW; ++c) {
      // one query tile: every corpus row is read by exactly one workgroup, once -- non-temporal (aux = 2) keeps the
      // stream from displacing the query tile in L2; with several query tiles the row tile is shared through L2
      if (HX_SCAN_NT && nq == 1)
        __builtin_amdgcn_global_load_lds(GLB_PTR(a_tile + a_off[c] + koff), LDS_PTR(sbase + (wave + NW * c) * 1024),
                                         16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds(GLB_PTR(a_tile + a_off[c] + koff), LDS_PTR(sbase + (wave + NW * c) * 1024),
                                         16, 0, 0);
    }
    if constexpr (!QRES) {
#pragma unroll
      for (int c = 0; c < B_LPW; ++c)
        __builtin_amdgcn_global_load_lds(GLB_PTR(q_tile + b_off[c] + koff),
                                         LDS_PTR(sbase + A_BYTES + (wave + NW * c) * 1024), 16, 0, 0);
    }
    ++l_step;
    if (++l_kt == KT) {
      l_kt = 0;
      lj += per_xcd;
      if (l_step < total_steps) set_tile();
    }
  };

  // ---- accumulators -----------------------------------------------------------
  using acc_t = typename std::conditional<KIND == KIND_F16, f32x16, i32x16>::type;
  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  if constexpr (QRES) {     // the whole query tile, once (launch_scan: nq_tiles == 1, KT <= QRES_KT)
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int c = 0; c < B_LPW; ++c)
        __builtin_amdgcn_global_load_lds(GLB_PTR(q_tile + b_off[c] + ((int64_t)kt << 7)),
                                         LDS_PTR(qres + kt * B_BYTES + (wave + NW * c) * 1024), 16, 0, 0);
    }
    wait_vmcnt<0>();        // this wave's pieces; the other waves' are behind the first barrier of the loop
  }
  // prologue: NSTAGE-1 steps in flight
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (l_step < total_steps) issue_load();

  int cj = i0;
  int c_kt = 0;
  for (int64_t s = 0; s < total_steps; ++s) {
    // retire the loads of step s (all but the youngest NSTAGE-2 groups)
    if (s + NSTAGE - 2 < total_steps) {
      wait_vmcnt<LPW*(NSTAGE - 2)>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // loads(s) of every wave landed; compute(s-1) done everywhere
    if (l_step < total_steps) issue_load();  // into the stage compute(s-1) just released

    const uint8_t* As = lds + (int)(s % NSTAGE) * STAGE;
    const uint8_t* Bs = QRES ? qres + c_kt * B_BYTES : As + A_BYTES;
    // all fragments of the k-step first (16-byte LDS reads, conflict free), then the MFMAs:
    // hipcc interleaves them behind counted lgkmcnt waits
    half8 af[4][TM], bf[4][TN];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + r;
        af[kk][i] = *(const half8*)(As + row * 128 + ((((kk << 1) | h) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + r;
        bf[kk][j] = *(const half8*)(Bs + row * 128 + ((((kk << 1) | h) ^ ((row >> 1) & 7)) << 4));
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (KIND == KIND_F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, af[kk][i]),
                                                              __builtin_bit_cast(i32x4, bf[kk][j]), acc[i][j], 0, 0, 0);
          }
        }

    if (++c_kt == KT) {
      // ---- epilogue: threshold filter + append ----------------------------------
      c_kt = 0;
      const int rt = (int)(cj / nq) * 8 + xcd, qt = (int)(cj % nq);
      cj += per_xcd;
      float rxm = 0.f;
      if constexpr (KIND == KIND_I8) {
        typedef __attribute__((address_space(4))) const float CF;   // uniform index: a scalar load
        rxm = ((CF*)a.rinv_tile_max)[phys_row0(rt) >> 8];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int q = qt * BN + wn * (TN * 32) + j * 32 + r;
        const bool qok = q < a.B;
        float tau, rq = 0.f;
        if (tabs) {
          tau = lds_tau[q];
          rq = lds_rq[q];
        } else {
          tau = qok ? g_tau[q] : __builtin_inff();
          if constexpr (KIND == KIND_I8) rq = qok ? g_rinv_q[q] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int64_t lrow0 = a.row_begin + (int64_t)rt * BM + wm * (TM * 32) + i * 32 + 4 * h;   // logical
          const int64_t row0 = phys_row0(rt) + wm * (TM * 32) + i * 32 + 4 * h;                    // physical
          if constexpr (KIND == KIND_I8) {
            // (f32(dot) * largest row scale of the 256-row tile) * query scale bounds the row's score from above
            // (non-negative factors, monotone rounding; scan8.hip).  No per-row scale -- a VGPR-destination load -- is
            // fetched in the stream: a row whose bound reaches the threshold is staged as (dot, row) and scored with its
            // own scale when the workgroup has finished its tiles.
            if (!a.all_pass) {
              int im = acc[i][j][0];
#pragma unroll
              for (int e = 1; e < 16; ++e) im = acc[i][j][e] > im ? acc[i][j][e] : im;
              const float bound = im > 0 ? ((float)im * rxm) * rq : 0.f;
              if (__builtin_amdgcn_ballot_w64(bound >= tau) != 0ull) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                  const int d = acc[i][j][e];
                  const float be = d > 0 ? ((float)d * rxm) * rq : 0.f;
                  if (be >= tau && row < a.n_total) {
                    const int lp = __hip_atomic_fetch_add(&lds_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (lp < LCAP) {
                      lds_key[lp] = ((uint64_t)(uint32_t)d << 32) | (uint32_t)row;
                      lds_kq[lp] = q;
                    } else {             // staging area full: score and append at once
                      const float sc1 = ((float)d * g_rinv_x[row]) * rq;
                      if (sc1 >= tau) {
                        const int pos = __hip_atomic_fetch_add(g_cnt + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (pos < a.cap)
                          g_cand[(int64_t)q * a.cap + pos] = make_key(sc1, (uint32_t)(a.id_base + row));
                        else
                          g_ovf[q] = 1;
                      }
                    }
                  }
                }
              }
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
              continue;
            }
          }
          float sc[16];
          if constexpr (KIND == KIND_F16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[e] = acc[i][j][e];
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              // rows row0+8g .. +3 ; rinv_x is padded past n_rows so this never faults
              const f32x4 rx = *(const GF4*)(g_rinv_x + (row0 + 8 * g < a.n_total ? row0 + 8 * g : 0));
              sc[4 * g + 0] = ((float)acc[i][j][4 * g + 0] * rx.x) * rq;
              sc[4 * g + 1] = ((float)acc[i][j][4 * g + 1] * rx.y) * rq;
              sc[4 * g + 2] = ((float)acc[i][j][4 * g + 2] * rx.z) * rq;
              sc[4 * g + 3] = ((float)acc[i][j][4 * g + 3] * rx.w) * rq;
            }
          }
          bool hit = false;
#pragma unroll
          for (int e = 0; e < 16; ++e) hit |= (sc[e] >= tau);
          if (a.all_pass || __builtin_amdgcn_ballot_w64(hit) != 0ull) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
              if (a.all_pass) {   // first chunk: slot = logical row, nothing to count
                const int64_t slot = lrow0 + (e & 3) + 8 * (e >> 2) - a.row_begin;
                if (qok && slot < a.row_end - a.row_begin)
                  g_cand[(int64_t)q * a.cap + slot] =
                      (row < a.n_total && sc[e] >= tau) ? make_key(sc[e], (uint32_t)(a.id_base + row)) : 0ull;
              } else if (sc[e] >= tau && row < a.n_total) {
                const uint64_t key = make_key(sc[e], (uint32_t)(a.id_base + row));
                const int lp = __hip_atomic_fetch_add(&lds_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lp < LCAP) {            // staged: placed when the workgroup has finished its tiles
                  lds_key[lp] = key;
                  lds_kq[lp] = q;
                } else {
                  const int pos = __hip_atomic_fetch_add(g_cnt + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if (pos < a.cap)
                    g_cand[(int64_t)q * a.cap + pos] = key;
                  else
                    g_ovf[q] = 1;
                }
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        }
      }
    }
  }
  // the staged appends of this workgroup
  __syncthreads();
  const int staged = lds_n < LCAP ? lds_n : LCAP;
  for (int t = tid; t < staged; t += BM * 2) {
    const int q = lds_kq[t];
    if constexpr (KIND == KIND_I8) {      // (dot, row) -> the oracle's score, with the row's own scale
      const int d = (int)(uint32_t)(lds_key[t] >> 32);
      const uint32_t row = (uint32_t)lds_key[t];
      const float sc1 = ((float)d * g_rinv_x[row]) * g_rinv_q[q];
      if (!(sc1 >= g_tau[q])) continue;
      lds_key[t] = make_key(sc1, (uint32_t)(a.id_base + row));
    }
    const int pos = __hip_atomic_fetch_add(g_cnt + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pos < a.cap)
      g_cand[(int64_t)q * a.cap + pos] = lds_key[t];
    else
      g_ovf[q] = 1;
  }
}

template <int KIND, int BM, int BN, int NSTAGE, bool QRES = false>
static void launch(const ScanArgs& a, int64_t tiles, hipStream_t st) {
  constexpr int per_cu = BM == 256 ? 1 : 2;
  int64_t cap = 256 * per_cu;
  if (a.oversub > 1 && BM != 256) cap = cap * a.oversub - 32;   // (kernels.hpp: ScanArgs.oversub)
  int64_t g = tiles < cap ? tiles : cap;
  g = (g + 7) / 8 * 8;  // whole XCD groups; blocks without items exit at once
  hipLaunchKernelGGL((k_scan<KIND, BM, BN, NSTAGE, QRES>), dim3((unsigned)g), dim3(BM * 2), 0, st, a);
}

void launch_scan(const ScanArgs& a, int kind, int bn, hipStream_t st, hipEvent_t after_kernel) {
  const int64_t n_rows = a.row_end - a.row_begin;
  if (n_rows <= 0 || a.B <= 0) return;
  HX_CHECK((a.row_bytes & 127) == 0, "scan: row_bytes must be a multiple of 128");
  if (scan8_usable(a, bn)) return launch_scan8(a, kind, st, after_kernel);
  const int bm = bn == 256 ? 256 : 128;
  const int64_t tiles = (n_rows + bm - 1) / bm * a.nq_tiles;
  static const bool qres_on = !getenv("HX_DEBUG_NO_QRES");      // diagnostics: the streamed query tile for every shape
  const bool qres = qres_on && bn == 32 && a.nq_tiles == 1 && a.row_bytes <= QRES_KT * 128;
  // (Round 4, measured and dropped: the resident query tile for 64 / 128 queries too -- 48 / 96 KiB of LDS beside a
  // three-stage ring of corpus rows leaves ONE 4-wave workgroup per CU: 2.57-3.37 ms per 10M-row pass against 1.51-1.91
  // with two workgroups streaming both operands; a 256-row tile for 64 / 128 queries (8 waves, one workgroup per CU,
  // three stages): 1.82-2.22 ms against 1.56-1.86 -- profiles/r04_mid_batch.txt)
  if (kind == KIND_F16) {
    if (bn == 256) launch<KIND_F16, 256, 256, 2>(a, tiles, st);
    else if (bn == 128) launch<KIND_F16, 128, 128, 2>(a, tiles, st);
    else if (bn == 64) launch<KIND_F16, 128, 64, 3>(a, tiles, st);
    else if (qres) launch<KIND_F16, 128, 32, 3, true>(a, tiles, st);
    else launch<KIND_F16, 128, 32, 3>(a, tiles, st);
  } else {
    if (bn == 256) launch<KIND_I8, 256, 256, 2>(a, tiles, st);
    else if (bn == 128) launch<KIND_I8, 128, 128, 2>(a, tiles, st);
    else if (bn == 64) launch<KIND_I8, 128, 64, 3>(a, tiles, st);
    else if (qres) launch<KIND_I8, 128, 32, 3, true>(a, tiles, st);
    else launch<KIND_I8, 128, 32, 3>(a, tiles, st);
  }
  HX_HIP(hipGetLastError());
  if (after_kernel) HX_HIP(hipEventRecord(after_kernel, st));
}

}  // namespace hx
