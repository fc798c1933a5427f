// K3/K5 for large batches -- the whole-collection scan S = X . Q^T as a 256 x 256 tile
// per 512-thread workgroup, software-pipelined in half-tile phases with the two wave
// groups of the workgroup staggered by half a phase.
//
// Same contract as scan.hip (reference: Prefetch(query=..., using="dense"|"quantized",
// limit=...) of app/core/vector_store/qdrant/qdrant_handler.py:327-339, "rank the whole
// collection by cosine"); this file only changes HOW the tile is computed.
//
// Geometry.  8 waves as 2 (rows) x 4 (queries); a wave owns 128 rows x 64 queries as four
// quadrants C[ha][hb] of 64 x 32 (two 32x32 MFMA tiles each).  A k-tile is 128 bytes of
// every row (64 halves / 128 int8).  The operands of one k-tile travel as four HALF-TILES
// of 128 rows x 128 B = 16 KiB:
//     A0, A1 : corpus rows  {wm*128 + h*64 + i}   (the h-th 64 rows of both wave rows)
//     B0, B1 : query rows   {wn*64  + h*32 + j}   (the h-th 32 queries of all four wave columns)
// Each half-tile is two 1-KiB global_load_lds pieces per wave (16 B per lane, lane-linear
// LDS image, 16-byte-slot XOR swizzle applied to the SOURCE address and again on the
// ds_read_b128 address -- conflict free, see scan.hip).
//
// Schedule.  Stream order of half-tiles: g = 4*T + {A0, B0, B1, A1}; slot = g mod 8
// (8 x 16 KiB ring).  A k-tile is two phases of two quadrants each (HX_S8_PH = 2; the
// one-quadrant-per-phase form, HX_S8_PH = 4, has twice the barriers and measured 5-7 % slower):
//     L segment : ds_read the fragments this phase needs, the four loads of half-tiles g+6, g+7 (HX_S8_LDMA = 1,
//                 round 2), one counted s_waitcnt vmcnt
//     barrier
//     M segment : the 32 MFMAs of two quadrants, back to back (round 1 issued the four loads between them: a
//                 global_load_lds costs the MFMA stream 40-60 cycles of issue, the wave in its L segment has them
//                 to spare -- +2.6 ... +3.4 % on the 10M-row scan, interleaved A/B on one box)
//     barrier
//   X: reads A0, B0, B1(T), stages B1, A1 of T+1 | vmcnt(8) | C00 += A0.B0, C01 += A0.B1
//   Y: reads A1(T),         stages A0, B0 of T+2 | vmcnt(6) | C11 += A1.B1, C10 += A1.B0
// RAW: the wait of X retires g <= 4T+3 (4T+4 .. 4T+7 may be in flight), the wait of Y retires
// g <= 4T+6 (4T+7 .. 4T+9 in flight), each ahead of the barrier before the reading phase.
// WAR: slot(g) is restaged by g+8 in the L segment one phase after its last read by THIS wave group; the
// other group reads one barrier later, so the restage is still one barrier behind every read of the slot
// (which every wave retires with lgkmcnt(0) first).  Intervals I0 = L(X)T of group 0, group 1 one behind:
// A1(T) is read at I2 / I3 and restaged at I4 / I5; A0, B0(T) read at I0 / I1, restaged at I2 / I3; B1(T)
// read at I0 / I1, restaged at I4 / I5.
// Waves 4..7 (wm = 1) run one barrier behind waves 0..3, so on every SIMD one wave is in
// its M segment while its partner is in its L segment: the matrix pipe never waits for
// LDS reads or load issue.  Per-query thresholds live in LDS (no VGPR-destination global
// loads inside the pipeline: those would make hipcc drain the LDS-DMA queue).
//
// Appends.  A returning atomic on the per-query counter stalls the wave -- and through the
// barriers its whole workgroup -- for a memory round trip per passing score (measured: +34 %
// kernel time at 1,200 passes per query).  Instead every wave owns a log in global memory and
// writes the passing lane's column of scores with plain stores at a scalar position it keeps
// itself; k_scatter_log picks the passing rows and moves them into the per-query candidate
// buffers after the scan.
#include <stdlib.h>
#include <type_traits>
#include "hx_common.hpp"
#include "kernels.hpp"

namespace hx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

#ifndef HX_S8_PH
#define HX_S8_PH 2   // phases per k-tile: 4 (one quadrant each) or 2 (two quadrants each)
#endif
#ifndef HX_S8_PRIO
#define HX_S8_PRIO 1   // 1: raise the wave's priority for its M segment; 0: never; 2: waves 4..7 at priority 1 throughout
#endif
constexpr int S8_HT = 16384;       // half-tile bytes
constexpr int S8_MAXQ = 4096;      // queries whose thresholds fit the LDS table

// TS = MFMA tile side: 32 (v_mfma_*_32x32x16_f16 / 32x32x32_i8) or 16 (16x16x32_f16 / 16x16x64_i8).
// Same LDS image, same number of 16-byte fragment reads and of matrix-pipe cycles per phase.
// HQ ("half the queries", round 4): a 256-row x 128-query tile for batches of 65..128 queries.  Same skeleton, slots and
// barriers; a wave owns 128 rows x 32 queries (the hb = 0 quadrants only), so the B1 half-tile is never staged or read and the
// quadrants C01 / C11 are never computed: half the matrix work and three quarters of the LDS-DMA pieces per byte of corpus --
// at 128 queries the 256-wide form spent half its MFMAs on padding columns.  Query q sits at row (q / 32) * 32 + q % 32 of a
// 128-row query tile (wave column wn = (q / 32) % 4).  The only schedule change: phase X stages one half-tile (A1 of T + 1),
// so its counted wait leaves 6 pieces in flight instead of 8 (g = 4T+4, 4T+5, 4T+7).
template <int KIND, int TS, int DBG, bool HQ = false>
__global__ __launch_bounds__(512, 2) void k_scan8(ScanArgs a) {
  constexpr int MT = 64 / TS;          // row tiles of a quadrant
  constexpr int NT = 32 / TS;          // query tiles of a quadrant
  constexpr int KS = TS == 32 ? 4 : 2; // MFMA k-steps of a k-tile
  constexpr int EPT = TS * TS / 64;    // accumulator registers of one tile
  constexpr int AUX = S8_MAXQ * 4;
  // timing builds (wrong results; -DHX_SCAN_DBG, scripts/scan8_ablate.sh): 1 one corpus tile over and over, 2 no loads in the
  // loop, 3 no MFMAs, 4 no filter, 5 no fragment reads in the loop, 6 neither loads nor reads nor filter (MFMAs + barriers),
  // 7 as 6 without the barriers, 8 no loads and no filter, 9 / 10 below
  constexpr bool NO_READS = DBG == 5 || DBG == 6 || DBG == 7;
  constexpr bool NO_GLDS = DBG == 2 || DBG == 6 || DBG == 7 || DBG == 8;
  constexpr bool NO_FILTER = DBG == 4 || DBG == 6 || DBG == 7 || DBG == 8 || DBG == 9 || DBG == 10 || DBG == 11;
  constexpr bool NO_VMWAIT = DBG == 9;     // 9: no filter and no counted waits (the loads are issued, nothing waits for them)
  constexpr bool ONE_TILE = DBG == 1 || DBG == 10;   // 10: 1 without the filter
  constexpr bool L2_TILES = DBG == 11;               // 11: no filter, every XCD alternates between TWO row tiles: rows from L2, not from L1
  constexpr bool NO_BAR = DBG == 7;
  __shared__ __attribute__((aligned(1024))) uint8_t lds[8 * S8_HT + AUX];
  float* lds_tau = (float*)(lds + 8 * S8_HT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & (TS - 1), hh = lane / TS;   // row/column inside a tile, k-group

  typedef __attribute__((address_space(1))) const float GF;
  typedef __attribute__((address_space(1))) const f32x4 GF4;
  GF* g_tau = (GF*)a.tau;
  GF* g_rinv_q = (GF*)a.rinv_q;
  GF* g_rinv_x = (GF*)a.rinv_x;
  auto* g_ovf = (__attribute__((address_space(1))) int*)a.overflow;

  const int KT = (int)(a.row_bytes >> 7);
  const int64_t n_rows = a.row_end - a.row_begin;
  const int n_row_tiles = (int)((n_rows + 255) >> 8);
  const int nq = a.nq_tiles;

  // XCD-aware tile walk (as scan.hip): XCD x owns row tiles == x (mod 8), query tile fastest
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7;
  const int per_xcd = G >> 3;
  const int my_rt = (n_row_tiles - xcd + 7) >> 3;
  const int items_x = my_rt * nq;
  const int i0 = blockIdx.x >> 3;
  if (i0 >= items_x) {
    if (lane == 0) a.hitcnt[blockIdx.x * 8 + wave] = 0;
    return;
  }
  const int my_items = (items_x - i0 + per_xcd - 1) / per_xcd;
  const int total_T = my_items * KT;   // k-tiles of this block (launch_scan8 bounds it below 2^31)

  // thresholds (and int8 query scales) of every query -> LDS, +inf for the padding
  constexpr int QT = HQ ? 128 : 256;     // queries of a query tile
  constexpr int QW = HQ ? 32 : 64;       // ... of a wave column
  const int Bpad = nq * QT;
  for (int q = tid; q < Bpad; q += 512) {
    float t = q < a.B ? g_tau[q] : __builtin_inff();
    if constexpr (KIND == KIND_I8) {
      // threshold of f32(dot) * max row scale: tau / rinv_q pushed down by 2^-20 (the filter may let more columns
      // through than pass, never fewer); a non-positive threshold passes every column, a zero query scale with a
      // positive threshold none (all its scores are +0)
      const float rq = q < a.B ? g_rinv_q[q] : 0.f;
      if (t <= 0.f) t = -__builtin_inff();
      else if (!(rq > 0.f)) t = __builtin_inff();
      else t = (t / rq) * 0.99999905f;
    }
    lds_tau[q] = t;
  }
  __syncthreads();

  // ---- per-lane constants ---------------------------------------------------------
  // source byte offsets of this lane's 16-byte slot in each (half-tile kind, piece)
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int rr = (wave + 8 * c) * 8 + (lane >> 3);          // row of the half-tile image
    const uint32_t sl = (uint32_t)(((lane & 7) ^ ((rr >> 1) & 7)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      offA[h][c] = (uint32_t)((rr >> 6) * 128 + h * 64 + (rr & 63)) * (uint32_t)a.row_bytes + sl;
      offB[h][c] = (uint32_t)((rr >> 5) * QW + h * 32 + (rr & 31)) * (uint32_t)a.row_bytes + sl;
    }
  }
  // fragment read offsets inside a half-tile: row*128 + ((ks*(8/KS) + hh) ^ swz)*16
  const int swz = (r >> 1) & 7;
  uint32_t rdA[KS], rdB[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint32_t s = (uint32_t)(((ks * (8 / KS) + hh) ^ swz) << 4);
    rdA[ks] = (uint32_t)(wm * 64 + r) * 128 + s;     // + mt*TS*128
    rdB[ks] = (uint32_t)(wn * 32 + r) * 128 + s;     // + nt*TS*128
  }

  // ---- load cursors: k-tile T+1 (c1) and T+2 (c2) -----------------------------------
  struct Cur {
    const uint8_t* a;
    const uint8_t* q;
    int koff;
  };
  int lj = i0;          // item of cursor c2
  // physical 256-row tile of the rt-th tile of this launch (kernels.hpp: scan order); scalar
  auto phys_tile = [&](int rt) __attribute__((always_inline)) {
    return (int)__builtin_amdgcn_readfirstlane(
        (int)scan_phys_tile((uint32_t)((a.row_begin >> 8) + rt), a.perm_mul, a.perm_n, a.perm_inv));
  };
  auto tile_ptrs = [&](int j, Cur& c) __attribute__((always_inline)) {
    const int d = __builtin_amdgcn_readfirstlane(j / nq);   // keep the cursor in scalar registers
    const int rt = ONE_TILE ? 0 : (L2_TILES ? (d & 1) * 8 + xcd : d * 8 + xcd), qt = j - d * nq;
    c.a = a.A + (int64_t)phys_tile(rt) * 256 * a.row_bytes;
    c.q = a.Q + (int64_t)qt * QT * a.row_bytes;
  };
  auto advance = [&](Cur& c) __attribute__((always_inline)) {
    c.koff += 128;
    if (c.koff == (int)a.row_bytes) {
      c.koff = 0;
      if (lj + per_xcd < items_x) {   // past the last item: keep re-loading it (never read)
        lj += per_xcd;
        tile_ptrs(lj, c);
      }
    }
  };
  bool in_loop = false;
  auto stage1 = [&](const uint8_t* base, int koff, const uint32_t (&off)[2], int slot, int c) __attribute__((always_inline)) {
    if (NO_GLDS && in_loop) return;
    // (the empty asm keeps the zero-extension of the lane offset in this block: hoisted out of the loop as a 64-bit pair,
    // instruction selection no longer sees `scalar base + zext(32-bit lane offset)` and falls back to a 64-bit VALU add
    // per piece into ONE temporary pair -- a VALU->VMEM->VALU chain in the L segment -- instead of the saddr form)
    uint32_t o = off[c];
    asm volatile("" : "+v"(o));
    __builtin_amdgcn_global_load_lds(GLB_PTR(base + koff + o),
                                     LDS_PTR(lds + slot * S8_HT + (wave + 8 * c) * 1024), 16, 0, 0);
  };
  auto stage = [&](const uint8_t* base, int koff, const uint32_t (&off)[2], int slot) __attribute__((always_inline)) {
    stage1(base, koff, off, slot, 0);
    stage1(base, koff, off, slot, 1);
  };

  using frag_t = half8;
  using acc_t = typename std::conditional<
      KIND == KIND_F16, typename std::conditional<TS == 32, f32x16, f32x4>::type,
      typename std::conditional<TS == 32, i32x16, i32x4>::type>::type;
  acc_t acc[2][2][MT][NT];   // [ha][hb][mt][nt]
  frag_t af[MT][KS];         // current A half
  frag_t bA[NT][KS], bB[NT][KS];  // query fragments: B0 / B1 alternate between the two sets

  auto read_a = [&](int slot) __attribute__((always_inline)) {
    if (NO_READS && in_loop) return;
    const uint8_t* s = lds + slot * S8_HT;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) af[mt][ks] = *(const frag_t*)(s + rdA[ks] + mt * (TS * 128));
  };
  auto read_b = [&](int slot, frag_t (&b)[NT][KS]) __attribute__((always_inline)) {
    if (NO_READS && in_loop) return;
    const uint8_t* s = lds + slot * S8_HT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[nt][ks] = *(const frag_t*)(s + rdB[ks] + nt * (TS * 128));
  };

  // ---- prologue: half-tiles g = 0..5 ---------------------------------------------------
  Cur c1{}, c2{};
  c2.koff = 0;
  tile_ptrs(lj, c2);
  stage(c2.a, c2.koff, offA[0], 0);
  stage(c2.q, c2.koff, offB[0], 1);
  if (!HQ) stage(c2.q, c2.koff, offB[1], 2);
  stage(c2.a, c2.koff, offA[1], 3);
  advance(c2);                       // T = 1
  stage(c2.a, c2.koff, offA[0], 4);
  stage(c2.q, c2.koff, offB[0], 5);
  c1 = c2;
  advance(c2);                       // T = 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#if HX_S8_PH == 4
  read_b(1, bA);
#endif
  if (wm == 1) __builtin_amdgcn_s_barrier();   // the stagger
  if (NO_READS) {
    read_a(0);
    read_b(1, bA);
    read_b(2, bB);
  }

  int cj = i0, c_kt = 0;
  in_loop = true;
#if HX_S8_PRIO == 2
  if (wm == 1) __builtin_amdgcn_s_setprio(1);
#endif

  // quadrant epilogue: threshold filter; passing (key, query) pairs go to this wave's log
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int S8_ENTRY = 1 + MT * (EPT / 4);   // 16-byte words per log entry
  auto* g_log = (__attribute__((address_space(1))) u32x4*)a.hitlog +
                (int64_t)(blockIdx.x * 8 + wave) * a.logcap * S8_ENTRY;
  int wpos = 0;   // entries this wave has logged (scalar)
  // (Log stores count on the same vmcnt as the LDS-DMA loads, in issue order, so a counted wait behind an M segment
  // that logged also waits for the stores.  Letting the wait leave that many more operations in flight -- 5 / 10 / 20
  // by the stores issued -- measured nothing on the 10M step: 6.81-6.83 ms of scans without, 6.84-6.95 with, same box.)
  bool pend = false;           // the lower quadrants of the last finished item still wait for their filter
  int pend_rt = 0, pend_qt = 0;
  // Thresholds of the item's query columns (lane: column r of tile nt of half hb) and, int8, the row-scale bound of
  // its corpus tile: fetched once per item at its FIRST k-tile, so that the filter -- which runs in the M segment
  // and holds the SIMD's matrix slot while it does -- waits for no LDS or scalar load (it did: ~100 cycles of
  // ds_read latency and ~200 of s_load per quadrant, 13 % of the matrix cycles of a 6-k-tile item).
  // int8: lds_tau holds tau / rinv_q, rounded down (kernel prologue), and the column bound is f32(max dot) * rxm:
  // one multiplication, still an upper bound of every score of the column (k_scatter_log computes the exact ones).
  float tau_it[2][NT];
  float rxm_it = 0.f;
  auto fetch_thresholds = [&](int rt, int qt) __attribute__((always_inline)) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) tau_it[hb][nt] = lds_tau[qt * QT + wn * QW + (HQ ? 0 : hb * 32) + nt * TS + r];
    if constexpr (KIND == KIND_I8) {
      typedef __attribute__((address_space(4))) const float CF;   // constant address space + uniform index = s_load_dword
      rxm_it = ((CF*)a.rinv_tile_max)[rt];
    }
  };
  auto filter = [&](int rt, int qt, int ha, int hb, acc_t (&c)[MT][NT]) __attribute__((always_inline)) {
    if constexpr (NO_FILTER) {       // timing build: no filter at all (the accumulators are kept alive)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(c[mt][nt]));
      return;
    }
    // lane owns query column r of each of the NT tiles and, per row tile, EPT rows:
    //   TS = 32: rows 8*(e>>2) + 4*hh + (e&3);  TS = 16: rows 4*hh + e   (4 consecutive per group)
    // (rt is the PHYSICAL tile here: ktile maps it once per item)
    const int q0 = qt * QT + wn * QW + (HQ ? 0 : hb * 32) + r;
    const int64_t rowq = (int64_t)rt * 256 + wm * 128 + ha * 64 + 4 * hh;
    // (plain fmaxf / max chains: hipcc forms v_max3 itself, and -- unlike an inline-asm v_max3 --
    // gets the MFMA-result wait states its hazard recognizer inserts for instructions it knows)
    auto max3 = [](float x, float y, float z) __attribute__((always_inline)) {
      return __builtin_fmaxf(__builtin_fmaxf(x, y), z);
    };
    auto imax3 = [](int x, int y, int z) __attribute__((always_inline)) {
      const int t = x > y ? x : y;
      return t > z ? t : z;
    };
    (void)max3;
    (void)imax3;
    constexpr int NG = EPT / 4;   // groups of 4 consecutive rows per tile
    // m[nt] >= (an upper bound of) the best score of the lane's column of tile nt
    float tau[NT], m[NT];
    const float rxm = rxm_it;
    bool any = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      tau[nt] = hb ? tau_it[1][nt] : tau_it[0][nt];
      if constexpr (KIND == KIND_F16) {
        // (v_max3_f32 returns the maximum of the non-NaN operands; garbage rows past the end of
        // the matrix may hold NaNs and are dropped by the row < row_end test anyway)
        float mm = c[0][nt][0];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            mm = max3(max3(mm, c[mt][nt][4 * g], c[mt][nt][4 * g + 1]), c[mt][nt][4 * g + 2], c[mt][nt][4 * g + 3]);
        m[nt] = mm;
      } else {
        // int8: score = (f32(dot) * rinv_x[row]) * rinv_q[q], all factors >= 0 for dot > 0 and
        // rounding is monotone, so (f32(max dot) * max rinv_x of the tile) * rinv_q bounds the
        // column from above without touching the per-row scales (a VGPR-destination load here would make
        // hipcc drain the LDS-DMA queue); k_scatter_log computes the exact scores.
        int im = c[0][nt][0];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            im = imax3(imax3(im, c[mt][nt][4 * g], c[mt][nt][4 * g + 1]), c[mt][nt][4 * g + 2], c[mt][nt][4 * g + 3]);
        m[nt] = im > 0 ? (float)im * rxm : 0.f;
      }
      any |= (m[nt] >= tau[nt]);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) == 0ull, 1)) return;
    // Rare path, kept SHORT (it is inlined in every phase of every k-tile body, and a long one
    // pushed the kernel far past the instruction cache): a lane whose column reached its
    // threshold (bound) logs the whole column -- {query, first row} + its MT x EPT accumulators
    // (fp16: the scores; int8: the integer dots) -- with plain stores; k_scatter_log picks the
    // passing rows.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool hit = m[nt] >= tau[nt];
      const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
      if (mask == 0ull) continue;
      const int idx = wpos + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      if (hit) {
        const int q = q0 + nt * TS;
        if (idx < a.logcap) {
          auto* e = g_log + (int64_t)idx * S8_ENTRY;
          e[0] = u32x4{(uint32_t)q, (uint32_t)rowq, (uint32_t)((uint64_t)rowq >> 32), 0u};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              // (element copies first: __builtin_bit_cast applied directly to an ext-vector element
              // reference yields element 0 -- hipcc, ROCm 7.2)
              const auto x0 = c[mt][nt][4 * g + 0], x1 = c[mt][nt][4 * g + 1], x2 = c[mt][nt][4 * g + 2],
                         x3 = c[mt][nt][4 * g + 3];
              e[1 + mt * NG + g] = u32x4{__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, x1),
                                         __builtin_bit_cast(uint32_t, x2), __builtin_bit_cast(uint32_t, x3)};
            }
        } else {
          g_ovf[q] = 1;
        }
      }
      wpos += __builtin_popcountll(mask);
    }
  };

  // MFMAs of k-steps [ks0, ks1) of one quadrant
  auto mma = [&](acc_t (&c)[MT][NT], const frag_t (&b)[NT][KS], bool first, int ks0, int ks1) __attribute__((always_inline)) {
    if constexpr (DBG == 3) {
#pragma unroll
      for (int ks = ks0; ks < ks1; ++ks) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(b[nt][ks]));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(af[mt][ks]));
      }
      if (first && ks0 == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) c[mt][nt] = acc_t{};
      }
      return;
    }
#pragma unroll
    for (int ks = ks0; ks < ks1; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const acc_t z = acc_t{};
          const acc_t cin = (first && ks == 0) ? z : c[mt][nt];
          if constexpr (KIND == KIND_F16 && TS == 32)
            c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][ks], b[nt][ks], cin, 0, 0, 0);
          else if constexpr (KIND == KIND_F16 && TS == 16)
            c[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][ks], b[nt][ks], cin, 0, 0, 0);
          else if constexpr (TS == 32)
            c[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, af[mt][ks]),
                                                              __builtin_bit_cast(i32x4, b[nt][ks]), cin, 0, 0, 0);
          else
            c[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[mt][ks]),
                                                              __builtin_bit_cast(i32x4, b[nt][ks]), cin, 0, 0, 0);
        }
  };

#ifndef HX_S8_SPLIT
#define HX_S8_SPLIT 2
#endif
  // Where the two loads of a phase are issued: 0 = both in the L segment (vmcnt 6), 1 = one in L
  // and one between the MFMAs (vmcnt 5), 2 = both between the MFMAs (vmcnt 4).
#define S8_WAIT() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 - HX_S8_SPLIT) : "memory")
#define S8_PHASE(READS, BASE, KOFF, OFF, SLOT, ACC, BF, HA, HB)                 \
  READS;                                                                        \
  if (HX_S8_SPLIT <= 1) stage1(BASE, KOFF, OFF, SLOT, 0);                       \
  if (HX_S8_SPLIT == 0) stage1(BASE, KOFF, OFF, SLOT, 1);                       \
  S8_WAIT();                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                            \
  __builtin_amdgcn_s_barrier();                                                 \
  __builtin_amdgcn_sched_barrier(0);                                            \
  __builtin_amdgcn_s_setprio(1);                                                \
  mma(ACC, BF, FIRST, 0, 1);                                                    \
  if (HX_S8_SPLIT >= 1) {                                                       \
    __builtin_amdgcn_sched_barrier(0);                                          \
    stage1(BASE, KOFF, OFF, SLOT, HX_S8_SPLIT == 1 ? 1 : 0);                    \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }                                                                             \
  mma(ACC, BF, FIRST, 1, KS == 4 ? 3 : 2);                                      \
  if (HX_S8_SPLIT == 2) {                                                       \
    __builtin_amdgcn_sched_barrier(0);                                          \
    stage1(BASE, KOFF, OFF, SLOT, 1);                                           \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }                                                                             \
  mma(ACC, BF, FIRST, KS == 4 ? 3 : 2, KS);                                     \
  if (__builtin_expect(last, 0)) filter(rt, qt, HA, HB, ACC);                   \
  __builtin_amdgcn_s_setprio(0);                                                \
  __builtin_amdgcn_sched_barrier(0);                                            \
  __builtin_amdgcn_s_barrier();                                                 \
  __builtin_amdgcn_sched_barrier(0);

#if HX_S8_PH == 4
  // one k-tile = four phases.  PAR = T & 1 (ring half and the B register set holding B0).
  auto ktile = [&](auto par_c, auto first_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int S0 = 4 * PAR, N0 = 4 * (1 - PAR);
    frag_t(&b0)[NT][KS] = PAR ? bB : bA;
    frag_t(&b1)[NT][KS] = PAR ? bA : bB;
    const bool last = (c_kt == KT - 1);
    int rt = 0, qt = 0;
    if (last) {
      const int d = __builtin_amdgcn_readfirstlane(cj / nq);
      rt = phys_tile(d * 8 + xcd);
      qt = cj - d * nq;
    }

    // C00 += A0.B0 | C01 += A0.B1 | C11 += A1.B1 | C10 += A1.B0 (B0 of the next k-tile replaces B1)
    S8_PHASE(read_a(S0 + 0), c1.q, c1.koff, offB[1], N0 + 2, acc[0][0], b0, 0, 0)
    S8_PHASE(read_b(S0 + 2, b1), c1.a, c1.koff, offA[1], N0 + 3, acc[0][1], b1, 0, 1)
    S8_PHASE(read_a(S0 + 3), c2.a, c2.koff, offA[0], S0 + 0, acc[1][1], b1, 1, 1)
    S8_PHASE(read_b(N0 + 1, b1), c2.q, c2.koff, offB[0], S0 + 1, acc[1][0], b0, 1, 0)

    c1 = c2;
    advance(c2);
    c_kt = last ? 0 : c_kt + 1;
    cj += last ? per_xcd : 0;
  };
#else
  // Two phases per k-tile, two quadrants each: half the barriers per MFMA.
  //   X: reads A0, B0, B1 | C00 += A0.B0, C01 += A0.B1 | stages B1, A1 of T+1 | vmcnt(4)
  //   Y: reads A1         | C11 += A1.B1, C10 += A1.B0 | stages A0, B0 of T+2 | vmcnt(2)
  // Half-tile g = 4T + {A0, B0, B1, A1} is read in phase X (i < 3) or Y of k-tile T and staged
  // 6 half-tiles ahead, always in an M segment.  RAW: the wait of X retires g <= 4T+3 (g = 4T+4,
  // 4T+5 may be in flight), the wait of Y retires g <= 4T+6 (only 4T+7 in flight), each ahead
  // of the barrier before the reading phase.  WAR: slot(g) is restaged by g+8 in the M segment
  // of the phase after its last read, i.e. behind one more barrier than the reads.
#ifndef HX_S8_FILTER_SPLIT
#define HX_S8_FILTER_SPLIT 0  // 1: the upper quadrants in phase Y of the last k-tile, the lower ones in the next phase X
#endif                        //    (spills: the Y site has the B fragments live); 0: the whole tile in the next phase X
#ifndef HX_S8_FILTER_IN_M
#define HX_S8_FILTER_IN_M 1   // 1: the filter behind each quadrant's MFMAs, inside the M segment; 2: both quadrants' filters behind
                              //    the phase's last MFMA (measured: 7.55 ms per step's scans against 7.39-7.48 for 1); 0: deferred to the next L
                              //    segments (tried in round 3: hipcc then spills 36-380 bytes per lane INSIDE the k-loop, behind vmcnt(0))
#endif
#ifndef HX_S8_LDMA
#define HX_S8_LDMA 1   // where the four global_load_lds of a phase are issued: 0 between the MFMAs of its M segment (round 1),
#endif                 // 1 in its L segment -- by the wave that is NOT on the matrix pipe -- 2 half and half

#define S8_QUAD(ACC, BF, HA, HB, BASE, KOFF, OFF, SLOT, INM)                    \
  mma(ACC, BF, FIRST, 0, KS / 2);                                               \
  if (INM) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                          \
    stage1(BASE, KOFF, OFF, SLOT, 0);                                           \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }                                                                             \
  mma(ACC, BF, FIRST, KS / 2, KS);                                              \
  if (INM) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                          \
    stage1(BASE, KOFF, OFF, SLOT, 1);                                           \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }                                                                             \
  if (HX_S8_FILTER_IN_M == 1 && __builtin_expect(last, 0)) filter(rt, qt, HA, HB, ACC);
#define S8_L_END(N)                                                             \
  if (!NO_VMWAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");      \
  __builtin_amdgcn_sched_barrier(0);                                            \
  if (!NO_BAR) __builtin_amdgcn_s_barrier();                                    \
  __builtin_amdgcn_sched_barrier(0);                                            \
  if (HX_S8_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#define S8_M_END()                                                              \
  if (HX_S8_PRIO == 1) __builtin_amdgcn_s_setprio(0);                           \
  __builtin_amdgcn_sched_barrier(0);                                            \
  if (!NO_BAR) __builtin_amdgcn_s_barrier();                                    \
  __builtin_amdgcn_sched_barrier(0);
  // The threshold filter of a finished tile runs in the L segments that FOLLOW its last MFMAs, not behind them in the
  // M segment: there this wave would hold the SIMD's matrix slot without using it (its partner is in its L segment,
  // behind the barrier) -- ~50 VALU instructions per quadrant, four quadrants per tile: 6.5 % of the matrix cycles of
  // a 12-k-tile item (fp16, dim 768) and 13 % of a 6-k-tile one (int8).  C00, C01 are complete after phase X of the
  // last k-tile and are filtered in its phase Y; C11, C10 are complete after phase Y and are filtered in phase X of
  // the NEXT k-tile (the first of the next item, which overwrites them in ITS phase Y only), or after the loop.
  auto ktile = [&](auto par_c, auto first_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int S0 = 4 * PAR, N0 = 4 * (1 - PAR);
    const bool last = (c_kt == KT - 1);
    int rt = 0, qt = 0;
    if (FIRST || last) {
      const int d = __builtin_amdgcn_readfirstlane(cj / nq);
      rt = phys_tile(d * 8 + xcd);
      qt = cj - d * nq;
    }
    if (FIRST) fetch_thresholds(rt, qt);
    // phase X  (the filter comes first: the fragment registers of the previous phase are dead, the new ones not yet
    // loaded -- placed behind the reads it pushed the kernel over its 256 VGPRs)
    if (!HX_S8_FILTER_IN_M && FIRST && __builtin_expect(pend, 1)) {     // the previous item's tile
#if HX_S8_FILTER_SPLIT == 0
      filter(pend_rt, pend_qt, 0, 0, acc[0][0]);
      filter(pend_rt, pend_qt, 0, 1, acc[0][1]);
#endif
      filter(pend_rt, pend_qt, 1, 1, acc[1][1]);
      filter(pend_rt, pend_qt, 1, 0, acc[1][0]);
      pend = false;
      __builtin_amdgcn_sched_barrier(0);
    }
    read_a(S0 + 0);
    read_b(S0 + 1, bA);
    if (!HQ) read_b(S0 + 2, bB);
    // HX_S8_LDMA 1: both half-tiles of the phase staged here (the wait then leaves 8 pieces in flight: g = 4T+4 ..
    // 4T+7); 2: B1 here, A1 between the MFMAs of the second quadrant (6 in flight: 4T+4 .. 4T+6)
    // HQ: there is no B1 -- A1 of T + 1 alone, 6 pieces stay in flight (g = 4T+4, 4T+5, 4T+7)
    if (HX_S8_LDMA) {
      __builtin_amdgcn_sched_barrier(0);
      if (!HQ) stage(c1.q, c1.koff, offB[1], N0 + 2);
      if (HX_S8_LDMA == 1 || HQ) stage(c1.a, c1.koff, offA[1], N0 + 3);
    }
    S8_L_END(HQ ? 6 : (HX_S8_LDMA == 1 ? 8 : (HX_S8_LDMA == 2 ? 6 : 4)))
    S8_QUAD(acc[0][0], bA, 0, 0, c1.q, c1.koff, offB[1], N0 + 2, HX_S8_LDMA == 0 && !HQ)
    if (!HQ) { S8_QUAD(acc[0][1], bB, 0, 1, c1.a, c1.koff, offA[1], N0 + 3, HX_S8_LDMA != 1) }
    if (HX_S8_FILTER_IN_M == 2 && __builtin_expect(last, 0)) {   // both quadrants behind the phase's last MFMA: the first
      filter(rt, qt, 0, 0, acc[0][0]);                           // one's results are long there, its VALU work runs under
      filter(rt, qt, 0, 1, acc[0][1]);                           // the second one's MFMAs still in the pipe
    }
    S8_M_END()
    // phase Y
#if HX_S8_FILTER_SPLIT
    if (!HX_S8_FILTER_IN_M && __builtin_expect(last, 0)) {              // this item's upper quadrants
      filter(rt, qt, 0, 0, acc[0][0]);
      filter(rt, qt, 0, 1, acc[0][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    read_a(S0 + 3);
    if (HX_S8_LDMA) {      // (6 pieces in flight: g = 4T+7 .. 4T+9)
      __builtin_amdgcn_sched_barrier(0);
      stage(c2.a, c2.koff, offA[0], S0 + 0);
      stage(c2.q, c2.koff, offB[0], S0 + 1);
    }
    S8_L_END(HX_S8_LDMA ? 6 : 2)
    if (!HQ) { S8_QUAD(acc[1][1], bB, 1, 1, c2.a, c2.koff, offA[0], S0 + 0, HX_S8_LDMA == 0) }
    S8_QUAD(acc[1][0], bA, 1, 0, c2.q, c2.koff, offB[0], S0 + 1, HX_S8_LDMA == 0)
    if (HX_S8_FILTER_IN_M == 2 && __builtin_expect(last, 0)) {
      filter(rt, qt, 1, 1, acc[1][1]);
      filter(rt, qt, 1, 0, acc[1][0]);
    }
    S8_M_END()
    if (!HX_S8_FILTER_IN_M && last) {
      pend = true;
      pend_rt = rt;
      pend_qt = qt;
    }

    c1 = c2;
    advance(c2);
    c_kt = last ? 0 : c_kt + 1;
    cj += last ? per_xcd : 0;
  };
#undef S8_QUAD
#undef S8_L_END
#undef S8_M_END
#endif
  auto ktile_any = [&](auto par_c) __attribute__((always_inline)) {
    if (c_kt == 0)
      ktile(par_c, std::true_type{});
    else
      ktile(par_c, std::false_type{});
  };

  int T = 0;
  for (; T + 1 < total_T; T += 2) {
    ktile_any(std::integral_constant<int, 0>{});
    ktile_any(std::integral_constant<int, 1>{});
  }
  if (T < total_T) ktile_any(std::integral_constant<int, 0>{});
#if HX_S8_PH != 4
  if (!HX_S8_FILTER_IN_M && pend) {                  // the last item's tile
#if HX_S8_FILTER_SPLIT == 0
    filter(pend_rt, pend_qt, 0, 0, acc[0][0]);
    filter(pend_rt, pend_qt, 0, 1, acc[0][1]);
#endif
    filter(pend_rt, pend_qt, 1, 1, acc[1][1]);
    filter(pend_rt, pend_qt, 1, 0, acc[1][0]);
  }
#endif

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the never-read tail loads
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if (lane == 0) a.hitcnt[blockIdx.x * 8 + wave] = wpos;
#undef S8_PHASE
#undef S8_WAIT
}

// Logs -> per-query candidate buffers.  Every entry is a lane's column of one quadrant (see the
// filter); rows whose score reaches the query's threshold are appended (order is irrelevant,
// k_compact sorts).  tau is what the scan used: nothing updates it in between.
//
// Device-scope atomics run at the memory side (the XCDs' L2s are not coherent): one returning
// atomic per candidate made this kernel take 0.25-0.5 ms whatever the launch (2M scattered atomics).
// A workgroup therefore takes the logs of S8_SB scan workgroups x the two waves that share a query
// column group (wave & 3), i.e. at most 64 * nq_tiles distinct queries: pass 1 counts per query in
// LDS, ONE global atomic per (workgroup, query) reserves a range, pass 2 re-reads the logs and
// places the keys with LDS atomics.
#ifndef HX_S8_WPL
#define HX_S8_WPL 2     // waves of the scatter workgroup per log (the walk is a chain of dependent loads per entry: more waves
#endif                  // per log shorten it, at the price of more workgroups and one global atomic per workgroup and query;
                        // everything of a dense search but its scan kernels, B = 1024, L = 100: 0.857 / 0.75 / 0.747 ms at 1 / 2 / 4)
constexpr int S8_WPL = HX_S8_WPL;
constexpr int S8_SB = 8 / S8_WPL;
template <int KIND, int TS, bool HQ = false>
__global__ __launch_bounds__(1024) void k_scatter_log(uint4* log, const int* __restrict__ hitcnt,
                                                      int logcap, int n_scan_blocks, int nq_tiles,
                                                      const float* __restrict__ tau, int64_t row_end, int64_t id_base,
                                                      const float* __restrict__ rinv_x, const float* __restrict__ rinv_q,
                                                      uint64_t* __restrict__ cand, int* __restrict__ cnt,
                                                      int* __restrict__ ovf, int cap) {
  constexpr int MT = 64 / TS, EPT = TS * TS / 64, NG = EPT / 4, ENTRY = 1 + MT * NG;
  static_assert(MT * NG * 4 <= 32, "one mask bit per logged score");
  __shared__ int lcnt[S8_MAXQ / 4];    // per query of the column group: candidates, then next slot
  const int tid = threadIdx.x;
  const int wn = blockIdx.x & 3;                      // query column group
  const int sb0 = (blockIdx.x >> 2) * S8_SB;          // first scan workgroup
  constexpr int QT = HQ ? 128 : 256, QW = HQ ? 32 : 64, QTS = HQ ? 7 : 8;   // (k_scan8: queries per tile / per wave column)
  const int nloc = nq_tiles * QW;                     // queries of the group: q = qt*QT + wn*QW + j
  for (int i = tid; i < nloc; i += 1024) lcnt[i] = 0;
  __syncthreads();
  static_assert(2 * S8_SB * S8_WPL == 1024 / 64, "S8_WPL waves of the workgroup per log");
  const int l = (tid >> 6) / S8_WPL, lane = tid & 63; // waves l * S8_WPL ... walk log l: the logs advance together
  const int sub = (tid >> 6) % S8_WPL;
  const int sb = sb0 + (l >> 1);
  const int w = sb * 8 + (l & 1) * 4 + wn;            // waves wn and wn + 4 of the scan workgroup
  int n = sb < n_scan_blocks ? hitcnt[w] : 0;
  n = n < logcap ? n : logcap;
  uint4* base = log + (int64_t)w * logcap * ENTRY;
  typedef float f4 __attribute__((ext_vector_type(4)));
  // exact score of logged element (mt, g, k) of an entry: fp16 the score itself, int8 the oracle's (f32(dot) * rinv_x) * rinv_q
  // Pass 0 scores all 16 elements of an entry (four consecutive rows per word: their scales come as one 16-byte load,
  // not four gathers), counts the passing ones with ONE LDS atomic per entry and leaves their bit mask in the
  // entry's spare header word; pass 1 -- after the workgroup reserved a range per query -- touches only entries with a
  // mask and only their passing elements (about one of sixteen), with one LDS atomic per entry again.
  for (int i = lane + 64 * sub; i < n; i += 64 * S8_WPL) {
    uint4* e = base + (int64_t)i * ENTRY;
    const uint4 h = e[0];
    const int q = (int)h.x;
    const int64_t row0 = (int64_t)(((uint64_t)h.z << 32) | h.y);
    const float t = tau[q];
    float rq = 0.f;
    if constexpr (KIND == KIND_I8) rq = rinv_q[q];
    uint32_t mask = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const uint4 v = e[1 + mt * NG + g];
        const uint32_t bits[4] = {v.x, v.y, v.z, v.w};
        const int64_t r4 = row0 + mt * TS + 8 * g;    // multiple of 4: the four scales are one aligned load
        f4 sc = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (KIND == KIND_I8) sc = *(const f4*)(rinv_x + r4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s;
          if constexpr (KIND == KIND_F16) s = __builtin_bit_cast(float, bits[k]);
          else s = __fmul_rn(__fmul_rn((float)(int)bits[k], sc[k]), rq);
          if (r4 + k < row_end && s >= t) mask |= 1u << ((mt * NG + g) * 4 + k);
        }
      }
    ((uint32_t*)e)[3] = mask;
    if (mask) atomicAdd(&lcnt[(q >> QTS) * QW + (q & (QW - 1))], __builtin_popcount(mask));
  }
  __syncthreads();
  for (int i = tid; i < nloc; i += 1024) {
    const int c = lcnt[i];
    const int q = (i / QW) * QT + wn * QW + (i & (QW - 1));
    lcnt[i] = c > 0 ? atomicAdd(cnt + q, c) : 0;   // first slot of this workgroup's range
  }
  __syncthreads();
  for (int i = lane + 64 * sub; i < n; i += 64 * S8_WPL) {
    const uint4* e = base + (int64_t)i * ENTRY;
    const uint4 h = e[0];                          // (its mask word was written by this very lane above)
    uint32_t mask = h.w;
    if (!mask) continue;
    const int q = (int)h.x;
    const int64_t row0 = (int64_t)(((uint64_t)h.z << 32) | h.y);
    float rq = 0.f;
    if constexpr (KIND == KIND_I8) rq = rinv_q[q];
    int pos = atomicAdd(&lcnt[(q >> QTS) * QW + (q & (QW - 1))], __builtin_popcount(mask));
    while (mask) {
      const int bit = __builtin_ctz(mask);
      mask &= mask - 1;
      const int word = bit >> 2, k = bit & 3;      // word = mt * NG + g
      const int64_t row = row0 + (word / NG) * TS + 8 * (word % NG) + k;
      const uint32_t raw = ((const uint32_t*)(e + 1 + word))[k];
      float s;
      if constexpr (KIND == KIND_F16) s = __builtin_bit_cast(float, raw);
      else s = __fmul_rn(__fmul_rn((float)(int)raw, rinv_x[row]), rq);   // the oracle's int8 score, as in pass 0
      if (pos < cap) cand[(int64_t)q * cap + pos] = make_key(s, (uint32_t)(id_base + row));
      else ovf[q] = 1;
      ++pos;
    }
  }
}

bool scan8_usable(const ScanArgs& a, int bn) {
  return (bn == 256 || (bn == 128 && a.half_q)) && a.hitlog != nullptr && a.B <= S8_MAXQ && (a.row_begin & 255) == 0 &&
         a.row_bytes * 256 < (1ll << 31);
}

void launch_scan8(const ScanArgs& a, int kind, hipStream_t st, hipEvent_t after_kernel) {
  const int64_t n_rows = a.row_end - a.row_begin;
  if (n_rows <= 0 || a.B <= 0) return;
  HX_CHECK((a.row_bytes & 127) == 0, "scan: row_bytes must be a multiple of 128");
  const int64_t tiles = (n_rows + 255) / 256 * a.nq_tiles;
  HX_CHECK(tiles * (a.row_bytes >> 7) < (1ll << 31), "scan: launch too large");
  HX_CHECK(a.hitlog && a.hitcnt && a.logcap > 0, "scan8: no hit log");
  static const int grid_cap = getenv("HX_DEBUG_SCAN8_GRID") ? atoi(getenv("HX_DEBUG_SCAN8_GRID")) : 256;   // diagnostics
  const int cap = (grid_cap >= 8 && grid_cap <= 256) ? grid_cap / 8 * 8 : 256;
  int64_t g = tiles < cap ? tiles : cap;
  g = (g + 7) / 8 * 8;
#ifdef HX_SCAN_DBG
  static const int dbg = getenv("HX_SCAN_DBG") ? atoi(getenv("HX_SCAN_DBG")) : 0;
#define HX_DBG_CASE(K, D) \
  if (kind == K && dbg == D) hipLaunchKernelGGL((k_scan8<K, HX_S8_TS, D>), dim3((unsigned)g), dim3(512), 0, st, a); else
  HX_DBG_CASE(KIND_F16, 1) HX_DBG_CASE(KIND_F16, 2) HX_DBG_CASE(KIND_F16, 3) HX_DBG_CASE(KIND_F16, 4)
  HX_DBG_CASE(KIND_I8, 1) HX_DBG_CASE(KIND_I8, 2) HX_DBG_CASE(KIND_I8, 3) HX_DBG_CASE(KIND_I8, 4)
  HX_DBG_CASE(KIND_I8, 5) HX_DBG_CASE(KIND_I8, 6) HX_DBG_CASE(KIND_I8, 7) HX_DBG_CASE(KIND_I8, 8)
  HX_DBG_CASE(KIND_I8, 9) HX_DBG_CASE(KIND_I8, 10) HX_DBG_CASE(KIND_I8, 11)
#undef HX_DBG_CASE
#endif
  if (a.half_q) {      // 256 rows x 128 queries (65..128 queries)
    if (kind == KIND_F16)
      hipLaunchKernelGGL((k_scan8<KIND_F16, HX_S8_TS, 0, true>), dim3((unsigned)g), dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL((k_scan8<KIND_I8, HX_S8_TS, 0, true>), dim3((unsigned)g), dim3(512), 0, st, a);
  } else if (kind == KIND_F16)
    hipLaunchKernelGGL((k_scan8<KIND_F16, HX_S8_TS, 0>), dim3((unsigned)g), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((k_scan8<KIND_I8, HX_S8_TS, 0>), dim3((unsigned)g), dim3(512), 0, st, a);
  HX_HIP(hipGetLastError());
  if (after_kernel) HX_HIP(hipEventRecord(after_kernel, st));     // the profile times k_scan8 alone, not its log scatter
  const unsigned sg = (unsigned)((g + S8_SB - 1) / S8_SB) * 4;
  if (a.half_q) {
    if (kind == KIND_F16)
      hipLaunchKernelGGL((k_scatter_log<KIND_F16, HX_S8_TS, true>), dim3(sg), dim3(1024), 0, st, a.hitlog, a.hitcnt,
                         a.logcap, (int)g, a.nq_tiles, a.tau, a.n_total, a.id_base, a.rinv_x, a.rinv_q, a.cand, a.cnt,
                         a.overflow, a.cap);
    else
      hipLaunchKernelGGL((k_scatter_log<KIND_I8, HX_S8_TS, true>), dim3(sg), dim3(1024), 0, st, a.hitlog, a.hitcnt,
                         a.logcap, (int)g, a.nq_tiles, a.tau, a.n_total, a.id_base, a.rinv_x, a.rinv_q, a.cand, a.cnt,
                         a.overflow, a.cap);
  } else if (kind == KIND_F16)
    hipLaunchKernelGGL((k_scatter_log<KIND_F16, HX_S8_TS>), dim3(sg), dim3(1024), 0, st, a.hitlog, a.hitcnt, a.logcap,
                       (int)g, a.nq_tiles, a.tau, a.n_total, a.id_base, a.rinv_x, a.rinv_q, a.cand, a.cnt, a.overflow,
                       a.cap);
  else
    hipLaunchKernelGGL((k_scatter_log<KIND_I8, HX_S8_TS>), dim3(sg), dim3(1024), 0, st, a.hitlog, a.hitcnt, a.logcap,
                       (int)g, a.nq_tiles, a.tau, a.n_total, a.id_base, a.rinv_x, a.rinv_q, a.cand, a.cnt, a.overflow,
                       a.cap);
  HX_HIP(hipGetLastError());
}

}  // namespace hx
