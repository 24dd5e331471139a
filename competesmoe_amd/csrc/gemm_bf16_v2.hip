// Grouped expert GEMM v2 for gfx950 (row-space launches; the weight-gradient launch uses the persistent twin in
// gemm_bf16_v2p.hip -- kept as two translation units because the persistent loop changes hipcc's register allocation of the
// row-space kernels for the worse, measured -8..-11 %): 256x256 output tile, 512 threads (8 waves as 2 row-halves x 4 column-quarters, each
// wave 128x64 = 8x4 v_mfma_f32_16x16x32_bf16 accumulators), K-tile 64, 128 KiB of LDS, one workgroup per CU.
//
// LDS holds 4 kinds of 16 KiB images (the KC / KM images of gemm_tiles.h), two slots each (K-tile parity):
//     RL = rows  0..63  of both row halves      (what every wave reads in phase 1)      CL = columns  0..31 of the 4 quarters
//     RH = rows 64..127 of both row halves      (phase 3)                               CH = columns 32..63 of the 4 quarters
// Every K-tile s is 4 phases; a phase = {fragment ds_reads, issue ONE image by LDS-DMA (2 x 1 KiB per wave), counted vmcnt,
// lgkmcnt(0), s_barrier A, 16 MFMA (a 64x32 quadrant of the wave tile over K=64), s_barrier B}; wave (wm, wn) owns rows
// wm*64+0..63 of BOTH row images and columns wn*32+0..31 of BOTH column images, so each image is consumed in exactly one phase:
//     phase 1: read CL,RL(s)   MFMA C_lo x R_lo        phase 3: read RH(s)    MFMA C_hi x R_hi
//     phase 2: read CH(s)      MFMA C_hi x R_lo        phase 4: (C_lo kept)   MFMA C_lo x R_hi
// Two DMA schedules (template SCHED), both re-filling a slot only after the phase that read it and waiting with a COUNTED
// vmcnt so the loop never drains the DMA queue (guide §5 "Pipelining across barriers"):
//     SHALLOW: P1 RL(s+1)  P2 RH(s+1)  P3 CL(s+2)  P4 CH(s+2), vmcnt(4) in P4      (2-4 images in flight)
//     DEEP   : P1 RH(s+1)  P2 RL(s+2)  P3 CL(s+2)  P4 CH(s+2), vmcnt(10) in P1,P2,P4 (5-6 images = 80-96 KiB in flight)
//     WIDE   : TWO phases of 32 MFMA per K-tile (A: CL,RL,RH x C_lo; B: CH x C_hi), 2 images issued per phase, vmcnt(8)/(6)
// Measured (profiles/r01): WIDE > DEEP > SHALLOW on every layout (half the barriers: +6..16 %).
//     BAL    : WIDE's two phases cut by ROW image instead of by column image (A: CL,CH,RL x R_lo; B: RH x R_hi): 16 + 8 fragment
//              reads per phase instead of 20 + 4, one image issued in A and three in B, every image two phases ahead
// Same-box A/B at the headline launches (tools/gemm_bench.py, ms per launch GEMM1 / GEMM2-shaped): NT 5.80 / 4.89 with BAL against
// 6.02 / 5.12 with WIDE; NN 6.03 / 5.19 with BAL against 5.87 / 4.96 with WIDE.  Default: BAL for NT, WIDE for NN;
// CSMOE_GEMM_SCHED=1|2|3 forces one for A/B runs.  Two more issue placements were measured and removed: WIDE with one image
// issued in phase A and three in phase B (NT +1 %, NN -0.5 %), BAL with two images issued in each phase (NT +2 %, NN -3 %), BAL with
// all four images issued in phase B (RH one phase ahead only: NT -1 %, NN -6 %).
// Row half 1 (waves 4-7, the SIMD partners of waves 0-3) runs half a phase behind: one hardware barrier is A for one group and
// B for the other, so one group's ds_reads / DMA issue overlap its partners' MFMAs ("Two waves per SIMD" item 9 of the
// microarch guide; +9 % here).  Safety under the stagger: reads are retired (lgkmcnt(0)) and the counted vmcnt is taken BEFORE
// barrier A, so whichever group is ahead can neither re-fill a slot the other still reads nor read an image the other has not
// finished fetching.  K-tiles past the end are still "issued": their offsets are out of range, the buffer descriptor turns
// them into zero-fills of slots nobody reads, which keeps the wait counts uniform.
// In-kernel s_memtime stamps of this loop (diagnostic build of round 1; its listing was not kept -- the persistent twin's is
// profiles/r01/wgrad_stamps.txt): per phase ~450 cycles of
// read/issue/wait, ~380 of MFMA and ~170 of release latency per barrier -- an LDS-DMA issue costs its wave 100-185 cycles.
#include "gemm_epilogue.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

using namespace ggt;

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int CT2_LD = BN2 + 4;              // fp32 epilogue staging row stride (floats); 128 rows per pass
constexpr int LDS2_BYTES = EPI_LDS_BYTES;   // 135,168 B: the epilogue's bf16 staging tile (>= the 8 operand images = 131,072 B)

enum { WIDE = 2, BAL = 3 };

// Diagnostic build (-DCSMOE_STAMPS, tools/tile_stamps.py): every workgroup records where it ran (XCC / SE / CU) and the 100 MHz
// s_memrealtime at entry, at the start and the end of its K-loop and at exit, into a buffer of their own (CSMOE_STAMP_FILE gets
// the raw records after the launch): the gaps between one workgroup's exit and the next one's entry on the same CU are the
// re-dispatch cost of the non-persistent grid.
#ifdef CSMOE_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define RT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define RT_STAMP(var) do { } while (0)
#endif

template <int ROWK, int COLK, int MODE, int SCHED>
__global__ void __launch_bounds__(512, 2) gg8_kernel(FastArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  RT_STAMP(st_entry);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;       // row half, column quarter

  // ---------------- tile lookup ----------------
  int e, row0 = 0, rows = 0, tr0 = 0, tc0 = 0, red_len;
  const int nct = (p.NC + BN2 - 1) / BN2;
  if (MODE == 0) {
    TilePos tp;
    const int total = grouped_total_tiles(p.offsets, p.E, p.single_M, BM2, nct, lane, p.row_part);
    if ((int)blockIdx.x >= total) return;
    const int v = xcd_remap(blockIdx.x, total);
    if (!grouped_find_tile(p.offsets, p.E, p.single_M, BM2, nct, v, lane, tp, p.row_part)) return;
    e = tp.e;
    row0 = tp.o0 + tp.mt * BM2; rows = min(BM2, tp.o1 - row0);
    tc0 = tp.nt * BN2;
    red_len = p.Kd;
  } else {
    const int nrt = (p.NR + BM2 - 1) / BM2;
    const int per_e = nrt * nct;
    int v = xcd_remap(blockIdx.x, per_e * p.E);
    e = v / per_e;
    int local = v - e * per_e;
    tr0 = (local / nct) * BM2; tc0 = (local % nct) * BN2;
    row0 = p.offsets ? p.offsets[e] : 0;
    red_len = (p.offsets ? p.offsets[e + 1] : p.single_M) - row0;
  }
  e = __builtin_amdgcn_readfirstlane(e);
  row0 = __builtin_amdgcn_readfirstlane(row0);
  rows = __builtin_amdgcn_readfirstlane(rows);
  tr0 = __builtin_amdgcn_readfirstlane(tr0);
  tc0 = __builtin_amdgcn_readfirstlane(tc0);
  red_len = __builtin_amdgcn_readfirstlane(red_len);

  // ---------------- operand descriptors + per-lane DMA offsets of the 4 image kinds ----------------
  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldc_b = (unsigned)p.ld_c * 2u;
  __amdgpu_buffer_rsrc_t rs_r, rs_c;
  unsigned vb_rl[2], vb_rh[2], vb_cl[2], vb_ch[2];
  int ax_r[2], ax_c[2], ax_dummy[2];
  // RL / RH = tile rows [0,128) / [128,256); CL / CH = tile columns [0,128) / [128,256): contiguous, fully coalesced images.
  // Wave (wm, wn) owns rows {wm*64 + 0..63} of BOTH row images and columns {wn*32 + 0..31} of BOTH column images, i.e. four
  // 64x32 blocks of the 256x256 tile, so every image is consumed in exactly one phase by all eight waves.
  if (MODE == 0) {
#ifdef CSMOE_SAME_TILE   // diagnostic: every tile reads the operands of tile (expert 0, rows 0.., columns 0..): the loop with all fetches L2 hits
    rs_r = make_rsrc((const char*)p.R, (unsigned)rows * ldr_b);
#else
    rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);
#endif
    dma_setup<KC, 2>(vb_rl, ax_r, ldr_b, 0, 0, 7, 0, 0, wave, lane);
    dma_setup<KC, 2>(vb_rh, ax_dummy, ldr_b, 0, 0, 7, 0, 128, wave, lane);
#ifdef CSMOE_SAME_TILE
    const char* wb = (const char*)(p.c_ptrs_in ? p.c_ptrs_in[0] : p.single_B);
    const int tc0_rd = 0;
#else
    const char* wb = (const char*)(p.c_ptrs_in ? p.c_ptrs_in[e] : p.single_B);
    const int tc0_rd = tc0;
#endif
    if (COLK == KC) {
      int nrows = min(BN2, p.NC - tc0);
      rs_c = make_rsrc(wb + (int64_t)tc0_rd * ldc_b, (unsigned)nrows * ldc_b);
      dma_setup<KC, 2>(vb_cl, ax_c, ldc_b, 0, 0, 7, 0, 0, wave, lane);
      dma_setup<KC, 2>(vb_ch, ax_dummy, ldc_b, 0, 0, 7, 0, 128, wave, lane);
    } else {
      rs_c = make_rsrc(wb, (unsigned)p.Kd * ldc_b);
      dma_setup<KM, 2>(vb_cl, ax_c, ldc_b, tc0_rd, p.NC, 7, 0, 0, wave, lane);
      dma_setup<KM, 2>(vb_ch, ax_dummy, ldc_b, tc0_rd, p.NC, 7, 0, 128, wave, lane);
    }
  } else {
    rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)red_len * ldr_b);
    dma_setup<KM, 2>(vb_rl, ax_r, ldr_b, tr0, p.NR, 7, 0, 0, wave, lane);
    dma_setup<KM, 2>(vb_rh, ax_dummy, ldr_b, tr0, p.NR, 7, 0, 128, wave, lane);
    rs_c = make_rsrc((const char*)p.Cflat + (int64_t)row0 * ldc_b, (unsigned)red_len * ldc_b);
    dma_setup<KM, 2>(vb_cl, ax_c, ldc_b, tc0, p.NC, 7, 0, 0, wave, lane);
    dma_setup<KM, 2>(vb_ch, ax_dummy, ldc_b, tc0, p.NC, 7, 0, 128, wave, lane);
  }

  // ---------------- LDS read addressing ----------------
  const int g = lane >> 4, i16 = lane & 15;
  const int kc_lane = i16 * 128 + ((g ^ (i16 >> 1)) << 4);
  const int q = i16 >> 2, pp = i16 & 3;
  const int fk = q | ((g & 1) << 2);
  // this wave's 16-row / 16-column blocks inside an image: R images blocks wm*4 + 0..3, C images blocks wn*2 + 0..1
  const int r_blk0 = wm * 4, c_blk0 = wn * 2;
  int km_r[4], km_c[2];
#pragma unroll
  for (int b = 0; b < 4; ++b) km_r[b] = (8 * g + q) * 256 + (((r_blk0 + b) ^ fk) << 5) + pp * 8;
#pragma unroll
  for (int b = 0; b < 2; ++b) km_c[b] = (8 * g + q) * 256 + (((c_blk0 + b) ^ fk) << 5) + pp * 8;

  RT_STAMP(st_setup);
  f32x4 acc[4][8];   // [column block][row block]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (red_len + BK2 - 1) / BK2;

  // slot(kind, parity) = (parity * 4 + kind) * 16 KiB with kind RL=0, CL=1, CH=2, RH=3
#define SLOT(kind, tile) (smem + ((((tile) & 1) * 4 + (kind)) * TILE_B))
#define ISSUE_RL(tile) dma_tile<ROWK, 2>(rs_r, SLOT(0, tile), vb_rl, ax_r, (tile) * BK2, red_len, ldr_b, wave)
#define ISSUE_CL(tile) dma_tile<COLK, 2>(rs_c, SLOT(1, tile), vb_cl, ax_c, (tile) * BK2, red_len, ldc_b, wave)
#define ISSUE_CH(tile) dma_tile<COLK, 2>(rs_c, SLOT(2, tile), vb_ch, ax_c, (tile) * BK2, red_len, ldc_b, wave)
#define ISSUE_RH(tile) dma_tile<ROWK, 2>(rs_r, SLOT(3, tile), vb_rh, ax_r, (tile) * BK2, red_len, ldr_b, wave)
#define WAIT_DMA(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // barrier A closes a phase's read/issue section (reads retired, counted DMA wait taken), barrier B its MFMA section
#define PHASE_SYNC_IN()                                \
  __builtin_amdgcn_sched_barrier(0);                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_setprio(1)
  // epilogue barrier: LDS traffic only (__syncthreads() adds s_waitcnt vmcnt(0): a wait for the pass's own global stores)
#define EPI_SYNC()                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                        \
  asm volatile("" ::: "memory")
#define PHASE_SYNC_OUT()                               \
  __builtin_amdgcn_s_setprio(0);                       \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0)
  // ragged tiles: a wave whose 64-row / 32-column strips lie outside the tile skips those reads and MFMAs (it still issues
  // its share of the DMA and takes every barrier)
  const int rows_here = (MODE == 0 ? rows : min(BM2, p.NR - tr0)) - wm * 64;
  const int cols_here = min(BN2, p.NC - tc0) - wn * 32;

  // Tiles of at most 128 rows (the remainder tile of an expert: with ~1024 +- 32 rows per expert every other expert has one of
  // 1..60 rows) only ever use the row image RL: a K-tile is ONE phase (C_all x R_lo), three images to fetch (6 of the 8 pieces per
  // wave) and two barriers instead of four.  In the full-tile loops such a tile ran 70-83 % as long as a 256-row tile
  // (tools/tile_stamps.py: the loop is paced by the DMA issue and the barriers, not by the MFMAs it skips); CSMOE_THIN_LOOP=0
  // sends them through the full-tile loop again (A/B).
  if (MODE == 0 && rows <= 128 && p.thin_loop) {
    const bool rlo = rows_here > 0, clo = cols_here > 0, chi = cols_here > 128;
    ISSUE_CL(0); ISSUE_CH(0); ISSUE_RL(0); ISSUE_CL(1); ISSUE_CH(1); ISSUE_RL(1);
    WAIT_DMA(6);                                           // K-tile 0 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define THIN_FRAG_DECL bf16x8 fc[4][2], fr[4][2];
#define THIN_READ_CL(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[cb][ks] = (COLK == KC) ? frag_kc(i_cl, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_cl, km_c[cb], ks);
#define THIN_READ_CH(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[2 + cb][ks] = (COLK == KC) ? frag_kc(i_ch, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_ch, km_c[cb], ks);
#define THIN_READ_RL(rb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fr[rb][ks] = (ROWK == KC) ? frag_kc(i_rl, kc_lane, r_blk0 + rb, ks) : frag_km_raw(i_rl, km_r[rb], ks);
#define THIN_KSTEPS 2
#define THIN_MFMA(ks, cb, rb) acc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[cb][ks], fr[rb][ks], acc[cb][rb], 0, 0, 0);
#define THIN_ISSUE(s) ISSUE_CL(s + 2); ISSUE_CH(s + 2); ISSUE_RL(s + 2)
#define THIN_WAIT(s) WAIT_DMA(6)                           /* K-tile s+1 landed (s+2 stays in flight) */
#include "gemm_loop_thin.inc"
  } else if constexpr (SCHED == BAL) {
    // WIDE with the work of the two phases cut by ROW image instead of by column image, so that the LDS reads and the DMA issue
    // are spread over both phases (WIDE reads 20 fragments in phase A and 4 in phase B):
    //     phase A: read CL, CH, RL(s)  (16 fragments)   issue RH(s+1)            vmcnt(8)   MFMA C_all x R_lo
    //     phase B: read RH(s)          ( 8 fragments)   issue CL, CH, RL(s+2)    vmcnt(8)   MFMA C_all x R_hi
    // An image is waited for one phase before it is read (the counted wait sits in front of barrier A, the staggered half takes
    // the same wait one hardware barrier later), and a slot is re-filled in the phase after the one that read it.  Issue order
    // ... [CL,CH,RL](s+1) | RH(s+1) | [CL,CH,RL](s+2) | RH(s+2) ...: vmcnt(8) in A leaves {[CL,CH,RL](s+1), RH(s+1)} in flight
    // (RH(s) landed), vmcnt(8) in B leaves {RH(s+1), [CL,CH,RL](s+2)} ([CL,CH,RL](s+1) landed).
#ifdef CSMOE_DMA_ONLY     // diagnostic: the loop's DMA stream and barriers alone (no fragment reads, no MFMAs)
    const bool rlo = false, rhi = false, clo = cols_here > 0, chi = cols_here > 128;
#else
    const bool rlo = rows_here > 0, rhi = rows_here > 128, clo = cols_here > 0, chi = cols_here > 128;
#endif
    ISSUE_CL(0); ISSUE_CH(0); ISSUE_RL(0); ISSUE_RH(0); ISSUE_CL(1); ISSUE_CH(1); ISSUE_RL(1);
    WAIT_DMA(8);                                           // CL, CH, RL(0) landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();             // row half 1 starts half a phase late
    __builtin_amdgcn_sched_barrier(0);
#define BAL_FRAG_DECL bf16x8 fc[4][2], fr[4][2];
#define BAL_READ_CL(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[cb][ks] = (COLK == KC) ? frag_kc(i_cl, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_cl, km_c[cb], ks);
#define BAL_READ_CH(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[2 + cb][ks] = (COLK == KC) ? frag_kc(i_ch, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_ch, km_c[cb], ks);
#define BAL_READ_RL(rb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fr[rb][ks] = (ROWK == KC) ? frag_kc(i_rl, kc_lane, r_blk0 + rb, ks) : frag_km_raw(i_rl, km_r[rb], ks);
#define BAL_READ_RH(rb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fr[rb][ks] = (ROWK == KC) ? frag_kc(i_rh, kc_lane, r_blk0 + rb, ks) : frag_km_raw(i_rh, km_r[rb], ks);
#define BAL_PHASE_A_EXTRA
#define BAL_KSTEPS 2
#define BAL_MFMA(ks, cb, rb, arb) acc[cb][arb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[cb][ks], fr[rb][ks], acc[cb][arb], 0, 0, 0);
#define BAL_ISSUE_A(s) ISSUE_RH(s + 1)
#define BAL_ISSUE_B(s) ISSUE_CL(s + 2); ISSUE_CH(s + 2); ISSUE_RL(s + 2)
#define BAL_WAIT_A(s) WAIT_DMA(8)                          /* RH(s) landed */
#define BAL_WAIT_B(s) WAIT_DMA(8)                          /* CL, CH, RL(s+1) landed */
#include "gemm_loop_bal.inc"
    if (wm == 0) __builtin_amdgcn_s_barrier();             // row half 0 waits for the staggered half to finish
  } else if constexpr (SCHED == WIDE) {
    // TWO phases of 32 MFMA per K-tile (half the barriers of the 4-phase loop; all 8 row blocks stay in registers):
    //     phase A: read CL, RL, RH(s)   issue CL,CH(s+1)   vmcnt(8)   MFMA C_lo x R_all
    //     phase B: read CH(s)           issue RL,RH(s+2)   vmcnt(6)   MFMA C_hi x R_all
    const bool actA = rows_here > 0 && cols_here > 0, actAh = rows_here > 128 && cols_here > 0;
    const bool actB = rows_here > 0 && cols_here > 128, actBh = rows_here > 128 && cols_here > 128;
    ISSUE_RL(0); ISSUE_RH(0); ISSUE_CL(0); ISSUE_CH(0); ISSUE_RL(1); ISSUE_RH(1);
    WAIT_DMA(6);                                           // RL, RH, CL(0) landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();             // row half 1 starts half a phase late
    __builtin_amdgcn_sched_barrier(0);
#define WIDE_PRE_A(s)
#define WIDE_PRE_B(s)
#define WIDE_READ_CL(cb, ks) fc[cb][ks] = (COLK == KC) ? frag_kc(i_cl, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_cl, km_c[cb], ks);
#define WIDE_READ_CH(cb, ks) fc[cb][ks] = (COLK == KC) ? frag_kc(i_ch, kc_lane, c_blk0 + cb, ks) : frag_km_raw(i_ch, km_c[cb], ks);
#define WIDE_READ_RL(rb, ks) fr[rb][ks] = (ROWK == KC) ? frag_kc(i_rl, kc_lane, r_blk0 + rb, ks) : frag_km_raw(i_rl, km_r[rb], ks);
#define WIDE_READ_RH(rb, ks) fr[4 + rb][ks] = (ROWK == KC) ? frag_kc(i_rh, kc_lane, r_blk0 + rb, ks) : frag_km_raw(i_rh, km_r[rb], ks);
#define WIDE_ISSUE_A(s) ISSUE_CL(s + 1); ISSUE_CH(s + 1)
#define WIDE_ISSUE_B(s) ISSUE_RL(s + 2); ISSUE_RH(s + 2)
#define WIDE_WAIT_A(s) WAIT_DMA(8)                         /* CH(s) landed */
#define WIDE_WAIT_B(s) WAIT_DMA(6)                         /* RL, RH, CL(s+1) landed */
#include "gemm_loop_wide.inc"
    if (wm == 0) __builtin_amdgcn_s_barrier();             // row half 0 waits for the staggered half to finish
  }

  RT_STAMP(st_loop_end);
  // the zero-fill DMAs of the K-tiles past the end may still be writing LDS: drain before the staging tile reuses it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---------------- epilogue: two passes of 128 rows through an fp32 LDS tile ----------------
  // ---------------- epilogue ----------------
  if constexpr (MODE == 0) {
    const EpiArgs ea{p.C, p.C2, p.aux, p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias, p.ldc, p.epilogue, p.act, p.NC};
    rowspace_epilogue(ea, acc, smem, row0, rows, tc0, wm, wn, lane);
  } else {
    // weight-gradient form (kept for the MODE = 1 instantiation; the launched weight-gradient kernel is gemm_bf16_v2p.hip's):
    // two passes of 128 rows through an fp32 LDS tile
    float* stg = (float*)smem;
    const int ec = (threadIdx.x & 31) * 8, er = threadIdx.x >> 5;
    const int ncol = tc0 + ec;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const int m = wm * 64 + rb * 16 + i16;
          const int n = (cb >> 1) * 128 + wn * 32 + (cb & 1) * 16 + 4 * g;
          *(f32x4*)(stg + m * CT2_LD + n) = pass == 0 ? acc[cb][rb] : acc[cb][4 + rb];
        }
      EPI_SYNC();
      if (ncol < p.NC) {
      char* Ce = (char*)(p.out_ptrs ? p.out_ptrs[e] : p.single_C);
      const int rlim = min(128, p.NR - tr0 - pass * 128);
#pragma unroll 1
      for (int r = er; r < rlim; r += 16) {
        f32x4 lo = *(const f32x4*)(stg + r * CT2_LD + ec), hi = *(const f32x4*)(stg + r * CT2_LD + ec + 4);
        const int64_t o = (int64_t)(tr0 + pass * 128 + r) * p.ldc + ncol;
        if (p.out_f32) {
          f32x4* dst = (f32x4*)(Ce + o * 4);
          if (p.accumulate) { lo += dst[0]; hi += dst[1]; }
          dst[0] = lo; dst[1] = hi;
        } else {
          bf16x8* dst = (bf16x8*)(Ce + o * 2);
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (p.accumulate) {
            const bf16x8 old = *dst;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)old[j];
          }
          bf16x8 o8;
#pragma unroll
          for (int j = 0; j < 8; ++j) o8[j] = (bf16)v[j];
          *dst = o8;
        }
      }
      }
      EPI_SYNC();
    }
  }
#ifdef CSMOE_STAMPS
  if (threadIdx.x == 0 && g_stamp_buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the tile's own stores have left the wave
    unsigned long long* d = g_stamp_buf + (size_t)blockIdx.x * 8;
    d[0] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
    d[1] = st_entry; d[2] = st_setup; d[3] = st_loop_end; d[4] = __builtin_amdgcn_s_memrealtime();
    d[5] = ((unsigned long long)(unsigned)rows << 32) | (unsigned)nk;
  }
#endif
}

// CSMOE_GEMM_SCHED=2|3 forces WIDE / BAL for every row-space launch (A/B runs); unset = the measured best per layout: BAL for NT
// (both operands K-contiguous: +4 % over WIDE), WIDE for NN (K-major weights: BAL is 1-4 % slower there).  Round 3 removed the loops
// that had lost every comparison and were no longer instantiated by default -- the four-phase SHALLOW / DEEP loops (rounds 1-2:
// 6-16 % slower than WIDE), PP (one barrier per phase: 2-4 % slower, gpurun_out r2r), the class-sorted tile order
// (CSMOE_TILE_CLASSES: 0-3 % slower) -- they are in the history (commit 1d8bf29), their measurements in DESIGN.md section 3.
int sched_pref(int b_layout) {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("CSMOE_GEMM_SCHED");
    v = e ? atoi(e) : -1;
  }
  if (v == WIDE || v == BAL) return v;
  return b_layout == CSMOE_B_NK ? BAL : WIDE;
}

template <typename K>
int set_lds2(K kern) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  return CSMOE_OK;
}

}  // namespace

static int rowspace_launch(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                           const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                           const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                           hipStream_t st, int row_part) {
  FastArgs p{};
  p.row_part = row_part;
  p.single_M = M; p.single_B = single_B; p.single_bias = single_bias;
  p.R = A; p.ld_r = lda; p.c_ptrs_in = b_ptrs; p.ld_c = ldb; p.bias_ptrs = bias_ptrs; p.offsets = offsets; p.E = E;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  {
    static const int thin = [] { const char* e = getenv("CSMOE_THIN_LOOP"); return e ? atoi(e) : 1; }();
    p.thin_loop = thin;
  }
  int nct = (N + BN2 - 1) / BN2;
  int64_t grid = (int64_t)nct * ((M + BM2 - 1) / BM2 + E);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_gemm: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  int rc;
  const int sp = sched_pref(b_layout);
#ifdef CSMOE_STAMPS
  struct StampDump {
    unsigned long long* buf; int64_t n; hipStream_t st;
    ~StampDump() {
      const char* path = getenv("CSMOE_STAMP_FILE");
      if (!buf || !path) return;
      (void)hipStreamSynchronize(st);
      std::vector<unsigned long long> h((size_t)n * 8);
      (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
      if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      (void)hipFree(buf);
    }
  } stamp_dump{nullptr, grid, st};
  if (getenv("CSMOE_STAMP_FILE")) {
    (void)hipMalloc(&stamp_dump.buf, (size_t)grid * 64);
    (void)hipMemsetAsync(stamp_dump.buf, 0, (size_t)grid * 64, st);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stamp_buf), &stamp_dump.buf, sizeof(void*), 0, hipMemcpyHostToDevice, st);
  }
#endif
#define LAUNCH_SCHED(S)                                                                                              \
  do {                                                                                                                \
    if (b_layout == CSMOE_B_NK) {                                                                                     \
      if ((rc = set_lds2(gg8_kernel<KC, KC, 0, S>))) return rc;                                                       \
      hipLaunchKernelGGL((gg8_kernel<KC, KC, 0, S>), dim3((unsigned)grid), dim3(512), LDS2_BYTES, st, p);             \
    } else {                                                                                                          \
      if ((rc = set_lds2(gg8_kernel<KC, KM, 0, S>))) return rc;                                                       \
      hipLaunchKernelGGL((gg8_kernel<KC, KM, 0, S>), dim3((unsigned)grid), dim3(512), LDS2_BYTES, st, p);             \
    }                                                                                                                 \
  } while (0)
  if (sp == BAL) LAUNCH_SCHED(BAL); else LAUNCH_SCHED(WIDE);
  CSMOE_CHECK_LAUNCH("grouped_gemm(bf16 v2)");
  return CSMOE_OK;
}

int gg8_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                 const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                 const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                 hipStream_t st) {
  return rowspace_launch(A, lda, b_ptrs, b_layout, ldb, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act, single_B,
                         single_bias, st, 0);
}

// every row tile of every expert EXCEPT its first (the fp32-master GEMM's second launch, gemm_bf16_v2c.hip)
int gg8_rowspace_rest(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                      const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                      const void* aux, int64_t ldc, int epilogue, int act, hipStream_t st) {
  return rowspace_launch(A, lda, b_ptrs, b_layout, ldb, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act, nullptr,
                         nullptr, st, 2);
}
