// Grouped expert GEMM v4 for gfx950, row-space launches: ONE WAVE PER SIMD.
// (moe_pretrain_model/layers/cvmm.py:61-168 `cvmm_kernel`; the per-expert nn.Linear loop of moe_model/model/moe/moe.py:196-204)
//
// 256x256 output tile, K-tile 64, 256 threads = 4 waves as 2 x 2, each wave a contiguous 128x128 quadrant = 8 x 8
// v_mfma_f32_16x16x32_bf16 accumulators held IN PLACE in 256 accumulator registers, the fragments of both K-steps in 128 vector
// registers, operands brought in by LDS-DMA into the swizzled 16 KiB images of gemm_tiles.h (two slots of four images).
//
// Why (round 3).  The 8-wave kernel (gemm_bf16_v2.hip) runs its matrix pipe 57-75 % busy at 1.5-1.8 GHz; the vendor's kernels for the
// same 256x256x64 tile are 4-wave / 512-register kernels: 71 % at 1.78 GHz (generated), 92 % at 1.51 GHz (hand-written;
// tools/vendor_probe.sh, disassembly of hipBLASLt's Custom_Cijk_Alik_Bljk_..._MT256x256x64: 16 `buffer_load_dwordx4 ... lds`,
// 32 ds_read_b128 and 128 MFMAs per wave and K-tile, all four fragment sets in registers, accumulators in place in a[0:255]).  The
// chip is power-limited under these loops (busy x clock ~ constant for a given loop), so what counts is the work spent per MFMA:
//   * a wave that owns 128x128 reads 16 + 16 KiB of fragments per K-tile, four of them 128 KiB; eight waves of 128x64 read 192 KiB;
//   * one wave per SIMD: nothing to arbitrate, no stagger, TWO barriers per K-tile (four in v2), no s_setprio.
// Measured here (tools/gemm_bench.py, same box, ms per launch v2 -> v4, random routing): GEMM1 5.52 -> 5.36, GEMM2 4.86 -> 4.65,
// dXs 4.98 -> 4.71, dH 5.41 -> 5.47 (its epilogue is VALU work, which a lone wave issues at half the rate of two); on slower-clocked
// boxes the gap grows (5.23 -> 4.81).
//
// Three things hipcc (ROCm 7.2) does not do for such a kernel by itself, and what the file does about each:
//   1. Arrays stay in registers only when every index is a compile-time constant, and a `#pragma unroll` over the 128 MFMA slots with
//      tests on the slot number inside was NOT fully unrolled (G[] and the accumulators went through scratch): the slot table is a
//      template over the slot number, expanded by a fold expression.
//   2. With the MFMA builtin the 64 accumulators were allocated OUT of place (each MFMA wrote a fresh AGPR quad, ~500 v_accvgpr moves
//      beside 256 MFMAs): the MFMA is inline asm with a "+a" operand.  `volatile` also pins the issue order of the memory operations
//      around it, so the slot table IS the instruction order.
//   3. The hazard recogniser does not see inside inline asm: `settle_last_group` supplies the wait states before the epilogue's
//      v_accvgpr_reads, tied to the accumulators as operands (a `memory` clobber does not order register reads).
//
// Slot table of one K-tile s for wave W (slot I = MFMA number; K-step 0 = slots 0..63 on fr0 / fc0, K-step 1 = 64..127 on fr1 / fc1;
// CUR = s & 1 the LDS slot of K-tile s, NXT the other one, which holds K-tile s+1):
//   I = 0, 2, .. 30     read the 16 fragments of K-step 1 from CUR (odd waves on the odd slots)
//   I = 34              lgkmcnt(0) + barrier: every wave holds all of K-tile s in registers, CUR is free
//   I = 36 + 5 k + W    DMA piece k (< 16) of K-tile s+2 -> CUR.  The CU's 64 pieces are spread evenly over the K-tile and the four
//                       waves take turns: in two bursts of eight between three barriers, as the vendor kernel has them, the address
//                       path was offered twice what it moves and blocked the issuing waves (+1.1 ms on a 3.0 ms MFMA stream;
//                       `make ablate` + tools/v4_ablate.sh: MFMAs only 3.0 ms, + reads 3.5, + DMA 4.1, all 4.6 before the spreading)
//   I = 62              vmcnt(pieces issued so far) + barrier: K-tile s+1, issued a K-tile ago, is in NXT for every wave
//   I = 64, 68, ..      read the 16 fragments of K-step 0 of K-tile s+1 from NXT (wave W on slot 64 + 4 j + W)
// (tools/v4_tune.sh: a dozen tables with the reads and the DMA packed or spread differently, 4.66 .. 5.06 ms on one box; this one won.)
// A register-staged form of the same loop (buffer_load_dwordx4 -> ds_write_b128, one barrier per K-tile) was built first and measured
// level with the DMA form before its spreading; it is in the history (commit "One-wave-per-SIMD row-space kernel"), not here.
#include "gemm_epilogue.h"
#include <cstdlib>
#include <utility>

using namespace ggt;

namespace {

constexpr int BM4 = 256, BN4 = 256, BK4 = 64;
constexpr int LDS4_BYTES = EPI_LDS_BYTES;     // 135,168 B: the epilogue's staging tile (>= the 8 operand images = 131,072 B)

// image kinds in piece order: RL (rows 0..127), RH (rows 128..255), CL (columns 0..127), CH (columns 128..255); LDS slot order
// inside a K-tile slot follows gemm_bf16_v2.hip: RL = 0, CL = 1, CH = 2, RH = 3
__device__ __forceinline__ constexpr int kind_slot(int kind) { return kind == 0 ? 0 : kind == 1 ? 3 : kind == 2 ? 1 : 2; }

// Everything one wave carries through the K-loop besides its accumulators (every array index below is a template constant)
struct LoopState {
  bf16x8 fr0[8], fr1[8], fc0[8], fc1[8];   // row / column fragments of K-step 0 / 1
  __amdgpu_buffer_rsrc_t rs_r, rs_c;
  unsigned vb[16];                          // per-lane source offsets of the 16 pieces (image swizzle included): RL 0..3, RH 4..7, CL, CH
  unsigned kstep_r, kstep_c;                // bytes one K-tile advances the row / column operand
  char* smem;
  int wave4k;                               // wave * 4 KiB: the wave-uniform part of a piece's LDS destination
  int kc_lane, r_off, c_off;
  int km_c[8];
};

// acc += a x b in place (see the header, item 2)
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// (header, item 3; the first build read acc[7][*] of ONE epilogue variant too early: wrong values in columns 114, 118, .. 254 of
// every tile of the K-major kernel with the plain epilogue, right ones with the bias epilogue)
__device__ __forceinline__ void settle_last_group(f32x4 (&acc)[8][8]) {
  asm volatile("s_nop 15\n\ts_nop 15"
               : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]), "+a"(acc[7][6]),
                 "+a"(acc[7][7]));
}

template <int SL, int KS, int B>
__device__ __forceinline__ bf16x8 read_r(const LoopState& st) {
  return frag_kc(st.smem + SL * (4 * TILE_B) + st.r_off, st.kc_lane, B, KS);
}
// Column fragment B of K-step KS from slot SL: one ds_read_b128 (K-contiguous image) or two transposed 64-bit reads (K-major image).
// THE K-MAJOR FORM IS CORRECT BUT NOT LAUNCHED (gg4_rowspace_ok takes K-contiguous weights only): beside an LDS-DMA in flight hipcc
// puts vmcnt(0) in front of every ds_read_b64_tr_b16 BUILTIN (14 ms instead of 4.5 per launch), and both ways around it failed here:
//   * the transposed reads as inline asm (as the 8-wave kernels have them) are invisible to the waitcnt pass, and in this kernel the
//     allocator moved their results with v_mov copies placed straight behind the asm, before the data had landed, whenever it had
//     not given the two halves adjacent registers (full-epilogue build: every tile row of waves 2 and 3 wrong; plain-epilogue
//     tuning builds: none) -- 4.54 ms against 5.08 for the 8-wave kernel while it happened to work;
//   * the LDS-DMA as inline asm instead (so that the builtin reads stay tracked) gave wrong 16x16 blocks here and there in BOTH
//     layouts (hazards between SALU / VALU results and an asm VMEM the recogniser cannot see).
// The K-major row-space launches (dH, dXs) stay on the 8-wave kernel.
template <int COLK, int SL, int KS, int B>
__device__ __forceinline__ bf16x8 read_c(const LoopState& st) {
  if constexpr (COLK == KC) return frag_kc(st.smem + SL * (4 * TILE_B) + st.c_off, st.kc_lane, B, KS);
  else return frag_km(st.smem + SL * (4 * TILE_B) + st.c_off, st.km_c[B], KS);
}

#ifndef CSMOE_V4_ABL
#define CSMOE_V4_ABL 0      // diagnostic twins (`make ablate`): bit 0 no fragment reads in the loop, bit 1 no LDS-DMA, bit 2 no landing wait
#endif
// One 1 KiB piece by LDS-DMA: the wave-uniform LDS destination goes to M0, the per-lane source offset carries the image swizzle, the
// SGPR offset the K-tile.
template <int I, int SL>
__device__ __forceinline__ void dma_piece4(LoopState& st, int tile) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(I < 8 ? st.rs_r : st.rs_c,
                                           (lds_void*)(st.smem + SL * (4 * TILE_B) + kind_slot(I >> 2) * TILE_B + (I & 3) * 1024 + st.wave4k),
                                           16, st.vb[I], (unsigned)tile * (I < 8 ? st.kstep_r : st.kstep_c), 0, 0);
}

// slot table parameters (tools/v4_tune.sh builds twins with other values)
#ifndef CSMOE_V4_WHOLE
#define CSMOE_V4_WHOLE 8      // tile order: experts of at most this many row tiles are one band (common.h grouped_find_tile); 0 = bands of 4 always
#endif
#ifndef CSMOE_V4_RSTEP
#define CSMOE_V4_RSTEP 2      // early reads: one every RSTEP slots from slot 0
#define CSMOE_V4_BAR 34       // slot of the "K-tile in registers" barrier
#define CSMOE_V4_D0 36        // first DMA slot
#define CSMOE_V4_DSTEP 5      // DMA piece k at D0 + DSTEP k + W
#define CSMOE_V4_LAND 62      // landing barrier
#define CSMOE_V4_R0 64        // late reads from here, one every R0STEP slots
#define CSMOE_V4_R0STEP 4
#endif

template <int COLK, int RBN, int CUR, int W, int I>
__device__ __forceinline__ void slot(f32x4 (&acc)[8][8], LoopState& st, int t2) {
  constexpr int NXT = CUR ^ 1;
  constexpr int gi = I >> 3, ks = gi >> 3, cb = gi & 7, rb = I & 7;
  constexpr bool RD = !(CSMOE_V4_ABL & 1), DM = !(CSMOE_V4_ABL & 2);
  constexpr int RS = CSMOE_V4_RSTEP, BAR = CSMOE_V4_BAR, DSTEP = CSMOE_V4_DSTEP, LAND = CSMOE_V4_LAND, R0 = CSMOE_V4_R0, R0S = CSMOE_V4_R0STEP;
  constexpr int P = RS > 1 ? (W % RS) : 0;     // read slots: the waves take turns
  constexpr int P0 = R0S > 1 ? (W % R0S) : 0;
  constexpr int D0 = CSMOE_V4_D0 + W;
  static_assert(16 * RS <= BAR && BAR < CSMOE_V4_D0 && R0 + 16 * R0S <= 128 && LAND < R0 && CSMOE_V4_D0 + 3 + 15 * DSTEP < 128, "slot table");
  if constexpr (rb < RBN) {
    if constexpr (ks == 0) mfma_acc(acc[cb][rb], st.fc0[cb], st.fr0[rb]);
    else                   mfma_acc(acc[cb][rb], st.fc1[cb], st.fr1[rb]);
  }
  if constexpr (RD && I < 16 * RS && (I % RS) == P) {          // K-step 1's 16 fragments, early: slot CUR is free after them
    constexpr int k = I / RS;
    if constexpr (k < 8) { if constexpr (k < RBN) st.fr1[k] = read_r<CUR, 1, k>(st); }
    else st.fc1[k - 8] = read_c<COLK, CUR, 1, (k - 8)>(st);
  }
  if constexpr (I == BAR) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (I == LAND) {                                   // K-tile s+1 (issued a K-tile ago) landed, for every wave
    constexpr int issued = LAND <= D0 ? 0 : ((LAND - D0 + DSTEP - 1) / DSTEP > 16 ? 16 : (LAND - D0 + DSTEP - 1) / DSTEP);
    if constexpr (DM && !(CSMOE_V4_ABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(issued) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (DM && I >= D0 && I <= D0 + 15 * DSTEP && ((I - D0) % DSTEP) == 0) dma_piece4<((I - D0) / DSTEP), CUR>(st, t2);
  if constexpr (RD && I >= R0 && I < R0 + 16 * R0S && ((I - R0) % R0S) == P0) {
    constexpr int k = (I - R0) / R0S;
    if constexpr (k < 8) { if constexpr (k < RBN) st.fr0[k] = read_r<NXT, 0, k>(st); }
    else st.fc0[k - 8] = read_c<COLK, NXT, 0, (k - 8)>(st);
  }
}
template <int COLK, int RBN, int CUR, int W, int... Is>
__device__ __forceinline__ void k_tile(f32x4 (&acc)[8][8], LoopState& st, int t2, std::integer_sequence<int, Is...>) {
  (slot<COLK, RBN, CUR, W, Is>(acc, st, t2), ...);
}
template <int SL, int... Is>
__device__ __forceinline__ void dma_all(LoopState& st, int tile, std::integer_sequence<int, Is...>) { (dma_piece4<Is, SL>(st, tile), ...); }
template <int COLK, int RBN, int... Is>
__device__ __forceinline__ void read_first(LoopState& st, std::integer_sequence<int, Is...>) {
  ((Is < RBN ? (void)(st.fr0[Is] = read_r<0, 0, Is>(st)) : (void)0), ...);
  ((st.fc0[Is] = read_c<COLK, 0, 0, Is>(st)), ...);
}

template <int COLK, int RBN, int W>
__device__ __forceinline__ void k_loop_w(f32x4 (&acc)[8][8], LoopState& st, int nk) {
  using S128 = std::make_integer_sequence<int, 128>;
  for (int s = 0; s < nk; s += 2) {                        // nk is even (gg4_rowspace_ok): the body is two K-tiles, one per slot parity;
    k_tile<COLK, RBN, 0, W>(acc, st, min(s + 2, nk - 1), S128{});   // K-tiles past the end re-load the last one into a slot nobody reads
    k_tile<COLK, RBN, 1, W>(acc, st, min(s + 3, nk - 1), S128{});
  }
  settle_last_group(acc);      // in every copy of the loop: the moves that reconcile the copies' register assignments come behind it
}

// RBN = live 16-row blocks of the wave's quadrant (8, 4 or 0: MFMAs and row-fragment reads of the others are not emitted)
template <int COLK, int RBN>
__device__ __forceinline__ void k_loop(f32x4 (&acc)[8][8], LoopState& st, int nk, int wave) {
  using S16 = std::make_integer_sequence<int, 16>;
  dma_all<0>(st, 0, S16{});
  dma_all<1>(st, 1, S16{});
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // K-tile 0 landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_first<COLK, RBN>(st, std::make_integer_sequence<int, 8>{});
  switch (wave) {                                          // wave-uniform: four copies of the loop, one slot shift each
    case 0: k_loop_w<COLK, RBN, 0>(acc, st, nk); break;
    case 1: k_loop_w<COLK, RBN, 1>(acc, st, nk); break;
    case 2: k_loop_w<COLK, RBN, 2>(acc, st, nk); break;
    default: k_loop_w<COLK, RBN, 3>(acc, st, nk); break;
  }
}

template <int COLK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gg4_kernel(FastArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---------------- tile lookup: (expert, column tile, row tile) order, one contiguous chunk of it per XCD ----------------
  const int nct = (p.NC + BN4 - 1) / BN4;
  TilePos tp;
  {
    const int total = grouped_total_tiles(p.offsets, p.E, p.single_M, BM4, nct, lane);
    if ((int)blockIdx.x >= total) return;
    const int v = xcd_remap(blockIdx.x, total);
    if (!grouped_find_tile<CSMOE_V4_WHOLE>(p.offsets, p.E, p.single_M, BM4, nct, v, lane, tp)) return;
  }
  const int e = __builtin_amdgcn_readfirstlane(tp.e);
  const int row0 = __builtin_amdgcn_readfirstlane(tp.o0 + tp.mt * BM4);
  const int rows = __builtin_amdgcn_readfirstlane(min(BM4, tp.o1 - row0));
  const int tc0 = __builtin_amdgcn_readfirstlane(tp.nt * BN4);

  // ---------------- operand descriptors + per-lane source offsets of the 16 pieces ----------------
  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldc_b = (unsigned)p.ld_c * 2u;
  LoopState st;
  st.rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);     // rows past the expert's range read zeros
  const char* wb = (const char*)(p.c_ptrs_in ? p.c_ptrs_in[e] : p.single_B);
  {
    unsigned t[4];
    int ax[4];
    dma_setup<KC, 4>(t, ax, ldr_b, 0, 0, 7, 0, 0, wave, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) st.vb[j] = t[j];
    dma_setup<KC, 4>(t, ax, ldr_b, 0, 0, 7, 0, 128, wave, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) st.vb[4 + j] = t[j];
    if (COLK == KC) {
      const int nrows = min(BN4, p.NC - tc0);
      st.rs_c = make_rsrc(wb + (int64_t)tc0 * ldc_b, (unsigned)nrows * ldc_b);
      dma_setup<KC, 4>(t, ax, ldc_b, 0, 0, 7, 0, 0, wave, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) st.vb[8 + j] = t[j];
      dma_setup<KC, 4>(t, ax, ldc_b, 0, 0, 7, 0, 128, wave, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) st.vb[12 + j] = t[j];
    } else {
      st.rs_c = make_rsrc(wb, (unsigned)p.Kd * ldc_b);
      dma_setup<KM, 4>(t, ax, ldc_b, tc0, p.NC, 7, 0, 0, wave, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) st.vb[8 + j] = t[j];
      dma_setup<KM, 4>(t, ax, ldc_b, tc0, p.NC, 7, 0, 128, wave, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) st.vb[12 + j] = t[j];
    }
  }
  st.kstep_r = BK4 * 2u;                                  // bytes one K-tile advances a K-contiguous row (the SGPR offset of the DMA:
  st.kstep_c = COLK == KC ? BK4 * 2u : BK4 * ldc_b;       // no per-lane K bound is needed, K is a whole number of K-tiles)
  {
    const int g = lane >> 4, i16 = lane & 15;
    st.kc_lane = i16 * 128 + ((g ^ (i16 >> 1)) << 4);
    const int q = i16 >> 2, pp = i16 & 3;
    const int fk = q | ((g & 1) << 2);
#pragma unroll
    for (int b = 0; b < 8; ++b) st.km_c[b] = (8 * g + q) * 256 + ((b ^ fk) << 5) + pp * 8;
    st.r_off = kind_slot(wm ? 1 : 0) * TILE_B;
    st.c_off = kind_slot(wn ? 3 : 2) * TILE_B;
    st.smem = smem;
    st.wave4k = wave * 4096;
  }

  f32x4 acc[8][8];     // [column block][row block]: rows wm*128 + rb*16 + lane%16, columns wn*128 + cb*16 + 4*(lane/16) + 0..3 (Lay4)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kd / BK4;
  // a wave whose rows lie past the expert's range issues its DMA pieces and takes every barrier but no MFMA; the remainder tile of an
  // expert (1..64 rows) keeps half of wave-row 0's
  const int rows_here = rows - wm * 128;
  if (rows_here > 64) k_loop<COLK, 8>(acc, st, nk, wave);
  else if (rows_here > 0) k_loop<COLK, 4>(acc, st, nk, wave);
  else k_loop<COLK, 0>(acc, st, nk, wave);

  // the DMA pieces of the K-tiles past the end have landed and every wave's last fragment reads are done before the staging tile
  // overlays the images
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  const EpiArgs ea{p.C, p.C2, p.aux, p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias, p.ldc, p.epilogue, p.act, p.NC};
#ifdef CSMOE_V4_DEV      // tools/v4_tune.sh: the plain epilogue only (seconds instead of minutes to compile)
  epi_run<EC_PLAIN, 0, Lay4>(ea, acc, smem, row0, rows, tc0, wm, wn, lane);
#else
  rowspace_epilogue<Lay4>(ea, acc, smem, row0, rows, tc0, wm, wn, lane);
#endif
}

template <typename K>
int set_lds4(K kern) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_BYTES);
    if (e != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  return CSMOE_OK;
}

}  // namespace

// shapes the one-wave-per-SIMD kernel takes beyond gg_fast_rowspace_ok: whole PAIRS of K-tiles (the loop body is two K-tiles, one per
// LDS slot parity, and its DMA carries no per-lane K bound)
bool gg4_rowspace_ok(int Kd, int b_layout) { return Kd % (2 * BK4) == 0 && b_layout == CSMOE_B_NK; }

int gg4_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                 const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                 const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                 hipStream_t st) {
  FastArgs p{};
  p.single_M = M; p.single_B = single_B; p.single_bias = single_bias;
  p.R = A; p.ld_r = lda; p.c_ptrs_in = b_ptrs; p.ld_c = ldb; p.bias_ptrs = bias_ptrs; p.offsets = offsets; p.E = E;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  const int nct = (N + BN4 - 1) / BN4;
  const int64_t grid = (int64_t)nct * ((M + BM4 - 1) / BM4 + E);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_gemm: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  int rc;
  if (b_layout == CSMOE_B_NK) {
    if ((rc = set_lds4(gg4_kernel<KC>))) return rc;
    hipLaunchKernelGGL((gg4_kernel<KC>), dim3((unsigned)grid), dim3(256), LDS4_BYTES, st, p);
  } else {
    if ((rc = set_lds4(gg4_kernel<KM>))) return rc;
    hipLaunchKernelGGL((gg4_kernel<KM>), dim3((unsigned)grid), dim3(256), LDS4_BYTES, st, p);
  }
  CSMOE_CHECK_LAUNCH("grouped_gemm(bf16 v4)");
  return CSMOE_OK;
}
