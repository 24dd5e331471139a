// Grouped expert GEMM v4 for gfx950, row-space launches: ONE WAVE PER SIMD.
//
// 256x256 output tile, K-tile 64, 256 threads = 4 waves as 2 x 2, each wave a contiguous 128x128 quadrant = 8 x 8
// v_mfma_f32_16x16x32_bf16 accumulators (256 of the wave's 512 registers), both K-steps' fragments in registers (128), one K-tile
// of global loads in flight in registers (64).
//
// Why (round 3; gpurun_out/r3b_vendor/vendor_kernels.txt): the 8-wave kernel (gemm_bf16_v2.hip) runs its matrix pipe 57-69 % busy at
// 1.61-1.84 GHz; the vendor's 256x256x64 kernels for the same shapes are 4-wave / 512-register kernels that hold 71 % at 1.78 GHz
// (generated) and 92 % at 1.51 GHz (hand-written).  Same tile, so what differs is the energy and the issue slots spent per MFMA:
//   * LDS read traffic: a wave that owns 128x128 reads 16 + 16 KiB of fragments per K-tile, four of them 128 KiB; eight waves of
//     128x64 read 192 KiB for the same MFMAs;
//   * no second wave on the SIMD: nothing to arbitrate, no stagger, ONE barrier per K-tile (four in v2);
//   * operands come in by plain buffer_load_dwordx4 into registers and go to LDS by ds_write_b128 (an LDS-DMA issue blocks its
//     wave for 60-185 cycles -- MI355X_MICROARCH.md, cycle constants -- which a lone wave cannot hide behind a partner).
// The LDS images are the ones of gemm_tiles.h (swizzled 16 KiB KC / KM images, two slots of four): the store of piece j goes to
// the linear address the LDS-DMA would have written, the per-lane SOURCE offset carries the swizzle.
//
// Per K-tile s and wave, 128 MFMA slots (K-step 0: slots 0..63 on fragment set F0, K-step 1: 64..127 on F1), memory operations
// placed between them:
// (see the table in front of the loop in k_loop)
#include "gemm_epilogue.h"
#include <cstdlib>
#include <utility>

using namespace ggt;

namespace {

constexpr int BM4 = 256, BN4 = 256, BK4 = 64;
constexpr int LDS4_BYTES = EPI_LDS_BYTES;     // 135,168 B: the epilogue's staging tile (>= the 8 operand images = 131,072 B)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// image kinds in load order: RL (rows 0..127), RH (rows 128..255), CL (columns 0..127), CH (columns 128..255); LDS slot order
// inside a K-tile slot follows gemm_bf16_v2.hip: RL = 0, CL = 1, CH = 2, RH = 3
__device__ __forceinline__ constexpr int kind_slot(int kind) { return kind == 0 ? 0 : kind == 1 ? 3 : kind == 2 ? 1 : 2; }

// Everything one wave carries through the K-loop besides its accumulators.  All array indices below are template constants:
// hipcc keeps an array in registers only when every access has a compile-time index (a `#pragma unroll` over the 128 slots with
// run-time tests inside was NOT fully unrolled by ROCm 7.2 and sent G[] and the accumulators through scratch).
struct LoopState {
  u32x4_t G[16];                      // one K-tile of global loads in flight: pieces RL 0..3, RH 4..7, CL 8..11, CH 12..15
  bf16x8 fr0[8], fr1[8], fcq[4];      // row fragments of both K-steps; column fragments in a ring of four MFMA groups
  bf16x8 fc0[8], fc1[8];              // LDS-DMA form: the column fragments of both K-steps (there is no G[] then)
  __amdgpu_buffer_rsrc_t rs_r, rs_c;
  unsigned vb[16];                    // per-lane source offsets of the 16 pieces (swizzle included)
  unsigned kstep_r, kstep_c;          // bytes one K-tile advances the row / column operand
  char* smem;
  char* st_base;                      // smem + wave * 4 KiB + lane * 16: where this lane's 16 bytes of piece 0 of image slot 0 go
  int wave4k;                         // wave * 4 KiB (LDS-DMA form: the wave-uniform part of a piece's destination)
  int kc_lane, r_off, c_off;
  int km_c[8];
};

// acc += a x b, IN PLACE in accumulator registers.  As inline asm: with the builtin, hipcc (ROCm 7.2) allocates the 64 accumulators
// of a 512-register kernel out of place -- every MFMA wrote a fresh AGPR quad and ~2 v_accvgpr_read / _write / _mov per MFMA
// shuffled them back (504 moves beside 256 MFMAs in the first build of this loop).  `volatile` also pins the order of the memory
// operations around the MFMAs (the scheduler does not move loads / stores across an asm with side effects): the slot table below is
// the issue order.  The hazard recogniser does not see inside: the loop never reads an accumulator, and k_loop ends with the wait
// states a VALU read of the last MFMA's result needs.
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// The wait states a VALU read (v_accvgpr_read in the epilogue) of an MFMA result needs are inserted by hipcc's hazard recogniser
// for MFMAs it knows about -- not for inline asm.  The last group's eight accumulators are tied to the s_nops as operands: a
// `memory` clobber alone does not order register reads, and the first build read acc[7][*] of one epilogue variant too early
// (wrong values in columns 114, 118, ..., 254 of every tile; tools/v4_debug.py).
__device__ __forceinline__ void settle_last_group(f32x4 (&acc)[8][8]) {
  asm volatile("s_nop 15\n\ts_nop 15"
               : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]), "+a"(acc[7][6]),
                 "+a"(acc[7][7]));
}

template <int I>
__device__ __forceinline__ void load_piece(LoopState& st, int tile) {
  st.G[I] = __builtin_amdgcn_raw_buffer_load_b128(I < 8 ? st.rs_r : st.rs_c, st.vb[I], (unsigned)tile * (I < 8 ? st.kstep_r : st.kstep_c), 0);
}
template <int I, int SL>
__device__ __forceinline__ void store_piece(LoopState& st) {
  *(u32x4_t*)(st.st_base + SL * (4 * TILE_B) + kind_slot(I >> 2) * TILE_B + (I & 3) * 1024) = st.G[I];
}
template <int SL, int KS, int B>
__device__ __forceinline__ bf16x8 read_r(const LoopState& st) {
  return frag_kc(st.smem + SL * (4 * TILE_B) + st.r_off, st.kc_lane, B, KS);
}
template <int COLK, int SL, int KS, int B>
__device__ __forceinline__ bf16x8 read_c(const LoopState& st) {
  if constexpr (COLK == KC) return frag_kc(st.smem + SL * (4 * TILE_B) + st.c_off, st.kc_lane, B, KS);
  else return frag_km(st.smem + SL * (4 * TILE_B) + st.c_off, st.km_c[B], KS);
}

// One K-tile = 16 groups of 8 MFMAs (group gi: K-step gi >> 3, column block gi & 7, the 8 row blocks), slot I = 8 gi + rb:
//   I % 8 == 0      read the column fragment of group gi + 2 (groups 16, 17 = groups 0, 1 of the NEXT K-tile, slot NXT)
//   I = 4 + 8 k     read row fragment k of K-step 1 (k < 8)              I = 96 + 4 k   row fragment k of the next K-tile's step 0
//   I = 2 + 5 k     store piece k of K-tile s+1 into slot NXT, re-issue its load for K-tile s+2 (k < 16: I <= 77)
//   I = 90          lgkmcnt(0) + barrier: slot NXT complete, slot CUR's last reader (row fragment 7 of K-step 1, I = 60) retired
template <int COLK, int RBN, int CUR, int I>
__device__ __forceinline__ void slot(f32x4 (&acc)[8][8], LoopState& st, int t2) {
  constexpr int NXT = CUR ^ 1;
  constexpr int gi = I >> 3, ks = gi >> 3, cb = gi & 7, rb = I & 7;
  if constexpr ((I & 7) == 0) {
    constexpr int gn = gi + 2;
    if constexpr (gn < 16) st.fcq[gn & 3] = read_c<COLK, CUR, (gn >> 3), (gn & 7)>(st);
    else st.fcq[gn & 3] = read_c<COLK, NXT, 0, (gn & 7)>(st);
  }
  if constexpr (rb < RBN) {
    if constexpr (ks == 0) mfma_acc(acc[cb][rb], st.fcq[gi & 3], st.fr0[rb]);
    else                   mfma_acc(acc[cb][rb], st.fcq[gi & 3], st.fr1[rb]);
  }
  if constexpr (I < 64 && (I & 7) == 4 && (I >> 3) < RBN) st.fr1[I >> 3] = read_r<CUR, 1, (I >> 3)>(st);
  if constexpr (I >= 96 && (I & 3) == 0 && ((I - 96) >> 2) < RBN) st.fr0[(I - 96) >> 2] = read_r<NXT, 0, ((I - 96) >> 2)>(st);
  if constexpr (I >= 2 && I <= 77 && (I - 2) % 5 == 0) {
    constexpr int k = (I - 2) / 5;
    store_piece<k, NXT>(st);
    load_piece<k>(st, t2);
  }
  if constexpr (I == 90) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#ifdef CSMOE_V4_SB
  __builtin_amdgcn_sched_barrier(0);
#endif
}

template <int COLK, int RBN, int CUR, int... Is>
__device__ __forceinline__ void k_tile(f32x4 (&acc)[8][8], LoopState& st, int t2, std::integer_sequence<int, Is...>) {
  (slot<COLK, RBN, CUR, Is>(acc, st, t2), ...);
}
template <int... Is>
__device__ __forceinline__ void load_all(LoopState& st, int tile, std::integer_sequence<int, Is...>) { (load_piece<Is>(st, tile), ...); }
template <int... Is>
__device__ __forceinline__ void store_all0(LoopState& st, std::integer_sequence<int, Is...>) { (store_piece<Is, 0>(st), ...); }
template <int RBN, int... Is>
__device__ __forceinline__ void read_r0_all(LoopState& st, std::integer_sequence<int, Is...>) {
  ((Is < RBN ? (void)(st.fr0[Is] = read_r<0, 0, Is>(st)) : (void)0), ...);
}

// ---- the same K-tile with the operands brought in by LDS-DMA (buffer_load ... lds) instead of registers + ds_write_b128 --------
// What the vendor's hand-written 256x256x64 kernel does (disassembly of hipBLASLt's Custom_Cijk_Alik_Bljk_..._MT256x256x64: 4 waves,
// 128 x 128 per wave, 16 `buffer_load_dwordx4 ... lds` + 32 ds_read_b128 + 3 s_barrier per wave and K-tile, all four fragment sets in
// registers, 92 % MFMA-busy): no VGPR staging, no LDS store instructions, no address VALU.  A slot half can only be re-filled once
// EVERY wave has read it, so all of K-step 1's fragments are read early (every wave then holds the whole K-tile in registers) and the K-tile takes two barriers:
//   I =  0..31          read the 16 fragments of K-step 1 (slot CUR), one every other slot
//   I = 34              lgkmcnt(0) + barrier: every wave holds all of K-tile s in registers, slot CUR is free
//   I = 36 + 5 k + W    DMA piece k (< 16) of K-tile s+2 -> slot CUR: the CU's 64 pieces spread evenly over 3/4 of the K-tile (in
//                       two bursts of 8 between three barriers the address path was offered twice what it moves and blocked the
//                       issuing waves: +1.1 ms on a 3.0 ms MFMA stream, tools/v4_ablate.sh)
//   I = 94              vmcnt(pieces issued so far) + barrier: K-tile s+1, issued a K-tile ago, is in slot NXT for everyone
//   I = 96..127         read the 16 fragments of K-step 0 of K-tile s+1 (slot NXT)
#ifndef CSMOE_V4_ABL
#define CSMOE_V4_ABL 0      // diagnostic twins (tools/v4_ablate.sh): bit 0 = no fragment reads in the loop, bit 1 = no LDS-DMA in the loop, bit 2 = no landing wait (wrong results: timing only)
#endif
template <int I, int SL>
__device__ __forceinline__ void dma_piece4(LoopState& st, int tile) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(I < 8 ? st.rs_r : st.rs_c,
                                           (lds_void*)(st.smem + SL * (4 * TILE_B) + kind_slot(I >> 2) * TILE_B + (I & 3) * 1024 + st.wave4k),
                                           16, st.vb[I], (unsigned)tile * (I < 8 ? st.kstep_r : st.kstep_c), 0, 0);
}

// W = the wave's number: the four waves of the workgroup run the SAME slot table shifted against each other -- wave W issues its DMA
// pieces one slot (one MFMA, 16 cycles) after wave W-1 and its fragment reads on the other slot parity -- so that the CU's one
// address path and one LDS take the four waves' requests one after the other.  (All four at the same slots, which is where the
// barriers leave them: +1.15 ms of DMA issue time on a 3.0 ms MFMA stream, tools/v4_ablate.sh; the vendor's kernel runs two
// instruction orders selected by SIMD parity for the same reason.)
template <int COLK, int RBN, int CUR, int W, int I>
__device__ __forceinline__ void slot_dma(f32x4 (&acc)[8][8], LoopState& st, int t2) {
  constexpr int NXT = CUR ^ 1;
  constexpr int gi = I >> 3, ks = gi >> 3, cb = gi & 7, rb = I & 7;
  constexpr bool RD = !(CSMOE_V4_ABL & 1), DM = !(CSMOE_V4_ABL & 2);
  constexpr int P = W & 1;                     // read slots: parity
  constexpr int D0 = 36 + W, DSTEP = 5;        // DMA piece k at slot D0 + 5 k: the 64 pieces of a K-tile spread over 3/4 of it
  constexpr int LAND = 94;                     // slot of the landing barrier
  if constexpr (rb < RBN) {
    if constexpr (ks == 0) mfma_acc(acc[cb][rb], st.fc0[cb], st.fr0[rb]);
    else                   mfma_acc(acc[cb][rb], st.fc1[cb], st.fr1[rb]);
  }
  if constexpr (RD && I < 32 && (I & 1) == P) {                // K-step 1's 16 fragments, early: slot CUR is free after them
    constexpr int k = I >> 1;
    if constexpr (k < 8) { if constexpr (k < RBN) st.fr1[k] = read_r<CUR, 1, k>(st); }
    else st.fc1[k - 8] = read_c<COLK, CUR, 1, (k - 8)>(st);
  }
  if constexpr (I == 34) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (I == LAND) {                                   // K-tile s+1 (issued a K-tile ago) landed, for every wave
    constexpr int issued = (LAND - D0 + DSTEP - 1) / DSTEP;    // this K-tile's pieces issued so far (slots < LAND) stay in flight
    if constexpr (DM && !(CSMOE_V4_ABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(issued) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (DM && I >= D0 && I <= D0 + 15 * DSTEP && ((I - D0) % DSTEP) == 0) dma_piece4<((I - D0) / DSTEP), CUR>(st, t2);
  if constexpr (RD && I >= 96 && (I & 1) == P) {
    constexpr int k = (I - 96) >> 1;
    if constexpr (k < 8) { if constexpr (k < RBN) st.fr0[k] = read_r<NXT, 0, k>(st); }
    else st.fc0[k - 8] = read_c<COLK, NXT, 0, (k - 8)>(st);
  }
}
template <int COLK, int RBN, int CUR, int W, int... Is>
__device__ __forceinline__ void k_tile_dma(f32x4 (&acc)[8][8], LoopState& st, int t2, std::integer_sequence<int, Is...>) {
  (slot_dma<COLK, RBN, CUR, W, Is>(acc, st, t2), ...);
}
template <int SL, int... Is>
__device__ __forceinline__ void dma_all(LoopState& st, int tile, std::integer_sequence<int, Is...>) { (dma_piece4<Is, SL>(st, tile), ...); }
template <int COLK, int... Is>
__device__ __forceinline__ void read_c0_all(LoopState& st, std::integer_sequence<int, Is...>) {
  ((st.fc0[Is] = read_c<COLK, 0, 0, Is>(st)), ...);
}

template <int COLK, int RBN, int W>
__device__ __forceinline__ void k_loop_dma_w(f32x4 (&acc)[8][8], LoopState& st, int nk) {
  using S128 = std::make_integer_sequence<int, 128>;
  for (int s = 0; s < nk; s += 2) {
    k_tile_dma<COLK, RBN, 0, W>(acc, st, min(s + 2, nk - 1), S128{});
    k_tile_dma<COLK, RBN, 1, W>(acc, st, min(s + 3, nk - 1), S128{});
  }
}

template <int COLK, int RBN>
__device__ __forceinline__ void k_loop_dma(f32x4 (&acc)[8][8], LoopState& st, int nk, int wave) {
  using S16 = std::make_integer_sequence<int, 16>;
  using S8 = std::make_integer_sequence<int, 8>;
  dma_all<0>(st, 0, S16{});
  dma_all<1>(st, nk > 1 ? 1 : 0, S16{});
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // K-tile 0 landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_r0_all<RBN>(st, S8{});
  read_c0_all<COLK>(st, S8{});
  switch (wave) {                                          // wave-uniform: four copies of the loop, one slot shift each
    case 0: k_loop_dma_w<COLK, RBN, 0>(acc, st, nk); break;
    case 1: k_loop_dma_w<COLK, RBN, 1>(acc, st, nk); break;
    case 2: k_loop_dma_w<COLK, RBN, 2>(acc, st, nk); break;
    default: k_loop_dma_w<COLK, RBN, 3>(acc, st, nk); break;
  }
  settle_last_group(acc);
}

template <int COLK, int RBN>
__device__ __forceinline__ void k_loop(f32x4 (&acc)[8][8], LoopState& st, int nk) {
  using S16 = std::make_integer_sequence<int, 16>;
  using S128 = std::make_integer_sequence<int, 128>;
  // ---- prologue: K-tile 0 into slot 0, K-tile 1 in flight, the first fragments of K-tile 0 in registers
  load_all(st, 0, S16{});
  store_all0(st, S16{});
  load_all(st, nk > 1 ? 1 : 0, S16{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_r0_all<RBN>(st, std::make_integer_sequence<int, 8>{});
  st.fcq[0] = read_c<COLK, 0, 0, 0>(st);
  st.fcq[1] = read_c<COLK, 0, 0, 1>(st);
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 0; s < nk; s += 2) {             // nk is even (gg4_rowspace_ok); K-tiles past the end re-load the last one
    k_tile<COLK, RBN, 0>(acc, st, min(s + 2, nk - 1), S128{});
    k_tile<COLK, RBN, 1>(acc, st, min(s + 3, nk - 1), S128{});
  }
  settle_last_group(acc);
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
    if (!grouped_find_tile(p.offsets, p.E, p.single_M, BM4, nct, v, lane, tp)) return;
  }
  const int e = __builtin_amdgcn_readfirstlane(tp.e);
  const int row0 = __builtin_amdgcn_readfirstlane(tp.o0 + tp.mt * BM4);
  const int rows = __builtin_amdgcn_readfirstlane(min(BM4, tp.o1 - row0));
  const int tc0 = __builtin_amdgcn_readfirstlane(tp.nt * BN4);

  // ---------------- operand descriptors + per-lane source offsets of the 16 pieces ----------------
  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldc_b = (unsigned)p.ld_c * 2u;
  LoopState st;
  st.rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);
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
  st.kstep_r = BK4 * 2u;                                  // bytes one K-tile advances a K-contiguous row
  st.kstep_c = COLK == KC ? BK4 * 2u : BK4 * ldc_b;       // ... / 64 rows of a K-major matrix
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
    st.st_base = smem + wave * 4096 + lane * 16;
    st.wave4k = wave * 4096;
  }

  f32x4 acc[8][8];     // [column block][row block]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kd / BK4;
  // live 16-row blocks of this wave's quadrant: a wave whose rows lie past the expert's range issues its loads and stores and takes
  // every barrier but no MFMA; the remainder tile of an expert (1..64 rows) keeps half of wave-row 0's
  const int rows_here = rows - wm * 128;
  (void)rows_here;
  if (p.tile_classes) k_loop<COLK, 8>(acc, st, nk);          // CSMOE_V4_STAGE=reg: operands staged through registers (A/B)
  else k_loop_dma<COLK, 8>(acc, st, nk, wave);

  // every wave's last fragment reads are done before the staging tile overlays the images
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  const EpiArgs ea{p.C, p.C2, p.aux, p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias, p.ldc, p.epilogue, p.act, p.NC};
#ifdef CSMOE_V4_DEV
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

// shapes the one-wave-per-SIMD kernel takes beyond gg_fast_rowspace_ok: whole K-tiles (no per-lane K bound in its loads)
bool gg4_rowspace_ok(int Kd) { return Kd % (2 * BK4) == 0; }   // whole PAIRS of K-tiles: the loop body is two K-tiles (slot parity)

int gg4_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                 const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                 const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                 hipStream_t st) {
  FastArgs p{};
  p.single_M = M; p.single_B = single_B; p.single_bias = single_bias;
  p.R = A; p.ld_r = lda; p.c_ptrs_in = b_ptrs; p.ld_c = ldb; p.bias_ptrs = bias_ptrs; p.offsets = offsets; p.E = E;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  {
    static const int reg_stage = [] { const char* e = getenv("CSMOE_V4_STAGE"); return e && e[0] == 'r' ? 1 : 0; }();
    p.tile_classes = reg_stage;
  }
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
