// Grouped expert GEMM on the block-scaled fp8 matrix pipe of gfx950 (BASELINE config 5): C[m, :] = epilogue(sum_k A[m,k] B_e[n,k])
// with MXFP8 operands -- OCP e4m3 elements, one e8m0 scale per 32 elements along k -- on `v_mfma_scale_f32_16x16x128_f8f6f4`
// (2x the bf16 rate, dequantisation fused into the instruction), fp32 accumulators, bf16 outputs through the same epilogues as the
// bf16 kernel.  No counterpart upstream (the reference has no fp8).
//
// Structure = the 256x256, 8-wave, two-phase ("BAL") loop of gemm_bf16_v2.hip, byte for byte: a K-tile is 128 fp8 elements = the
// same 128-byte image rows as 64 bf16, so the LDS-DMA fill, the source-side swizzle, the four image kinds (RL / RH / CL / CH), the
// counted vmcnt discipline and the half-phase stagger carry over unchanged.  What differs:
//   * a fragment is 32 k-bytes of one row in the instruction's own k order, measured with tools/probes/mfma_scale_probe.hip: lane l
//     (row l & 15, group g = l >> 4) holds k = 16 g .. 16 g + 15 in its first four registers and k = 64 + 16 g .. + 15 in the last
//     four -- the SAME two 16-byte chunks (g, g + 4) the bf16 kernel reads for its two K = 32 steps -- and the scale of the
//     32-element block b (k = 32 b .. 32 b + 31) is taken from the lanes of group b;
//   * one K = 128 MFMA per (column block, row block) and K-tile instead of two K = 32 ones: 16 per phase at 32 cycles each;
//   * the e8m0 scales travel by LDS-DMA too (4 bytes per row and K-tile: one 256-byte wave-instruction per wave and K-tile --
//     waves 0-3 the 256 row scales, waves 4-7 the 256 column scales), are read as bytes in phase A and handed to the MFMA as
//     per-lane scale operands (lane l supplies the scale of ITS 32-element block).
#include "gemm_epilogue.h"
#include <algorithm>

using namespace ggt;

namespace {

constexpr int BM8 = 256, BN8 = 256;
constexpr int SC_OFF = 8 * TILE_B;                       // scale slots behind the 8 image slots: 2 x (1 KiB rows + 1 KiB columns)
constexpr int LDS8_BYTES = SC_OFF + 4096;                // 8 image slots + the scale slots behind them
static_assert(LDS8_BYTES >= EPI_LDS_BYTES, "the epilogue's staging tile overlays the operand images");

typedef __attribute__((ext_vector_type(8))) int i32x8;

struct Fp8Args {
  const uint8_t* R; int64_t ld_r;                 // A elements [M, Kd]
  const uint8_t* RS; int64_t ld_rs;               // A scales   [M, Kd/32]
  const void* const* c_ptrs; const void* const* cs_ptrs; int64_t ld_c, ld_cs;     // per-expert B [N, Kd], scales [N, Kd/32]
  const void* single_B; const void* single_BS;
  const void* const* bias_ptrs; const void* single_bias;
  const int32_t* offsets; int E; int single_M;
  int NC, Kd;
  void* C; void* C2; const void* aux; int64_t ldc;
  int epilogue, act;
};

__global__ void __launch_bounds__(512, 2) gg8f_kernel(Fp8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int e, row0, rows, tc0;
  const int nct = (p.NC + BN8 - 1) / BN8;
  {
    const int total = grouped_total_tiles(p.offsets, p.E, p.single_M, BM8, nct, lane);
    if ((int)blockIdx.x >= total) return;
    const int v = xcd_remap(blockIdx.x, total);
    TilePos tp;
    if (!grouped_find_tile(p.offsets, p.E, p.single_M, BM8, nct, v, lane, tp)) return;
    e = tp.e;
    row0 = tp.o0 + tp.mt * BM8; rows = min(BM8, tp.o1 - row0);
    tc0 = tp.nt * BN8;
  }
  e = __builtin_amdgcn_readfirstlane(e);
  row0 = __builtin_amdgcn_readfirstlane(row0);
  rows = __builtin_amdgcn_readfirstlane(rows);
  tc0 = __builtin_amdgcn_readfirstlane(tc0);
  // the DMA helpers count the reduction in 2-byte units (they were written for bf16): 64 units = one 128-byte K-tile
  const int red_len = p.Kd >> 1;
  const int ncols = min(BN8, p.NC - tc0);

  const unsigned ldr_b = (unsigned)p.ld_r, ldc_b = (unsigned)p.ld_c;
  const unsigned ldrs = (unsigned)p.ld_rs, ldcs = (unsigned)p.ld_cs;
  const unsigned nsb = (unsigned)(p.Kd >> 5);          // scale bytes per row
  __amdgpu_buffer_rsrc_t rs_r, rs_c, rs_rs, rs_cs;
  unsigned vb_rl[2], vb_rh[2], vb_cl[2], vb_ch[2];
  int ax_r[2], ax_c[2], ax_dummy[2];
  rs_r = make_rsrc(p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);
  dma_setup<KC, 2>(vb_rl, ax_r, ldr_b, 0, 0, 7, 0, 0, wave, lane);
  dma_setup<KC, 2>(vb_rh, ax_dummy, ldr_b, 0, 0, 7, 0, 128, wave, lane);
  const char* wb = (const char*)(p.c_ptrs ? p.c_ptrs[e] : p.single_B);
  const char* wsb = (const char*)(p.cs_ptrs ? p.cs_ptrs[e] : p.single_BS);
  rs_c = make_rsrc(wb + (int64_t)tc0 * ldc_b, (unsigned)ncols * ldc_b);
  dma_setup<KC, 2>(vb_cl, ax_c, ldc_b, 0, 0, 7, 0, 0, wave, lane);
  dma_setup<KC, 2>(vb_ch, ax_dummy, ldc_b, 0, 0, 7, 0, 128, wave, lane);
  rs_rs = make_rsrc(p.RS + (int64_t)row0 * ldrs, (unsigned)rows * ldrs);
  rs_cs = make_rsrc(wsb + (int64_t)tc0 * ldcs, (unsigned)ncols * ldcs);
  // scale piece of this wave: waves 0-3 rows (wave * 64 + lane) of the tile, waves 4-7 columns ((wave - 4) * 64 + lane)
  const bool sc_rows = wave < 4;
  const unsigned sc_vbase = (unsigned)((wave & 3) * 64 + lane) * (sc_rows ? ldrs : ldcs);

  // ---------------- LDS read addressing ----------------
  const int g = lane >> 4, i16 = lane & 15;
  const int sw = (i16 >> 1) & 7;
  const int f_lo = i16 * 128 + ((g ^ sw) << 4);                // chunk g     of this lane's row (source-side swizzle of dma_setup)
  const int f_hi = i16 * 128 + (((g + 4) ^ sw) << 4);          // chunk g + 4
  const int r_blk0 = wm * 4, c_blk0 = wn * 2;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.Kd + 127) / 128;

#define SLOT(kind, tile) (smem + ((((tile) & 1) * 4 + (kind)) * TILE_B))
#define SCSLOT(tile) (smem + SC_OFF + ((tile) & 1) * 2048)
#define ISSUE_RL(tile) dma_tile<KC, 2>(rs_r, SLOT(0, tile), vb_rl, ax_r, (tile) * 64, red_len, ldr_b, wave)
#define ISSUE_CL(tile) dma_tile<KC, 2>(rs_c, SLOT(1, tile), vb_cl, ax_c, (tile) * 64, red_len, ldc_b, wave)
#define ISSUE_CH(tile) dma_tile<KC, 2>(rs_c, SLOT(2, tile), vb_ch, ax_c, (tile) * 64, red_len, ldc_b, wave)
#define ISSUE_RH(tile) dma_tile<KC, 2>(rs_r, SLOT(3, tile), vb_rh, ax_r, (tile) * 64, red_len, ldr_b, wave)
  // 4 scale bytes (the K-tile's four 32-element blocks) of one row / column per lane
#define ISSUE_SC(tile)                                                                                              \
  do {                                                                                                              \
    unsigned vo_ = sc_vbase + (unsigned)(tile) * 4u;                                                                \
    if ((unsigned)(tile) * 4u >= nsb) vo_ = OOB;                                                                    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sc_rows ? rs_rs : rs_cs, (lds_void*)(SCSLOT(tile) + wave * 256), 4,    \
                                             vo_, 0, 0, 0);                                                         \
  } while (0)
#define WAIT_DMA(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PHASE_SYNC_IN()                                \
  __builtin_amdgcn_sched_barrier(0);                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_setprio(1)
#define PHASE_SYNC_OUT()                               \
  __builtin_amdgcn_s_setprio(0);                       \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0)
#define FRAG(img, blk) ({                                                              \
    const i32x4 lo_ = *(const i32x4*)((img) + (blk) * 2048 + f_lo);                     \
    const i32x4 hi_ = *(const i32x4*)((img) + (blk) * 2048 + f_hi);                     \
    i32x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]}; })
#define MFMA8(ACC, FC, FR, SC, SR) \
  ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(FC, FR, ACC, 0, 0, 0, SC, 0, SR)

  const int rows_here = rows - wm * 64;
  const int cols_here = ncols - wn * 32;
  const bool rlo = rows_here > 0, rhi = rows_here > 128, clo = cols_here > 0, chi = cols_here > 128;

  if (rows <= 128) {
    // Thin tile (the remainder tile of an expert: at 128 experts x ~512 rows half of the experts end in one of a few dozen rows):
    // the one-phase loop of gemm_bf16_v2.hip -- no RH image, one barrier pair and 7 pieces per K-tile, every wave's 4 x 4 blocks
    // of C_all x R_lo in one MFMA section.
    ISSUE_CL(0); ISSUE_CH(0); ISSUE_RL(0); ISSUE_SC(0); ISSUE_CL(1); ISSUE_CH(1); ISSUE_RL(1); ISSUE_SC(1);
    WAIT_DMA(7);                                           // K-tile 0 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define THIN_FRAG_DECL const char* sc = SCSLOT(s); i32x8 fc[4], fr[4]; int sc_c[4], sc_r[4];
#define THIN_READ_CL(cb) fc[cb] = FRAG(i_cl, c_blk0 + cb); sc_c[cb] = *(const uint8_t*)(sc + 1024 + ((c_blk0 + cb) * 16 + i16) * 4 + g);
#define THIN_READ_CH(cb) fc[2 + cb] = FRAG(i_ch, c_blk0 + cb); sc_c[2 + cb] = *(const uint8_t*)(sc + 1024 + (128 + (c_blk0 + cb) * 16 + i16) * 4 + g);
#define THIN_READ_RL(rb) fr[rb] = FRAG(i_rl, r_blk0 + rb); sc_r[rb] = *(const uint8_t*)(sc + ((r_blk0 + rb) * 16 + i16) * 4 + g);
#define THIN_KSTEPS 1
#define THIN_MFMA(ks, cb, rb) MFMA8(acc[cb][rb], fc[cb], fr[rb], sc_c[cb], sc_r[rb]);
#define THIN_ISSUE(s) ISSUE_CL(s + 2); ISSUE_CH(s + 2); ISSUE_RL(s + 2); ISSUE_SC(s + 2)
#define THIN_WAIT(s) WAIT_DMA(7)                           /* K-tile s+1 landed (s+2 stays in flight) */
#include "gemm_loop_thin.inc"
  } else {
  // issue order ... [CL,CH,RL,SC](s+1) | RH(s+1) | [CL,CH,RL,SC](s+2) | RH(s+2) ...: 7 + 2 pieces per K-tile and wave, so the
  // counted waits of the bf16 loop (8) become 9
  ISSUE_CL(0); ISSUE_CH(0); ISSUE_RL(0); ISSUE_SC(0); ISSUE_RH(0); ISSUE_CL(1); ISSUE_CH(1); ISSUE_RL(1); ISSUE_SC(1);
  WAIT_DMA(9);                                           // CL, CH, RL, SC(0) landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();             // row half 1 starts half a phase late
  __builtin_amdgcn_sched_barrier(0);
  // phase A also reads ALL scales of K-tile s (the RH rows' too); a K = 128 MFMA per (column block, row block) and K-tile
#define BAL_FRAG_DECL const char* sc = SCSLOT(s); i32x8 fc[4], fr[4]; int sc_c[4], sc_r[8];
#define BAL_READ_CL(cb) fc[cb] = FRAG(i_cl, c_blk0 + cb); sc_c[cb] = *(const uint8_t*)(sc + 1024 + ((c_blk0 + cb) * 16 + i16) * 4 + g);
#define BAL_READ_CH(cb) fc[2 + cb] = FRAG(i_ch, c_blk0 + cb); sc_c[2 + cb] = *(const uint8_t*)(sc + 1024 + (128 + (c_blk0 + cb) * 16 + i16) * 4 + g);
#define BAL_READ_RL(rb) fr[rb] = FRAG(i_rl, r_blk0 + rb); sc_r[rb] = *(const uint8_t*)(sc + ((r_blk0 + rb) * 16 + i16) * 4 + g);
#define BAL_READ_RH(rb) fr[rb] = FRAG(i_rh, r_blk0 + rb);
#define BAL_PHASE_A_EXTRA                                                                                                      \
  if (rhi && clo) {                                                                                                            \
    _Pragma("unroll") for (int rb = 0; rb < 4; ++rb) sc_r[4 + rb] = *(const uint8_t*)(sc + (128 + (r_blk0 + rb) * 16 + i16) * 4 + g); \
  }
#define BAL_KSTEPS 1
#define BAL_MFMA(ks, cb, rb, arb) MFMA8(acc[cb][arb], fc[cb], fr[rb], sc_c[cb], sc_r[arb]);
#define BAL_ISSUE_A(s) ISSUE_RH(s + 1)
#define BAL_ISSUE_B(s) ISSUE_CL(s + 2); ISSUE_CH(s + 2); ISSUE_RL(s + 2); ISSUE_SC(s + 2)
#define BAL_WAIT_A(s) WAIT_DMA(9)                          /* RH(s) landed */
#define BAL_WAIT_B(s) WAIT_DMA(9)                          /* CL, CH, RL, SC(s+1) landed */
#include "gemm_loop_bal.inc"
  if (wm == 0) __builtin_amdgcn_s_barrier();
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---------------- epilogue: the bf16 kernels' (gemm_epilogue.h) ----------------
  const EpiArgs ea{p.C, p.C2, p.aux, p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias, p.ldc, p.epilogue, p.act, p.NC};
  rowspace_epilogue(ea, acc, smem, row0, rows, tc0, wm, wn, lane);
}

}  // namespace

int gg8f_rowspace(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* const* bq_ptrs, const void* const* bs_ptrs,
                  int64_t ldb, int64_t ldbs, const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd,
                  void* C, void* C2, const void* aux, int64_t ldc, int epilogue, int act, const void* single_B,
                  const void* single_BS, const void* single_bias, hipStream_t st) {
  Fp8Args p{};
  p.R = (const uint8_t*)Aq; p.ld_r = lda; p.RS = (const uint8_t*)As; p.ld_rs = ldas;
  p.c_ptrs = bq_ptrs; p.cs_ptrs = bs_ptrs; p.ld_c = ldb; p.ld_cs = ldbs; p.single_B = single_B; p.single_BS = single_BS;
  p.bias_ptrs = bias_ptrs; p.single_bias = single_bias; p.offsets = offsets; p.E = E; p.single_M = M;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  const int nct = (N + BN8 - 1) / BN8;
  const int64_t grid = (int64_t)nct * ((M + BM8 - 1) / BM8 + E);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_gemm_mxfp8: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  static bool done = false;
  if (!done) {
    hipError_t er = hipFuncSetAttribute((const void*)gg8f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8_BYTES);
    if (er != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(er)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  hipLaunchKernelGGL(gg8f_kernel, dim3((unsigned)grid), dim3(512), LDS8_BYTES, st, p);
  CSMOE_CHECK_LAUNCH("grouped_gemm_mxfp8");
  return CSMOE_OK;
}
