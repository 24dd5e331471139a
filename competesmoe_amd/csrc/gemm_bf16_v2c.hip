// Row-space grouped GEMM whose weight operand is read straight from FP32 MASTERS: C[m, :] = epilogue(A[m, :] (bf16) x B_e) with
// B_e [Kd, N] fp32 (the pretrain stack's `keys [E, D, F]` / `values [E, F, D]` under bf16 autocast).  The reference's Triton kernel
// converts its operand tiles in registers (`a.to(tl.bfloat16)`, moe_pretrain_model/layers/cvmm.py:126-140); round 1 cast the whole
// tensors to bf16 first, an HBM-bound pass of ~6 ms per step at the headline shape (34.5 GB).  Here the cast is part of the tile fill:
//
//   * same 256x256 tile, 8 waves, two phases per K-tile and half-phase stagger as the WIDE loop of gemm_bf16_v2.hip; the row (A)
//     images RL / RH still arrive by LDS-DMA;
//   * the column (weight) images CL / CH are filled THROUGH REGISTERS: every thread owns two 16-byte pieces of each 64 x 128 K-major
//     image (the piece -> (k, 8 columns) map of dma_setup<KM>, so the fragment reads do not change), fetches their 8 fp32 with two
//     `buffer_load_dwordx4` (inline asm: hipcc would wait vmcnt(0) for an ordinary load beside LDS-DMA, guide section 5 item 4b),
//     converts with v_cvt_pk_bf16_f32 (round to nearest even = torch's cast, so results are bit-identical to "cast, then GEMM")
//     and writes one `ds_write_b128`.  One register set of 16 VGPRs: image X is loaded during one phase and written at the start
//     of the next -- CL(s+1) loaded in phase A(s), written in B(s), read in A(s+1); CH(s+1) loaded in B(s), written in A(s+1), read
//     in B(s+1).  The loads share the in-order vmcnt queue with the DMA pieces: vmcnt(4) at the start of A (the 4 row pieces issued
//     after the CH loads stay in flight), vmcnt(0) at the start of B (everything older was issued a full phase ago).
//     Measured at the headline shape (gpurun_out r2k / r2l, same box): 7.74 ms per launch against 6.36 ms for the LDS-DMA kernel on
//     pre-cast bf16 weights, the step 41.0 ms against 41.8 ms with the two cast passes -- a net 2 %, because the conversion's issue work
//     (8 loads, 16 cvt, 4 ds_write_b128 per wave and K-tile) sits in the read sections between the barriers, not under the MFMAs.  A
//     row-cut (BAL) variant with the quads fetched TWO phases ahead (32 staging VGPRs) was bit-identical too and SLOWER (8.17 ms): load
//     latency is not what costs.  What costs is the vector-memory stream itself: 8 fp32 quads + 4 row pieces per wave and K-tile are 12
//     instructions and 96 KiB per CU where the bf16 kernel has 8 and 64 KiB, and the bf16 loop is already paced by that stream
//     (DESIGN.md section 3, round-2 stamps: the DMA stream alone takes 1.21 us of a 1.7 us K-tile);
//   * the tiles of an expert's FIRST row tile also store the converted pieces to a bf16 copy of the weights (`b_copy`), which the two
//     backward GEMMs of the step read through the plain LDS-DMA kernels -- experts without rows write nothing and are read by nobody;
//   * round 3: with a copy requested, ONLY the first row tiles run this kernel (`row_part` = 1); the expert's other row tiles follow
//     in a second launch of the LDS-DMA kernel over the copy (gg8_rowspace_rest, stream order makes the copy visible).  At the
//     headline shape an expert has ~4 row tiles, so 3/4 of the work runs at the bf16 kernel's rate and the fp32 panel of an
//     (expert, column tile) is fetched once instead of by four workgroups.
#include "gemm_epilogue.h"
#include <algorithm>

using namespace ggt;

namespace {

constexpr int BMc = 256, BNc = 256, BKc = 64;
constexpr int LDSc_BYTES = EPI_LDS_BYTES;     // the epilogue's staging tile (>= the 8 operand images)

struct CvtArgs {
  const void* R; int64_t ld_r;
  const void* const* b_ptrs; const void* single_B; int64_t ld_c;          // fp32 [Kd, N]
  void* const* copy_ptrs; void* single_copy;                              // bf16 [Kd, N] (leading dim NC) or null
  const void* const* bias_ptrs; const void* single_bias;
  const int32_t* offsets; int E; int single_M;
  int NC, Kd;
  void* C; void* C2; const void* aux; int64_t ldc;
  int epilogue, act;
  int row_part;                                                           // 0 every row tile, 1 every expert's first only (common.h)
};

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
  return v;
}

__global__ void __launch_bounds__(512, 2) gg8c_kernel(CvtArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int e, row0, rows, tc0, mt;
  const int nct = (p.NC + BNc - 1) / BNc;
  {
    const int total = grouped_total_tiles(p.offsets, p.E, p.single_M, BMc, nct, lane, p.row_part);
    if ((int)blockIdx.x >= total) return;
    const int v = xcd_remap(blockIdx.x, total);
    TilePos tp;
    if (!grouped_find_tile(p.offsets, p.E, p.single_M, BMc, nct, v, lane, tp, p.row_part)) return;
    e = tp.e; mt = tp.mt;
    row0 = tp.o0 + tp.mt * BMc; rows = min(BMc, tp.o1 - row0);
    tc0 = tp.nt * BNc;
  }
  e = __builtin_amdgcn_readfirstlane(e);
  mt = __builtin_amdgcn_readfirstlane(mt);
  row0 = __builtin_amdgcn_readfirstlane(row0);
  rows = __builtin_amdgcn_readfirstlane(rows);
  tc0 = __builtin_amdgcn_readfirstlane(tc0);
  const int red_len = p.Kd;

  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldw_b = (unsigned)p.ld_c * 4u;
  __amdgpu_buffer_rsrc_t rs_r, rs_w;
  unsigned vb_rl[2], vb_rh[2];
  int ax_r[2], ax_dummy[2];
  rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);
  dma_setup<KC, 2>(vb_rl, ax_r, ldr_b, 0, 0, 7, 0, 0, wave, lane);
  dma_setup<KC, 2>(vb_rh, ax_dummy, ldr_b, 0, 0, 7, 0, 128, wave, lane);
  const char* wb = (const char*)(p.b_ptrs ? p.b_ptrs[e] : p.single_B);
  rs_w = make_rsrc(wb, (unsigned)p.Kd * ldw_b);
  bf16* copy = (bf16*)(p.copy_ptrs ? p.copy_ptrs[e] : p.single_copy);
  const bool do_copy = copy != nullptr && mt == 0;

  // this thread's two pieces of a K-major image: piece P = (wave * 2 + j) * 64 + lane -> (k, 8 columns from `ic`), as dma_setup<KM>
  unsigned w_off[2][2];        // [image CL / CH][piece]: byte offset inside a K-tile of the fp32 source, or OOB
  int c_k[2], c_col[2][2];     // k row of the piece; first column (tile-relative, image included) for the bf16 copy
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int P = (wave * 2 + j) * 64 + lane;
    const int k = P >> 4, pc = P & 15;
    const int f = (k & 3) | (((k >> 3) & 1) << 2);
    const int seg = (pc >> 1) ^ f;
    const int ic = seg * 16 + (pc & 1) * 8;
    c_k[j] = k;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      const int col = tc0 + im * 128 + ic;
      c_col[im][j] = col;
      w_off[im][j] = (col < p.NC) ? ((unsigned)k * ldw_b + (unsigned)col * 4u) : OOB;       // N % 8 == 0: a piece is all in or all out
    }
  }

  const int g = lane >> 4, i16 = lane & 15;
  const int kc_lane = i16 * 128 + ((g ^ (i16 >> 1)) << 4);
  const int q = i16 >> 2, pp = i16 & 3;
  const int fk = q | ((g & 1) << 2);
  const int r_blk0 = wm * 4, c_blk0 = wn * 2;
  int km_c[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) km_c[b] = (8 * g + q) * 256 + (((c_blk0 + b) ^ fk) << 5) + pp * 8;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (red_len + BKc - 1) / BKc;

#define SLOT(kind, tile) (smem + ((((tile) & 1) * 4 + (kind)) * TILE_B))
#define ISSUE_RL(tile) dma_tile<KC, 2>(rs_r, SLOT(0, tile), vb_rl, ax_r, (tile) * BKc, red_len, ldr_b, wave)
#define ISSUE_RH(tile) dma_tile<KC, 2>(rs_r, SLOT(3, tile), vb_rh, ax_r, (tile) * BKc, red_len, ldr_b, wave)
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
  // the four fp32 quads of one image of K-tile `tile` (rows past Kd fall off the descriptor: zeros)
#define LOAD_IMG(im, tile)                                                                           \
  do {                                                                                               \
    const unsigned kt_ = (unsigned)(tile) * (unsigned)BKc * ldw_b;                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
      const unsigned o_ = w_off[im][j] == OOB ? OOB : w_off[im][j] + kt_;                            \
      wr[2 * j] = bload4(rs_w, o_);                                                                  \
      wr[2 * j + 1] = bload4(rs_w, o_ == OOB ? OOB : o_ + 16u);                                      \
    }                                                                                                \
  } while (0)
  // wait until all but the N youngest vector-memory operations of this wave are done; the asm "modifies" the staging registers so
  // that no conversion of them can be scheduled above the wait
#define WAIT_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]) :: "memory")
  // convert the staged quads and write them as the two pieces of image `kind` (1 = CL, 2 = CH) of K-tile `tile`
#define STORE_IMG(kind, im, tile)                                                                    \
  do {                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
      const f32x4 a_ = wr[2 * j], b_ = wr[2 * j + 1];                                                \
      bf16x8 o_;                                                                                     \
      o_[0] = (bf16)a_[0]; o_[1] = (bf16)a_[1]; o_[2] = (bf16)a_[2]; o_[3] = (bf16)a_[3];            \
      o_[4] = (bf16)b_[0]; o_[5] = (bf16)b_[1]; o_[6] = (bf16)b_[2]; o_[7] = (bf16)b_[3];            \
      *(bf16x8*)(SLOT(kind, tile) + ((wave * 2 + j) * 64 + lane) * 16) = o_;                         \
      if (do_copy) {                                                                                 \
        const int kk_ = (tile) * BKc + c_k[j];                                                       \
        if (kk_ < p.Kd && c_col[im][j] < p.NC) *(bf16x8*)(copy + (int64_t)kk_ * p.NC + c_col[im][j]) = o_;  \
      }                                                                                              \
    }                                                                                                \
  } while (0)

  const int rows_here = rows - wm * 64;
  const int cols_here = min(BNc, p.NC - tc0) - wn * 32;
  const bool actA = rows_here > 0 && cols_here > 0, actAh = rows_here > 128 && cols_here > 0;
  const bool actB = rows_here > 0 && cols_here > 128, actBh = rows_here > 128 && cols_here > 128;

  f32x4 wr[4];
  // prologue: rows of K-tiles 0 and 1 by DMA; CL(0) through registers; CH(0) left in flight for phase A(0)
  ISSUE_RL(0); ISSUE_RH(0);
  LOAD_IMG(0, 0);
  WAIT_W(0);
  STORE_IMG(1, 0, 0);
  LOAD_IMG(1, 0);
  ISSUE_RL(1); ISSUE_RH(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // phase A: CH(s) from the registers into LDS; fragments of CL, RL, RH(s); loads of CL(s+1)
  // phase B: CL(s+1) from the registers into LDS; fragments of CH(s); loads of CH(s+1); row pieces of K-tile s+2
#define WIDE_PRE_A(s) WAIT_W(4); /* CH(s) quads are in (the 4 row pieces issued after them stay in flight) */ STORE_IMG(2, 1, s);
#define WIDE_PRE_B(s) WAIT_W(0); /* CL(s+1) quads are in; so are the row pieces of K-tile s+1 (a phase old) */ STORE_IMG(1, 0, s + 1);
#define WIDE_READ_CL(cb, ks) fc[cb][ks] = frag_km_raw(i_cl, km_c[cb], ks);
#define WIDE_READ_CH(cb, ks) fc[cb][ks] = frag_km_raw(i_ch, km_c[cb], ks);
#define WIDE_READ_RL(rb, ks) fr[rb][ks] = frag_kc(i_rl, kc_lane, r_blk0 + rb, ks);
#define WIDE_READ_RH(rb, ks) fr[4 + rb][ks] = frag_kc(i_rh, kc_lane, r_blk0 + rb, ks);
#define WIDE_ISSUE_A(s) LOAD_IMG(0, s + 1)
#define WIDE_ISSUE_B(s) LOAD_IMG(1, s + 1); ISSUE_RL(s + 2); ISSUE_RH(s + 2)
#define WIDE_WAIT_A(s)
#define WIDE_WAIT_B(s)
#include "gemm_loop_wide.inc"
  if (wm == 0) __builtin_amdgcn_s_barrier();

  // the loads / zero-fill DMAs of the K-tiles past the end may still be in flight: drain before the staging tile reuses LDS
  WAIT_W(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---------------- epilogue (gemm_epilogue.h) ----------------
  const EpiArgs ea{p.C, p.C2, p.aux, p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias, p.ldc, p.epilogue, p.act, p.NC};
  rowspace_epilogue(ea, acc, smem, row0, rows, tc0, wm, wn, lane);
}

}  // namespace

int gg8c_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int64_t ldb, void* const* copy_ptrs,
                  const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2, const void* aux,
                  int64_t ldc, int epilogue, int act, const void* single_B, void* single_copy, const void* single_bias, hipStream_t st,
                  int row_part) {
  CvtArgs p{};
  p.row_part = row_part;
  p.R = A; p.ld_r = lda; p.b_ptrs = b_ptrs; p.single_B = single_B; p.ld_c = ldb; p.copy_ptrs = copy_ptrs; p.single_copy = single_copy;
  p.bias_ptrs = bias_ptrs; p.single_bias = single_bias; p.offsets = offsets; p.E = E; p.single_M = M;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  const int nct = (N + BNc - 1) / BNc;
  const int64_t grid = row_part == 1 ? (int64_t)nct * E : (int64_t)nct * ((M + BMc - 1) / BMc + E);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_gemm_f32w: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  static bool done = false;
  if (!done) {
    hipError_t er = hipFuncSetAttribute((const void*)gg8c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSc_BYTES);
    if (er != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(er)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  hipLaunchKernelGGL(gg8c_kernel, dim3((unsigned)grid), dim3(512), LDSc_BYTES, st, p);
  CSMOE_CHECK_LAUNCH("grouped_gemm_f32w");
  return CSMOE_OK;
}
