// Persistent weight-gradient GEMM for gfx950: dW[e] = A[rows of e]^T @ B[rows of e]  (bf16 operands, fp32 accumulate).
// Same 256x256 tile / 8 waves / K-tile 64 / two-phase staggered loop as the row-space kernel in gemm_bf16_v2.hip (see its
// header for the image kinds, the half-phase stagger of waves 4-7 and the counted-vmcnt rules); what differs is the outer
// structure and the placement of the DMA issue.  The headline launch is 44k tiles of only ~16 K-tiles each, so what happens
// BETWEEN tiles matters as much as the K-loop (s_memtime stamps of this kernel, -DCSMOE_STAMPS, profiles/r01/wgrad_stamps.txt:
// per tile ~48k cycles K-loop = 3.0k per K-tile of which 2,048 are MFMA, ~8k epilogue, ~1.7k everything else):
//   * one workgroup per CU walks its XCD's contiguous chunk of the tile order (no re-dispatch of a 512-thread / 130 KiB
//     workgroup per tile);
//   * the epilogue stages through the UPPER 64 KiB of LDS only, so K-tile 0 of the NEXT tile (4 images, the parity-0 slots in
//     the lower 64 KiB) is fetched by LDS-DMA while this tile is converted and stored; the wait for that prefetch is taken in
//     the last pass, BEFORE its global stores are issued (vmcnt retires in order on gfx9: waiting later would also wait for
//     the stores);
//   * bf16 output without accumulate (the training path) rounds in registers and stages bf16: two passes of 128 rows with
//     ds_write_b64, half the LDS write bytes and half the barriers of the fp32 staging (4 passes of 64 rows) that fp32 output /
//     accumulate keep (~8.1k vs ~10.1k cycles per tile);
//   * barriers of the epilogue wait for LDS only (__syncthreads() adds s_waitcnt vmcnt(0) = the prefetch and the stores), the
//     offsets / output-pointer tables are read with scalar loads (hipcc's vector loads cost a vmcnt(0) per use).
// Tried and dropped (same box A/B, tools/gemm_bench.py): storing the accumulators straight from registers (32-byte row
// segments per store: -6 %); a software-pipelined loop with single-buffered just-in-time fragments and ONE barrier per K-tile
// (2.64k cycles per K-tile with cache-resident operands, but 3.2k with real ones: its prefetch lead is one K-tile where the
// staggered loop has 1.5); a column-cut schedule with the DMA pieces of two images moved between the MFMAs (5.62 ms per headline
// launch against 5.55 for the row-cut loop; removed in round 3, commit 5bc2f9d has it).
// Both operands are K-major ([rows, features]): KM images read with ds_read_b64_tr_b16.
#include "gemm_tiles.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

using namespace ggt;

namespace {

#ifndef CSMOE_WG_VARIANT
#define CSMOE_WG_VARIANT 0
#endif
constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int CT2_LD = BN2 + 4;                       // fp32 staging row stride (floats)
constexpr int STG_OFF = 4 * TILE_B;                   // staging tile = upper half (the parity-1 slots)
constexpr int LDS2_BYTES = STG_OFF + 64 * CT2_LD * 4; // 132,096 B  (>= 8 * TILE_B = 131,072 B)

// Diagnostic build (-DCSMOE_STAMPS): workgroup 0 / thread 0 records s_memtime at the marked points of its first 24 tiles
// into the buffer passed in FastArgs::aux; gg8_wgrad prints the per-section cycle counts (tools/gemm_bench.py --which tn1).
#ifdef CSMOE_STAMPS
#define STAMP(k)                                                                                             \
  do {                                                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0 && tile_i < 24)                                                  \
      ((unsigned long long*)p.aux)[tile_i * 16 + (k)] = __builtin_readcyclecounter();                        \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// Gradient tiles are written once and read by the optimizer much later: non-temporal stores keep them from displacing the operand
// panels in L2.  CSMOE_WG_NT=0: plain stores (A/B).
#ifndef CSMOE_WG_NT
#define CSMOE_WG_NT 1
#endif
#if CSMOE_WG_NT
#define WG_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define WG_STORE(ptr, val) (*(ptr) = (val))
#endif

struct TileW { int e, row0, red_len, tr0, tc0; };     // wave-uniform
struct DmaW {
  __amdgpu_buffer_rsrc_t rs_r, rs_c;
  unsigned vb_rl[2], vb_rh[2], vb_cl[2], vb_ch[2];
};

// Scalar (SMEM) load of 8 bytes from a wave-uniform address.  Written as asm because hipcc turns a plain load of the offsets /
// pointer tables into a VECTOR load once the kernel also stores to global memory, and every use of such a load costs an
// `s_waitcnt vmcnt(0)`: a wait for the prefetch DMA and for all earlier stores of the epilogue.  The tables are written before
// the launch, so the scalar cache is coherent for them.
__device__ __forceinline__ uint64_t sload_b64(const void* ptr) {
  const uint64_t a = (uint64_t)ptr;
  const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)a);
  uint64_t r;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(u) : "memory");
  return r;
}

// v = position in this workgroup's tile list.  Without p.xcd_order: the global tile order (expert-major).  With it: the tiles of
// the experts dealt to this workgroup's XCD (csmoe_expert_order): entry [x * slots + k] = k-th expert of XCD x, heaviest first
__device__ __forceinline__ TileW tile_of(const FastArgs& p, int v, int nct, int per_e) {
  TileW t;
  int e = v / per_e;
  int local = v - e * per_e;
  if (p.xcd_order) {
    const int slots = (p.E + 7) >> 3;
    e = (int)(unsigned)sload_b64(p.xcd_order + (blockIdx.x & 7) * slots + e);     // 8-byte scalar load, low word used
  }
  int row0 = 0, red_len = p.single_M;
  if (p.offsets) {
    const uint64_t oo = sload_b64(p.offsets + e);          // offsets[e], offsets[e + 1]
    row0 = (int)(unsigned)oo;
    red_len = (int)(unsigned)(oo >> 32) - row0;
  }
  t.e = __builtin_amdgcn_readfirstlane(e);
  t.row0 = __builtin_amdgcn_readfirstlane(row0);
  t.red_len = __builtin_amdgcn_readfirstlane(red_len);
  // Tiles of an expert in bands of WG_BAND row tiles, column-major inside a band: the ~32 workgroups of an XCD hold consecutive
  // positions of this order, i.e. a WG_BAND x 8 block of tiles that shares WG_BAND row panels and 8 column panels (12 panels of
  // 256 x red_len for 32 tiles) where the row-major order had them share ONE row panel and fetch 32 column panels.
  // CSMOE_WGRAD_BAND=1 (A/B) is the row-major order.
  const int band = p.tile_band > 0 ? p.tile_band : 1;
  const int nrt = per_e / nct;
  const int grp = local / (band * nct), rem = local - grp * band * nct;
  const int g_eff = min(band, nrt - grp * band);
  t.tr0 = __builtin_amdgcn_readfirstlane((grp * band + rem % g_eff) * BM2);
  t.tc0 = __builtin_amdgcn_readfirstlane((rem / g_eff) * BN2);
  return t;
}

__device__ __forceinline__ DmaW dma_of(const FastArgs& p, const TileW& t_in, unsigned ldr_b, unsigned ldc_b, int wave, int lane) {
  DmaW d;
  int ax[2];
#ifdef CSMOE_FAKE_LOCAL
  TileW t = t_in; t.row0 = 0; t.tr0 = 0; t.tc0 = 0;      // timing experiment: every tile reads the same (cache-resident) panels
#else
  const TileW& t = t_in;
#endif
  d.rs_r = make_rsrc((const char*)p.R + (int64_t)t.row0 * ldr_b, (unsigned)t.red_len * ldr_b);
  d.rs_c = make_rsrc((const char*)p.Cflat + (int64_t)t.row0 * ldc_b, (unsigned)t.red_len * ldc_b);
  dma_setup<KM, 2>(d.vb_rl, ax, ldr_b, t.tr0, p.NR, 7, 0, 0, wave, lane);
  dma_setup<KM, 2>(d.vb_rh, ax, ldr_b, t.tr0, p.NR, 7, 0, 128, wave, lane);
  dma_setup<KM, 2>(d.vb_cl, ax, ldc_b, t.tc0, p.NC, 7, 0, 0, wave, lane);
  dma_setup<KM, 2>(d.vb_ch, ax, ldc_b, t.tc0, p.NC, 7, 0, 128, wave, lane);
  return d;
}

// The K-loop is the row-cut two-phase loop shared with the row-space kernels (gemm_loop_bal.inc): phases cut by ROW image,
// see gemm_bf16_v2.hip -- A reads CL, CH, RL(s) and issues RH(s+1), B reads RH(s) and issues CL, CH, RL(s+2), vmcnt(8) in both.
__global__ void __launch_bounds__(512, 2) gg8w_kernel(FastArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nct = (p.NC + BN2 - 1) / BN2;
  const int nrt = (p.NR + BM2 - 1) / BM2;
  const int per_e = nrt * nct;
  const int total = per_e * p.E;
  int v_begin, v_end, v_step;
  if (p.xcd_order) {
    // Experts DEALT to the 8 XCDs by row count (a tile's duration is proportional to its expert's rows): every XCD gets whole
    // experts (their operand panels stay in ITS L2) and about the same number of rows.  With a contiguous expert range per XCD a
    // skewed router (BASELINE's second regime: 8 hot experts with consecutive indices) put all hot experts on one XCD: 10.6 ms
    // per launch instead of 6.5.  Only the last slot of an XCD can be empty (E % 8 != 0).
    const int x = blockIdx.x & 7, slots = (p.E + 7) >> 3;
    const int last_pos = ((slots - 1) & 1) ? 7 - x : x;
    const int n_mine = slots - 1 + (8 * (slots - 1) + last_pos < p.E ? 1 : 0);
    v_begin = blockIdx.x >> 3; v_end = n_mine * per_e; v_step = gridDim.x >> 3;
  } else if ((gridDim.x & 7) == 0) {             // one contiguous chunk of the tile order per XCD (blocks id, id+8 share one)
    const int x = blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
    const int cs = (x < r8) ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8;
    v_begin = cs + (blockIdx.x >> 3); v_end = cs + q8 + (x < r8 ? 1 : 0); v_step = gridDim.x >> 3;
  } else {
    v_begin = blockIdx.x; v_end = total; v_step = gridDim.x;
  }
  if (v_begin >= v_end) return;
  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldc_b = (unsigned)p.ld_c * 2u;
  const int dummy_ax[2] = {0, 0};

  // slot(kind, parity) = (parity * 4 + kind) * 16 KiB with kind RL=0, CL=1, CH=2, RH=3
#define SLOT(kind, tile) (smem + ((((tile) & 1) * 4 + (kind)) * TILE_B))
#define ISSUE_RL(D, T, tile) dma_tile<KM, 2>(D.rs_r, SLOT(0, tile), D.vb_rl, dummy_ax, (tile) * BK2, T.red_len, ldr_b, wave)
#define ISSUE_CL(D, T, tile) dma_tile<KM, 2>(D.rs_c, SLOT(1, tile), D.vb_cl, dummy_ax, (tile) * BK2, T.red_len, ldc_b, wave)
#define ISSUE_CH(D, T, tile) dma_tile<KM, 2>(D.rs_c, SLOT(2, tile), D.vb_ch, dummy_ax, (tile) * BK2, T.red_len, ldc_b, wave)
#define ISSUE_RH(D, T, tile) dma_tile<KM, 2>(D.rs_r, SLOT(3, tile), D.vb_rh, dummy_ax, (tile) * BK2, T.red_len, ldr_b, wave)
#define WAIT_DMA(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // barrier A closes a phase's read/issue section (reads retired, counted DMA wait taken), barrier B its MFMA section
#define PHASE_SYNC_IN()                                \
  __builtin_amdgcn_sched_barrier(0);                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_setprio(1)
  // epilogue barrier: LDS traffic only.  __syncthreads() would add s_waitcnt vmcnt(0) = wait for the next tile's prefetch and
  // for this tile's global stores at every pass.
#define EPI_SYNC()                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                        \
  asm volatile("" ::: "memory")
#define PHASE_SYNC_OUT()                               \
  __builtin_amdgcn_s_setprio(0);                       \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0)

  // ---------------- first tile: descriptors + K-tile 0 ----------------
  TileW cur;
  DmaW dcur;
  {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < 512);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cur = tile_of(p, v_begin, nct, per_e);
    dcur = dma_of(p, cur, ldr_b, ldc_b, wave, lane);
    ISSUE_RL(dcur, cur, 0); ISSUE_RH(dcur, cur, 0); ISSUE_CL(dcur, cur, 0); ISSUE_CH(dcur, cur, 0);
  }

  bool first = true;
  int tile_i = -1;
  for (int v = v_begin; v < v_end; v += v_step) {
    ++tile_i;
    (void)tile_i;
    STAMP(0);
    // Everything derived from the thread id is tile-invariant; left visible, hipcc hoists dozens of such values out of this loop,
    // keeps them live across the K-loop and spills.  An opaque copy per tile pins them inside the iteration (~100 VALU per tile).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < 512);       // give the range back: LDS offsets fold into ds_read immediates again
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;       // row half, column quarter

    // ---------------- LDS read addressing ----------------
    const int g = lane >> 4, i16 = lane & 15;
    const int q = i16 >> 2, pp = i16 & 3;
    const int fk = q | ((g & 1) << 2);
    const int r_blk0 = wm * 4, c_blk0 = wn * 2;    // this wave's 16-wide blocks inside the R / C images
    int km_r[4], km_c[2];
#pragma unroll
    for (int b = 0; b < 4; ++b) km_r[b] = (8 * g + q) * 256 + (((r_blk0 + b) ^ fk) << 5) + pp * 8;
#pragma unroll
    for (int b = 0; b < 2; ++b) km_c[b] = (8 * g + q) * 256 + (((c_blk0 + b) ^ fk) << 5) + pp * 8;

    f32x4 acc[4][8];   // [column block][row block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (cur.red_len + BK2 - 1) / BK2;
    // ragged tiles: a wave whose strips lie outside the tile skips those reads and MFMAs (it still issues its DMA share)
    const int rows_here = min(BM2, p.NR - cur.tr0) - wm * 64;
    const int cols_here = min(BN2, p.NC - cur.tc0) - wn * 32;
    {
    const bool rlo = rows_here > 0, rhi = rows_here > 128, clo = cols_here > 0, chi = cols_here > 128;
    ISSUE_CL(dcur, cur, 1); ISSUE_CH(dcur, cur, 1); ISSUE_RL(dcur, cur, 1);
    if (first) {
      WAIT_DMA(6);                                           // all of K-tile 0 landed
      __builtin_amdgcn_s_barrier();
    }                                                        // later tiles: K-tile 0 confirmed in the previous epilogue
    if (wm == 1) __builtin_amdgcn_s_barrier();               // row half 1 starts half a phase late
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
#define BAL_FRAG_DECL bf16x8 fc[4][2], fr[4][2];
#define BAL_READ_CL(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[cb][ks] = frag_km_raw(i_cl, km_c[cb], ks);
#define BAL_READ_CH(cb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fc[2 + cb][ks] = frag_km_raw(i_ch, km_c[cb], ks);
#define BAL_READ_RL(rb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fr[rb][ks] = frag_km_raw(i_rl, km_r[rb], ks);
#define BAL_READ_RH(rb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fr[rb][ks] = frag_km_raw(i_rh, km_r[rb], ks);
#define BAL_PHASE_A_EXTRA
#define BAL_KSTEPS 2
#define BAL_MFMA(ks, cb, rb, arb) acc[cb][arb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[cb][ks], fr[rb][ks], acc[cb][arb], 0, 0, 0);
#define BAL_ISSUE_A(s) ISSUE_RH(dcur, cur, s + 1)
#define BAL_ISSUE_B(s) ISSUE_CL(dcur, cur, s + 2); ISSUE_CH(dcur, cur, s + 2); ISSUE_RL(dcur, cur, s + 2)
#define BAL_WAIT_A(s) if (s > 0 || first) WAIT_DMA(8)     /* RH(s) landed (later tiles: K-tile 0 confirmed in the previous epilogue) */
#define BAL_WAIT_B(s) WAIT_DMA(8)                          /* CL, CH, RL(s+1) landed */
#include "gemm_loop_bal.inc"
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();             // row half 0 waits for the staggered half to finish
    STAMP(2);

    // the zero-fill DMAs of the K-tiles past the end may still be writing LDS: drain before anything reuses the slots
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    STAMP(3);
    // ---------------- next tile: descriptors, K-tile 0 into the (now dead) parity-0 slots ----------------
    const bool has_next = v + v_step < v_end;
    TileW nxt = cur;
    DmaW dnxt = dcur;
    if (has_next) {
      nxt = tile_of(p, v + v_step, nct, per_e);
      dnxt = dma_of(p, nxt, ldr_b, ldc_b, wave, lane);
      ISSUE_RL(dnxt, nxt, 0); ISSUE_RH(dnxt, nxt, 0); ISSUE_CL(dnxt, nxt, 0); ISSUE_CH(dnxt, nxt, 0);
    }

    STAMP(4);
    // ---------------- epilogue: four passes of 64 rows through an fp32 tile in the upper half of LDS ----------------
    // pass (h, u): row blocks 4h + 2u + {0,1} of every wave -> staging row wm*32 + i*16 + (lane & 15)
    //              = tile row h*128 + wm*64 + u*32 + i*16 + (lane & 15)
    float* stg = (float*)(smem + STG_OFF);
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, g_e = lane_e >> 4, i16_e = lane_e & 15;
    const int ec = (tid_e & 31) * 8;             // this thread's 8 columns inside the 256-wide tile
    const int er = tid_e >> 5;                   // 0..15
    const int ncol = cur.tc0 + ec;
    typedef __attribute__((address_space(1))) char gchar;
    gchar* Ce = (gchar*)(p.out_ptrs ? (char*)sload_b64(p.out_ptrs + cur.e) : (char*)p.single_C);
    const int nrows = p.NR - cur.tr0;            // valid tile rows (may exceed 256)
    if (!p.out_f32 && !p.accumulate) {
      // bf16 output, no read-modify-write: round in registers and stage bf16 -- two passes of 128 rows (row stride 520 B:
      // the 8-byte ds_write_b64 of 16 consecutive rows fall on 16 different bank pairs), half the LDS write bytes and half
      // the barriers of the fp32 path below
      char* stg_b = smem + STG_OFF;
      constexpr int SB = BN2 * 2 + 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            const int m = wm * 64 + rb * 16 + i16_e;
            const int n = (cb >> 1) * 128 + wn * 32 + (cb & 1) * 16 + 4 * g_e;
            const f32x4 a = acc[cb][4 * h + rb];
            bf16x4 o4;
#pragma unroll
            for (int t = 0; t < 4; ++t) o4[t] = (bf16)a[t];
            *(bf16x4*)(stg_b + m * SB + n * 2) = o4;
          }
        EPI_SYNC();
        if (h == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's K-tile 0 landed; before this pass's stores
#pragma unroll
        for (int jj = 0; jj < 8; jj += 4) {
          bf16x8 row[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) row[j] = *(const bf16x8*)(stg_b + (er + 16 * (jj + j)) * SB + ec * 2);
          if (ncol < p.NC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int trow = h * 128 + er + 16 * (jj + j);
              if (trow < nrows) {
                typedef __attribute__((address_space(1))) bf16x8 gbf16x8;
                WG_STORE((gbf16x8*)(Ce + ((int64_t)(cur.tr0 + trow) * p.ldc + ncol) * 2), row[j]);
              }
            }
          }
        }
        EPI_SYNC();
        STAMP(5 + h);
      }
    } else
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int h = pass >> 1, u = pass & 1;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = wm * 32 + i * 16 + i16_e;
          const int n = (cb >> 1) * 128 + wn * 32 + (cb & 1) * 16 + 4 * g_e;
          *(f32x4*)(stg + m * CT2_LD + n) = acc[cb][4 * h + 2 * u + i];
        }
      EPI_SYNC();
      if (p.out_f32) {
        // fp32 output (the pretrain stack's master-weight gradients): one store instruction = ONE 1-KiB row segment (64 lanes x 16 B,
        // eight full 128-B lines) instead of two rows half-written per instruction (lanes 32 B apart, the other halves in the next
        // instruction): -4..7 % per launch at 128 experts x 512 rows, -3.5 % at 64 x 1024 (tools/wgrad_ab.sh, round 3)
        const int ec4 = (tid_e & 63) * 4, er8 = tid_e >> 6;
        const int ncol4 = cur.tc0 + ec4;
        f32x4 rw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rw[j] = *(const f32x4*)(stg + (er8 + 8 * j) * CT2_LD + ec4);
        if (pass == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ncol4 < p.NC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = er8 + 8 * j;
            const int trow = h * 128 + (r >> 5) * 64 + u * 32 + (r & 31);
            if (trow < nrows) {
              typedef __attribute__((address_space(1))) f32x4 gf32x4;
              gf32x4* dst = (gf32x4*)(Ce + ((int64_t)(cur.tr0 + trow) * p.ldc + ncol4) * 4);
              f32x4 a = rw[j];
              if (p.accumulate) a += dst[0];
#if CSMOE_WG_VARIANT == 2                 // timing experiment: no global stores at all (results are wrong): what the stores cost
              if (a[0] == 1.2345e30f) dst[0] = a;
#else
              WG_STORE(dst, a);
#endif
            }
          }
        }
      } else {
      f32x4 lo[4], hi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = er + 16 * j;
        lo[j] = *(const f32x4*)(stg + r * CT2_LD + ec);
        hi[j] = *(const f32x4*)(stg + r * CT2_LD + ec + 4);
      }
      if (pass == 3) {
        // next tile's K-tile 0 (issued four passes ago) must have landed before the closing barrier; taken before this
        // pass's stores so that it does not wait for them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (ncol < p.NC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = er + 16 * j;
          const int trow = h * 128 + (r >> 5) * 64 + u * 32 + (r & 31);
          if (trow < nrows) {
            const int64_t o = (int64_t)(cur.tr0 + trow) * p.ldc + ncol;
            if (p.out_f32) {
              typedef __attribute__((address_space(1))) f32x4 gf32x4;
              gf32x4* dst = (gf32x4*)(Ce + o * 4);
              f32x4 a = lo[j], b = hi[j];
              if (p.accumulate) { a += dst[0]; b += dst[1]; }
              WG_STORE(dst, a); WG_STORE(dst + 1, b);
            } else {
              typedef __attribute__((address_space(1))) bf16x8 gbf16x8;
              gbf16x8* dst = (gbf16x8*)(Ce + o * 2);
              float vv[8] = {lo[j][0], lo[j][1], lo[j][2], lo[j][3], hi[j][0], hi[j][1], hi[j][2], hi[j][3]};
              if (p.accumulate) {
                const bf16x8 old = *dst;
#pragma unroll
                for (int t = 0; t < 8; ++t) vv[t] += (float)old[t];
              }
              bf16x8 o8;
#pragma unroll
              for (int t = 0; t < 8; ++t) o8[t] = (bf16)vv[t];
              WG_STORE(dst, o8);
            }
          }
        }
      }
      }
      EPI_SYNC();
      STAMP(5 + pass);
    }
    cur = nxt;
    dcur = dnxt;
    first = false;
  }   // persistent tile loop
}

int persistent_grid(int64_t tiles_upper) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  const char* e = getenv("CSMOE_GEMM_PERSISTENT");
  if (e && atoi(e) == 0) return (int)tiles_upper;        // A/B: one workgroup per tile
  return (int)std::min<int64_t>(tiles_upper, ncu);
}

int set_lds2() {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)gg8w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  return CSMOE_OK;
}

}  // namespace

int gg8_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na, int Nb,
              void* const* c_ptrs, int64_t ldc, int out_dtype, int accumulate, int single_M, void* single_C, hipStream_t st,
              const int32_t* xcd_order) {
  FastArgs p{};
  p.single_M = single_M; p.single_C = single_C;
  p.R = A; p.ld_r = lda; p.Cflat = B; p.ld_c = ldb; p.offsets = offsets; p.E = E; p.NR = Na; p.NC = Nb;
  p.out_ptrs = c_ptrs; p.ldc = ldc; p.accumulate = accumulate; p.out_f32 = (out_dtype == CSMOE_F32);
  // band height of the tile order (tile_of): 5 for the experts of a grouped launch, 4 for one dense matrix.  Swept again with the
  // non-temporal stores in place (bands of 2 / 3 / 4 / 5 / 6 / 8, sum of the headline's two launches: 10.40 / 10.23 / 10.33 / 10.20 /
  // 10.19 / 10.48 ms; fp32 gradients 4 / 5 / 6: 12.15 / 11.80 / 11.84 ms at 64 experts, 15.35 / 14.72 / 15.1 at 128 x 512 rows; the
  // dense launches of the competition pass 2.148 / 2.178 / 2.166 ms).  CSMOE_WGRAD_BAND=<n> forces one height for both (A/B).
  static const int band_env = [] { const char* e = getenv("CSMOE_WGRAD_BAND"); return e ? atoi(e) : 0; }();
  p.tile_band = band_env > 0 ? band_env : (offsets ? 5 : 4);
  int64_t grid = (int64_t)E * ((Na + BM2 - 1) / BM2) * ((Nb + BN2 - 1) / BN2);
  // the dealt order needs the persistent grid to be a multiple of 8 (workgroup id % 8 = XCD) and enough experts to deal: every XCD
  // gets WHOLE experts, so 4 experts (one group of an expert-parallel rank's 8, ep.py) kept 4 of the 8 XCDs idle -- 2 x the time
  // per launch -- and 12 would give four XCDs twice the others' work.  Otherwise: one contiguous chunk of the tile order per XCD.
  const bool deal = offsets && xcd_order && (E % 8 == 0 || E >= 64);
  p.xcd_order = (deal && persistent_grid(grid) % 8 == 0 && persistent_grid(grid) < grid) ? xcd_order : nullptr;
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_wgrad: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  int rc;
  if ((rc = set_lds2())) return rc;
#ifdef CSMOE_STAMPS
  static unsigned long long* dbg = nullptr;
  if (!dbg) (void)hipMalloc(&dbg, 24 * 16 * 8);
  (void)hipMemsetAsync(dbg, 0, 24 * 16 * 8, st);
  p.aux = dbg;
#endif
  hipLaunchKernelGGL(gg8w_kernel, dim3((unsigned)persistent_grid(grid)), dim3(512), LDS2_BYTES, st, p);
#ifdef CSMOE_STAMPS
  {
    static unsigned long long h[24 * 16];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[9] = {"top->loop", "K-loop", "drain", "next-setup+K0", "pass0", "pass1", "pass2", "pass3", "to-next-top"};
    for (int t = 2; t < 24 && h[t * 16]; t += 7) {
      fprintf(stderr, "[stamps] tile %d:", t);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%lld", nm[k], (long long)(h[t * 16 + k + 1] - h[t * 16 + k]));
      if (h[(t + 1) * 16]) fprintf(stderr, " %s=%lld total=%lld", nm[8], (long long)(h[(t + 1) * 16] - h[t * 16 + 8]),
                                   (long long)(h[(t + 1) * 16] - h[t * 16]));
      fprintf(stderr, "\n");
    }
  }
#endif
  CSMOE_CHECK_LAUNCH("grouped_wgrad(bf16 v2)");
  return CSMOE_OK;
}
