// One-pass router for gfx950: x is read ONCE and leaves as logits (bf16), their fp32 softmax, the top-K expert ids with their
// renormalised weights and the per-block expert histogram of the token binning -- `self.gate(x)` + `topk_expert` + the counting
// pass of `compute_moe` (moe_model/model/moe/smoe.py:42-44, moe.py:113-132, 189-191; moe_pretrain_model/layers/moe/moe.py:121,
// deepseekv2.py:140-142) -- in one launch instead of gate GEMM, router_select and bin_hist.
//
// The gate product is HBM-bound (T x D bf16 in, T x E out, E <= 64): a 128-row GEMM tile with N = E leaves one workgroup per CU
// streaming 1 MB rows at 38 % of the HBM rate (profiles/r01).  Here a workgroup of 4 waves owns 64 token rows, 16 per wave:
//   * A fragments (16 rows x 32 k) go straight from global memory to the MFMA operand registers (lane l: row l%16, 16 bytes at
//     k = 8*(l/16)): every byte of x is used once, the loads of the next TWO 256-wide chunks are in flight (16 per lane, 16 KiB per wave, 128 KiB
//     per CU with two workgroups resident): one chunk ahead left every wave waiting out an HBM round trip per chunk (2.2 TB/s);
//   * the gate matrix is re-read by every workgroup (E x D, L2-resident): K-chunks of 256 columns are staged through LDS
//     (E rows of 528 bytes, double-buffered, filled by all 256 threads a chunk ahead) and read as B fragments by ds_read_b128;
//   * v_mfma_f32_16x16x32_bf16 over K in ascending order, fp32 accumulate, rounded to bf16 once (as the GEMM path does);
//   * each wave then selects its 16 rows, four at a time with 16 lanes per row, by the 16-lane form of the routine
//     router_select_kernel uses (router_select.h: same reduction trees): the fused and the two-launch path give identical bits;
//   * the expert ids of a workgroup's 64 rows are one block of the binning histogram (csmoe_bin_tokens_hist takes it from here).
#include "router_select.h"

namespace {

#ifndef CSMOE_RF_DIAG
#define CSMOE_RF_DIAG 0      // 1: no selection phase; 2: no gate staging (the first chunk's gate tile is reused for every chunk)
#endif
#ifndef CSMOE_RF_WAVES
#define CSMOE_RF_WAVES 4
#endif
constexpr int RF_WAVES = CSMOE_RF_WAVES;   // waves per workgroup, 16 token rows each
constexpr int RF_ROWS = 16 * RF_WAVES;     // token rows per workgroup
#ifndef CSMOE_RF_RING
#define CSMOE_RF_RING 3
#endif
constexpr int RF_RING = CSMOE_RF_RING;     // x chunks held per wave: one in use, RF_RING - 1 in flight
constexpr int RF_KCH = 256;           // K-chunk staged in LDS
constexpr int RF_LD = RF_KCH + 8;     // bf16 elements per staged gate row (528 B)

// NE = ceil(E / 16) column blocks.  SPEC: the selection rule and K are compile-time constants (softmax top-2 with the bf16-rounded K-sum:
// `smoe`'s, the headline's) -- the four row passes of a wave then are straight-line code the scheduler interleaves; with run-time
// `mode` / `K` every pass is a chain of ~600 dependent instructions behind uniform branches, two waves per SIMD to hide it
// (tools/router_diag.py: 21 us of selection behind a 52 us stream at E = 64).
template <int NE, bool SPEC = false>
__global__ void __launch_bounds__(64 * RF_WAVES) gate_select_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wg, int T, int D, int E,
                                                          int K_rt, int mode_rt, int round_sum_rt, float sel_param,
                                                          bf16* __restrict__ logits, float* __restrict__ softmax,
                                                          int32_t* __restrict__ idx, float* __restrict__ w,
                                                          int32_t* __restrict__ block_hist) {
  const int K = SPEC ? 2 : K_rt;
  const int mode = SPEC ? (int)CSMOE_SEL_SOFTMAX : mode_rt;
  const int round_sum_bf16 = SPEC ? 1 : round_sum_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* wbuf = (bf16*)smem;                                   // 2 x [NE*16][RF_LD]
  constexpr int WB = NE * 16 * RF_LD;                         // elements per buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i16 = lane & 15;
  const int row0 = blockIdx.x * RF_ROWS + wave * 16;
  const int my_row = min(row0 + i16, T - 1);                  // rows past the end re-read the last row; their results are dropped
  const bf16* xrow = x + (int64_t)my_row * D + 8 * g;

  // staging map: thread -> (gate row, 16-byte segment) of a chunk: 32 threads per row, 2 rows per wave and pass
  const int s_seg = tid & 31, s_row = tid >> 5;
  constexpr int RPP = 2 * RF_WAVES;                           // gate rows per pass
  constexpr int NPASS = NE * 16 / RPP;
  bf16x8 wreg[NPASS];
  bf16x8 a[RF_RING][8];                                       // ring: chunk ch in a[ch % RF_RING], the next RF_RING - 1 in flight
  f32x4 acc[NE];
#pragma unroll
  for (int c = 0; c < NE; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nch = (D + RF_KCH - 1) / RF_KCH;
  auto load_w = [&](int ch) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int e = p * RPP + s_row, k = ch * RF_KCH + s_seg * 8;
      wreg[p] = (e < E && k < D) ? *(const bf16x8*)(wg + (int64_t)e * D + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) *(bf16x8*)(wbuf + buf * WB + (p * RPP + s_row) * RF_LD + s_seg * 8) = wreg[p];
  };
  auto load_a = [&](bf16x8 (&av)[8], int ch) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = ch * RF_KCH + ks * 32;
      av[ks] = (k + 8 * g < D) ? *(const bf16x8*)(xrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  // one chunk: fetch the gate chunk ch+1 and the x chunk ch+2, multiply chunk ch, publish gate chunk ch+1
  auto step = [&](int ch, bf16x8 (&a_use)[8], bf16x8 (&a_fill)[8]) {
    const bool more = (CSMOE_RF_DIAG == 2) ? false : ch + 1 < nch;
    if (more) load_w(ch + 1);
    if (ch + RF_RING - 1 < nch) load_a(a_fill, ch + RF_RING - 1);
    const bf16* wb = wbuf + ((CSMOE_RF_DIAG == 2) ? 0 : (ch & 1)) * WB + i16 * RF_LD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int c = 0; c < NE; ++c) {
        const bf16x8 b = *(const bf16x8*)(wb + c * 16 * RF_LD + ks * 32);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_use[ks], b, acc[c], 0, 0, 0);
      }
    }
    if (more) store_w((ch + 1) & 1);                           // the other buffer: last read in iteration ch - 1, behind a barrier
    // LDS-only barrier: __syncthreads() also waits vmcnt(0), i.e. for the x chunk fetched two steps ahead -- it left ONE chunk in
    // flight per wave across every step boundary instead of two
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  load_w(0);
#pragma unroll
  for (int j = 0; j < RF_RING - 1; ++j)
    if (j < nch) load_a(a[j], j);
  store_w(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ch += RF_RING) {
#pragma unroll
    for (int j = 0; j < RF_RING; ++j)
      if (ch + j < nch) step(ch + j, a[j], a[(j + RF_RING - 1) % RF_RING]);
  }

#if CSMOE_RF_DIAG == 1      // timing experiment (wrong results): the stream + MFMA loop alone, no selection
  if (acc[0][0] == 1.2345e30f) logits[0] = (bf16)acc[0][0];
  return;
#endif
  // ---- logits of this wave's 16 rows -> LDS (fp32 value of the bf16-rounded logit), then row by row through the selection routine
  float* lrow = (float*)smem + wave * 16 * 65;                 // [16][65] per wave, over the gate buffers (all reads are done)
#pragma unroll
  for (int c = 0; c < NE; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) lrow[(4 * g + j) * 65 + c * 16 + i16] = (float)(bf16)acc[c][j];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int32_t* hist = (int32_t*)((float*)smem + RF_WAVES * 16 * 65);      // [64] behind the logit tiles
  if (block_hist) {
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
  }
  // four rows per pass, sixteen lanes per row (select_row_g16: the bits of the wave-per-row routine)
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int r = r4 * 4 + g;
    const int t = row0 + r;
    const bool live = t < T;
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = i16 + 16 * j;
      s[j] = e < E ? lrow[r * 65 + e] : -INFINITY;
      if (live && e < E) logits[(int64_t)t * E + e] = (bf16)s[j];
    }
    const int64_t tt = live ? t : 0;
    int myi[4];
    select_row_g16(s, i16, E, K, mode, round_sum_bf16, sel_param, CSMOE_BF16, live, softmax ? softmax + tt * E : nullptr,
                   idx + tt * K, w + tt * K, myi);
    if (block_hist && live) {
      // the ids this lane selected, from its registers: reading them back from `idx` put a store acknowledgement and a global load
      // (~2 us) into each of the four passes of every wave -- the tail in which the E = 64 router lost its 4.9 TB/s
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i16 + 16 * j < K) atomicAdd(&hist[myi[j]], 1);
    }
  }
  if (block_hist) {
    __syncthreads();
    if (tid < E) block_hist[(int64_t)blockIdx.x * E + tid] = hist[tid];
  }
}

}  // namespace

bool k_gate_select_ok(int T, int D, int E, int K, int dtype, const void* x, const void* wg) {
  return dtype == CSMOE_BF16 && E >= 1 && E <= 64 && K <= E && D % 8 == 0 && ((((uintptr_t)x | (uintptr_t)wg) & 15) == 0);
}

int k_gate_select_blocks(int T) { return (T + RF_ROWS - 1) / RF_ROWS; }
int k_gate_select_rows() { return RF_ROWS; }

int k_gate_select(const void* x, const void* wg, int T, int D, int E, int K, int mode, int round_sum_bf16, float sel_param,
                  void* logits, float* softmax, int32_t* idx, float* w, int32_t* block_hist, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  const int ne = (E + 15) / 16;
  const dim3 grid(k_gate_select_blocks(T)), block(64 * RF_WAVES);
  const size_t lds = std::max<size_t>((size_t)2 * ne * 16 * RF_LD * 2, (size_t)RF_WAVES * 16 * 65 * 4 + 256);
  const bool spec = K == 2 && mode == CSMOE_SEL_SOFTMAX && round_sum_bf16;
#define GS_LAUNCH(NE)                                                                                                    \
  if (spec)                                                                                                              \
    hipLaunchKernelGGL((gate_select_kernel<NE, true>), grid, block, lds, st, (const bf16*)x, (const bf16*)wg, T, D, E, K, mode,  \
                       round_sum_bf16, sel_param, (bf16*)logits, softmax, idx, w, block_hist);                           \
  else                                                                                                                   \
    hipLaunchKernelGGL((gate_select_kernel<NE, false>), grid, block, lds, st, (const bf16*)x, (const bf16*)wg, T, D, E, K, mode,  \
                       round_sum_bf16, sel_param, (bf16*)logits, softmax, idx, w, block_hist)
  switch (ne) {
    case 1: GS_LAUNCH(1); break;
    case 2: GS_LAUNCH(2); break;
    case 3: GS_LAUNCH(3); break;
    default: GS_LAUNCH(4); break;
  }
#undef GS_LAUNCH
  CSMOE_CHECK_LAUNCH("gate_select");
  return CSMOE_OK;
}
