// Grouped expert GEMM, bf16 in / fp32 accumulate, for gfx950 (MI355X).
//
// Structure (v1, "128x128 tile, one barrier per K-tile"):  256 threads = 4 waves in a 2x2 arrangement, each wave owns a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators; K-tile = 64; two LDS stages of (16 KiB + 16 KiB), filled by
// LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) one K-tile ahead of the MFMAs.  Buffer descriptors
// give hardware zero-fill for ragged expert row ranges and matrix edges, so tiles never read another expert's rows.
//
// LDS images (the DMA destination is lane-linear, so the swizzle lives in the per-lane SOURCE address and the same
// involution is applied on the read -- guide §5.4 rule 21):
//   KC  "K-contiguous"  128 rows x 64 k (128-B rows): 16-B chunk c of row r is stored at chunk c ^ ((r>>1)&7);
//       fragments by ds_read_b128, conflict-free for the 16-lane groups of that instruction.
//   KM  "K-major"       64 k-rows x 128 cols (256-B rows): 32-B segment s of k-row k at segment s ^ ((k&3)|((k>>3&1)<<2));
//       fragments by ds_read_b64_tr_b16 (hardware transpose, guide T10), conflict-free per 32-lane half.
//
// MFMA operand roles are swapped on purpose: the weight ("column") operand feeds the A input and the token ("row")
// operand the B input, so each lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one output row ->
// 8-byte (bf16) / 16-byte (fp32) row-major stores straight from registers.
//
// Replaces: cvmm_kernel / cvmm_backward_kernel3 (moe_pretrain_model/layers/cvmm.py:61-168, 194-345) and the per-expert
// nn.Linear loop of compute_moe (moe_model/model/moe/moe.py:196-204).
#include "gemm_tiles.h"
#include <algorithm>

using namespace ggt;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_B = 2 * TILE_B;
constexpr int CT_LD = BN + 4;               // fp32 epilogue tile row stride (floats)
constexpr int LDS_BYTES = BM * CT_LD * 4;   // 67,584 B >= the two K-loop stages (65,536 B)

template <int ROWK, int COLK, int MODE>
__global__ void __launch_bounds__(256, 2) gg_fast_kernel(FastArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---------------- tile lookup (wave-uniform scalars) ----------------
  int e, row0 = 0, rows = 0, tr0 = 0, tc0 = 0, red_len;
  const int nct = (p.NC + BN - 1) / BN;
  if (MODE == 0) {
    const int total = grouped_total_tiles(p.offsets, p.E, p.single_M, BM, nct, lane);
    if ((int)blockIdx.x >= total) return;
    const int v = xcd_remap(blockIdx.x, total);
    TilePos tp;
    if (!grouped_find_tile(p.offsets, p.E, p.single_M, BM, nct, v, lane, tp)) return;
    e = tp.e;
    row0 = tp.o0 + tp.mt * BM; rows = min(BM, tp.o1 - row0);
    tc0 = tp.nt * BN;
    red_len = p.Kd;
  } else {
    const int nrt = (p.NR + BM - 1) / BM;
    const int per_e = nrt * nct;
    int v = xcd_remap(blockIdx.x, per_e * p.E);
    e = v / per_e;
    int local = v - e * per_e;
    tr0 = (local / nct) * BM; tc0 = (local % nct) * BN;
    row0 = p.offsets ? p.offsets[e] : 0;
    red_len = (p.offsets ? p.offsets[e + 1] : p.single_M) - row0;
  }
  e = __builtin_amdgcn_readfirstlane(e);
  row0 = __builtin_amdgcn_readfirstlane(row0);
  rows = __builtin_amdgcn_readfirstlane(rows);
  tr0 = __builtin_amdgcn_readfirstlane(tr0);
  tc0 = __builtin_amdgcn_readfirstlane(tc0);
  red_len = __builtin_amdgcn_readfirstlane(red_len);

  // ---------------- operand descriptors ----------------
  const unsigned ldr_b = (unsigned)p.ld_r * 2u, ldc_b = (unsigned)p.ld_c * 2u;
  __amdgpu_buffer_rsrc_t rs_r, rs_c;
  unsigned vb_r[4], vb_c[4];
  int ax_r[4], ax_c[4];
  if (MODE == 0) {
    // rows of this expert's m-tile, [rows, Kd]
    rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)rows * ldr_b);
    dma_setup<KC, 4>(vb_r, ax_r, ldr_b, 0, 0, 7, 0, 0, wave, lane);
    const char* wb = (const char*)(p.c_ptrs_in ? p.c_ptrs_in[e] : p.single_B);
    if (COLK == KC) {   // weight [N, Kd]: tile rows = n
      int nrows = min(BN, p.NC - tc0);
      rs_c = make_rsrc(wb + (int64_t)tc0 * ldc_b, (unsigned)nrows * ldc_b);
      dma_setup<KC, 4>(vb_c, ax_c, ldc_b, 0, 0, 7, 0, 0, wave, lane);
    } else {            // weight [Kd, N]: tile rows = k, cols = n
      rs_c = make_rsrc(wb, (unsigned)p.Kd * ldc_b);
      dma_setup<KM, 4>(vb_c, ax_c, ldc_b, tc0, p.NC, 7, 0, 0, wave, lane);
    }
  } else {
    rs_r = make_rsrc((const char*)p.R + (int64_t)row0 * ldr_b, (unsigned)red_len * ldr_b);
    dma_setup<KM, 4>(vb_r, ax_r, ldr_b, tr0, p.NR, 7, 0, 0, wave, lane);
    rs_c = make_rsrc((const char*)p.Cflat + (int64_t)row0 * ldc_b, (unsigned)red_len * ldc_b);
    dma_setup<KM, 4>(vb_c, ax_c, ldc_b, tc0, p.NC, 7, 0, 0, wave, lane);
  }

  // ---------------- LDS read addressing ----------------
  const int g = lane >> 4, i16 = lane & 15;
  // KC: row (lane&15) of a 16-row block, chunk (g ^ (row>>1))
  const int kc_lane = i16 * 128 + ((g ^ (i16 >> 1)) << 4);
  // KM: k-row 8g+q, 8-byte piece p of the 32-B segment
  const int q = i16 >> 2, pp = i16 & 3;
  const int fk = q | ((g & 1) << 2);
  int km_r[4], km_c[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    km_r[cb] = (8 * g + q) * 256 + ((((wr * 4) + cb) ^ fk) << 5) + pp * 8;
    km_c[cb] = (8 * g + q) * 256 + ((((wc * 4) + cb) ^ fk) << 5) + pp * 8;
  }

  f32x4 acc[4][4];   // [col block][row block]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (red_len + BK - 1) / BK;
  if (nk > 0) {
    dma_tile<ROWK, 4>(rs_r, smem, vb_r, ax_r, 0, red_len, ldr_b, wave);
    dma_tile<COLK, 4>(rs_c, smem + TILE_B, vb_c, ax_c, 0, red_len, ldc_b, wave);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_B;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_B;
    if (kt + 1 < nk) {
      dma_tile<ROWK, 4>(rs_r, nxt, vb_r, ax_r, (kt + 1) * BK, red_len, ldr_b, wave);
      dma_tile<COLK, 4>(rs_c, nxt + TILE_B, vb_c, ax_c, (kt + 1) * BK, red_len, ldc_b, wave);
    }
    const char* tr = cur;
    const char* tc = cur + TILE_B;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fr[4], fc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        fr[b] = (ROWK == KC) ? frag_kc(tr, kc_lane, wr * 4 + b, s) : frag_km(tr, km_r[b], s);
        fc[b] = (COLK == KC) ? frag_kc(tc, kc_lane, wc * 4 + b, s) : frag_km(tc, km_c[b], s);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[cb], fr[rb], acc[cb][rb], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // D[row = 4*(lane>>4)+j -> output column][col = lane&15 -> output row]: each lane owns 4 consecutive output columns.
  // Stage the fp32 tile through LDS (row stride 132 floats: conflict-free b128 writes), then every thread handles
  // 8 consecutive columns of one row per step -> 16-byte coalesced global accesses, activation code emitted once.
  float* ct = (float*)smem;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = wr * 64 + rb * 16 + i16, n = wc * 64 + cb * 16 + 4 * g;
      *(f32x4*)(ct + m * CT_LD + n) = acc[cb][rb];
    }
  __syncthreads();
  const int ec = (threadIdx.x & 15) * 8;       // first of this thread's 8 columns inside the tile
  const int er = threadIdx.x >> 4;             // row inside a 16-row step
  const int ncol = tc0 + ec;
  if (ncol >= p.NC) return;
  if (MODE == 0) {
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool post_bias = p.epilogue == CSMOE_EPI_ROUND_BIAS32_ACT;   // fp32 bias added to the ROUNDED product (cvmm + bias)
    if (p.epilogue == CSMOE_EPI_BIAS || p.epilogue == CSMOE_EPI_BIAS_ACT || post_bias) {
      const void* bias = p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias;
      if (bias && post_bias) {
        const f32x4 b0 = *(const f32x4*)((const float*)bias + ncol), b1 = *(const f32x4*)((const float*)bias + ncol + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bv[j] = b0[j]; bv[4 + j] = b1[j]; }
      } else if (bias) {
        bf16x8 b8 = *(const bf16x8*)((const bf16*)bias + ncol);
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (float)b8[j];
      }
    }
#pragma unroll 1
    for (int r = er; r < rows; r += 16) {
      const f32x4 lo = *(const f32x4*)(ct + r * CT_LD + ec), hi = *(const f32x4*)(ct + r * CT_LD + ec + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const int64_t o = (int64_t)(row0 + r) * p.ldc + ncol;
      bf16x8 o0;
      if (p.epilogue == CSMOE_EPI_ACTGRAD || p.epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE) {
        const bf16x8 h8 = *(const bf16x8*)((const bf16*)p.aux + o);
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = (float)h8[j]; v[j] = (float)(bf16)v[j]; }
        if (p.epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE) {      // per-row scale (C2 slot) applied to the rounded product
          float* dot_tab = (float*)(p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias);
          if (dot_tab) {      // sum over this thread's 8 columns of product * aux -> FP32 table [M][NC / 8] (the small-shape kernel
            float d = 0.f;    // leaves the row reduction to csmoe_affinity_finish; the 256-tile kernel reduces to 128-column halves)
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fmaf(v[j], h[j], d);
            dot_tab[(int64_t)(row0 + r) * (p.NC >> 3) + (ncol >> 3)] = d;
          }
          const float sc = ((const float*)p.C2)[row0 + r];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (float)(bf16)(sc * v[j]);
        }
        act_bwd8(h, p.act);
#pragma unroll
        for (int j = 0; j < 8; ++j) o0[j] = (bf16)(v[j] * h[j]);
        *(bf16x8*)((bf16*)p.C + o) = o0;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (post_bias) { v[j] = (float)(bf16)v[j] + bv[j]; o0[j] = (bf16)v[j]; }
          else { o0[j] = (bf16)(v[j] + bv[j]); v[j] = (float)o0[j]; }
        }
        if (p.C) *(bf16x8*)((bf16*)p.C + o) = o0;
        if ((p.epilogue == CSMOE_EPI_BIAS_ACT || post_bias) && p.C2) {
          act_fwd8(v, p.act);
          bf16x8 o1;
#pragma unroll
          for (int j = 0; j < 8; ++j) o1[j] = (bf16)v[j];
          *(bf16x8*)((bf16*)p.C2 + o) = o1;
        }
      }
    }
  } else {
    char* Ce = (char*)(p.out_ptrs ? p.out_ptrs[e] : p.single_C);
    const int rlim = min(BM, p.NR - tr0);
#pragma unroll 1
    for (int r = er; r < rlim; r += 16) {
      f32x4 lo = *(const f32x4*)(ct + r * CT_LD + ec), hi = *(const f32x4*)(ct + r * CT_LD + ec + 4);
      const int64_t o = (int64_t)(tr0 + r) * p.ldc + ncol;
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
}

template <typename K>
int set_lds(K kern) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { csmoe_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
    done = true;
  }
  return CSMOE_OK;
}

}  // namespace

bool gg_fast_rowspace_ok(int64_t lda, int64_t ldb, int64_t ldc, int M, int N, int Kd, const void* A, const void* C) {
  if (N % 8 || Kd % 8 || lda % 8 || ldb % 8 || ldc % 8) return false;
  if (((uintptr_t)A & 15) || ((uintptr_t)C & 15)) return false;
  if ((int64_t)M * lda * 2 >= (int64_t)OOB) return false;
  if ((int64_t)std::max(N, Kd) * ldb * 2 >= (int64_t)OOB) return false;
  return true;
}

int gg_fast_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                     const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                     const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                     hipStream_t st) {
  FastArgs p{};
  p.single_M = M; p.single_B = single_B; p.single_bias = single_bias;
  p.R = A; p.ld_r = lda; p.c_ptrs_in = b_ptrs; p.ld_c = ldb; p.bias_ptrs = bias_ptrs; p.offsets = offsets; p.E = E;
  p.NC = N; p.Kd = Kd; p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  int nct = (N + BN - 1) / BN;
  int64_t grid = (int64_t)nct * ((M + BM - 1) / BM + E);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_gemm: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  int rc;
  if (b_layout == CSMOE_B_NK) {
    if ((rc = set_lds(gg_fast_kernel<KC, KC, 0>))) return rc;
    hipLaunchKernelGGL((gg_fast_kernel<KC, KC, 0>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
  } else {
    if ((rc = set_lds(gg_fast_kernel<KC, KM, 0>))) return rc;
    hipLaunchKernelGGL((gg_fast_kernel<KC, KM, 0>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
  }
  CSMOE_CHECK_LAUNCH("grouped_gemm(bf16)");
  return CSMOE_OK;
}

bool gg_fast_wgrad_ok(int64_t lda, int64_t ldb, int64_t ldc, int M, int Na, int Nb, const void* A, const void* B) {
  if (Na % 8 || Nb % 8 || lda % 8 || ldb % 8 || ldc % 8) return false;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return false;
  if ((int64_t)M * std::max(lda, ldb) * 2 >= (int64_t)OOB) return false;
  return true;
}

int gg_fast_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na, int Nb,
                  void* const* c_ptrs, int64_t ldc, int out_dtype, int accumulate, int single_M, void* single_C,
                  hipStream_t st) {
  FastArgs p{};
  p.single_M = single_M; p.single_C = single_C;
  p.R = A; p.ld_r = lda; p.Cflat = B; p.ld_c = ldb; p.offsets = offsets; p.E = E; p.NR = Na; p.NC = Nb;
  p.out_ptrs = c_ptrs; p.ldc = ldc; p.accumulate = accumulate; p.out_f32 = (out_dtype == CSMOE_F32);
  int64_t grid = (int64_t)E * ((Na + BM - 1) / BM) * ((Nb + BN - 1) / BN);
  if (grid <= 0) return CSMOE_OK;
  if (grid > 0x7fffffff) { csmoe_set_error("grouped_wgrad: grid too large"); return CSMOE_ERR_UNSUPPORTED; }
  int rc;
  if ((rc = set_lds(gg_fast_kernel<KM, KM, 1>))) return rc;
  hipLaunchKernelGGL((gg_fast_kernel<KM, KM, 1>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
  CSMOE_CHECK_LAUNCH("grouped_wgrad(bf16)");
  return CSMOE_OK;
}
