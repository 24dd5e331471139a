// Epilogue of the 256x256 row-space grouped-GEMM kernels (gemm_bf16_v2.hip, gemm_bf16_v2c.hip, gemm_fp8.hip): the 8 waves' fp32
// accumulators (wave (wm, wn): acc[cb][rb] = rows (rb>>2)*128 + wm*64 + (rb&3)*16 + lane%16, columns (cb>>1)*128 + wn*32 +
// (cb&1)*16 + 4*(lane/16) + 0..3) -> bf16 rows in global memory with the bias / activation / activation-gradient variants of
// include/csmoe.h.
//
// Per-workgroup stamps of the previous form (tools/tile_stamps.py, -DCSMOE_STAMPS; headline shape, balanced routing) priced the
// epilogue at 12.5 us of a 300 us tile with a plain store and 24-27 us of a 134 us tile with the GELU variants -- and the SAME with
// the global stores compiled out: the time was the instruction stream (run-time `epilogue` / `act` tests per element, fp32 staging
// in two passes of 128 rows, four barriers).  This form
//   * rounds to bf16 in registers and stages the whole 256-row tile ONCE (ds_write_b64, 132 KiB incl. row padding, one barrier):
//     every variant of the interface applies its activation to the ROUNDED product, so nothing is lost by rounding first;
//   * is instantiated per (variant, activation) with the choice made once per tile, so each row loop is straight-line code;
//   * walks 16 rows per thread fully unrolled (row = t/32 + 16 i, 8 columns per thread: 512-byte row segments per half wave);
//   * fetches the saved pre-activations of the activation-gradient variant before the staging, 16 loads in flight per thread.
#pragma once
#include "gemm_tiles.h"

namespace ggt {

struct EpiArgs {
  void* C; void* C2; const void* aux; const void* bias;   // bias: this expert's row (bf16, or fp32 for ROUND_BIAS32_ACT) or null
  int64_t ldc; int epilogue, act, NC;
};

constexpr int EPI_LD = 264;                         // staged row stride in bf16 elements (528 B: 16-byte aligned, 4 banks per row)
constexpr int EPI_LDS_BYTES = 256 * EPI_LD * 2;     // 135,168 B

// How a kernel's waves hold the 256x256 accumulator tile (lane = 16 g + i16; every f32x4 = 4 consecutive columns of one row):
//   Lay8: 8 waves as 2 row halves x 4 column quarters, acc[4][8], 64x32 blocks interleaved over the four operand images
//         (gemm_bf16_v2.hip, gemm_bf16_v2c.hip, gemm_fp8.hip);
//   Lay4: 4 waves as 2 x 2, one wave per SIMD, acc[8][8] = a contiguous 128x128 quadrant (gemm_bf16_v4.hip).
struct Lay8 {
  static constexpr int THREADS = 512, NCB = 4, NRB = 8;
  static __device__ __forceinline__ int m(int wm, int rb, int i16) { return (rb >> 2) * 128 + wm * 64 + (rb & 3) * 16 + i16; }
  static __device__ __forceinline__ int n(int wn, int cb, int g) { return (cb >> 1) * 128 + wn * 32 + (cb & 1) * 16 + 4 * g; }
};
struct Lay4 {
  static constexpr int THREADS = 256, NCB = 8, NRB = 8;
  static __device__ __forceinline__ int m(int wm, int rb, int i16) { return wm * 128 + rb * 16 + i16; }
  static __device__ __forceinline__ int n(int wn, int cb, int g) { return wn * 128 + cb * 16 + 4 * g; }
};

enum { EC_PLAIN = 0, EC_BIAS = 1, EC_BIAS_ACT = 2, EC_R32_ACT = 3, EC_ACTGRAD = 4, EC_ACTGRAD_RS = 5 };   // _RS: + per-row scale (C2 slot)
constexpr int ACT_RT = -1;                          // activation chosen at run time (the rarer ones share one instantiation)

template <int ACT>
__device__ __forceinline__ void epi_act_fwd8(float (&v)[8], int act_rt) {
  if constexpr (ACT == ACT_RT) act_fwd8(v, act_rt); else act_fwd8(v, ACT);
}
template <int ACT>
__device__ __forceinline__ void epi_act_bwd8(float (&h)[8], int act_rt) {
  if constexpr (ACT == ACT_RT) act_bwd8(h, act_rt); else act_bwd8(h, ACT);
}

template <int CLS, int ACT, typename LAY = Lay8>
__device__ __forceinline__ void epi_run(const EpiArgs& p, const f32x4 (&acc)[LAY::NCB][LAY::NRB], char* smem, int row0, int rows, int tc0,
                                        int wm, int wn, int lane) {
  constexpr int RSTEP = LAY::THREADS / 32;     // rows one pass of the workgroup covers (32 threads x 8 columns per row)
  constexpr int RPT = 256 / RSTEP;             // rows per thread: 16 (512 threads) / 32 (256 threads)
  bf16* stg = (bf16*)smem;
  const int g = lane >> 4, i16 = lane & 15;
  const int ec = (threadIdx.x & 31) * 8;       // this thread's 8 columns inside the 256-wide tile
  const int er = threadIdx.x >> 5;             // 0..RSTEP-1
  const int ncol = tc0 + ec;
  const bool col_ok = ncol < p.NC;

  // rows go through in chunks of 16 per thread (one chunk for the 512-thread layout, two for the 256-thread one: a 32-row
  // straight-line body times the 12 (variant, activation) instantiations was 1.3 MB of code and minutes of compile time)
  constexpr int CH = 16, NCH = RPT / CH;
  bf16x8 hq[CH], hq_next[CH];
  float rs[CH], rs_next[CH];                   // EC_ACTGRAD_RS: this thread's row scales, fetched with the pre-activations
  auto fetch_aux = [&](int h, bf16x8 (&q)[CH], float (&sc)[CH]) {
    if constexpr (CLS == EC_ACTGRAD || CLS == EC_ACTGRAD_RS) {
      if (col_ok) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int r = er + RSTEP * (h * CH + i);
          if (r < rows) q[i] = *(const bf16x8*)((const bf16*)p.aux + (int64_t)(row0 + r) * p.ldc + ncol);
        }
      }
    }
    if constexpr (CLS == EC_ACTGRAD_RS) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int r = er + RSTEP * (h * CH + i);
        sc[i] = r < rows ? ((const float*)p.C2)[row0 + r] : 0.f;
      }
    }
  };
  fetch_aux(0, hq, rs);                        // before the staging: 16 loads in flight per thread while the tile goes to LDS
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (CLS == EC_R32_ACT) {           // fp32 bias added to the ROUNDED product (cvmm + bias)
    if (p.bias && col_ok) {
      const f32x4 b0 = *(const f32x4*)((const float*)p.bias + ncol), b1 = *(const f32x4*)((const float*)p.bias + ncol + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { bv[j] = b0[j]; bv[4 + j] = b1[j]; }
    }
  }

  // ---- stage: accumulators (+ bf16 bias, added before the rounding as F.linear does) -> bf16 -> LDS
#pragma unroll
  for (int cb = 0; cb < LAY::NCB; ++cb) {
    const int n = LAY::n(wn, cb, g);
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CLS == EC_BIAS || CLS == EC_BIAS_ACT) {
      if (p.bias && tc0 + n < p.NC) {
        const bf16x4 b4 = *(const bf16x4*)((const bf16*)p.bias + tc0 + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (float)b4[j];
      }
    }
#pragma unroll
    for (int rb = 0; rb < LAY::NRB; ++rb) {
      const int m = LAY::m(wm, rb, i16);
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (CLS == EC_BIAS || CLS == EC_BIAS_ACT) o[j] = (bf16)(acc[cb][rb][j] + b[j]);
        else o[j] = (bf16)acc[cb][rb][j];
      }
      *(bf16x4*)(stg + m * EPI_LD + n) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS traffic only: __syncthreads() would also wait for the loads above
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- rows
  // EC_ACTGRAD_RS with a dot table (EpiArgs::bias slot, FP32 [rows of the launch][ceil(NC / 128)]): per row and 128-column half,
  // sum_n product[m,n] * aux[m,n] -- the gradient of the reduction weight as the reference forms it, `grad_x_full @ x`
  // (cvmm.py:544) -- reduced over the 16 threads of the half by shuffles: every thread stays for them
  float* const dot_tab = (CLS == EC_ACTGRAD_RS) ? (float*)p.bias : nullptr;
  if (!col_ok && dot_tab == nullptr) return;
#pragma unroll 1
  for (int hc = 0; hc < NCH; ++hc) {
  if constexpr (NCH > 1) {                      // the next chunk's pre-activations travel while this chunk is processed
    if (hc > 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) { hq[i] = hq_next[i]; rs[i] = rs_next[i]; }
    }
    if (hc + 1 < NCH) fetch_aux(hc + 1, hq_next, rs_next);
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int r = er + RSTEP * (hc * CH + i);
    if (r < rows) {
      bf16x8 o0 = *(const bf16x8*)(stg + r * EPI_LD + ec);
      const int64_t o = (int64_t)(row0 + r) * p.ldc + ncol;
      if constexpr (CLS == EC_PLAIN || CLS == EC_BIAS) {
        *(bf16x8*)((bf16*)p.C + o) = o0;
      } else if constexpr (CLS == EC_ACTGRAD || CLS == EC_ACTGRAD_RS) {
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = col_ok ? (float)hq[i][j] : 0.f;
        if constexpr (CLS == EC_ACTGRAD_RS) {
          if (dot_tab) {                           // wave-uniform
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fmaf(col_ok ? (float)o0[j] : 0.f, h[j], d);
#pragma unroll
            for (int sh = 8; sh > 0; sh >>= 1) d += __shfl_xor(d, sh, 16);
            const int half = (threadIdx.x & 31) >> 4;             // a half wholly past the last column has no table entry
            if ((threadIdx.x & 15) == 0 && tc0 + half * 128 < p.NC)
              dot_tab[(int64_t)(row0 + r) * ((p.NC + 127) >> 7) + (tc0 >> 7) + half] = d;
          }
        }
        epi_act_bwd8<ACT>(h, p.act);
        bf16x8 o1;
        if constexpr (CLS == EC_ACTGRAD_RS) {      // the reduction weight multiplies the ROUNDED product (cvmm.py:527-543)
          const float sc = rs[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) o1[j] = (bf16)((float)(bf16)(sc * (float)o0[j]) * h[j]);
          if (!col_ok) continue;                   // stayed for the shuffles only
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o1[j] = (bf16)((float)o0[j] * h[j]);
        }
        *(bf16x8*)((bf16*)p.C + o) = o1;
      } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)o0[j];
        if constexpr (CLS == EC_R32_ACT) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { v[j] += bv[j]; o0[j] = (bf16)v[j]; }     // the activation sees the fp32 sum
        }
        if (p.C) *(bf16x8*)((bf16*)p.C + o) = o0;            // null: the caller keeps the activated output only (ReLU)
        if (p.C2) {
          epi_act_fwd8<ACT>(v, p.act);
          bf16x8 o1;
#pragma unroll
          for (int j = 0; j < 8; ++j) o1[j] = (bf16)v[j];
          *(bf16x8*)((bf16*)p.C2 + o) = o1;
        }
      }
    }
  }
  }
}

// The two affinity epilogues of the competition pass: y = round(acc + bias) is staged like every other variant, then
//   GRAD = false: each row's 256 staged columns are reduced to sum softplus(y) (8 per thread, then the 32 threads of the row by
//                 shuffles: a fixed order) into the FP32 table C[row][column tile];
//   GRAD = true : C[row][col] = round(g * sigmoid(y)) with g = aux[row] / N (aux: FP32 d aff, one per row), g rounded to bf16
//                 first under flag bit 1.
// flags (EpiArgs::act): bit 0 precise exp / log1p, bit 1 round every softplus / product to bf16 first (x.dtype tensor ops).
template <bool GRAD, typename LAY = Lay8>
__device__ __forceinline__ void epi_softplus(const EpiArgs& p, const f32x4 (&acc)[LAY::NCB][LAY::NRB], char* smem, int row0, int rows, int tc0,
                                             int wm, int wn, int lane) {
  constexpr int RSTEP = LAY::THREADS / 32, RPT = 256 / RSTEP;
  bf16* stg = (bf16*)smem;
  const int g = lane >> 4, i16 = lane & 15;
  const int ec = (threadIdx.x & 31) * 8, er = threadIdx.x >> 5;
  const int ncol = tc0 + ec;
  const bool col_ok = ncol < p.NC;
  const bool precise = p.act & 1, rnd = p.act & 2;
#pragma unroll
  for (int cb = 0; cb < LAY::NCB; ++cb) {
    const int n = LAY::n(wn, cb, g);
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && tc0 + n < p.NC) {
      const bf16x4 b4 = *(const bf16x4*)((const bf16*)p.bias + tc0 + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = (float)b4[j];
    }
#pragma unroll
    for (int rb = 0; rb < LAY::NRB; ++rb) {
      const int m = LAY::m(wm, rb, i16);
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = p.bias ? (bf16)(acc[cb][rb][j] + b[j]) : (bf16)acc[cb][rb][j];
      *(bf16x4*)(stg + m * EPI_LD + n) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int nt = tc0 >> 8;
#pragma unroll 4
  for (int i = 0; i < RPT; ++i) {
    const int r = er + RSTEP * i;                   // the two rows of a wave go through the loop together (shuffles below)
    const bf16x8 o0 = *(const bf16x8*)(stg + r * EPI_LD + ec);
    if constexpr (!GRAD) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sp = softplus_rt((float)o0[j], precise);
        s += rnd ? (float)(bf16)sp : sp;
      }
      if (!col_ok) s = 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
      if ((threadIdx.x & 31) == 0 && r < rows) ((float*)p.C)[(int64_t)(row0 + r) * p.ldc + nt] = s;
    } else {
      if (col_ok && r < rows) {
        const float g0 = ((const float*)p.aux)[row0 + r] / (float)p.NC;          // d aff / D, as softplus_mean_bwd_kernel forms it
        const float sc = rnd ? (float)(bf16)g0 : g0;
        bf16x8 o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) o1[j] = (bf16)(sc * softplus_grad_rt((float)o0[j], precise));
        *(bf16x8*)((bf16*)p.C + (int64_t)(row0 + r) * p.ldc + ncol) = o1;
      }
    }
  }
}

template <int CLS, typename LAY = Lay8>
__device__ __forceinline__ void epi_by_act(const EpiArgs& p, const f32x4 (&acc)[LAY::NCB][LAY::NRB], char* smem, int row0, int rows, int tc0,
                                           int wm, int wn, int lane) {
  if (p.act == CSMOE_ACT_GELU) epi_run<CLS, CSMOE_ACT_GELU, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane);
  else if (p.act == CSMOE_ACT_RELU) epi_run<CLS, CSMOE_ACT_RELU, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane);
  else epi_run<CLS, ACT_RT, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane);
}

// Called by all of the workgroup's threads after the K-loop's LDS traffic has drained (the staging tile overlays the operand images).
template <typename LAY = Lay8>
__device__ __forceinline__ void rowspace_epilogue(const EpiArgs& p, const f32x4 (&acc)[LAY::NCB][LAY::NRB], char* smem, int row0, int rows,
                                                  int tc0, int wm, int wn, int lane) {
  switch (p.epilogue) {
    case CSMOE_EPI_ACTGRAD: epi_by_act<EC_ACTGRAD, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_ACTGRAD_ROWSCALE: epi_by_act<EC_ACTGRAD_RS, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_SOFTPLUS_ROWSUM: epi_softplus<false, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_SOFTPLUS_GRAD: epi_softplus<true, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_ROUND_BIAS32_ACT: epi_by_act<EC_R32_ACT, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_BIAS_ACT: epi_by_act<EC_BIAS_ACT, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
    case CSMOE_EPI_BIAS:
      if (p.bias) { epi_run<EC_BIAS, 0, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break; }
      [[fallthrough]];
    default: epi_run<EC_PLAIN, 0, LAY>(p, acc, smem, row0, rows, tc0, wm, wn, lane); break;
  }
}

}  // namespace ggt
