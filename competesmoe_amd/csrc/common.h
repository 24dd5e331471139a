// Shared device/host helpers for the CompeteSMoE MI355X (gfx950) kernels.
// CDNA4 only: 64-wide wavefronts, MFMA, LDS-DMA.  No CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/csmoe.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CSMOE_WAVE 64

// ------------------------------------------------------------------ error plumbing (host)
void csmoe_set_error(const char* fmt, ...);
#define CSMOE_CHECK_ARG(cond, ...)                      \
  do {                                                  \
    if (!(cond)) {                                      \
      csmoe_set_error(__VA_ARGS__);                     \
      return CSMOE_ERR_INVALID;                         \
    }                                                   \
  } while (0)
#define CSMOE_CHECK_LAUNCH(name)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      csmoe_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return CSMOE_ERR_LAUNCH;                                                       \
    }                                                                                \
  } while (0)

// ------------------------------------------------------------------ dtype helpers (device)
template <typename T> struct DT;
template <> struct DT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct DT<bf16> {
  static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
  static __device__ __forceinline__ float rnd(float v) { return (float)(bf16)v; }
};

// ------------------------------------------------------------------ activations (fp32 math)
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case CSMOE_ACT_RELU: return x > 0.f ? x : 0.f;
    case CSMOE_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    case CSMOE_ACT_GELU_TANH: {
      const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
      float inner = k0 * (x + k1 * x * x * x);
      return 0.5f * x * (1.f + tanhf(inner));
    }
    case CSMOE_ACT_SILU: return x / (1.f + __expf(-x));
    case CSMOE_ACT_QUICK_GELU: return x / (1.f + __expf(-1.702f * x));
    default: return x;
  }
}
// d act(x) / dx
__device__ __forceinline__ float act_bwd(float x, int act) {
  switch (act) {
    case CSMOE_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case CSMOE_ACT_GELU: {
      float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case CSMOE_ACT_GELU_TANH: {
      const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
      float x2 = x * x;
      float inner = k0 * (x + k1 * x * x2);
      float t = tanhf(inner);
      float dinner = k0 * (1.f + 3.f * k1 * x2);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * dinner;
    }
    case CSMOE_ACT_SILU: {
      float s = 1.f / (1.f + __expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case CSMOE_ACT_QUICK_GELU: {
      float s = 1.f / (1.f + __expf(-1.702f * x));
      return s * (1.f + 1.702f * x * (1.f - s));
    }
    default: return 1.f;
  }
}

// quick-GELU as the reference computes it on x.dtype tensors, `input * torch.sigmoid(1.702 * input)` (transformers
// QuickGELUActivation): THREE elementwise ops, each rounding to x.dtype (the caller rounds the final product)
template <typename T>
__device__ __forceinline__ float quick_gelu_rounded(float x) {
  const float t = DT<T>::rnd(1.702f * x);
  const float s = DT<T>::rnd(1.f / (1.f + __expf(-t)));
  return x * s;
}
template <typename T>
__device__ __forceinline__ float quick_gelu_grad_rounded(float x) {
  const float t = DT<T>::rnd(1.702f * x);
  const float s = DT<T>::rnd(1.f / (1.f + __expf(-t)));
  return s + 1.702f * x * s * (1.f - s);
}

// ------------------------------------------------------------------ wave reductions (64 lanes)
// softplus with torch's threshold (x > 20 -> x) and its derivative.  precise: expf / log1pf as torch computes them; otherwise the
// hardware exp / log (the values are averaged over D right after).  Shared by the affinity kernels (moe_kernels.hip) and the
// affinity epilogues of the row-space GEMM (gemm_epilogue.h).
__device__ __forceinline__ float softplus_rt(float x, bool precise) {
  if (precise) return x > 20.f ? x : log1pf(expf(x));
  return x > 20.f ? x : __logf(1.f + __expf(x));
}
__device__ __forceinline__ float softplus_grad_rt(float x, bool precise) {
  if (precise) return 1.f / (1.f + expf(-x));
  return x > 20.f ? 1.f : __frcp_rn(1.f + __expf(-x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id (guide T1): blocks id and id+8 share an XCD, so give
// every XCD one contiguous chunk of the virtual tile order.  n = number of live tiles.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  int q = n >> 3, r = n & 7, x = id & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (id >> 3);
}

// Wave-parallel tile lookup for the grouped GEMMs: virtual tile v (after the XCD remap) -> expert, its row range and the
// tile's position.  Lane l handles expert l (+64, +128, ...): one batch of offset loads and a 6-step shuffle scan replace a
// serial walk over E dependent scalar loads per workgroup.  Tiles are ordered (expert, column tile, row tile), row tile fastest.
// Returns false when v is past the last tile.  All results are wave-uniform (broadcast from the owning lane).
#ifndef CSMOE_TILE_BAND
#define CSMOE_TILE_BAND 4
#endif
constexpr int TILE_BAND = CSMOE_TILE_BAND;      // row tiles per band of the row-space tile order (grouped_find_tile)
struct TilePos { int e, o0, o1, mt, nt; };
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// `part` selects which row tiles of every expert a launch covers: 0 all, 1 the first only, 2 all but the first (the fp32-master GEMM
// converts the weights in the launch over the first row tiles and runs the others from the bf16 copy that launch wrote)
__device__ __forceinline__ int part_tiles(int mt_e, int part) { return part == 1 ? min(mt_e, 1) : part == 2 ? max(mt_e - 1, 0) : mt_e; }
__device__ __forceinline__ int grouped_total_tiles(const int32_t* offsets, int E, int single_M, int BMt, int nct, int lane, int part = 0) {
  int total = 0;
  for (int base = 0; base < E; base += 64) {
    int e = base + lane;
    int cnt = 0;
    if (e < E) cnt = offsets ? (offsets[e + 1] - offsets[e]) : single_M;
    int tiles = part_tiles((cnt + BMt - 1) / BMt, part) * nct;
    tiles = wave_incl_scan(tiles, lane);
    total += __shfl(tiles, 63, 64);
  }
  return total;
}
// WHOLE > 0: an expert with at most WHOLE row tiles is ONE band (column tiles in order, all its row tiles under each) -- the
// one-wave-per-SIMD kernel's NT launches run 1.3 % faster that way at the headline's 4-5 row tiles per expert (tools/band_ab.sh: 5.06
// -> 4.99 ms with bands of 5, 6 or 8), the 8-wave kernel's NN launches 0.6 % slower, and tall experts want bands of 4 on both.
template <int WHOLE = 0>
__device__ __forceinline__ bool grouped_find_tile(const int32_t* offsets, int E, int single_M, int BMt, int nct, int v, int lane,
                                                  TilePos& out, int part = 0) {
  int acc = 0;
  for (int base = 0; base < E; base += 64) {
    int e = base + lane;
    int o0 = 0, o1 = 0;
    if (e < E) { o0 = offsets ? offsets[e] : 0; o1 = offsets ? offsets[e + 1] : single_M; }
    int mt_e = part_tiles((o1 - o0 + BMt - 1) / BMt, part);
    int incl = wave_incl_scan(mt_e * nct, lane) + acc;
    unsigned long long hit = __ballot(incl > v);
    if (hit) {
      int src = __ffsll((long long)hit) - 1;
      int excl = __shfl(incl - mt_e * nct, src, 64);
      int mte = __shfl(mt_e, src, 64);
      int local = v - excl;
      out.e = base + src;
      out.o0 = __shfl(o0, src, 64);
      out.o1 = __shfl(o1, src, 64);
      // row tiles in bands of TILE_BAND, column tiles inside a band, row tiles of the band fastest: the ~32 tiles an XCD's CUs hold at
      // a time are then a 4 x 8 block sharing 4 row panels and 8 weight panels.  With at most TILE_BAND row tiles per expert (the
      // headline: ~4) this IS the plain (n-tile, m-tile) order; with many (a dense always-on expert: 128 row tiles) the plain order
      // made 32 concurrent tiles share ONE weight panel and stream 32 row panels.
      const int bh = (WHOLE > 0 && mte <= WHOLE) ? max(mte, 1) : TILE_BAND;          // band height
      const int band = local / (bh * nct), rem = local - band * bh * nct;
      const int g_eff = min(bh, mte - band * bh);
      out.mt = band * bh + rem % g_eff + (part == 2 ? 1 : 0);
      out.nt = rem / g_eff;
      return true;
    }
    acc = __shfl(incl, 63, 64);
  }
  return false;
}
