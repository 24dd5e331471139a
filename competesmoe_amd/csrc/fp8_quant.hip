// MXFP8 quantisation for the block-scaled MFMA path (BASELINE config 5): OCP e4m3 elements with one e8m0 scale per 32 elements
// ALONG THE REDUCTION DIMENSION of the GEMM that will consume the tensor (OCP Microscaling spec: shared exponent =
// floor(log2(amax)) - emax_elem, emax(e4m3) = 8; elements = round-to-nearest-even(x * 2^-shared), saturated at +-448).
// The reference has no fp8: nothing upstream to cite; the oracle restates this file in torch.
//   rows:       x [R, C] -> q [R, C] u8, s [R, C/32] u8          (blocks along the contiguous dim: activations, K-contiguous weights)
//   transposed: x [R, C] -> q [C, R] u8, s [C, R/32] u8          (blocks along dim 0: the same weights for the transposed product)
// Inputs bf16 or fp32 (the pretrain stack's fp32 master weights are quantised directly: no bf16 cast pass).
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

__device__ __forceinline__ int e8m0_of_amax(float amax) {
  // biased exponent byte of 2^(floor(log2(amax)) - 8); amax == 0 (or subnormal) -> the smallest scale
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xff);      // floor(log2(amax)) + 127 for normal amax
  const int s = be - 8;
  return s < 0 ? 0 : (s > 254 ? 254 : s);
}
__device__ __forceinline__ float inv_scale_of(int sbyte) {         // 2^-(sbyte - 127), exact
  const int e = 254 - sbyte;                                       // biased exponent of the reciprocal
  return e <= 0 ? __uint_as_float(0x00400000u >> (-e)) : __uint_as_float((unsigned)e << 23);
}
// saturate to +-448 but let a NaN through (fminf / fmaxf return their non-NaN operand, which would turn NaN into -448): the
// conversion below encodes it as the e4m3 NaN, so a NaN activation still poisons its output row as it does on the bf16 path
__device__ __forceinline__ float sat448(float v) { return v != v ? v : fminf(fmaxf(v, -448.f), 448.f); }
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  a = sat448(a); b = sat448(b); c = sat448(c); d = sat448(d);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

template <typename T> __device__ __forceinline__ void load32(const T* p, float (&v)[32]);
template <> __device__ __forceinline__ void load32<bf16>(const bf16* p, float (&v)[32]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 x = *(const bf16x8*)(p + 8 * c);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[8 * c + j] = (float)x[j];
  }
}
template <> __device__ __forceinline__ void load32<float>(const float* p, float (&v)[32]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 x = *(const f32x4*)(p + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * c + j] = x[j];
  }
}

// one lane per 32-element block; blocks of a row are consecutive lanes.  grid.y = matrix (expert) index
template <typename T>
__global__ void __launch_bounds__(256) quant_rows_kernel(const void* const* x_ptrs, const T* x_single, int64_t ldx, int R, int C,
                                                         uint8_t* q, uint8_t* s, int64_t q_mat, int64_t s_mat) {
  const int e = blockIdx.y;
  const T* x = x_ptrs ? (const T*)x_ptrs[e] : x_single;
  uint8_t* qe = q + (int64_t)e * q_mat;
  uint8_t* se = s + (int64_t)e * s_mat;
  const int nb = C >> 5;
  const int64_t total = (int64_t)R * nb;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / nb), b = (int)(i - (int64_t)r * nb);
    float v[32];
    load32<T>(x + (int64_t)r * ldx + b * 32, v);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
    const int sb = e8m0_of_amax(amax);
    const float inv = inv_scale_of(sb);
    u32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0[j] = pack4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
      o1[j] = pack4_e4m3(v[16 + 4 * j] * inv, v[17 + 4 * j] * inv, v[18 + 4 * j] * inv, v[19 + 4 * j] * inv);
    }
    uint8_t* dst = qe + (int64_t)r * C + b * 32;
    *(u32x4*)dst = o0;
    *(u32x4*)(dst + 16) = o1;
    se[(int64_t)r * nb + b] = (uint8_t)sb;
  }
}

// transposed: one wave per 32-row x 64-column tile; lane = column, the 32 rows of the block sit in the lane's registers
template <typename T>
__global__ void __launch_bounds__(256) quant_cols_kernel(const void* const* x_ptrs, const T* x_single, int64_t ldx, int R, int C,
                                                         uint8_t* q, uint8_t* s, int64_t q_mat, int64_t s_mat) {
  const int e = blockIdx.y;
  const T* x = x_ptrs ? (const T*)x_ptrs[e] : x_single;
  uint8_t* qe = q + (int64_t)e * q_mat;       // [C, R]
  uint8_t* se = s + (int64_t)e * s_mat;       // [C, R/32]
  const int lane = threadIdx.x & 63;
  const int rb_n = R >> 5, cb_n = (C + 63) >> 6;
  const int64_t tiles = (int64_t)rb_n * cb_n;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += (int64_t)gridDim.x * 4) {
    const int rb = (int)(t / cb_n), cb = (int)(t - (int64_t)rb * cb_n);
    const int c = cb * 64 + lane;
    if (c >= C) continue;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      v[i] = (float)x[(int64_t)(rb * 32 + i) * ldx + c];
      amax = fmaxf(amax, fabsf(v[i]));
    }
    const int sb = e8m0_of_amax(amax);
    const float inv = inv_scale_of(sb);
    u32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0[j] = pack4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
      o1[j] = pack4_e4m3(v[16 + 4 * j] * inv, v[17 + 4 * j] * inv, v[18 + 4 * j] * inv, v[19 + 4 * j] * inv);
    }
    uint8_t* dst = qe + (int64_t)c * R + rb * 32;
    *(u32x4*)dst = o0;
    *(u32x4*)(dst + 16) = o1;
    se[(int64_t)c * rb_n + rb] = (uint8_t)sb;
  }
}

// both orientations in ONE pass over the source (weights: the forward product needs blocks along one dim, the backward product
// along the other; reading 4-byte masters twice was the larger half of the fp8 step's quantisation time).  One 256-thread
// workgroup per 64-row x 128-column tile: 16-byte global loads (512 B per row and wave-instruction, 8 in flight per thread) into an
// fp32 LDS tile (33 KiB: four workgroups per CU), then the 256 (row, 32-column block) units and the 256 (column, 32-row block) units
// are quantised one each per thread straight from LDS -- rows by ds_read_b128 along the row, columns by ds_read_b32 down the
// column (consecutive lanes = consecutive banks) -- and every output leaves as 16-byte stores; two threads complete a 64-byte run
// of a transposed row.  6 B of HBM traffic per element with fp32 masters (4 read + 2 written).
// Tile height, same box (gpurun_out r2x / r3n): 32 rows (lone 32-byte pieces of the transposed rows, 8 workgroups per CU) 4.50 ms per
// call = 3.9 TB/s; 64 rows 4.18 ms = 4.2 TB/s; 128 rows (66 KiB of LDS, 2 workgroups per CU: too few to keep HBM busy while a
// workgroup's load and quantise phases do not overlap) 5.49 ms.
constexpr int QB_R = 64, QB_C = 128, QB_LD = QB_C + 4;       // +4 floats: rows 16 B apart in bank space (b128 row reads)

template <typename T>
__global__ void __launch_bounds__(256) quant_both_kernel(const void* const* x_ptrs, const T* x_single, int64_t ldx, int R, int C,
                                                         uint8_t* q, uint8_t* s, uint8_t* qt, uint8_t* st, int64_t q_mat,
                                                         int64_t s_mat, int64_t st_mat) {
  __shared__ __attribute__((aligned(16))) float tile[QB_R * QB_LD];
  const int e = blockIdx.y;
  const T* x = x_ptrs ? (const T*)x_ptrs[e] : x_single;
  uint8_t* qe = q + (int64_t)e * q_mat;         // [R, C]
  uint8_t* se = s + (int64_t)e * s_mat;         // [R, C/32]
  uint8_t* qte = qt + (int64_t)e * q_mat;       // [C, R]
  uint8_t* ste = st + (int64_t)e * st_mat;      // [C, R/32]
  const int t = threadIdx.x;
  const int rt_n = (R + QB_R - 1) / QB_R, cb_n = (C + QB_C - 1) / QB_C, nbc = C >> 5, nbr = R >> 5;
  const int64_t tiles = (int64_t)rt_n * cb_n;
  // A tile's eight 16-byte loads per thread are all issued into registers before the first LDS write (round 3: 4.2 -> 4.5 TB/s; written
  // as load-then-store per row, hipcc waited for each load in turn).  When a workgroup has more than one tile (grids beyond 16 384
  // tiles per expert) the next tile's loads are issued before the current one is quantised.
  f32x4 nxt[8];
  auto fetch = [&](int64_t ti) {
    const int rt = (int)(ti / cb_n), cb = (int)(ti - (int64_t)rt * cb_n);
    const int r0 = rt * QB_R, c0 = cb * QB_C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (t >> 5) + 8 * i, c = (t & 31) * 4;      // thread -> (row t / 32 + 8 i, 4 columns (t % 32) * 4)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + c < C && r0 + r < R) {
        const T* src = x + (int64_t)(r0 + r) * ldx + c0 + c;
        if constexpr (std::is_same<T, float>::value) {
          v = *(const f32x4*)src;
        } else {
          const bf16x4 h = *(const bf16x4*)src;
          v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
      }
      nxt[i] = v;
    }
  };
  if ((int64_t)blockIdx.x < tiles) fetch(blockIdx.x);
  for (int64_t ti = blockIdx.x; ti < tiles; ti += gridDim.x) {
    const int rt = (int)(ti / cb_n), cb = (int)(ti - (int64_t)rt * cb_n);
    const int r0 = rt * QB_R, c0 = cb * QB_C;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)(tile + ((t >> 5) + 8 * i) * QB_LD + (t & 31) * 4) = nxt[i];
    __syncthreads();
    if (ti + gridDim.x < tiles) fetch(ti + gridDim.x);
    {
      // ---- row-major output: unit (row t / 4, column block t % 4)
      const int r = t >> 2, bk = t & 3;
      if (c0 + bk * 32 < C && r0 + r < R) {
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 w4 = *(const f32x4*)(tile + r * QB_LD + bk * 32 + 4 * j);
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[4 * j + k] = w4[k]; amax = fmaxf(amax, fabsf(w4[k])); }
        }
        const int sb = e8m0_of_amax(amax);
        const float inv = inv_scale_of(sb);
        u32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o0[j] = pack4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
          o1[j] = pack4_e4m3(v[16 + 4 * j] * inv, v[17 + 4 * j] * inv, v[18 + 4 * j] * inv, v[19 + 4 * j] * inv);
        }
        uint8_t* dst = qe + (int64_t)(r0 + r) * C + c0 + bk * 32;
        *(u32x4*)dst = o0;
        *(u32x4*)(dst + 16) = o1;
        se[(int64_t)(r0 + r) * nbc + (c0 >> 5) + bk] = (uint8_t)sb;
      }
    }
    {
      // ---- transposed output: column t % 128, row block t / 128 (two threads complete a 64-byte run of the column)
      const int c = t & 127, rbk = t >> 7;
      if (c0 + c < C && r0 + rbk * 32 < R) {
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { v[i] = tile[(rbk * 32 + i) * QB_LD + c]; amax = fmaxf(amax, fabsf(v[i])); }
        const int sb = e8m0_of_amax(amax);
        const float inv = inv_scale_of(sb);
        u32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o0[j] = pack4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
          o1[j] = pack4_e4m3(v[16 + 4 * j] * inv, v[17 + 4 * j] * inv, v[18 + 4 * j] * inv, v[19 + 4 * j] * inv);
        }
        uint8_t* dst = qte + (int64_t)(c0 + c) * R + r0 + rbk * 32;
        *(u32x4*)dst = o0;
        *(u32x4*)(dst + 16) = o1;
        ste[(int64_t)(c0 + c) * nbr + (r0 >> 5) + rbk] = (uint8_t)sb;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int k_quantize_mxfp8_both(const void* const* x_ptrs, const void* x_single, int E, int64_t ldx, int R, int C, int in_dtype, void* q,
                          void* s, void* qt, void* st, hipStream_t stream) {
  if (E <= 0 || R <= 0 || C <= 0) return CSMOE_OK;
  const int64_t tiles = (int64_t)((R + QB_R - 1) / QB_R) * ((C + QB_C - 1) / QB_C);
  // one tile per workgroup where the grid allows (16 384 per expert): a persistent grid walking several tiles per workgroup with
  // the next tile prefetched measured SLOWER (2 048 workgroups 4.86 ms, 65 536 4.33 ms, one per tile 3.92 ms per 128-expert table)
  dim3 grid((unsigned)std::min<int64_t>(tiles, 16384), (unsigned)E), block(256);
  const int64_t q_mat = (int64_t)R * C, s_mat = (int64_t)R * (C / 32), st_mat = (int64_t)C * (R / 32);
  if (in_dtype == CSMOE_BF16)
    hipLaunchKernelGGL((quant_both_kernel<bf16>), grid, block, 0, stream, x_ptrs, (const bf16*)x_single, ldx, R, C, (uint8_t*)q,
                       (uint8_t*)s, (uint8_t*)qt, (uint8_t*)st, q_mat, s_mat, st_mat);
  else
    hipLaunchKernelGGL((quant_both_kernel<float>), grid, block, 0, stream, x_ptrs, (const float*)x_single, ldx, R, C, (uint8_t*)q,
                       (uint8_t*)s, (uint8_t*)qt, (uint8_t*)st, q_mat, s_mat, st_mat);
  CSMOE_CHECK_LAUNCH("quantize_mxfp8_both");
  return CSMOE_OK;
}

int k_quantize_mxfp8(const void* const* x_ptrs, const void* x_single, int E, int64_t ldx, int R, int C, int in_dtype, int transpose,
                     void* q, void* s, hipStream_t st) {
  if (E <= 0 || R <= 0 || C <= 0) return CSMOE_OK;
  const int64_t q_mat = (int64_t)R * C;
  const int64_t s_mat = transpose ? (int64_t)C * (R / 32) : (int64_t)R * (C / 32);
  if (!transpose) {
    const int64_t total = (int64_t)R * (C / 32);
    dim3 grid((unsigned)std::min<int64_t>((total + 255) / 256, 8192), (unsigned)E), block(256);
    if (in_dtype == CSMOE_BF16)
      hipLaunchKernelGGL((quant_rows_kernel<bf16>), grid, block, 0, st, x_ptrs, (const bf16*)x_single, ldx, R, C, (uint8_t*)q, (uint8_t*)s, q_mat, s_mat);
    else
      hipLaunchKernelGGL((quant_rows_kernel<float>), grid, block, 0, st, x_ptrs, (const float*)x_single, ldx, R, C, (uint8_t*)q, (uint8_t*)s, q_mat, s_mat);
  } else {
    const int64_t tiles = (int64_t)(R / 32) * ((C + 63) / 64);
    dim3 grid((unsigned)std::min<int64_t>((tiles + 3) / 4, 8192), (unsigned)E), block(256);
    if (in_dtype == CSMOE_BF16)
      hipLaunchKernelGGL((quant_cols_kernel<bf16>), grid, block, 0, st, x_ptrs, (const bf16*)x_single, ldx, R, C, (uint8_t*)q, (uint8_t*)s, q_mat, s_mat);
    else
      hipLaunchKernelGGL((quant_cols_kernel<float>), grid, block, 0, st, x_ptrs, (const float*)x_single, ldx, R, C, (uint8_t*)q, (uint8_t*)s, q_mat, s_mat);
  }
  CSMOE_CHECK_LAUNCH("quantize_mxfp8");
  return CSMOE_OK;
}
