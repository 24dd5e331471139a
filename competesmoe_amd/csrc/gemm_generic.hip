// Generic grouped GEMM (any shape / stride / dtype) on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Role: the fp32 path (reference tolerance 1e-5: the f32-input MFMA is bit-for-bit a k-ordered fmaf chain,
// guide §3 "FP32-input MFMA") and the shape-agnostic fallback of the bf16 path (odd K, unaligned leading
// dimensions).  64x64 output tile per 256-thread workgroup, 4 waves of 32x32, K-step 32, operands staged through
// LDS as fp32 with a +1 pad (ds_read_b32 of 32 distinct rows -> 32 distinct banks).
//
// One kernel covers every operand layout through generic (row-stride, k-stride) pairs:
//   row-space GEMM  C[m,n]  = sum_k A[m,k] * B_e[n,k]|B_e[k,n]      (cvmm.py:61-168 semantics)
//   weight gradient C_e[i,j] = sum_m A[m,i] * B[m,j]                 (cvmm.py:194-345 semantics, no atomics)
#include "common.h"

namespace {

constexpr int GT = 64;   // tile edge
constexpr int GK = 32;   // k step

template <typename T>
__device__ __forceinline__ void stage_tile(float (*dst)[GK + 1], const T* base, int64_t rs, int64_t ks,
                                           int rows_valid, int k0, int kmax, bool k_contig) {
  // 64 rows x 32 k = 2048 elements, 8 per thread; the unit-stride dimension runs across lanes.
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int r, k;
    if (k_contig) { k = t & 31; r = (t >> 5) + 8 * i; }
    else          { r = t & 63; k = (t >> 6) + 4 * i; }
    float v = 0.f;
    if (r < rows_valid && (k0 + k) < kmax) v = DT<T>::ld(base + (int64_t)r * rs + (int64_t)(k0 + k) * ks);
    dst[r][k] = v;
  }
}

struct GGArgs {
  const void* A; int64_t a_rs, a_ks;          // "row" operand: row stride / reduction stride (elements)
  const void* const* b_ptrs; const void* Bflat; int64_t b_rs, b_ks;   // "col" operand
  const void* const* bias_ptrs;
  const int32_t* offsets; int E;
  int single_M; const void* single_B; const void* single_bias; void* single_C;
  int N, Kd;
  void* C; void* C2; const void* aux; int64_t ldc;
  void* const* c_ptrs;                         // wgrad outputs
  int epilogue, act, accumulate;
};

// mode 0: row-space GEMM. blockIdx.x = n tile, blockIdx.y = m-tile slot (ceil(M/64)+E slots, most live).
// mode 1: wgrad.          blockIdx.x = j tile, blockIdx.y = i tile, blockIdx.z = expert.
template <typename T, typename TOut, int MODE>
__global__ void __launch_bounds__(256) gg_generic_kernel(GGArgs p) {
  __shared__ float As[GT][GK + 1];
  __shared__ float Bs[GT][GK + 1];

  int e, row0 = 0, rows = 0, red_len = 0;
  const T* Abase; const T* Bbase;
  int tile_r0, tile_c0, r_lim, c_lim;
  if (MODE == 0) {
    // find (expert, m-tile) for this slot
    int slot = blockIdx.y, acc = 0;
    e = -1;
    for (int i = 0; i < p.E; ++i) {
      int o0 = p.offsets ? p.offsets[i] : 0, o1 = p.offsets ? p.offsets[i + 1] : p.single_M;
      int nt = (o1 - o0 + GT - 1) / GT;
      if (slot < acc + nt) { e = i; row0 = o0 + (slot - acc) * GT; rows = min(GT, o1 - row0); break; }
      acc += nt;
    }
    if (e < 0) return;
    tile_c0 = blockIdx.x * GT;
    Abase = (const T*)p.A + (int64_t)row0 * p.a_rs;
    Bbase = (const T*)(p.b_ptrs ? p.b_ptrs[e] : p.single_B) + (int64_t)tile_c0 * p.b_rs;
    red_len = p.Kd;
    tile_r0 = row0; r_lim = rows; c_lim = min(GT, p.N - tile_c0);
  } else {
    e = blockIdx.z;
    int o0 = p.offsets ? p.offsets[e] : 0, o1 = p.offsets ? p.offsets[e + 1] : p.single_M;
    red_len = o1 - o0;
    tile_r0 = blockIdx.y * GT; tile_c0 = blockIdx.x * GT;
    // A-operand rows = output rows i (stride 1 along i), reduction = binned rows m
    Abase = (const T*)p.A + (int64_t)o0 * p.a_ks + (int64_t)tile_r0 * p.a_rs;
    Bbase = (const T*)p.Bflat + (int64_t)o0 * p.b_ks + (int64_t)tile_c0 * p.b_rs;
    r_lim = min(GT, p.N - tile_r0);      // N = Na here
    c_lim = min(GT, p.Kd - tile_c0);     // Kd = Nb here
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool a_kc = (p.a_ks == 1), b_kc = (p.b_ks == 1);

  for (int k0 = 0; k0 < red_len; k0 += GK) {
    stage_tile<T>(As, Abase, p.a_rs, p.a_ks, r_lim, k0, red_len, a_kc);
    stage_tile<T>(Bs, Bbase, p.b_rs, p.b_ks, c_lim, k0, red_len, b_kc);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      float a = As[wr * 32 + (lane & 31)][kk + (lane >> 5)];
      float b = Bs[wc * 32 + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = wc * 32 + (lane & 31);
  if (col >= c_lim) return;
  if (MODE == 0) {
    const int n = tile_c0 + col;
    float bias = 0.f;
    const bool post_bias = p.epilogue == CSMOE_EPI_ROUND_BIAS32_ACT;   // fp32 bias added to the ROUNDED product (cvmm + bias)
    if (p.epilogue == CSMOE_EPI_BIAS || p.epilogue == CSMOE_EPI_BIAS_ACT || post_bias) {
      const void* bp = p.bias_ptrs ? p.bias_ptrs[e] : p.single_bias;
      if (bp) bias = post_bias ? ((const float*)bp)[n] : DT<T>::ld((const T*)bp + n);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= r_lim) continue;
      int64_t o = (int64_t)(tile_r0 + row) * p.ldc + n;
      float v = acc[r];
      if (p.epilogue == CSMOE_EPI_ACTGRAD || p.epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE) {
        float g = DT<T>::rnd(v);
        if (p.epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE) g = DT<T>::rnd(((const float*)p.C2)[tile_r0 + row] * g);
        float h = DT<T>::ld((const T*)p.aux + o);
        DT<T>::st((T*)p.C + o, g * (p.act == CSMOE_ACT_QUICK_GELU ? quick_gelu_grad_rounded<T>(h) : act_bwd(h, p.act)));
      } else {
        float h = post_bias ? DT<T>::rnd(v) + bias : DT<T>::rnd(v + bias);
        if (p.C) DT<T>::st((T*)p.C + o, h);
        if ((p.epilogue == CSMOE_EPI_BIAS_ACT || post_bias) && p.C2)
          DT<T>::st((T*)p.C2 + o, p.act == CSMOE_ACT_QUICK_GELU ? quick_gelu_rounded<T>(h) : act_fwd(h, p.act));
      }
    }
  } else {
    TOut* Ce = (TOut*)(p.c_ptrs ? p.c_ptrs[e] : p.single_C);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= r_lim) continue;
      int64_t o = (int64_t)(tile_r0 + row) * p.ldc + tile_c0 + col;
      float v = acc[r];
      if (p.accumulate) v += DT<TOut>::ld(Ce + o);
      DT<TOut>::st(Ce + o, v);
    }
  }
}

}  // namespace

int gg_generic_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                        const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C,
                        void* C2, const void* aux, int64_t ldc, int epilogue, int act, int dtype, const void* single_B,
                        const void* single_bias, hipStream_t st) {
  GGArgs p{};
  p.single_M = M; p.single_B = single_B; p.single_bias = single_bias;
  p.A = A; p.a_rs = lda; p.a_ks = 1;
  p.b_ptrs = b_ptrs;
  if (b_layout == CSMOE_B_NK) { p.b_rs = ldb; p.b_ks = 1; } else { p.b_rs = 1; p.b_ks = ldb; }
  p.bias_ptrs = bias_ptrs; p.offsets = offsets; p.E = E; p.N = N; p.Kd = Kd;
  p.C = C; p.C2 = C2; p.aux = aux; p.ldc = ldc; p.epilogue = epilogue; p.act = act;
  dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT + E);
  if (grid.y > 65535) { csmoe_set_error("grouped_gemm(generic): too many row tiles (%u)", grid.y); return CSMOE_ERR_UNSUPPORTED; }
  if (dtype == CSMOE_F32) hipLaunchKernelGGL((gg_generic_kernel<float, float, 0>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gg_generic_kernel<bf16, bf16, 0>), grid, dim3(256), 0, st, p);
  CSMOE_CHECK_LAUNCH("grouped_gemm(generic)");
  return CSMOE_OK;
}

int gg_generic_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na,
                     int Nb, void* const* c_ptrs, int64_t ldc, int dtype, int out_dtype, int accumulate, int single_M,
                     void* single_C, hipStream_t st) {
  GGArgs p{};
  p.single_M = single_M; p.single_C = single_C;
  p.A = A; p.a_rs = 1; p.a_ks = lda;
  p.Bflat = B; p.b_rs = 1; p.b_ks = ldb;
  p.offsets = offsets; p.E = E; p.N = Na; p.Kd = Nb; p.c_ptrs = c_ptrs; p.ldc = ldc; p.accumulate = accumulate;
  dim3 grid((Nb + GT - 1) / GT, (Na + GT - 1) / GT, E);
  if (dtype == CSMOE_F32) {
    if (out_dtype != CSMOE_F32) { csmoe_set_error("wgrad: fp32 inputs need fp32 output"); return CSMOE_ERR_INVALID; }
    hipLaunchKernelGGL((gg_generic_kernel<float, float, 1>), grid, dim3(256), 0, st, p);
  } else if (out_dtype == CSMOE_F32) {
    hipLaunchKernelGGL((gg_generic_kernel<bf16, float, 1>), grid, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((gg_generic_kernel<bf16, bf16, 1>), grid, dim3(256), 0, st, p);
  }
  CSMOE_CHECK_LAUNCH("grouped_wgrad(generic)");
  return CSMOE_OK;
}
