// Shared pieces of the bf16 grouped-GEMM kernels: argument block, buffer descriptors, the two swizzled 16 KiB LDS tile
// images (KC / KM, see gemm_bf16.hip) with their LDS-DMA fill and fragment reads, and the 8-wide activation helpers.
#pragma once
#include "common.h"

namespace ggt {

constexpr unsigned OOB = 0x80000000u;       // any offset >= num_records reads as zero
constexpr int TILE_B = 16384;               // one 128x64 (KC) / 64x128 (KM) bf16 tile image

enum { KC = 0, KM = 1 };

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct FastArgs {
  // row operand ("tokens" / wgrad A): [*, ld_r]; col operand (weights / wgrad B)
  const void* R; int64_t ld_r;
  const void* Cflat; int64_t ld_c;            // wgrad col operand (flat [M, Nb])
  const void* const* c_ptrs_in;               // row-space: per-expert weight pointers
  const void* const* bias_ptrs;
  const int32_t* offsets; int E;
  int single_M; const void* single_B; const void* single_bias; void* single_C;
  int NR;      // row-space: unused;  wgrad: Na (output rows)
  int NC;      // output columns (N or Nb)
  int Kd;      // row-space reduction length
  void* C; void* C2; const void* aux; int64_t ldc;
  void* const* out_ptrs;                      // wgrad outputs
  int epilogue, act, accumulate, out_f32;
  const int32_t* xcd_order;                   // persistent wgrad: experts dealt to XCDs (csmoe_expert_order), or null
  int tile_band;                              // persistent weight gradient: row tiles per band of the tile order (gemm_bf16_v2p.hip tile_of)
  int row_part;                               // row-space v2: which row tiles of every expert (common.h part_tiles)
  int thin_loop;                              // row-space v2: tiles of <= 128 rows take the one-phase loop
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}


// ---- LDS-DMA fill of one 16 KiB tile image: NJ wave-instructions per wave (NJ * waves = 16) ----------------------
// KC: tile rows are operand rows (stride ld_bytes), 64 reduction elements per row starting at red0.
// KM: tile rows are 64 reduction rows, 128 operand columns starting at col0.
template <int KIND, int NJ>
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, const unsigned (&vbase)[NJ],
                                         const int (&aux)[NJ], int red0, int red_len, unsigned ld_bytes, int wave) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    unsigned voff;
    if (KIND == KC) {
      voff = vbase[j] + (unsigned)red0 * 2u;            // aux[j] = first reduction element of this lane's chunk
      if (red0 + aux[j] >= red_len) voff = OOB;
    } else {
      voff = vbase[j] + (unsigned)red0 * ld_bytes;      // vbase already OOB for out-of-range columns; rows past the
    }                                                   // end of the reduction fall off num_records
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * NJ + j) * 1024), 16, voff, 0, 0, 0);
  }
}

// one 1 KiB piece (wave-instruction) of an image: lets a caller interleave the NJ pieces with other work
template <int KIND, int NJ>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, unsigned vbase_j, int aux_j, int j,
                                          int red0, int red_len, unsigned ld_bytes, int wave) {
  unsigned voff;
  if (KIND == KC) {
    voff = vbase_j + (unsigned)red0 * 2u;
    if (red0 + aux_j >= red_len) voff = OOB;
  } else {
    voff = vbase_j + (unsigned)red0 * ld_bytes;
  }
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * NJ + j) * 1024), 16, voff, 0, 0, 0);
}

// per-lane, loop-invariant DMA source offsets.  Image row / column i (0..127) is fetched from source row / column
//   src(i) = (i >> gshift) * gstride + (i & ((1 << gshift) - 1)) + sub
// (identity for gshift = 7).  The v2 kernel uses it to build images out of the 64-row / 32-column strips each wave reads in
// one phase.
template <int KIND, int NJ>
__device__ __forceinline__ void dma_setup(unsigned (&vbase)[NJ], int (&aux)[NJ], unsigned ld_bytes, int col0, int ncols,
                                          int gshift, int gstride, int sub, int wave, int lane) {
  const int gmask = (1 << gshift) - 1;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int P = (wave * NJ + j) * 64 + lane;
    if (KIND == KC) {
      int r = P >> 3, pc = P & 7;
      int c = pc ^ ((r >> 1) & 7);
      int src = (r >> gshift) * gstride + (r & gmask) + sub;
      vbase[j] = (unsigned)src * ld_bytes + (unsigned)c * 16u;
      aux[j] = c * 8;
    } else {
      int k = P >> 4, pc = P & 15;
      int f = (k & 3) | (((k >> 3) & 1) << 2);
      int seg = (pc >> 1) ^ f;
      int ic = seg * 16 + (pc & 1) * 8;
      int col = col0 + (ic >> gshift) * gstride + (ic & gmask) + sub;
      vbase[j] = (col < ncols) ? ((unsigned)k * ld_bytes + (unsigned)col * 2u) : OOB;
      aux[j] = 0;
    }
  }
}

// ---- fragment reads ------------------------------------------------------------------------------------------------
// KC: block b = 16 operand rows; returns the 8 reduction elements k = s*32 + 8*(lane>>4) .. +7 of row (lane&15)
__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int lane_off, int b, int s) {
  return *(const bf16x8*)(tile + b * 2048 + (lane_off ^ (s << 6)));
}
// KM: column block cb; two transposed 4x16 reads
__device__ __forceinline__ bf16x8 frag_km(const char* tile, int addr_cb, int s) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + addr_cb + s * 8192));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + addr_cb + s * 8192 + 1024));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Same fragment by inline asm, for the hand-scheduled v2 loops.  hipcc's waitcnt pass cannot tell which LDS bytes a pending
// LDS-DMA (buffer_load ... lds) will write and puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 builtin: that
// drained the whole prefetch queue twice per K-tile in every kernel with a K-major image (profiles/r01).  The asm form is
// invisible to that pass; the CALLER owns the ordering: the counted vmcnt + s_barrier before the slot is read, and
// `s_waitcnt lgkmcnt(0)` before the first use of the result (PHASE_SYNC_IN does both).
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ bf16x8 frag_km_raw(const char* tile, int addr_cb, int s) {
  const unsigned a = lds_addr(tile) + (unsigned)addr_cb;
  i32x2 lo, hi;
  if (s == 0) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a) : "memory");
  } else {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(hi) : "v"(a) : "memory");
  }
  i32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- fast GELU for the bf16 epilogues -------------------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below half a bf16 ulp), one v_exp + one v_rcp: the OCML erff costs
// ~50 instructions and made the bias+GELU epilogue ~30 % of a K=4096 tile (profiles/r01).  The fp32 path (gemm_generic.hip)
// keeps the exact erff.
// Both formulas (GELU and its derivative) over 8 values as 4 pairs, written on 2-vectors so that the multiplies / adds / FMAs become packed fp32
// instructions (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32: two elements per issue slot); v_rcp_f32 / v_exp_f32 stay per element.
// Per tile of the bias+GELU epilogue this arithmetic was ~15 us of a 134 us tile (tools/tile_stamps.py): VALU-bound, no MFMA beside it.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float c) { return f32x2{c, c}; }
// erf(z) for a pair; e_out = exp(-z*z)
__device__ __forceinline__ f32x2 erf_as2(f32x2 z, f32x2& e_out) {
  const f32x2 az = {fabsf(z[0]), fabsf(z[1])};
  const f32x2 d = pk_fma(pk_splat(0.3275911f), az, pk_splat(1.f));
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 a2 = (az * az) * pk_splat(-1.44269504088896340736f);           // -z^2 * log2(e)
  const f32x2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
  f32x2 p = pk_fma(pk_splat(1.061405429f), t, pk_splat(-1.453152027f));
  p = pk_fma(p, t, pk_splat(1.421413741f));
  p = pk_fma(p, t, pk_splat(-0.284496736f));
  p = pk_fma(p, t, pk_splat(0.254829592f));
  const f32x2 r = pk_fma(-(p * t), e, pk_splat(1.f));
  e_out = e;
  return f32x2{copysignf(r[0], z[0]), copysignf(r[1], z[1])};
}
__device__ __forceinline__ void gelu_fast8(float (&v)[8]) {
#ifdef CSMOE_EPI_NOMATH     // timing experiment (wrong results): what the GELU arithmetic of the epilogues costs
  return;
#endif
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 x = {v[2 * k], v[2 * k + 1]};
    f32x2 e;
    const f32x2 erf = erf_as2(x * pk_splat(0.70710678118654752440f), e);
    const f32x2 y = (x * pk_splat(0.5f)) * (erf + pk_splat(1.f));
    v[2 * k] = y[0]; v[2 * k + 1] = y[1];
  }
}
__device__ __forceinline__ void gelu_grad_fast8(float (&h)[8]) {
#ifdef CSMOE_EPI_NOMATH
  return;
#endif
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 x = {h[2 * k], h[2 * k + 1]};
    f32x2 e;   // exp(-x^2/2)
    const f32x2 erf = erf_as2(x * pk_splat(0.70710678118654752440f), e);
    const f32x2 cdf = pk_fma(erf, pk_splat(0.5f), pk_splat(0.5f));
    const f32x2 y = pk_fma(x * pk_splat(0.39894228040143267794f), e, cdf);
    h[2 * k] = y[0]; h[2 * k + 1] = y[1];
  }
}

// activation over 8 values, `switch` outside the element loop so each formula is emitted once
__device__ __forceinline__ void act_fwd8(float (&v)[8], int act) {
  switch (act) {
    case CSMOE_ACT_RELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j], CSMOE_ACT_RELU);
      break;
    case CSMOE_ACT_GELU:
      gelu_fast8(v);
      break;
    case CSMOE_ACT_GELU_TANH:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j], CSMOE_ACT_GELU_TANH);
      break;
    case CSMOE_ACT_SILU:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j], CSMOE_ACT_SILU);
      break;
    case CSMOE_ACT_QUICK_GELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = quick_gelu_rounded<bf16>(v[j]);
      break;
    default: break;
  }
}
// h[j] <- act'(h[j])
__device__ __forceinline__ void act_bwd8(float (&h)[8], int act) {
  switch (act) {
    case CSMOE_ACT_RELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = act_bwd(h[j], CSMOE_ACT_RELU);
      break;
    case CSMOE_ACT_GELU:
      gelu_grad_fast8(h);
      break;
    case CSMOE_ACT_GELU_TANH:
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = act_bwd(h[j], CSMOE_ACT_GELU_TANH);
      break;
    case CSMOE_ACT_SILU:
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = act_bwd(h[j], CSMOE_ACT_SILU);
      break;
    case CSMOE_ACT_QUICK_GELU:
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = quick_gelu_grad_rounded<bf16>(h[j]);
      break;
    default:
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = 1.f;
      break;
  }
}

}  // namespace ggt
