// Router selection of ONE token row by one wave (lane l holds scores l, l+64, ...: VPL values per lane, E <= 64*VPL): fp32 softmax,
// K rounds of wave arg-max with the lowest-index tie break, and the renormalised weights of the five selection modes of
// include/csmoe.h.  Shared by router_select_kernel (moe_kernels.hip) and the one-pass router (router_fused.hip), so that the fused
// path selects and weighs bit-identically.
// (moe_model/model/moe/moe.py:113-132, smoe.py:44, competesmoe.py:246-255; moe_pretrain_model/layers/moe/deepseekv2.py:140-142,
//  deepseekv3.py:147-151)
#pragma once
#include "common.h"

__device__ __forceinline__ float round_dt(float v, int dtype) { return dtype == CSMOE_BF16 ? (float)(bf16)v : v; }
__device__ __forceinline__ float load_score(const void* p, int64_t i, int dtype) {
  return dtype == CSMOE_BF16 ? (float)((const bf16*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void store_score(void* p, int64_t i, float v, int dtype) {
  if (dtype == CSMOE_BF16) ((bf16*)p)[i] = (bf16)v; else ((float*)p)[i] = v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// wave arg-max with lowest-index tie break over (val, idx) pairs held one per lane
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

template <int VPL>
__device__ __forceinline__ void select_row(const float (&s)[VPL], int lane, int E, int K, int mode, int round_sum_bf16, float sel_param,
                                           int dtype, float* sm_row, int32_t* idx_row, float* w_row) {
  float key[VPL];
  float mx = -INFINITY;
#pragma unroll
  for (int v = 0; v < VPL; ++v) mx = fmaxf(mx, s[v]);
  mx = wave_max(mx);
  // fp32 softmax of the scores (always produced when requested: losses need it)
  float ex[VPL], sum = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int e = lane + 64 * v;
    ex[v] = e < E ? expf(s[v] - mx) : 0.f;
    sum += ex[v];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int e = lane + 64 * v;
    float p = ex[v] / sum;
    if (sm_row && e < E) sm_row[e] = p;
    if (mode == CSMOE_SEL_SOFTMAX) key[v] = p;
    else if (mode == CSMOE_SEL_SIGMOID) key[v] = round_dt(sigmoidf_(s[v]), dtype);
    else key[v] = s[v];
    if (e >= E) key[v] = -INFINITY;
  }
  // K rounds of wave arg-max, removing the winner each round
  float vsum = 0.f;
  float myv = 0.f;    // lane k keeps the k-th value
  int myi = 0;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      int e = lane + 64 * v;
      if (key[v] > bv || (key[v] == bv && e < bi)) { bv = key[v]; bi = e; }
    }
    wave_argmax(bv, bi);
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      if (lane + 64 * v == bi) key[v] = -INFINITY;
    if (lane == k) { myv = bv; myi = bi; }
    vsum += bv;   // summed in k order, identical in every lane
  }
  if (mode == CSMOE_SEL_TOPK_SOFTMAX) {
    // softmax over the K selected logits (fp32)
    float top = __shfl(myv, 0, 64);
    float ek = lane < K ? expf(myv - top) : 0.f;
    float es = wave_sum(ek);
    if (lane < K) { w_row[lane] = ek / es; idx_row[lane] = myi; }
    return;
  }
  float denom, wk;
  if (mode == CSMOE_SEL_TOPK_SIGMOID) {
    // w_k = sigmoid(logit_k / scale) in the logits' dtype, renormalised by the fp32 K-sum rounded to x.dtype (quotient fp32)
    const float sv = lane < K ? round_dt(sigmoidf_(round_dt(myv / sel_param, dtype)), dtype) : 0.f;
    float ssum = 0.f;
    for (int k = 0; k < K; ++k) ssum += __shfl(sv, k, 64);
    denom = round_sum_bf16 ? (float)(bf16)ssum : ssum;
    wk = sv / denom;
  } else if (mode == CSMOE_SEL_SOFTMAX) {
    denom = round_sum_bf16 ? (float)(bf16)vsum : vsum;
    wk = myv / denom;
  } else if (mode == CSMOE_SEL_RAW) {
    denom = round_dt(vsum, dtype);
    wk = round_dt(myv / denom, dtype);
  } else {  // SIGMOID: fp32 sum of the K sigmoids (+1e-20), fp32 quotient
    denom = vsum + 1e-20f;
    wk = myv / denom;
  }
  if (lane < K) { w_row[lane] = wk; idx_row[lane] = myi; }
}

// ---- the same selection with SIXTEEN lanes per row (E <= 64), four rows per wave at once ------------------------------------------
// Lane l16 of a row's 16-lane group stands for the four "virtual lanes" vl = l16 + 16 j (j < 4) of select_row<1>: it holds the
// scores, keys and (for vl < K) the selected values of those four.  Every reduction reproduces the 64-lane butterfly of
// wave_sum / wave_max level by level -- (x[vl] + x[vl^32]) + (x[vl^16] + x[vl^16^32]) inside the lane, then xor 8, 4, 2, 1 across
// the group -- so the softmax, the K-sums and the weights carry the bits select_row<1> produces; arg-max ties go to the lowest
// expert id in both.  Used by the one-pass router, where a wave owns 16 rows and 16 sequential wave-wide selections were a
// 40 us tail on a 55 us stream (tools/router_bench.py).
// Cross-lane steps as DPP row rotations (VALU speed) instead of __shfl_xor (ds_bpermute: an LDS round trip per step, ~25 dependent
// ones per row pass -- a 21 us selection tail behind a 52 us stream at E = 64, tools/router_diag.py).  Inside a 16-lane row, after the
// step with partner l ^ 8 (= rotation by 8) every value is replicated with period 8, so the partner l ^ 4 can be fetched as lane
// l + 4 (rotation by 4), then l + 2, l + 1: the same pairs as the xor butterfly, level by level, hence the same bits (fp add and
// max are commutative; the arg-max order is total).
template <int N>
__device__ __forceinline__ int dpp_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ float dpp_ror(float v) { return __builtin_bit_cast(float, dpp_ror_i<N>(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ float g16_max(float v) {
  v = fmaxf(v, dpp_ror<8>(v)); v = fmaxf(v, dpp_ror<4>(v)); v = fmaxf(v, dpp_ror<2>(v)); v = fmaxf(v, dpp_ror<1>(v));
  return v;
}
__device__ __forceinline__ float g16_sum64(const float (&x)[4]) {
  float v = (x[0] + x[2]) + (x[1] + x[3]);
  v += dpp_ror<8>(v); v += dpp_ror<4>(v); v += dpp_ror<2>(v); v += dpp_ror<1>(v);
  return v;
}
__device__ __forceinline__ void g16_argmax(float& v, int& i) {
#define CSMOE_G16_STEP(N)                                                  \
  {                                                                        \
    const float ov = dpp_ror<N>(v);                                        \
    const int oi = dpp_ror_i<N>(i);                                        \
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }                 \
  }
  CSMOE_G16_STEP(8) CSMOE_G16_STEP(4) CSMOE_G16_STEP(2) CSMOE_G16_STEP(1)
#undef CSMOE_G16_STEP
}

// s[j] = score of expert l16 + 16 j (-inf for experts >= E); `live` = this group's row exists (all lanes run the shuffles anyway)
// `myi` (out): lane l16 holds in myi[j] the (l16 + 16 j)-th selected expert id (the value it writes to idx_row), so that a caller that
// needs the ids again (the binning histogram of the one-pass router) does not read them back from global memory
__device__ __forceinline__ void select_row_g16(const float (&s)[4], int l16, int E, int K, int mode, int round_sum_bf16,
                                               float sel_param, int dtype, bool live, float* sm_row, int32_t* idx_row, float* w_row,
                                               int (&myi)[4]) {
  float key[4], ex[4];
  float mx = g16_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
#pragma unroll
  for (int j = 0; j < 4; ++j) ex[j] = (l16 + 16 * j) < E ? expf(s[j] - mx) : 0.f;
  const float sum = g16_sum64(ex);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = l16 + 16 * j;
    const float p = ex[j] / sum;
    if (live && sm_row && e < E) sm_row[e] = p;
    if (mode == CSMOE_SEL_SOFTMAX) key[j] = p;
    else if (mode == CSMOE_SEL_SIGMOID) key[j] = round_dt(sigmoidf_(s[j]), dtype);
    else key[j] = s[j];
    if (e >= E) key[j] = -INFINITY;
  }
  float vsum = 0.f;
  float myv[4] = {0.f, 0.f, 0.f, 0.f};     // virtual lane k keeps the k-th value
#pragma unroll
  for (int j = 0; j < 4; ++j) myi[j] = 0;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = l16 + 16 * j;
      if (key[j] > bv || (key[j] == bv && e < bi)) { bv = key[j]; bi = e; }
    }
    g16_argmax(bv, bi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (l16 + 16 * j == bi) key[j] = -INFINITY;
      if (l16 + 16 * j == k) { myv[j] = bv; myi[j] = bi; }
    }
    vsum += bv;
  }
  if (mode == CSMOE_SEL_TOPK_SOFTMAX) {
    const float top = __shfl(myv[0], 0, 16);
    float ek[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ek[j] = (l16 + 16 * j) < K ? expf(myv[j] - top) : 0.f;
    const float es = g16_sum64(ek);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (live && l16 + 16 * j < K) { w_row[l16 + 16 * j] = ek[j] / es; idx_row[l16 + 16 * j] = myi[j]; }
    return;
  }
  float denom, wk[4];
  if (mode == CSMOE_SEL_TOPK_SIGMOID) {
    float sv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sv[j] = (l16 + 16 * j) < K ? round_dt(sigmoidf_(round_dt(myv[j] / sel_param, dtype)), dtype) : 0.f;
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int kk = 0; kk < 16 && 16 * j + kk < K; ++kk) ssum += __shfl(sv[j], kk, 16);       // in k order, as the 64-lane form
    denom = round_sum_bf16 ? (float)(bf16)ssum : ssum;
#pragma unroll
    for (int j = 0; j < 4; ++j) wk[j] = sv[j] / denom;
  } else if (mode == CSMOE_SEL_SOFTMAX) {
    denom = round_sum_bf16 ? (float)(bf16)vsum : vsum;
#pragma unroll
    for (int j = 0; j < 4; ++j) wk[j] = myv[j] / denom;
  } else if (mode == CSMOE_SEL_RAW) {
    denom = round_dt(vsum, dtype);
#pragma unroll
    for (int j = 0; j < 4; ++j) wk[j] = round_dt(myv[j] / denom, dtype);
  } else {
    denom = vsum + 1e-20f;
#pragma unroll
    for (int j = 0; j < 4; ++j) wk[j] = myv[j] / denom;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (live && l16 + 16 * j < K) { w_row[l16 + 16 * j] = wk[j]; idx_row[l16 + 16 * j] = myi[j]; }
}
__device__ __forceinline__ void select_row_g16(const float (&s)[4], int l16, int E, int K, int mode, int round_sum_bf16,
                                               float sel_param, int dtype, bool live, float* sm_row, int32_t* idx_row, float* w_row) {
  int myi[4];
  select_row_g16(s, l16, E, K, mode, round_sum_bf16, sel_param, dtype, live, sm_row, idx_row, w_row, myi);
}
