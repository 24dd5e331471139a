// HBM-bound kernels of the sparse-MoE hot path for gfx950: router selection, token binning, dispatch / combine,
// bias-gradient column sums and the competition affinity reduction.  One wave (64 lanes) per token row, 16-byte
// vector accesses, wavefront shuffles for the row reductions / arg-max.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "router_select.h"
#include <algorithm>

namespace {

// =====================================================================================================================
// Router selection (moe_model/model/moe/moe.py:113-132, smoe.py:44, competesmoe.py:246-255;
//                   moe_pretrain_model/layers/moe/deepseekv2.py:140-142, deepseekv3.py:147-151)
// One wave per token; lane l holds scores l, l+64, ... (VPL values per lane, E <= 64*VPL).
// =====================================================================================================================
template <int VPL>
__global__ void __launch_bounds__(256) router_select_kernel(const void* scores, int dtype, int T, int E, int K, int mode,
                                                            int round_sum_bf16, float sel_param, float* softmax, int32_t* idx,
                                                            float* w) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int64_t base = (int64_t)t * E;
  float s[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int e = lane + 64 * v;
    s[v] = e < E ? load_score(scores, base + e, dtype) : -INFINITY;
  }
  select_row<VPL>(s, lane, E, K, mode, round_sum_bf16, sel_param, dtype, softmax ? softmax + base : nullptr, idx + (int64_t)t * K,
                  w + (int64_t)t * K);
}

template <int VPL>
__global__ void __launch_bounds__(256) router_select_bwd_kernel(const void* scores, int dtype, int T, int E, int K, int mode,
                                                                int round_sum_bf16, float sel_param, const float* softmax,
                                                                const int32_t* idx,
                                                                const float* w, const float* dw, const float* dsoftmax,
                                                                void* dscores) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int64_t base = (int64_t)t * E;
  // lane k (< K) holds slot k
  int myi = -1;
  float mydw = 0.f, myw = 0.f, myv = 0.f;
  if (lane < K) {
    myi = idx[(int64_t)t * K + lane];
    mydw = dw ? dw[(int64_t)t * K + lane] : 0.f;
    myw = w[(int64_t)t * K + lane];
    if (mode == CSMOE_SEL_SOFTMAX) myv = softmax[base + myi];
    else if (mode == CSMOE_SEL_SIGMOID) myv = round_dt(sigmoidf_(load_score(scores, base + myi, dtype)), dtype);
    else if (mode == CSMOE_SEL_TOPK_SIGMOID)
      myv = round_dt(sigmoidf_(round_dt(load_score(scores, base + myi, dtype) / sel_param, dtype)), dtype);
    else myv = load_score(scores, base + myi, dtype);
  }
  float dv = 0.f;   // gradient w.r.t. the selected value of slot `lane`
  if (mode == CSMOE_SEL_TOPK_SOFTMAX) {
    float dot = wave_sum(mydw * myw);
    dv = myw * (mydw - dot);
  } else {
    float ssum = wave_sum(myv);
    float denom;
    if (mode == CSMOE_SEL_SOFTMAX || mode == CSMOE_SEL_TOPK_SIGMOID) denom = round_sum_bf16 ? (float)(bf16)ssum : ssum;
    else if (mode == CSMOE_SEL_RAW) denom = round_dt(ssum, dtype);
    else denom = ssum + 1e-20f;
    float dot = wave_sum(mydw * myv);
    // d/d(denominator) = -sum_k dw_k v_k / denom^2.  Where the reference casts the K-sum to x.dtype (`.to(x.dtype)`, smoe.py:44) the
    // denominator is a bf16 TENSOR, and autograd hands a bf16 tensor a bf16 gradient: that term is rounded before it is broadcast
    // back to the K values (observed: d gate.weight 3.2e-3 off and half its elements different without the rounding).
    float dden = -dot / (denom * denom);
    if (round_sum_bf16 && (mode == CSMOE_SEL_SOFTMAX || mode == CSMOE_SEL_TOPK_SIGMOID)) dden = (float)(bf16)dden;
    dv = mydw / denom + dden;
    if (mode == CSMOE_SEL_RAW) {
      // competition step (competesmoe.py:253-255): `weights` is an x.dtype TENSOR taken from the affinities and used twice -- as the
      // numerator and in the K-sum -- so under bf16 it receives its own gradient rounded (the weighted sum's backward hands back a
      // bf16 tensor), then two bf16 gradients (quotient, sum), added in bf16; the result is scattered into d affinity.  fp32: no-ops.
      // The denominator's gradient in autograd's own sequence for `self / other` (derivatives.yaml: -grad * ((self / other) / other),
      // every op a bf16 tensor op), then the K-sum of `sum`'s backward rounded once: emulated on the CPU this reproduces torch's
      // gradient bit for bit, where one rounding of the closed form -sum(dw v) / denom^2 was 4.2e-3 off it (and d affinity with it:
      // tools/comp_grad_probe.py).
      const float dwr = round_dt(mydw, dtype);
      const float q = round_dt(round_dt(myv / denom, dtype) / denom, dtype);
      const float ddr = round_dt(wave_sum(round_dt(-dwr * q, dtype)), dtype);
      dv = round_dt(round_dt(dwr / denom, dtype) + ddr, dtype);
    }
    if (mode == CSMOE_SEL_SIGMOID) {
      // deepseekv3 under autocast (deepseekv3.py:147-151): the K sigmoids are a bf16 TENSOR used twice -- as the numerator and,
      // through the fp32-policy `sum`, in the denominator -- so autograd hands it two gradients, each rounded to bf16 (the quotient's
      // backward; the cast in front of the sum), adds them in bf16, and sigmoid's backward multiplies the rounded sum (one more
      // rounding at the store below).  Without these three roundings d w_gate was 4.6e-3 off the reference's (fp32 runs: no-ops).
      dv = round_dt(round_dt(mydw / denom, dtype) + round_dt(dden, dtype), dtype);
      dv *= myv * (1.f - myv);
    }
    if (mode == CSMOE_SEL_TOPK_SIGMOID) dv *= myv * (1.f - myv) / sel_param;
  }
  // scatter dv to the expert positions; softmax-path gradient
  float g[VPL], p[VPL];
  float inner = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int e = lane + 64 * v;
    p[v] = (softmax && e < E) ? softmax[base + e] : 0.f;
    g[v] = (dsoftmax && e < E) ? dsoftmax[base + e] : 0.f;   // d loss / d softmax
  }
  float direct[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) direct[v] = 0.f;
  for (int k = 0; k < K; ++k) {
    int ei = __shfl(myi, k, 64);
    float d = __shfl(dv, k, 64);
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      if (lane + 64 * v == ei) {
        if (mode == CSMOE_SEL_SOFTMAX) g[v] += d; else direct[v] += d;
      }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) inner += g[v] * p[v];
  inner = wave_sum(inner);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    int e = lane + 64 * v;
    // SEL_RAW: the scores feed the fp32 softmax AND the top-K values: two gradient tensors in the scores' dtype, added in that dtype
    const float sp = p[v] * (g[v] - inner);
    if (e < E) store_score(dscores, base + e, mode == CSMOE_SEL_RAW ? round_dt(sp, dtype) + direct[v] : sp + direct[v], dtype);
  }
}

// =====================================================================================================================
// Binning: stable counting sort of expert ids (replaces cvmm_prepare_sel2's sort, cvmm.py:580-593, and the E torch.where
// scans of compute_moe, moe.py:189-191).  1024 ids per workgroup; ranks inside a wave by ballot "match-any".
// =====================================================================================================================
constexpr int BIN_CHUNK = 1024;

__global__ void __launch_bounds__(256) bin_hist_kernel(const int32_t* idx, int n, int E, int32_t* block_hist) {
  extern __shared__ int32_t h[];
  for (int i = threadIdx.x; i < E; i += 256) h[i] = 0;
  __syncthreads();
  int base = blockIdx.x * BIN_CHUNK;
  for (int i = threadIdx.x; i < BIN_CHUNK; i += 256) {
    int j = base + i;
    if (j < n) {
      int e = idx[j];
      if (e >= 0 && e < E) atomicAdd(&h[e], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += 256) block_hist[(int64_t)blockIdx.x * E + i] = h[i];
}

__global__ void __launch_bounds__(1024) bin_scan_kernel(const int32_t* block_hist, int nb, int E, int P, int32_t* counts,
                                                        int32_t* offsets, int32_t* block_base) {
  // P threads per expert, each over a contiguous range of blocks (thread i: expert i % E, part i / E: a block's E counts are
  // read by consecutive threads).  One thread per expert walking all nb blocks was 100 us at nb = 512 (dependent L2 round trips).
  extern __shared__ int32_t c[];   // [P*E] part sums -> exclusive part offsets, then [E] expert offsets
  int32_t* eoff = c + P * E;
  const int per = (nb + P - 1) / P;
  for (int i = threadIdx.x; i < P * E; i += blockDim.x) {
    const int e = i % E, p = i / E;
    const int b0 = p * per, b1 = min(nb, b0 + per);
    int s = 0;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += block_hist[(int64_t)b * E + e];
    c[i] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    int run = 0;
    for (int p = 0; p < P; ++p) { const int v = c[p * E + e]; c[p * E + e] = run; run += v; }
    counts[e] = run;
    eoff[e] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < E; ++e) { const int v = eoff[e]; eoff[e] = run; offsets[e] = run; run += v; }
    offsets[E] = run;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P * E; i += blockDim.x) {
    const int e = i % E, p = i / E;
    const int b0 = p * per, b1 = min(nb, b0 + per);
    int run = eoff[e] + c[i];
#pragma unroll 8
    for (int b = b0; b < b1; ++b) {
      block_base[(int64_t)b * E + e] = run;
      run += block_hist[(int64_t)b * E + e];
    }
  }
}

__global__ void __launch_bounds__(256) bin_scatter_kernel(const int32_t* idx, int n, int E, int ebits, int chunk,
                                                          const int32_t* block_base, int32_t* perm, int32_t* slot_of) {
  extern __shared__ int32_t sm[];
  int32_t* running = sm;          // [E]
  int32_t* wcnt = sm + E;         // [4][E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < E; i += 256) running[i] = block_base[(int64_t)blockIdx.x * E + i];
  for (int i = threadIdx.x; i < 4 * E; i += 256) wcnt[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * chunk;                 // `chunk` ids per workgroup: the block size the histogram was built with
  for (int round = 0; round < (chunk + 255) / 256; ++round) {
    int j = base + round * 256 + threadIdx.x;
    bool valid = j < n && round * 256 + (int)threadIdx.x < chunk;
    int e = valid ? idx[j] : -1;
    valid = valid && e >= 0 && e < E;
    // lanes of this wave holding the same expert
    unsigned long long mask = __ballot(valid);
    for (int b = 0; b < ebits; ++b) {
      unsigned long long bm = __ballot((e >> b) & 1);
      mask &= ((e >> b) & 1) ? bm : ~bm;
    }
    int rank = __popcll(mask & ((1ull << lane) - 1ull));
    int cnt = __popcll(mask);
    if (valid && rank == cnt - 1) wcnt[wave * E + e] = cnt;   // one writer per (wave, expert)
    __syncthreads();
    if (valid) {
      int pos = running[e] + rank;
      for (int w2 = 0; w2 < wave; ++w2) pos += wcnt[w2 * E + e];
      perm[pos] = j;
      slot_of[j] = pos;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E; i += 256) {
      running[i] += wcnt[i] + wcnt[E + i] + wcnt[2 * E + i] + wcnt[3 * E + i];
      wcnt[i] = 0; wcnt[E + i] = 0; wcnt[2 * E + i] = 0; wcnt[3 * E + i] = 0;
    }
    __syncthreads();
  }
}

// =====================================================================================================================
// Dispatch (gather rows into the binned row space) -- moe.py:201 `x[batch_idx, token_idx]`, cvmm.py:114-119
// One wave per destination row, 16 B per lane per access, rows visited with a grid-stride.
// =====================================================================================================================
__global__ void __launch_bounds__(256) dispatch_rows_kernel(const char* x, const int32_t* perm, int K, char* xs, int n,
                                                            int row_bytes, int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  const int nv = vec_ok ? (row_bytes >> 4) : 0;
  for (int m = wave_g; m < n; m += nw) {
    const int t = perm[m] / K;
    const i32x4* src = (const i32x4*)(x + (int64_t)t * row_bytes);
    i32x4* dst = (i32x4*)(xs + (int64_t)m * row_bytes);
    for (int c0 = 0; c0 < nv; c0 += 512) {           // eight 16-byte loads in flight per lane, then the stores
      i32x4 r[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int i = c0 + c * 64 + lane;
        if (i < nv) r[c] = src[i];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int i = c0 + c * 64 + lane;
        if (i < nv) dst[i] = r[c];
      }
    }
    // tail (row_bytes not a multiple of 16): 2-byte granularity
    for (int b = (nv << 4) + lane * 2; b < row_bytes; b += 128)
      *(short*)((char*)dst + b) = *(const short*)((const char*)src + b);
  }
}

// Token-major form: one wave per TOKEN copies its row to the K binned rows slot_of[t*K+k]; the source row is re-read from
// L1/L2 for k > 0, so HBM sees it once (the row-major form above fetches x once per selected expert: 1.33x the bytes).
__global__ void __launch_bounds__(256) dispatch_tokens_kernel(const char* x, const int32_t* slot_of, int K, char* xs, int T,
                                                              int row_bytes, int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  const int nv = vec_ok ? (row_bytes >> 4) : 0;
  for (int t = wave_g; t < T; t += nw) {
    const i32x4* src = (const i32x4*)(x + (int64_t)t * row_bytes);
    for (int k = 0; k < K; ++k) {
      const int m = slot_of[(int64_t)t * K + k];
      i32x4* dst = (i32x4*)(xs + (int64_t)m * row_bytes);
      for (int i = lane; i < nv; i += 64) dst[i] = src[i];
      for (int b = (nv << 4) + lane * 2; b < row_bytes; b += 128)
        *(short*)((char*)dst + b) = *(const short*)((const char*)src + b);
    }
  }
}

// The same with the source row held in registers: up to eight 16-byte loads in flight per lane (one 8 KiB row per wave pass),
// then K x 8 stores; two loads in flight per lane (the plain loop above) leave an HBM-bound copy at 4.9 TB/s.
__global__ void __launch_bounds__(256) dispatch_tokens_reg_kernel(const char* x, const int32_t* slot_of, int K, char* xs, int T,
                                                                  int row_bytes) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  const int nv = row_bytes >> 4;
  for (int t = wave_g; t < T; t += nw) {
    const i32x4* src = (const i32x4*)(x + (int64_t)t * row_bytes);
    for (int c0 = 0; c0 < nv; c0 += 512) {
      i32x4 r[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int i = c0 + c * 64 + lane;
        if (i < nv) r[c] = src[i];
      }
      for (int k = 0; k < K; ++k) {
        const int m = slot_of[(int64_t)t * K + k];
        i32x4* dst = (i32x4*)(xs + (int64_t)m * row_bytes);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int i = c0 + c * 64 + lane;
          if (i < nv) dst[i] = r[c];
        }
      }
    }
  }
}

// =====================================================================================================================
// Combine  (moe.py:204 / cvmm.py:481-483) and the dispatch backward gather-sum (cvmm.py:544-545)
// =====================================================================================================================

constexpr int COMBINE_SEQ_DESC = 3;   // internal: CSMOE_COMBINE_SEQ with the slots visited in descending expert order (dispatch backward)
// TO = type of the residual `add` and of `out`: T, or float with T = bf16 (the pretrain stack's fp32 residual stream under
// bf16 autocast: the combine result is rounded to bf16 like the reference's cvmm output, then added to the fp32 residual in fp32)
// TA = type of `add` alone when it differs from the output's: bf16 with TO = float is the dispatch backward of that stack, where the
// gate's bf16 gradient of x meets the experts' in the fp32 stream (csmoe_dispatch_rows_bwd_mixed)
template <typename T, int VEC, typename TO = T, typename TA = TO>
__global__ void __launch_bounds__(256) combine_kernel(const T* y, const int32_t* slot_of, const int32_t* idx, const float* w,
                                                      const T* obias, const TA* add, TO* out, int Tn, int K, int D, int mode,
                                                      const T* pre) {
#pragma clang fp contract(off)   // the sequential rule is "multiply, round, add, round": no FMA contraction
  constexpr bool O32 = !std::is_same<T, TO>::value;
  static_assert(!O32 || (std::is_same<TO, float>::value && VEC == 8), "mixed combine: bf16 rows, fp32 residual / output, 8-column chunks");
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    // lane k (< K) holds slot k; visit order is found with ballots so nothing is runtime-indexed (no scratch)
    int myslot = 0, mye = 0, rank = 0;
    float myw = 1.f;
    if (lane < K) {
      myslot = slot_of[(int64_t)t * K + lane];
      if (w) myw = w[(int64_t)t * K + lane];
      if (idx) mye = idx[(int64_t)t * K + lane];
    }
    if (mode == COMBINE_SEQ_DESC && idx) {
      // dispatch backward of the LLaVA stack: autograd runs the experts' backward nodes last-created first, i.e. in DESCENDING
      // expert order, and adds each one's dx into the leaf's buffer in x.dtype
      for (int j = 0; j < K; ++j) {
        int ej = __shfl(mye, j, 64);
        rank += (ej > mye || (ej == mye && j < lane)) ? 1 : 0;
      }
    } else if (mode != CSMOE_COMBINE_DOT && idx) {
      // experts in ascending index order (the reference loops `for i, expert in enumerate(self.experts)`)
      for (int j = 0; j < K; ++j) {
        int ej = __shfl(mye, j, 64);
        rank += (ej < mye || (ej == mye && j < lane)) ? 1 : 0;
      }
    } else {
      rank = lane;
    }
    // the d-loop is wave-uniform (ballot / shuffle inside need every lane); `live` guards the memory accesses
    for (int dbase = 0; dbase < D; dbase += 64 * VEC) {
      const int d0 = dbase + lane * VEC;
      const bool live = d0 < D;
      float acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
      if (pre && live) {                // the chain starts from this row (a gradient stream that reached the leaf earlier)
        const T* prow = pre + (int64_t)t * D + d0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = DT<T>::ld(prow + v);
      }
      // the residual / extra addend of this chunk: one vector load issued ahead of the K gathers
      float addv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) addv[v] = 0.f;
      if (add && live) {
        const TA* arow = add + (int64_t)t * D + d0;
        if constexpr (O32 && std::is_same<TA, float>::value) {
          const f32x4 ra = *(const f32x4*)arow, rb = *(const f32x4*)(arow + 4);
#pragma unroll
          for (int v = 0; v < 4; ++v) { addv[v] = ra[v]; addv[4 + v] = rb[v]; }
        } else if constexpr (VEC == 8) {
          bf16x8 r8 = *(const bf16x8*)arow;
#pragma unroll
          for (int v = 0; v < VEC; ++v) addv[v] = (float)r8[v];
        } else if constexpr (VEC == 4) {
          f32x4 r4 = *(const f32x4*)arow;
#pragma unroll
          for (int v = 0; v < VEC; ++v) addv[v] = r4[v];
        } else {
          addv[0] = DT<TA>::ld(arow);
        }
      }
      for (int kk = 0; kk < K; ++kk) {
        const int src = __ffsll((unsigned long long)__ballot(lane < K && rank == kk)) - 1;
        const int sl = __shfl(myslot, src, 64);
        const float wv = __shfl(myw, src, 64);
        float yv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) yv[v] = 0.f;
        if (live) {
          const T* row = y + (int64_t)sl * D + d0;
          if constexpr (VEC == 8) {
            bf16x8 r8 = *(const bf16x8*)row;
#pragma unroll
            for (int v = 0; v < VEC; ++v) yv[v] = (float)r8[v];
          } else if constexpr (VEC == 4) {
            f32x4 r4 = *(const f32x4*)row;
#pragma unroll
            for (int v = 0; v < VEC; ++v) yv[v] = r4[v];
          } else {
            yv[0] = DT<T>::ld(row);
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          if (mode == CSMOE_COMBINE_DOT) {
            acc[v] = fmaf(wv, yv[v], acc[v]);
          } else {
            // written out (not __fmul_rn/__fadd_rn: those inline as contractable a*b / a+b) under contract(off)
            float prod = wv * yv[v];
            if (mode == CSMOE_COMBINE_SEQ_RW) prod = DT<T>::rnd(prod);
            float sum = acc[v] + prod;
            acc[v] = DT<T>::rnd(sum);
          }
        }
      }
      if (!live) continue;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float r = acc[v];
        if (mode == CSMOE_COMBINE_DOT) r = DT<T>::rnd(r);
        if (obias) r = DT<T>::rnd(r + DT<T>::ld(obias + d0 + v));
        if (add) r = DT<TO>::rnd(r + addv[v]);
        acc[v] = r;
      }
      TO* o = out + (int64_t)t * D + d0;
      if constexpr (O32) {
        *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
      } else if constexpr (VEC == 8) {
        bf16x8 o8;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o8[v] = (bf16)acc[v];
        *(bf16x8*)o = o8;
      } else if constexpr (VEC == 4) {
        *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
      } else {
        DT<TO>::st(o, acc[0]);
      }
    }
  }
}

// K == 2, bf16 rows, D a multiple of 512 (the headline case): the same arithmetic with eight row loads (+ four residual loads) in
// flight per lane -- four 8-column chunks of both selected rows are fetched before the first add.  The generic kernel above has one
// dependent load per (chunk, k) in flight and stays at 4.6-5.0 TB/s.
template <typename TO, typename TA = TO>
__global__ void __launch_bounds__(256) combine_k2_kernel(const bf16* y, const int32_t* slot_of, const int32_t* idx, const float* w,
                                                         const bf16* obias, const TA* add, TO* out, int Tn, int D, int mode) {
#pragma clang fp contract(off)
  constexpr bool O32 = std::is_same<TO, float>::value;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    // visit order: ascending expert index for the sequential rules (ties: slot order), slot order otherwise -- wave-uniform
    int s0 = slot_of[(int64_t)t * 2], s1 = slot_of[(int64_t)t * 2 + 1];
    float w0 = w ? w[(int64_t)t * 2] : 1.f, w1 = w ? w[(int64_t)t * 2 + 1] : 1.f;
    if (mode != CSMOE_COMBINE_DOT && idx && idx[(int64_t)t * 2 + 1] < idx[(int64_t)t * 2]) {
      const int si = s0; s0 = s1; s1 = si;
      const float wi = w0; w0 = w1; w1 = wi;
    }
    const bf16* r0 = y + (int64_t)s0 * D;
    const bf16* r1 = y + (int64_t)s1 * D;
    for (int dbase = 0; dbase < D; dbase += 4 * 512) {
      bf16x8 a[4], b[4];
      float addv[4][8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d0 = dbase + c * 512 + lane * 8;
        if (d0 < D) {
          a[c] = *(const bf16x8*)(r0 + d0);
          b[c] = *(const bf16x8*)(r1 + d0);
          if (add) {
            const TA* arow = add + (int64_t)t * D + d0;
            if constexpr (std::is_same<TA, float>::value) {
              const f32x4 ra = *(const f32x4*)arow, rb = *(const f32x4*)(arow + 4);
#pragma unroll
              for (int v = 0; v < 4; ++v) { addv[c][v] = ra[v]; addv[c][4 + v] = rb[v]; }
            } else {
              const bf16x8 r8 = *(const bf16x8*)arow;
#pragma unroll
              for (int v = 0; v < 8; ++v) addv[c][v] = (float)r8[v];
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d0 = dbase + c * 512 + lane * 8;
        if (d0 >= D) continue;
        float acc[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float ya = (float)a[c][v], yb = (float)b[c][v];
          float r;
          if (mode == CSMOE_COMBINE_DOT) {
            r = DT<bf16>::rnd(fmaf(w1, yb, fmaf(w0, ya, 0.f)));
          } else {
            float p0 = w0 * ya, p1 = w1 * yb;
            if (mode == CSMOE_COMBINE_SEQ_RW) { p0 = DT<bf16>::rnd(p0); p1 = DT<bf16>::rnd(p1); }
            float s = 0.f + p0;
            s = DT<bf16>::rnd(s);
            s = s + p1;
            r = DT<bf16>::rnd(s);
          }
          if (obias) r = DT<bf16>::rnd(r + (float)obias[d0 + v]);
          if (add) r = DT<TO>::rnd(r + addv[c][v]);
          acc[v] = r;
        }
        TO* o = out + (int64_t)t * D + d0;
        if constexpr (O32) {
          *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
          *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
          bf16x8 o8;
#pragma unroll
          for (int v = 0; v < 8; ++v) o8[v] = (bf16)acc[v];
          *(bf16x8*)o = o8;
        }
      }
    }
  }
}

// combine backward: one wave per TOKEN, inner loop over its K slots (dout[t] comes from L1/L2 after the first slot, so HBM
// reads it once): dy[slot] = round(w * dout[t]), dw[t,k] = <dout[t], y[slot]>
// TG = type of the upstream gradient: T, or float with T = bf16 (fp32 residual stream: autograd casts the gradient to bf16 first,
// i.e. one rounding on load)
template <typename T, int VEC, typename TG = T>
__global__ void __launch_bounds__(256) combine_bwd_kernel(const TG* dout, const T* y, const int32_t* slot_of, const float* w, T* dy,
                                                          float* dw, int Tn, int K, int D, int round_prod) {
  constexpr bool G32 = !std::is_same<T, TG>::value;
  static_assert(!G32 || (std::is_same<TG, float>::value && VEC == 8), "mixed combine_bwd: fp32 gradient, bf16 rows, 8-column chunks");
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    for (int k = 0; k < K; ++k) {
      const int flat = t * K + k;
      const int m = slot_of[flat];
      const float wk = w ? w[flat] : 1.f;
      float dot = 0.f;
      for (int d0 = lane * VEC; d0 < D; d0 += 64 * VEC) {
        const TG* g = dout + (int64_t)t * D + d0;
        float gv[VEC], yv[VEC];
        if constexpr (VEC == 8) {
          if constexpr (G32) {
            const f32x4 ga = *(const f32x4*)g, gb = *(const f32x4*)(g + 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) { gv[v] = DT<bf16>::rnd(ga[v]); gv[4 + v] = DT<bf16>::rnd(gb[v]); }
          } else {
            bf16x8 g8 = *(const bf16x8*)g;
#pragma unroll
            for (int v = 0; v < VEC; ++v) gv[v] = (float)g8[v];
          }
          if (y) {
            bf16x8 y8 = *(const bf16x8*)(y + (int64_t)m * D + d0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) yv[v] = (float)y8[v];
          }
          bf16x8 o8;
#pragma unroll
          for (int v = 0; v < VEC; ++v) o8[v] = (bf16)(gv[v] * wk);
          *(bf16x8*)(dy + (int64_t)m * D + d0) = o8;
        } else if constexpr (VEC == 4) {
          f32x4 g4 = *(const f32x4*)g;
#pragma unroll
          for (int v = 0; v < VEC; ++v) gv[v] = g4[v];
          if (y) {
            f32x4 y4 = *(const f32x4*)(y + (int64_t)m * D + d0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) yv[v] = y4[v];
          }
          *(f32x4*)(dy + (int64_t)m * D + d0) = f32x4{gv[0] * wk, gv[1] * wk, gv[2] * wk, gv[3] * wk};
        } else {
          gv[0] = DT<TG>::ld(g);
          if (y) yv[0] = DT<T>::ld(y + (int64_t)m * D + d0);
          DT<T>::st(dy + (int64_t)m * D + d0, gv[0] * wk);
        }
        if (y) {
          if (round_prod) {        // the weights are a T-typed tensor upstream: `grad * out` is rounded element by element (moe.py:204)
#pragma unroll
            for (int v = 0; v < VEC; ++v) dot += DT<T>::rnd(gv[v] * yv[v]);
          } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) dot = fmaf(gv[v], yv[v], dot);
          }
        }
      }
      if (y && dw) {
        dot = wave_sum(dot);
        if (lane == 0) dw[flat] = round_prod ? DT<T>::rnd(dot) : dot;
      }
    }
  }
}

// =====================================================================================================================
// Per-expert column sums (bias gradients).  grid = (ceil(N/256), E); 4 waves split the rows, lane = 4 columns.
// =====================================================================================================================
template <typename T, typename TOut>
__global__ void __launch_bounds__(256) colsum_kernel(const T* G, int64_t ldg, const int32_t* offsets, int single_M, int N,
                                                     void* const* out_ptrs, void* single_out) {
  // generic (any N / alignment): lane = 4 columns, 4 waves split the rows
  __shared__ float part[4][256];
  const int e = blockIdx.y;
  const int r0 = offsets ? offsets[e] : 0, r1 = offsets ? offsets[e + 1] : single_M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + lane * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + wave; r < r1; r += 4) {
    const T* row = G + (int64_t)r * ldg + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < N) s[j] += DT<T>::ld(row + j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) part[wave][lane * 4 + j] = s[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    TOut* o = (TOut*)(out_ptrs ? out_ptrs[e] : single_out);
    DT<TOut>::st(o + c, v);
  }
}

// bf16, N % 8 == 0, 16-B aligned rows: lane = 8 columns (16 B), a wave covers 512 columns, 4 waves split the rows and keep
// 4 row loads in flight each
template <typename TOut>
__global__ void __launch_bounds__(256) colsum_bf16x8_kernel(const bf16* G, int64_t ldg, const int32_t* offsets, int single_M, int N,
                                                            void* const* out_ptrs, void* single_out) {
  __shared__ float part[4][512];
  const int e = blockIdx.y;
  const int r0 = offsets ? offsets[e] : 0, r1 = offsets ? offsets[e + 1] : single_M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {
      bf16x8 v0 = *(const bf16x8*)(G + (int64_t)r * ldg + c0);
      bf16x8 v1 = *(const bf16x8*)(G + (int64_t)(r + 4) * ldg + c0);
      bf16x8 v2 = *(const bf16x8*)(G + (int64_t)(r + 8) * ldg + c0);
      bf16x8 v3 = *(const bf16x8*)(G + (int64_t)(r + 12) * ldg + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += ((float)v0[j] + (float)v1[j]) + ((float)v2[j] + (float)v3[j]);
    }
    for (; r < r1; r += 4) {
      bf16x8 v0 = *(const bf16x8*)(G + (int64_t)r * ldg + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[wave][lane * 8 + j] = s[j];
  __syncthreads();
  TOut* o = (TOut*)(out_ptrs ? out_ptrs[e] : single_out);
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int c = blockIdx.x * 512 + i;
    if (c < N) DT<TOut>::st(o + c, part[0][i] + part[1][i] + part[2][i] + part[3][i]);
  }
}

// fp32 rows, very wide and few (the fp32 partials of a split-K weight gradient: 2-4 rows of millions of columns): a thread owns 8
// consecutive columns (two 16-byte loads per row), four rows in flight, one 16 / 32-byte store.  Used when the column blocks alone
// fill the chip.
template <typename TOut>
__global__ void __launch_bounds__(256) colsum_f32_wide_kernel(const float* G, int64_t ldg, const int32_t* offsets, int single_M, int N,
                                                              void* const* out_ptrs, void* single_out) {
  const int e = blockIdx.y;
  const int r0 = offsets ? offsets[e] : 0, r1 = offsets ? offsets[e + 1] : single_M;
  const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (c0 >= N) return;
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* p = G + (int64_t)(r + u) * ldg + c0;
      a[u] = *(const f32x4*)p;
      b[u] = *(const f32x4*)(p + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { sa += a[u]; sb += b[u]; }
  }
  for (; r < r1; ++r) {
    const float* p = G + (int64_t)r * ldg + c0;
    sa += *(const f32x4*)p;
    sb += *(const f32x4*)(p + 4);
  }
  TOut* o = (TOut*)(out_ptrs ? out_ptrs[e] : single_out) + c0;
  if constexpr (std::is_same<TOut, float>::value) {
    *(f32x4*)o = sa;
    *(f32x4*)(o + 4) = sb;
  } else {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (bf16)sa[j]; v[4 + j] = (bf16)sb[j]; }
    *(bf16x8*)o = v;
  }
}

// =====================================================================================================================
// Competition affinity: aff[r] = mean_d softplus(y[r,d])  (competesmoe.py:242) and its backward
// =====================================================================================================================
template <bool PRECISE> __device__ __forceinline__ float softplusf_(float x) { return softplus_rt(x, PRECISE); }
template <bool PRECISE> __device__ __forceinline__ float sp_sigmoidf_(float x) { return softplus_grad_rt(x, PRECISE); }

template <typename T> struct Vec16;
template <> struct Vec16<bf16> { static constexpr int N = 8; typedef bf16x8 V; };
template <> struct Vec16<float> { static constexpr int N = 4; typedef f32x4 V; };

// wave per row, 16-byte accesses when VEC (D % N == 0, 16-byte aligned rows).  TA = dtype of the affinities: T (the LLaVA stack:
// softplus and mean are x.dtype tensor ops, every softplus value is rounded to T before the mean) or float around bf16 rows (the
// pretrain stack under CUDA autocast: F.softplus is an fp32-policy op, so softplus, mean and everything after are fp32).
template <typename T, typename TA, bool VEC, bool PRECISE>
__global__ void __launch_bounds__(256) softplus_mean_kernel(const T* y, TA* aff, int R, int D) {
  typedef typename Vec16<T>::V V;
  constexpr int N = Vec16<T>::N;
  constexpr bool RND = std::is_same<T, TA>::value;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave_g; r < R; r += nw) {
    float s = 0.f;
    if constexpr (VEC) {
      for (int d = lane * N; d < D; d += 64 * N) {
        const V v = *(const V*)(y + (int64_t)r * D + d);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float sp = softplusf_<PRECISE>((float)v[j]);
          s += RND ? DT<T>::rnd(sp) : sp;
        }
      }
    } else {
      for (int d = lane; d < D; d += 64) {
        const float sp = softplusf_<PRECISE>(DT<T>::ld(y + (int64_t)r * D + d));
        s += RND ? DT<T>::rnd(sp) : sp;
      }
    }
    s = wave_sum(s);
    if (lane == 0) DT<TA>::st(aff + r, s / (float)D);
  }
}

// TA == T:     dy[r,d] = round(round(daff[r] / D) * sigmoid(y[r,d])) (+ dy_add[r,d])
// TA == float: dy[r,d] = round(daff[r] / D * sigmoid(y[r,d]))        (+ dy_add[r,d])   (fp32 chain, one cast: autograd of .float())
template <typename T, typename TA, bool VEC, bool PRECISE>
__global__ void __launch_bounds__(256) softplus_mean_bwd_kernel(const T* y, const TA* daff, const T* dy_add, T* dy, int R, int D) {
  typedef typename Vec16<T>::V V;
  constexpr int N = Vec16<T>::N;
  constexpr bool RND = std::is_same<T, TA>::value;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int r = wave_g; r < R; r += nw) {
    const float g0 = DT<TA>::ld(daff + r) / (float)D;
    const float g = RND ? DT<T>::rnd(g0) : g0;
    if constexpr (VEC) {
      for (int d = lane * N; d < D; d += 64 * N) {
        const int64_t o = (int64_t)r * D + d;
        const V v = *(const V*)(y + o);
        V a;
        if (dy_add) a = *(const V*)(dy_add + o);
        V out;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float t = DT<T>::rnd(g * sp_sigmoidf_<PRECISE>((float)v[j]));
          if (dy_add) t = DT<T>::rnd(t + (float)a[j]);
          out[j] = (T)t;
        }
        *(V*)(dy + o) = out;
      }
    } else {
      for (int d = lane; d < D; d += 64) {
        const int64_t o = (int64_t)r * D + d;
        float v = DT<T>::rnd(g * sp_sigmoidf_<PRECISE>(DT<T>::ld(y + o)));
        if (dy_add) v = DT<T>::rnd(v + DT<T>::ld(dy_add + o));
        DT<T>::st(dy + o, v);
      }
    }
  }
}

// =====================================================================================================================
// Diversity loss of the competition step (moe.py:133-171, competesmoe.py:180-218): per token the sum over ordered pairs i != j
// of the cosine similarity of the K selected experts' outputs, <y_i / max(|y_i|, 1e-12), y_j / max(|y_j|, 1e-12)> in fp32
// (F.normalize + bmm + zeroed diagonal upstream: 32768 batched 2x4096x2 matmuls, 15 ms through rocBLAS at the headline).
// One wave per token, K <= 8; forward writes the per-token sum (the caller sums T values and divides by T*K*K), backward
// recomputes norms and dots and writes dy = g * 2/|y_k| * (s_k - <n_k, s_k> n_k), s_k = sum_{j != k} n_j, in x.dtype.
// =====================================================================================================================
constexpr int PC_MAXK = 8;

template <typename T, bool BWD, bool VEC>
__global__ void __launch_bounds__(256) pair_cosine_kernel(const T* y, float* tok_loss, const float* gscale, T* dy, int Tn, int K, int D) {
  typedef typename Vec16<T>::V V;
  constexpr int N = VEC ? Vec16<T>::N : 1;         // elements per lane and step (16-byte accesses when VEC)
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    const T* base = y + (int64_t)t * K * D;
    float dot[PC_MAXK][PC_MAXK];                 // upper triangle incl. diagonal (norms)
#pragma unroll
    for (int i = 0; i < PC_MAXK; ++i)
#pragma unroll
      for (int j = 0; j < PC_MAXK; ++j) dot[i][j] = 0.f;
    for (int d = lane * N; d < D; d += 64 * N) {
      float v[PC_MAXK][N];
#pragma unroll
      for (int i = 0; i < PC_MAXK; ++i) {
        if (i < K) {
          if constexpr (VEC) {
            const V x = *(const V*)(base + (int64_t)i * D + d);
#pragma unroll
            for (int e = 0; e < N; ++e) v[i][e] = (float)x[e];
          } else {
            v[i][0] = DT<T>::ld(base + (int64_t)i * D + d);
          }
        } else {
#pragma unroll
          for (int e = 0; e < N; ++e) v[i][e] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < PC_MAXK; ++i)
#pragma unroll
        for (int j = i; j < PC_MAXK; ++j)
          if (j < K) {
#pragma unroll
            for (int e = 0; e < N; ++e) dot[i][j] += v[i][e] * v[j][e];
          }
    }
    float inv[PC_MAXK];
#pragma unroll
    for (int i = 0; i < PC_MAXK; ++i)
#pragma unroll
      for (int j = i; j < PC_MAXK; ++j)
        if (j < K) dot[i][j] = wave_sum(dot[i][j]);
#pragma unroll
    for (int i = 0; i < PC_MAXK; ++i) inv[i] = i < K ? 1.f / fmaxf(sqrtf(dot[i][i]), 1e-12f) : 0.f;
    if constexpr (!BWD) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PC_MAXK; ++i)
#pragma unroll
        for (int j = i + 1; j < PC_MAXK; ++j)
          if (j < K) s += 2.f * dot[i][j] * inv[i] * inv[j];
      if (lane == 0) tok_loss[t] = s;
    } else {
      const float g2 = 2.f * gscale[0];
      // <n_k, s_k> = sum_{j != k} cos(k, j)
      float ns[PC_MAXK];
#pragma unroll
      for (int k = 0; k < PC_MAXK; ++k) {
        ns[k] = 0.f;
#pragma unroll
        for (int j = 0; j < PC_MAXK; ++j)
          if (j < K && j != k) ns[k] += (j > k ? dot[k][j] : dot[j][k]) * inv[k] * inv[j];
      }
      T* dbase = dy + (int64_t)t * K * D;
      for (int d = lane * N; d < D; d += 64 * N) {
        float n[PC_MAXK][N], tot[N];
#pragma unroll
        for (int e = 0; e < N; ++e) tot[e] = 0.f;
#pragma unroll
        for (int i = 0; i < PC_MAXK; ++i) {
          if (i < K) {
            if constexpr (VEC) {
              const V x = *(const V*)(base + (int64_t)i * D + d);
#pragma unroll
              for (int e = 0; e < N; ++e) n[i][e] = (float)x[e] * inv[i];
            } else {
              n[i][0] = DT<T>::ld(base + (int64_t)i * D + d) * inv[i];
            }
#pragma unroll
            for (int e = 0; e < N; ++e) tot[e] += n[i][e];
          }
        }
#pragma unroll
        for (int k = 0; k < PC_MAXK; ++k)
          if (k < K) {
            if constexpr (VEC) {
              V o;
#pragma unroll
              for (int e = 0; e < N; ++e) o[e] = (T)(g2 * inv[k] * ((tot[e] - n[k][e]) - ns[k] * n[k][e]));
              *(V*)(dbase + (int64_t)k * D + d) = o;
            } else {
              DT<T>::st(dbase + (int64_t)k * D + d, g2 * inv[k] * ((tot[0] - n[k][0]) - ns[k] * n[k][0]));
            }
          }
      }
    }
  }
}

// =====================================================================================================================
// Skinny gate projection for few experts (E <= 16; the reference's LLaVA configs use 4): the 128x128 MFMA tiles of the GEMM path
// waste >= 7/8 of their columns there (0.09 / 0.17 ms for logits / weight gradient at [12800, 1152] x 4 experts, next to a
// 3 ms step).  Plain HBM-bound row passes instead: logits = x @ Wg^T, dx = dlogits @ Wg, dWg = dlogits^T @ x.
// =====================================================================================================================
constexpr int GS_MAXE = 16;

template <typename T>
__global__ void __launch_bounds__(256) gate_small_fwd_kernel(const T* x, const T* wg, T* logits, int Tn, int D, int E) {
  typedef typename Vec16<T>::V V;
  constexpr int N = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    float acc[GS_MAXE];
#pragma unroll
    for (int e = 0; e < GS_MAXE; ++e) acc[e] = 0.f;
    for (int d = lane * N; d < D; d += 64 * N) {
      const V xv = *(const V*)(x + (int64_t)t * D + d);
#pragma unroll
      for (int e = 0; e < GS_MAXE; ++e) {
        if (e < E) {
          const V wv = *(const V*)(wg + (int64_t)e * D + d);
#pragma unroll
          for (int j = 0; j < N; ++j) acc[e] += (float)xv[j] * (float)wv[j];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < GS_MAXE; ++e)
      if (e < E) {
        const float sum = wave_sum(acc[e]);
        if (lane == 0) DT<T>::st(logits + (int64_t)t * E + e, sum);
      }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gate_small_dx_kernel(const T* dl, const T* wg, T* dx, int Tn, int D, int E) {
  typedef typename Vec16<T>::V V;
  constexpr int N = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    float g[GS_MAXE];
#pragma unroll
    for (int e = 0; e < GS_MAXE; ++e) g[e] = e < E ? DT<T>::ld(dl + (int64_t)t * E + e) : 0.f;
    for (int d = lane * N; d < D; d += 64 * N) {
      float o[N];
#pragma unroll
      for (int j = 0; j < N; ++j) o[j] = 0.f;
#pragma unroll
      for (int e = 0; e < GS_MAXE; ++e) {
        if (e < E) {
          const V wv = *(const V*)(wg + (int64_t)e * D + d);
#pragma unroll
          for (int j = 0; j < N; ++j) o[j] += g[e] * (float)wv[j];
        }
      }
      V ov;
#pragma unroll
      for (int j = 0; j < N; ++j) ov[j] = (T)o[j];
      *(V*)(dx + (int64_t)t * D + d) = ov;
    }
  }
}

// grid (ceil(D / (64*N)), nranges), block 64: a lane owns one 16-byte column chunk for all E experts over the rows of its range;
// partial[range][e][D] fp32, reduced over the ranges by a column-sum launch (deterministic)
template <typename T>
__global__ void __launch_bounds__(64) gate_small_dw_kernel(const T* dl, const T* x, float* partial, int Tn, int D, int E, int rows_per) {
  typedef typename Vec16<T>::V V;
  constexpr int N = Vec16<T>::N;
  const int d = (blockIdx.x * 64 + threadIdx.x) * N;
  const int r0 = blockIdx.y * rows_per, r1 = min(Tn, r0 + rows_per);
  float acc[GS_MAXE][N];
#pragma unroll
  for (int e = 0; e < GS_MAXE; ++e)
#pragma unroll
    for (int j = 0; j < N; ++j) acc[e][j] = 0.f;
  if (d < D) {
    for (int t = r0; t < r1; ++t) {
      const V xv = *(const V*)(x + (int64_t)t * D + d);
#pragma unroll
      for (int e = 0; e < GS_MAXE; ++e) {
        if (e < E) {
          const float g = DT<T>::ld(dl + (int64_t)t * E + e);
#pragma unroll
          for (int j = 0; j < N; ++j) acc[e][j] += g * (float)xv[j];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < GS_MAXE; ++e)
      if (e < E)
#pragma unroll
        for (int j = 0; j < N; ++j) partial[((int64_t)blockIdx.y * E + e) * D + d + j] = acc[e][j];
  }
}

// =====================================================================================================================
// Router auxiliary losses of the LLaVA stack in two launches (+ one for the backward) instead of ~35 tiny tensor ops:
//   balance (moe.py:90-110) = mean_{b,e}( mean_n softmax[b,n,e] * mean_n [idx[b,n,0] == e] ) * E^2          (fp32)
//   z-loss  (moe.py:71-88)  = mean_t( logsumexp(logits[t,:])^2 ), evaluated on x.dtype tensors like the reference: the
//                             logsumexp, its square and the mean are each rounded to x.dtype
// logsumexp(logits[t]) = logits[t,i1] - log(softmax[t,i1]) with i1 the top-1 expert (softmax is the fp32 softmax of the same
// logits, so no second reduction over E is needed).  Deterministic: per-workgroup partial rows, then one reducing workgroup.
// =====================================================================================================================
constexpr int RA_MAXQ = 4;                        // E <= 1024: up to 4 column chunks of 256

// Tokens per workgroup: about 512 workgroups over the chip, at least 32 tokens each.
static inline int ra_chunk(int B, int N) {
  const int64_t T = (int64_t)B * N;
  int c = (int)((T + 511) / 512);
  c = (c + 31) / 32 * 32;
  return c < 32 ? 32 : c;
}

// threads = (row r, column col): cols = 2^lc columns (E rounded up to a power of two, at most 256) x 256/cols token rows, so that
// consecutive lanes read consecutive experts of consecutive tokens (coalesced for any E, no idle lanes at E = 4).
template <typename T>
__global__ void __launch_bounds__(256) router_aux_partial_kernel(const T* logits, const float* sm, const int32_t* idx, float* lse_out,
                                                                 float* partial, int N, int E, int K, int nchunk, int chunk, int lc) {
  __shared__ float red[2][RA_MAXQ][256];
  __shared__ float zred[256];
  const int b = blockIdx.x / nchunk, c = blockIdx.x % nchunk;
  const int tid = threadIdx.x, cols = 1 << lc, rows = 256 >> lc;
  const int col = tid & (cols - 1), r = tid >> lc;
  const int n0 = c * chunk, n1 = min(N, n0 + chunk);
  float px[RA_MAXQ], dn[RA_MAXQ], zs = 0.f;
#pragma unroll
  for (int q = 0; q < RA_MAXQ; ++q) { px[q] = 0.f; dn[q] = 0.f; }
  for (int n = n0 + r; n < n1; n += rows) {
    const int64_t t = (int64_t)b * N + n;
    const int i1 = idx[t * K];
#pragma unroll
    for (int q = 0; q < RA_MAXQ; ++q) {
      const int e = q * cols + col;
      if (e < E) {
        px[q] += sm[t * E + e];
        dn[q] += (e == i1) ? 1.f : 0.f;
      }
    }
    if (logits && col == 0) {
      const float l = DT<T>::rnd(DT<T>::ld(logits + t * E + i1) - __logf(sm[t * E + i1]));
      lse_out[t] = l;
      zs += DT<T>::rnd(l * l);
    }
  }
#pragma unroll
  for (int q = 0; q < RA_MAXQ; ++q) { red[0][q][tid] = px[q]; red[1][q][tid] = dn[q]; }
  zred[tid] = zs;
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * (2 * E + 1);
  for (int o = tid; o < 2 * E + 1; o += 256) {
    float acc = 0.f;
    if (o < 2 * E) {
      const int which = o >= E ? 1 : 0, e = o - which * E;
      const int q = e >> lc, cl = e & (cols - 1);
      for (int rr = 0; rr < rows; ++rr) acc += red[which][q][(rr << lc) + cl];
    } else {
      for (int rr = 0; rr < rows; ++rr) acc += zred[rr << lc];
    }
    out[o] = acc;
  }
}

// one workgroup of 1024 threads: partial rows -> dens[b,e] (kept for the backward), balance, z.  One (b, e) output per thread and
// eight chunk rows in flight per thread: the launch is a chain of dependent loads, not bandwidth.
template <typename T>
__global__ void __launch_bounds__(1024) router_aux_final_kernel(const float* partial, float* dens, float* out2, int B, int N, int E,
                                                                int nchunk, int has_z) {
  __shared__ float acc[1024];
  const int tid = threadIdx.x;
  const int64_t rs = 2 * E + 1;
  float bal = 0.f, z = 0.f;
  for (int o = tid; o < B * E; o += 1024) {
    const int b = o / E, e = o - b * E;
    const float* row = partial + (int64_t)b * nchunk * rs + e;
    float p = 0.f, d = 0.f;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
      float pv[8], dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { pv[u] = row[(c + u) * rs]; dv[u] = row[(c + u) * rs + E]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { p += pv[u]; d += dv[u]; }
    }
    for (; c < nchunk; ++c) { p += row[c * rs]; d += row[c * rs + E]; }
    p /= (float)N;
    d /= (float)N;
    dens[o] = d;
    bal += p * d;
  }
  if (has_z)
    for (int o = tid; o < B * nchunk; o += 1024) z += partial[(int64_t)o * rs + 2 * E];
  acc[tid] = bal;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (tid < st) acc[tid] += acc[tid + st]; __syncthreads(); }
  const float bal_tot = acc[0];
  __syncthreads();
  acc[tid] = z;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (tid < st) acc[tid] += acc[tid + st]; __syncthreads(); }
  if (tid == 0) {
    out2[0] = bal_tot / (float)(B * E) * (float)E * (float)E;
    out2[1] = has_z ? DT<T>::rnd(acc[0] / (float)((int64_t)B * N)) : 0.f;
  }
}

// dsoftmax[t,e] = g_bal * E^2 / (B*E) * dens[b,e] / N   (fp32);   dlogits[t,e] = g_z * 2 * lse[t] / T * softmax[t,e]   (x.dtype)
template <typename T>
__global__ void __launch_bounds__(256) router_aux_bwd_kernel(const float* sm, const float* dens, const float* lse, const float* g_bal,
                                                             const float* g_z, float* dsm, T* dlogits, int B, int N, int E) {
  const int64_t total = (int64_t)B * N * E;
  const float kb = g_bal ? g_bal[0] * (float)E / (float)B / (float)N : 0.f;
  const float kz = (g_z && dlogits) ? g_z[0] * 2.f / (float)((int64_t)B * N) : 0.f;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int64_t t = o / E;
    const int e = (int)(o - t * E);
    const int b = (int)(t / N);
    if (dsm) dsm[o] = kb * dens[b * E + e];
    if (dlogits) DT<T>::st(dlogits + o, kz * lse[t] * sm[o]);
  }
}

// =====================================================================================================================
// Experts dealt to the 8 XCDs by row count for the persistent weight-gradient kernel: rank r (0 = most rows, ties by index) goes
// to XCD (r % 8) in snake order; order[x * slots + k] = the k-th expert of XCD x (slots = ceil(E / 8)), -1 for an empty slot.
// One workgroup; O(E^2 / 64) per lane.
// =====================================================================================================================
__global__ void __launch_bounds__(256) expert_order_kernel(const int32_t* offsets, int E, int32_t* order) {
  const int slots = (E + 7) >> 3;
  for (int o = threadIdx.x; o < 8 * slots + 1; o += 256) order[o] = -1;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    const int c = offsets[e + 1] - offsets[e];
    int r = 0;
    for (int j = 0; j < E; ++j) {
      const int cj = offsets[j + 1] - offsets[j];
      r += (cj > c || (cj == c && j < e)) ? 1 : 0;
    }
    const int k = r >> 3, pos = r & 7;
    const int x = (k & 1) ? 7 - pos : pos;
    order[x * slots + k] = e;
  }
}

// grid-stride row kernels: 4 rows (waves) per workgroup; caps of 2048..16384 workgroups measure the same (tools/hbm_bench.py)
inline int stride_grid(int rows) { return std::max(1, std::min((rows + 3) / 4, 4096)); }

// combine backward for K == 2, bf16 rows, D a multiple of 512: the upstream-gradient chunk and both y chunks of four 8-column
// chunks (12 loads) are in flight per lane before the first store; same per-lane accumulation order of the dot products as the
// generic kernel (bit-identical dw).
template <typename TG>
__global__ void __launch_bounds__(256) combine_bwd_k2_kernel(const TG* dout, const bf16* y, const int32_t* slot_of, const float* w,
                                                             bf16* dy, float* dw, int Tn, int D) {
  constexpr bool G32 = std::is_same<TG, float>::value;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave_g; t < Tn; t += nw) {
    const int m0 = slot_of[(int64_t)t * 2], m1 = slot_of[(int64_t)t * 2 + 1];
    const float w0 = w ? w[(int64_t)t * 2] : 1.f, w1 = w ? w[(int64_t)t * 2 + 1] : 1.f;
    float dot0 = 0.f, dot1 = 0.f;
    for (int dbase = 0; dbase < D; dbase += 4 * 512) {
      float gv[4][8];
      bf16x8 ya[4], yb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d0 = dbase + c * 512 + lane * 8;
        if (d0 < D) {
          const TG* g = dout + (int64_t)t * D + d0;
          if constexpr (G32) {
            const f32x4 ga = *(const f32x4*)g, gb = *(const f32x4*)(g + 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) { gv[c][v] = DT<bf16>::rnd(ga[v]); gv[c][4 + v] = DT<bf16>::rnd(gb[v]); }
          } else {
            const bf16x8 g8 = *(const bf16x8*)g;
#pragma unroll
            for (int v = 0; v < 8; ++v) gv[c][v] = (float)g8[v];
          }
          if (y) {
            ya[c] = *(const bf16x8*)(y + (int64_t)m0 * D + d0);
            yb[c] = *(const bf16x8*)(y + (int64_t)m1 * D + d0);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d0 = dbase + c * 512 + lane * 8;
        if (d0 >= D) continue;
        bf16x8 o0, o1;
#pragma unroll
        for (int v = 0; v < 8; ++v) { o0[v] = (bf16)(gv[c][v] * w0); o1[v] = (bf16)(gv[c][v] * w1); }
        *(bf16x8*)(dy + (int64_t)m0 * D + d0) = o0;
        *(bf16x8*)(dy + (int64_t)m1 * D + d0) = o1;
        if (y) {
#pragma unroll
          for (int v = 0; v < 8; ++v) { dot0 = fmaf(gv[c][v], (float)ya[c][v], dot0); dot1 = fmaf(gv[c][v], (float)yb[c][v], dot1); }
        }
      }
    }
    if (y && dw) {
      dot0 = wave_sum(dot0);
      dot1 = wave_sum(dot1);
      if (lane == 0) { dw[(int64_t)t * 2] = dot0; dw[(int64_t)t * 2 + 1] = dot1; }
    }
  }
}

// Every expert's row range cut into P chunks (the pseudo-experts of a split-K weight gradient / a chunked column sum):
// out[e * P + j] = min(off[e] + roundup(cnt_e * j / P, align), off[e + 1]),  out[E * P] = off[E].
__global__ void __launch_bounds__(256) chunk_offsets_kernel(const int32_t* offsets, int E, int P, int align, int32_t* out) {
  for (int o = blockIdx.x * 256 + threadIdx.x; o <= E * P; o += gridDim.x * 256) {
    if (o == E * P) { out[o] = offsets[E]; continue; }
    const int e = o / P, j = o - e * P;
    const int lo = offsets[e], hi = offsets[e + 1];
    int64_t st = ((int64_t)(hi - lo) * j) / P;
    st = (st + align - 1) / align * align;
    out[o] = (int)min((int64_t)hi, (int64_t)lo + st);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host launchers
#define SEL_DISPATCH(KERNEL, ...)                                                                 \
  do {                                                                                            \
    int vpl = (E + 63) / 64;                                                                      \
    dim3 grid((T + 3) / 4), block(256);                                                           \
    if (vpl <= 1) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, st, __VA_ARGS__);               \
    else if (vpl <= 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, st, __VA_ARGS__);          \
    else if (vpl <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, st, __VA_ARGS__);          \
    else if (vpl <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, block, 0, st, __VA_ARGS__);          \
    else hipLaunchKernelGGL((KERNEL<16>), grid, block, 0, st, __VA_ARGS__);                       \
  } while (0)

int k_router_select(const void* scores, int dtype, int T, int E, int K, int mode, int round_sum_bf16, float sel_param,
                    float* softmax, int32_t* idx, float* w, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  SEL_DISPATCH(router_select_kernel, scores, dtype, T, E, K, mode, round_sum_bf16, sel_param, softmax, idx, w);
  CSMOE_CHECK_LAUNCH("router_select");
  return CSMOE_OK;
}

int k_router_select_bwd(const void* scores, int dtype, int T, int E, int K, int mode, int round_sum_bf16, float sel_param,
                        const float* softmax, const int32_t* idx, const float* w, const float* dw, const float* dsoftmax,
                        void* dscores, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  SEL_DISPATCH(router_select_bwd_kernel, scores, dtype, T, E, K, mode, round_sum_bf16, sel_param, softmax, idx, w, dw, dsoftmax,
               dscores);
  CSMOE_CHECK_LAUNCH("router_select_bwd");
  return CSMOE_OK;
}

// threads per expert in bin_scan_kernel: as many as 1024 threads give, not more than there are blocks
static int scan_parts(int nb, int E) {
  int P = E <= 512 ? 1024 / E : 1;
  if (P > nb) P = nb;
  return P < 1 ? 1 : P;
}

int64_t k_bin_workspace_bytes(int n, int E) {
  int64_t nb = (n + BIN_CHUNK - 1) / BIN_CHUNK;
  return 2 * nb * (int64_t)E * 4 + 64;
}

int k_bin_tokens(const int32_t* idx, int n, int E, int32_t* counts, int32_t* offsets, int32_t* perm, int32_t* slot_of,
                 void* workspace, hipStream_t st) {
  int nb = (n + BIN_CHUNK - 1) / BIN_CHUNK;
  if (nb == 0) nb = 1;
  int32_t* block_hist = (int32_t*)workspace;
  int32_t* block_base = block_hist + (int64_t)nb * E;
  int ebits = 0;
  while ((1 << ebits) < E) ++ebits;
  hipLaunchKernelGGL(bin_hist_kernel, dim3(nb), dim3(256), E * 4, st, idx, n, E, block_hist);
  { const int P = scan_parts(nb, E); hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), (P + 1) * E * 4, st, block_hist, nb, E, P, counts, offsets, block_base); }
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(nb), dim3(256), 5 * E * 4, st, idx, n, E, ebits, BIN_CHUNK, block_base, perm, slot_of);
  CSMOE_CHECK_LAUNCH("bin_tokens");
  return CSMOE_OK;
}

// The same sort from a histogram somebody else built (the one-pass router, router_fused.hip): block b of the histogram counts ids
// [b*chunk, (b+1)*chunk); scan + scatter only.
int k_bin_tokens_hist(const int32_t* idx, int n, int E, int chunk, const int32_t* block_hist, int32_t* block_base, int32_t* counts,
                      int32_t* offsets, int32_t* perm, int32_t* slot_of, hipStream_t st) {
  int nb = (n + chunk - 1) / chunk;
  if (nb == 0) nb = 1;
  int ebits = 0;
  while ((1 << ebits) < E) ++ebits;
  { const int P = scan_parts(nb, E); hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), (P + 1) * E * 4, st, block_hist, nb, E, P, counts, offsets, block_base); }
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(nb), dim3(256), 5 * E * 4, st, idx, n, E, ebits, chunk, block_base, perm, slot_of);
  CSMOE_CHECK_LAUNCH("bin_tokens_hist");
  return CSMOE_OK;
}

int k_dispatch_rows(const void* x, const int32_t* perm, int K, void* xs, int n, int row_bytes, int vec_ok, hipStream_t st) {
  if (n == 0) return CSMOE_OK;
  hipLaunchKernelGGL(dispatch_rows_kernel, dim3(stride_grid(n)), dim3(256), 0, st, (const char*)x, perm, K, (char*)xs, n,
                     row_bytes, vec_ok);
  CSMOE_CHECK_LAUNCH("dispatch_rows");
  return CSMOE_OK;
}

int k_dispatch_tokens(const void* x, const int32_t* slot_of, int K, void* xs, int T, int row_bytes, int vec_ok, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  static const bool plain = getenv("CSMOE_DISPATCH_PLAIN") != nullptr;       // A/B switch
  if (vec_ok && !plain)
    hipLaunchKernelGGL(dispatch_tokens_reg_kernel, dim3(stride_grid(T)), dim3(256), 0, st, (const char*)x, slot_of, K, (char*)xs, T,
                       row_bytes);
  else
    hipLaunchKernelGGL(dispatch_tokens_kernel, dim3(stride_grid(T)), dim3(256), 0, st, (const char*)x, slot_of, K, (char*)xs, T,
                       row_bytes, vec_ok);
  CSMOE_CHECK_LAUNCH("dispatch_tokens");
  return CSMOE_OK;
}

int k_combine(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const void* obias, const void* add,
              void* out, int T, int K, int D, int dtype, int mode, hipStream_t st, const void* pre) {
  if (T == 0) return CSMOE_OK;
  dim3 grid(stride_grid(T)), block(256);
  const bool al = (((uintptr_t)y | (uintptr_t)out | (uintptr_t)obias | (uintptr_t)add | (uintptr_t)pre) & 15) == 0;
  static const bool generic_only = getenv("CSMOE_COMBINE_GENERIC") != nullptr;       // A/B switch
  if (dtype == CSMOE_BF16) {
    if (K == 2 && D % 512 == 0 && al && !generic_only && !pre && mode != COMBINE_SEQ_DESC)
      hipLaunchKernelGGL((combine_k2_kernel<bf16>), grid, block, 0, st, (const bf16*)y, slot_of, idx, w, (const bf16*)obias,
                         (const bf16*)add, (bf16*)out, T, D, mode);
    else if (D % 8 == 0 && al)
      hipLaunchKernelGGL((combine_kernel<bf16, 8>), grid, block, 0, st, (const bf16*)y, slot_of, idx, w, (const bf16*)obias,
                         (const bf16*)add, (bf16*)out, T, K, D, mode, (const bf16*)pre);
    else
      hipLaunchKernelGGL((combine_kernel<bf16, 1>), grid, block, 0, st, (const bf16*)y, slot_of, idx, w, (const bf16*)obias,
                         (const bf16*)add, (bf16*)out, T, K, D, mode, (const bf16*)pre);
  } else {
    if (D % 4 == 0 && al)
      hipLaunchKernelGGL((combine_kernel<float, 4>), grid, block, 0, st, (const float*)y, slot_of, idx, w, (const float*)obias,
                         (const float*)add, (float*)out, T, K, D, mode, (const float*)pre);
    else
      hipLaunchKernelGGL((combine_kernel<float, 1>), grid, block, 0, st, (const float*)y, slot_of, idx, w, (const float*)obias,
                         (const float*)add, (float*)out, T, K, D, mode, (const float*)pre);
  }
  CSMOE_CHECK_LAUNCH("combine");
  return CSMOE_OK;
}

// dispatch backward into an fp32 stream: bf16 rows, bf16 `add`, fp32 output; D % 8 == 0 and 16-byte alignment checked by the caller
int k_dispatch_rows_bwd_mixed(const void* dxs, const int32_t* slot_of, int K, const void* add, float* dx, int T, int D, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  static const bool generic_only = getenv("CSMOE_COMBINE_GENERIC") != nullptr;
  if (K == 2 && D % 512 == 0 && !generic_only)
    hipLaunchKernelGGL((combine_k2_kernel<float, bf16>), dim3(stride_grid(T)), dim3(256), 0, st, (const bf16*)dxs, slot_of,
                       (const int32_t*)nullptr, (const float*)nullptr, (const bf16*)nullptr, (const bf16*)add, dx, T, D,
                       (int)CSMOE_COMBINE_DOT);
  else
    hipLaunchKernelGGL((combine_kernel<bf16, 8, float, bf16>), dim3(stride_grid(T)), dim3(256), 0, st, (const bf16*)dxs, slot_of,
                       (const int32_t*)nullptr, (const float*)nullptr, (const bf16*)nullptr, (const bf16*)add, dx, T, K, D,
                       (int)CSMOE_COMBINE_DOT, (const bf16*)nullptr);
  CSMOE_CHECK_LAUNCH("dispatch_rows_bwd_mixed");
  return CSMOE_OK;
}

int k_combine_bwd(const void* dout, const void* y, const int32_t* perm, const float* w, void* dy, float* dw, int n, int K, int D,
                  int dtype, int round_prod, hipStream_t st) {
  // `perm` here is slot_of (token-major traversal); n = T
  if (n == 0) return CSMOE_OK;
  dim3 grid(stride_grid(n)), block(256);
  const bool al = (((uintptr_t)y | (uintptr_t)dout | (uintptr_t)dy) & 15) == 0;
  static const bool generic_only = getenv("CSMOE_COMBINE_GENERIC") != nullptr;       // A/B switch
  if (dtype == CSMOE_BF16) {
    if (K == 2 && D % 512 == 0 && al && !generic_only && !round_prod)
      hipLaunchKernelGGL((combine_bwd_k2_kernel<bf16>), grid, block, 0, st, (const bf16*)dout, (const bf16*)y, perm, w, (bf16*)dy,
                         dw, n, D);
    else if (D % 8 == 0 && al)
      hipLaunchKernelGGL((combine_bwd_kernel<bf16, 8>), grid, block, 0, st, (const bf16*)dout, (const bf16*)y, perm, w, (bf16*)dy,
                         dw, n, K, D, round_prod);
    else
      hipLaunchKernelGGL((combine_bwd_kernel<bf16, 1>), grid, block, 0, st, (const bf16*)dout, (const bf16*)y, perm, w, (bf16*)dy,
                         dw, n, K, D, round_prod);
  } else {
    if (D % 4 == 0 && al)
      hipLaunchKernelGGL((combine_bwd_kernel<float, 4>), grid, block, 0, st, (const float*)dout, (const float*)y, perm, w,
                         (float*)dy, dw, n, K, D, round_prod);
    else
      hipLaunchKernelGGL((combine_bwd_kernel<float, 1>), grid, block, 0, st, (const float*)dout, (const float*)y, perm, w,
                         (float*)dy, dw, n, K, D, round_prod);
  }
  CSMOE_CHECK_LAUNCH("combine_bwd");
  return CSMOE_OK;
}

// bf16 rows, fp32 residual / output (pretrain stack under autocast); D % 8 == 0 and 16-byte alignment checked by the caller
int k_combine_mixed(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const float* add, float* out, int T, int K,
                    int D, int mode, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  static const bool generic_only = getenv("CSMOE_COMBINE_GENERIC") != nullptr;
  if (K == 2 && D % 512 == 0 && !generic_only)
    hipLaunchKernelGGL((combine_k2_kernel<float>), dim3(stride_grid(T)), dim3(256), 0, st, (const bf16*)y, slot_of, idx, w,
                       (const bf16*)nullptr, add, out, T, D, mode);
  else
    hipLaunchKernelGGL((combine_kernel<bf16, 8, float>), dim3(stride_grid(T)), dim3(256), 0, st, (const bf16*)y, slot_of, idx, w,
                       (const bf16*)nullptr, add, out, T, K, D, mode, (const bf16*)nullptr);
  CSMOE_CHECK_LAUNCH("combine_mixed");
  return CSMOE_OK;
}

int k_combine_bwd_mixed(const float* dout, const void* y, const int32_t* slot_of, const float* w, void* dy, float* dw, int T, int K,
                        int D, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  static const bool generic_only = getenv("CSMOE_COMBINE_GENERIC") != nullptr;
  if (K == 2 && D % 512 == 0 && !generic_only)
    hipLaunchKernelGGL((combine_bwd_k2_kernel<float>), dim3(stride_grid(T)), dim3(256), 0, st, dout, (const bf16*)y, slot_of, w,
                       (bf16*)dy, dw, T, D);
  else
    hipLaunchKernelGGL((combine_bwd_kernel<bf16, 8, float>), dim3(stride_grid(T)), dim3(256), 0, st, dout, (const bf16*)y, slot_of, w,
                       (bf16*)dy, dw, T, K, D, 0);
  CSMOE_CHECK_LAUNCH("combine_bwd_mixed");
  return CSMOE_OK;
}

// out[i] = (float(a[i]) + float(b[i])) + float(c[i])  (b, c may be null): bf16 gradient streams of an fp32 tensor, each widened by
// its cast's backward and added by the autograd engine in fp32 (csmoe_widen_sum) -- one pass instead of a cast per stream and an
// fp32 add per pair.  Four 16-byte chunks per stream in flight per lane; the tail (n % 8 elements) is scalar.
__global__ void __launch_bounds__(256) widen_sum_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, const bf16* __restrict__ c,
                                                        float* __restrict__ out, int64_t n) {
#pragma clang fp contract(off)
  const int64_t n8 = n >> 3, step = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n8; i0 += 4 * step) {
    bf16x8 va[4], vb[4], vc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * step;
      if (i < n8) {
        va[u] = ((const bf16x8*)a)[i];
        if (b) vb[u] = ((const bf16x8*)b)[i];
        if (c) vc[u] = ((const bf16x8*)c)[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * step;
      if (i >= n8) continue;
      float r[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        r[v] = (float)va[u][v];
        if (b) r[v] = r[v] + (float)vb[u][v];
        if (c) r[v] = r[v] + (float)vc[u][v];
      }
      float* o = out + i * 8;
      *(f32x4*)o = f32x4{r[0], r[1], r[2], r[3]};
      *(f32x4*)(o + 4) = f32x4{r[4], r[5], r[6], r[7]};
    }
  }
  if (blockIdx.x == 0 && (n8 << 3) + threadIdx.x < n) {
    const int64_t i = (n8 << 3) + threadIdx.x;
    float r = (float)a[i];
    if (b) r = r + (float)b[i];
    if (c) r = r + (float)c[i];
    out[i] = r;
  }
}

int k_widen_sum(const void* a, const void* b, const void* c, float* out, int64_t n, hipStream_t st) {
  if (n == 0) return CSMOE_OK;
  const int64_t n8 = n >> 3;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n8 + 1023) / 1024, 8192));
  hipLaunchKernelGGL(widen_sum_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)c, out, n);
  CSMOE_CHECK_LAUNCH("widen_sum");
  return CSMOE_OK;
}

int k_colsum(const void* G, int64_t ldg, const int32_t* offsets, int E, int single_M, int N, void* const* out_ptrs,
             void* single_out, int dtype, int out_dtype, hipStream_t st) {
  if (N == 0 || E == 0) return CSMOE_OK;
  dim3 grid((N + 255) / 256, E), block(256);
  if (dtype == CSMOE_BF16 && N % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)G & 15) == 0) {
    dim3 g8((N + 511) / 512, E);
    if (out_dtype == CSMOE_F32)
      hipLaunchKernelGGL((colsum_bf16x8_kernel<float>), g8, block, 0, st, (const bf16*)G, ldg, offsets, single_M, N, out_ptrs, single_out);
    else
      hipLaunchKernelGGL((colsum_bf16x8_kernel<bf16>), g8, block, 0, st, (const bf16*)G, ldg, offsets, single_M, N, out_ptrs, single_out);
    CSMOE_CHECK_LAUNCH("grouped_colsum");
    return CSMOE_OK;
  }
  if (dtype == CSMOE_F32 && N % 8 == 0 && ldg % 8 == 0 && (((uintptr_t)G | (uintptr_t)single_out) & 15) == 0 &&
      (int64_t)((N / 8 + 255) / 256) * E >= 256) {
    // output alignment: rows of the out tensors start at multiples of N elements from 16-byte aligned bases (torch allocations)
    dim3 gw((N / 8 + 255) / 256, E);
    if (out_dtype == CSMOE_F32)
      hipLaunchKernelGGL((colsum_f32_wide_kernel<float>), gw, block, 0, st, (const float*)G, ldg, offsets, single_M, N, out_ptrs, single_out);
    else
      hipLaunchKernelGGL((colsum_f32_wide_kernel<bf16>), gw, block, 0, st, (const float*)G, ldg, offsets, single_M, N, out_ptrs, single_out);
    CSMOE_CHECK_LAUNCH("grouped_colsum");
    return CSMOE_OK;
  }
  if (dtype == CSMOE_F32 && out_dtype == CSMOE_F32)
    hipLaunchKernelGGL((colsum_kernel<float, float>), grid, block, 0, st, (const float*)G, ldg, offsets, single_M, N, out_ptrs,
                       single_out);
  else if (dtype == CSMOE_F32)          // fp32 partial rows summed straight into a bf16 gradient (split-K weight gradients)
    hipLaunchKernelGGL((colsum_kernel<float, bf16>), grid, block, 0, st, (const float*)G, ldg, offsets, single_M, N, out_ptrs,
                       single_out);
  else if (out_dtype == CSMOE_F32)
    hipLaunchKernelGGL((colsum_kernel<bf16, float>), grid, block, 0, st, (const bf16*)G, ldg, offsets, single_M, N, out_ptrs,
                       single_out);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)G, ldg, offsets, single_M, N, out_ptrs,
                       single_out);
  CSMOE_CHECK_LAUNCH("grouped_colsum");
  return CSMOE_OK;
}

template <typename T, typename TA>
static void softplus_launch(const void* y, void* aff, int R, int D, bool vec, int precise, dim3 grid, dim3 block, hipStream_t st) {
  if (vec) {
    if (precise) hipLaunchKernelGGL((softplus_mean_kernel<T, TA, true, true>), grid, block, 0, st, (const T*)y, (TA*)aff, R, D);
    else         hipLaunchKernelGGL((softplus_mean_kernel<T, TA, true, false>), grid, block, 0, st, (const T*)y, (TA*)aff, R, D);
  } else {
    if (precise) hipLaunchKernelGGL((softplus_mean_kernel<T, TA, false, true>), grid, block, 0, st, (const T*)y, (TA*)aff, R, D);
    else         hipLaunchKernelGGL((softplus_mean_kernel<T, TA, false, false>), grid, block, 0, st, (const T*)y, (TA*)aff, R, D);
  }
}

__global__ void __launch_bounds__(256) affinity_finish_kernel(const float* partial, int M, int nt, int D, void* aff, int64_t stride,
                                                              int aff_dtype) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s = 0.f;
  for (int j = 0; j < nt; ++j) s += partial[(int64_t)m * nt + j];          // column tiles in ascending order: deterministic
  s /= (float)D;
  if (aff_dtype == CSMOE_BF16) ((bf16*)aff)[(int64_t)m * stride] = (bf16)s; else ((float*)aff)[(int64_t)m * stride] = s;
}

int k_affinity_finish(const float* partial, int M, int nt, int D, void* aff, int64_t stride, int aff_dtype, hipStream_t st) {
  if (M == 0) return CSMOE_OK;
  hipLaunchKernelGGL(affinity_finish_kernel, dim3((M + 255) / 256), dim3(256), 0, st, partial, M, nt, D, aff, stride, aff_dtype);
  CSMOE_CHECK_LAUNCH("affinity_finish");
  return CSMOE_OK;
}

int k_softplus_mean(const void* y, void* aff, int R, int D, int dtype, int aff_dtype, int precise, hipStream_t st) {
  if (R == 0) return CSMOE_OK;
  dim3 grid(stride_grid(R)), block(256);
  const bool al = ((uintptr_t)y & 15) == 0;
  if (dtype == CSMOE_BF16) {
    const bool vec = D % 8 == 0 && al;
    if (aff_dtype == CSMOE_BF16) softplus_launch<bf16, bf16>(y, aff, R, D, vec, precise, grid, block, st);
    else                         softplus_launch<bf16, float>(y, aff, R, D, vec, precise, grid, block, st);
  } else {
    softplus_launch<float, float>(y, aff, R, D, D % 4 == 0 && al, precise, grid, block, st);
  }
  CSMOE_CHECK_LAUNCH("softplus_mean");
  return CSMOE_OK;
}

template <typename T, typename TA>
static void softplus_bwd_launch(const void* y, const void* daff, const void* dy_add, void* dy, int R, int D, bool vec, int precise,
                                dim3 grid, dim3 block, hipStream_t st) {
#define SPB(V, P) hipLaunchKernelGGL((softplus_mean_bwd_kernel<T, TA, V, P>), grid, block, 0, st, (const T*)y, (const TA*)daff, \
                                     (const T*)dy_add, (T*)dy, R, D)
  if (vec) { if (precise) SPB(true, true); else SPB(true, false); }
  else     { if (precise) SPB(false, true); else SPB(false, false); }
#undef SPB
}

int k_softplus_mean_bwd(const void* y, const void* daff, const void* dy_add, void* dy, int R, int D, int dtype, int aff_dtype,
                        int precise, hipStream_t st) {
  if (R == 0) return CSMOE_OK;
  dim3 grid(stride_grid(R)), block(256);
  const bool al = (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dy_add) & 15) == 0;
  if (dtype == CSMOE_BF16) {
    const bool vec = D % 8 == 0 && al;
    if (aff_dtype == CSMOE_BF16) softplus_bwd_launch<bf16, bf16>(y, daff, dy_add, dy, R, D, vec, precise, grid, block, st);
    else                         softplus_bwd_launch<bf16, float>(y, daff, dy_add, dy, R, D, vec, precise, grid, block, st);
  } else {
    softplus_bwd_launch<float, float>(y, daff, dy_add, dy, R, D, D % 4 == 0 && al, precise, grid, block, st);
  }
  CSMOE_CHECK_LAUNCH("softplus_mean_bwd");
  return CSMOE_OK;
}

int k_pair_cosine(const void* y, float* tok_loss, int T, int K, int D, int dtype, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  dim3 grid(stride_grid(T)), block(256);
  const bool vec = ((uintptr_t)y & 15) == 0 && D % (dtype == CSMOE_BF16 ? 8 : 4) == 0;
  if (dtype == CSMOE_BF16) {
    if (vec) hipLaunchKernelGGL((pair_cosine_kernel<bf16, false, true>), grid, block, 0, st, (const bf16*)y, tok_loss, (const float*)nullptr, (bf16*)nullptr, T, K, D);
    else     hipLaunchKernelGGL((pair_cosine_kernel<bf16, false, false>), grid, block, 0, st, (const bf16*)y, tok_loss, (const float*)nullptr, (bf16*)nullptr, T, K, D);
  } else {
    if (vec) hipLaunchKernelGGL((pair_cosine_kernel<float, false, true>), grid, block, 0, st, (const float*)y, tok_loss, (const float*)nullptr, (float*)nullptr, T, K, D);
    else     hipLaunchKernelGGL((pair_cosine_kernel<float, false, false>), grid, block, 0, st, (const float*)y, tok_loss, (const float*)nullptr, (float*)nullptr, T, K, D);
  }
  CSMOE_CHECK_LAUNCH("pair_cosine");
  return CSMOE_OK;
}

int k_pair_cosine_bwd(const void* y, const float* gscale, void* dy, int T, int K, int D, int dtype, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  dim3 grid(stride_grid(T)), block(256);
  const bool vec = ((((uintptr_t)y | (uintptr_t)dy) & 15) == 0) && D % (dtype == CSMOE_BF16 ? 8 : 4) == 0;
  if (dtype == CSMOE_BF16) {
    if (vec) hipLaunchKernelGGL((pair_cosine_kernel<bf16, true, true>), grid, block, 0, st, (const bf16*)y, (float*)nullptr, gscale, (bf16*)dy, T, K, D);
    else     hipLaunchKernelGGL((pair_cosine_kernel<bf16, true, false>), grid, block, 0, st, (const bf16*)y, (float*)nullptr, gscale, (bf16*)dy, T, K, D);
  } else {
    if (vec) hipLaunchKernelGGL((pair_cosine_kernel<float, true, true>), grid, block, 0, st, (const float*)y, (float*)nullptr, gscale, (float*)dy, T, K, D);
    else     hipLaunchKernelGGL((pair_cosine_kernel<float, true, false>), grid, block, 0, st, (const float*)y, (float*)nullptr, gscale, (float*)dy, T, K, D);
  }
  CSMOE_CHECK_LAUNCH("pair_cosine_bwd");
  return CSMOE_OK;
}

int k_expert_order(const int32_t* offsets, int E, int32_t* order, hipStream_t st) {
  hipLaunchKernelGGL(expert_order_kernel, dim3(1), dim3(256), 0, st, offsets, E, order);
  CSMOE_CHECK_LAUNCH("expert_order");
  return CSMOE_OK;
}

bool k_gate_small_ok(int D, int E, int dtype, const void* a, const void* b) {
  const int n = dtype == CSMOE_BF16 ? 8 : 4;
  // E <= 4 only: the row passes are VALU work proportional to E (0.22 ms at E = 8, D = 4096, T = 32768 against 0.09 ms for the
  // MFMA path), they win where the GEMM tiles are nearly empty (0.024 vs 0.09 ms at the reference's E = 4, D = 1152)
  return E <= 4 && D % n == 0 && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
}

int k_gate_small_fwd(const void* x, const void* wg, void* logits, int T, int D, int E, int dtype, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  dim3 grid(stride_grid(T)), block(256);
  if (dtype == CSMOE_BF16) hipLaunchKernelGGL((gate_small_fwd_kernel<bf16>), grid, block, 0, st, (const bf16*)x, (const bf16*)wg, (bf16*)logits, T, D, E);
  else                     hipLaunchKernelGGL((gate_small_fwd_kernel<float>), grid, block, 0, st, (const float*)x, (const float*)wg, (float*)logits, T, D, E);
  CSMOE_CHECK_LAUNCH("gate_logits(small E)");
  return CSMOE_OK;
}

int k_gate_small_dx(const void* dl, const void* wg, void* dx, int T, int D, int E, int dtype, hipStream_t st) {
  if (T == 0) return CSMOE_OK;
  dim3 grid(stride_grid(T)), block(256);
  if (dtype == CSMOE_BF16) hipLaunchKernelGGL((gate_small_dx_kernel<bf16>), grid, block, 0, st, (const bf16*)dl, (const bf16*)wg, (bf16*)dx, T, D, E);
  else                     hipLaunchKernelGGL((gate_small_dx_kernel<float>), grid, block, 0, st, (const float*)dl, (const float*)wg, (float*)dx, T, D, E);
  CSMOE_CHECK_LAUNCH("gate_bwd_dx");
  return CSMOE_OK;
}

int k_gate_small_dw_ranges(int T, int D, int dtype) {
  const int n = dtype == CSMOE_BF16 ? 8 : 4;
  const int colgroups = (D + 64 * n - 1) / (64 * n);
  int r = (2048 + colgroups - 1) / colgroups;        // ~2048 waves in flight
  const int maxr = (T + 31) / 32;                    // at least 32 rows per range
  if (r > maxr) r = maxr;
  return r < 1 ? 1 : r;
}

int k_gate_small_dw(const void* dl, const void* x, float* partial, int T, int D, int E, int dtype, int nranges, hipStream_t st) {
  if (T == 0 || nranges <= 0) return CSMOE_OK;
  const int n = dtype == CSMOE_BF16 ? 8 : 4;
  dim3 grid((D + 64 * n - 1) / (64 * n), nranges), block(64);
  const int rows_per = (T + nranges - 1) / nranges;
  if (dtype == CSMOE_BF16) hipLaunchKernelGGL((gate_small_dw_kernel<bf16>), grid, block, 0, st, (const bf16*)dl, (const bf16*)x, partial, T, D, E, rows_per);
  else                     hipLaunchKernelGGL((gate_small_dw_kernel<float>), grid, block, 0, st, (const float*)dl, (const float*)x, partial, T, D, E, rows_per);
  CSMOE_CHECK_LAUNCH("gate_bwd_dw");
  return CSMOE_OK;
}

int64_t k_router_aux_workspace_floats(int B, int N, int E) {
  const int chunk = ra_chunk(B, N), nchunk = (N + chunk - 1) / chunk;
  return (int64_t)B * nchunk * (2 * E + 1);
}

int k_router_aux_fwd(const void* logits, const float* sm, const int32_t* idx, float* lse, float* partial, float* dens, float* out2,
                     int B, int N, int E, int K, int dtype, hipStream_t st) {
  const int chunk = ra_chunk(B, N), nchunk = (N + chunk - 1) / chunk;
  int lc = 1;
  while ((1 << lc) < E && lc < 8) ++lc;
  dim3 grid(B * nchunk), block(256);
  if (dtype == CSMOE_BF16) {
    hipLaunchKernelGGL((router_aux_partial_kernel<bf16>), grid, block, 0, st, (const bf16*)logits, sm, idx, lse, partial, N, E, K, nchunk, chunk, lc);
    hipLaunchKernelGGL((router_aux_final_kernel<bf16>), dim3(1), dim3(1024), 0, st, partial, dens, out2, B, N, E, nchunk, logits ? 1 : 0);
  } else {
    hipLaunchKernelGGL((router_aux_partial_kernel<float>), grid, block, 0, st, (const float*)logits, sm, idx, lse, partial, N, E, K, nchunk, chunk, lc);
    hipLaunchKernelGGL((router_aux_final_kernel<float>), dim3(1), dim3(1024), 0, st, partial, dens, out2, B, N, E, nchunk, logits ? 1 : 0);
  }
  CSMOE_CHECK_LAUNCH("router_aux_fwd");
  return CSMOE_OK;
}

int k_router_aux_bwd(const float* sm, const float* dens, const float* lse, const float* g_bal, const float* g_z, float* dsm,
                     void* dlogits, int B, int N, int E, int dtype, hipStream_t st) {
  const int64_t total = (int64_t)B * N * E;
  dim3 grid((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), block(256);
  if (dtype == CSMOE_BF16) hipLaunchKernelGGL((router_aux_bwd_kernel<bf16>), grid, block, 0, st, sm, dens, lse, g_bal, g_z, dsm, (bf16*)dlogits, B, N, E);
  else                     hipLaunchKernelGGL((router_aux_bwd_kernel<float>), grid, block, 0, st, sm, dens, lse, g_bal, g_z, dsm, (float*)dlogits, B, N, E);
  CSMOE_CHECK_LAUNCH("router_aux_bwd");
  return CSMOE_OK;
}

int k_chunk_offsets(const int32_t* offsets, int E, int P, int align, int32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(chunk_offsets_kernel, dim3(std::min((E * P + 256) / 256, 64)), dim3(256), 0, st, offsets, E, P, align, out);
  CSMOE_CHECK_LAUNCH("chunk_offsets");
  return CSMOE_OK;
}
