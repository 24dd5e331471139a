/* csmoe.h -- C ABI of the MI355X-native (gfx950) sparse-MoE hot path for CompeteSMoE.
 *
 * Drop-in boundary: the reference hides this path behind Python nn.Module classes
 * (`moe_model/model/moe/` and `moe_pretrain_model/layers/moe/`) whose only native code is the two
 * Triton kernels of `moe_pretrain_model/layers/cvmm.py`.  Every entry point below replaces a piece of
 * that path and cites it.  The host side (`competesmoe_amd/`) keeps the reference's module / registry
 * surface and calls these functions through ctypes with raw device pointers -- no torch types cross
 * this boundary.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked host; tensors are dense row-major;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); nothing synchronises;
 *   - `dtype` is CSMOE_F32 or CSMOE_BF16 (the activation / weight storage type; accumulation is fp32);
 *   - T tokens, D model dim, F expert hidden dim, E experts, K selected per token, n = T*K binned rows;
 *   - "binned row space": rows sorted by expert (stable counting sort), expert e owns rows
 *     [offsets[e], offsets[e+1]);
 *   - return value: 0 on success, CSMOE_ERR_* otherwise; csmoe_last_error() gives the message.
 */
#ifndef CSMOE_H
#define CSMOE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* csmoe_stream_t;

enum { CSMOE_F32 = 0, CSMOE_BF16 = 1 };
enum { CSMOE_ACT_NONE = 0, CSMOE_ACT_RELU = 1, CSMOE_ACT_GELU = 2, CSMOE_ACT_GELU_TANH = 3, CSMOE_ACT_SILU = 4,
       CSMOE_ACT_QUICK_GELU = 5 /* x * sigmoid(1.702 x): CLIP towers (clip_smoe.py CLIPMLP, hidden_act "quick_gelu") */ };
enum { CSMOE_OK = 0, CSMOE_ERR_INVALID = 1, CSMOE_ERR_LAUNCH = 2, CSMOE_ERR_UNSUPPORTED = 3 };

/* Router selection rule (what top-k runs on, how the K weights are normalised). */
enum {
  CSMOE_SEL_SOFTMAX = 0,  /* softmax(fp32) -> topk -> w/sum          moe_model/model/moe/moe.py:129-130, smoe.py:44      */
  CSMOE_SEL_RAW = 1,      /* topk on raw scores -> w/sum (competition) moe_model/model/moe/competesmoe.py:253-255          */
  CSMOE_SEL_TOPK_SOFTMAX = 2, /* topk(logits) -> softmax over the K  moe_pretrain_model/layers/moe/deepseekv2.py:140-142 */
  CSMOE_SEL_SIGMOID = 3,  /* topk(sigmoid in the scores' dtype) -> w/(fp32 sum+1e-20), fp32 quotient (under CUDA autocast `sum` is an
                             fp32-policy op)                          moe_pretrain_model/layers/moe/deepseekv3.py:147-151 */
  CSMOE_SEL_TOPK_SIGMOID = 4 /* topk(logits) -> sigmoid(v / sel_param) -> w/sum   (`norm_sigmoid` + `scale_weight`,
                                moe_pretrain_model/layers/moe/competesmoe.py:476-483)                                    */
};

/* Combine rounding rule. */
enum {
  CSMOE_COMBINE_SEQ = 0,  /* LLaVA compute_moe: experts visited in index order, accumulator rounded to the
                             activation dtype after every add (moe_model/model/moe/moe.py:204)                        */
  CSMOE_COMBINE_DOT = 1,  /* cvmm reduction_weight: one fp32 dot over K, rounded once
                             (moe_pretrain_model/layers/cvmm.py:481-483)                                              */
  CSMOE_COMBINE_SEQ_RW = 2 /* as SEQ, but the product w*y is rounded to the activation dtype before the add: the
                             competition branch hands compute_moe weights already in x.dtype
                             (moe_model/model/moe/competesmoe.py:253-255 -> moe.py:204)                               */
};

/* GEMM operand layouts for the grouped expert GEMMs. */
enum {
  CSMOE_B_NK = 0,  /* per-expert B stored [N, Kd] (nn.Linear weight: y = a @ B^T)  */
  CSMOE_B_KN = 1   /* per-expert B stored [Kd, N] (cvmm keys/values: y = a @ B)    */
};

/* Epilogues of the row-space grouped GEMM. */
enum {
  CSMOE_EPI_PLAIN = 0,     /* C = round(acc)                                                         */
  CSMOE_EPI_BIAS = 1,      /* C = round(acc + bias_e[n])                                            */
  CSMOE_EPI_BIAS_ACT = 2,  /* C = round(acc + bias_e[n]);  C2 = round(act(C))   (bias may be null; C may be null when only
                              the activated output is kept: for ReLU act'(pre) = (act(pre) > 0), so ACTGRAD takes C2 as aux) */
  CSMOE_EPI_ACTGRAD = 3,   /* C = round(round(acc) * act'(aux[m,n]))             (GELU/ReLU backward) */
  CSMOE_EPI_ROUND_BIAS32_ACT = 4 /* u = round(acc) + bias_e[n] with an FP32 bias table; C = round(u) (may be null); C2 = round(act(u)):
                              the pretrain stack's `scores = cvmm(...) + bias[sel]` under autocast -- the cvmm output is already
                              rounded to bf16, the fp32 master bias promotes the sum to fp32, the activation runs on that and the
                              next cvmm rounds once more (moe_pretrain_model/layers/moe/moe.py:400-405) */
  ,
  /* The competition pass without its [T,E,D] outputs (csmoe_dense_gemm only, bf16 fast path; `act` carries two flags: bit 0 =
   * precise expf / log1pf, bit 1 = every softplus value / the row scale's product is rounded to bf16 as x.dtype tensor ops do):
   *   SOFTPLUS_ROWSUM: y = round(acc + bias[n]) is never stored; C is an FP32 table [M, ceil(N/256)] (ldc = its row stride) that
   *                    receives sum_n softplus(y[m,n]) over each 256-column tile -- `torch.mean(F.softplus(expert(x)), dim=-1)`
   *                    (moe_model/model/moe/competesmoe.py:240-243) is the row sum of that table over D (csmoe_affinity_finish);
   *   SOFTPLUS_GRAD:   C[m,n] = round(g[m] * sigmoid(y[m,n])), g[m] = aux[m] / N (rounded to bf16 under bit 1) with aux an FP32
   *                    vector [M] = d affinity[m]: the gradient of that mean with respect to y, from a recomputed y (autograd of
   *                    the same lines). */
  CSMOE_EPI_SOFTPLUS_ROWSUM = 5,
  CSMOE_EPI_SOFTPLUS_GRAD = 6,
  /* C[m,n] = round(round(scale[m] * round(acc)) * act'(aux[m,n])) with scale an FP32 vector [M] passed in the C2 slot: the
   * backward of the pretrain stack's weighted cvmm, which multiplies by the reduction weight AFTER the product has been rounded --
   * `grad_x_full = cvmm(grad_output, ..)` (unscaled, bf16), then `reduction_weight @ grad_x_full`
   * (moe_pretrain_model/layers/cvmm.py:527-543) -- so A holds the UNSCALED upstream rows.  (csmoe_grouped_gemm)
   * Optional dot table in the bias slot (every bias_ptrs[e] the SAME FP32 table [M][csmoe_grouped_gemm_rowdot_cols(..)]): receives
   * partial sums over column groups of round(acc)[m,n] * aux[m,n]; their row sums (csmoe_affinity_finish with D = 1) are the gradient
   * of the reduction weights as the reference forms it, `grad_x_full @ x` (cvmm.py:544) -- meaningful when aux is the activated
   * input of the second product (ReLU experts). */
  CSMOE_EPI_ACTGRAD_ROWSCALE = 7
};

int csmoe_version(void);
const char* csmoe_last_error(void);
/* Number of compute units / name of the device `stream` belongs to (host ints). */
int csmoe_device_info(int* n_cu, int* lds_bytes, char* name, int name_len);

/* ---- router ------------------------------------------------------------------------------------------
 * gate projection  logits[T,E] = x[T,D] @ w_gate[E,D]^T, rounded to `dtype`
 * replaces `self.gate(x)` (moe_model/model/moe/smoe.py:42, competesmoe.py:314) and
 * `F.linear(x, self.w_gate)` (moe_pretrain_model/layers/moe/moe.py:121). */
int csmoe_gate_logits(const void* x, const void* w_gate, void* logits, int T, int D, int E, int dtype,
                      csmoe_stream_t stream);

/* scores[T,E] (dtype) -> softmax[T,E] fp32 (of the scores), idx[T,K] int32 (descending value, ties ->
 * lowest index), w[T,K] fp32.  `round_sum_bf16` != 0 rounds the K-sum to bf16 before the division
 * (`.to(x.dtype)` on the denominator, smoe.py:44).  softmax may be null for SEL_TOPK_SOFTMAX/SIGMOID.
 * `sel_param`: the scale of SEL_TOPK_SIGMOID (args.scale_weight), ignored by the other rules.
 * replaces topk_expert + renorm (moe_model/model/moe/moe.py:113-132; smoe.py:19-44;
 * competesmoe.py:246-255; pretrain smoe.py:123-143, deepseekv2.py:140-142, deepseekv3.py:147-151). */
int csmoe_router_select(const void* scores, int dtype, int T, int E, int K, int sel_mode, int round_sum_bf16,
                        float sel_param, float* softmax, int32_t* idx, float* w, csmoe_stream_t stream);

/* backward of csmoe_router_select: given dw[T,K] (fp32) and optional dsoftmax[T,E] (fp32, from the aux
 * losses), produce dscores[T,E] in `dtype`.  (autograd of the same reference lines.) */
int csmoe_router_select_bwd(const void* scores, int dtype, int T, int E, int K, int sel_mode, int round_sum_bf16,
                            float sel_param, const float* softmax, const int32_t* idx, const float* w, const float* dw,
                            const float* dsoftmax, void* dscores, csmoe_stream_t stream);

/* ---- binning -----------------------------------------------------------------------------------------
 * Stable counting sort of idx[n] (expert id per (token,k) slot, n = T*K) into the binned row space.
 *   counts[E], offsets[E+1]; perm[n]: flat (t*K+k) held by binned row m; slot_of[n]: binned row of flat j.
 * `workspace` must hold csmoe_bin_workspace_bytes(n, E) bytes.
 * replaces cvmm_prepare_sel2's sort (moe_pretrain_model/layers/cvmm.py:580-593) and the E `torch.where`
 * scans of compute_moe (moe_model/model/moe/moe.py:189-191). */
int64_t csmoe_bin_workspace_bytes(int n, int E);
int csmoe_bin_tokens(const int32_t* idx, int n, int E, int32_t* counts, int32_t* offsets, int32_t* perm,
                     int32_t* slot_of, void* workspace, csmoe_stream_t stream);

/* the same sort from a histogram built elsewhere (csmoe_gate_select): block_hist[nb][E] counts the ids of
 * [b*chunk, (b+1)*chunk), nb = ceil(n / chunk); block_base[nb][E] is scratch.  Same outputs as csmoe_bin_tokens. */
int csmoe_bin_tokens_hist(const int32_t* idx, int n, int E, int chunk, const int32_t* block_hist, int32_t* block_base,
                          int32_t* counts, int32_t* offsets, int32_t* perm, int32_t* slot_of, csmoe_stream_t stream);

/* ---- one-pass router -----------------------------------------------------------------------------------
 * csmoe_gate_logits + csmoe_router_select (+ the counting pass of csmoe_bin_tokens) in ONE launch that reads x once:
 *   logits[T,E] (dtype) = x[T,D] @ w_gate[E,D]^T;  softmax[T,E] fp32 (nullable), idx[T,K], w[T,K] as csmoe_router_select
 *   gives them on those logits (identical bits: same selection routine);  block_hist (nullable): [ceil(T / rows)][E] counts of
 *   the ids of each block of csmoe_gate_select_rows() token rows, i.e. chunk = rows*K for csmoe_bin_tokens_hist.
 * Shapes it takes: csmoe_gate_select_ok(...) = 1 (bf16, E <= 64, D % 8 == 0); others use the two separate entries.
 * replaces `self.gate(x)` + `topk_expert` (moe_model/model/moe/smoe.py:42-44, moe.py:113-132) and `F.linear(x, w_gate)` + top-k
 * (moe_pretrain_model/layers/moe/moe.py:121, smoe.py:30-40, deepseekv2.py:140-142). */
int csmoe_gate_select_ok(int T, int D, int E, int K, int dtype);
int csmoe_gate_select_rows(void);
int csmoe_gate_select(const void* x, const void* w_gate, int T, int D, int E, int K, int sel_mode, int round_sum_bf16,
                      float sel_param, int dtype, void* logits, float* softmax, int32_t* idx, float* w, int32_t* block_hist,
                      csmoe_stream_t stream);

/* aff[m] = (sum_j partial[m, j]) / D in `aff_dtype` (bf16 or fp32): closes CSMOE_EPI_SOFTPLUS_ROWSUM; `aff` is strided
 * (element stride `aff_stride`) so that expert e's column of an affinity matrix [T, E] is written in place. */
int csmoe_affinity_finish(const float* partial, int M, int ntiles, int D, void* aff, int64_t aff_stride, int aff_dtype,
                          csmoe_stream_t stream);

/* ---- dispatch / combine ------------------------------------------------------------------------------
 * dispatch: xs[m,:] = x[perm[m] / K, :]                      (gather of x rows, moe.py:201 `x[batch_idx, token_idx]`;
 *                                                             cvmm.py:114-119 remap_offs_am) */
int csmoe_dispatch_rows(const void* x, const int32_t* perm, int K, void* xs, int n, int D, int dtype,
                        csmoe_stream_t stream);
/* token-major dispatch (same result; reads every x row from HBM once): xs[slot_of[t*K+k], :] = x[t, :] */
int csmoe_dispatch_tokens(const void* x, const int32_t* slot_of, int K, void* xs, int T, int D, int dtype,
                          csmoe_stream_t stream);
/* dispatch backward: dx[t,:] = round(round(sum_k dxs[slot_of[t*K+k], :]) + add[t,:])  (add may be null): the K-sum in fp32, rounded
 * once -- CVMM.backward's reduction (cvmm.py:544-545) and, for K <= 2, the LLaVA stack's autograd.
 * `idx` [T,K] != null selects what autograd does for the LLaVA stack's per-expert modules (moe.py:196-204): every expert's `x[...]`
 * is its own use of x, the engine runs their backward nodes last-created first and adds each result into the leaf's buffer in
 * x.dtype -- dx[t] = (((pre[t] + dxs[slot of the HIGHEST expert]) + ...) + dxs[slot of the lowest]) + add[t], rounded after every
 * add.  `pre` [T,D] (may be null) = a gradient stream that reached x before the experts' (the always-on expert of smoe_share /
 * deepseekv3, shard_smoe.py:53, created after the routed experts); `add` = one that arrives after them (the gate's). */
int csmoe_dispatch_rows_bwd(const void* dxs, const int32_t* slot_of, int K, const void* add, void* dx, int T, int D,
                            int dtype, const int32_t* idx, const void* pre, csmoe_stream_t stream);
/* combine: out[t,:] = sum_k w[t,k] * y[slot_of[t*K+k], :]  (+ obias[:] if non-null) with the rounding rule `mode`
 * (moe.py:204; cvmm.py:481-483).  idx gives the expert of each slot (visit order for COMBINE_SEQ).
 * `residual` [T,D] (may be null) is added last, as one more x.dtype addition: out = round(out + residual) -- the
 * `hidden_states = residual + results` of the block around the layer (siglip_smoe.py:155; relative_moe_transformer.py:161). */
int csmoe_combine(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const void* obias,
                  const void* residual, void* out, int T, int K, int D, int dtype, int mode, csmoe_stream_t stream);
/* combine backward: dy[m,:] = round(w * dout[t,:]) in the binned row space; dw[t,k] = <dout[t,:], y[slot,:]>
 * (y may be null -> dw not written).  (autograd of moe.py:204; cvmm.py:497-499,543)
 * `round_products` != 0: dw[t,k] = round(sum_d round(dout[t,d] * y[slot,d])), every product and the sum rounded to `dtype` -- what
 * autograd returns for `weights * out_exp` when the weights are an x.dtype tensor (the competition step's affinity weights,
 * competesmoe.py:253-258 -> moe.py:204); 0: one fp32 dot product (fp32 weights: router steps, cvmm). */
int csmoe_combine_bwd(const void* dout, const void* y, const int32_t* perm, const int32_t* slot_of, const float* w,
                      void* dy, float* dw, int T, int K, int D, int dtype, int round_products, csmoe_stream_t stream);

/* ---- grouped expert GEMM -----------------------------------------------------------------------------
 * Row-space GEMM: for every expert e and binned row m in [offsets[e], offsets[e+1]):
 *     C[m, 0:N] = epilogue( A[m, 0:Kd] (x) B_e )
 * B_e = b_ptrs[e] is [N,Kd] (CSMOE_B_NK) or [Kd,N] (CSMOE_B_KN) with leading dimension ldb.
 * bias_ptrs[e] -> [N] or null; C2 / aux are [M,N] with ldc.  M = offsets[E] rows in total (upper bound `M`).
 * replaces cvmm_kernel (moe_pretrain_model/layers/cvmm.py:61-168, 354-398), the per-expert nn.Linear calls of
 * compute_moe (moe_model/model/moe/moe.py:196-204) and their grad-input (cvmm.py:519-536).
 * `kernel` (CSMOE_KERNEL_*): 0 = the library chooses (one-wave-per-SIMD 256x256 kernel for K-contiguous weights where the shape allows, else the 8-wave
 * 256x256 kernel, the 128x128 kernel for small / narrow problems, the generic kernel for fp32 and unaligned operands); 1 forces the
 * generic kernel; 2 / 4 ask for the 8-wave / one-wave-per-SIMD kernel where the shape allows (tests and A/B runs: every kernel is
 * reachable through the interface, none through hidden state).  A CSMOE_EPI_ACTGRAD_ROWSCALE launch that passes a dot table and
 * lands on a kernel that cannot fill it returns CSMOE_ERR_UNSUPPORTED instead of leaving the table unwritten. */
#define CSMOE_KERNEL_AUTO 0
#define CSMOE_KERNEL_GENERIC 1
#define CSMOE_KERNEL_V2 2
#define CSMOE_KERNEL_V4 4
int csmoe_grouped_gemm(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                       const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd,
                       void* C, void* C2, const void* aux, int64_t ldc, int epilogue, int act, int dtype,
                       int kernel, csmoe_stream_t stream);

/* partial sums per row written into the dot table of CSMOE_EPI_ACTGRAD_ROWSCALE for this launch shape; 0: no table (generic kernel) */
int csmoe_grouped_gemm_rowdot_cols(int M, int N, int Kd, int64_t lda, int64_t ldb, int64_t ldc, int dtype);

/* Dense (single weight matrix) form of the same kernel: C[M,N] = epilogue(A[M,Kd] (x) B), used for the gate projection and
 * the always-on shared expert (moe_model/model/moe/shard_smoe.py:53, deepseekv3.py:44; pretrain deepseekv2.py:154-165). */
int csmoe_dense_gemm(const void* A, int64_t lda, const void* B, int b_layout, int64_t ldb, const void* bias, int M, int N,
                     int Kd, void* C, void* C2, const void* aux, int64_t ldc, int epilogue, int act, int dtype,
                     int kernel, csmoe_stream_t stream);

/* Weight-gradient GEMM: for every expert e:  C_e[0:Na, 0:Nb] = sum_{m in expert e} A[m,0:Na]^T B[m,0:Nb]
 * c_ptrs[e] -> [Na, Nb] (leading dim ldc) written in `out_dtype` (CSMOE_F32 or CSMOE_BF16); empty experts get zeros.
 * Deterministic (no atomics).  `accumulate` != 0 adds into the existing C_e.
 * replaces cvmm_backward_kernel3 (cvmm.py:194-345, 421-457) and autograd's per-expert weight grads. */
int csmoe_grouped_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int M,
                        int Na, int Nb, void* const* c_ptrs, int64_t ldc, int dtype, int out_dtype, int accumulate,
                        int force_generic, const int32_t* xcd_order, csmoe_stream_t stream);
/* Load balance of the persistent weight-gradient kernel under skewed routing: experts dealt to the 8 XCDs by row count (rank r,
 * most rows first, goes to XCD r % 8 in snake order).  order: 8 * ceil(E / 8) + 1 int32, entry [x * ceil(E/8) + k] = k-th
 * expert of XCD x or -1.  Computed once per routing decision, passed as `xcd_order` (may be null: contiguous expert ranges). */
int csmoe_expert_order(const int32_t* offsets, int E, int32_t* order, csmoe_stream_t stream);

/* Row ranges of the E experts cut into P chunks each, out [E * P + 1]: out[e * P + j] = min(offsets[e] + roundup(count_e * j / P,
 * align), offsets[e + 1]), out[E * P] = offsets[E].  The chunks are handed to csmoe_grouped_wgrad / csmoe_grouped_colsum as
 * E * P pseudo-experts (split-K with fp32 partials) when E experts alone give too few workgroups for the chip -- the reference's
 * LLaVA configurations use 4 experts (sft.sh:19-20).  No counterpart upstream. */
int csmoe_chunk_offsets(const int32_t* offsets, int E, int P, int align, int32_t* out, csmoe_stream_t stream);

/* Dense form: C[Na,Nb] = A[M,Na]^T B[M,Nb]. */
int csmoe_dense_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int Na, int Nb, void* C, int64_t ldc,
                      int dtype, int out_dtype, int accumulate, int force_generic, csmoe_stream_t stream);

/* Per-expert column sums (bias gradients): out_e[0:N] = sum_{m in e} G[m, 0:N]; out in `out_dtype`. */
int csmoe_grouped_colsum(const void* G, int64_t ldg, const int32_t* offsets, int E, int N, void* const* out_ptrs,
                         int dtype, int out_dtype, csmoe_stream_t stream);

int csmoe_dense_colsum(const void* G, int64_t ldg, int M, int N, void* out, int dtype, int out_dtype, csmoe_stream_t stream);

/* Router auxiliary losses of the LLaVA stack (moe.py:71-110): out2[0] = balance loss (fp32), out2[1] = z-loss (each step rounded
 * to x.dtype like the reference's tensor ops; 0 when logits == null).  softmax [B,N,E] fp32 = softmax of logits, idx [B,N,K]
 * (column 0 = top-1).  Keeps dens [B,E] (fraction of tokens whose top-1 is e) and lse [B*N] for the backward; workspace =
 * csmoe_router_aux_workspace_floats(B, N, E) floats.  Backward: dsoftmax (fp32, from d balance) and dlogits (x.dtype, from d z)
 * given the device scalars g_balance / g_z; either output may be null. */
int64_t csmoe_router_aux_workspace_floats(int B, int N, int E);
int csmoe_router_aux(const void* logits, const float* softmax, const int32_t* idx, float* lse, float* workspace, float* dens,
                     float* out2, int B, int N, int E, int K, int dtype, csmoe_stream_t stream);
int csmoe_router_aux_bwd(const float* softmax, const float* dens, const float* lse, const float* g_balance, const float* g_z,
                         float* dsoftmax, void* dlogits, int B, int N, int E, int dtype, csmoe_stream_t stream);

/* Gate backward for few experts (E <= 4, D a multiple of 8 (bf16) / 4 (fp32), 16-byte aligned rows; csmoe_gate_bwd_small_ok):
 * dx[T,D] = dlogits[T,E] @ w_gate[E,D] and the partial sums of dWg = dlogits^T @ x over csmoe_gate_bwd_dw_ranges(T, D, dtype)
 * row ranges (partial [nranges][E][D] fp32; their sum over the ranges -- csmoe_dense_colsum over [nranges, E*D] -- is dWg).
 * HBM-bound row passes replacing the MFMA GEMMs whose 128-wide tiles are empty at E = 4 (autograd of `self.gate(x)`, smoe.py:42;
 * csmoe_gate_logits switches to the matching forward kernel by itself). */
int csmoe_gate_bwd_small_ok(int D, int E, int dtype);
int csmoe_gate_bwd_dx(const void* dlogits, const void* w_gate, void* dx, int T, int D, int E, int dtype, csmoe_stream_t stream);
int csmoe_gate_bwd_dw_ranges(int T, int D, int dtype);
int csmoe_gate_bwd_dw(const void* dlogits, const void* x, float* partial, int T, int D, int E, int dtype, int nranges,
                      csmoe_stream_t stream);

/* diversity loss of the competition step (moe.py:133-171, competesmoe.py:180-218): tok_loss[t] = sum over ordered pairs
 * i != j of <y[t,i,:] / max(|y[t,i,:]|, 1e-12), y[t,j,:] / ...> (fp32), y = [T, K, D] selected expert outputs, K <= 8; the loss
 * is sum_t tok_loss[t] / (T*K*K).  Backward: dy = gscale[0] * d(sum_t tok_loss)/dy in x.dtype (gscale: device scalar). */
int csmoe_pair_cosine(const void* y, float* tok_loss, int T, int K, int D, int dtype, csmoe_stream_t stream);
int csmoe_pair_cosine_bwd(const void* y, const float* gscale, void* dy, int T, int K, int D, int dtype, csmoe_stream_t stream);

/* ---- the MoE half of a pre-LN block around the layer: out = x + MoE(LayerNorm(x))  (SURVEY.md section 8 f1;
 *      siglip_smoe.py:141-157 SiglipEncoderMoELayer.forward; relative_moe_transformer.py:153-161, preln) ----
 * LayerNorm over the last dimension with fp32 statistics (torch.nn.LayerNorm semantics): xn = (x - mean) * rstd * gamma + beta,
 * mean/rstd [T] fp32 are kept for the backward.  gamma / beta may be null.  With w_gate != null the router's gate projection is
 * computed from the rounded xn by the same call (second launch on the stream: the gate GEMM of csmoe_gate_logits reads the rows
 * the LayerNorm just wrote): logits[T,E] = xn @ w_gate^T (gate of moe.py:46 / F.linear(x, w_gate) moe.py:121).
 * D <= 4096, D % 8 == 0 (bf16) / D % 4 == 0 (fp32), 16-byte aligned operands. */
int csmoe_layernorm_gate(const void* x, const void* gamma, const void* beta, float eps, void* xn, float* mean, float* rstd,
                         int T, int D, int dtype, const void* w_gate, void* logits, int E, csmoe_stream_t stream);
/* LayerNorm backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * gamma, xhat = (x - mean) * rstd.  With
 * dxn2 != null the gradient of xn is round(dxn + dxn2) (expert path + gate path, summed as autograd would).  With
 * add != null the residual-path gradient is added in the same pass: dx = round(dx) + add.  `partial` receives
 * csmoe_layernorm_bwd_blocks(T) rows of [2][D] fp32 (row sums of dxn * xhat and of dxn): their column sums (csmoe_dense_colsum)
 * are dgamma and dbeta -- deterministic, no atomics. */
int csmoe_layernorm_bwd(const void* dxn, const void* dxn2, const void* x, const void* gamma, const float* mean, const float* rstd,
                        const void* add, void* dx, float* partial, int T, int D, int dtype, csmoe_stream_t stream);
int csmoe_layernorm_bwd_blocks(int T);

/* ---- the same block with an fp32 residual stream around bf16 activations: the pretrain stack under bf16 autocast
 *      (relative_moe_transformer.py:153-161 inside simple_task.py:295's autocast).  There LayerNorm is an fp32 op on fp32 x with
 *      fp32 gamma / beta, its output is cast to bf16 by the gate's F.linear (moe.py:121) and by cvmm (cvmm.py:29-32), the bf16 MoE
 *      output is added to the fp32 residual in fp32 (type promotion), and in the backward autograd casts the fp32 gradient to
 *      bf16 for the MoE, casts the two bf16 gradients of xn back to fp32 and sums them.  These entries read and write the fp32
 *      stream directly (no cast passes): xn / logits / y / dy / dxn are bf16, x / residual / out / dout / dx are fp32.
 *      D <= 4096, D % 8 == 0, 16-byte aligned operands. */
int csmoe_layernorm_gate_mixed(const float* x, const float* gamma, const float* beta, float eps, void* xn, float* mean, float* rstd,
                               int T, int D, const void* w_gate, void* logits, int E, csmoe_stream_t stream);
/* dx = LayerNorm backward of (float(dxn) + float(dxn2)) [+ add], all sums in fp32 */
int csmoe_layernorm_bwd_mixed(const void* dxn, const void* dxn2, const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* add, float* dx, float* partial, int T, int D, csmoe_stream_t stream);
/* out[t] = float(round_bf16(combine of the bf16 rows y)) + residual[t]   (csmoe_combine with an fp32 residual / output) */
int csmoe_combine_mixed(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const float* residual, float* out,
                        int T, int K, int D, int mode, csmoe_stream_t stream);
/* csmoe_combine_bwd with an fp32 upstream gradient, rounded to bf16 on load (the cast autograd would insert) */
int csmoe_combine_bwd_mixed(const float* dout, const void* y, const int32_t* slot_of, const float* w, void* dy, float* dw, int T, int K,
                            int D, csmoe_stream_t stream);
/* csmoe_dispatch_rows_bwd into the fp32 stream: dx[t,:] = float(round_bf16(sum_k dxs[slot_of[t*K+k], :])) + float(add[t,:]), the sum
 * of the two in fp32 -- what autograd leaves in an fp32 x under bf16 autocast when the experts (CVMM.backward's bf16 reduction,
 * cvmm.py:544-545) and the gate (F.linear's bf16 dx, moe.py:121) each read their own bf16 cast of x: every cast's backward widens
 * its bf16 gradient and the engine adds the fp32 streams.  dxs, add (may be null): bf16; D % 8 == 0, 16-byte aligned operands. */
int csmoe_dispatch_rows_bwd_mixed(const void* dxs, const int32_t* slot_of, int K, const void* add, float* dx, int T, int D,
                                  csmoe_stream_t stream);
/* out[i] = (float(a[i]) + float(b[i])) + float(c[i]), a / b / c bf16 (b, c may be null, in that order), n elements: the gradient an
 * fp32 x receives under bf16 autocast when several ops read it -- the gate's F.linear, the experts' cvmm, the shared expert
 * (moe.py:121, cvmm.py:29-32,445; deepseekv2.py:154-165) each cast x themselves, every cast's backward widens its bf16 stream and the
 * autograd engine adds the fp32 streams.  One pass over the streams; 16-byte aligned operands. */
int csmoe_widen_sum(const void* a, const void* b, const void* c, float* out, int64_t n, csmoe_stream_t stream);

/* ---- competition affinity ----------------------------------------------------------------------------
 * aff[r] = mean_d softplus(y[r, d])  (competesmoe.py:242; pretrain competesmoe.py:401) and its backward
 * dy[r,d] = round(daff[r] / D * sigmoid(y[r,d])) (+ dy_add if non-null).
 * `aff_dtype` == `dtype`: the LLaVA stack's x.dtype tensor ops (every softplus value rounded to dtype before the mean, the mean
 * rounded to dtype).  `aff_dtype` == CSMOE_F32 around bf16 rows: the pretrain stack under CUDA autocast, where F.softplus is an
 * fp32-policy op (simple_task.py:295 + torch's autocast table): softplus, mean and the affinities are fp32, daff is fp32 and the
 * gradient is rounded to bf16 once.  `precise` != 0 evaluates exp / log1p as torch does instead of with the hardware
 * approximations. */
int csmoe_softplus_mean(const void* y, void* aff, int R, int D, int dtype, int aff_dtype, int precise, csmoe_stream_t stream);
int csmoe_softplus_mean_bwd(const void* y, const void* daff, const void* dy_add, void* dy, int R, int D, int dtype, int aff_dtype,
                            int precise, csmoe_stream_t stream);

/* Row-space grouped GEMM with the weights read straight from FP32 MASTERS: A [M, Kd] bf16, B_e = b_ptrs[e] [Kd, N] FP32 (leading
 * dimension ldb), converted to bf16 inside the tile fill (global -> registers -> cvt -> LDS) -- what the reference's Triton kernel
 * does per tile (`a.to(tl.bfloat16)`, moe_pretrain_model/layers/cvmm.py:126-140, fp32 `keys` / `values` under bf16 autocast) instead of
 * a separate cast pass over the weights.  Bit-identical to "cast to bf16, then csmoe_grouped_gemm(CSMOE_B_KN)".  b_copy_ptrs[e]
 * (may be null) receives the bf16 copy [Kd, N] of every expert that has rows -- the operand of the step's two backward products;
 * experts without rows are not written.  Same epilogues, outputs and bias conventions as csmoe_grouped_gemm with dtype bf16.
 * N % 8 == 0, Kd % 8 == 0, 16-byte aligned rows (else CSMOE_ERR_UNSUPPORTED: cast and use csmoe_grouped_gemm). */
int csmoe_grouped_gemm_f32w(const void* A, int64_t lda, const void* const* b_ptrs, int64_t ldb, void* const* b_copy_ptrs,
                            const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                            const void* aux, int64_t ldc, int epilogue, int act, csmoe_stream_t stream);

/* ---- MXFP8 expert GEMMs (BASELINE.json config 5: shared-expert variant on the fp8 matrix pipe) -------------------------
 * The reference has no fp8 path: these entries have no upstream counterpart; the contract is the bf16 path's
 * result within the quantisation error (tests: <= 3e-2 relative to the bf16 oracle, <= 2e-3 to the oracle's own MXFP8 emulation).
 * Format: OCP e4m3 elements, one e8m0 scale (2^(s - 127)) per 32 consecutive elements ALONG THE REDUCTION dimension of the GEMM
 * that consumes the tensor; scale = 2^(floor(log2(amax)) - 8), elements round-to-nearest-even, saturated at +-448.
 *
 * csmoe_quantize_mxfp8: E matrices [R, C] (x_ptrs[e], or the single matrix x when x_ptrs == null and E == 1), leading dimension
 * ldx, dtype CSMOE_BF16 or CSMOE_F32 (fp32 master weights are quantised directly).  transpose == 0: q [E, R, C], s [E, R, C/32]
 * (blocks along C; C % 32 == 0).  transpose != 0: q [E, C, R], s [E, C, R/32] (blocks along R; R % 32 == 0) -- the operand of the
 * transposed product (dX = dY . W for a weight stored [N, K]). */
int csmoe_quantize_mxfp8(const void* x, const void* const* x_ptrs, int E, int64_t ldx, int R, int C, int dtype, int transpose,
                         void* q, void* s, csmoe_stream_t stream);
/* Both orientations in one pass over the source (weights are needed along one dim by the forward product and along the other by
 * the backward product): q [E, R, C] / s [E, R, C/32] and qt [E, C, R] / st [E, C, R/32], bit-identical to the two separate calls.
 * R % 32 == 0, C % 32 == 0, 16-byte aligned rows. */
int csmoe_quantize_mxfp8_both(const void* x, const void* const* x_ptrs, int E, int64_t ldx, int R, int C, int dtype, void* q, void* s,
                              void* qt, void* st, csmoe_stream_t stream);
/* Row-space grouped GEMM on v_mfma_scale_f32_16x16x128_f8f6f4: C[m, 0:N] = epilogue(sum_k A[m,k] B_e[n,k]) for the binned rows of
 * expert e; A [M, Kd] e4m3 (lda) with scales [M, Kd/32] (ldas); B_e = bq_ptrs[e] [N, Kd] e4m3 (ldb), scales bs_ptrs[e] [N, Kd/32]
 * (ldbs); bf16 outputs / bias / aux and the epilogues PLAIN / BIAS / BIAS_ACT / ACTGRAD of csmoe_grouped_gemm.
 * Kd % 128 == 0, N % 8 == 0, 16-byte aligned operands, lda % 16 == 0, ldas % 4 == 0 (else CSMOE_ERR_UNSUPPORTED). */
int csmoe_grouped_gemm_mxfp8(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* const* bq_ptrs,
                             const void* const* bs_ptrs, int64_t ldb, int64_t ldbs, const void* const* bias_ptrs,
                             const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2, const void* aux, int64_t ldc,
                             int epilogue, int act, csmoe_stream_t stream);
/* Dense form (one weight matrix: the always-on shared expert). */
int csmoe_dense_gemm_mxfp8(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* Bq, const void* Bs, int64_t ldb,
                           int64_t ldbs, const void* bias, int M, int N, int Kd, void* C, void* C2, const void* aux, int64_t ldc,
                           int epilogue, int act, csmoe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSMOE_H */
