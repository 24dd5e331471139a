// C ABI (include/csmoe.h): argument validation, fast/generic kernel selection, error strings.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

// ---- kernels implemented in the other translation units
int gg_generic_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                        const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                        const void* aux, int64_t ldc, int epilogue, int act, int dtype, const void* single_B,
                        const void* single_bias, hipStream_t st);
int gg_generic_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na, int Nb,
                     void* const* c_ptrs, int64_t ldc, int dtype, int out_dtype, int accumulate, int single_M, void* single_C,
                     hipStream_t st);
bool gg_fast_rowspace_ok(int64_t lda, int64_t ldb, int64_t ldc, int M, int N, int Kd, const void* A, const void* C);
int gg_fast_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                     const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                     const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                     hipStream_t st);
bool gg_fast_wgrad_ok(int64_t lda, int64_t ldb, int64_t ldc, int M, int Na, int Nb, const void* A, const void* B);
int gg_fast_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na, int Nb,
                  void* const* c_ptrs, int64_t ldc, int out_dtype, int accumulate, int single_M, void* single_C, hipStream_t st);
int gg8_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                 const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                 const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                 hipStream_t st);
bool gg4_rowspace_ok(int Kd, int b_layout);
int gg4_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                 const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                 const void* aux, int64_t ldc, int epilogue, int act, const void* single_B, const void* single_bias,
                 hipStream_t st);
int gg8_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int Na, int Nb,
              void* const* c_ptrs, int64_t ldc, int out_dtype, int accumulate, int single_M, void* single_C, hipStream_t st, const int32_t* xcd_order = nullptr);
int gg8c_rowspace(const void* A, int64_t lda, const void* const* b_ptrs, int64_t ldb, void* const* copy_ptrs,
                  const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2, const void* aux,
                  int64_t ldc, int epilogue, int act, const void* single_B, void* single_copy, const void* single_bias, hipStream_t st,
                  int row_part);
int gg8_rowspace_rest(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                      const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                      const void* aux, int64_t ldc, int epilogue, int act, hipStream_t st);
int k_quantize_mxfp8(const void* const* x_ptrs, const void* x_single, int E, int64_t ldx, int R, int C, int in_dtype, int transpose,
                     void* q, void* s, hipStream_t st);
int k_quantize_mxfp8_both(const void* const* x_ptrs, const void* x_single, int E, int64_t ldx, int R, int C, int in_dtype, void* q,
                          void* s, void* qt, void* st, hipStream_t stream);
int gg8f_rowspace(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* const* bq_ptrs, const void* const* bs_ptrs,
                  int64_t ldb, int64_t ldbs, const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd,
                  void* C, void* C2, const void* aux, int64_t ldc, int epilogue, int act, const void* single_B,
                  const void* single_BS, const void* single_bias, hipStream_t st);
int k_router_select(const void*, int, int, int, int, int, int, float, float*, int32_t*, float*, hipStream_t);
int k_router_select_bwd(const void*, int, int, int, int, int, int, float, const float*, const int32_t*, const float*, const float*,
                        const float*, void*, hipStream_t);
int64_t k_bin_workspace_bytes(int n, int E);
int k_bin_tokens(const int32_t*, int, int, int32_t*, int32_t*, int32_t*, int32_t*, void*, hipStream_t);
int k_affinity_finish(const float*, int, int, int, void*, int64_t, int, hipStream_t);
int k_bin_tokens_hist(const int32_t*, int, int, int, const int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, hipStream_t);
bool k_gate_select_ok(int T, int D, int E, int K, int dtype, const void* x, const void* wg);
int k_gate_select_rows();
int k_gate_select(const void*, const void*, int, int, int, int, int, int, float, void*, float*, int32_t*, float*, int32_t*, hipStream_t);
int k_dispatch_rows(const void*, const int32_t*, int, void*, int, int, int, hipStream_t);
int k_dispatch_tokens(const void*, const int32_t*, int, void*, int, int, int, hipStream_t);
int k_combine(const void*, const int32_t*, const int32_t*, const float*, const void*, const void*, void*, int, int, int, int, int,
              hipStream_t, const void* pre);
int k_combine_bwd(const void*, const void*, const int32_t*, const float*, void*, float*, int, int, int, int, int, hipStream_t);
int k_colsum(const void*, int64_t, const int32_t*, int, int, int, void* const*, void*, int, int, hipStream_t);
int k_softplus_mean(const void*, void*, int, int, int, int, int, hipStream_t);
int k_pair_cosine(const void* y, float* tok_loss, int T, int K, int D, int dtype, hipStream_t st);
int k_expert_order(const int32_t* offsets, int E, int32_t* order, hipStream_t st);
int k_chunk_offsets(const int32_t* offsets, int E, int P, int align, int32_t* out, hipStream_t st);
int k_combine_mixed(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const float* add, float* out, int T, int K,
                    int D, int mode, hipStream_t st);
int k_combine_bwd_mixed(const float* dout, const void* y, const int32_t* slot_of, const float* w, void* dy, float* dw, int T, int K,
                        int D, hipStream_t st);
int k_dispatch_rows_bwd_mixed(const void* dxs, const int32_t* slot_of, int K, const void* add, float* dx, int T, int D, hipStream_t st);
int k_widen_sum(const void* a, const void* b, const void* c, float* out, int64_t n, hipStream_t st);
int k_layernorm_fwd_mixed(const float* x, const float* gamma, const float* beta, float eps, void* xn, float* mean, float* rstd, int T,
                          int D, hipStream_t st);
int k_layernorm_bwd_mixed(const void* dxn, const void* dxn2, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* add, float* dx, float* partial, int T, int D, hipStream_t st);
int64_t k_router_aux_workspace_floats(int B, int N, int E);
int k_router_aux_fwd(const void* logits, const float* sm, const int32_t* idx, float* lse, float* partial, float* dens, float* out2,
                     int B, int N, int E, int K, int dtype, hipStream_t st);
int k_router_aux_bwd(const float* sm, const float* dens, const float* lse, const float* g_bal, const float* g_z, float* dsm,
                     void* dlogits, int B, int N, int E, int dtype, hipStream_t st);
bool k_gate_small_ok(int D, int E, int dtype, const void* a, const void* b);
int k_gate_small_fwd(const void* x, const void* wg, void* logits, int T, int D, int E, int dtype, hipStream_t st);
int k_gate_small_dx(const void* dl, const void* wg, void* dx, int T, int D, int E, int dtype, hipStream_t st);
int k_gate_small_dw_ranges(int T, int D, int dtype);
int k_gate_small_dw(const void* dl, const void* x, float* partial, int T, int D, int E, int dtype, int nranges, hipStream_t st);
int k_pair_cosine_bwd(const void* y, const float* gscale, void* dy, int T, int K, int D, int dtype, hipStream_t st);
int k_layernorm_max_d(int dtype);
int k_layernorm_fwd(const void* x, const void* gamma, const void* beta, float eps, void* xn, float* mean, float* rstd, int T,
                    int D, int dtype, hipStream_t st);
int k_layernorm_bwd_blocks(int T);
int k_layernorm_bwd(const void* dxn, const void* dxn2, const void* x, const void* gamma, const float* mean, const float* rstd,
                    const void* add, void* dx, float* partial, int T, int D, int dtype, hipStream_t st);
int k_softplus_mean_bwd(const void*, const void*, const void*, void*, int, int, int, int, int, hipStream_t);

static thread_local char g_err[512] = "";

void csmoe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Kernel choice for the bf16 grouped GEMMs: v2 (256x256 tile, pipelined LDS-DMA) for big problems, v1 (128x128) for
// small / narrow ones.  CSMOE_GEMM_KERNEL=v1|v2 forces one (A/B runs); read once.
static int gemm_pref() {
  static int pref = -1;
  if (pref < 0) {
    const char* e = getenv("CSMOE_GEMM_KERNEL");
    pref = (e && !strcmp(e, "v1")) ? 1 : (e && !strcmp(e, "v2")) ? 2 : (e && !strcmp(e, "v4")) ? 4 : 0;
  }
  return pref;
}
// Row-space 256x256 kernel for a launch the v2 rule accepts: the one-wave-per-SIMD kernel (gemm_bf16_v4.hip) for K-contiguous
// weights when K is a whole number of K-tile pairs, else the 8-wave kernel.  `kernel` = the caller's CSMOE_KERNEL_* request; CSMOE_GEMM_KERNEL=v2|v4 (A/B runs,
// read once) overrides the automatic choice.
static bool use_v2_rowspace(int M, int N, int Kd) {
  int p = gemm_pref();
  if (p == 1) return false;
  if (p == 2 || p == 4) return true;
  return N >= 256 && Kd >= 128 && M >= 2048;
}
static bool use_v4_rowspace(int Kd, int b_layout, int kernel) {
  if (!gg4_rowspace_ok(Kd, b_layout)) return false;
  if (kernel == CSMOE_KERNEL_V2) return false;
  if (kernel == CSMOE_KERNEL_V4) return true;
  return gemm_pref() != 2;
}
typedef int (*rowspace_fn)(const void*, int64_t, const void* const*, int, int64_t, const void* const*, const int32_t*, int, int, int, int,
                           void*, void*, const void*, int64_t, int, int, const void*, const void*, hipStream_t);
static rowspace_fn pick_rowspace(int M, int N, int Kd, int b_layout, int kernel) {
  const bool big = (kernel == CSMOE_KERNEL_V2 || kernel == CSMOE_KERNEL_V4) ? (N >= 256 && Kd >= 128) : use_v2_rowspace(M, N, Kd);
  if (!big) return gg_fast_rowspace;
  return use_v4_rowspace(Kd, b_layout, kernel) ? gg4_rowspace : gg8_rowspace;
}
static bool use_v2_wgrad(int M, int Na, int Nb) {
  int p = gemm_pref();
  if (p == 1) return false;
  if (p == 2) return true;
  return Na >= 256 && Nb >= 256 && M >= 512;
}

static inline bool dtype_ok(int d) { return d == CSMOE_F32 || d == CSMOE_BF16; }
static inline int esize(int d) { return d == CSMOE_F32 ? 4 : 2; }

extern "C" {

int csmoe_version(void) { return 100; }
const char* csmoe_last_error(void) { return g_err; }

int csmoe_device_info(int* n_cu, int* lds_bytes, char* name, int name_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { csmoe_set_error("hipGetDevice: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) { csmoe_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return CSMOE_ERR_LAUNCH; }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  return CSMOE_OK;
}

int csmoe_gate_logits(const void* x, const void* w_gate, void* logits, int T, int D, int E, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "gate_logits: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && E > 0, "gate_logits: bad shape T=%d D=%d E=%d", T, D, E);
  CSMOE_CHECK_ARG(w_gate && (T == 0 || (x && logits)), "gate_logits: null pointer");
  if (T == 0) return CSMOE_OK;      // empty batch: zero-row buffers have no address
  hipStream_t st = (hipStream_t)stream;
  if (k_gate_small_ok(D, E, dtype, x, w_gate)) return k_gate_small_fwd(x, w_gate, logits, T, D, E, dtype, st);
  // one dense "expert" over all T rows: logits = x @ w_gate^T
  if (dtype == CSMOE_BF16 && gg_fast_rowspace_ok(D, D, E, T, E, D, x, logits))
    return gg_fast_rowspace(x, D, nullptr, CSMOE_B_NK, D, nullptr, nullptr, 1, T, E, D, logits, nullptr, nullptr, E,
                            CSMOE_EPI_PLAIN, CSMOE_ACT_NONE, w_gate, nullptr, st);
  return gg_generic_rowspace(x, D, nullptr, CSMOE_B_NK, D, nullptr, nullptr, 1, T, E, D, logits, nullptr, nullptr, E,
                             CSMOE_EPI_PLAIN, CSMOE_ACT_NONE, dtype, w_gate, nullptr, st);
}

int csmoe_router_select(const void* scores, int dtype, int T, int E, int K, int sel_mode, int round_sum_bf16, float sel_param,
                        float* softmax, int32_t* idx, float* w, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "router_select: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(T >= 0 && E > 0 && E <= 1024, "router_select: E=%d out of range (1..1024)", E);
  CSMOE_CHECK_ARG(K > 0 && K <= E && K <= 64, "router_select: K=%d out of range (1..min(E,64))", K);
  CSMOE_CHECK_ARG(sel_mode >= 0 && sel_mode <= 4, "router_select: bad mode %d", sel_mode);
  CSMOE_CHECK_ARG(sel_mode != CSMOE_SEL_TOPK_SIGMOID || sel_param != 0.f, "router_select: SEL_TOPK_SIGMOID needs a non-zero scale");
  if (T == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(scores && idx && w, "router_select: null pointer");
  return k_router_select(scores, dtype, T, E, K, sel_mode, round_sum_bf16, sel_param, softmax, idx, w, (hipStream_t)stream);
}

int csmoe_router_select_bwd(const void* scores, int dtype, int T, int E, int K, int sel_mode, int round_sum_bf16,
                            float sel_param, const float* softmax, const int32_t* idx, const float* w, const float* dw, const float* dsoftmax,
                            void* dscores, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "router_select_bwd: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(T >= 0 && E > 0 && E <= 1024 && K > 0 && K <= E && K <= 64, "router_select_bwd: bad shape");
  CSMOE_CHECK_ARG(sel_mode >= 0 && sel_mode <= 4, "router_select_bwd: bad mode %d", sel_mode);
  CSMOE_CHECK_ARG(sel_mode != CSMOE_SEL_TOPK_SIGMOID || sel_param != 0.f, "router_select_bwd: SEL_TOPK_SIGMOID needs a non-zero scale");
  if (T == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(scores && idx && w && dscores, "router_select_bwd: null pointer");
  CSMOE_CHECK_ARG(softmax || (sel_mode != CSMOE_SEL_SOFTMAX && !dsoftmax), "router_select_bwd: softmax required");
  return k_router_select_bwd(scores, dtype, T, E, K, sel_mode, round_sum_bf16, sel_param, softmax, idx, w, dw, dsoftmax, dscores,
                             (hipStream_t)stream);
}

int64_t csmoe_bin_workspace_bytes(int n, int E) { return k_bin_workspace_bytes(n, E); }

int csmoe_bin_tokens(const int32_t* idx, int n, int E, int32_t* counts, int32_t* offsets, int32_t* perm, int32_t* slot_of,
                     void* workspace, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(n >= 0 && E > 0 && E <= 8192, "bin_tokens: bad n=%d E=%d", n, E);
  CSMOE_CHECK_ARG(counts && offsets && workspace && (n == 0 || (idx && perm && slot_of)), "bin_tokens: null pointer");
  return k_bin_tokens(idx, n, E, counts, offsets, perm, slot_of, workspace, (hipStream_t)stream);
}

int csmoe_bin_tokens_hist(const int32_t* idx, int n, int E, int chunk, const int32_t* block_hist, int32_t* block_base, int32_t* counts,
                          int32_t* offsets, int32_t* perm, int32_t* slot_of, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(n >= 0 && E > 0 && E <= 8192 && chunk > 0, "bin_tokens_hist: bad n=%d E=%d chunk=%d", n, E, chunk);
  CSMOE_CHECK_ARG(counts && offsets && block_hist && block_base && (n == 0 || (idx && perm && slot_of)), "bin_tokens_hist: null pointer");
  return k_bin_tokens_hist(idx, n, E, chunk, block_hist, block_base, counts, offsets, perm, slot_of, (hipStream_t)stream);
}

int csmoe_affinity_finish(const float* partial, int M, int ntiles, int D, void* aff, int64_t aff_stride, int aff_dtype,
                          csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(M >= 0 && ntiles > 0 && D > 0 && aff_stride > 0 && dtype_ok(aff_dtype), "affinity_finish: bad arguments");
  CSMOE_CHECK_ARG(M == 0 || (partial && aff), "affinity_finish: null pointer");
  return k_affinity_finish(partial, M, ntiles, D, aff, aff_stride, aff_dtype, (hipStream_t)stream);
}

int csmoe_gate_select_ok(int T, int D, int E, int K, int dtype) {
  return T >= 0 && D > 0 && K > 0 && k_gate_select_ok(T, D, E, K, dtype, nullptr, nullptr) ? 1 : 0;
}

int csmoe_gate_select_rows(void) { return k_gate_select_rows(); }

int csmoe_gate_select(const void* x, const void* w_gate, int T, int D, int E, int K, int sel_mode, int round_sum_bf16, float sel_param,
                      int dtype, void* logits, float* softmax, int32_t* idx, float* w, int32_t* block_hist, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && E > 0 && K > 0 && K <= E && K <= 64, "gate_select: bad shape T=%d D=%d E=%d K=%d", T, D, E, K);
  CSMOE_CHECK_ARG(sel_mode >= 0 && sel_mode <= 4, "gate_select: bad mode %d", sel_mode);
  CSMOE_CHECK_ARG(sel_mode != CSMOE_SEL_TOPK_SIGMOID || sel_param != 0.f, "gate_select: SEL_TOPK_SIGMOID needs a non-zero scale");
  CSMOE_CHECK_ARG(k_gate_select_ok(T, D, E, K, dtype, x, w_gate),
                  "gate_select: needs bf16, E <= 64, D a multiple of 8, 16-byte aligned operands (csmoe_gate_select_ok)");
  if (T == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(x && w_gate && logits && idx && w, "gate_select: null pointer");
  return k_gate_select(x, w_gate, T, D, E, K, sel_mode, round_sum_bf16, sel_param, logits, softmax, idx, w, block_hist,
                       (hipStream_t)stream);
}

int csmoe_dispatch_rows(const void* x, const int32_t* perm, int K, void* xs, int n, int D, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && K > 0 && n >= 0 && D > 0, "dispatch_rows: bad arguments");
  CSMOE_CHECK_ARG(n == 0 || (x && perm && xs), "dispatch_rows: null pointer");
  int row_bytes = D * esize(dtype);
  // 16-B vector path needs 16-B aligned rows; otherwise the kernel copies 2 bytes at a time
  int vec_ok = ((((uintptr_t)x | (uintptr_t)xs) & 15) == 0 && (row_bytes & 15) == 0) ? 1 : 0;
  return k_dispatch_rows(x, perm, K, xs, n, row_bytes, vec_ok, (hipStream_t)stream);
}

int csmoe_dispatch_tokens(const void* x, const int32_t* slot_of, int K, void* xs, int T, int D, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && K > 0 && T >= 0 && D > 0, "dispatch_tokens: bad arguments");
  CSMOE_CHECK_ARG(T == 0 || (x && slot_of && xs), "dispatch_tokens: null pointer");
  int row_bytes = D * esize(dtype);
  int vec_ok = ((((uintptr_t)x | (uintptr_t)xs) & 15) == 0 && (row_bytes & 15) == 0) ? 1 : 0;
  return k_dispatch_tokens(x, slot_of, K, xs, T, row_bytes, vec_ok, (hipStream_t)stream);
}

int csmoe_dispatch_rows_bwd(const void* dxs, const int32_t* slot_of, int K, const void* add, void* dx, int T, int D, int dtype,
                            const int32_t* idx, const void* pre, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && K > 0 && K <= 64 && T >= 0 && D > 0, "dispatch_rows_bwd: bad arguments");
  CSMOE_CHECK_ARG(T == 0 || (dxs && slot_of && dx), "dispatch_rows_bwd: null pointer");
  CSMOE_CHECK_ARG(!pre || idx, "dispatch_rows_bwd: `pre` is the first addend of the sequential form, which needs idx");
  // idx given: sequential accumulation in x.dtype, slots in DESCENDING expert order, starting from `pre` (internal mode 3)
  return k_combine(dxs, slot_of, idx, nullptr, nullptr, add, dx, T, K, D, dtype, idx ? 3 : CSMOE_COMBINE_DOT, (hipStream_t)stream, pre);
}

int csmoe_combine(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const void* obias,
                  const void* residual, void* out, int T, int K, int D, int dtype, int mode, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && K > 0 && K <= 64 && T >= 0 && D > 0, "combine: bad arguments (K<=64)");
  CSMOE_CHECK_ARG(mode >= 0 && mode <= 2, "combine: bad mode %d", mode);
  CSMOE_CHECK_ARG(T == 0 || (y && slot_of && w && out), "combine: null pointer");
  CSMOE_CHECK_ARG(T == 0 || mode == CSMOE_COMBINE_DOT || idx, "combine: idx required for the sequential rounding rule");
  return k_combine(y, slot_of, idx, w, obias, residual, out, T, K, D, dtype, mode, (hipStream_t)stream, nullptr);
}

int csmoe_combine_bwd(const void* dout, const void* y, const int32_t* perm, const int32_t* slot_of, const float* w, void* dy,
                      float* dw, int T, int K, int D, int dtype, int round_products, csmoe_stream_t stream) {
  (void)perm;
  CSMOE_CHECK_ARG(dtype_ok(dtype) && K > 0 && T >= 0 && D > 0, "combine_bwd: bad arguments");
  CSMOE_CHECK_ARG(T == 0 || (dout && slot_of && dy), "combine_bwd: null pointer");
  return k_combine_bwd(dout, y, slot_of, w, dy, dw, T, K, D, dtype, round_products, (hipStream_t)stream);
}

int csmoe_grouped_gemm(const void* A, int64_t lda, const void* const* b_ptrs, int b_layout, int64_t ldb,
                       const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                       const void* aux, int64_t ldc, int epilogue, int act, int dtype, int kernel,
                       csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "grouped_gemm: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(E > 0 && M >= 0 && N > 0 && Kd > 0, "grouped_gemm: bad shape E=%d M=%d N=%d Kd=%d", E, M, N, Kd);
  CSMOE_CHECK_ARG(b_layout == CSMOE_B_NK || b_layout == CSMOE_B_KN, "grouped_gemm: bad B layout %d", b_layout);
  CSMOE_CHECK_ARG(((epilogue >= 0 && epilogue <= 4) || epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE) && act >= 0 && act <= 5,
                  "grouped_gemm: bad epilogue/act");
  CSMOE_CHECK_ARG(b_ptrs && offsets && (M == 0 || (A && (C || ((epilogue == CSMOE_EPI_BIAS_ACT || epilogue == CSMOE_EPI_ROUND_BIAS32_ACT) && C2)))), "grouped_gemm: null pointer");
  CSMOE_CHECK_ARG(M == 0 || (epilogue != CSMOE_EPI_ACTGRAD && epilogue != CSMOE_EPI_ACTGRAD_ROWSCALE) || aux,
                  "grouped_gemm: ACTGRAD epilogue needs aux");
  CSMOE_CHECK_ARG(M == 0 || epilogue != CSMOE_EPI_ACTGRAD_ROWSCALE || (C && C2),
                  "grouped_gemm: ACTGRAD_ROWSCALE takes its FP32 row scales [M] in the C2 slot");
  CSMOE_CHECK_ARG(lda >= Kd && ldc >= N && ldb >= (b_layout == CSMOE_B_NK ? Kd : N), "grouped_gemm: leading dimension too small");
  if (M == 0) return CSMOE_OK;
  hipStream_t st = (hipStream_t)stream;
  CSMOE_CHECK_ARG(kernel == CSMOE_KERNEL_AUTO || kernel == CSMOE_KERNEL_GENERIC || kernel == CSMOE_KERNEL_V2 || kernel == CSMOE_KERNEL_V4,
                  "grouped_gemm: bad kernel selector %d", kernel);
  // ACTGRAD_ROWSCALE with a dot table (passed in the bias slot): only the tiled bf16 kernels fill it
  const bool wants_dot = epilogue == CSMOE_EPI_ACTGRAD_ROWSCALE && bias_ptrs != nullptr;
  if (kernel != CSMOE_KERNEL_GENERIC && dtype == CSMOE_BF16 && gg_fast_rowspace_ok(lda, ldb, ldc, M, N, Kd, A, C ? C : C2))
    return pick_rowspace(M, N, Kd, b_layout, kernel)(A, lda, b_ptrs, b_layout, ldb, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act,
                                           nullptr, nullptr, st);
  if (wants_dot) {
    csmoe_set_error("grouped_gemm: ACTGRAD_ROWSCALE was given a dot table, but this launch (dtype %d, alignment, kernel %d) runs on the "
                    "generic kernel, which writes none (csmoe_grouped_gemm_rowdot_cols reports 0 for it)", dtype, kernel);
    return CSMOE_ERR_UNSUPPORTED;
  }
  return gg_generic_rowspace(A, lda, b_ptrs, b_layout, ldb, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act,
                             dtype, nullptr, nullptr, st);
}

int csmoe_grouped_gemm_rowdot_cols(int M, int N, int Kd, int64_t lda, int64_t ldb, int64_t ldc, int dtype) {
  // partial sums per row that CSMOE_EPI_ACTGRAD_ROWSCALE writes into its dot table for this launch shape (16-byte aligned
  // operands assumed): one per 128-column half from the 256-tile kernel, one per 8 columns from the 128-tile kernel, 0 = the
  // generic kernel would run (no table)
  if (dtype != CSMOE_BF16 || M <= 0 || N <= 0 || Kd <= 0 || !gg_fast_rowspace_ok(lda, ldb, ldc, M, N, Kd, nullptr, nullptr)) return 0;
  return use_v2_rowspace(M, N, Kd) ? (N + 127) / 128 : N / 8;
}

int csmoe_dense_gemm(const void* A, int64_t lda, const void* B, int b_layout, int64_t ldb, const void* bias, int M, int N,
                     int Kd, void* C, void* C2, const void* aux, int64_t ldc, int epilogue, int act, int dtype,
                     int kernel, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && M >= 0 && N > 0 && Kd > 0, "dense_gemm: bad arguments");
  CSMOE_CHECK_ARG(M == 0 || (A && B && (C || ((epilogue == CSMOE_EPI_BIAS_ACT || epilogue == CSMOE_EPI_ROUND_BIAS32_ACT) && C2))), "dense_gemm: null pointer");
  CSMOE_CHECK_ARG(M == 0 || epilogue != CSMOE_EPI_ACTGRAD || aux, "dense_gemm: ACTGRAD epilogue needs aux");
  if (M == 0) return CSMOE_OK;
  hipStream_t st = (hipStream_t)stream;
  if (epilogue == CSMOE_EPI_SOFTPLUS_ROWSUM || epilogue == CSMOE_EPI_SOFTPLUS_GRAD) {
    // the affinity epilogues of the competition pass: the 256x256-tile bf16 kernel only
    const bool rowsum = epilogue == CSMOE_EPI_SOFTPLUS_ROWSUM;
    CSMOE_CHECK_ARG(epilogue == CSMOE_EPI_SOFTPLUS_ROWSUM || aux, "dense_gemm: SOFTPLUS_GRAD needs aux (the FP32 row scales)");
    CSMOE_CHECK_ARG(!rowsum || ldc >= (N + 255) / 256, "dense_gemm: SOFTPLUS_ROWSUM: ldc is the row stride of the FP32 table [M, ceil(N/256)]");
    if (dtype != CSMOE_BF16 || !gg_fast_rowspace_ok(lda, ldb, rowsum ? (int64_t)N : ldc, M, N, Kd, A, rowsum ? A : C) ||
        ((uintptr_t)C & 15)) {
      csmoe_set_error("dense_gemm: the affinity epilogues need bf16 operands on the fast path (dimensions multiples of 8, "
                      "16-byte aligned, under 2 GiB)");
      return CSMOE_ERR_UNSUPPORTED;
    }
    return (use_v4_rowspace(Kd, b_layout, kernel) ? gg4_rowspace : gg8_rowspace)(A, lda, nullptr, b_layout, ldb, nullptr, nullptr, 1, M, N, Kd, C, C2,
                                                                       aux, ldc, epilogue, act, B, bias, st);
  }
  if (kernel != CSMOE_KERNEL_GENERIC && dtype == CSMOE_BF16 && gg_fast_rowspace_ok(lda, ldb, ldc, M, N, Kd, A, C ? C : C2))
    return pick_rowspace(M, N, Kd, b_layout, kernel)(A, lda, nullptr, b_layout, ldb, nullptr, nullptr, 1, M, N, Kd, C, C2, aux, ldc, epilogue, act, B,
                                           bias, st);
  return gg_generic_rowspace(A, lda, nullptr, b_layout, ldb, nullptr, nullptr, 1, M, N, Kd, C, C2, aux, ldc, epilogue, act,
                             dtype, B, bias, st);
}

int csmoe_grouped_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, const int32_t* offsets, int E, int M, int Na,
                        int Nb, void* const* c_ptrs, int64_t ldc, int dtype, int out_dtype, int accumulate, int force_generic,
                        const int32_t* xcd_order, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && dtype_ok(out_dtype), "grouped_wgrad: bad dtype");
  CSMOE_CHECK_ARG(E > 0 && M >= 0 && Na > 0 && Nb > 0, "grouped_wgrad: bad shape");
  CSMOE_CHECK_ARG(c_ptrs && offsets && (M == 0 || (A && B)), "grouped_wgrad: null pointer");
  CSMOE_CHECK_ARG(lda >= Na && ldb >= Nb && ldc >= Nb, "grouped_wgrad: leading dimension too small");
  CSMOE_CHECK_ARG(!(dtype == CSMOE_F32 && out_dtype != CSMOE_F32), "grouped_wgrad: fp32 inputs need fp32 output");
  hipStream_t st = (hipStream_t)stream;
  if (!force_generic && dtype == CSMOE_BF16 && gg_fast_wgrad_ok(lda, ldb, ldc, M, Na, Nb, A, B)) {
    if (use_v2_wgrad(M, Na, Nb))
      return gg8_wgrad(A, lda, B, ldb, offsets, E, Na, Nb, c_ptrs, ldc, out_dtype, accumulate, 0, nullptr, st, xcd_order);
    return gg_fast_wgrad(A, lda, B, ldb, offsets, E, Na, Nb, c_ptrs, ldc, out_dtype, accumulate, 0, nullptr, st);
  }
  return gg_generic_wgrad(A, lda, B, ldb, offsets, E, Na, Nb, c_ptrs, ldc, dtype, out_dtype, accumulate, 0, nullptr, st);
}

int64_t csmoe_router_aux_workspace_floats(int B, int N, int E) { return (B > 0 && N > 0 && E > 0) ? k_router_aux_workspace_floats(B, N, E) : 0; }

int csmoe_router_aux(const void* logits, const float* softmax, const int32_t* idx, float* lse, float* workspace, float* dens,
                     float* out2, int B, int N, int E, int K, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && B > 0 && N > 0 && E > 0 && E <= 1024 && K > 0, "router_aux: bad arguments (E <= 1024)");
  CSMOE_CHECK_ARG(softmax && idx && workspace && dens && out2 && (logits == nullptr || lse), "router_aux: null pointer");
  return k_router_aux_fwd(logits, softmax, idx, lse, workspace, dens, out2, B, N, E, K, dtype, (hipStream_t)stream);
}

int csmoe_router_aux_bwd(const float* softmax, const float* dens, const float* lse, const float* g_balance, const float* g_z,
                         float* dsoftmax, void* dlogits, int B, int N, int E, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && B > 0 && N > 0 && E > 0, "router_aux_bwd: bad arguments");
  CSMOE_CHECK_ARG(softmax && dens && (dsoftmax || dlogits) && (!dlogits || (lse && g_z)) && (!dsoftmax || g_balance), "router_aux_bwd: null pointer");
  return k_router_aux_bwd(softmax, dens, lse, g_balance, g_z, dsoftmax, dlogits, B, N, E, dtype, (hipStream_t)stream);
}

int csmoe_gate_bwd_small_ok(int D, int E, int dtype) { return dtype_ok(dtype) && D > 0 && E > 0 && k_gate_small_ok(D, E, dtype, nullptr, nullptr) ? 1 : 0; }

int csmoe_gate_bwd_dx(const void* dlogits, const void* w_gate, void* dx, int T, int D, int E, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && T >= 0 && D > 0 && E > 0, "gate_bwd_dx: bad arguments");
  CSMOE_CHECK_ARG(k_gate_small_ok(D, E, dtype, w_gate, dx), "gate_bwd_dx: needs E <= 4, D a multiple of the 16-byte chunk, aligned operands");
  CSMOE_CHECK_ARG(T == 0 || (dlogits && w_gate && dx), "gate_bwd_dx: null pointer");
  return k_gate_small_dx(dlogits, w_gate, dx, T, D, E, dtype, (hipStream_t)stream);
}

int csmoe_gate_bwd_dw_ranges(int T, int D, int dtype) { return dtype_ok(dtype) && T > 0 && D > 0 ? k_gate_small_dw_ranges(T, D, dtype) : 0; }

int csmoe_gate_bwd_dw(const void* dlogits, const void* x, float* partial, int T, int D, int E, int dtype, int nranges,
                      csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && T >= 0 && D > 0 && E > 0 && nranges > 0, "gate_bwd_dw: bad arguments");
  CSMOE_CHECK_ARG(k_gate_small_ok(D, E, dtype, x, nullptr), "gate_bwd_dw: needs E <= 4, D a multiple of the 16-byte chunk, aligned operands");
  CSMOE_CHECK_ARG(T == 0 || (dlogits && x && partial), "gate_bwd_dw: null pointer");
  return k_gate_small_dw(dlogits, x, partial, T, D, E, dtype, nranges, (hipStream_t)stream);
}

int csmoe_expert_order(const int32_t* offsets, int E, int32_t* order, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(E > 0 && E <= 8192 && offsets && order, "expert_order: bad arguments");
  return k_expert_order(offsets, E, order, (hipStream_t)stream);
}

int csmoe_chunk_offsets(const int32_t* offsets, int E, int P, int align, int32_t* out, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(E > 0 && P > 0 && align > 0 && (int64_t)E * P < (1 << 24) && offsets && out, "chunk_offsets: bad arguments");
  return k_chunk_offsets(offsets, E, P, align, out, (hipStream_t)stream);
}

int csmoe_dense_wgrad(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int Na, int Nb, void* C, int64_t ldc,
                      int dtype, int out_dtype, int accumulate, int force_generic, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && dtype_ok(out_dtype) && M >= 0 && Na > 0 && Nb > 0, "dense_wgrad: bad arguments");
  CSMOE_CHECK_ARG(C && (M == 0 || (A && B)), "dense_wgrad: null pointer");
  CSMOE_CHECK_ARG(!(dtype == CSMOE_F32 && out_dtype != CSMOE_F32), "dense_wgrad: fp32 inputs need fp32 output");
  hipStream_t st = (hipStream_t)stream;
  if (!force_generic && dtype == CSMOE_BF16 && gg_fast_wgrad_ok(lda, ldb, ldc, M, Na, Nb, A, B)) {
    if (use_v2_wgrad(M, Na, Nb))
      return gg8_wgrad(A, lda, B, ldb, nullptr, 1, Na, Nb, nullptr, ldc, out_dtype, accumulate, M, C, st);
    return gg_fast_wgrad(A, lda, B, ldb, nullptr, 1, Na, Nb, nullptr, ldc, out_dtype, accumulate, M, C, st);
  }
  return gg_generic_wgrad(A, lda, B, ldb, nullptr, 1, Na, Nb, nullptr, ldc, dtype, out_dtype, accumulate, M, C, st);
}

int csmoe_grouped_colsum(const void* G, int64_t ldg, const int32_t* offsets, int E, int N, void* const* out_ptrs, int dtype,
                         int out_dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && dtype_ok(out_dtype) && E > 0 && N > 0, "grouped_colsum: bad arguments");
  // G may be null when every bin is empty (the row count lives in `offsets`, on the device)
  CSMOE_CHECK_ARG(offsets && out_ptrs, "grouped_colsum: null pointer");
  return k_colsum(G, ldg, offsets, E, 0, N, out_ptrs, nullptr, dtype, out_dtype, (hipStream_t)stream);
}

int csmoe_dense_colsum(const void* G, int64_t ldg, int M, int N, void* out, int dtype, int out_dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && dtype_ok(out_dtype) && M >= 0 && N > 0, "dense_colsum: bad arguments");
  CSMOE_CHECK_ARG(out && (M == 0 || G), "dense_colsum: null pointer");
  return k_colsum(G, ldg, nullptr, 1, M, N, nullptr, out, dtype, out_dtype, (hipStream_t)stream);
}

int csmoe_softplus_mean(const void* y, void* aff, int R, int D, int dtype, int aff_dtype, int precise, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && R >= 0 && D > 0, "softplus_mean: bad arguments");
  CSMOE_CHECK_ARG(aff_dtype == dtype || aff_dtype == CSMOE_F32, "softplus_mean: affinities are in the rows' dtype or fp32");
  CSMOE_CHECK_ARG(R == 0 || (y && aff), "softplus_mean: null pointer");
  return k_softplus_mean(y, aff, R, D, dtype, aff_dtype, precise, (hipStream_t)stream);
}

int csmoe_softplus_mean_bwd(const void* y, const void* daff, const void* dy_add, void* dy, int R, int D, int dtype, int aff_dtype, int precise,
                            csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && R >= 0 && D > 0, "softplus_mean_bwd: bad arguments");
  CSMOE_CHECK_ARG(aff_dtype == dtype || aff_dtype == CSMOE_F32, "softplus_mean_bwd: affinities are in the rows' dtype or fp32");
  CSMOE_CHECK_ARG(R == 0 || (y && daff && dy), "softplus_mean_bwd: null pointer");
  return k_softplus_mean_bwd(y, daff, dy_add, dy, R, D, dtype, aff_dtype, precise, (hipStream_t)stream);
}

}  // extern "C"

int csmoe_pair_cosine(const void* y, float* tok_loss, int T, int K, int D, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && T >= 0 && K > 0 && K <= 8 && D > 0, "pair_cosine: bad arguments (K <= 8)");
  CSMOE_CHECK_ARG(T == 0 || (y && tok_loss), "pair_cosine: null pointer");
  return k_pair_cosine(y, tok_loss, T, K, D, dtype, (hipStream_t)stream);
}

int csmoe_pair_cosine_bwd(const void* y, const float* gscale, void* dy, int T, int K, int D, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && T >= 0 && K > 0 && K <= 8 && D > 0, "pair_cosine_bwd: bad arguments (K <= 8)");
  CSMOE_CHECK_ARG(T == 0 || (y && gscale && dy), "pair_cosine_bwd: null pointer");
  return k_pair_cosine_bwd(y, gscale, dy, T, K, D, dtype, (hipStream_t)stream);
}

int csmoe_layernorm_gate(const void* x, const void* gamma, const void* beta, float eps, void* xn, float* mean, float* rstd,
                         int T, int D, int dtype, const void* w_gate, void* logits, int E, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "layernorm_gate: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && D <= k_layernorm_max_d(dtype) && D % (dtype == CSMOE_BF16 ? 8 : 4) == 0,
                  "layernorm_gate: D=%d unsupported (multiple of %d, at most %d)", D, dtype == CSMOE_BF16 ? 8 : 4,
                  k_layernorm_max_d(dtype));
  CSMOE_CHECK_ARG(eps >= 0.f, "layernorm_gate: eps < 0");
  CSMOE_CHECK_ARG((w_gate == nullptr) == (logits == nullptr) && (w_gate == nullptr || E > 0), "layernorm_gate: w_gate / logits / E");
  if (T == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(x && xn && mean && rstd, "layernorm_gate: null pointer");
  CSMOE_CHECK_ARG(((((uintptr_t)x | (uintptr_t)xn | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w_gate) & 15) == 0),
                  "layernorm_gate: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = k_layernorm_fwd(x, gamma, beta, eps, xn, mean, rstd, T, D, dtype, st);
  if (rc || !w_gate) return rc;
  return csmoe_gate_logits(xn, w_gate, logits, T, D, E, dtype, stream);
}

int csmoe_layernorm_bwd_blocks(int T) { return k_layernorm_bwd_blocks(T); }

int csmoe_layernorm_bwd(const void* dxn, const void* dxn2, const void* x, const void* gamma, const float* mean, const float* rstd,
                        const void* add, void* dx, float* partial, int T, int D, int dtype, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype), "layernorm_bwd: bad dtype %d", dtype);
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && D <= k_layernorm_max_d(dtype) && D % (dtype == CSMOE_BF16 ? 8 : 4) == 0,
                  "layernorm_bwd: D=%d unsupported", D);
  CSMOE_CHECK_ARG(partial, "layernorm_bwd: null pointer");
  CSMOE_CHECK_ARG(T == 0 || (dxn && x && mean && rstd && dx), "layernorm_bwd: null pointer");
  CSMOE_CHECK_ARG(((((uintptr_t)x | (uintptr_t)dxn | (uintptr_t)dxn2 | (uintptr_t)gamma | (uintptr_t)add | (uintptr_t)dx) & 15) == 0),
                  "layernorm_bwd: operands must be 16-byte aligned");
  return k_layernorm_bwd(dxn, dxn2, x, gamma, mean, rstd, add, dx, partial, T, D, dtype, (hipStream_t)stream);
}

// ---- fp32 residual stream around bf16 activations (pretrain stack under autocast): mixed-precision forms of the block kernels
int csmoe_layernorm_gate_mixed(const float* x, const float* gamma, const float* beta, float eps, void* xn, float* mean, float* rstd,
                               int T, int D, const void* w_gate, void* logits, int E, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && D <= k_layernorm_max_d(CSMOE_BF16) && D % 8 == 0,
                  "layernorm_gate_mixed: D=%d unsupported (multiple of 8, at most %d)", D, k_layernorm_max_d(CSMOE_BF16));
  CSMOE_CHECK_ARG(eps >= 0.f, "layernorm_gate_mixed: eps < 0");
  CSMOE_CHECK_ARG((w_gate == nullptr) == (logits == nullptr) && (w_gate == nullptr || E > 0), "layernorm_gate_mixed: w_gate / logits / E");
  if (T == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(x && xn && mean && rstd, "layernorm_gate_mixed: null pointer");
  CSMOE_CHECK_ARG(((((uintptr_t)x | (uintptr_t)xn | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w_gate) & 15) == 0),
                  "layernorm_gate_mixed: operands must be 16-byte aligned");
  int rc = k_layernorm_fwd_mixed(x, gamma, beta, eps, xn, mean, rstd, T, D, (hipStream_t)stream);
  if (rc || !w_gate) return rc;
  return csmoe_gate_logits(xn, w_gate, logits, T, D, E, CSMOE_BF16, stream);
}

int csmoe_layernorm_bwd_mixed(const void* dxn, const void* dxn2, const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* add, float* dx, float* partial, int T, int D, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(T >= 0 && D > 0 && D <= k_layernorm_max_d(CSMOE_BF16) && D % 8 == 0, "layernorm_bwd_mixed: D=%d unsupported", D);
  CSMOE_CHECK_ARG(partial, "layernorm_bwd_mixed: null pointer");
  CSMOE_CHECK_ARG(T == 0 || (dxn && x && mean && rstd && dx), "layernorm_bwd_mixed: null pointer");
  CSMOE_CHECK_ARG(((((uintptr_t)x | (uintptr_t)dxn | (uintptr_t)dxn2 | (uintptr_t)gamma | (uintptr_t)add | (uintptr_t)dx) & 15) == 0),
                  "layernorm_bwd_mixed: operands must be 16-byte aligned");
  return k_layernorm_bwd_mixed(dxn, dxn2, x, gamma, mean, rstd, add, dx, partial, T, D, (hipStream_t)stream);
}

int csmoe_combine_mixed(const void* y, const int32_t* slot_of, const int32_t* idx, const float* w, const float* residual, float* out,
                        int T, int K, int D, int mode, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(K > 0 && K <= 64 && T >= 0 && D > 0 && D % 8 == 0, "combine_mixed: bad arguments (K <= 64, D a multiple of 8)");
  CSMOE_CHECK_ARG(mode >= 0 && mode <= 2, "combine_mixed: bad mode %d", mode);
  CSMOE_CHECK_ARG(T == 0 || (y && slot_of && w && out), "combine_mixed: null pointer");
  CSMOE_CHECK_ARG(T == 0 || mode == CSMOE_COMBINE_DOT || idx, "combine_mixed: idx required for the sequential rounding rule");
  CSMOE_CHECK_ARG((((uintptr_t)y | (uintptr_t)residual | (uintptr_t)out) & 15) == 0, "combine_mixed: operands must be 16-byte aligned");
  return k_combine_mixed(y, slot_of, idx, w, residual, out, T, K, D, mode, (hipStream_t)stream);
}

int csmoe_combine_bwd_mixed(const float* dout, const void* y, const int32_t* slot_of, const float* w, void* dy, float* dw, int T, int K,
                            int D, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(K > 0 && T >= 0 && D > 0 && D % 8 == 0, "combine_bwd_mixed: bad arguments (D a multiple of 8)");
  CSMOE_CHECK_ARG(T == 0 || (dout && slot_of && dy), "combine_bwd_mixed: null pointer");
  CSMOE_CHECK_ARG((((uintptr_t)y | (uintptr_t)dout | (uintptr_t)dy) & 15) == 0, "combine_bwd_mixed: operands must be 16-byte aligned");
  return k_combine_bwd_mixed(dout, y, slot_of, w, dy, dw, T, K, D, (hipStream_t)stream);
}

int csmoe_dispatch_rows_bwd_mixed(const void* dxs, const int32_t* slot_of, int K, const void* add, float* dx, int T, int D,
                                  csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(K > 0 && K <= 64 && T >= 0 && D > 0 && D % 8 == 0, "dispatch_rows_bwd_mixed: bad arguments (K <= 64, D a multiple of 8)");
  CSMOE_CHECK_ARG(T == 0 || (dxs && slot_of && dx), "dispatch_rows_bwd_mixed: null pointer");
  CSMOE_CHECK_ARG((((uintptr_t)dxs | (uintptr_t)add | (uintptr_t)dx) & 15) == 0, "dispatch_rows_bwd_mixed: operands must be 16-byte aligned");
  return k_dispatch_rows_bwd_mixed(dxs, slot_of, K, add, dx, T, D, (hipStream_t)stream);
}

int csmoe_widen_sum(const void* a, const void* b, const void* c, float* out, int64_t n, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(n >= 0, "widen_sum: bad length");
  CSMOE_CHECK_ARG(n == 0 || (a && out), "widen_sum: null pointer");
  CSMOE_CHECK_ARG(!c || b, "widen_sum: streams are given in order (c without b)");
  CSMOE_CHECK_ARG((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0, "widen_sum: operands must be 16-byte aligned");
  return k_widen_sum(a, b, c, out, n, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------------ MXFP8 path
int csmoe_quantize_mxfp8(const void* x, const void* const* x_ptrs, int E, int64_t ldx, int R, int C, int dtype, int transpose,
                         void* q, void* s, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && E >= 1 && R >= 0 && C >= 0, "quantize_mxfp8: bad arguments");
  CSMOE_CHECK_ARG((x_ptrs != nullptr) || (x != nullptr && E == 1) || R == 0 || C == 0, "quantize_mxfp8: no input");
  if (R == 0 || C == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(q && s && ldx >= C, "quantize_mxfp8: null output or leading dimension too small");
  const int es = dtype == CSMOE_BF16 ? 2 : 4;
  if (transpose) {
    if (R % 32 != 0) { csmoe_set_error("quantize_mxfp8: transposed form needs R %% 32 == 0 (R=%d)", R); return CSMOE_ERR_UNSUPPORTED; }
  } else {
    if (C % 32 != 0 || (ldx * es) % 16 != 0 || (x && ((uintptr_t)x & 15))) {
      csmoe_set_error("quantize_mxfp8: needs C %% 32 == 0 and 16-byte aligned rows (C=%d ldx=%lld)", C, (long long)ldx);
      return CSMOE_ERR_UNSUPPORTED;
    }
  }
  return k_quantize_mxfp8(x_ptrs, x, E, ldx, R, C, dtype, transpose, q, s, (hipStream_t)stream);
}

int csmoe_quantize_mxfp8_both(const void* x, const void* const* x_ptrs, int E, int64_t ldx, int R, int C, int dtype, void* q, void* s,
                              void* qt, void* st, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(dtype_ok(dtype) && E >= 1 && R >= 0 && C >= 0, "quantize_mxfp8_both: bad arguments");
  CSMOE_CHECK_ARG((x_ptrs != nullptr) || (x != nullptr && E == 1) || R == 0 || C == 0, "quantize_mxfp8_both: no input");
  if (R == 0 || C == 0) return CSMOE_OK;
  CSMOE_CHECK_ARG(q && s && qt && st && ldx >= C, "quantize_mxfp8_both: null output or leading dimension too small");
  const int es2 = dtype == CSMOE_BF16 ? 2 : 4;
  if (R % 32 != 0 || C % 32 != 0 || (ldx * es2) % 16 != 0 || (x && ((uintptr_t)x & 15))) {
    csmoe_set_error("quantize_mxfp8_both: needs R %% 32 == 0, C %% 32 == 0 and 16-byte aligned rows (R=%d C=%d)", R, C);
    return CSMOE_ERR_UNSUPPORTED;
  }
  return k_quantize_mxfp8_both(x_ptrs, x, E, ldx, R, C, dtype, q, s, qt, st, (hipStream_t)stream);
}

static int fp8_shape_ok(int64_t lda, int64_t ldas, int64_t ldb, int64_t ldbs, int64_t ldc, int N, int Kd, const void* A, const void* C,
                        const char* who) {
  if (Kd % 128 != 0 || N % 8 != 0 || lda % 16 != 0 || ldb % 16 != 0 || ldas % 4 != 0 || ldbs % 4 != 0 || ldc % 8 != 0 ||
      ((uintptr_t)A & 15) || ((uintptr_t)C & 15)) {
    csmoe_set_error("%s: needs Kd %% 128 == 0, N %% 8 == 0, 16-byte aligned operands (N=%d Kd=%d)", who, N, Kd);
    return CSMOE_ERR_UNSUPPORTED;
  }
  return CSMOE_OK;
}

int csmoe_grouped_gemm_mxfp8(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* const* bq_ptrs,
                             const void* const* bs_ptrs, int64_t ldb, int64_t ldbs, const void* const* bias_ptrs,
                             const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2, const void* aux, int64_t ldc,
                             int epilogue, int act, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(E > 0 && M >= 0 && N > 0 && Kd > 0, "grouped_gemm_mxfp8: bad shape E=%d M=%d N=%d Kd=%d", E, M, N, Kd);
  CSMOE_CHECK_ARG(epilogue >= 0 && epilogue <= 3 && act >= 0 && act <= 5, "grouped_gemm_mxfp8: bad epilogue/act");
  CSMOE_CHECK_ARG(bq_ptrs && bs_ptrs && offsets && (M == 0 || (Aq && As && (C || (epilogue == CSMOE_EPI_BIAS_ACT && C2)))),
                  "grouped_gemm_mxfp8: null pointer");
  CSMOE_CHECK_ARG(M == 0 || epilogue != CSMOE_EPI_ACTGRAD || aux, "grouped_gemm_mxfp8: ACTGRAD epilogue needs aux");
  CSMOE_CHECK_ARG(lda >= Kd && ldb >= Kd && ldc >= N && ldas >= Kd / 32 && ldbs >= Kd / 32, "grouped_gemm_mxfp8: leading dimension too small");
  if (M == 0) return CSMOE_OK;
  if (int rc = fp8_shape_ok(lda, ldas, ldb, ldbs, ldc, N, Kd, Aq, C ? C : C2, "grouped_gemm_mxfp8")) return rc;
  return gg8f_rowspace(Aq, lda, As, ldas, bq_ptrs, bs_ptrs, ldb, ldbs, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act,
                       nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int csmoe_dense_gemm_mxfp8(const void* Aq, int64_t lda, const void* As, int64_t ldas, const void* Bq, const void* Bs, int64_t ldb,
                           int64_t ldbs, const void* bias, int M, int N, int Kd, void* C, void* C2, const void* aux, int64_t ldc,
                           int epilogue, int act, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(M >= 0 && N > 0 && Kd > 0, "dense_gemm_mxfp8: bad shape");
  CSMOE_CHECK_ARG(epilogue >= 0 && epilogue <= 3 && act >= 0 && act <= 5, "dense_gemm_mxfp8: bad epilogue/act");
  CSMOE_CHECK_ARG(M == 0 || (Aq && As && Bq && Bs && (C || (epilogue == CSMOE_EPI_BIAS_ACT && C2))), "dense_gemm_mxfp8: null pointer");
  CSMOE_CHECK_ARG(M == 0 || epilogue != CSMOE_EPI_ACTGRAD || aux, "dense_gemm_mxfp8: ACTGRAD epilogue needs aux");
  if (M == 0) return CSMOE_OK;
  if (int rc = fp8_shape_ok(lda, ldas, ldb, ldbs, ldc, N, Kd, Aq, C ? C : C2, "dense_gemm_mxfp8")) return rc;
  return gg8f_rowspace(Aq, lda, As, ldas, nullptr, nullptr, ldb, ldbs, nullptr, nullptr, 1, M, N, Kd, C, C2, aux, ldc, epilogue, act,
                       Bq, Bs, bias, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------------ fp32 master weights in the GEMM
int csmoe_grouped_gemm_f32w(const void* A, int64_t lda, const void* const* b_ptrs, int64_t ldb, void* const* b_copy_ptrs,
                            const void* const* bias_ptrs, const int32_t* offsets, int E, int M, int N, int Kd, void* C, void* C2,
                            const void* aux, int64_t ldc, int epilogue, int act, csmoe_stream_t stream) {
  CSMOE_CHECK_ARG(E > 0 && M >= 0 && N > 0 && Kd > 0, "grouped_gemm_f32w: bad shape E=%d M=%d N=%d Kd=%d", E, M, N, Kd);
  CSMOE_CHECK_ARG(epilogue >= 0 && epilogue <= 4 && act >= 0 && act <= 5, "grouped_gemm_f32w: bad epilogue/act");
  CSMOE_CHECK_ARG(b_ptrs && offsets && (M == 0 || (A && (C || ((epilogue == CSMOE_EPI_BIAS_ACT || epilogue == CSMOE_EPI_ROUND_BIAS32_ACT) && C2)))),
                  "grouped_gemm_f32w: null pointer");
  CSMOE_CHECK_ARG(M == 0 || epilogue != CSMOE_EPI_ACTGRAD || aux, "grouped_gemm_f32w: ACTGRAD epilogue needs aux");
  CSMOE_CHECK_ARG(lda >= Kd && ldc >= N && ldb >= N, "grouped_gemm_f32w: leading dimension too small");
  if (M == 0) return CSMOE_OK;
  const void* Cany = C ? C : C2;
  if (N % 8 || Kd % 8 || lda % 8 || ldb % 4 || ldc % 8 || ((uintptr_t)A & 15) || ((uintptr_t)Cany & 15) ||
      (int64_t)M * lda * 2 >= 0x80000000ll || (int64_t)Kd * ldb * 4 >= 0x80000000ll) {
    csmoe_set_error("grouped_gemm_f32w: needs N %% 8 == 0, Kd %% 8 == 0, 16-byte aligned rows and operands under 2 GiB each "
                    "(N=%d Kd=%d); cast the weights and use csmoe_grouped_gemm", N, Kd);
    return CSMOE_ERR_UNSUPPORTED;
  }
  // With a bf16 copy requested: the launch over every expert's FIRST row tile converts the fp32 masters in its tile fill and writes
  // the copy; the other row tiles run the LDS-DMA kernel over that copy (same stream: the copy is complete when they start).
  // CSMOE_F32W_SPLIT=0 (A/B): every row tile converts, as in round 2.
  static const bool split = [] { const char* e = getenv("CSMOE_F32W_SPLIT"); return !e || atoi(e) != 0; }();
  if (b_copy_ptrs && split && (int64_t)Kd * N * 2 < 0x80000000ll) {
    if (int rc = gg8c_rowspace(A, lda, b_ptrs, ldb, b_copy_ptrs, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act, nullptr,
                               nullptr, nullptr, (hipStream_t)stream, 1))
      return rc;
    return gg8_rowspace_rest(A, lda, (const void* const*)b_copy_ptrs, CSMOE_B_KN, N, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc,
                             epilogue, act, (hipStream_t)stream);
  }
  return gg8c_rowspace(A, lda, b_ptrs, ldb, b_copy_ptrs, bias_ptrs, offsets, E, M, N, Kd, C, C2, aux, ldc, epilogue, act, nullptr,
                       nullptr, nullptr, (hipStream_t)stream, 0);
}
