/*
 * afr.h -- C ABI of libafr.so, the MI355X (gfx950) training hot path of ai-font-renderer.
 *
 * The reference has no FFI/plugin layer: its hot path is reached only through Python
 * (reference model.py:129-204 forward, :268-270 loss, :292-310 step).  Each entry point below
 * names the reference code it replaces.  Conventions:
 *   - plain pointers and sizes, no torch types; every device buffer is CALLER-allocated and
 *     caller-owned (the library never frees or retains them past afr_plan_destroy);
 *   - all work is enqueued on the caller's hipStream_t (passed as void*), asynchronously: no
 *     hidden device synchronisation, no allocation inside any call except afr_plan_create;
 *   - every call returns 0 on success or a negative AFR_E* code; afr_last_error() gives the text
 *     (thread-local).  Nothing throws across the boundary;
 *   - a plan is not thread-safe: one plan per rank, one process per GPU.
 */
#ifndef AFR_H
#define AFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFR_VERSION 1

enum { AFR_OK = 0, AFR_EINVAL = -1, AFR_ESTATE = -2, AFR_EHIP = -3, AFR_EUNSUPPORTED = -4 };

enum { AFR_KIND_SHEET = 0,   /* AttentionFontRenderer, model.py:129-204                         */
       AFR_KIND_GLYPH = 1,   /* per-glyph MLP (BASELINE.json configs C1-C4)                     */
       AFR_KIND_PIXEL = 2 }; /* per-pixel-token transformer (BASELINE.json configs[4]): embed_dim = d_model
                                (<= 512, = 64 * heads), fc_dim = ff width, n_hidden = blocks, out_h * out_w = pixel tokens.
                                Every entry point serves it; one backward stage per block, last block first.                */
enum { AFR_F32 = 0,          /* exact-f32 MFMA everywhere: parity mode (<=1e-4 vs reference)     */
       AFR_BF16 = 1 };       /* bf16 MFMA operands, f32 accumulate, f32 master weights           */
enum { AFR_TARGET_U8 = 0,    /* 8-bit pixels as stored in the BMPs; k/255.0f on device           */
       AFR_TARGET_F32 = 1 }; /* float32 targets as helpers.load_string_dataset returns them      */

#define AFR_MAX_HIDDEN 8

typedef struct afr_config {
    int32_t kind;        /* AFR_KIND_*                                                            */
    int32_t dtype;       /* AFR_F32 | AFR_BF16                                                    */
    int32_t max_batch;   /* largest B any call will pass                                          */
    int32_t vocab;       /* embedding rows (128, model.py:136)                                    */
    int32_t embed_dim;   /* EMBEDDING_DIM (32, model.py:79)                                       */
    int32_t out_h, out_w;/* SHEET_HEIGHT x SHEET_WIDTH (model.py:64-65) or glyph bitmap size       */
    /* sheet model */
    int32_t max_length;  /* MAX_CHARS_PER_SHEET (model.py:66)                                     */
    int32_t heads;       /* NUM_ATTENTION_HEADS (model.py:81)                                     */
    int32_t fc_dim;      /* fc1 width (64, model.py:148)                                          */
    float p_embed, p_attn, p_fc; /* dropout rates (model.py:137,144,149)                          */
    float ln_eps;        /* LayerNorm eps (1e-5)                                                  */
    /* glyph model */
    int32_t n_hidden;
    int32_t hidden[AFR_MAX_HIDDEN];
    int32_t n_fonts;     /* 0 = no font-id embedding                                              */
    /* dropout stream */
    uint64_t seed;
    int32_t rank;        /* data-parallel rank: gives each replica its own dropout stream         */
    int32_t reserved;    /* bit 0: keep the optimizer un-fused in afr_train_step (gradients of every tensor are
                            then materialised; otherwise the sheet model's fc_output.weight is updated inside its
                            weight-gradient GEMM and its gradient never reaches HBM)
                            bit 1: one launch per product in backward (no grouped dW+dX launches): A/B measurements
                            bit 2: small one-hidden-layer glyph nets through the generic per-layer kernels instead of
                            the fused whole-step kernel of afr_train_step (A/B measurements, parity cross-checks)
                            bit 3: OPT IN to in-launch split-K (afr_op_gemm_fix) for the sheet model's fc_output products
                            that have no fused loss / optimizer tail (today: the input gradient)
                            bit 4: the glyph nets' folded first layer backward through the weight-gradient GEMM + post-pass
                            instead of the fused kernel (A/B measurements, parity cross-checks)
                            bit 5: weight gradients of grouped 256x256 launches as split-K partial slabs summed by the grouped
                            reduce, instead of the cooperative split-K whose slices meet inside the launch (A/B measurements,
                            parity cross-checks: the two give bitwise equal results)
                            bit 6: the glyph nets' first layer through the gather kernel + dense h1 in training steps too,
                            instead of the (character, font) combination table gathered inside the consuming products
                            bit 7: ReLU masks of the input-gradient products read from the stored activations instead of the
                            bit masks the forward epilogues leave (bits 6, 7: A/B measurements; bitwise equal results)      */
} afr_config;

typedef struct afr_plan afr_plan;

int afr_version(void);
const char* afr_last_error(void);

/* Build the launch plan (shapes, workspace carve-up).  Replaces nn.Module construction,
 * model.py:130-156 (no parameters are created: see afr_bind). */
int afr_plan_create(const afr_config* cfg, afr_plan** out);
int afr_plan_destroy(afr_plan* plan);

/* Flat parameter layout: all tensors of state_dict(), in state_dict order, live in
 * ONE float32 buffer of afr_param_elems() elements; tensor i starts at offset[i] (multiple of 64
 * elements).  Gradients and the two AdamW moments use the same layout. */
int64_t afr_param_elems(const afr_plan* plan);
int afr_param_count(const afr_plan* plan);
int afr_param_info(const afr_plan* plan, int index, char* name, int name_cap, int64_t* offset,
                   int64_t* numel, int32_t* ndim, int64_t shape[4]);

size_t afr_workspace_bytes(const afr_plan* plan);

/* Attach caller-owned device buffers.  params/grads/exp_avg/exp_avg_sq: afr_param_elems() floats
 * each (grads, moments may be NULL for inference-only use); workspace: afr_workspace_bytes(). */
int afr_bind(afr_plan* plan, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
             void* workspace, size_t workspace_bytes);

/* Re-derive the bf16 shadow weights from the f32 masters after the caller changed them
 * (load_state_dict, helpers.py:100).  No-op in AFR_F32 mode. */
int afr_sync_params(afr_plan* plan, void* stream);

/* forward(x): model.py:158-204.  x int64 [B, L] (sheet; L>max_length truncated, L<max_length
 * zero-padded features) or int64 [B] glyph codes with optional font ids.  y: float32 [B, out_h*out_w]
 * clamped to [0,1], or NULL when only the saved pre-activation is wanted (training).
 * training!=0 enables the three dropouts with the counter-hash stream (seed, rank, step). */
int afr_forward(afr_plan* plan, const int64_t* x, const int64_t* font, int B, int L, float* y,
                int training, uint64_t step, void* stream);

/* compute_loss + the first backward step: F.mse_loss(clamp(u,0,1), target) (model.py:268-270) and
 * d(loss)/du with the inclusive clamp mask.  Uses the pre-activation the last afr_forward left in
 * the workspace.  mean_elems = B_global*out_h*out_w (the mean's denominator; lets data-parallel
 * shards weight a short last batch exactly).  *loss_accum (device float) += this shard's share. */
int afr_loss_grad(afr_plan* plan, const void* target, int target_dtype, int B, int64_t mean_elems,
                  float* loss_accum, void* stream);

/* Entry for a caller-owned loss (torch.autograd): dy = d(loss)/d(y) for the clamped output y [B, pixels], float32.
 * Applies the clamp's gradient mask (0 <= u <= 1, inclusive) and leaves du where afr_backward expects it. */
int afr_set_output_grad(afr_plan* plan, const float* dy, int B, void* stream);

/* loss.backward(): model.py:309.  Overwrites the flat gradient buffer (zero_grad, model.py:292,
 * is implied). */
int afr_backward(afr_plan* plan, void* stream);

/* The same backward in stages (last layer first), for overlapping the gradient all-reduce with the rest of the
 * backward pass under data parallelism: stage k leaves the flat-gradient range [*grad_offset, +*grad_elems) final.
 * Stages must be called in order 0 .. afr_backward_stages()-1. */
int afr_backward_stages(const afr_plan* plan);
int afr_backward_stage(afr_plan* plan, int stage, int64_t* grad_offset, int64_t* grad_elems, void* stream);

/* Training forward with the loss and d(loss)/du computed in the epilogue of the last layer (u never reaches HBM):
 * afr_forward(training) + afr_loss_grad in one pass; follow with afr_backward / afr_backward_stage. */
int afr_forward_loss(afr_plan* plan, const int64_t* x, const int64_t* font, const void* target, int target_dtype,
                     int B, int L, int64_t mean_elems, float* loss_accum, uint64_t step, void* stream);

/* optimizer.step(): torch.optim.AdamW as configured at model.py:273.  t = 1,2,...; grad_scale
 * multiplies every gradient first (1/world after a sum all-reduce; 1 otherwise). */
int afr_adamw_step(afr_plan* plan, float lr, float beta1, float beta2, float eps, float weight_decay,
                   int64_t t, float grad_scale, void* stream);

/* One whole iteration of the loop body model.py:292-310 on this rank's shard:
 * forward(training) -> loss+grad -> backward [-> AdamW when do_step!=0].  With do_step!=0 the loss is fused into
 * the last forward GEMM and, for the sheet model, the AdamW update of fc_output.weight into its dW GEMM. */
int afr_train_step(afr_plan* plan, const int64_t* x, const int64_t* font, const void* target,
                   int target_dtype, int B, int L, int64_t mean_elems, float* loss_accum,
                   uint64_t step, int do_step, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t t, void* stream);

/* Set / read the device-side error word (bit 0: an embedding index outside [0,vocab), the
 * condition on which the reference raises IndexError; model.py:136,167; bit 1: a cooperative split-K
 * workgroup gave up waiting for its partners -- the step's results are invalid).  Reading synchronises and clears. */
int afr_error_flags(afr_plan* plan, void* stream, uint32_t* flags_out);

/* Name and average duration (ms, hipEvent-timed on the launch stream) of the plan's dominant
 * kernel over the calls since the last reset; bench.py's roofline leg.  mode 0: off; 1: time every
 * launch (to find the dominant kernel); 2: from now on time only the kernel that dominated the
 * mode-1 recording (two events per launch of that kernel); 3: like 2 but only every 4th launch of it -- an event
 * record also holds back the next kernel's start (~3.5 us each on MI355X), a sample keeps a timed region honest. */
int afr_profile_dominant(afr_plan* plan, int mode);
int afr_profile_read(afr_plan* plan, char* name, int name_cap, double* avg_ms, int64_t* launches,
                     double* algo_flops, double* algo_bytes);
/* Text table of everything recorded: one "kernel\tlaunches\ttotal_ms\tavg_ms\tflops\tbytes" line per kernel. */
int afr_profile_dump(afr_plan* plan, char* buf, int cap);

/* Inspection for stage-by-stage validation: copy one internal activation buffer of the last call (in the plan's
 * activation dtype: f32, or bf16 in AFR_BF16 mode) to dst (device or host pointer).  *bytes_out = bytes copied. */
enum { AFR_BUF_U = 0,      /* pre-clamp output u [B, pixels]; after afr_loss_grad it holds du                     */
       AFR_BUF_Z = 1,      /* sheet: flattened fc1 features z [B, max_length*fc_dim]                               */
       AFR_BUF_DZ = 2,     /* sheet: gradient w.r.t. z                                                            */
       AFR_BUF_W1T = 4,    /* small glyph nets, bf16: the transposed operand copy W1^T [E][N1] the fused step reads */
       AFR_BUF_W2T = 5,    /*                         and W2^T [N1][pixels] (always bf16)                          */
       AFR_BUF_ACT = 16 }; /* glyph: AFR_BUF_ACT + i = activation i (0 = embedding sum, i = output of hidden i)   */
int afr_debug_copy(afr_plan* plan, int which, void* dst, size_t dst_bytes, size_t* bytes_out, void* stream);
/* Inspection of the sheet model's in-kernel embedding gather (model.py:136,167): runs the front end of an eval forward on
 * x [B, L] and leaves the rows it gathered, Emb[x[b][l]] for l < min(L, max_length), in e0 (device, float32
 * [B][min(L, max_length)][embed_dim]) -- before dropout and the positional encoding.  The north star asks this gather to
 * be bit-exact; the glyph nets expose theirs as AFR_BUF_ACT + 0. */
int afr_debug_sheet_gather(afr_plan* plan, const int64_t* x, int B, int L, float* e0, void* stream);

/* ---- single-kernel entry points (unit tests and re-use by callers either side of the path) ---- */
enum { AFR_GEMM_BIAS = 1, AFR_GEMM_RELU = 2, AFR_GEMM_RELU_MASK = 4, AFR_GEMM_OUT_BF16 = 8,
       AFR_GEMM_A_KSTRIDED = 16, AFR_GEMM_B_KSTRIDED = 32 };
/* C[m][n] = sum_k A(m,k) * B(n,k) (+bias[n]) (relu) (* (aux[m][n] > 0)).
 * A(m,k) = A[m*lda+k], or A[k*lda+m] with AFR_GEMM_A_KSTRIDED; B likewise.  dtype selects f32 or
 * bf16 operands (aux has the operand dtype); C is f32 unless AFR_GEMM_OUT_BF16.  splitk>1 writes
 * splitk partial f32 slabs of M*ldc elements each, to be summed by the caller (afr_op_reduce). */
int afr_op_gemm(int dtype, int flags, const void* A, const void* B, void* C, const float* bias,
                const void* aux, int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                int splitk, void* stream);

/* The same product (bf16 operands, any AFR_GEMM_* epilogue) as split-K WITH the reduction inside the launch, on
 * 256x256 output tiles: the first head_tiles tiles of the kernel's walk are computed whole by one workgroup each, every
 * remaining tile as `splitk` K-slices that are parked in `workspace` and added, in slice order, by whichever slice
 * workgroup arrives last, which then runs the epilogue.  For products whose tile count leaves the chip's last round
 * mostly empty (the sheet model's fc_output forward: 300 tiles = one full round of 256 + 44 tiles cut 5 ways; its input
 * gradient: 100 tiles cut 2 ways).  Measured on R0's input gradient (1024 x 6400 x 19200): 220 us against the 128x128
 * kernel's 276 us with the weight operand warm in the infinity cache, but 343 us against 281 us inside a training step,
 * where the 246 MB weight shadow streams from HBM and one 160-KiB workgroup per CU hides that latency worse than two
 * 64-KiB ones -- so afr_train_step uses it only when config.reserved bit 3 asks for it.  workspace: afr_op_gemm_fix_workspace_bytes() bytes, 16-byte
 * aligned, whose trailing counter words (one per tail tile) are ZERO before the first call; the kernel leaves them zero. */
size_t afr_op_gemm_fix_workspace_bytes(int M, int N, int head_tiles, int splitk);
int afr_op_gemm_fix(int flags, const void* A, const void* B, void* C, const float* bias, const void* aux,
                    int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int head_tiles, int splitk,
                    void* workspace, size_t workspace_bytes, void* stream);
/* A Linear layer's two gradient products exactly as afr_train_step issues them in bf16 mode when the pair fills the chip
 * with 256x256 tiles -- ONE grouped launch (reference: what autograd derives from nn.Linear, model.py:148,152,309):
 *     dW[N][K]      = dy^T . x          f32; the split-K slices are summed INSIDE the launch (cooperative split-K)
 *     db_part[s][N] = column sums of dy over K-slice s, s < splitk (the caller adds the rows)
 *     dX[B][K]      = (dy . W) * (aux > 0)   bf16 (aux NULL: no mask)
 * dy [B][N], x [B][K], W [N][K], aux [B][K]: bf16, row-major, dense.  afr_op_gemm_pair_plan returns AFR_OK with the split
 * and the workspace size, or AFR_EUNSUPPORTED when the shape does not take this form (the step then uses separate
 * launches).  workspace: 256-byte aligned; its contents need not be preserved between calls. */
int afr_op_gemm_pair_plan(int B, int N, int K, int* splitk, size_t* workspace_bytes);
int afr_op_gemm_pair(const void* dy, const void* x, const void* W, const void* aux, float* dW, float* db_part, void* dX,
                     int B, int N, int K, void* workspace, size_t workspace_bytes, void* stream);
int afr_op_reduce(float* dst, const float* slabs, int nslabs, int64_t slab_stride, int64_t n,
                  float scale, int accumulate, void* stream);
/* Several slab reductions in ONE launch (what a backward pass uses for all its split-K / per-block partial gradients):
 * dst[i][0..n[i]) = sum_s slabs[i][s*stride[i] + ...], s < nslabs[i], fixed order.  At most 32 segments: more is an
 * error (AFR_EINVAL), never a silent drop.  n[i] must be a multiple of 4. */
int afr_op_reduce_group(int nseg, float* const* dst, const float* const* slabs, const int* nslabs,
                        const int64_t* stride, const int64_t* n, void* stream);
int afr_op_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int64_t t, float grad_scale,
                 void* stream);
/* scratch: >= 1040 floats, zero before the first call (holds per-block partials and the arrival counter) */
int afr_op_mse_grad(int act_dtype, const void* u, const void* target, int target_dtype, void* du,
                    int64_t rows, int64_t cols, int64_t mean_elems, float* loss_accum, float* scratch,
                    void* stream);
int afr_op_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ---- fp8 building blocks of BASELINE configs[4] ("fp8 MFMA weights on CDNA4"; no counterpart in the reference) ----
 * Operands are OCP e4m3fn bytes (gfx950's native fp8: exponent bias 7, largest finite 448, no infinities) with ONE float
 * scale per tensor: value = scale * e4m3.  afr_op_f32_to_fp8: dst[i] = e4m3(src[i] / scale), round to nearest even,
 * saturating.  afr_op_gemm_fp8: C[m][n] = scale_ab * sum_k A[m*lda+k] * B[n*ldb+k] (+bias[n]) (relu), f32 accumulation on
 * the MX-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: fp8 at twice the bf16 rate),
 * C f32 or bf16 (AFR_GEMM_OUT_BF16); both operands k-contiguous (the forward form x . W^T), K, lda, ldb multiples of 16,
 * N and ldc multiples of 8.  flags: AFR_GEMM_BIAS | AFR_GEMM_RELU | AFR_GEMM_OUT_BF16. */
int afr_op_f32_to_fp8(const float* src, void* dst_e4m3, int64_t n, float scale, void* stream);
int afr_op_gemm_fp8(int flags, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda, int ldb,
                    int ldc, float scale_ab, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFR_H */
