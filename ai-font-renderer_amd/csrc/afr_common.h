// afr_common.h -- shared device helpers and the internal launcher interface of libafr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---------------------------------------------------------------------------------------------
// Dropout counter hash.  Bit-for-bit twin of ai-font-renderer_amd/synth.py:dropout_keep_mask.
// Stands in for the reference's torch bernoulli_ stream (model.py:137,144,149).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t afr_hash32(uint64_t idx, uint32_t key) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t h = lo * 0x9E3779B1u + (key + hi * 0x85EBCA6Bu);
    h ^= h >> 16; h *= 0x21F0AAADu;
    h ^= h >> 15; h *= 0x735A2D97u;
    h ^= h >> 15;
    return h;
}
// keep element idx iff the top 24 hash bits are below thr24 = keep_prob * 2^24
__host__ __device__ __forceinline__ bool afr_keep(uint64_t idx, uint32_t key, uint32_t thr24) {
    return (afr_hash32(idx, key) >> 8) < thr24;
}

static inline uint64_t afr_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
enum { AFR_STREAM_EMBED = 1, AFR_STREAM_ATTN = 2, AFR_STREAM_FC = 3 };
static inline uint32_t afr_dropout_key(uint64_t seed, uint64_t step, uint64_t stream, uint64_t rank) {
    uint64_t v = seed ^ (step * 0x9E3779B97F4A7C15ull) ^ (stream * 0xC2B2AE3D27D4EB4Full) ^ (rank * 0x165667B19E3779F9ull);
    return (uint32_t)(afr_splitmix64(v) & 0xFFFFFFFFull);
}
static inline uint32_t afr_keep_threshold(float keep_prob) { return (uint32_t)((double)keep_prob * 16777216.0); }

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-level finish of a loss reduction: `block_sum` (valid in thread 0) is published as this block's partial; the
// LAST block to arrive sums all partials in block order (deterministic) and adds sum * inv_n to *loss_accum.
// Hand-off form: sc1 (write-through) partial store, drained, agent-scope ticket; the reader uses sc1 loads only
// (cdna_hip_programming.md Guideline 16 R1: no release fence -- it would flush every dirty line of the XCD's L2).
// Must be called by all threads of the block; `sh` is >= blockDim.x floats of LDS free for use.
__device__ __forceinline__ void loss_block_finish(float block_sum, float* partial, unsigned* counter, float* loss_accum,
                                                  float inv_n, float* sh) {
    __shared__ unsigned ticket_;
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket_ = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (ticket_ != gridDim.x - 1) return;
    float a = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
        a += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_accum[0] += sh[0] * inv_n;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next call
    }
}

__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f32_to_bf16(float x) { return (bf16_t)x; }

// ---------------------------------------------------------------------------------------------
// internal launchers (implemented in the .hip files, called from afr_api.cpp)
// ---------------------------------------------------------------------------------------------
struct GemmParams {
    const void* A; const void* B; void* C; const float* bias; const void* aux;
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int flags;            // AFR_GEMM_* bits
    int splitk;           // >=1
    long long slab_stride;  // elements between split-K slabs of C
    // optional fused bias gradient (A must be k-strided): colsum[z*colsum_stride + m] = sum_k A(m,k) over split z
    float* colsum = nullptr;
    long long colsum_stride = 0;
    // optional fused loss (last forward layer of a training step): instead of u = A.B^T + bias the epilogue writes
    // du = 2 (clamp(u,0,1) - t) / mean_elems * [0<=u<=1] and accumulates the MSE (reference model.py:156,268-270)
    const void* mse_target = nullptr;   // [M][N] uint8 or float32
    int mse_target_dtype = 0;           // AFR_TARGET_*
    float mse_inv_n = 0.f;              // 1 / mean_elems
    float* mse_partial = nullptr;       // per-block partial sums (>= grid floats)
    unsigned* mse_counter = nullptr;    // arrival counter, zero on entry, re-armed by the last block
    float* mse_loss_accum = nullptr;    // device scalar: += sum(partials) / mean_elems
    // optional fused optimizer (dW GEMMs only, single GPU): C is the weight's gradient tile; instead of storing it
    // the epilogue applies AdamW to the matching tile of p/m/v (same [M][ldc] layout) and refreshes the bf16 shadow
    float* ad_p = nullptr; float* ad_m = nullptr; float* ad_v = nullptr; bf16_t* ad_shadow = nullptr;
    float ad_decay = 1.f, ad_b1 = 0.f, ad_b2 = 0.f, ad_eps = 0.f, ad_step = 0.f, ad_rsqrt_bc2 = 1.f;
    // optional in-launch split-K (bf16, 256x256 tiles; gemm.hip gemm_bf16_256_body): the first head_tiles tiles of the
    // walk are computed whole, each remaining tile as `splitk` K-slices parked in fix_ws (256 KiB per slice) and summed in
    // slice order by the slice block that arrives last (fix_cnt: one zeroed counter per tail tile, re-armed by the kernel).
    // The output then takes the full epilogue (bias / ReLU / mask / bf16), unlike plain split-K's f32 partial slabs.
    int head_tiles = 0; float* fix_ws = nullptr; unsigned* fix_cnt = nullptr;
    // optional COOPERATIVE split-K (bf16, 256x256 tiles, grouped launches; gemm.hip gemm_bf16_256_body): the `splitk` (2, 4 or
    // 8) slice workgroups of a tile park their accumulators in coop_ws (256 KiB per slice, write-through), arrive on the
    // tile's counter and WAIT for each other (every workgroup of the launch is resident: one per CU); then slice z adds rows
    // [z, z+1) * 128 / splitk ... of every wave's accumulator image over all slices, in slice order, and finishes that strip:
    // AdamW on p/m/v (ad_p set) or a plain store into C.  No partial slabs leave the kernel and no second kernel re-reads
    // them.  coop_cnt: one counter per tile, monotonic: a launch waits for coop_target = launches so far * splitk.
    float* coop_ws = nullptr; unsigned* coop_cnt = nullptr; unsigned coop_target = 0; uint32_t* err = nullptr;
    // optional row gather (bf16): memory row r of the operand (a row m of a k-contiguous A, a row k of a k-strided B, a row m
    // of aux) is row rowmap[r] of the table the operand pointer names.  The glyph nets' first-layer output as a combination
    // table (elementwise.hip glyph_combo_kernel).  Supported: A k-contiguous on the 256x128 ring kernel, B k-strided on the
    // 256x256 kernel, aux with a bf16 output; the launchers refuse anything else.
    const int* a_rowmap = nullptr; const int* b_rowmap = nullptr; const int* aux_rowmap = nullptr;
    float out_scale = 1.f;     // fp8 products: scale_a * scale_b, applied to the accumulators ahead of bias / ReLU
    // optional ReLU mask as BITS (bf16 output only; N a multiple of 8): a forward layer with AFR_GEMM_RELU leaves bit (n & 7)
    // of mask_out[m * ldmask + n / 8] = [stored activation (m, n) > 0]; an input-gradient product with AFR_GEMM_RELU_MASK
    // reads mask_in in that layout INSTEAD of the activation aux (1/16 of its bytes: C3's 16.8 MB mask read becomes 1 MB)
    unsigned char* mask_out = nullptr; const unsigned char* mask_in = nullptr; int ldmask = 0;
#ifdef AFR_GEMM_TIMING
    int dbg_slot = 0;     // kernel-development builds: which 1024-block region of the stamp buffer this launch writes
#endif
};
constexpr uint32_t AFR_ERR_INDEX = 1u, AFR_ERR_COOP_TIMEOUT = 2u;    // bits of the plan's device error word
// torch.optim.AdamW element update (reference model.py:273,310); shared by adamw_kernel and the fused GEMM epilogue
__device__ __forceinline__ void adamw_elem(float& p, float& m, float& v, float g, float decay, float b1, float b2,
                                           float eps, float step_size, float rsqrt_bc2) {
    p *= decay;
    m = m + (g - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * rsqrt_bc2 + eps;
    p -= step_size * (m / denom);
}
hipError_t afr_launch_gemm(int dtype, const GemmParams& p, hipStream_t s);
hipError_t afr_launch_gemm_fp8(const GemmParams& p, hipStream_t s);          // e4m3 x e4m3, both k-contiguous (gemm.hip fp8k)
hipError_t afr_launch_f32_to_fp8(const float* src, unsigned char* dst, long long n, float inv_scale, hipStream_t s);
// several independent products in one launch (falls back to one launch each when one of them does not qualify)
bool afr_gemm_groupable(int dtype, const GemmParams& p);
hipError_t afr_launch_gemm_group(int dtype, const GemmParams* ps, int n, int tile256, hipStream_t s);
void afr_gemm_pair_plan(int B, int n_out, int k_in, int* tile256, int* splitk);
// in-launch split-K plan for one bf16 product on 256x256 tiles: true when it beats the ring kernels by the launch model;
// workspace need is (tiles - head_tiles) * splitk * AFR_FIX_SLICE_BYTES, never more than AFR_FIX_WS_BYTES
constexpr size_t AFR_FIX_SLICE_BYTES = 256 * 256 * 4, AFR_FIX_MAX_SLICES = 256, AFR_FIX_WS_BYTES = AFR_FIX_SLICE_BYTES * AFR_FIX_MAX_SLICES;
bool afr_gemm_fix_plan(int M, int N, int K, int* head_tiles, int* splitk);
hipError_t afr_launch_gemm_fix(const GemmParams& p, hipStream_t s);
const char* afr_gemm_kernel_name(int dtype, const GemmParams& p);
bool afr_gemm_wide_ok(int M, int N, int K);      // a bf16 product of this shape (no split, no fused optimizer) runs on the 256x128 ring kernel

hipError_t afr_launch_reduce(float* dst, const float* slabs, int nslabs, long long slab_stride, long long n,
                             float scale, int accumulate, hipStream_t s);
// grouped reduction: every gradient tensor that was produced as partial slabs, in ONE launch
struct RSeg {
    float* dst; const float* src; long long stride; long long n4; int nslabs; int blk0; int nblk; int deep;
    // optional (with the fused optimizer): a TRANSPOSED bf16 copy of this [tN][tK] weight, shT[k][n], kept current too
    bf16_t* shT = nullptr; int tN = 0, tK = 0;
};
// every gradient tensor of the deepest glyph net (AFR_MAX_HIDDEN + 1 Linears: weight + bias each) plus the folded first
// layer's extra segments (compact dW1, embedding and font partials) fits; afr_api.hip static_asserts it
constexpr int AFR_RT_MAXSEG = 32;
constexpr int AFR_L1F_MAX_SPLIT = 4;   // column ranges per row block of the fused first-layer backward (2 segments each)
struct RTable {
    int nseg = 0; int nblocks = 0; int overflow = 0;
    // optional fused optimizer: the summed gradient is not stored; AdamW is applied to p/m/v at the same flat offset
    // (offset of seg.dst from `gbase`)
    int adam = 0; float ad_decay, ad_b1, ad_b2, ad_eps, ad_step, ad_rsqrt_bc2;
    const float* gbase; float* P; float* M; float* V; bf16_t* shadow;
    RSeg seg[AFR_RT_MAXSEG];
};
void afr_rtable_add(RTable& t, float* dst, const float* src, int nslabs, long long stride, long long n);
hipError_t afr_launch_reduce_group(const RTable& t, hipStream_t s);
hipError_t afr_launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long long n, float lr,
                            float beta1, float beta2, float eps, float wd, float bc1, float bc2, float grad_scale,
                            hipStream_t s);
// loss: u (act dtype) [rows][cols] -> du in place or to `du`; per-block partial sums to scratch, then
// a 1-block finisher adds sum(scratch) to *loss_accum (deterministic order).
int afr_mse_blocks(long long rows, long long cols);
// scratch: >= 1028 floats; scratch[1024] (as unsigned) is the arrival counter, zero before the first call
hipError_t afr_launch_mse_grad(int act_dtype, const void* u, const void* target, int target_dtype, void* du,
                               long long rows, long long cols, long long mean_elems, float* loss_accum,
                               float* scratch, hipStream_t s);
hipError_t afr_launch_f32_to_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s);
hipError_t afr_launch_clamp_bwd(int act_dtype, void* u_inout, const float* dy, long long n, hipStream_t s);
hipError_t afr_launch_clamp_out(int act_dtype, const void* u, float* y, long long n, hipStream_t s);
// glyph embedding gather / deterministic scatter-add
hipError_t afr_launch_glyph_embed(int act_dtype, const float* emb, const float* font_emb, const int64_t* x,
                                  const int64_t* font, int B, int E, int vocab, int n_fonts, void* out,
                                  uint32_t* err_flag, hipStream_t s);
int afr_embed_bwd_blocks(int B);
hipError_t afr_launch_glyph_l1_fwd(int act_dtype, const float* emb, const float* font_emb, const float* W1, const float* b1,
                                   const int64_t* x, const int64_t* font, int B, int E, int N1, int vocab, int n_fonts,
                                   float* table, void* h0, void* h1, uint32_t* err_flag, hipStream_t s, void* w1t = nullptr);
hipError_t afr_launch_glyph_combo(const float* emb, const float* font_emb, const float* W1, const float* b1, const int64_t* x,
                                  const int64_t* font, int B, int E, int N1, int vocab, int n_fonts, float* table, void* h1c,
                                  int ld1 /* row stride of h1c, elements */, void* h0c, int* cidx, uint32_t* err_flag, hipStream_t s,
                                  void* w1t);
int afr_glyph_k0(int E, int vocab, int n_fonts);
int afr_glyph_l1_bwd_blocks(int N1);
// the same backward in one kernel (throughput mode, gemm.hip: glyph_l1_bwd_fused_kernel); W1T = bf16 [E][N1] from the forward
bool afr_glyph_l1_bwd_fused_eligible(int dtype, int E, int N1, int vocab, int n_fonts);
int afr_glyph_l1_bwd_fused_split(int B, int N1);                                  // column ranges per block of 64 glyphs
int afr_glyph_l1_bwd_fused_blocks(int B, int N1);
long long afr_glyph_l1_bwd_fused_slab_floats(int B, int N1, int vocab, int n_fonts);   // [dW1 nc*E | db1 nc | dTab rows*E], nc = N1 / split
hipError_t afr_launch_glyph_l1_bwd_fused(const void* d1, int ldd, const void* h0, int ldh, const void* W1T, const int64_t* x,
                                         const int64_t* font, int B, int N1, int vocab, int n_fonts, float* slabs, hipStream_t s,
                                         const int* h0_rowmap = nullptr);
hipError_t afr_launch_glyph_l1_bwd(const float* slabs, int nslabs, long long slab_stride, const float* W1, int N1, int E,
                                   int vocab, int n_fonts, float* dw1, float* dtab_part, hipStream_t s);
hipError_t afr_launch_glyph_embed_bwd(int act_dtype, const void* d, const int64_t* x, const int64_t* font, int B,
                                      int E, int vocab, int n_fonts, float* slabs /*[blocks][(vocab+n_fonts)*E]*/,
                                      hipStream_t s);

// sheet front end (sheet.hip)
struct SheetDims { int L, Lmax, E, H, F, vocab; };
constexpr int AFR_SHEET_SAVE_PER_POS = 56;   // 32 + 4 + 4 + 16, see SheetDrop::save
struct SheetParams {   // device pointers into the flat f32 parameter buffer
    const float *pos, *emb, *w_in, *b_in, *w_o, *b_o, *ln_g, *ln_b, *w1, *b1;
};
struct SheetDrop {
    uint32_t key_e, key_a, key_f, thr_e, thr_a, thr_f; float sc_e, sc_a, sc_f; int training;
    // optional per-string save area written by a training forward and read by backward: [B][L*AFR_SHEET_SAVE_PER_POS]
    // words = attention output o [L][32], softmax row max [4][L], 1/row-sum [4][L], and the attention-dropout keep bits
    // [4][L][4] (row (h,i), key j: bit (j>>1)&31 of word (j&1)*2 + (j>>6)).  Spares backward the attention recompute
    // and both of its passes the per-element counter hash (3 quarter-rate integer multiplies per probability).
    float* save;
    // optional inspection output (afr_debug_sheet_gather): the rows the kernel's embedding gather fetched, e0 [B][L][32] f32,
    // before dropout and positional encoding (model.py:167)
    float* dbg_e0 = nullptr;
};
// offsets (in floats) of the 10 small tensors inside one partial-gradient slab == their flat-buffer offsets
struct SheetSlabOff { int pos, emb, win, bin, wo, bo, g, b, w1, b1, total; };
int afr_sheet_blocks(int B);
size_t afr_sheet_fwd_lds_bytes(const SheetDims& d);
size_t afr_sheet_bwd_lds_bytes(const SheetDims& d);
hipError_t afr_launch_sheet_fwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr,
                                const int64_t* x, int ldx, int B, void* z, float ln_eps, uint32_t* err_flag,
                                hipStream_t s);
hipError_t afr_launch_sheet_bwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr,
                                const int64_t* x, int ldx, int B, const void* dz, float ln_eps, float* slabs,
                                const SheetSlabOff& so, hipStream_t s);

// fused step of the small one-hidden-layer glyph nets (glyph_fused.hip)
struct Glyph1Args {
    const int64_t* x; const int64_t* font; const void* target; int tdtype;
    int B, E, N1, P, vocab, n_fonts;
    const float *emb, *femb, *b1, *b2;       // f32 masters
    const void *W1, *W2;                     // [N1][E], [P][N1] in the operand type (f32 masters / bf16 shadow)
    const void *W1T, *W2T;                   // bf16 mode: [E][N1], [N1][P]; f32 mode: the f32 masters again (gathered)
    float* slabs; long long slab_stride;     // slab b: this block's partial gradients, flat-buffer layout
    long long o_emb, o_font, o_w1, o_b1, o_w2, o_b2;
    float inv_n; float* loss_partial; unsigned* counter; float* loss_accum; uint32_t* err;
    int cs = 1;                              // column split: cs blocks share a row block, each owning P / cs output columns
};
int afr_glyph1_rows(int dtype);
int afr_glyph1_colsplit(int dtype, int B, int P);     // blocks per row block (1, 2 or 4) for a batch of B
int afr_glyph1_max_blocks(int dtype, int max_batch, int P);   // the most blocks any batch <= max_batch launches (slab / loss-partial count)
bool afr_glyph1_eligible(int E, int N1, int P, int vocab, int n_fonts);
size_t afr_glyph1_lds_bytes(int dtype, int E, int N1, int P, int table_rows);
hipError_t afr_launch_transpose_bf16(const float* W, bf16_t* WT, int N, int K, hipStream_t s);
hipError_t afr_launch_glyph1_step(int dtype, const Glyph1Args& a, hipStream_t s);

// per-pixel-token transformer (pixel.hip): the token-wise kernels of BASELINE configs[4], forward and backward
hipError_t afr_launch_pixel_ctx(int act_dtype, const float* emb, const float* femb, const int64_t* x, const int64_t* font, int B, int d,
                                int vocab, int n_fonts, void* ctx, uint32_t* err, hipStream_t s);
hipError_t afr_launch_pixel_add_ln(int act_dtype, const float* hin, float* h, const float* pos, const void* add, const float* g, const float* b, void* n,
                                   long long rows, int Tk, int d, float eps, hipStream_t s);
hipError_t afr_launch_pixel_attn(int act_dtype, const void* q, const void* kv, void* o, long long rows, int Tk, int d, int heads, int C, hipStream_t s);
hipError_t afr_launch_pixel_head(int act_dtype, const float* hin, float* h, const void* add, const float* g, const float* b, const float* w_out, const float* b_out,
                                 float* u, float* y, long long rows, int d, float eps, hipStream_t s);
int afr_pixel_bwd_blocks(long long rows);           // blocks (= partial slabs) of the head / LayerNorm backward kernels
int afr_pixel_attn_chunk(int Tk);                   // tokens per attention-backward block
hipError_t afr_launch_pixel_head_bwd(int act_dtype, const float* du, const float* hf, const float* g, const float* b, const float* w_out, float* dh,
                                     void* dhT, float* part /*[blocks][4][d]: dgamma, dbeta, dw_out, db_out*/, long long rows, int d, float eps, hipStream_t s);
hipError_t afr_launch_pixel_ln_bwd(int act_dtype, const void* dy, const float* hin, const float* g, float* dh, void* dhT, float* part /*[blocks][2][d]*/,
                                   long long rows, int d, float eps, hipStream_t s);
hipError_t afr_launch_pixel_attn_bwd(int act_dtype, const void* dO, const void* q, const void* kv, void* dq, float* dkv_part /*[chunks][B][2][2d]*/,
                                     int B, int Tk, int d, int C, hipStream_t s);
hipError_t afr_launch_pixel_ctx_bwd(const float* dctx, const int64_t* x, const int64_t* font, int B, int d, int vocab, int n_fonts, float* demb, float* dfont,
                                    hipStream_t s);
hipError_t afr_launch_pixel_accum(int act_dtype, float* acc, const void* src, long long n, int first, hipStream_t s);
hipError_t afr_launch_pixel_cast(int act_dtype, void* dst, const float* src, long long rows, int w, int ld_src, hipStream_t s);
