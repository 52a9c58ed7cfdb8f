// pixel.hip -- the token-wise (non-GEMM) kernels of the per-pixel-token transformer, BASELINE configs[4] as DESIGN.md 8 defines
// it (config.PixelConfig; oracle.pixel_forward): context gather, positional initialisation, residual-add + LayerNorm,
// cross-attention of a pixel token to the glyph's <= 2 context tokens, and the LayerNorm + Linear(d -> 1) + clamp head.
// The Linear layers run on the GEMM kernels of gemm.hip.  Backward twins below (no counterpart in the reference: SURVEY 8 f5;
// the layer idioms are model.py:136,140-145,148,152-156).  One wave per token row, 8 channels per lane: d_model <= 512,
// a multiple of 64 x ... (checked by the launcher); the residual stream h stays float32, GEMM operands take the plan's
// activation dtype T.
#include "afr_common.h"
#include "../../include/afr.h"

namespace {
template <typename T> __device__ __forceinline__ T pcvt(float v);
template <> __device__ __forceinline__ float pcvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t pcvt<bf16_t>(float v) { return (bf16_t)v; }

// 8 consecutive channels as ONE 16-byte (bf16) / two 16-byte (f32) accesses
template <typename T> __device__ __forceinline__ void ld8v(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)w[j];
    } else {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
template <typename T> __device__ __forceinline__ void st8v(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8*>(p) = w;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}
// LayerNorm of one row held 8 channels per lane (biased variance, as nn.LayerNorm): returns the normalised, affine row
__device__ __forceinline__ void row_layernorm(float (&v)[8], const float* __restrict__ g, const float* __restrict__ b, int c0, int d, float eps, bool live) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += live ? v[j] : 0.f;
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] -= mu; q = live ? fmaf(v[j], v[j], q) : q; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    if (live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j] * rstd, g[c0 + j], b[c0 + j]);
    }
}
// the same with the affine parameters of the lane's 8 channels already in registers
__device__ __forceinline__ void row_layernorm_r(float (&v)[8], const float (&g)[8], const float (&b)[8], int d, float eps, bool live) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += live ? v[j] : 0.f;
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] -= mu; q = live ? fmaf(v[j], v[j], q) : q; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j] * rstd, g[j], b[j]);
}
}  // namespace

// ctx[b][0] = Emb[x_b], ctx[b][1] = Font[f_b]  (model.py:136,167 gather; index check as the glyph kernels)
template <typename T>
__global__ __launch_bounds__(256) void pixel_ctx_kernel(const float* __restrict__ emb, const float* __restrict__ femb, const int64_t* __restrict__ x,
                                                        const int64_t* __restrict__ font, int B, int d, int vocab, int n_fonts, T* __restrict__ ctx,
                                                        uint32_t* err) {
    const int C = n_fonts > 0 ? 2 : 1;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < (long long)B * C * d; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % d), c = (int)((i / d) % C), b = (int)(i / ((long long)d * C));
        long long xi = x[b], fi = (n_fonts > 0 && font) ? font[b] : 0;
        if (xi < 0 || xi >= vocab) { if (k == 0) atomicOr(err, 1u); xi = min(max(xi, 0ll), (long long)vocab - 1); }
        if (n_fonts > 0 && (fi < 0 || fi >= n_fonts)) { if (k == 0) atomicOr(err, 1u); fi = min(max(fi, 0ll), (long long)n_fonts - 1); }
        ctx[i] = pcvt<T>(c == 0 ? emb[xi * d + k] : femb[fi * d + k]);
    }
}
// h = (first ? pos[t] : h) + (add ? add : 0);  n = LayerNorm(h) * g + b   -- rows = B * Tk tokens, one wave per row
template <typename T>
__global__ __launch_bounds__(256) void pixel_add_ln_kernel(const float* __restrict__ hin, float* __restrict__ h, const float* __restrict__ pos,
                                                           const T* __restrict__ add, const float* __restrict__ g, const float* __restrict__ b,
                                                           T* __restrict__ n, long long rows, int Tk, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int c0 = 8 * lane;
    const bool live = c0 < d;
    // (g, b stay per-row loads (cache hits): held in 16 registers across the row loop this kernel ran 7 % slower)
    for (long long r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (live) {
            const float* src = pos ? pos + (size_t)(r % Tk) * d + c0 : hin + (size_t)r * d + c0;
            const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
            if (add) {
                float av[8];
                ld8v(add + (size_t)r * d + c0, av);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += av[j];
            }
            float* hd = h + (size_t)r * d + c0;
            *reinterpret_cast<float4*>(hd) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(hd + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (n) {
            row_layernorm(v, g, b, c0, d, eps, live);
            if (live) st8v(n + (size_t)r * d + c0, v);
        }
    }
}
// o[r][head] = softmax_c(q_head . k[b][c][head] * sqrt(1/D)) . v[b][c][head], c < C <= 2 context tokens (nn.MultiheadAttention,
// model.py:144: q scaled by sqrt(1/D), softmax over the keys); kv [B][C][2 d] = [k | v]; head_dim D = 64: 8 lanes per head
template <typename T>
__global__ __launch_bounds__(256) void pixel_attn_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ o, long long rows,
                                                         int Tk, int d, int C) {
    const int lane = threadIdx.x & 63, c0 = 8 * lane;
    const bool live = c0 < d;
    for (long long r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const long long b = r / Tk;
        float qv[8], s[2] = {0.f, 0.f}, vv[2][8], kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = 0.f;
        if (live) ld8v(q + (size_t)r * d + c0, qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= 0.125f;                                                     // sqrt(1/64)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < C) {
                const T* kr = kv + ((size_t)b * C + c) * 2 * d;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { kk[j] = 0.f; vv[c][j] = 0.f; }
                if (live) { ld8v(kr + c0, kk); ld8v(kr + d + c0, vv[c]); }
#pragma unroll
                for (int j = 0; j < 8; ++j) a = fmaf(qv[j], kk[j], a);
                a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);       // the head's 8 lanes
                s[c] = a;
            }
        }
        float p0 = 1.f, p1 = 0.f;
        if (C == 2) {
            const float m = fmaxf(s[0], s[1]);
            const float e0 = __expf(s[0] - m), e1 = __expf(s[1] - m), inv = 1.f / (e0 + e1);
            p0 = e0 * inv; p1 = e1 * inv;
        }
        if (live) {
            float ov[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = C == 2 ? fmaf(p1, vv[1][j], p0 * vv[0][j]) : vv[0][j];
            st8v(o + (size_t)r * d + c0, ov);
        }
    }
}
// h += add;  u = LayerNorm_f(h) . w_out + b_out;  y = clamp(u, 0, 1)      (model.py:152-156 idiom)
template <typename T>
__global__ __launch_bounds__(256) void pixel_head_kernel(const float* __restrict__ hin, float* __restrict__ h, const T* __restrict__ add, const float* __restrict__ g,
                                                         const float* __restrict__ b, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                         float* __restrict__ u, float* __restrict__ y, long long rows, int d, float eps) {
    const int lane = threadIdx.x & 63, c0 = 8 * lane;
    const bool live = c0 < d;
    float gv[8], bv[8], wv[8];                            // per-lane constants once; the next row is requested before this one is worked on
#pragma unroll
    for (int j = 0; j < 8; ++j) { gv[j] = live ? g[c0 + j] : 0.f; bv[j] = live ? b[c0 + j] : 0.f; wv[j] = live ? w_out[c0 + j] : 0.f; }
    const float bo = b_out[0];
    const long long stride = (long long)gridDim.x * 4;
    long long r = blockIdx.x * 4ll + (threadIdx.x >> 6);
    float v[8], av[8], vn[8], avn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = av[j] = vn[j] = avn[j] = 0.f;
    if (r < rows && live) { ld8v(hin + (size_t)r * d + c0, v); ld8v(add + (size_t)r * d + c0, av); }
    for (; r < rows; r += stride) {
        const long long rn = r + stride;
        if (rn < rows && live) { ld8v(hin + (size_t)rn * d + c0, vn); ld8v(add + (size_t)rn * d + c0, avn); }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += av[j];
        if (live) {
            float* hd = h + (size_t)r * d + c0;
            *reinterpret_cast<float4*>(hd) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(hd + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        row_layernorm_r(v, gv, bv, d, eps, live);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a = fmaf(v[j], wv[j], a);
        a = wave_sum(a) + bo;
        if (lane == 0) {
            if (u) u[r] = a;
            if (y) y[r] = fminf(fmaxf(a, 0.f), 1.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = vn[j]; av[j] = avn[j]; }
    }
}

static inline int pix_grid(long long rows) { long long g = (rows + 3) / 4; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
#define PIX_DISPATCH(KERNEL, GRID, ...)                                                                        \
    do {                                                                                                       \
        if (act_dtype == AFR_BF16) hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<float>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__);                    \
    } while (0)
hipError_t afr_launch_pixel_ctx(int act_dtype, const float* emb, const float* femb, const int64_t* x, const int64_t* font, int B, int d,
                                int vocab, int n_fonts, void* ctx, uint32_t* err, hipStream_t s) {
    const long long n = (long long)B * (n_fonts > 0 ? 2 : 1) * d;
    const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_ctx_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, emb, femb, x, font, B, d, vocab, n_fonts, (bf16_t*)ctx, err);
    else hipLaunchKernelGGL(pixel_ctx_kernel<float>, dim3(grid), dim3(256), 0, s, emb, femb, x, font, B, d, vocab, n_fonts, (float*)ctx, err);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_add_ln(int act_dtype, const float* hin, float* h, const float* pos, const void* add, const float* g, const float* b, void* n,
                                   long long rows, int Tk, int d, float eps, hipStream_t s) {
    if (d > 512 || (d & 7)) return hipErrorInvalidValue;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_add_ln_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, hin, h, pos, (const bf16_t*)add, g, b, (bf16_t*)n, rows, Tk, d, eps);
    else hipLaunchKernelGGL(pixel_add_ln_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, hin, h, pos, (const float*)add, g, b, (float*)n, rows, Tk, d, eps);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_attn(int act_dtype, const void* q, const void* kv, void* o, long long rows, int Tk, int d, int heads, int C, hipStream_t s) {
    if (d > 512 || d != heads * 64 || C < 1 || C > 2) return hipErrorInvalidValue;       // 8 lanes x 8 channels per head
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_attn_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)o, rows, Tk, d, C);
    else hipLaunchKernelGGL(pixel_attn_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, (const float*)q, (const float*)kv, (float*)o, rows, Tk, d, C);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_head(int act_dtype, const float* hin, float* h, const void* add, const float* g, const float* b, const float* w_out, const float* b_out,
                                 float* u, float* y, long long rows, int d, float eps, hipStream_t s) {
    if (d > 512 || (d & 7)) return hipErrorInvalidValue;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_head_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, hin, h, (const bf16_t*)add, g, b, w_out, b_out, u, y, rows, d, eps);
    else hipLaunchKernelGGL(pixel_head_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, hin, h, (const float*)add, g, b, w_out, b_out, u, y, rows, d, eps);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------- backward (round 3)
// Every kernel below walks token rows one wave per row like its forward twin and leaves its parameter-gradient partial sums
// as ONE slab per block ([nblk][...], block order), which the grouped reduce adds in fixed order: bitwise reproducible.
namespace {
constexpr int PIX_BWD_BLOCKS = 512;        // blocks of the backward token kernels = slabs per LayerNorm / head gradient
constexpr int PIX_BWD_WAVES = 16;          // waves per block: 2 blocks x 16 waves fill a CU's wave slots (these kernels live on loads in flight:
                                           // with 4-wave blocks pixel_ln_bwd ran at 3.8 TB/s and pixel_head_bwd at 1.6 TB/s)
constexpr int PIX_BWD_THREADS = 64 * PIX_BWD_WAVES;
// sum the waves' per-lane partials (8 channels each, K arrays) through LDS in wave order and store the total: slab[k][d]
// (sh: (PIX_BWD_WAVES - 1) * 512 floats; lane-major rows of 8 floats + 1 pad keep the 64 lanes on distinct banks)
constexpr int PIX_SH_FLOATS = (PIX_BWD_WAVES - 1) * 64 * 9;
template <int K>
__device__ __forceinline__ void block_store_partials(float (&acc)[K][8], float* slab, int d, int c0, bool live, float* sh) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sh[((wave - 1) * 64 + lane) * 9 + j] = acc[k][j];
        }
        __syncthreads();
        if (wave == 0 && live) {
            float out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = acc[k][j];
            for (int w = 0; w < PIX_BWD_WAVES - 1; ++w)
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] += sh[(w * 64 + lane) * 9 + j];
            *reinterpret_cast<float4*>(slab + (size_t)k * d + c0) = make_float4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<float4*>(slab + (size_t)k * d + c0 + 4) = make_float4(out[4], out[5], out[6], out[7]);
        }
        __syncthreads();
    }
}
// LayerNorm backward of one row: x (pre-norm input), dy (gradient of the affine output) -> dx; accumulates dgamma, dbeta
__device__ __forceinline__ void row_ln_bwd(const float (&x)[8], const float (&dy)[8], const float (&g)[8], int d, float eps, bool live,
                                           float (&dx)[8], float (&dg)[8], float (&db)[8]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += live ? x[j] : 0.f;
    const float mu = wave_sum(s) / (float)d;
    float xc[8], q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { xc[j] = x[j] - mu; q = live ? fmaf(xc[j], xc[j], q) : q; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float gg[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xh = xc[j] * rstd;
        xc[j] = xh;
        gg[j] = live ? dy[j] * g[j] : 0.f;
        m1 += gg[j]; m2 = fmaf(gg[j], xh, m2);
        if (live) { dg[j] = fmaf(dy[j], xh, dg[j]); db[j] += dy[j]; }
    }
    m1 = wave_sum(m1) / (float)d; m2 = wave_sum(m2) / (float)d;
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[j] = (gg[j] - m1 - xc[j] * m2) * rstd;
}
}  // namespace
// head backward: du [rows] -> dh = LN_f-backward(du * w_out); partials [nblk][4][d]: dgamma_f, dbeta_f, dw_out, (db_out in [3][0])
template <typename T>
__global__ __launch_bounds__(PIX_BWD_THREADS) void pixel_head_bwd_kernel(const float* __restrict__ du, const float* __restrict__ hf, const float* __restrict__ g,
                                                             const float* __restrict__ bta, const float* __restrict__ w_out, float* __restrict__ dh,
                                                             T* __restrict__ dhT, float* __restrict__ part, long long rows, int d, float eps) {
    __shared__ float sh[PIX_SH_FLOATS];
    const int lane = threadIdx.x & 63, c0 = 8 * lane;
    const bool live = c0 < d;
    float acc[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    // per-lane constants (8 channels) once; the next row's loads are requested before this row's arithmetic (a wave walks
    // only rows / (blocks * waves) rows: without the prefetch every row pays a full memory latency)
    float gv[8], bv[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gv[j] = live ? g[c0 + j] : 0.f; bv[j] = live ? bta[c0 + j] : 0.f; wv[j] = live ? w_out[c0 + j] : 0.f; }
    const long long stride = (long long)gridDim.x * PIX_BWD_WAVES;
    long long r = (long long)blockIdx.x * PIX_BWD_WAVES + (threadIdx.x >> 6);
    float x[8], xn[8], dur = 0.f, durn = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = xn[j] = 0.f;
    if (r < rows) { if (live) ld8v(hf + (size_t)r * d + c0, x); dur = du[r]; }
    for (; r < rows; r += stride) {
        const long long rn = r + stride;
        if (rn < rows) { if (live) ld8v(hf + (size_t)rn * d + c0, xn); durn = du[rn]; }
        float dy[8], dx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dy[j] = dur * wv[j];
        // nf = LN_f(hf) for dw_out: recomputed (xhat * g + b)
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[j];
        row_layernorm_r(xs, gv, bv, d, eps, live);
        row_ln_bwd(x, dy, gv, d, eps, live, dx, acc[0], acc[1]);
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[2][j] = fmaf(dur, xs[j], acc[2][j]);
            if (lane == 0) acc[3][0] += dur;
            float* o = dh + (size_t)r * d + c0;
            *reinterpret_cast<float4*>(o) = make_float4(dx[0], dx[1], dx[2], dx[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(dx[4], dx[5], dx[6], dx[7]);
            if (dhT) st8v(dhT + (size_t)r * d + c0, dx);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = xn[j];
        dur = durn;
    }
    block_store_partials<4>(acc, part + (size_t)blockIdx.x * 4 * d, d, c0, live, sh);
}
// LayerNorm backward with the residual: dh <- dh + LN-backward(dy; x = hin);  partials [nblk][2][d]: dgamma, dbeta
template <typename T>
__global__ __launch_bounds__(PIX_BWD_THREADS) void pixel_ln_bwd_kernel(const T* __restrict__ dyT, const float* __restrict__ hin, const float* __restrict__ g,
                                                           float* __restrict__ dh, T* __restrict__ dhT, float* __restrict__ part, long long rows, int d,
                                                           float eps) {
    __shared__ float sh[PIX_SH_FLOATS];
    const int lane = threadIdx.x & 63, c0 = 8 * lane;
    const bool live = c0 < d;
    float acc[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    float gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gv[j] = live ? g[c0 + j] : 0.f;
    const long long stride = (long long)gridDim.x * PIX_BWD_WAVES;
    long long r = (long long)blockIdx.x * PIX_BWD_WAVES + (threadIdx.x >> 6);
    float x[8], dy[8], res[8], xn[8], dyn[8], resn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = dy[j] = res[j] = xn[j] = dyn[j] = resn[j] = 0.f;
    if (r < rows && live) { ld8v(hin + (size_t)r * d + c0, x); ld8v(dyT + (size_t)r * d + c0, dy); ld8v(dh + (size_t)r * d + c0, res); }
    for (; r < rows; r += stride) {
        const long long rn = r + stride;            // (a row is read and written by this wave only: the prefetch of row rn never races a store)
        if (rn < rows && live) { ld8v(hin + (size_t)rn * d + c0, xn); ld8v(dyT + (size_t)rn * d + c0, dyn); ld8v(dh + (size_t)rn * d + c0, resn); }
        float dx[8];
        row_ln_bwd(x, dy, gv, d, eps, live, dx, acc[0], acc[1]);
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] += dx[j];
            st8v(dh + (size_t)r * d + c0, res);
            if (dhT) st8v(dhT + (size_t)r * d + c0, res);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = xn[j]; dy[j] = dyn[j]; res[j] = resn[j]; }
    }
    block_store_partials<2>(acc, part + (size_t)blockIdx.x * 2 * d, d, c0, live, sh);
}
// cross-attention backward (C <= 2 keys): do, q, kv -> dq [rows][d]; dk | dv of the sample's context tokens summed over a
// chunk of its tokens: dkv_part [chunks][B][2][2 d] -- one slab per chunk, so ONE slab reduce over the chunks finishes every
// sample's sums (block = (sample, chunk); the waves' sums are added in wave order)
template <typename T>
__global__ __launch_bounds__(PIX_BWD_THREADS) void pixel_attn_bwd_kernel(const T* __restrict__ dO, const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ dq,
                                                             float* __restrict__ dkv_part, int Tk, int chunk, int d, int C) {
    __shared__ float sh[PIX_SH_FLOATS];
    const int lane = threadIdx.x & 63, c0 = 8 * lane, wave = threadIdx.x >> 6;
    const bool live = c0 < d;
    const int chunks = (Tk + chunk - 1) / chunk;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    float kk[2][8], vv[2][8], acc[4][8];               // acc: dk0, dv0, dk1, dv1
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[c][j] = vv[c][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (int c = 0; c < C; ++c)
        if (live) { const T* kr = kv + ((size_t)b * C + c) * 2 * d; ld8v(kr + c0, kk[c]); ld8v(kr + d + c0, vv[c]); }
    const int t1 = min(Tk, (ch + 1) * chunk);
    for (int t = ch * chunk + wave; t < t1; t += PIX_BWD_WAVES) {
        const size_t r = (size_t)b * Tk + t;
        float qv[8], dov[8], s[2] = {0.f, 0.f}, dp[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = dov[j] = 0.f;
        if (live) { ld8v(q + r * d + c0, qv); ld8v(dO + r * d + c0, dov); }
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= 0.125f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float a = 0.f, e = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a = fmaf(qv[j], kk[c][j], a); e = fmaf(dov[j], vv[c][j], e); }
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
            e += __shfl_xor(e, 1, 64); e += __shfl_xor(e, 2, 64); e += __shfl_xor(e, 4, 64);
            s[c] = a; dp[c] = e;
        }
        float p0 = 1.f, p1 = 0.f;
        if (C == 2) {
            const float m = fmaxf(s[0], s[1]);
            const float e0 = __expf(s[0] - m), e1 = __expf(s[1] - m), inv = 1.f / (e0 + e1);
            p0 = e0 * inv; p1 = e1 * inv;
        }
        const float dot = p0 * dp[0] + p1 * dp[1];
        const float ds0 = p0 * (dp[0] - dot), ds1 = p1 * (dp[1] - dot);       // softmax backward (zero when C == 1)
        float dqv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dqv[j] = (ds0 * kk[0][j] + ds1 * kk[1][j]) * 0.125f;
            acc[0][j] = fmaf(ds0, qv[j], acc[0][j]); acc[1][j] = fmaf(p0, dov[j], acc[1][j]);
            acc[2][j] = fmaf(ds1, qv[j], acc[2][j]); acc[3][j] = fmaf(p1, dov[j], acc[3][j]);
        }
        if (live) st8v(dq + r * d + c0, dqv);
    }
    // partial layout [C][2 d] = [dk_c | dv_c]: rows (k-arrays) 0..3 = dk0, dv0, dk1, dv1 are consecutive d-vectors
    block_store_partials<4>(acc, dkv_part + ((size_t)ch * (gridDim.x / chunks) + b) * 4 * d, d, c0, live, sh);
}
// dEmb[v] = sum over glyphs b with x_b == v of dctx[b][0];  dFont[f] likewise with dctx[b][1]   (embedding_dense_backward)
__global__ __launch_bounds__(256) void pixel_ctx_bwd_kernel(const float* __restrict__ dctx, const int64_t* __restrict__ x, const int64_t* __restrict__ font,
                                                            int B, int d, int vocab, int n_fonts, float* __restrict__ demb, float* __restrict__ dfont) {
    const int row = blockIdx.x, C = n_fonts > 0 ? 2 : 1;
    const bool is_font = row >= vocab;
    const int target = is_font ? row - vocab : row;
    for (int k = threadIdx.x; k < d; k += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) {
            long long id = is_font ? font[b] : x[b];
            id = min(max(id, 0ll), (long long)(is_font ? n_fonts : vocab) - 1);
            if ((int)id == target) a += dctx[((size_t)b * C + (is_font ? 1 : 0)) * d + k];
        }
        (is_font ? dfont : demb)[(size_t)target * d + k] = a;
    }
}
// acc (f32) += src (T)   /  acc = src when first
template <typename T>
__global__ __launch_bounds__(256) void pixel_accum_kernel(float* __restrict__ acc, const T* __restrict__ src, long long n, int first) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc[i] = (first ? 0.f : acc[i]) + (float)src[i];
}
// dst (T) [rows][w] = src (f32) [rows][ld_src], the first w columns of every row
template <typename T>
__global__ __launch_bounds__(256) void pixel_cast_kernel(T* __restrict__ dst, const float* __restrict__ src, long long rows, int w, int ld_src) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < rows * w; i += (long long)gridDim.x * 256) {
        const long long r = i / w;
        dst[i] = pcvt<T>(src[r * ld_src + (i - r * w)]);
    }
}
int afr_pixel_bwd_blocks(long long rows) { long long g = (rows + PIX_BWD_WAVES - 1) / PIX_BWD_WAVES; return (int)(g < 1 ? 1 : (g > PIX_BWD_BLOCKS ? PIX_BWD_BLOCKS : g)); }
hipError_t afr_launch_pixel_head_bwd(int act_dtype, const float* du, const float* hf, const float* g, const float* b, const float* w_out, float* dh,
                                     void* dhT, float* part, long long rows, int d, float eps, hipStream_t s) {
    const int grid = afr_pixel_bwd_blocks(rows);
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_head_bwd_kernel<bf16_t>, dim3(grid), dim3(PIX_BWD_THREADS), 0, s, du, hf, g, b, w_out, dh, (bf16_t*)dhT, part, rows, d, eps);
    else hipLaunchKernelGGL(pixel_head_bwd_kernel<float>, dim3(grid), dim3(PIX_BWD_THREADS), 0, s, du, hf, g, b, w_out, dh, (float*)nullptr, part, rows, d, eps);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_ln_bwd(int act_dtype, const void* dy, const float* hin, const float* g, float* dh, void* dhT, float* part, long long rows,
                                   int d, float eps, hipStream_t s) {
    const int grid = afr_pixel_bwd_blocks(rows);
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_ln_bwd_kernel<bf16_t>, dim3(grid), dim3(PIX_BWD_THREADS), 0, s, (const bf16_t*)dy, hin, g, dh, (bf16_t*)dhT, part, rows, d, eps);
    else hipLaunchKernelGGL(pixel_ln_bwd_kernel<float>, dim3(grid), dim3(PIX_BWD_THREADS), 0, s, (const float*)dy, hin, g, dh, (float*)nullptr, part, rows, d, eps);
    return hipGetLastError();
}
int afr_pixel_attn_chunk(int Tk) { return Tk <= 256 ? Tk : 256; }
hipError_t afr_launch_pixel_attn_bwd(int act_dtype, const void* dO, const void* q, const void* kv, void* dq, float* dkv_part, int B, int Tk, int d, int C,
                                     hipStream_t s) {
    const int chunk = afr_pixel_attn_chunk(Tk), chunks = (Tk + chunk - 1) / chunk;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_attn_bwd_kernel<bf16_t>, dim3(B * chunks), dim3(PIX_BWD_THREADS), 0, s, (const bf16_t*)dO, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)dq, dkv_part, Tk, chunk, d, C);
    else hipLaunchKernelGGL(pixel_attn_bwd_kernel<float>, dim3(B * chunks), dim3(PIX_BWD_THREADS), 0, s, (const float*)dO, (const float*)q, (const float*)kv, (float*)dq, dkv_part, Tk, chunk, d, C);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_ctx_bwd(const float* dctx, const int64_t* x, const int64_t* font, int B, int d, int vocab, int n_fonts, float* demb, float* dfont,
                                    hipStream_t s) {
    hipLaunchKernelGGL(pixel_ctx_bwd_kernel, dim3(vocab + (n_fonts > 0 ? n_fonts : 0)), dim3(256), 0, s, dctx, x, font, B, d, vocab, n_fonts, demb, dfont);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_accum(int act_dtype, float* acc, const void* src, long long n, int first, hipStream_t s) {
    const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_accum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, acc, (const bf16_t*)src, n, first);
    else hipLaunchKernelGGL(pixel_accum_kernel<float>, dim3(grid), dim3(256), 0, s, acc, (const float*)src, n, first);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_cast(int act_dtype, void* dst, const float* src, long long rows, int w, int ld_src, hipStream_t s) {
    const long long n = rows * w;
    const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)dst, src, rows, w, ld_src);
    else hipLaunchKernelGGL(pixel_cast_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)dst, src, rows, w, ld_src);
    return hipGetLastError();
}
