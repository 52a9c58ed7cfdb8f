// pixel.hip -- the token-wise (non-GEMM) kernels of the per-pixel-token transformer, BASELINE configs[4] as DESIGN.md 8 defines
// it (config.PixelConfig; oracle.pixel_forward): context gather, positional initialisation, residual-add + LayerNorm,
// cross-attention of a pixel token to the glyph's <= 2 context tokens, and the LayerNorm + Linear(d -> 1) + clamp head.
// The Linear layers run on the GEMM kernels of gemm.hip.  FORWARD ONLY so far (no counterpart in the reference: SURVEY 8 f5;
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
__global__ __launch_bounds__(256) void pixel_add_ln_kernel(float* __restrict__ h, const float* __restrict__ pos, const T* __restrict__ add,
                                                           const float* __restrict__ g, const float* __restrict__ b, T* __restrict__ n,
                                                           long long rows, int Tk, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int c0 = 8 * lane;
    const bool live = c0 < d;
    for (long long r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (live) {
            const float* src = pos ? pos + (size_t)(r % Tk) * d + c0 : h + (size_t)r * d + c0;
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
__global__ __launch_bounds__(256) void pixel_head_kernel(float* __restrict__ h, const T* __restrict__ add, const float* __restrict__ g,
                                                         const float* __restrict__ b, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                         float* __restrict__ u, float* __restrict__ y, long long rows, int d, float eps) {
    const int lane = threadIdx.x & 63, c0 = 8 * lane;
    const bool live = c0 < d;
    for (long long r = blockIdx.x * 4ll + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        float v[8], av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = av[j] = 0.f;
        if (live) { ld8v(h + (size_t)r * d + c0, v); ld8v(add + (size_t)r * d + c0, av); }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += av[j];
        if (live) {
            float* hd = h + (size_t)r * d + c0;
            *reinterpret_cast<float4*>(hd) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(hd + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        row_layernorm(v, g, b, c0, d, eps, live);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a = live ? fmaf(v[j], w_out[c0 + j], a) : a;
        a = wave_sum(a) + b_out[0];
        if (lane == 0) {
            if (u) u[r] = a;
            if (y) y[r] = fminf(fmaxf(a, 0.f), 1.f);
        }
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
hipError_t afr_launch_pixel_add_ln(int act_dtype, float* h, const float* pos, const void* add, const float* g, const float* b, void* n,
                                   long long rows, int Tk, int d, float eps, hipStream_t s) {
    if (d > 512 || (d & 7)) return hipErrorInvalidValue;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_add_ln_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, h, pos, (const bf16_t*)add, g, b, (bf16_t*)n, rows, Tk, d, eps);
    else hipLaunchKernelGGL(pixel_add_ln_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, h, pos, (const float*)add, g, b, (float*)n, rows, Tk, d, eps);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_attn(int act_dtype, const void* q, const void* kv, void* o, long long rows, int Tk, int d, int heads, int C, hipStream_t s) {
    if (d > 512 || d != heads * 64 || C < 1 || C > 2) return hipErrorInvalidValue;       // 8 lanes x 8 channels per head
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_attn_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)o, rows, Tk, d, C);
    else hipLaunchKernelGGL(pixel_attn_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, (const float*)q, (const float*)kv, (float*)o, rows, Tk, d, C);
    return hipGetLastError();
}
hipError_t afr_launch_pixel_head(int act_dtype, float* h, const void* add, const float* g, const float* b, const float* w_out, const float* b_out,
                                 float* u, float* y, long long rows, int d, float eps, hipStream_t s) {
    if (d > 512 || (d & 7)) return hipErrorInvalidValue;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(pixel_head_kernel<bf16_t>, dim3(pix_grid(rows)), dim3(256), 0, s, h, (const bf16_t*)add, g, b, w_out, b_out, u, y, rows, d, eps);
    else hipLaunchKernelGGL(pixel_head_kernel<float>, dim3(pix_grid(rows)), dim3(256), 0, s, h, (const float*)add, g, b, w_out, b_out, u, y, rows, d, eps);
    return hipGetLastError();
}
