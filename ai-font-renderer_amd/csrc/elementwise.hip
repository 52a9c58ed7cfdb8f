// elementwise.hip -- the HBM-bound kernels of the hot path: loss/grad reduction, AdamW, slab reduction,
// the glyph embedding gather and its deterministic scatter-add (bias gradients are fused into the dW GEMM).
// All are 16-byte-per-lane streaming kernels with grid-stride loops; reductions are shuffle -> LDS ->
// per-block partial -> fixed-order finish, so every result is bitwise reproducible run to run.
#include <algorithm>
#include "afr_common.h"
#include "../../include/afr.h"

int afr_glyph_k0(int E, int vocab, int n_fonts);
static inline int grid_for(long long work_items, int block, int max_blocks = 2048) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ------------------------------------------------------------------------------------- reduce
// dst[i] = (accumulate ? dst[i] : 0) + scale * sum_s slabs[s*stride + i]   (fixed s order)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* __restrict__ dst, const float* __restrict__ slabs,
                                                           int nslabs, long long stride, long long n, float scale,
                                                           int accumulate) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < nslabs; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + s * stride + 4 * i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
        float4* d = reinterpret_cast<float4*>(dst) + i;
        if (accumulate) { float4 o = *d; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *d = a;
    }
    // tail (n not a multiple of 4)
    for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a = 0.f;
        for (int s = 0; s < nslabs; ++s) a += slabs[s * stride + i];
        a *= scale;
        if (accumulate) a += dst[i];
        dst[i] = a;
    }
}
hipError_t afr_launch_reduce(float* dst, const float* slabs, int nslabs, long long slab_stride, long long n,
                             float scale, int accumulate, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, dst, slabs, nslabs,
                       slab_stride, n, scale, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ grouped reduce
// (Optionally fused with AdamW: single-GPU steps update p/m/v right here and never materialise those gradients.)
// One launch sums every slab-produced gradient of a backward pass (split-K dW slabs, fused bias partials, embedding
// partials) into the flat gradient buffer, each in fixed slab order.  Block -> segment by a scan of <= 24 entries.
// sum of slabs [s0, s1) of one float4 column, 8 independent 16-byte loads in flight, fixed order
__device__ __forceinline__ float4 slab_sum(const float* src, long long stride, int s0, int s1) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long long)(s + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
#pragma unroll 4
    for (; s < s1; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long long)s * stride);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    return a;
}
__device__ __forceinline__ void reduce_finish(const RTable& t, const RSeg& sg, long long i, float4 a, float4 pp, float4 mm, float4 vv) {
    if (t.adam) {
        const long long off = (sg.dst - t.gbase) + 4 * i;
        adamw_elem(pp.x, mm.x, vv.x, a.x, t.ad_decay, t.ad_b1, t.ad_b2, t.ad_eps, t.ad_step, t.ad_rsqrt_bc2);
        adamw_elem(pp.y, mm.y, vv.y, a.y, t.ad_decay, t.ad_b1, t.ad_b2, t.ad_eps, t.ad_step, t.ad_rsqrt_bc2);
        adamw_elem(pp.z, mm.z, vv.z, a.z, t.ad_decay, t.ad_b1, t.ad_b2, t.ad_eps, t.ad_step, t.ad_rsqrt_bc2);
        adamw_elem(pp.w, mm.w, vv.w, a.w, t.ad_decay, t.ad_b1, t.ad_b2, t.ad_eps, t.ad_step, t.ad_rsqrt_bc2);
        *reinterpret_cast<float4*>(t.P + off) = pp;
        *reinterpret_cast<float4*>(t.M + off) = mm;
        *reinterpret_cast<float4*>(t.V + off) = vv;
        if (t.shadow) {
            bf16x4 o = {(bf16_t)pp.x, (bf16_t)pp.y, (bf16_t)pp.z, (bf16_t)pp.w};
            *reinterpret_cast<bf16x4*>(t.shadow + off) = o;
        }
        if (sg.shT) {                      // 4 consecutive k of one row n (tK is a multiple of 4)
            const int n = (int)((4 * i) / sg.tK), k = (int)(4 * i - (long long)n * sg.tK);
            sg.shT[(size_t)k * sg.tN + n] = (bf16_t)pp.x;
            sg.shT[(size_t)(k + 1) * sg.tN + n] = (bf16_t)pp.y;
            sg.shT[(size_t)(k + 2) * sg.tN + n] = (bf16_t)pp.z;
            sg.shT[(size_t)(k + 3) * sg.tN + n] = (bf16_t)pp.w;
        }
    } else {
        reinterpret_cast<float4*>(sg.dst)[i] = a;
    }
}
__global__ __launch_bounds__(256) void reduce_group_kernel(RTable t) {
    int si = 0;
    for (int k = 1; k < t.nseg; ++k) if ((int)blockIdx.x >= t.seg[k].blk0) si = k;
    const RSeg sg = t.seg[si];
    float4 pp = make_float4(0.f, 0.f, 0.f, 0.f), mm = pp, vv = pp;
    if (sg.deep) {
        // many slabs, few columns (the per-block partials of the sheet backward): a block owns 64 float4 columns and
        // each of its 4 waves sums a quarter of the slabs; the quarters meet in LDS and are added in wave order.
        __shared__ float4 part[3][64];
        const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const int per = (sg.nslabs + 3) >> 2;
        for (long long i0 = (long long)(blockIdx.x - sg.blk0) * 64; i0 < sg.n4; i0 += (long long)sg.nblk * 64) {
            const long long i = i0 + col;
            const bool live = i < sg.n4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                if (grp == 0 && t.adam) {
                    const long long off = (sg.dst - t.gbase) + 4 * i;
                    pp = *reinterpret_cast<float4*>(t.P + off); mm = *reinterpret_cast<float4*>(t.M + off); vv = *reinterpret_cast<float4*>(t.V + off);
                }
                const int s0 = grp * per, s1 = min(sg.nslabs, s0 + per);
                a = slab_sum(sg.src + 4 * i, sg.stride, s0, s1);
            }
            if (grp) part[grp - 1][col] = a;
            __syncthreads();
            if (grp == 0 && live) {
#pragma unroll
                for (int g = 0; g < 3; ++g) { const float4 v = part[g][col]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
                reduce_finish(t, sg, i, a, pp, mm, vv);
            }
            __syncthreads();
        }
        return;
    }
    for (long long i = (long long)(blockIdx.x - sg.blk0) * 256 + threadIdx.x; i < sg.n4; i += (long long)sg.nblk * 256) {
        if (t.adam) {   // issued ahead of the slab loads so that everything this element needs is in flight at once
            const long long off = (sg.dst - t.gbase) + 4 * i;
            pp = *reinterpret_cast<float4*>(t.P + off); mm = *reinterpret_cast<float4*>(t.M + off); vv = *reinterpret_cast<float4*>(t.V + off);
        }
        const float4 a = slab_sum(sg.src + 4 * i, sg.stride, 0, sg.nslabs);
        reduce_finish(t, sg, i, a, pp, mm, vv);
    }
}
void afr_rtable_add(RTable& t, float* dst, const float* src, int nslabs, long long stride, long long n) {
    if (n <= 0) return;
    if (t.nseg >= AFR_RT_MAXSEG) { t.overflow = 1; return; }     // the launch refuses an overflowed table: never a silent drop
    RSeg& sg = t.seg[t.nseg];
    sg.dst = dst; sg.src = src; sg.stride = stride; sg.n4 = n / 4; sg.nslabs = nslabs; sg.blk0 = t.nblocks;
    sg.shT = nullptr; sg.tN = sg.tK = 0;
    sg.deep = nslabs >= 32;
    const int cols = sg.deep ? 64 : 256;
    long long nb = (sg.n4 + cols - 1) / cols;       // one float4 per thread where possible: measured 2x faster than
    if (nb > 1024) nb = 1024;                       // 4 per thread (the kernel lives on memory-level parallelism)
    sg.nblk = (int)nb;
    t.nblocks += (int)nb;
    t.nseg++;
}
hipError_t afr_launch_reduce_group(const RTable& t, hipStream_t s) {
    if (t.overflow) return hipErrorInvalidValue;
    if (t.nseg == 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_group_kernel, dim3(t.nblocks), dim3(256), 0, s, t);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------- AdamW
// torch.optim.AdamW single-tensor update (reference model.py:273,310), one pass over p,g,m,v:
//   p *= 1 - lr*wd;  m += (g-m)*(1-b1);  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Also refreshes the bf16 shadow copy the bf16 GEMMs read.  28 (+2) bytes per element of HBM traffic.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ shadow, long long n4, float lr, float b1,
                                                    float b2, float eps, float wd, float step_size, float rsqrt_bc2,
                                                    float gscale) {
    const float decay = 1.f - lr * wd;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) adamw_elem(pa[k], ma[k], va[k], ga[k] * gscale, decay, b1, b2, eps, step_size, rsqrt_bc2);
        reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
        if (shadow) {
            bf16x4 o = {(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3]};
            reinterpret_cast<bf16x4*>(shadow)[i] = o;
        }
    }
}
hipError_t afr_launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long long n, float lr,
                            float beta1, float beta2, float eps, float wd, float bc1, float bc2, float grad_scale,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 3) return hipErrorInvalidValue;   // flat buffers are padded to multiples of 64
    const float step_size = lr / bc1;
    const float rsqrt_bc2 = (float)(1.0 / sqrt((double)bc2));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, s, p, g, m, v, shadow, n / 4, lr,
                       beta1, beta2, eps, wd, step_size, rsqrt_bc2, grad_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- f32 -> bf16
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                          long long n) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[i];
        bf16x4 o = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w};
        reinterpret_cast<bf16x4*>(dst)[i] = o;
    }
    for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = (bf16_t)src[i];
}
hipError_t afr_launch_f32_to_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------- f32 -> fp8
// dst = e4m3(src * inv_scale), OCP e4m3fn (gfx950's native fp8: bias 7, max 448, no infinities), round to nearest even,
// saturating at +-448.  The per-tensor scale is the caller's (max |src| / 448 is the usual choice).
__global__ __launch_bounds__(256) void f32_to_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, long long n,
                                                         float inv_scale) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[i];
        const float x0 = fminf(fmaxf(a.x * inv_scale, -448.f), 448.f), x1 = fminf(fmaxf(a.y * inv_scale, -448.f), 448.f);
        const float x2 = fminf(fmaxf(a.z * inv_scale, -448.f), 448.f), x3 = fminf(fmaxf(a.w * inv_scale, -448.f), 448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
        reinterpret_cast<int*>(dst)[i] = w;
    }
    for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = fminf(fmaxf(src[i] * inv_scale, -448.f), 448.f);
        dst[i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xFF);
    }
}
hipError_t afr_launch_f32_to_fp8(const float* src, unsigned char* dst, long long n, float inv_scale, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(f32_to_fp8_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, src, dst, n, inv_scale);
    return hipGetLastError();
}

// -------------------------------------------------------------------- clamp output (eval path)
// y = clamp(u, 0, 1) as float32: the model's output activation (reference model.py:156,202)
template <typename T>
__global__ __launch_bounds__(256) void clamp_out_kernel(const T* __restrict__ u, float* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = fminf(fmaxf((float)u[i], 0.f), 1.f);
}
hipError_t afr_launch_clamp_out(int act_dtype, const void* u, float* y, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (act_dtype == AFR_BF16)
        hipLaunchKernelGGL(clamp_out_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const bf16_t*)u, y, n);
    else
        hipLaunchKernelGGL(clamp_out_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const float*)u, y, n);
    return hipGetLastError();
}

// du = dy * [0 <= u <= 1], in place over u: torch.clamp's backward (reference model.py:156) for a caller-side loss
template <typename T>
__global__ __launch_bounds__(256) void clamp_bwd_kernel(T* __restrict__ u, const float* __restrict__ dy, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float uv = (float)u[i];
        u[i] = (T)((uv >= 0.f && uv <= 1.f) ? dy[i] : 0.f);
    }
}
hipError_t afr_launch_clamp_bwd(int act_dtype, void* u, const float* dy, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (act_dtype == AFR_BF16) hipLaunchKernelGGL(clamp_bwd_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, (bf16_t*)u, dy, n);
    else hipLaunchKernelGGL(clamp_bwd_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, (float*)u, dy, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- MSE loss + gradient
// loss = sum((clamp(u,0,1) - t)^2) / mean_elems ;  du = 2 (y - t) / mean_elems * [0 <= u <= 1]
// (reference model.py:156,268-270 and the first step of loss.backward(), model.py:309).
// 8 pixels per lane per iteration: u as 2 x 16 B (f32) or 16 B (bf16), target as 8 B (u8) or 2 x 16 B (f32).
// du may alias u.  Per-lane sums -> wave shuffle -> LDS -> one partial per block -> fixed-order finisher.
template <typename T, typename TT>
__global__ __launch_bounds__(256) void mse_grad_kernel(const T* __restrict__ u, const TT* __restrict__ tgt,
                                                       T* __restrict__ du, long long n8, float inv_n,
                                                       float* __restrict__ partial, unsigned* __restrict__ counter,
                                                       float* __restrict__ loss_accum) {
    float lsum = 0.f;
    const float g2 = 2.f * inv_n;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float uu[8], tt[8];
        if (sizeof(T) == 4) {
            const float4 a = reinterpret_cast<const float4*>(u)[2 * i], b = reinterpret_cast<const float4*>(u)[2 * i + 1];
            uu[0] = a.x; uu[1] = a.y; uu[2] = a.z; uu[3] = a.w; uu[4] = b.x; uu[5] = b.y; uu[6] = b.z; uu[7] = b.w;
        } else {
            const bf16x8 a = reinterpret_cast<const bf16x8*>(u)[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) uu[k] = (float)a[k];
        }
        if (sizeof(TT) == 1) {
            const uint2 a = reinterpret_cast<const uint2*>(tgt)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tt[k] = (float)((a.x >> (8 * k)) & 0xFF) / 255.0f;       // helpers.py:121: uint8 / 255.0 in float32
                tt[4 + k] = (float)((a.y >> (8 * k)) & 0xFF) / 255.0f;
            }
        } else {
            const float4 a = reinterpret_cast<const float4*>(tgt)[2 * i], b = reinterpret_cast<const float4*>(tgt)[2 * i + 1];
            tt[0] = a.x; tt[1] = a.y; tt[2] = a.z; tt[3] = a.w; tt[4] = b.x; tt[5] = b.y; tt[6] = b.z; tt[7] = b.w;
        }
        float dd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float y = fminf(fmaxf(uu[k], 0.f), 1.f);
            const float diff = y - tt[k];
            lsum += diff * diff;
            dd[k] = (uu[k] >= 0.f && uu[k] <= 1.f) ? g2 * diff : 0.f;
        }
        if (sizeof(T) == 4) {
            reinterpret_cast<float4*>(du)[2 * i] = make_float4(dd[0], dd[1], dd[2], dd[3]);
            reinterpret_cast<float4*>(du)[2 * i + 1] = make_float4(dd[4], dd[5], dd[6], dd[7]);
        } else {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (bf16_t)dd[k];
            reinterpret_cast<bf16x8*>(du)[i] = o;
        }
    }
    __shared__ float wsum[4];
    __shared__ float sh[256];
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = lsum;
    __syncthreads();
    loss_block_finish((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), partial, counter, loss_accum, inv_n, sh);
}
int afr_mse_blocks(long long rows, long long cols) { return grid_for(rows * cols / 8, 256, 1024); }
hipError_t afr_launch_mse_grad(int act_dtype, const void* u, const void* target, int target_dtype, void* du,
                               long long rows, long long cols, long long mean_elems, float* loss_accum,
                               float* scratch, hipStream_t s) {
    const long long n = rows * cols;
    if (n <= 0) return hipSuccess;
    if (n & 7) return hipErrorInvalidValue;
    const int blocks = afr_mse_blocks(rows, cols);
    const float inv_n = (float)(1.0 / (double)mean_elems);
    dim3 g(blocks), b(256);
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 1024);
#define MSE(T, TT) hipLaunchKernelGGL((mse_grad_kernel<T, TT>), g, b, 0, s, (const T*)u, (const TT*)target, (T*)du, n / 8, inv_n, scratch, counter, loss_accum)
    if (act_dtype == AFR_BF16) {
        if (target_dtype == AFR_TARGET_U8) MSE(bf16_t, uint8_t); else MSE(bf16_t, float);
    } else {
        if (target_dtype == AFR_TARGET_U8) MSE(float, uint8_t); else MSE(float, float);
    }
#undef MSE
    return hipGetLastError();
}

// ------------------------------------------------------------------------- glyph embedding
// out[b][:] = Emb[x[b]][:] (+ Font[font[b]][:])  -- nn.Embedding gather (reference model.py:136,167): bit-exact row
// copy in f32 mode.  An index outside [0,vocab) sets bit 0 of *err_flag (the reference raises IndexError) and is
// clamped so the kernel never reads out of bounds.
template <typename T>
__global__ __launch_bounds__(256) void glyph_embed_kernel(const float* __restrict__ emb, const float* __restrict__ femb,
                                                          const int64_t* __restrict__ x, const int64_t* __restrict__ font,
                                                          int B, int E, int vocab, int n_fonts, T* __restrict__ out,
                                                          uint32_t* err_flag) {
    const long long total = (long long)B * E;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / E), c = (int)(i % E);
        long long xi = x[b];
        if (xi < 0 || xi >= vocab) { if (c == 0) atomicOr(err_flag, 1u); xi = min(max(xi, 0ll), (long long)vocab - 1); }
        float v = emb[xi * E + c];
        if (n_fonts > 0) {
            long long fi = font ? font[b] : 0;
            if (fi < 0 || fi >= n_fonts) { if (c == 0) atomicOr(err_flag, 1u); fi = min(max(fi, 0ll), (long long)n_fonts - 1); }
            v += femb[fi * E + c];
        }
        out[i] = (T)v;
    }
}
hipError_t afr_launch_glyph_embed(int act_dtype, const float* emb, const float* font_emb, const int64_t* x,
                                  const int64_t* font, int B, int E, int vocab, int n_fonts, void* out,
                                  uint32_t* err_flag, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    dim3 g(grid_for((long long)B * E, 256)), b(256);
    if (act_dtype == AFR_BF16)
        hipLaunchKernelGGL(glyph_embed_kernel<bf16_t>, g, b, 0, s, emb, font_emb, x, font, B, E, vocab, n_fonts, (bf16_t*)out, err_flag);
    else
        hipLaunchKernelGGL(glyph_embed_kernel<float>, g, b, 0, s, emb, font_emb, x, font, B, E, vocab, n_fonts, (float*)out, err_flag);
    return hipGetLastError();
}

// First Linear folded through the embedding tables.  The glyph model feeds h0 = Emb[x] + Font[f] straight into
// fc1 (no dropout, no nonlinearity in between), and a batch of thousands of glyphs draws from only vocab + n_fonts
// distinct rows, so   fc1(h0)[b] = T[x_b] + T[vocab + f_b] + b1   with   T = [Emb; Font] . W1^T   ((vocab+n_fonts) x N1).
// The table costs (vocab+n_fonts)*N1*E MACs per step instead of B*N1*E, stays in L2, and the layer becomes a gather.
// table[r][n] = sum_k tab(r)[k] * W1[n][k].  A block stages 256 fc1 rows (coalesced, padded to E+1 in LDS) and GT_ROWS table
// rows, one thread per n; grid (ceil(rows/8), ceil(N1/256)).
constexpr int GT_ROWS = 2;    // rows per block: the kernel is latency bound, more blocks (272 at C3) beat fewer, fatter ones (8: +1.5 us)
__global__ __launch_bounds__(256) void glyph_table_kernel(const float* __restrict__ emb, const float* __restrict__ femb,
                                                          const float* __restrict__ W1, int vocab, int rows, int E, int N1,
                                                          float* __restrict__ table, bf16_t* __restrict__ w1t) {
    extern __shared__ float sm[];                         // W [256][E+1] | tab [GT_ROWS][E]
    float* Ws = sm;
    float* tab = sm + 256 * (E + 1);
    const int r0 = blockIdx.x * GT_ROWS, n0 = blockIdx.y * 256;
    const int nn = min(256, N1 - n0), nr = min(GT_ROWS, rows - r0);
    const int E4 = E >> 2;
    for (int i = threadIdx.x; i < nn * E4; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(W1 + (size_t)n0 * E + 4 * i);
        float* d = Ws + (i / E4) * (E + 1) + 4 * (i % E4);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < nr * E; i += 256) {
        const int r = r0 + i / E, k = i % E;
        tab[i] = r < vocab ? emb[(size_t)r * E + k] : femb[(size_t)(r - vocab) * E + k];
    }
    __syncthreads();
    if ((int)threadIdx.x >= nn) return;
    const float* w = Ws + threadIdx.x * (E + 1);
    // the first row of blocks also leaves W1^T as bf16 [E][N1] (the fused first-layer backward reads fc1's weights k-contiguous)
    if (w1t && blockIdx.x == 0)
        for (int k = 0; k < E; ++k) w1t[(size_t)k * N1 + n0 + threadIdx.x] = (bf16_t)w[k];
    float a[GT_ROWS];
#pragma unroll
    for (int j = 0; j < GT_ROWS; ++j) a[j] = 0.f;
    for (int k = 0; k < E; ++k) {
        const float wv = w[k];
#pragma unroll
        for (int j = 0; j < GT_ROWS; ++j) a[j] = fmaf(tab[j * E + k], wv, a[j]);   // rows past nr read stale LDS; never stored
    }
    for (int j = 0; j < nr; ++j) table[(size_t)(r0 + j) * N1 + n0 + threadIdx.x] = a[j];
}
// h1[b][n] = relu(T[x_b][n] + T[vocab+f_b][n] + b1[n]), one lane per 8 consecutive n of one glyph; index checks as in
// glyph_embed_kernel.  Also leaves the backward's GEMM operand h0' [B][K0]: h0 = Emb[x_b] + Font[f_b] followed by the
// one-hot code of the two table rows the glyph used (see glyph_l1_bwd_kernel).
template <typename T>
__device__ __forceinline__ void store8(T* q, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 w;
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = (bf16_t)v[r];
        __builtin_nontemporal_store(w, reinterpret_cast<bf16x8*>(q));
    } else {
        const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        __builtin_nontemporal_store(lo, reinterpret_cast<f32x4*>(q));
        __builtin_nontemporal_store(hi, reinterpret_cast<f32x4*>(q) + 1);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void glyph_l1_fwd_kernel(const float* __restrict__ table, const float* __restrict__ b1,
                                                           const float* __restrict__ emb, const float* __restrict__ femb,
                                                           const int64_t* __restrict__ x, const int64_t* __restrict__ font,
                                                           int B, int E, int N1, int vocab, int n_fonts, int K0,
                                                           T* __restrict__ h0, T* __restrict__ h1, uint32_t* err_flag) {
    // one glyph per threadIdx.y, its 8-column chunks of h1 (N1/8) and then of h0' (K0/8) strided over threadIdx.x: no index
    // division (a flat index over B x (N1 + K0)/8 items cost a 64-bit divide + modulo by a runtime divisor per item)
    const int c1 = N1 >> 3, c0 = K0 >> 3;
    const int b = blockIdx.x * blockDim.y + threadIdx.y;
    if (b >= B) return;
    long long xi = x[b];
    if (xi < 0 || xi >= vocab) { if (threadIdx.x == 0) atomicOr(err_flag, 1u); xi = min(max(xi, 0ll), (long long)vocab - 1); }
    long long fi = 0;
    if (n_fonts > 0) {
        fi = font ? font[b] : 0;
        if (fi < 0 || fi >= n_fonts) { if (threadIdx.x == 0) atomicOr(err_flag, 1u); fi = min(max(fi, 0ll), (long long)n_fonts - 1); }
    }
    const float* trow_c = table + (size_t)xi * N1;
    const float* trow_f = table + (size_t)(vocab + fi) * N1;
    for (int c = threadIdx.x; c < c1; c += blockDim.x) {
        float v[8];
        const int n = 8 * c;
        const float4 a0 = *reinterpret_cast<const float4*>(trow_c + n), a1 = *reinterpret_cast<const float4*>(trow_c + n + 4);
        const float4 g0 = *reinterpret_cast<const float4*>(b1 + n), g1 = *reinterpret_cast<const float4*>(b1 + n + 4);
        v[0] = a0.x + g0.x; v[1] = a0.y + g0.y; v[2] = a0.z + g0.z; v[3] = a0.w + g0.w;
        v[4] = a1.x + g1.x; v[5] = a1.y + g1.y; v[6] = a1.z + g1.z; v[7] = a1.w + g1.w;
        if (n_fonts > 0) {
            const float4 f0 = *reinterpret_cast<const float4*>(trow_f + n), f1 = *reinterpret_cast<const float4*>(trow_f + n + 4);
            v[0] += f0.x; v[1] += f0.y; v[2] += f0.z; v[3] += f0.w; v[4] += f1.x; v[5] += f1.y; v[6] += f1.z; v[7] += f1.w;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
        store8(h1 + (size_t)b * N1 + n, v);
    }
    for (int c = threadIdx.x; c < c0; c += blockDim.x) {
        // h0' = [h0 | one-hot of x_b | one-hot of vocab + f_b | zero pad]
        float v[8];
        const int k0 = 8 * c;
        if (k0 < E) {                                     // E % 8 == 0: a chunk is all h0 or all one-hot
            const float* e = emb + (size_t)xi * E + k0;
            const float4 a0 = *reinterpret_cast<const float4*>(e), a1 = *reinterpret_cast<const float4*>(e + 4);
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
            if (n_fonts > 0) {
                const float* f = femb + (size_t)fi * E + k0;
                const float4 f0 = *reinterpret_cast<const float4*>(f), f1 = *reinterpret_cast<const float4*>(f + 4);
                v[0] += f0.x; v[1] += f0.y; v[2] += f0.z; v[3] += f0.w; v[4] += f1.x; v[5] += f1.y; v[6] += f1.z; v[7] += f1.w;
            }
        } else {
            const int r0 = k0 - E, hx = (int)xi, hf = n_fonts > 0 ? vocab + (int)fi : -1;
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = (r0 + r == hx || r0 + r == hf) ? 1.f : 0.f;
        }
        store8(h0 + (size_t)b * K0 + k0, v);
    }
}
hipError_t afr_launch_glyph_l1_fwd(int act_dtype, const float* emb, const float* font_emb, const float* W1, const float* b1,
                                   const int64_t* x, const int64_t* font, int B, int E, int N1, int vocab, int n_fonts,
                                   float* table, void* h0, void* h1, uint32_t* err_flag, hipStream_t s, void* w1t) {
    if (B <= 0) return hipSuccess;
    if ((N1 & 7) || (E & 7)) return hipErrorInvalidValue;
    const int K0 = afr_glyph_k0(E, vocab, n_fonts);
    const int rows = vocab + n_fonts;
    const size_t tlds = (size_t)(256 * (E + 1) + GT_ROWS * E) * sizeof(float);      // 35 KB at E = 32, 136 KB at the E = 128 limit
    static size_t tlds_set[16];                      // largest dynamic-LDS opt-in made so far, per device
    int dev = 0;
    if (tlds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && (dev < 0 || dev >= 16 || tlds_set[dev] < tlds)) {
        hipError_t e = hipFuncSetAttribute((const void*)glyph_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) tlds_set[dev] = tlds;
    }
    hipLaunchKernelGGL(glyph_table_kernel, dim3((rows + GT_ROWS - 1) / GT_ROWS, (N1 + 255) / 256), dim3(256), tlds, s, emb, font_emb,
                       W1, vocab, rows, E, N1, table, (bf16_t*)w1t);
    // threads: x over a glyph's 8-column chunks (a power of two up to 128), y over glyphs; 256 threads per block
    int tx = 32;
    while (tx < 128 && tx < N1 / 8) tx *= 2;
    dim3 b(tx, 256 / tx), g((B + b.y - 1) / b.y);
    if (act_dtype == AFR_BF16)
        hipLaunchKernelGGL(glyph_l1_fwd_kernel<bf16_t>, g, b, 0, s, table, b1, emb, font_emb, x, font, B, E, N1, vocab, n_fonts, K0,
                           (bf16_t*)h0, (bf16_t*)h1, err_flag);
    else
        hipLaunchKernelGGL(glyph_l1_fwd_kernel<float>, g, b, 0, s, table, b1, emb, font_emb, x, font, B, E, N1, vocab, n_fonts, K0,
                           (float*)h0, (float*)h1, err_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------- first layer as a COMBINATION table (training steps, bf16)
// A glyph is one of vocab x max(n_fonts, 1) (character, font) combinations, so the first layer's output has at most that many
// distinct rows.  A training step therefore does not materialise h1 [B][N1] (16 MB at C3, written once and read three times:
// by the next layer's forward, by its weight gradient and by its ReLU mask) but only
//     H1c[c][n] = bf16(relu(T[x][n] + b1[n] + T[vocab + f][n]))      c = x * max(n_fonts, 1) + f   (the gather kernel's arithmetic)
//     H0c[c][k] = bf16(Emb[x][k] + Font[f][k])
//     cidx[b]   = c of glyph b
// (0.5 MB + 16 KB + 32 KB, L2-resident); the GEMM kernels that consume h1 / h0 gather their operand rows through cidx while
// staging (GemmParams::a_rowmap / b_rowmap / aux_rowmap; LDS-DMA takes a per-lane source address, so a gathered row costs
// what a dense one does).  Same values bit for bit as the gather kernel's h1 / h0, same FLOPs in every product.
// One kernel: a block takes CB_COMBOS combinations x 256 fc1 rows (grid x: combination groups, then the batch's index
// blocks; grid y: 256-row chunks of fc1), stages its W1 rows through LDS as glyph_table_kernel does and runs the SAME
// arithmetic as the table + gather kernels, in the same order: t_c = fma chain over k from 0 of Emb[x][k] W1[n][k], t_f likewise
// for the font row, v = (t_c + b1[n]) + t_f, ReLU, one rounding to bf16 -- bit for bit the dense path's h1.
constexpr int CB_COMBOS = 4;
__global__ __launch_bounds__(256) void glyph_combo_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ emb, const float* __restrict__ femb,
                                                          const int64_t* __restrict__ x, const int64_t* __restrict__ font,
                                                          int B, int E, int N1, int vocab, int n_fonts, int combo_blocks,
                                                          bf16_t* __restrict__ h1c, int ld1, bf16_t* __restrict__ h0c, int* __restrict__ cidx,
                                                          bf16_t* __restrict__ w1t, uint32_t* err_flag) {
    extern __shared__ float sm[];                         // W [256][E+1] | emb rows [CB][E] | font rows [CB][E]
    const int nf = max(n_fonts, 1), ncombo = vocab * nf;
    if ((int)blockIdx.x >= combo_blocks) {                // the batch's combination indices (with the index check of the gather)
        if (blockIdx.y != 0) return;
        const int b = ((int)blockIdx.x - combo_blocks) * 256 + threadIdx.x;
        if (b >= B) return;
        long long xi = x[b], fi = (n_fonts > 0 && font) ? font[b] : 0;
        if (xi < 0 || xi >= vocab) { atomicOr(err_flag, 1u); xi = min(max(xi, 0ll), (long long)vocab - 1); }
        if (n_fonts > 0 && (fi < 0 || fi >= n_fonts)) { atomicOr(err_flag, 1u); fi = min(max(fi, 0ll), (long long)n_fonts - 1); }
        cidx[b] = (int)(xi * nf + fi);
        return;
    }
    float* Ws = sm;
    float* er = sm + 256 * (E + 1);
    float* fr = er + CB_COMBOS * E;
    const int c0 = blockIdx.x * CB_COMBOS, n0 = blockIdx.y * 256;
    const int nn = min(256, N1 - n0), nc = min(CB_COMBOS, ncombo - c0);
    const int E4 = E >> 2;
    for (int i = threadIdx.x; i < nn * E4; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(W1 + (size_t)n0 * E + 4 * i);
        float* d = Ws + (i / E4) * (E + 1) + 4 * (i % E4);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < nc * E; i += 256) {
        const int c = c0 + i / E, k = i % E, xi = c / nf, fi = c - xi * nf;
        er[i] = emb[(size_t)xi * E + k];
        fr[i] = n_fonts > 0 ? femb[(size_t)fi * E + k] : 0.f;
        if (blockIdx.y == 0) h0c[(size_t)c * E + k] = (bf16_t)(n_fonts > 0 ? er[i] + fr[i] : er[i]);      // H0c = bf16(Emb[x] + Font[f])
    }
    __syncthreads();
    if ((int)threadIdx.x >= nn) return;
    const float* w = Ws + threadIdx.x * (E + 1);
    // the first group of blocks also leaves W1^T as bf16 [E][N1] (the fused first-layer backward reads fc1's weights k-contiguous)
    if (w1t && blockIdx.x == 0)
        for (int k = 0; k < E; ++k) w1t[(size_t)k * N1 + n0 + threadIdx.x] = (bf16_t)w[k];
    float tc[CB_COMBOS], tf[CB_COMBOS];
#pragma unroll
    for (int j = 0; j < CB_COMBOS; ++j) tc[j] = tf[j] = 0.f;
    for (int k = 0; k < E; ++k) {
        const float wv = w[k];
#pragma unroll
        for (int j = 0; j < CB_COMBOS; ++j) {             // (rows past nc read stale LDS; never stored)
            tc[j] = fmaf(er[j * E + k], wv, tc[j]);
            tf[j] = fmaf(fr[j * E + k], wv, tf[j]);
        }
    }
    const float bias = b1[n0 + threadIdx.x];
    for (int j = 0; j < nc; ++j) {
        float v = tc[j] + bias;
        if (n_fonts > 0) v += tf[j];
        h1c[(size_t)(c0 + j) * ld1 + n0 + threadIdx.x] = (bf16_t)fmaxf(v, 0.f);
    }
}
// combination rows + combination indices (+ W1^T) in ONE launch; h1c [vocab * max(n_fonts,1)][ld1], h0c [..][E], cidx [B]
hipError_t afr_launch_glyph_combo(const float* emb, const float* font_emb, const float* W1, const float* b1, const int64_t* x,
                                  const int64_t* font, int B, int E, int N1, int vocab, int n_fonts, float* table, void* h1c,
                                  int ld1, void* h0c, int* cidx, uint32_t* err_flag, hipStream_t s, void* w1t) {
    if (B <= 0) return hipSuccess;
    if ((N1 & 7) || (E & 7) || E > 256) return hipErrorInvalidValue;
    (void)table;
    const size_t lds = (size_t)(256 * (E + 1) + 2 * CB_COMBOS * E) * sizeof(float);
    if (lds > 48 * 1024) return hipErrorInvalidValue;                  // the combination path is planned for E <= 32 .. 40
    const int ncombo = vocab * (n_fonts > 0 ? n_fonts : 1);
    const int combo_blocks = (ncombo + CB_COMBOS - 1) / CB_COMBOS;
    hipLaunchKernelGGL(glyph_combo_kernel, dim3(combo_blocks + (B + 255) / 256, (N1 + 255) / 256), dim3(256), lds, s, W1, b1, emb, font_emb,
                       x, font, B, E, N1, vocab, n_fonts, combo_blocks, (bf16_t*)h1c, ld1, (bf16_t*)h0c, cidx, (bf16_t*)w1t, err_flag);
    return hipGetLastError();
}

// columns of h0': E + (vocab + n_fonts) rounded up to 8
int afr_glyph_k0(int E, int vocab, int n_fonts) { return E + (vocab + n_fonts + 7) / 8 * 8; }

// Backward of the folded first layer.  The weight-gradient GEMM ran against h0' = [h0 | one-hot], so its split-K slabs
// hold, per fc1 row n, both dW1[n][0..E) and S[n][r] = sum over the glyphs that used table row r of d1[b][n] -- the
// segment sums that nn.Embedding's backward (model.py:309) needs, obtained on MFMA instead of a scatter-add.  Then
//     dTab[r][k] = sum_n S[n][r] W1[n][k]        (dEmb = rows < vocab, dFont = the rest)
// replaces the B x E x N1 input-gradient GEMM and the per-glyph scatter.  A block owns GL1_ROWS fc1 rows: it sums their
// slabs (fixed order), emits the compact dW1 rows and its partial dTab, which the grouped reduce sums in block order.
constexpr int GL1_ROWS = 8, GL1_NT = 1024;
__global__ __launch_bounds__(GL1_NT) void glyph_l1_bwd_kernel(const float* __restrict__ slabs, int nslabs, long long slab_stride,
                                                              const float* __restrict__ W1, int N1, int E, int R, int K0,
                                                              float* __restrict__ dw1, float* __restrict__ dtab_part) {
    extern __shared__ float sm[];                 // S [GL1_ROWS][K0] | W [GL1_ROWS][E] | part [G-1][GL1_ROWS*K0]
    float* S = sm;
    float* W = sm + GL1_ROWS * K0;
    float* part = W + GL1_ROWS * E;
    const int n0 = blockIdx.x * GL1_ROWS;
    const int nr = min(GL1_ROWS, N1 - n0);
    const int cnt4 = nr * K0 / 4;                 // K0 % 8 == 0; <= GL1_ROWS*K0/4 float4 columns
    // the block's threads form G groups of cnt4 lanes; group g sums slabs g, g+G, ... (all loads of a lane independent),
    // then the groups are added in group order
    const int full4 = GL1_ROWS * K0 / 4;
    const int G = max(1, min(GL1_NT / full4, 8));
    const int g = threadIdx.x / full4, i = threadIdx.x % full4;
    if (g < G && i < cnt4) {
        const float* src = slabs + (size_t)n0 * K0 + 4 * i;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = g; z < nslabs; z += G) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long long)z * slab_stride);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float* dst = g == 0 ? S : part + (size_t)(g - 1) * GL1_ROWS * K0;
        *reinterpret_cast<float4*>(dst + 4 * i) = a;
    }
    for (int j = threadIdx.x; j < nr * E; j += GL1_NT) W[j] = W1[(size_t)n0 * E + j];
    __syncthreads();
    if ((int)threadIdx.x < cnt4) {
        float4 a = *reinterpret_cast<float4*>(S + 4 * threadIdx.x);
        for (int q = 1; q < G; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)(q - 1) * GL1_ROWS * K0 + 4 * threadIdx.x);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(S + 4 * threadIdx.x) = a;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nr * E; j += GL1_NT) dw1[(size_t)n0 * E + j] = S[(j / E) * K0 + (j % E)];
    float* out = dtab_part + (size_t)blockIdx.x * R * E;
    for (int j = threadIdx.x; j < R * E; j += GL1_NT) {
        const int r = j / E, k = j % E;
        float a = 0.f;
        for (int q = 0; q < nr; ++q) a = fmaf(S[q * K0 + E + r], W[q * E + k], a);
        out[j] = a;
    }
}
int afr_glyph_l1_bwd_blocks(int N1) { return (N1 + GL1_ROWS - 1) / GL1_ROWS; }
hipError_t afr_launch_glyph_l1_bwd(const float* slabs, int nslabs, long long slab_stride, const float* W1, int N1, int E,
                                   int vocab, int n_fonts, float* dw1, float* dtab_part, hipStream_t s) {
    const int K0 = afr_glyph_k0(E, vocab, n_fonts);
    if (GL1_ROWS * K0 / 4 > GL1_NT) return hipErrorInvalidValue;      // K0 <= 512: vocab + n_fonts + E within one block
    const int G = std::max(1, std::min(GL1_NT / (GL1_ROWS * K0 / 4), 8));
    const size_t lds = ((size_t)GL1_ROWS * (K0 + E) + (size_t)(G - 1) * GL1_ROWS * K0) * sizeof(float);
    hipLaunchKernelGGL(glyph_l1_bwd_kernel, dim3(afr_glyph_l1_bwd_blocks(N1)), dim3(GL1_NT), lds, s, slabs, nslabs, slab_stride, W1, N1,
                       E, vocab + n_fonts, K0, dw1, dtab_part);
    return hipGetLastError();
}

// embedding_dense_backward (model.py:309): dEmb[x[b]] += d[b].  Deterministic, atomic-free: a block takes 256 glyph
// rows; thread (slot = tid>>5, c = tid&31) is the ONLY writer of LDS rows v with v%8 == slot, column c, and walks the
// block's rows in order.  Block partials go to slabs[block][(vocab+n_fonts)*E]; afr_launch_reduce sums them in order.
constexpr int EMB_BWD_ROWS = 32;
template <typename T>
__global__ __launch_bounds__(256) void glyph_embed_bwd_kernel(const T* __restrict__ d, const int64_t* __restrict__ x,
                                                              const int64_t* __restrict__ font, int B, int E, int vocab,
                                                              int n_fonts, float* __restrict__ slabs) {
    extern __shared__ float sm[];                  // acc [(vocab+n_fonts)][E] | tile [256][E+1] | ids [256] | fids [256]
    const int rows_tot = vocab + n_fonts;
    const int LDT = E + 1;
    float* acc = sm;
    float* tile = sm + (size_t)rows_tot * E;
    int* ids = reinterpret_cast<int*>(tile + (size_t)EMB_BWD_ROWS * LDT);
    int* fids = ids + EMB_BWD_ROWS;
    const int b0 = blockIdx.x * EMB_BWD_ROWS;
    const int nb = min(EMB_BWD_ROWS, B - b0);
    for (int i = threadIdx.x; i < rows_tot * E; i += 256) acc[i] = 0.f;
    for (int i = threadIdx.x; i < nb * E; i += 256) tile[(i / E) * LDT + (i % E)] = (float)d[(size_t)b0 * E + i];
    if ((int)threadIdx.x < nb) {
        long long xi = x[b0 + threadIdx.x];
        ids[threadIdx.x] = (int)min(max(xi, 0ll), (long long)vocab - 1);
        long long fi = (n_fonts > 0 && font) ? font[b0 + threadIdx.x] : 0;
        fids[threadIdx.x] = (int)min(max(fi, 0ll), (long long)max(n_fonts, 1) - 1);
    }
    __syncthreads();
    const int slot = threadIdx.x >> 5, c0 = threadIdx.x & 31;
    for (int c = c0; c < E; c += 32) {
        for (int r = 0; r < nb; ++r) {
            const int v = ids[r];
            const float val = tile[r * LDT + c];
            if ((v & 7) == slot) acc[v * E + c] += val;
            if (n_fonts > 0) {
                const int f = fids[r];
                if ((f & 7) == slot) acc[(vocab + f) * E + c] += val;
            }
        }
    }
    __syncthreads();
    float* out = slabs + (size_t)blockIdx.x * rows_tot * E;
    for (int i = threadIdx.x; i < rows_tot * E; i += 256) out[i] = acc[i];
}
int afr_embed_bwd_blocks(int B) { return (B + EMB_BWD_ROWS - 1) / EMB_BWD_ROWS; }
hipError_t afr_launch_glyph_embed_bwd(int act_dtype, const void* d, const int64_t* x, const int64_t* font, int B, int E,
                                      int vocab, int n_fonts, float* slabs, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t lds = ((size_t)(vocab + n_fonts) * E + (size_t)EMB_BWD_ROWS * (E + 1)) * sizeof(float) + 2 * EMB_BWD_ROWS * sizeof(int);
    dim3 g(afr_embed_bwd_blocks(B)), b(256);
    if (act_dtype == AFR_BF16)
        hipLaunchKernelGGL(glyph_embed_bwd_kernel<bf16_t>, g, b, lds, s, (const bf16_t*)d, x, font, B, E, vocab, n_fonts, slabs);
    else
        hipLaunchKernelGGL(glyph_embed_bwd_kernel<float>, g, b, lds, s, (const float*)d, x, font, B, E, vocab, n_fonts, slabs);
    return hipGetLastError();
}
