// gemm.hip -- the dense products of the hot path, hand-written for gfx950 MFMA.
//
//   C[m][n] = sum_k A(m,k) * B(n,k)   (+bias[n]) (relu) (* (aux[m][n] > 0))
//
// covers all three products of every Linear in the model (reference model.py:148,152,196 and their
// autograd transposes, model.py:309):
//   forward   y  = x . W^T      A = x  [M][K]  k-contiguous,  B = W  [N][K] k-contiguous
//   input grad dx = dy . W      A = dy [M][N]  k-contiguous,  B = W  [n][k'] : reduction index is the ROW -> B k-strided
//   weight grad dW = dy^T . x   A = dy [m][n]  k-strided,     B = x  [m][k'] k-strided (reduction over the batch)
//
// Two operand types:
//   f32   v_mfma_f32_32x32x2_f32  : exact f32 (one rounding per product, k-ordered fma chain) -- the parity mode
//   bf16  v_mfma_f32_16x16x32_bf16: bf16 operands, f32 accumulation -- the throughput mode
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave), double-buffered LDS, one barrier per
// K-tile.  bf16: tiles go global -> LDS by LDS-DMA (buffer_load ... lds: no VGPR staging, no ds_write -- ds_write
// bandwidth was the measured limiter of the register-staged version), tile t+1 in flight during the MFMAs on t.
// f32: register-staged (loads issued before the MFMAs, written to the other buffer after them).
// k-strided operands keep their [k][x] orientation in LDS; the bf16 path reads them with ds_read_b64_tr_b16
// (hardware transpose), the f32 path by plain indexing (one f32 per lane per MFMA).
#include "afr_common.h"
#include "../../include/afr.h"

// Block -> (tile, k-split).  The grid is 1-D; blocks b, b+8, b+16 ... share an XCD (and its 4 MiB L2), so the XCD's
// blocks are given a CONTIGUOUS range of work ids (bijective remap).  Work ids run over the k-split slowest, then
// the tile index of the LARGER operand, so that operand's tiles are fetched into one XCD's L2 only and the small
// operand is the one re-read by all eight (placement changes speed only, never results).
__device__ __forceinline__ void tile_of_block(const GemmParams& p, int BM, int BN, int& tm, int& tn, int& z) {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    z = v / tiles;
    const int t = v - z * tiles;
    if (p.N > p.M) { tn = t / tiles_m; tm = t - tn * tiles_m; }
    else { tm = t / tiles_n; tn = t - tm * tiles_n; }
}

// ------------------------------------------------------------------------------------------- f32
namespace f32k {
constexpr int BM = 128, BN = 128, BK = 32;

template <int LAY> struct Lds { static constexpr int LD = LAY ? 132 : 129; };

// global -> registers: 4 float4 per thread per operand tile
template <int LAY>
__device__ __forceinline__ void load_tile(const float* __restrict__ G, int ld, int X, int x0, int k0, int kend,
                                          int tid, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LAY == 0) {
            int row = idx >> 3, kc = idx & 7;
            int gx = x0 + row, gk = k0 + 4 * kc;
            if (gx < X && gk < kend) v = *reinterpret_cast<const float4*>(G + (size_t)gx * ld + gk);
        } else {
            int kr = idx >> 5, xc = idx & 31;
            int gk = k0 + kr, gx = x0 + 4 * xc;
            if (gk < kend && gx < X) v = *reinterpret_cast<const float4*>(G + (size_t)gk * ld + gx);
        }
        r[i] = v;
    }
}
// registers -> LDS image [k][x]
template <int LAY>
__device__ __forceinline__ void store_tile(float* S, int tid, const float4 (&r)[4]) {
    constexpr int LD = Lds<LAY>::LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + 256 * i;
        if (LAY == 0) {
            int row = idx >> 3, kc = idx & 7;
            S[(4 * kc + 0) * LD + row] = r[i].x;
            S[(4 * kc + 1) * LD + row] = r[i].y;
            S[(4 * kc + 2) * LD + row] = r[i].z;
            S[(4 * kc + 3) * LD + row] = r[i].w;
        } else {
            int kr = idx >> 5, xc = idx & 31;
            *reinterpret_cast<float4*>(S + kr * LD + 4 * xc) = r[i];
        }
    }
}

template <int ALAY, int BLAY>
__global__ __launch_bounds__(256) void gemm_f32(GemmParams p) {
    constexpr int LDA = Lds<ALAY>::LD, LDB = Lds<BLAY>::LD;
    constexpr int TILE = BK * LDA + BK * LDB;
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tm, tn, z;
    tile_of_block(p, BM, BN, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int klen = ((p.K + p.splitk - 1) / p.splitk + BK - 1) / BK * BK;
    const int kbeg = z * klen;
    const int kend = min(p.K, kbeg + klen);
    const float* A = reinterpret_cast<const float*>(p.A);
    const float* B = reinterpret_cast<const float*>(p.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    const bool do_cs = (ALAY == 1) && p.colsum != nullptr && tn == 0;   // fused bias gradient: column sums of A tiles
    float cs = 0.f;
    const int nt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nt > 0) {
        load_tile<ALAY>(A, p.lda, p.M, m0, kbeg, kend, tid, ra);
        load_tile<BLAY>(B, p.ldb, p.N, n0, kbeg, kend, tid, rb);
        store_tile<ALAY>(smem, tid, ra);
        store_tile<BLAY>(smem + BK * LDA, tid, rb);
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const bool more = (t + 1 < nt);
        if (more) {
            load_tile<ALAY>(A, p.lda, p.M, m0, kbeg + (t + 1) * BK, kend, tid, ra);
            load_tile<BLAY>(B, p.ldb, p.N, n0, kbeg + (t + 1) * BK, kend, tid, rb);
        }
        const float* As = smem + cur * TILE;
        const float* Bs = As + BK * LDA;
        const int kh = lane >> 5, li = lane & 31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[(kk + kh) * LDA + wm * 64 + li];
            float a1 = As[(kk + kh) * LDA + wm * 64 + 32 + li];
            float b0 = Bs[(kk + kh) * LDB + wn * 64 + li];
            float b1 = Bs[(kk + kh) * LDB + wn * 64 + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (ALAY == 1 && do_cs) {
            const int xx = tid & 127, hf = tid >> 7;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) cs += As[(hf * (BK / 2) + kk) * LDA + xx];
        }
        if (more) {
            float* Sn = smem + (cur ^ 1) * TILE;
            store_tile<ALAY>(Sn, tid, ra);
            store_tile<BLAY>(Sn + BK * LDA, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (ALAY == 1 && do_cs) {
        smem[tid] = cs;
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) p.colsum[(size_t)z * p.colsum_stride + m0 + tid] = smem[tid] + smem[tid + 128];
    }
    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; rows are m, columns n
    const int flags = p.flags;
    const bool out_bf16 = flags & AFR_GEMM_OUT_BF16;
    float* Cf = reinterpret_cast<float*>(p.C) + (size_t)z * p.slab_stride;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const float* aux = reinterpret_cast<const float*>(p.aux);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.N) continue;
            const float bias = (flags & AFR_GEMM_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (flags & AFR_GEMM_RELU) v = fmaxf(v, 0.f);
                if (flags & AFR_GEMM_RELU_MASK) v = (aux[(size_t)m * p.ldaux + n] > 0.f) ? v : 0.f;
                if (out_bf16) Cb[(size_t)m * p.ldc + n] = f32_to_bf16(v);
                else Cf[(size_t)m * p.ldc + n] = v;
            }
        }
}
}  // namespace f32k

// ------------------------------------------------------------------------------------------ bf16
namespace bf16k {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile, either orientation

__device__ __forceinline__ int fswz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// LDS images (bytes) -- both are written LINEARLY by LDS-DMA (one wave-instruction = 64 lanes x 16 B = 1 KiB of
// consecutive LDS), so the bank swizzle is applied to the per-lane SOURCE address and again on the read:
//   LAY 0: [128 x][64 k]  128-B rows; position c' of row r holds global 16-B chunk c' ^ (r&7)   -> conflict-free ds_read_b128
//   LAY 1: [64 k][128 x]  256-B rows; position c' of row k holds global chunk c' ^ (fswz(k)<<1)  -> conflict-free tr reads
// stage_tile: this wave's share (4 of the tile's 16 KiB) of one operand tile, global -> LDS with no VGPR staging and
// no ds_write.  Rows / k beyond the operand get a voffset past the descriptor's range: the hardware returns zeros.
template <int LAY>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int X, int x0, int k0,
                                           int kend, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        int gx, gk;
        if (LAY == 0) {
            const int r = inst * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            gx = x0 + r; gk = k0 + 8 * c;
        } else {
            const int kr = inst * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (fswz(kr) << 1);
            gk = k0 + kr; gx = x0 + 8 * c;
        }
        unsigned off = (LAY == 0) ? (unsigned)(((size_t)gx * ld + gk) * 2) : (unsigned)(((size_t)gk * ld + gx) * 2);
        if (gx >= X || gk >= kend) off = 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_tile + inst * 1024), 16, off, 0, 0, 0);
    }
}
// fragment for 16 x-rows starting at xb (multiple of 16), k-step ks (32 k each): lane holds
// X(x = xb + (lane&15), k = 32*ks + 8*(lane>>4) + j), j = 0..7
template <int LAY>
__device__ __forceinline__ bf16x8 read_frag(const char* S, int xb, int ks, int lane) {
    if (LAY == 0) {
        int row = xb + (lane & 15);
        int c = (lane >> 4) + 4 * ks;
        return *reinterpret_cast<const bf16x8*>(S + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        int kb = 32 * ks + 8 * (lane >> 4);
        int q = (lane >> 2) & 3, pp = lane & 3;
        int xblk = xb >> 4;
        int k0r = kb + q, k1r = kb + 4 + q;
        const char* a0 = S + k0r * 256 + ((xblk ^ fswz(k0r)) << 5) + pp * 8;
        const char* a1 = S + k1r * 256 + ((xblk ^ fswz(k1r)) << 5) + pp * 8;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int ALAY, int BLAY>
__global__ __launch_bounds__(256, 2) void gemm_bf16(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // 2 buffers x (A,B) = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int tm, tn, z;
    tile_of_block(p, BM, BN, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int klen = ((p.K + p.splitk - 1) / p.splitk + BK - 1) / BK * BK;
    const int kbeg = z * klen;
    const int kend = min(p.K, kbeg + klen);
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_cs = (ALAY == 1) && p.colsum != nullptr && tn == 0;   // fused bias gradient: column sums of A tiles
    float cs = 0.f;
    const int nt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    const int wave = __builtin_amdgcn_readfirstlane(wid);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7FFFFFFF, 0x00020000);
    if (nt > 0) {
        stage_tile<ALAY>(rA, smem, p.lda, p.M, m0, kbeg, kend, wave, lane);
        stage_tile<BLAY>(rB, smem + TILE_BYTES, p.ldb, p.N, n0, kbeg, kend, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        // tile t+1 streams into the other buffer (LDS-DMA, in flight during the MFMAs below); every wave finished
        // reading that buffer before the barrier that ended the previous iteration
#ifndef AFR_ABLATE_NOLOAD
        if (t + 1 < nt) {
            char* Sn = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile<ALAY>(rA, Sn, p.lda, p.M, m0, kbeg + (t + 1) * BK, kend, wave, lane);
            stage_tile<BLAY>(rB, Sn + TILE_BYTES, p.ldb, p.N, n0, kbeg + (t + 1) * BK, kend, wave, lane);
        }
#endif
        const char* As = smem + cur * 2 * TILE_BYTES;
        const char* Bs = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<ALAY>(As, wm * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = read_frag<BLAY>(Bs, wn * 64 + 16 * j, ks, lane);
            // operands swapped on purpose: D'[n][m] so that a lane owns 4 consecutive n of one row m
#ifndef AFR_ABLATE_NOMFMA
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(af[i])); asm volatile("" ::"v"(bfr[i])); }
#endif
        }
        if (ALAY == 1 && do_cs) {
            const int xx = tid & 127, hf = tid >> 7;
#pragma unroll 8
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int k = hf * (BK / 2) + kk;
                cs += (float)*reinterpret_cast<const bf16_t*>(As + k * 256 + (((xx >> 4) ^ fswz(k)) << 5) + (xx & 15) * 2);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the DMA of tile t+1 has landed; my reads of tile t are done
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }

    if (ALAY == 1 && do_cs) {
        float* red = reinterpret_cast<float*>(smem);
        red[tid] = cs;
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) p.colsum[(size_t)z * p.colsum_stride + m0 + tid] = red[tid] + red[tid + 128];
    }
    // epilogue: acc[i][j][r] = C[m = m0 + wm*64 + 16i + (lane&15)][n = n0 + wn*64 + 16j + 4*(lane>>4) + r]
    const int flags = p.flags;
    const bool out_bf16 = flags & AFR_GEMM_OUT_BF16;
    float* Cf = reinterpret_cast<float*>(p.C) + (size_t)z * p.slab_stride;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const bf16_t* aux = reinterpret_cast<const bf16_t*>(p.aux);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * (lane >> 4);
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (flags & AFR_GEMM_BIAS) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (flags & AFR_GEMM_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (flags & AFR_GEMM_RELU_MASK) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(aux + (size_t)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((float)a[r] > 0.f) ? v[r] : 0.f;
            }
            if (out_bf16) {
                bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(Cb + (size_t)m * p.ldc + n) = o;
            } else {
                *reinterpret_cast<f32x4*>(Cf + (size_t)m * p.ldc + n) = v;
            }
        }
    }
}
}  // namespace bf16k

// ---------------------------------------------------------------------------------------- launch
const char* afr_gemm_kernel_name(int dtype, int flags) {
    const int a = (flags & AFR_GEMM_A_KSTRIDED) ? 1 : 0, b = (flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0;
    static const char* names[2][2][2] = {
        {{"gemm_f32<0,0>", "gemm_f32<0,1>"}, {"gemm_f32<1,0>", "gemm_f32<1,1>"}},
        {{"gemm_bf16<0,0>", "gemm_bf16<0,1>"}, {"gemm_bf16<1,0>", "gemm_bf16<1,1>"}}};
    return names[dtype == AFR_BF16][a][b];
}

hipError_t afr_launch_gemm(int dtype, const GemmParams& p, hipStream_t s) {
    const int a = (p.flags & AFR_GEMM_A_KSTRIDED) ? 1 : 0, b = (p.flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0;
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    dim3 grid(tiles * p.splitk, 1, 1), block(256, 1, 1);
    if (tiles <= 0) return hipSuccess;
#define LAUNCH(NS, KRN, AL, BL) hipLaunchKernelGGL((NS::KRN<AL, BL>), grid, block, 0, s, p)
    if (dtype == AFR_BF16) {
        if (!a && !b) LAUNCH(bf16k, gemm_bf16, 0, 0);
        else if (!a && b) LAUNCH(bf16k, gemm_bf16, 0, 1);
        else if (a && !b) LAUNCH(bf16k, gemm_bf16, 1, 0);
        else LAUNCH(bf16k, gemm_bf16, 1, 1);
    } else {
        if (!a && !b) LAUNCH(f32k, gemm_f32, 0, 0);
        else if (!a && b) LAUNCH(f32k, gemm_f32, 0, 1);
        else if (a && !b) LAUNCH(f32k, gemm_f32, 1, 0);
        else LAUNCH(f32k, gemm_f32, 1, 1);
    }
#undef LAUNCH
    return hipGetLastError();
}
