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
#include <cstdlib>
#include "afr_common.h"
#include "../../include/afr.h"

// Block -> (tile, k-split).  The grid is 1-D; blocks b, b+8, b+16 ... share an XCD (and its 4 MiB L2), so the XCD's
// blocks are given a CONTIGUOUS range of work ids (bijective remap).  Work ids run over the k-split slowest; inside a
// split, tiles are walked in groups of 8 along the dimension of the LARGER operand: for each of the group's 8 slow
// indices... i.e. the 8 big-operand tiles of a group stay hot while the small operand is swept once per GROUP rather
// than once per slow index (R0's dW GEMM re-read its 13 MB activation operand 75x, 1 GB of extra HBM reads per step,
// before this).  Placement changes speed only, never results.
// block -> position in its XCD's contiguous range of the walk (blocks b, b+8, ... share an XCD)
__device__ __forceinline__ int xcd_walk(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// walk position t (0 .. tiles-1) -> tile coordinates; q, r describe the XCD ranges the walk is cut into (nb / 8, nb % 8)
__device__ __forceinline__ void tile_coords(const GemmParams& p, int BM, int BN, int t, int q, int r, int& tm, int& tn) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const bool n_slow = p.N > p.M;                   // the slow (grouped) dimension is the larger operand's
    const int ts = n_slow ? tiles_n : tiles_m, tf = n_slow ? tiles_m : tiles_n;
    // 8 slow tiles per group, or fewer when that makes an XCD's contiguous range exactly one group: then the group's
    // slow-operand tiles enter ONE L2 (C3 forward: 8192x1024 output = 32x8 tiles, 32 per XCD -> groups of 4x8; with
    // groups of 8x8 two XCDs shared each group and the 16 MB activation operand was fetched twice)
    int G = 8;
    if (r == 0 && q % tf == 0 && q / tf >= 1 && q / tf < 8 && tiles % q == 0) G = q / tf;
    const int grp = t / (G * tf);
    const int s0 = grp * G;
    const int gs = min(G, ts - s0);
    const int rr = t - grp * G * tf;
    // Inside a group the FAST index runs fastest when the fast dimension is only a few tiles (R0's fc_output: 4 row tiles of
    // the batch against 150 column tiles of the 246 MB weight): the tiles that share one big-operand tile are then adjacent in
    // the walk, so an XCD range that does not start on a group boundary (600 tiles / 8 XCDs = 75, groups of 32) shares ONE
    // big-operand tile with its neighbour instead of a whole group's -- with the slow index fastest every straddled group's
    // weight tiles were fetched by two XCDs (PMC: 743 MB read against 259 MB of operands).
    const bool fast_first = tf <= 8;
    const int sidx = fast_first ? s0 + rr / tf : s0 + rr % gs, fidx = fast_first ? rr % tf : rr / gs;
    tm = n_slow ? fidx : sidx;
    tn = n_slow ? sidx : fidx;
}
__device__ __forceinline__ void tile_of_block(const GemmParams& p, int BM, int BN, int b, int nb, int& tm, int& tn, int& z) {
    const int v = xcd_walk(b, nb);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    z = v / tiles;
    tile_coords(p, BM, BN, v - z * tiles, nb >> 3, nb & 7, tm, tn);
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
    tile_of_block(p, BM, BN, blockIdx.x, gridDim.x, tm, tn, z);
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
    const bool mse = p.mse_target != nullptr;
    float* Cf = reinterpret_cast<float*>(p.C) + (size_t)z * p.slab_stride;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const float* aux = reinterpret_cast<const float*>(p.aux);
    float lsum = 0.f;
    const float g2 = 2.f * p.mse_inv_n;
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
                if (mse) {
                    if (out_bf16) v = (float)(bf16_t)v;
                    const size_t ti = (size_t)m * p.N + n;
                    const float t = p.mse_target_dtype == AFR_TARGET_U8 ? (float)reinterpret_cast<const uint8_t*>(p.mse_target)[ti] / 255.0f
                                                                        : reinterpret_cast<const float*>(p.mse_target)[ti];
                    const float diff = fminf(fmaxf(v, 0.f), 1.f) - t;
                    lsum += diff * diff;
                    v = (v >= 0.f && v <= 1.f) ? g2 * diff : 0.f;
                }
                if (p.ad_p) {                          // fused AdamW on weight element (m, n); v is its gradient
                    const size_t wi = (size_t)m * p.ldc + n;
                    float pp = p.ad_p[wi], mm = p.ad_m[wi], vv = p.ad_v[wi];
                    adamw_elem(pp, mm, vv, v, p.ad_decay, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step, p.ad_rsqrt_bc2);
                    p.ad_p[wi] = pp; p.ad_m[wi] = mm; p.ad_v[wi] = vv;
                } else if (out_bf16) Cb[(size_t)m * p.ldc + n] = f32_to_bf16(v);
                else Cf[(size_t)m * p.ldc + n] = v;
            }
        }
    if (mse) {
        __syncthreads();
        lsum = wave_sum(lsum);
        if (lane == 0) smem[wid] = lsum;
        __syncthreads();
        loss_block_finish((smem[0] + smem[1]) + (smem[2] + smem[3]), p.mse_partial, p.mse_counter, p.mse_loss_accum, p.mse_inv_n, smem + 16);
    }
}
}  // namespace f32k

// ------------------------------------------------------------------------------------------ bf16
namespace bf16k {
constexpr int BN = 128, BK = 64;
constexpr int SUB = 128 * 64 * 2;   // 16 KiB: one 128-wide operand sub-tile (either orientation)

__device__ __forceinline__ int fswz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// LDS sub-tile images (bytes) -- written LINEARLY by LDS-DMA (one wave-instruction = 64 lanes x 16 B = 1 KiB of
// consecutive LDS), so the bank swizzle is applied to the per-lane SOURCE address and again on the read:
//   LAY 0: [128 x][64 k]  128-B rows; position c' of row r holds global 16-B chunk c' ^ (r&7)   -> conflict-free ds_read_b128
//   LAY 1: [64 k][128 x]  256-B rows; position c' of row k holds global chunk c' ^ (fswz(k)<<1)  -> conflict-free tr reads
// stage_inst: wave-instruction `inst` (0..15) of one sub-tile, global -> LDS with no VGPR staging and no ds_write.
// Rows / k beyond the operand get a voffset past the descriptor's range: the hardware returns zeros.
// The DMA is issued from inline asm on purpose: when hipcc sees a pending LDS-DMA it puts `s_waitcnt vmcnt(0)` in
// front of the next ds_read_b64_tr_b16 / plain LDS load (measured: the whole 3-stage ring drained every K-step on the
// k-strided orientations).  Hidden from the compiler, the only waits are the counted ones placed by hand below.
// M0 (the LDS destination base) is compiler-reserved: saved and restored inside the same statement.
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_addr, unsigned voff) {
    unsigned keep;
#ifdef AFR_GEMM_TIMING
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // the stamp branches cost the compiler its uniformity proof
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
    return r;
}
// byte offset of lane `lane`'s 16 source bytes of piece `inst` of a sub-tile (out of range -> past the descriptor: zeros)
template <int LAY>
__device__ __forceinline__ unsigned piece_voff(int ld, int X, int x0, int k0, int kend, int inst, int lane) {
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
    return off;
}
// rowmap (optional): the operand's memory row (x for LAY 0, k for LAY 1) is row rowmap[row] of the table behind rsrc
template <int LAY>
__device__ __forceinline__ void stage_inst(i32x4 rsrc, unsigned lds_sub, int ld, int X, int x0, int k0,
                                           int kend, int inst, int lane, const int* rowmap = nullptr) {
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
    const bool oob = gx >= X || gk >= kend;
    int rx = gx, rk = gk;
    if (rowmap && !oob) { if (LAY == 0) rx = rowmap[gx]; else rk = rowmap[gk]; }
    unsigned off = (LAY == 0) ? (unsigned)(((size_t)rx * ld + gk) * 2) : (unsigned)(((size_t)rk * ld + gx) * 2);
    if (oob) off = 0x80000000u;
    dma16(rsrc, lds_sub + inst * 1024, off);
}
// fragment for 16 x-rows starting at xb (multiple of 16, within the sub-tile), k-step ks (32 k each): lane holds
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

// WM = waves along M: tile (64*WM) x 128, 2*WM waves of 64x64.
//   WM=2: 128x128, 256 threads, 2 LDS stages (64 KiB) -> 2 blocks/CU; for small grids.
//   WM=4: 256x128, 512 threads, 3-stage LDS ring (144 KiB), tile t+2 in flight behind a COUNTED vmcnt and a raw
//         s_barrier: the main loop was latency-bound on global->LDS with one tile in flight, and the bigger tile
//         needs 25 % fewer DMA bytes per FLOP.
// p/m/v accessors of the fused AdamW tails.  Stores are streaming: the values are not read again before the next step (measured
// -1 % on the R0 dW kernel); the loads of wave_epilogue's tail are streaming buffer loads (see there), strip_finish's are ordinary.
#define ADLD(ptr) (*reinterpret_cast<const float4*>(ptr))
__device__ __forceinline__ void nt_st4(float* q, float4 v) {
    const f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(q));
}
#define ADST(ptr, val) nt_st4(ptr, val)
// Kernel-development build (-DAFR_GEMM_TIMING, never the shipped library): thread 0 of every block leaves wall-clock
// stamps (s_memrealtime, 10 ns ticks) at entry / first tile landed / K loop done / stores drained, plus its XCC and CU
// ids, in a buffer of its own (tools/gemm_timeline.py); no output value depends on them.
#ifdef AFR_GEMM_TIMING
__device__ unsigned long long* g_gemm_stamps = nullptr;
#define GSLOT ((size_t)p.dbg_slot * 1024 + blockIdx.x)
#define GSTAMP(i) do { if (stamps && tid == 0) stamps[GSLOT * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
// The epilogue of one wave's 64x64 f32 accumulator tile whose top-left element is (mb, nb0) of the product; Wt = the
// wave's own 16 KiB of LDS.  Shared by the 256x128 / 128x128 kernels (one call) and the 256x256 kernel (two calls).
// the bias values a lane adds in the row-contiguous pass of wave_epilogue: its 8 (bf16 output) or 4 (f32 output) columns
__device__ __forceinline__ void epilogue_bias(const GemmParams& p, const int nb0, const int lane, float (&bia)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) bia[r] = 0.f;
    if (!(p.flags & AFR_GEMM_BIAS)) return;
    if (p.flags & AFR_GEMM_OUT_BF16) {
        const int n = nb0 + 8 * (lane & 7);
        if (n < p.N) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bia[0] = b0.x; bia[1] = b0.y; bia[2] = b0.z; bia[3] = b0.w; bia[4] = b1.x; bia[5] = b1.y; bia[6] = b1.z; bia[7] = b1.w;
        }
    } else if (nb0 + 4 * (lane & 15) < p.N) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + nb0 + 4 * (lane & 15));
        bia[0] = b0.x; bia[1] = b0.y; bia[2] = b0.z; bia[3] = b0.w;
    }
}
// (acc_at(i, j): the wave's accumulator fragment of rows 16 i .., columns 16 j .. -- an accessor, so that the 256x256 body can hand
// over either half of its 128 x 64 tile without copying 64 registers)
template <int ALAY, int BLAY, int WM, bool SCALE = false, bool EARLYB_ = true, class ACC>      // SCALE: the accumulators are multiplied by p.out_scale first (fp8 per-tensor scales)
__device__ __forceinline__ void wave_epilogue_at(const GemmParams& p, const ACC& acc_at, const int mb, const int nb0, const int z,
                                                 float* Wt, const int lane, float& lsum, const float* lut255 = nullptr,
                                                 const float (*pre_bias)[8] = nullptr) {
    // Epilogue through LDS: the MFMA accumulators hold 4 consecutive n of 16 different rows per lane-group, which as
    // direct stores would be 32-byte pieces.  Each wave parks its 64x64 f32 tile in its own 16 KiB of LDS (16-B chunks
    // XOR-swizzled by row: conflict-free both ways) and streams it out row-contiguous: 8 lanes x 8 values = one 64-col
    // row, so stores (and the aux / target loads of the fused tails) are whole 128-B / 256-B row segments.
    // bias / ReLU in the row pass: the forward layout only (what the engine's Linear layers use); the other layouts, where the
    // op-level API alone can ask for them, keep the per-fragment form and carry no extra code in their row passes
    constexpr bool EARLYB = EARLYB_ && ALAY == 0 && BLAY == 0;
    const int flags = p.flags;
    const bool out_bf16 = flags & AFR_GEMM_OUT_BF16;
    const bool mse = p.mse_target != nullptr;
    const int c8 = lane & 7;
    const int n = nb0 + 8 * c8;
    const bool ncol = n < p.N;
    // The bias is added (and the ReLU taken) in the row-contiguous pass below, where a lane owns the same 8 (bf16 output) or
    // 4 (f32 output) columns of every row: its few bias values are requested HERE, before the accumulators are parked, so the
    // load's latency (2 us per tile on the K = 512 products of the pixel transformer when it was taken per accumulator
    // fragment at this point) passes under the LDS round trip.  Same f32 additions, same results.
    // (pre_bias: the 256x256 body calls twice for the same columns and asks once, before its first call.  EARLYB_ = false keeps
    // the per-fragment form of every layout.)
    float bia[8];
    if (pre_bias) {
#pragma unroll
        for (int r = 0; r < 8; ++r) bia[r] = (*pre_bias)[r];
    } else if (EARLYB) epilogue_bias(p, nb0, lane, bia);
    const bool relu = EARLYB && (flags & AFR_GEMM_RELU);
    const bool rowbias = EARLYB && (flags & AFR_GEMM_BIAS);
    // the fused loss' uint8 targets of the wave's 8 row passes (8 bytes per lane and pass) are requested before the park as well
    // (forward layout on the ring kernels only: the 256x256 body takes no fused loss)
    uint2 tu8[8];
    const bool early_t = EARLYB && pre_bias == nullptr && mse && p.mse_target_dtype == AFR_TARGET_U8;
    if (early_t) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int m = mb + ps * 8 + (lane >> 3);
            if (m < p.M && ncol) tu8[ps] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.mse_target) + (size_t)m * p.N + n);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = 16 * j + 4 * (lane >> 4);
            f32x4 v = acc_at(i, j);
            if (SCALE) v *= p.out_scale;
            if (!EARLYB) {
                if ((flags & AFR_GEMM_BIAS) && nb0 + nl < p.N) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + nb0 + nl);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (flags & AFR_GEMM_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            *reinterpret_cast<f32x4*>(Wt + ml * 64 + (((nl >> 2) ^ (ml & 15)) << 2)) = v;
        }
    }
    float* Cf = reinterpret_cast<float*>(p.C) + (size_t)z * p.slab_stride;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const bf16_t* aux = reinterpret_cast<const bf16_t*>(p.aux);
    const float g2 = 2.f * p.mse_inv_n;
    const bool relu_mask = flags & AFR_GEMM_RELU_MASK;
    // the tails' global operands (aux for the ReLU mask, targets for the fused loss) are fetched for all 8 row passes
    // up front: one memory latency instead of eight serial ones
    bf16x8 auxv[8];
    // ReLU gate as bits (mask_in, one byte per 8 columns): the 64 rows x 8 bytes of the wave's tile are ONE 8-byte load per lane
    // (lane l: row mb + l), handed to the lane that owns (row, 8-column group) in each pass by two wave shuffles -- as 8 byte loads
    // per lane the dX products of the pixel transformer spent 15 us of a 30 us workgroup life in this tail.  Needs 8-byte aligned
    // rows (ldmask % 8 == 0, i.e. N % 64 == 0); otherwise the byte loads below.
    const bool wide_bits = ALAY == 0 && relu_mask && p.mask_in && out_bf16 && (p.ldmask & 7) == 0;      // (ALAY == 1: weight gradients, no gate)
    uint2 mrow = {0u, 0u};
    if (wide_bits && mb + lane < p.M && nb0 < p.N)
        mrow = *reinterpret_cast<const uint2*>(p.mask_in + (size_t)(mb + lane) * p.ldmask + (nb0 >> 3));
    if ((relu_mask && !wide_bits) || (mse && !early_t && p.mse_target_dtype == AFR_TARGET_U8)) {
        int am[8];                                   // aux rows (gathered through aux_rowmap when the mask operand is a table)
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int m = mb + ps * 8 + (lane >> 3);
            am[ps] = (relu_mask && p.aux_rowmap && m < p.M) ? p.aux_rowmap[m] : m;
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int m = mb + ps * 8 + (lane >> 3);
            if (m < p.M && ncol) {
                if (relu_mask && !wide_bits) {
                    if (p.mask_in) auxv[ps][0] = __builtin_bit_cast(bf16_t, (unsigned short)p.mask_in[(size_t)m * p.ldmask + (n >> 3)]);   // the 8 bits travel in element 0
                    else auxv[ps] = *reinterpret_cast<const bf16x8*>(aux + (size_t)am[ps] * p.ldaux + n);
                }
                if (mse && !early_t) tu8[ps] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.mse_target) + (size_t)m * p.N + n);
            }
        }
    }
    if (!out_bf16) {
        // f32 outputs (split-K slabs, direct gradients, fused AdamW): 16 lanes x 16 B cover one 64-column row, so every
        // wave-instruction moves four whole 256-B row segments (with 8 columns per lane as below, an f32 row would be
        // written and read as two interleaved half-filled passes over the same cache lines).  16 passes of 4 rows.
        // Fused AdamW (weight-gradient GEMMs only): p, m, v of the wave's 64x64 tile are 16 passes x 3 x 16 B per lane.  Such a launch is
        // bound by how many of those loads a CU keeps in flight (measured on R0: 770 / 745 / 726 / 713 us with 3 / 7 / 11 / 15
        // passes ahead), so the 4-wave kernel (256 VGPRs per lane) keeps 13 of the 16 passes in flight from the moment the accumulators are parked
        // in LDS.  (Issuing some before the K loop was slower: vmcnt is in-order, so the ring's counted waits then also wait
        // for these HBM loads.)
        constexpr bool ADAM = (ALAY == 1 && BLAY == 1);
        constexpr int ADF = !ADAM ? 1 : (WM == 2) ? 14 : 4;    // 14: the deepest that allocates without scratch
        const int c4 = lane & 15;
        const int nf = nb0 + 4 * c4;
        const bool okc = nf < p.N;
        float4 qp[ADF], qm[ADF], qv[ADF];
        // The optimizer state is read as STREAMING data (buffer loads with the nt bit): 3 GB of p/m/v pass through the L2s during
        // R0's weight-gradient product, and without the hint they displace the operand slabs the 64 concurrent tiles of an XCD
        // share (PMC: 2.37 GB fetched per launch against 1.47 GB of state + 52 MB of operands).  Measured on R0: 655 -> 602 us.
        // (sc1 / sc0|sc1 instead: no change.  The addresses are tile-relative 32-bit offsets on a descriptor of the tile's origin.)
        const size_t tile0 = (size_t)mb * p.ldc + nb0;
        const bool adam_ld = ADAM && p.ad_p != nullptr;
        const __amdgpu_buffer_rsrc_t rp_ = __builtin_amdgcn_make_buffer_rsrc((void*)(adam_ld ? p.ad_p + tile0 : nullptr), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm_ = __builtin_amdgcn_make_buffer_rsrc((void*)(adam_ld ? p.ad_m + tile0 : nullptr), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv_ = __builtin_amdgcn_make_buffer_rsrc((void*)(adam_ld ? p.ad_v + tile0 : nullptr), 0, 0x7FFFFFFF, 0x00020000);
        auto load_f = [&](int ps, int buf) {
            const int m = mb + ps * 4 + (lane >> 4);
            if (m < p.M && okc) {
                const unsigned vo = (unsigned)(((ps * 4 + (lane >> 4)) * p.ldc + 4 * c4) * 4);
                qp[buf] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp_, vo, 0, 2));
                qm[buf] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rm_, vo, 0, 2));
                qv[buf] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv_, vo, 0, 2));
            }
        };
        const bool adam = ADAM && p.ad_p != nullptr;
        if (adam) {
#pragma unroll
            for (int q = 0; q < ADF - 1; ++q) load_f(q, q);
        }
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 4 + (lane >> 4);
            if (adam && ps + ADF - 1 < 16) load_f(ps + ADF - 1, (ps + ADF - 1) % ADF);
            f32x4 g = *reinterpret_cast<const f32x4*>(Wt + rl * 64 + ((c4 ^ (rl & 15)) << 2));
#pragma unroll
            for (int r = 0; r < 4; ++r) { g[r] = rowbias ? g[r] + bia[r] : g[r]; g[r] = relu ? fmaxf(g[r], 0.f) : g[r]; }
            const int m = mb + rl;
            if (m >= p.M || !okc) continue;
            const size_t wi = (size_t)m * p.ldc + nf;
            if (relu_mask) {                            // only the op-level API combines the mask with f32 output
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(aux + (size_t)m * p.ldaux + nf);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = ((float)a[r] > 0.f) ? g[r] : 0.f;
            }
            if (adam) {
                const int bf = ps % ADF;
                float pp[4] = {qp[bf].x, qp[bf].y, qp[bf].z, qp[bf].w}, mm[4] = {qm[bf].x, qm[bf].y, qm[bf].z, qm[bf].w};
                float vv[4] = {qv[bf].x, qv[bf].y, qv[bf].z, qv[bf].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) adamw_elem(pp[r], mm[r], vv[r], g[r], p.ad_decay, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step, p.ad_rsqrt_bc2);
                ADST(p.ad_p + wi, make_float4(pp[0], pp[1], pp[2], pp[3]));
                ADST(p.ad_m + wi, make_float4(mm[0], mm[1], mm[2], mm[3]));
                ADST(p.ad_v + wi, make_float4(vv[0], vv[1], vv[2], vv[3]));
                if (p.ad_shadow) {
                    bf16x4 o = {(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(p.ad_shadow + wi));
                }
            } else {
                nt_st4(Cf + wi, make_float4(g[0], g[1], g[2], g[3]));
            }
        }
        return;
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int rl = ps * 8 + (lane >> 3);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Wt + rl * 64 + (((2 * c8) ^ (rl & 15)) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Wt + rl * 64 + (((2 * c8 + 1) ^ (rl & 15)) << 2));
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r] = rowbias ? v[r] + bia[r] : v[r]; v[r] = relu ? fmaxf(v[r], 0.f) : v[r]; }
        unsigned wbits = 0;
        if (wide_bits) {                                 // (shuffles before the divergent `continue`: every lane takes part)
            const unsigned lo_w = (unsigned)__shfl((int)mrow.x, rl, 64), hi_w = (unsigned)__shfl((int)mrow.y, rl, 64);
            wbits = ((c8 < 4 ? lo_w : hi_w) >> (8 * (c8 & 3))) & 0xFFu;
        }
        const int m = mb + rl;
        if (m >= p.M || !ncol) continue;
        if (relu_mask) {
            if (p.mask_in) {
                const unsigned bits = wide_bits ? wbits : (unsigned)__builtin_bit_cast(unsigned short, auxv[ps][0]);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = ((bits >> r) & 1u) ? v[r] : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = ((float)auxv[ps][r] > 0.f) ? v[r] : 0.f;
            }
        }
        if (mse) {
            float t[8];
            const size_t ti = (size_t)m * p.N + n;
            if (p.mse_target_dtype == AFR_TARGET_U8) {
                const uint2 w = tu8[ps];
                if (lut255) {                          // k / 255.0f looked up (the block computed the 256 quotients once): same values
#pragma unroll
                    for (int r = 0; r < 4; ++r) { t[r] = lut255[(w.x >> (8 * r)) & 0xFF]; t[4 + r] = lut255[(w.y >> (8 * r)) & 0xFF]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { t[r] = (float)((w.x >> (8 * r)) & 0xFF) / 255.0f; t[4 + r] = (float)((w.y >> (8 * r)) & 0xFF) / 255.0f; }
                }
            } else {
                const float4 w0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.mse_target) + ti);
                const float4 w1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.mse_target) + ti + 4);
                t[0] = w0.x; t[1] = w0.y; t[2] = w0.z; t[3] = w0.w; t[4] = w1.x; t[5] = w1.y; t[6] = w1.z; t[7] = w1.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float u = out_bf16 ? (float)(bf16_t)v[r] : v[r];     // the value the unfused path would store
                const float diff = fminf(fmaxf(u, 0.f), 1.f) - t[r];
                lsum += diff * diff;
                v[r] = (u >= 0.f && u <= 1.f) ? g2 * diff : 0.f;
            }
        }
        {
            bf16x8 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
            // streaming stores: a kernel's dirty L2 lines are written back at its end, before the next kernel may start
            // (the XCDs' L2s are not coherent with each other); write-through output leaves nothing to drain (C3 -3.7 %)
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(Cb + (size_t)m * p.ldc + n));
            if (p.mask_out) {                          // the ReLU mask of the STORED (bf16) values, one byte per 8 columns
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) bits |= ((float)o[r] > 0.f ? 1u : 0u) << r;
                p.mask_out[(size_t)m * p.ldmask + (n >> 3)] = (unsigned char)bits;
            }
        }
    }
}

template <int ALAY, int BLAY, int WM, bool SCALE = false, bool EARLYB_ = true>
__device__ __forceinline__ void wave_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], const int mb, const int nb0, const int z,
                                              float* Wt, const int lane, float& lsum, const float* lut255 = nullptr) {
    wave_epilogue_at<ALAY, BLAY, WM, SCALE, EARLYB_>(p, [&](int i, int j) { return acc[i][j]; }, mb, nb0, z, Wt, lane, lsum, lut255);
}

// Finish of one wave's 16 x 64 f32 strip of a weight gradient (cooperative split-K, gemm_bf16_256_body): v[j] holds rows
// mb + (lane & 15), columns nb0 + 16 j + 4 (lane >> 4) .. +3.  Through the wave's own LDS (Wt: 16 KiB; the first 4 KiB take
// the strip, XOR-swizzled like wave_epilogue) so that 16 lanes x 16 B cover one 256-byte row segment; then AdamW on
// p/m/v(/shadow) at the same [m][ldc] position (p.ad_p set) or a plain store of the gradient into C.
// strip_prefetch brings the strip's p/m/v into the other 12 KiB of Wt by LDS-DMA (no VGPRs: the kernel has none to spare
// around its K loop) in the layout of the finish -- pass ps covers rows mb + 4 ps + (lane >> 4), columns nb0 + 4 (lane & 15)
// ..+3; lane l's 16 bytes of tensor t land at Wt + 4 KiB + (3 ps + t) KiB + 16 l.  The cooperative tail issues it before it
// parks its accumulators, so the optimizer state arrives under the park / wait / slice-load chain; the caller's
// s_waitcnt vmcnt(0) (the park's drain) retires it before the finish reads it.
__device__ __forceinline__ void strip_prefetch(const GemmParams& p, const int mb, const int nb0, const int lane, float* Wt) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(Wt) + 4096);
    const i32x4 rp = make_rsrc(p.ad_p), rm = make_rsrc(p.ad_m), rv = make_rsrc(p.ad_v);
    const int nf = nb0 + 4 * (lane & 15);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int m = mb + ps * 4 + (lane >> 4);
        // (32-bit byte offsets: the launcher admits the fused optimizer here only for tensors below 2 GiB)
        const unsigned off = (m < p.M && nf < p.N) ? (unsigned)(((size_t)m * p.ldc + nf) * 4) : 0x80000000u;
        dma16(rp, lds + (3 * ps + 0) * 1024, off);
        dma16(rm, lds + (3 * ps + 1) * 1024, off);
        dma16(rv, lds + (3 * ps + 2) * 1024, off);
    }
}
// pre: the strip's p/m/v were prefetched into Wt (strip_prefetch); else they are loaded here
__device__ __forceinline__ void strip_finish(const GemmParams& p, const f32x4 (&v)[4], const int mb, const int nb0, float* Wt, const int lane,
                                             const bool pre) {
    const int ml = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + (lane >> 4);
        *reinterpret_cast<f32x4*>(Wt + ml * 64 + ((c ^ ml) << 2)) = v[j];
    }
    const int c4 = lane & 15, nf = nb0 + 4 * c4;
    const bool okc = nf < p.N;
    float* Cf = reinterpret_cast<float*>(p.C);
    const bool adam = p.ad_p != nullptr;
    const float4* pmv = reinterpret_cast<const float4*>(Wt + 1024) + lane;       // + (3 ps + t) * 64
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int rl = ps * 4 + (lane >> 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(Wt + rl * 64 + ((c4 ^ rl) << 2));
        const int m = mb + rl;
        if (m >= p.M || !okc) continue;
        const size_t wi = (size_t)m * p.ldc + nf;
        if (adam) {
            float4 qp, qm, qv;
            if (pre) { qp = pmv[(3 * ps + 0) * 64]; qm = pmv[(3 * ps + 1) * 64]; qv = pmv[(3 * ps + 2) * 64]; }
            else { qp = ADLD(p.ad_p + wi); qm = ADLD(p.ad_m + wi); qv = ADLD(p.ad_v + wi); }
            float pp[4] = {qp.x, qp.y, qp.z, qp.w}, mm[4] = {qm.x, qm.y, qm.z, qm.w}, vv[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) adamw_elem(pp[r], mm[r], vv[r], g[r], p.ad_decay, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step, p.ad_rsqrt_bc2);
            ADST(p.ad_p + wi, make_float4(pp[0], pp[1], pp[2], pp[3]));
            ADST(p.ad_m + wi, make_float4(mm[0], mm[1], mm[2], mm[3]));
            ADST(p.ad_v + wi, make_float4(vv[0], vv[1], vv[2], vv[3]));
            if (p.ad_shadow) {
                bf16x4 o = {(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(p.ad_shadow + wi));
            }
        } else {
            nt_st4(Cf + wi, make_float4(g[0], g[1], g[2], g[3]));
        }
    }
}

template <int WM> struct RingGeom {
    static constexpr int ASUB = WM / 2, STAGES = (WM == 4) ? 3 : 2, STAGE_BYTES = (ASUB + 1) * SUB, LDS_BYTES = STAGES * STAGE_BYTES;
};
// One output tile (and k-split) of one product: block `bid` of the `nblk` blocks that product was given.  A plain launch
// passes its own blockIdx / gridDim; a grouped launch (gemm_bf16_group) a sub-range of its grid.
// ABL (kernel-development builds only, -DAFR_GEMM_LAB): ablation bits for the ring loop -- 1: no DMA inside the loop,
// 2: no fragment reads inside the loop, 4: no MFMAs.  Results are wrong by construction; only the time is read.
// GA: the k-contiguous A operand's rows are gathered through p.a_rowmap (ALAY == 0, WM == 4 only)
template <int ALAY, int BLAY, int WM, int ABL = 0, int GA = 0>
__device__ __forceinline__ void gemm_bf16_body(const GemmParams& p, const int bid, const int nblk, char* smem) {
    static_assert(!GA || (ALAY == 0 && WM == 4), "row gather: k-contiguous A on the 256x128 ring kernel");
    constexpr int BM = 64 * WM, NW = 2 * WM, ASUB = WM / 2;
    constexpr int STAGES = RingGeom<WM>::STAGES;
    constexpr int STAGE_BYTES = RingGeom<WM>::STAGE_BYTES;
    constexpr int A_PER_WAVE = ASUB * 16 / NW, B_PER_WAVE = 16 / NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#ifdef AFR_GEMM_TIMING
    unsigned long long* stamps = g_gemm_stamps;
    GSTAMP(0);
    if (stamps && tid == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        stamps[GSLOT * 8 + 4] = xcc; stamps[GSLOT * 8 + 5] = hwid;
    }
#endif
    int tm, tn, z;
    tile_of_block(p, BM, BN, bid, nblk, tm, tn, z);
    if (z >= p.splitk) return;                       // padding block of a grouped launch (ranges are rounded up to 8)
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
    const int nt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    // (Measured and dropped: requesting the fused loss' uint8 targets here, ahead of the prologue's DMA, to hold them in 16
    // VGPRs for the epilogue: the prologue got 1.0 us longer, the tail 0.5-1 us shorter.)
    const i32x4 rA = make_rsrc(A), rB = make_rsrc(B);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // Interior blocks (whole tile inside M x N, K range a multiple of 64) address their DMA pieces the FAST way: the
    // lane-dependent part of a piece's source offset depends on the lane and the wave only (for k-strided A images also
    // on whether the piece is in the wave's first or second pair), so it is computed once; which piece and which K-tile
    // are wave-uniform byte offsets passed as the instruction's SGPR soffset -- no per-piece address arithmetic, bounds
    // selects or exec-mask juggling inside the K loop.
    const bool fast = (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((kend - kbeg) % BK == 0) && nt > 0;
    unsigned fvA[2] = {0, 0}, fvB[2] = {0, 0};         // [first | second half of the wave's pieces] (equal unless k-strided)
    {
        const int ia0 = (wave * A_PER_WAVE) & 15, asub = (wave * A_PER_WAVE) >> 4, ib0 = wave * B_PER_WAVE;
        if (ALAY == 0) {
            const int r8 = lane >> 3;
            fvA[0] = fvA[1] = (unsigned)(((size_t)(m0 + asub * 128 + ia0 * 8 + r8) * p.lda + kbeg + 8 * ((lane & 7) ^ (r8 & 7))) * 2);
        } else {
            const int q4 = lane >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int hi = A_PER_WAVE >= 4 ? h : ((ia0 >> 1) & 1);
                const int c = (lane & 15) ^ (((q4 & 3) | (hi << 2)) << 1);
                fvA[h] = (unsigned)(((size_t)(kbeg + ia0 * 4 + q4) * p.lda + m0 + asub * 128 + 8 * c) * 2);
            }
        }
        if (BLAY == 0) {
            const int r8 = lane >> 3;
            fvB[0] = fvB[1] = (unsigned)(((size_t)(n0 + ib0 * 8 + r8) * p.ldb + kbeg + 8 * ((lane & 7) ^ (r8 & 7))) * 2);
        } else {
            const int q4 = lane >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int hi = B_PER_WAVE >= 4 ? h : ((ib0 >> 1) & 1);
                const int c = (lane & 15) ^ (((q4 & 3) | (hi << 2)) << 1);
                fvB[h] = (unsigned)(((size_t)(kbeg + ib0 * 4 + q4) * p.ldb + n0 + 8 * c) * 2);
            }
        }
    }
    // gathered A rows: a piece's 8 rows are no longer 8 * lda apart, so every piece of the wave gets its own lane offset
    // (looked up ONCE: a block's rows are the same for every K-tile) and only the K-tile remains in the SGPR offset
    unsigned fvAq[GA ? A_PER_WAVE : 1];
    if (GA) {
#pragma unroll
        for (int q = 0; q < A_PER_WAVE; ++q) {
            const int g = wave * A_PER_WAVE + q, r8 = lane >> 3;
            const int row = m0 + (g >> 4) * 128 + (g & 15) * 8 + r8;
            const int trow = row < p.M ? p.a_rowmap[row] : 0;
            fvAq[q] = (unsigned)(((size_t)trow * p.lda + kbeg + 8 * ((lane & 7) ^ (r8 & 7))) * 2);
        }
    }
    const unsigned fpA = (ALAY == 0 ? 8u : 4u) * p.lda * 2, fpB = (BLAY == 0 ? 8u : 4u) * p.ldb * 2;     // bytes per piece step
    const unsigned ftA = ALAY == 0 ? 128u : 64u * p.lda * 2, ftB = BLAY == 0 ? 128u : 64u * p.ldb * 2;   // bytes per K-tile
    // (Measured and dropped: the nt bit on the pieces of the operand that streams -- R0's 246 MB weight shadow.  Its tiles are
    // shared by the row tiles running beside each other on the XCD, and marked streaming they leave the L2 before the last of
    // those has read them: forward 283 -> 309 us, input gradient 304 -> 340 us.)
    auto dma_s = [&](i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %4, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(soff), "s"(rsrc) : "memory");
    };
    // piece q of the wave (q < A_PER_WAVE: A pieces, then B pieces) of K-tile t into ring slot `slot`
    auto stage_piece_fast = [&](int t, int slot, int q) {
        const unsigned S = lds0 + slot * STAGE_BYTES;
        if (q < A_PER_WAVE) {
            const int g = wave * A_PER_WAVE + q;
            if (GA) dma_s(rA, S + (g >> 4) * SUB + (g & 15) * 1024, fvAq[q], (unsigned)t * ftA);
            else dma_s(rA, S + (g >> 4) * SUB + (g & 15) * 1024, fvA[A_PER_WAVE >= 4 ? (q >> 1) : 0], (unsigned)t * ftA + (unsigned)q * fpA);
        } else {
            const int i = q - A_PER_WAVE;
            dma_s(rB, S + ASUB * SUB + (wave * B_PER_WAVE + i) * 1024, fvB[B_PER_WAVE >= 4 ? (i >> 1) : 0], (unsigned)t * ftB + (unsigned)i * fpB);
        }
    };
    auto stage = [&](int t, int slot) {
        if (fast) {
#pragma unroll
            for (int q = 0; q < A_PER_WAVE + B_PER_WAVE; ++q) stage_piece_fast(t, slot, q);
            return;
        }
        const unsigned S = lds0 + slot * STAGE_BYTES;
        const int k0 = kbeg + t * BK;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            const int g = wave * A_PER_WAVE + i;
            const int sub = g >> 4;
            stage_inst<ALAY>(rA, S + sub * SUB, p.lda, p.M, m0 + sub * 128, k0, kend, g & 15, lane, GA ? p.a_rowmap : nullptr);
        }
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i)
            stage_inst<BLAY>(rB, S + ASUB * SUB, p.ldb, p.N, n0, k0, kend, wave * B_PER_WAVE + i, lane);
    };

    auto read_a = [&](const char* S, int ks, bf16x8 (&f)[4]) {
        const char* As = S + (wm >> 1) * SUB;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = read_frag<ALAY>(As, (wm & 1) * 64 + 16 * i, ks, lane);
    };
    auto read_b = [&](const char* S, int ks, bf16x8 (&f)[4]) {
        const char* Bs = S + ASUB * SUB;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = read_frag<BLAY>(Bs, wn * 64 + 16 * j, ks, lane);
    };
    // operands swapped on purpose: D'[n][m] so that a lane owns 4 consecutive n of one row m
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#ifndef AFR_ABLATE_NOMFMA
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(fa[i])); asm volatile("" ::"v"(fb[i])); }
#endif
    };
    // k-step-1 MFMAs of tile t with the refill of the freed slot (tile t+3) and the fragment reads of tile t+1 (k-step 0)
    // issued BETWEEN the rows of MFMAs instead of ahead of them: the memory instructions go out while the matrix pipe works.
    auto stage_piece = [&](int t, int slot, int q) {
        if (fast) { stage_piece_fast(t, slot, q); return; }
        const unsigned S = lds0 + slot * STAGE_BYTES;
        const int k0 = kbeg + t * BK;
        if (q < A_PER_WAVE) {
            const int g = wave * A_PER_WAVE + q;
            const int sub = g >> 4;
            stage_inst<ALAY>(rA, S + sub * SUB, p.lda, p.M, m0 + sub * 128, k0, kend, g & 15, lane, GA ? p.a_rowmap : nullptr);
        } else {
            stage_inst<BLAY>(rB, S + ASUB * SUB, p.ldb, p.N, n0, k0, kend, wave * B_PER_WAVE + (q - A_PER_WAVE), lane);
        }
    };
    auto mma_mem = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], bool do_stage, int tn, int slot, bool do_read, const char* Sn,
                       int ks, bf16x8 (&ra)[4], bf16x8 (&rb)[4]) {
        constexpr int NP = A_PER_WAVE + B_PER_WAVE;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!(ABL & 4)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(fa[i]), "v"(fb[i]));
            }
            if (do_stage && !(ABL & 1)) {
#pragma unroll
                for (int q = i * NP / 4; q < (i + 1) * NP / 4; ++q) stage_piece(tn, slot, q);
            }
            if (do_read && !(ABL & 2)) {
                ra[i] = read_frag<ALAY>(Sn + (wm >> 1) * SUB, (wm & 1) * 64 + 16 * i, ks, lane);
                rb[i] = read_frag<BLAY>(Sn + ASUB * SUB, wn * 64 + 16 * i, ks, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // fused bias gradient: column sums of the k-strided A tile [64 k][BM x].  A thread owns one 16-byte chunk (8
    // consecutive x) and walks 4 of the 64 k-rows; 16 threads share a chunk and are combined once after the K loop.
    constexpr int CH = BM / 8;                       // chunks per k-row; NTHR / CH == 16 row groups
    float cs8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cs8[r] = 0.f;
    auto colsum_tile = [&](const char* S) {
        const int c = tid % CH, rg = tid / CH;
        const char* Ac = S + (c >> 4) * SUB;
        const int cl = c & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = rg + 16 * i;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Ac + k * 256 + ((cl ^ (fswz(k) << 1)) << 4));
#pragma unroll
            for (int r = 0; r < 8; ++r) cs8[r] += (float)v[r];
        }
    };

    if (STAGES == 3) {
        // Software-pipelined ring.  Per K-tile t (slot t%3), each wave:
        //   B  runs the MFMAs of k-step 0 (fragments read earlier) and, between its rows of MFMAs, A issues the LDS reads
        //      of k-step 1 of tile t
        //   C  waits until the DMA of tile t+1 has landed (tile t+2 stays in flight: counted vmcnt) and meets the
        //      other waves at the one barrier of the tile; every wave's reads of tile t are complete by then
        //   F  runs the MFMAs of k-step 1 and, between its rows of MFMAs (mma_mem), D re-fills the freed slot with tile
        //      t+3 and E issues the reads of k-step 0 of tile t+1 (+2..7 % over issuing D and E ahead of F)
        // so every MFMA block has the next block's LDS reads in flight under it, and two tiles of DMA are in flight.
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nt) stage(t, t);
        // start as soon as tile 0 has landed: tiles 1 and 2 (the newest 2 x 6 wave-instructions) stay in flight
        if (nt >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (nt == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GSTAMP(1);
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        if (nt > 0) { read_a(smem, 0, a0); read_b(smem, 0, b0); }
        int slot = 0;
        for (int t = 0; t < nt; ++t) {
            const char* S = smem + slot * STAGE_BYTES;
            if (ALAY == 1 && do_cs) colsum_tile(S);
            if ((ABL & 8) && wave >= 4) {
                // waves 4..7 (the SIMD partners of 0..3) re-fill the ring in the FIRST half of an iteration, waves 0..3 in
                // the second: at any time only one of a SIMD's two waves is issuing DMA (an LDS-DMA piece holds its wave for
                // ~100 cycles), the other has bare MFMAs + LDS reads for the matrix pipe.  Slot (t+2)%3 held tile t-1, which
                // every wave finished reading before the barrier of iteration t-1.
                int ps = slot + 2; if (ps >= 3) ps -= 3;
                mma_mem(a0, b0, t >= 1 && t + 2 < nt, t + 2, ps, true, S, 1, a1, b1);
            } else {
                mma_mem(a0, b0, false, 0, 0, true, S, 1, a1, b1);                     // A inside B
            }
            if (t + 1 < nt) {
                if ((ABL & 8) && wave >= 4) {
                    // outstanding here: tile t+1 (6 pieces, issued one iteration ago) and tile t+2 (just issued)
                    if (t >= 1 && t + 2 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    else if (t == 0 && nt > 2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // prologue's tiles 1, 2
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                } else {
                    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // C
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                int ns = slot + 1; if (ns == 3) ns = 0;
                if ((ABL & 8) && wave >= 4) mma_mem(a1, b1, false, 0, 0, true, smem + ns * STAGE_BYTES, 0, a0, b0);
                else mma_mem(a1, b1, t + 3 < nt, t + 3, slot, true, smem + ns * STAGE_BYTES, 0, a0, b0);   // D, E inside F
                slot = ns;
            } else {
                mma(a1, b1);
            }

        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        if (nt > 0) stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GSTAMP(1);
        int slot = 0;
        for (int t = 0; t < nt; ++t) {
            // tile t+1 streams into the other slot, which every wave finished reading before the last barrier
#ifndef AFR_ABLATE_NOLOAD
            if (t + 1 < nt) stage(t + 1, slot ^ 1);
#endif
            const char* S = smem + slot * STAGE_BYTES;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                read_a(S, ks, fa); read_b(S, ks, fb);
                mma(fa, fb);
            }
            if (ALAY == 1 && do_cs) colsum_tile(S);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }

    GSTAMP(2);
    if (ALAY == 1 && do_cs) {
        float* red = reinterpret_cast<float*>(smem);           // [16 row groups][BM]
        const int c = tid % CH, rg = tid / CH;
#pragma unroll
        for (int r = 0; r < 8; ++r) red[rg * BM + 8 * c + r] = cs8[r];
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) a += red[g * BM + tid];
            p.colsum[(size_t)z * p.colsum_stride + m0 + tid] = a;
        }
        __syncthreads();                       // the staging below reuses this LDS
    }
    const bool mse = p.mse_target != nullptr;
    float lsum = 0.f;
    // uint8 targets of the fused loss are pixel / 255.0f (helpers.py:121): 256 true divisions per block instead of one per
    // output element (the 8-wave kernel's LDS has room behind the waves' staging tiles)
    const float* lut255 = nullptr;
    if (WM == 4 && mse && p.mse_target_dtype == AFR_TARGET_U8) {
        float* l = reinterpret_cast<float*>(smem) + NW * 4096;
        if (tid < 256) l[tid] = (float)tid / 255.0f;
        __syncthreads();
        lut255 = l;
    }
    wave_epilogue<ALAY, BLAY, WM>(p, acc, m0 + wm * 64, n0 + wn * 64, z, reinterpret_cast<float*>(smem) + wave * 4096, lane, lsum, lut255);
    if (mse) {
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();                       // every wave is done with its staging tile
        lsum = wave_sum(lsum);
        if (lane == 0) red[wave] = lsum;
        __syncthreads();
        float bs = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) bs += red[w];
        loss_block_finish(bs, p.mse_partial, p.mse_counter, p.mse_loss_accum, p.mse_inv_n, red + 16);
    }
#ifdef AFR_GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GSTAMP(3);
#endif
}

// ------------------------------------------------------------------ 256x256 tile, 8 waves of 128x64, 4 phases per K-tile
// For products whose 256x256 tiling still fills the chip (a layer's dW + dX in one grouped launch: 128 + 128 blocks).
// Against the 256x128 ring kernel a K-tile moves 2/3 of the DMA bytes and 3/4 of the LDS fragment reads per FLOP, and the
// two waves of a SIMD never want the matrix pipe at the same time:
//   * waves as 2 (M) x 4 (N); wave (wr, wc) owns rows wr*128.. (A sub-tile wr) and columns (wc&1)*64.. of B sub-tile
//     wc>>1 (a sub-tile = 128 x 64 operand elements, 16 KiB, in either orientation: the images of stage_inst / read_frag);
//   * a K-tile is four PHASES, one 64x32 quadrant of the wave's accumulators each (16 MFMAs = 256 matrix-pipe cycles):
//       phase 0  reads all of B (8 fragments) + A rows 0..63 (8)   -> quadrant (0,0)      B sub-tiles free after this
//       phase 1  (no reads)                                         -> quadrant (0,1)
//       phase 2  reads A rows 64..127 (8)                           -> quadrant (1,1)      A sub-tiles free after this
//       phase 3  (no reads)                                         -> quadrant (1,0)
//     a phase = { issue its LDS reads; move one sub-tile of tile t+2 along; barrier; lgkmcnt(0); 16 MFMAs; barrier }.
//     Waves 4..7 (wr = 1, the SIMD partners of waves 0..3) run ONE BARRIER BEHIND: while one group is in its MFMAs the
//     other does the memory work (an LDS-DMA piece holds its wave ~100 cycles: issued behind a wave's own MFMAs it
//     lengthened every phase by that much), then they swap;
//   * a CU takes LDS-DMA pieces at only ~1 KiB per 38 cycles (~30 B/clk): the 64 pieces of a 256x256x64 K-tile are 1.3 us
//     against 1.03 us of MFMAs, so this kernel is DMA-bound (measured 1.5 us per K-tile; with the DMA removed 1.03), as is
//     the 256x128 ring kernel (48 pieces: 0.88 us against 0.55 us of MFMAs) -- per FLOP 1.35x faster here.  Measured
//     alternatives: every operand piece through registers (buffer_load -> ds_write_b128) saturates the LDS store path
//     instead (1.7-1.9 us); A by DMA + B through registers spills (1.6 us);
//   * LDS, all 160 KiB: A sub-tiles in a ring of 3 stages, B sub-tiles in 2.  Phase q of tile t issues sub-tile q of tile
//     t+2 (A0, A1, B0, B1): the A stage of tile t+2 was last read in phase 2 of tile t-1, the B stage in phase 0 of tile t
//     -- at least two phases before it is overwritten, so also the trailing group's reads are retired.  ONE counted wait
//     per K-tile, in phase 3 ahead of its first barrier: the four newest sub-tiles (tile t+2) stay in flight, tile t+1 has
//     landed; its first read is two barriers later (the trailing group's wait sits one barrier after the leading one's).
// GB: the k-strided B operand's rows (its k index: batch rows of a weight-gradient product) are gathered through p.b_rowmap
template <int ALAY, int BLAY, int ABL = 0, int GB = 0>
__device__ __forceinline__ void gemm_bf16_256_body(const GemmParams& p, const int bid, const int nblk, char* smem) {
    static_assert(!GB || BLAY == 1, "row gather: k-strided B on the 256x256 kernel");
    constexpr int BM = 256, BNN = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int tm, tn, z, ksplit = p.splitk;
    int tail_idx = -1;                                  // >= 0: a K-slice of tail tile tail_idx, reduced in-kernel (fix_ws)
    if (p.fix_ws) {
        // Split-K with the reduction INSIDE the launch: the first head_tiles tiles of the walk are computed whole by one
        // block each (typically the full rounds of the chip); each remaining ("tail") tile is cut into splitk K-slices so
        // that the last, partial round is spread over all CUs.  A slice block parks its accumulators in fix_ws (a private
        // register-image layout: every store and load is a whole 1 KiB wave access) and takes a ticket; the block that
        // draws the tile's last ticket adds the slices in slice order -- whoever that block is, so results do not depend
        // on arrival order -- and runs the normal epilogue (bias / ReLU / mask / bf16).  Nobody waits for anybody.
        const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BNN - 1) / BNN);
        const int head = min(p.head_tiles, tiles), ntail = tiles - head;
        int t;
        if (bid < head) { t = xcd_walk(bid, head); z = 0; ksplit = 1; }
        else {
            const int b2 = bid - head, nb2 = ntail * p.splitk;
            if (b2 >= nb2) return;
            const int v2 = xcd_walk(b2, nb2);
            z = v2 / ntail;
            tail_idx = v2 - z * ntail;
            t = head + tail_idx;
        }
        tile_coords(p, BM, BNN, t, tiles >> 3, tiles & 7, tm, tn);
    } else {
        tile_of_block(p, BM, BNN, bid, nblk, tm, tn, z);
        if (z >= p.splitk) return;
    }
#ifdef AFR_GEMM_TIMING
    unsigned long long* stamps = g_gemm_stamps;
    GSTAMP(0);
    if (stamps && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[GSLOT * 8 + 4] = xcc; stamps[GSLOT * 8 + 5] = (unsigned)(p.coop_ws ? 1 : 0);
    }
#endif
    const int m0 = tm * BM, n0 = tn * BNN;
    const int klen = ((p.K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    const int kbeg = z * klen;
    const int kend = min(p.K, kbeg + klen);
    const int nt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    const i32x4 rA = make_rsrc(p.A), rB = make_rsrc(p.B);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS map: A ring [3][A0, A1] at 0, B ring [2][B0, B1] at 6 * SUB.  sub-tile q of K-tile t: q = 0, 1 -> A0, A1 into A stage
    // sa; q = 2, 3 -> B0, B1 into B stage t & 1.  2 pieces per wave.
    // Interior blocks (whole tile inside M x N, K range a multiple of 64) take the FAST form: the lane-dependent part of a
    // piece's source address is the same for every piece a wave ever issues for an operand (the row-in-piece / chunk
    // swizzle of stage_inst depends on the lane and, for k-strided images, on the wave's parity only), so it is ONE VGPR
    // per operand computed here; which piece, which sub-tile and which K-tile only add a wave-uniform byte offset, passed
    // as the instruction's SGPR soffset.  Both pieces of a sub-tile go out in one asm statement (one M0 save/restore).
    const bool fast = (m0 + BM <= p.M) && (n0 + BNN <= p.N) && ((kend - kbeg) % BK == 0);
    unsigned voffA, voffB;
    {
        const int w1 = wave & 1;
        if (ALAY == 0) { const int r = lane >> 3; voffA = (unsigned)(((size_t)(m0 + r) * p.lda + kbeg + 8 * ((lane & 7) ^ (r & 7))) * 2); }
        else { const int kr = lane >> 4, c = (lane & 15) ^ ((((kr & 3) | (w1 << 2))) << 1); voffA = (unsigned)(((size_t)(kbeg + kr) * p.lda + m0 + 8 * c) * 2); }
        if (BLAY == 0) { const int r = lane >> 3; voffB = (unsigned)(((size_t)(n0 + r) * p.ldb + kbeg + 8 * ((lane & 7) ^ (r & 7))) * 2); }
        else { const int kr = lane >> 4, c = (lane & 15) ^ ((((kr & 3) | (w1 << 2))) << 1); voffB = (unsigned)(((size_t)(GB ? 0 : kbeg + kr) * p.ldb + n0 + 8 * c) * 2); }
    }
    // Gathered B rows (GB): k-row kbeg + 64 t + 4 (2 wave + i) + (lane >> 4) of piece i is row b_rowmap[..] of the table.  The
    // wave's two pieces cover 8 consecutive k-rows, so their table rows are ONE scalar load of 8 indices per K-tile (issued in
    // phase 0 for tile t+2, behind that phase's own lgkmcnt(0) by phase 1, where each lane picks its row: 2 VGPRs per tile).
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    unsigned gvb[2] = {0u, 0u};
    // (the base pointer is formed ONCE: re-reading p.b_rowmap from the kernel arguments inside the loop puts a scalar-load
    // wait, i.e. a wait for the phase's LDS reads too, in front of the phase's DMA issue)
    const __attribute__((address_space(4))) int* bm_base = nullptr;
    if (GB) {
        const unsigned long long a0 = (unsigned long long)(p.b_rowmap + kbeg + wave * 8);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a0), hi = __builtin_amdgcn_readfirstlane((unsigned)(a0 >> 32));
        bm_base = (const __attribute__((address_space(4))) int*)(((unsigned long long)hi << 32) | lo);
    }
    auto bmap_load = [&](int t) -> i32x8 {
#ifdef AFR_FAKE_BMAP      // kernel-development build: arithmetic rows instead of the scalar load (wrong results, timing only)
        const int k = kbeg + t * BK + wave * 8;
        return (i32x8){k & 255, (k + 1) & 255, (k + 2) & 255, (k + 3) & 255, (k + 4) & 255, (k + 5) & 255, (k + 6) & 255, (k + 7) & 255};
#else
        return *reinterpret_cast<const __attribute__((address_space(4))) i32x8*>(bm_base + t * BK);
#endif
    };
    auto bmap_apply = [&](const i32x8 mv) {
        const int kr = lane >> 4;
        const int r0 = kr == 0 ? mv[0] : kr == 1 ? mv[1] : kr == 2 ? mv[2] : mv[3];
        const int r1 = kr == 0 ? mv[4] : kr == 1 ? mv[5] : kr == 2 ? mv[6] : mv[7];
        gvb[0] = voffB + (unsigned)r0 * ((unsigned)p.ldb * 2u);
        gvb[1] = voffB + (unsigned)r1 * ((unsigned)p.ldb * 2u);
    };
    auto dma_pair2 = [&](i32x4 rsrc, unsigned lds_a, unsigned voff_a, unsigned voff_b, unsigned soff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %5, %4 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_a), "v"(voff_b), "s"(lds_a), "s"(soff), "s"(rsrc) : "memory");
    };
    // byte offsets (wave-uniform): per piece row-block, per second sub-tile (+128 rows / +128 elements), per K-tile
    const unsigned pieceA = (ALAY == 0 ? 8u : 4u) * p.lda * 2, pieceB = (BLAY == 0 ? 8u : 4u) * p.ldb * 2;
    const unsigned subA = ALAY == 0 ? 128u * p.lda * 2 : 256u, subB = BLAY == 0 ? 128u * p.ldb * 2 : 256u;
    const unsigned tileA = ALAY == 0 ? 128u : 64u * p.lda * 2, tileB = BLAY == 0 ? 128u : 64u * p.ldb * 2;
    auto dma_pair = [&](i32x4 rsrc, unsigned lds_a, unsigned voff, unsigned soff_a, unsigned soff_b) {
        if constexpr (ABL & 4) {
            i32x4 d0, d1;
            asm volatile("s_nop 3\n\tbuffer_load_dwordx4 %0, %2, %5, %3 offen\n\tbuffer_load_dwordx4 %1, %2, %5, %4 offen"
                         : "=&v"(d0), "=&v"(d1) : "v"(voff), "s"(soff_a), "s"(soff_b), "s"(rsrc) : "memory");
            return;
        }
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %5, %3 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_a), "s"(soff_a), "s"(soff_b), "s"(rsrc) : "memory");
    };
    auto stage_sub = [&](int t, int sa, int q) {
        if (fast) {
            const unsigned i0 = wave * 2;
            if (q < 2) {
                const unsigned so = (unsigned)t * tileA + (unsigned)q * subA + i0 * pieceA;
                dma_pair(rA, lds0 + (sa * 2 + q) * SUB + i0 * 1024, voffA, so, so + pieceA);
            } else if (GB) {
                dma_pair2(rB, lds0 + (6 + (t & 1) * 2 + (q - 2)) * SUB + i0 * 1024, gvb[0], gvb[1], (unsigned)(q - 2) * subB);
            } else {
                const unsigned so = (unsigned)t * tileB + (unsigned)(q - 2) * subB + i0 * pieceB;
                dma_pair(rB, lds0 + (6 + (t & 1) * 2 + (q - 2)) * SUB + i0 * 1024, voffB, so, so + pieceB);
            }
            return;
        }
        const int k0 = kbeg + t * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int inst = wave * 2 + i;
            if (q < 2) stage_inst<ALAY>(rA, lds0 + (sa * 2 + q) * SUB, p.lda, p.M, m0 + q * 128, k0, kend, inst, lane);
            else stage_inst<BLAY>(rB, lds0 + (6 + (t & 1) * 2 + (q - 2)) * SUB, p.ldb, p.N, n0 + (q - 2) * 128, k0, kend, inst, lane, GB ? p.b_rowmap : nullptr);
        }
    };
    const bool do_cs = (ALAY == 1) && p.colsum != nullptr && tn == 0;
    constexpr int CH = BM / 8;
    float cs8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cs8[r] = 0.f;
    auto colsum_tile = [&](const char* S) {
        const int c = tid % CH, rg = tid / CH;
        const char* Ac = S + (c >> 4) * SUB;        // S = this tile's A stage: A0, A1 are adjacent
        const int cl = c & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = rg + 16 * i;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Ac + k * 256 + ((cl ^ (fswz(k) << 1)) << 4));
#pragma unroll
            for (int r = 0; r < 8; ++r) cs8[r] += (float)v[r];
        }
    };

    // prologue: tiles 0 and 1 (what the phases of "tiles -2, -1" would have issued); start when tile 0 has landed
    if (nt > 0) { if (GB && fast) bmap_apply(bmap_load(0)); stage_sub(0, 0, 0); stage_sub(0, 0, 1); stage_sub(0, 0, 2); stage_sub(0, 0, 3); }
    if (nt > 1) { if (GB && fast) bmap_apply(bmap_load(1)); stage_sub(1, 1, 0); stage_sub(1, 1, 1); stage_sub(1, 1, 2); stage_sub(1, 1, 3); }
    if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile 0 has landed; tile 1 is retired by the wait of phase 3
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GSTAMP(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4..7 run one barrier behind from here on

    bf16x8 fa[2][4], fb[2][4];                          // [k-step][fragment]: A half (64 rows), all 64 B columns
    int sa = 0;                                         // A ring stage of tile t; tile t+2 goes to stage sa + 2 (mod 3)
    i32x8 mvn = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < nt; ++t) {
        const char* As = smem + (sa * 2 + wr) * SUB;
        const char* Bs = smem + (6 + (t & 1) * 2 + (wc >> 1)) * SUB;
        const int bx = (wc & 1) * 64;
        int sa2 = sa + 2; if (sa2 >= 3) sa2 -= 3;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            // ---- this phase's LDS reads
            if (ph == 0 && !((ABL & 2) && t > 0)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[ks][j] = read_frag<BLAY>(Bs, bx + 16 * j, ks, lane);
            }
            if ((ph == 0 || ph == 2) && !((ABL & 2) && t > 0)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[ks][i] = read_frag<ALAY>(As, (ph >> 1) * 64 + 16 * i, ks, lane);
            }
            if (ALAY == 1 && do_cs && ph == 1) colsum_tile(smem + sa * 2 * SUB);
            if (GB && fast && t + 2 < nt) {
                if (ph == 0) mvn = bmap_load(t + 2);
                if (ph == 1) bmap_apply(mvn);
            }
            // ---- this phase's DMA: sub-tile ph of tile t+2
            if (t + 2 < nt && !(ABL & 1)) stage_sub(t + 2, sa2, ph);
            if (ph == 3) {
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // tile t+2 (8 pieces per wave) stays in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            {
                const int ia = (ph >> 1) * 4;                        // accumulator rows: phases 0, 1 -> 0..3; phases 2, 3 -> 4..7
                const int jb = (ph == 1 || ph == 2) ? 2 : 0;         // B columns 32..63 in phases 1, 2; 0..31 in phases 0, 3
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ia + i][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][jb + j], fa[ks][i], acc[ia + i][jb + j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        sa = sa + 1; if (sa == 3) sa = 0;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // the leading group meets the trailing group's last barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GSTAMP(2);

    if (ALAY == 1 && do_cs) {
        float* red = reinterpret_cast<float*>(smem);           // [16 row groups][BM]
        const int c = tid % CH, rg = tid / CH;
#pragma unroll
        for (int r = 0; r < 8; ++r) red[rg * BM + 8 * c + r] = cs8[r];
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) a += red[g * BM + tid];
            p.colsum[(size_t)z * p.colsum_stride + m0 + tid] = a;
        }
        __syncthreads();
    }
    if ((ALAY == 1 && BLAY == 1) && p.coop_ws) {      // (weight-gradient products only: both operands k-strided)
        // Cooperative split-K (GemmParams::coop_ws): the S = splitk slice workgroups of this tile exchange their partial
        // sums inside the launch.  Hand-off form (MI355X_MICROARCH.md, visibility table, first row): every payload byte is
        // stored write-through (sc1) and drained by its wave, the workgroup meets at a barrier, ONE lane adds to the tile's
        // counter; wave 0 polls that counter with sc1 loads (bounded: a missing partner sets AFR_ERR_COOP_TIMEOUT in the
        // error word instead of hanging the GPU), the other waves pass the next barrier behind it, and every load of the
        // parked bytes is an sc1 load -- no fence on either side.  A grouped launch has at most one workgroup per CU (160
        // KiB of LDS) and these come first in its grid, so all S partners are resident whenever >= their number of CUs is
        // free; the input-gradient workgroups behind them never wait for anybody.
        // Parking layout (a register image, every access a whole 1 KiB wave access): slice z of tile T holds, per wave w and
        // accumulator (i, j), 64 lanes x 16 B at (((T*S + z)*8 + w)*32 + 4 i + j) KiB.  Slice z then owns accumulator rows
        // i in [z, z+1) * 8/S of every wave and adds them over the slices IN SLICE ORDER: the sums do not depend on arrival.
        const int S = p.splitk, per = 8 / S;
        const int T = tm * ((p.N + BNN - 1) / BNN) + tn;
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.coop_ws, 0, 0x7FFFFFFF, 0x00020000);
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const unsigned tile_b = (unsigned)T * (unsigned)S * (8u * 32u * 1024u);          // < 2 GiB: checked by the launcher
        float* Wt = reinterpret_cast<float*>(smem) + wave * 4096;
        {
            const unsigned mine = tile_b + ((unsigned)z * 8u + (unsigned)wave) * (32u * 1024u) + (unsigned)lane * 16u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rws, mine + (unsigned)(i * 4 + j) * 1024u, 0, 16);
        }
        // (Measured and dropped: requesting the first strip's optimizer state by LDS-DMA before / behind the park's stores, so
        // that it arrives under the park-wait-load chain -- strip_prefetch.  The finish got 4 us shorter, the park and the
        // wait 7 us longer: this tail is bound by the bytes the whole chip moves at that moment, 32 MB of slices out and in
        // next to the input-gradient workgroups' 17-34 MB, not by the latency of any one request.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its own stores
        __syncthreads();
        GSTAMP(6);
        if (wave == 0) {
            if (lane == 0) (void)__hip_atomic_fetch_add(p.coop_cnt + T, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(p.coop_cnt + T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.coop_target) < 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22)) {                              // seconds: a partner never ran
                    if (lane == 0 && p.err) atomicOr(p.err, AFR_ERR_COOP_TIMEOUT);
                    break;
                }
            }
        }
        __syncthreads();
        GSTAMP(7);
        for (int ii = 0; ii < per; ++ii) {
            const int i = z * per + ii;
            // two rounds of 4 slices (16 loads, 64 VGPRs in flight per lane; more made the kernel spill inside its K loop);
            // slices >= S read past the descriptor (zeros): one code path for every S, and the sum starts from +0 exactly as
            // the slab reduction's does (bitwise equal results).  The range check looks at the VGPR offset only.
            const unsigned src = tile_b + (unsigned)wave * (32u * 1024u) + (unsigned)(i * 4) * 1024u + (unsigned)lane * 16u;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 q[4][4];
#pragma unroll
                for (int zq = 0; zq < 4; ++zq) {
                    const int zz = 4 * h + zq;
                    const unsigned vo = zz < S ? src + (unsigned)zz * (8u * 32u * 1024u) : 0x80000000u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        q[zq][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, vo, (unsigned)j * 1024u, 16));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int zq = 0; zq < 4; ++zq) v[j] += q[zq][j];
            }
            strip_finish(p, v, m0 + wr * 128 + 16 * i, n0 + wc * 64, Wt, lane, false);
        }
#ifdef AFR_GEMM_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GSTAMP(3);
#endif
        return;
    }
    if (tail_idx >= 0) {
        // park this K-slice, take a ticket; only the tile's last arrival goes on (see the mapping above)
        f32x4* ws = reinterpret_cast<f32x4*>(p.fix_ws) + ((size_t)tail_idx * p.splitk * 8 + wave) * 2048 + lane;
        {
            f32x4* mine = ws + (size_t)z * 8 * 2048;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[(i * 4 + j) * 64] = acc[i][j];
        }
        __threadfence();                                 // every wave: its stores are visible device-wide before the ticket
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *flag = atomicAdd(p.fix_cnt + tail_idx, 1u);
        __syncthreads();
        const unsigned ticket = *reinterpret_cast<volatile unsigned*>(flag);
        __syncthreads();                                 // the flag word is part of wave 0's epilogue staging area
        if (ticket != (unsigned)p.splitk - 1u) return;
        __threadfence();                                 // acquire: the other slices' stores (other CUs, other XCDs' L2s)
        if (tid == 0) p.fix_cnt[tail_idx] = 0u;          // re-armed for the next launch (everyone has arrived)
        for (int zz = 0; zz < p.splitk; ++zz) {
            const f32x4* src = ws + (size_t)zz * 8 * 2048;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = src[(i * 4 + j) * 64];
                    acc[i][j] = zz == 0 ? v : acc[i][j] + v;
                }
        }
        z = 0;
    }
    // two passes of the 64x64 wave epilogue: accumulator rows 0..3 (tile rows wr*128 .. +63), then 4..7
    float lsum = 0.f;
    float* Wt = reinterpret_cast<float*>(smem) + wave * 4096;
    // (a real loop over the halves -- two inlined copies of the epilogue break the DMA asm's scalar operands -- with the half
    // picked by selects, never by a dynamic register index, which would be scratch)
    float bia[8];
    if (ALAY == 0 && BLAY == 0) epilogue_bias(p, n0 + wc * 64, lane, bia);       // one request for both halves, ahead of the first park
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wave_epilogue_at<ALAY, BLAY, 4, false, true>(p, [&](int i, int j) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = h ? acc[4 + i][j][r] : acc[i][j][r];
            return v; }, m0 + wr * 128 + h * 64, n0 + wc * 64, z, Wt, lane, lsum, nullptr, (ALAY == 0 && BLAY == 0) ? &bia : nullptr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS reads of pass 0 are done before pass 1 overwrites Wt
    }
#ifdef AFR_GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GSTAMP(3);
#endif
}

template <int ALAY, int BLAY, int WM, int ABL = 0, int GA = 0>
__global__ __launch_bounds__(128 * WM, 2) void gemm_bf16(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[RingGeom<WM>::LDS_BYTES];
    gemm_bf16_body<ALAY, BLAY, WM, ABL, GA>(p, blockIdx.x, gridDim.x, smem);
}

// Several independent products in ONE launch (a layer's weight gradient and input gradient both consume the same dy):
// blocks [blk0[i], blk0[i+1]) work on product i.  The long-running products are listed first (the dispatcher hands out
// blocks in order, so a CU that drew a split-K weight-gradient block of 2x the K-tiles is balanced by one that draws two
// input-gradient blocks), and nothing waits for a device-wide kernel boundary in between: a CU's next block starts while
// other CUs are still in their epilogues.  Products with a fused loss epilogue cannot be grouped (its arrival counter
// counts the blocks of ONE launch).
struct GemmGroup { int n; int blk0[5]; GemmParams p[4]; };
__global__ __launch_bounds__(512, 2) void gemm_bf16_group(GemmGroup g) {
    __shared__ __attribute__((aligned(16))) char smem[RingGeom<4>::LDS_BYTES];
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && b >= g.blk0[i + 1]) ++i;
    const GemmParams& p = g.p[i];
    const int bid = b - g.blk0[i], nblk = g.blk0[i + 1] - g.blk0[i];
    const int lay = ((p.flags & AFR_GEMM_A_KSTRIDED) ? 2 : 0) | ((p.flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0);
    if (lay == 3) gemm_bf16_body<1, 1, 4>(p, bid, nblk, smem);
    else if (lay == 1) gemm_bf16_body<0, 1, 4>(p, bid, nblk, smem);
    else if (lay == 0) gemm_bf16_body<0, 0, 4>(p, bid, nblk, smem);
    else gemm_bf16_body<1, 0, 4>(p, bid, nblk, smem);
}
__global__ __launch_bounds__(512, 2) void gemm_bf16_group256(GemmGroup g) {
    __shared__ __attribute__((aligned(16))) char smem[10 * SUB];   // all 160 KiB: A ring 3 x 32 KiB + B ring 2 x 32 KiB
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && b >= g.blk0[i + 1]) ++i;
    const GemmParams& p = g.p[i];
    const int bid = b - g.blk0[i], nblk = g.blk0[i + 1] - g.blk0[i];
    const int lay = ((p.flags & AFR_GEMM_A_KSTRIDED) ? 2 : 0) | ((p.flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0);
    if (lay == 3 && p.b_rowmap) gemm_bf16_256_body<1, 1, 0, 1>(p, bid, nblk, smem);
    else if (lay == 3) gemm_bf16_256_body<1, 1>(p, bid, nblk, smem);
    else if (lay == 1) gemm_bf16_256_body<0, 1>(p, bid, nblk, smem);
    else if (lay == 0) gemm_bf16_256_body<0, 0>(p, bid, nblk, smem);
    else gemm_bf16_256_body<1, 0>(p, bid, nblk, smem);
}
#ifdef AFR_GEMM_LAB
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_bf16_lab256(GemmGroup g) {
    __shared__ __attribute__((aligned(16))) char smem[10 * SUB];
    gemm_bf16_256_body<0, 0, ABL>(g.p[0], blockIdx.x, gridDim.x, smem);
}
#endif

// ------------------------------------------------------------ folded first layer of the glyph nets: backward in ONE kernel
// (throughput mode).  Inputs: d1 = d(loss)/d(pre-activation of fc1) [B][N1] (the layer above already applied the ReLU mask),
// h0 = Emb[x] + Font[f] [B][E] (the first E columns of the forward's h0'), the codes, and W1^T (bf16 [E][N1], left behind by
// the forward's table kernel).  A workgroup owns 64 glyphs, stages their d1 rows ONCE (128 KiB: the [k = glyph][x = n] images
// of the GEMM's k-strided operand, so both the row-contiguous and the hardware-transposed reads below hit the same bytes)
// and computes, all on the matrix cores:
//     dh0[b][e]  = sum_n d1[b][n] W1[n][e]                       (A rows b from LDS, B rows e of W1^T from L2)
//     dW1[n][e]  = sum_b d1[b][n] h0[b][e],  db1[n] = sum_b d1[b][n]      (h0 image carries a column of ones at x = E)
//     dTab[v][e] = sum_b onehot[b][v] dh0[b][e]                   (nn.Embedding's scatter-add as a product; model.py:309)
// and leaves them in its slab [dW1 | db1 | dTab]; the grouped reduce adds the slabs in block order.  This replaces a
// weight-gradient GEMM against the 168-column widened operand (2.8 GFLOP, 14.5 us of mostly fixed cost, 22 MB of split-K
// slabs) and the kernel that post-processed its slabs (9 us).
struct L1BwdArgs {
    const bf16_t* d1; const bf16_t* h0; const bf16_t* W1T; const int64_t* x; const int64_t* font;
    const int* h0_rowmap;            // optional: glyph b's h0 row is row h0_rowmap[b] of the table h0 (combination table)
    int ldd, ldh, B, N1, vocab, n_fonts;
    int CS, ncols;                   // the N1 columns are cut into CS ranges of ncols (a multiple of 128): block = (row block, range)
    float* slabs; long long slab_stride; int o_b, o_tab;
};
constexpr int L1_E = 32, L1_R = 64, L1_LDR = L1_R + 8;       // embedding width (compile time), glyphs per block, padded row of the overlay
__global__ __launch_bounds__(512) void glyph_l1_bwd_fused_kernel(L1BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int NS = a.ncols >> 7;                             // 128-column sub-tiles of this block's d1 columns
    const int rb = blockIdx.x / a.CS, n_lo = (blockIdx.x - rb * a.CS) * a.ncols;
    const int b0 = rb * L1_R, nb = min(L1_R, a.B - b0);
    char* Dimg = sm;                                         // [NS] images [64 k][128 x], 16 KiB each
    char* Himg = sm + NS * SUB;                              // h0 image (x < E valid, x == E: ones)
    int* ids = reinterpret_cast<int*>(sm + (NS + 1) * SUB);  // [64] codes, [64] vocab + font id (or -1)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#ifdef AFR_L1F_DEBUG
    unsigned long long ts_[8]; int nts_ = 0;
#define L1STAMP() do { __syncthreads(); ts_[nts_++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define L1STAMP() do { } while (0)
#endif
    // this wave's W1^T operands of the dh0 product (16 bytes per lane and k-step, <= 32 k-steps): independent of the staging
    // below, so they are requested first and arrive under it
    constexpr int L1_MAXKS = 32;
    bf16x8 wv[L1_MAXKS];
    {
        const int erow = (wave & 1) * 16 + r;
        const bf16_t* wrow = a.W1T + (size_t)erow * a.N1 + n_lo + 8 * q;
        const int nks = a.ncols >> 5;
#pragma unroll
        for (int j = 0; j < L1_MAXKS; ++j)
            if (j < nks) wv[j] = *reinterpret_cast<const bf16x8*>(wrow + j * 32);
    }
    L1STAMP();
    {
        const i32x4 rD = make_rsrc(a.d1), rH = make_rsrc(a.h0);
        const int pieces = (NS + 1) * 16;
        for (int pc = wave; pc < pieces; pc += 8) {
            const int sidx = pc >> 4, inst = pc & 15;
            if (sidx < NS) stage_inst<1>(rD, lds0 + sidx * SUB, a.ldd, a.N1, n_lo + sidx * 128, b0, b0 + nb, inst, lane);
            else stage_inst<1>(rH, lds0 + NS * SUB, a.ldh, L1_E, 0, b0, b0 + nb, inst, lane, a.h0_rowmap);
        }
        if (tid < L1_R) {
            long long xi = tid < nb ? a.x[b0 + tid] : 0, fi = (tid < nb && a.n_fonts > 0 && a.font) ? a.font[b0 + tid] : 0;
            xi = min(max(xi, 0ll), (long long)a.vocab - 1);                  // out-of-range codes were flagged by the forward
            if (a.n_fonts > 0) fi = min(max(fi, 0ll), (long long)a.n_fonts - 1);
            ids[tid] = (int)xi; ids[L1_R + tid] = a.n_fonts > 0 ? a.vocab + (int)fi : -1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < nb) {                                      // the ones column: element (k = tid, x = E) of the h0 image
            const int k = tid, xx = L1_E;
            *reinterpret_cast<unsigned short*>(Himg + k * 256 + (((xx >> 4) ^ fswz(k)) << 5) + (xx & 15) * 2) = 0x3F80;   // bf16 1.0
        }
        __syncthreads();
    }
    L1STAMP();
    // ---- dh0 tile of this wave: glyph rows (wave>>1)*16.., embedding columns (wave&1)*16..; reduction over all N1
    f32x4 dh0 = {0.f, 0.f, 0.f, 0.f};
    {
        const int brow = (wave >> 1) * 16 + r;
        const int nks = a.ncols >> 5;
#pragma unroll
        for (int j = 0; j < L1_MAXKS; ++j) {
            if (j < nks) {
                const int n = j * 32 + 8 * q, xx = n & 127;
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(Dimg + (n >> 7) * SUB + brow * 256 + (((xx >> 4) ^ fswz(brow)) << 5) + (xx & 15) * 2);
                dh0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv[j], dh0, 0, 0, 0);
            }
        }
    }
    L1STAMP();
    // ---- dW1 / db1: output tiles [16 n][48 e'] (e' = 32: the ones column), reduction over the block's 64 glyphs
    {
        bf16x8 hf[3][2];
#pragma unroll
        for (int et = 0; et < 3; ++et)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) hf[et][ks] = read_frag<1>(Himg, et * 16, ks, lane);
        for (int nt = wave; nt < (a.ncols >> 4); nt += 8) {
            const char* S = Dimg + (nt >> 3) * SUB;
            const bf16x8 d0 = read_frag<1>(S, (nt & 7) * 16, 0, lane), d1v = read_frag<1>(S, (nt & 7) * 16, 1, lane);
            f32x4 acc[3];
#pragma unroll
            for (int et = 0; et < 3; ++et) {
                acc[et] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[et][0], d0, acc[et], 0, 0, 0);    // D[e' = 4q + i][n = r]
                acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[et][1], d1v, acc[et], 0, 0, 0);
            }
            const int n = nt * 16 + r;                       // within this block's column range
            *reinterpret_cast<f32x4*>(slab + (size_t)n * L1_E + 4 * q) = acc[0];
            *reinterpret_cast<f32x4*>(slab + (size_t)n * L1_E + 16 + 4 * q) = acc[1];
            if (q == 0) slab[a.o_b + n] = acc[2][0];
        }
    }
    L1STAMP();
    __syncthreads();                                         // every read of the d1 / h0 images is done: the area is reused
    // ---- embedding_dense_backward as one more product (as glyph1_step_kernel): dh0^T [E][64] and the one-hot image [VT][64]
    bf16_t* dh0T = reinterpret_cast<bf16_t*>(sm);
    bf16_t* ohT = dh0T + L1_E * L1_LDR;
    const int rows_tot = a.vocab + a.n_fonts, VT = (rows_tot + 15) & ~15;
    for (int i = tid; i < VT * L1_LDR * 2 / 16; i += 512) reinterpret_cast<i32x4*>(ohT)[i] = (i32x4){0, 0, 0, 0};
    {
        bf16_t* d = dh0T + ((wave & 1) * 16 + r) * L1_LDR + (wave >> 1) * 16 + 4 * q;       // D[b = 4q + i][e = r]: 4 consecutive b
        *reinterpret_cast<bf16x4*>(d) = (bf16x4){(bf16_t)dh0[0], (bf16_t)dh0[1], (bf16_t)dh0[2], (bf16_t)dh0[3]};
    }
    __syncthreads();
    if (tid < nb) {
        ohT[ids[tid] * L1_LDR + tid] = (bf16_t)1.f;
        if (ids[L1_R + tid] >= 0) ohT[ids[L1_R + tid] * L1_LDR + tid] = (bf16_t)1.f;
    }
    __syncthreads();
    for (int t = wave; t < (VT >> 4) * 2; t += 8) {
        const int vt = t >> 1, et = t & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(dh0T + (et * 16 + r) * L1_LDR + ks * 32 + 8 * q);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(ohT + (vt * 16 + r) * L1_LDR + ks * 32 + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);            // D[e = 4q + i][v = r]
        }
        const int v = vt * 16 + r;
        if (v < rows_tot) *reinterpret_cast<f32x4*>(slab + a.o_tab + (size_t)v * L1_E + et * 16 + 4 * q) = acc;
    }
    L1STAMP();
#ifdef AFR_L1F_DEBUG
    if (blockIdx.x == 5 && tid == 0) { printf("l1f phases (us):"); for (int i = 1; i < nts_; ++i) printf(" %.2f", (double)(ts_[i] - ts_[i - 1]) * 0.01); printf("\n"); }
#endif
}
}  // namespace bf16k

// ------------------------------------------------------------------------------------------ fp8
// C[m][n] = scale * sum_k A(m,k) B(n,k) (+bias) (relu), A and B OCP e4m3 bytes, both k-contiguous (the forward form x . W^T),
// f32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: the MX-scaled instruction is the one that runs
// fp8 at twice the bf16 rate (5 PF dense; the plain fp8 MFMAs run at the bf16 rate) -- `scale` is ONE per-tensor factor
// (scale_a * scale_b) applied in the epilogue.  First building block of BASELINE configs[4] (fp8 weights on CDNA4).
// Geometry = the 256x128 ring kernel's: 8 waves of 64x64, three LDS stages of 3 x 16 KiB filled by LDS-DMA; a sub-tile row is
// 128 BYTES = 128 k here (64 k in bf16), so one K-tile is one MFMA k-step: a wave's 16 MFMAs (32 cycles each) per 48 KiB
// staged, the same matrix-pipe time per staged byte as bf16 at twice the FLOPs.  Operand k order inside an MFMA is free as
// long as A and B agree (probed with exact integer data): lane (r = lane & 15, q = lane >> 4) takes bytes 32 q .. 32 q + 31 of
// row r for both.  One barrier per K-tile; the next tile's fragment reads and the refill of the slot just consumed are issued
// under the current tile's MFMAs (two register sets of fragments, loop unrolled by two).
namespace fp8k {
using bf16k::SUB; using bf16k::i32x4; using bf16k::dma16; using bf16k::make_rsrc;
constexpr int BM = 256, BN = 128, BK = 128, STAGE_BYTES = 3 * SUB, LDS_BYTES = 3 * STAGE_BYTES;
typedef __attribute__((ext_vector_type(8))) int i32x8;
// wave-instruction `inst` (0..15) of a sub-tile [128 rows][128 bytes]: rows inst*8 + (lane>>3), 16-byte chunk (lane&7) ^ (row&7)
__device__ __forceinline__ void stage_inst8(i32x4 rsrc, unsigned lds_sub, int ld, int X, int x0, int k0, int kend, int inst, int lane) {
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    const int gx = x0 + r, gk = k0 + 16 * c;
    unsigned off = (unsigned)((size_t)gx * ld + gk);
    if (gx >= X || gk >= kend) off = 0x80000000u;        // (K is a multiple of 16: a chunk is inside or outside as a whole)
    dma16(rsrc, lds_sub + inst * 1024, off);
}
__device__ __forceinline__ i32x8 read_frag8(const char* S, int xb, int lane) {
    const int row = xb + (lane & 15), c0 = 2 * (lane >> 4);
    const i32x4 lo = *reinterpret_cast<const i32x4*>(S + row * 128 + (((c0) ^ (row & 7)) << 4));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(S + row * 128 + (((c0 + 1) ^ (row & 7)) << 4));
    return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__global__ __launch_bounds__(512, 2) void gemm_fp8(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn, z;
    tile_of_block(p, BM, BN, blockIdx.x, gridDim.x, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nt = (p.K + BK - 1) / BK;
    const i32x4 rA = make_rsrc(p.A), rB = make_rsrc(p.B);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Interior blocks (whole tile inside M x N, K a multiple of 128) address their pieces the FAST way of the bf16 ring: the
    // lane-dependent part of a piece's source offset is the same for all of a wave's pieces of an operand (row-in-piece and
    // chunk swizzle depend on the lane only), so it is ONE VGPR per operand; which piece and which K-tile are a wave-uniform
    // SGPR offset.  No per-piece address arithmetic or bounds selects inside the K loop.
    const bool fast = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (p.K % BK == 0);
    unsigned fvA, fvB;
    {
        const int r8 = lane >> 3, c = (lane & 7) ^ (r8 & 7), ga = wave * 4, gb = wave * 2;
        fvA = (unsigned)((size_t)(m0 + (ga >> 4) * 128 + (ga & 15) * 8 + r8) * p.lda + 16 * c);
        fvB = (unsigned)((size_t)(n0 + gb * 8 + r8) * p.ldb + 16 * c);
    }
    const unsigned fpA = 8u * p.lda, fpB = 8u * p.ldb;
    auto dma_s = [&](i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %4, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(soff), "s"(rsrc) : "memory");
    };
    // piece q of the wave (0..3: A, 4..5: B) of K-tile t into ring slot `slot`
    auto stage_piece = [&](int t, int slot, int q) {
        const unsigned S = lds0 + slot * STAGE_BYTES;
        if (q < 4) {
            const int g = wave * 4 + q;
            if (fast) dma_s(rA, S + (g >> 4) * SUB + (g & 15) * 1024, fvA, (unsigned)t * BK + (unsigned)q * fpA);
            else stage_inst8(rA, S + (g >> 4) * SUB, p.lda, p.M, m0 + (g >> 4) * 128, t * BK, p.K, g & 15, lane);
        } else {
            const int i = q - 4;
            if (fast) dma_s(rB, S + 2 * SUB + (wave * 2 + i) * 1024, fvB, (unsigned)t * BK + (unsigned)i * fpB);
            else stage_inst8(rB, S + 2 * SUB, p.ldb, p.N, n0, t * BK, p.K, wave * 2 + i, lane);
        }
    };
    auto stage = [&](int t, int slot) {
#pragma unroll
        for (int q = 0; q < 6; ++q) stage_piece(t, slot, q);
    };
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nt) stage(t, t);
    if (nt >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nt == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto read_tile = [&](int slot, i32x8 (&fa)[4], i32x8 (&fb)[4]) {
        const char* S = smem + slot * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag8(S + (wm >> 1) * SUB, (wm & 1) * 64 + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag8(S + 2 * SUB, wn * 64 + 16 * j, lane);
    };
    // one K-tile: [tile t+1 has landed -> barrier -> refill the slot of tile t (everybody holds it in registers) -> issue the
    // fragment reads of tile t+1 into the OTHER register set] under the 16 MFMAs of tile t -> the reads are complete
    auto step = [&](int t, int slot, const i32x8 (&fa)[4], const i32x8 (&fb)[4], i32x8 (&na)[4], i32x8 (&nb)[4]) {
        const bool more = t + 1 < nt, refill = t + 3 < nt;
        if (more) {
            // outstanding, oldest first: tile t+1, tile t+2 (6 pieces each)
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const char* Sn = smem + (slot + 1 == 3 ? 0 : slot + 1) * STAGE_BYTES;
        __builtin_amdgcn_s_setprio(1);
        // operands swapped as in the bf16 kernels: D'[n][m], a lane owns 4 consecutive n of one row m.  The refill's DMA pieces
        // and the next tile's fragment reads go out BETWEEN the rows of MFMAs (an LDS-DMA piece holds its wave ~100 cycles:
        // issued in one burst ahead of the MFMAs, both waves of a SIMD stall together)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (refill) {
#pragma unroll
                for (int q = (i * 6) / 4; q < ((i + 1) * 6) / 4; ++q) stage_piece(t + 3, slot, q);
            }
            if (more) {
                na[i] = read_frag8(Sn + (wm >> 1) * SUB, (wm & 1) * 64 + 16 * i, lane);
                nb[i] = read_frag8(Sn + 2 * SUB, wn * 64 + 16 * i, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    i32x8 fa0[4], fb0[4], fa1[4], fb1[4];
    if (nt > 0) read_tile(0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
    for (int t = 0; t < nt; t += 2) {
        step(t, slot, fa0, fb0, fa1, fb1);
        slot = slot + 1 == 3 ? 0 : slot + 1;
        if (t + 1 < nt) {
            step(t + 1, slot, fa1, fb1, fa0, fb0);
            slot = slot + 1 == 3 ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float lsum = 0.f;
    bf16k::wave_epilogue<0, 0, 4, true>(p, acc, m0 + wm * 64, n0 + wn * 64, 0, reinterpret_cast<float*>(smem) + wave * 4096, lane, lsum);
}
}  // namespace fp8k
hipError_t afr_launch_gemm_fp8(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if ((p.flags & (AFR_GEMM_A_KSTRIDED | AFR_GEMM_B_KSTRIDED)) || p.splitk != 1 || (p.K & 15) || (p.lda & 15) || (p.ldb & 15) ||
        p.mse_target || p.ad_p || p.colsum || (p.flags & AFR_GEMM_RELU_MASK)) return hipErrorInvalidValue;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    hipLaunchKernelGGL(fp8k::gemm_fp8, dim3(tiles), dim3(512), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- launch
#ifdef AFR_GEMM_LAB
static int g_gemm_variant = getenv("AFR_GEMM_VARIANT") ? atoi(getenv("AFR_GEMM_VARIANT")) : 0;
extern "C" void afr_dbg_set_gemm_variant(int v) { g_gemm_variant = v; }
#endif
#ifdef AFR_GEMM_TIMING
static int g_dbg_slot = 0;
extern "C" int afr_dbg_gemm_stamps(void* devbuf) {
    g_dbg_slot = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bf16k::g_gemm_stamps), &devbuf, sizeof(void*));
}
extern "C" void afr_dbg_gemm_slot_reset(void) { g_dbg_slot = 0; }
#define DBG_SLOT(q) (q).dbg_slot = g_dbg_slot++ & 63
#else
#define DBG_SLOT(q) do { } while (0)
#endif
// bf16: the 256x128 / 8-wave kernel when its grid fills most of the 256 CUs, else 128x128 / 4 waves
static bool bf16_use_wide(const GemmParams& p) {
    // a fused-AdamW epilogue moves 26 B per output element and is the longer half of such a kernel; two 128x128 blocks
    // per CU (64 KiB of LDS each) let one block's epilogue run under the other's K loop, one 256x128 block cannot
    if (p.ad_p) return false;
    const long long t = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128) * p.splitk;
    // a wide grid that leaves CUs idle loses to the 128x128 kernel at two blocks per CU (R0 dX: 200 wide tiles 323 us,
    // 400 narrow ones 302 us); from a full round on the two run level and wide needs fewer L2->LDS bytes
    return t >= 232 && p.K / p.splitk >= 256;      // the 3-stage ring needs a few K-tiles to pay
}
// A plain product on the 256x256 body (a one-member grouped launch): when its tiles make four or more full rounds of the
// chip, so that the ragged last round is small change.  Measured on the pixel transformer's products (131072 rows): 8-20 %
// faster than the 256x128 ring (K = 512: 466 -> 390 us forward, 104 -> 87 us input gradient; K = 2048: 322 -> 266 us).
static bool bf16_use_body256(const GemmParams& p) {
    static const int off = getenv("AFR_GEMM_NO_BODY256") ? atoi(getenv("AFR_GEMM_NO_BODY256")) : 0;      // kernel A/B measurements
    if (off || p.mse_target || p.ad_p || p.a_rowmap || p.b_rowmap || p.coop_ws || p.fix_ws) return false;
    const long long t = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    // weight gradients (both operands k-strided, split-K slabs): when the slices of the 256x256 tiles make whole rounds of the chip
    const bool kk = (p.flags & AFR_GEMM_A_KSTRIDED) && (p.flags & AFR_GEMM_B_KSTRIDED);
    if (kk && p.splitk > 1) return (t * p.splitk) % 256 == 0 && p.K / p.splitk >= 1024;
    if (p.splitk != 1 || p.colsum) return false;
    return t >= 1024 && p.K >= 256;
}
bool afr_gemm_wide_ok(int M, int N, int K) {
    GemmParams q;
    q.M = M; q.N = N; q.K = K; q.splitk = 1;
    return bf16_use_wide(q);
}
// the symbol rocprofv3 will report for this launch (without the "void bf16k::" decoration)
const char* afr_gemm_kernel_name(int dtype, const GemmParams& p) {
    const int a = (p.flags & AFR_GEMM_A_KSTRIDED) ? 1 : 0, b = (p.flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0;
    static const char* f32n[2][2] = {{"gemm_f32<0,0>", "gemm_f32<0,1>"}, {"gemm_f32<1,0>", "gemm_f32<1,1>"}};
    static const char* bfn[2][2][2] = {{{"gemm_bf16<0,0,2>", "gemm_bf16<0,0,4>"}, {"gemm_bf16<0,1,2>", "gemm_bf16<0,1,4>"}},
                                       {{"gemm_bf16<1,0,2>", "gemm_bf16<1,0,4>"}, {"gemm_bf16<1,1,2>", "gemm_bf16<1,1,4>"}}};
    if (dtype != AFR_BF16) return f32n[a][b];
    if (bf16_use_body256(p)) return "gemm_bf16_group256";
    return bfn[a][b][bf16_use_wide(p) ? 1 : 0];
}
// true when the product would run on the 256x128 ring kernel by itself (what a grouped launch is built from)
bool afr_gemm_groupable(int dtype, const GemmParams& p) {
    return dtype == AFR_BF16 && p.M > 0 && p.N > 0 && !p.mse_target && (!p.ad_p || p.coop_ws) && p.K / p.splitk >= 256;
}
// How a layer's gradient pair (dX: B x K_in over N_out; dW: N_out x K_in over the batch) is launched: with 256x256 tiles
// when those fill most of the chip in ONE round (dW split so that its blocks run as many K-tiles as dX's), else with
// 256x128 tiles and dW blocks of twice the K-tiles (a CU draws one long block or two short ones).
void afr_gemm_pair_plan(int B, int n_out, int k_in, int* tile256, int* splitk) {
    const long long dx256 = (long long)((B + 255) / 256) * ((k_in + 255) / 256);
    const long long dw256 = (long long)((n_out + 255) / 256) * ((k_in + 255) / 256);
    int sk = n_out > 0 ? (B + n_out - 1) / n_out : 1;                 // dW K-tiles per block == dX K-tiles per block
    if (sk < 1) sk = 1;
    const long long total = dx256 + dw256 * sk;
    const char* force = getenv("AFR_GEMM_PAIR_TILE");                  // 128 | 256: kernel A/B measurements
    bool use256 = total >= 192 && total <= 272 && B / sk >= 256;
    if (force) use256 = atoi(force) == 256;
    if (use256) { *tile256 = 1; *splitk = sk; return; }
    *tile256 = 0;
    sk = n_out > 0 ? (B + 2 * n_out - 1) / (2 * n_out) : 1;
    if (sk < 1) sk = 1;
    if (B / sk < 256) sk = 0;                                         // 0: no grouped launch
    *splitk = sk;
}
// In-launch split-K on 256x256 tiles (GemmParams::fix_ws): worth it when the tile count leaves the last round of the chip
// mostly empty -- the tail tiles are then cut along K so that the round's work is spread over every CU -- and the product
// is deep enough that the 256x256 kernel's fewer staged bytes per FLOP outweigh the parked slices.  Launch model in
// microseconds per K-tile (measured, operands in L2/MALL): 1.5 per block-round of the 256x256 kernel, 0.92 of the 256x128
// ring, 0.95 of the 128x128 kernel at two blocks per CU; 12 us for parking and summing the slices.
bool afr_gemm_fix_plan(int M, int N, int K, int* head_tiles, int* splitk) {
    static const int force = getenv("AFR_GEMM_FIX") ? atoi(getenv("AFR_GEMM_FIX")) : -1;     // 0 | 1: kernel A/B measurements
    if (force == 0 || M < 256 || N < 256 || K < 2048) return false;
    const int kt = (K + 63) / 64;
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const int rounds = tiles / 256, rem = tiles % 256;
    int sk = 1;
    if (rem) {
        sk = 256 / rem;
        if (sk > 8) sk = 8;
        while (sk > 1 && kt / sk < 8) --sk;
    }
    if (getenv("AFR_GEMM_FIX_SK") && rem) sk = atoi(getenv("AFR_GEMM_FIX_SK"));            // kernel A/B measurements
    if ((size_t)rem * sk > AFR_FIX_MAX_SLICES) return false;
    const double c256 = (rounds * (double)kt + (rem ? (kt + sk - 1) / sk : 0)) * 1.5 + (sk > 1 ? 12.0 : 0.0);
    const long long tw = (long long)((M + 255) / 256) * ((N + 127) / 128), tn = (long long)((M + 127) / 128) * ((N + 127) / 128);
    const double cw = (double)((tw + 255) / 256) * kt * 0.92, cn = (double)((tn + 511) / 512) * kt * 0.95;
    const double cr = cw < cn ? cw : cn;
    if (force != 1 && !(c256 < 0.9 * cr)) return false;
    *head_tiles = tiles - rem;
    *splitk = sk;
    return true;
}
hipError_t afr_launch_gemm_fix(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int head = p.head_tiles < tiles ? p.head_tiles : tiles;
    bf16k::GemmGroup g;
    g.n = 1; g.p[0] = p; g.blk0[0] = 0;
    g.blk0[1] = head + (tiles - head) * p.splitk;
    hipLaunchKernelGGL(bf16k::gemm_bf16_group256, dim3(g.blk0[1]), dim3(512), 0, s, g);
    return hipGetLastError();
}
hipError_t afr_launch_gemm_group(int dtype, const GemmParams* ps, int n, int tile256, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    bool ok = n <= 4, coop = false;
    for (int i = 0; i < n && ok; ++i) ok = afr_gemm_groupable(dtype, ps[i]);
    for (int i = 0; i < n; ++i) {          // row gathers a grouped launch supports: B k-strided with A k-strided on 256x256 tiles; aux
        const bool kk = (ps[i].flags & AFR_GEMM_A_KSTRIDED) && (ps[i].flags & AFR_GEMM_B_KSTRIDED);
        if (ps[i].a_rowmap || (ps[i].b_rowmap && !(ok && n > 1 && tile256 && kk)) || (ps[i].aux_rowmap && !(ps[i].flags & AFR_GEMM_OUT_BF16)))
            return hipErrorInvalidValue;
    }
    for (int i = 0; i < n; ++i) {
        if (!ps[i].coop_ws) continue;
        // cooperative split-K exists in the 256x256 body only, one workgroup per CU, partners resident together
        const long long tiles = (long long)((ps[i].M + 255) / 256) * ((ps[i].N + 255) / 256);
        const int S = ps[i].splitk;
        // (listed first: the dispatcher hands out workgroups in grid order, so the waiting ones all become resident)
        if (i != 0 || !ok || !tile256 || !ps[i].coop_cnt || (S != 2 && S != 4 && S != 8) || tiles * S > 256 ||
            tiles * S * (long long)AFR_FIX_SLICE_BYTES >= (1ll << 31)) return hipErrorInvalidValue;
        coop = true;
    }
    if (!ok || (n == 1 && !coop)) {
        for (int i = 0; i < n; ++i) { hipError_t e = afr_launch_gemm(dtype, ps[i], s); if (e != hipSuccess) return e; }
        return hipSuccess;
    }
    bf16k::GemmGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = ps[i];
        g.blk0[i] = total;
        int nb = ((ps[i].M + 255) / 256) * ((ps[i].N + (tile256 ? 255 : 127)) / (tile256 ? 256 : 128)) * ps[i].splitk;
        total += (nb + 7) & ~7;        // each product's range starts on a multiple of 8: blocks b, b+8, ... keep sharing an XCD
    }
    g.blk0[n] = total;
#ifdef AFR_GEMM_TIMING
    { const int slot = g_dbg_slot++ & 63; for (int i = 0; i < n; ++i) g.p[i].dbg_slot = slot; }
#endif
    if (tile256) hipLaunchKernelGGL(bf16k::gemm_bf16_group256, dim3(total), dim3(512), 0, s, g);
    else hipLaunchKernelGGL(bf16k::gemm_bf16_group, dim3(total), dim3(512), 0, s, g);
    return hipGetLastError();
}
hipError_t afr_launch_gemm(int dtype, const GemmParams& p_in, hipStream_t s) {
#ifdef AFR_GEMM_TIMING
    GemmParams p = p_in;
    DBG_SLOT(p);
#else
    const GemmParams& p = p_in;
#endif
    const int a = (p.flags & AFR_GEMM_A_KSTRIDED) ? 1 : 0, b = (p.flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0;
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.a_rowmap || p.b_rowmap || p.aux_rowmap) {
        // row gathers: A k-contiguous on the 256x128 ring kernel; aux with a bf16 output (B: grouped 256x256 launches only)
        if (dtype != AFR_BF16 || p.b_rowmap || (p.a_rowmap && (a || b || !bf16_use_wide(p))) ||
            (p.aux_rowmap && !(p.flags & AFR_GEMM_OUT_BF16))) return hipErrorInvalidValue;
    }
    if (dtype == AFR_BF16) {
        if (bf16_use_body256(p)) {
            bf16k::GemmGroup g;
            g.n = 1; g.p[0] = p; g.blk0[0] = 0;
            g.blk0[1] = ((p.M + 255) / 256) * ((p.N + 255) / 256) * p.splitk;
            hipLaunchKernelGGL(bf16k::gemm_bf16_group256, dim3(g.blk0[1]), dim3(512), 0, s, g);
            return hipGetLastError();
        }
        const bool wide = bf16_use_wide(p);
        const int bm = wide ? 256 : 128;
        const int tiles = ((p.M + bm - 1) / bm) * ((p.N + 127) / 128);
        dim3 grid(tiles * p.splitk, 1, 1);
#define LB(AL, BL) do { if (wide) hipLaunchKernelGGL((bf16k::gemm_bf16<AL, BL, 4>), grid, dim3(512), 0, s, p); \
                        else hipLaunchKernelGGL((bf16k::gemm_bf16<AL, BL, 2>), grid, dim3(256), 0, s, p); } while (0)
#ifdef AFR_GEMM_LAB
        if (g_gemm_variant == 260 && !a && !b) {
            bf16k::GemmGroup g;
            g.n = 1; g.p[0] = p; g.blk0[0] = 0;
            g.blk0[1] = ((((p.M + 255) / 256) * ((p.N + 255) / 256) * p.splitk) + 7) & ~7;
            hipLaunchKernelGGL((bf16k::gemm_bf16_lab256<6>), dim3(g.blk0[1]), dim3(512), 0, s, g);
            return hipGetLastError();
        }
        if (g_gemm_variant >= 257 && g_gemm_variant <= 259 && !a && !b) {
            bf16k::GemmGroup g;
            g.n = 1; g.p[0] = p; g.blk0[0] = 0;
            g.blk0[1] = ((((p.M + 255) / 256) * ((p.N + 255) / 256) * p.splitk) + 7) & ~7;
            if (g_gemm_variant == 257) hipLaunchKernelGGL((bf16k::gemm_bf16_lab256<1>), dim3(g.blk0[1]), dim3(512), 0, s, g);
            else if (g_gemm_variant == 258) hipLaunchKernelGGL((bf16k::gemm_bf16_lab256<2>), dim3(g.blk0[1]), dim3(512), 0, s, g);
            else hipLaunchKernelGGL((bf16k::gemm_bf16_lab256<3>), dim3(g.blk0[1]), dim3(512), 0, s, g);
            return hipGetLastError();
        }
        if (g_gemm_variant == 256 || g_gemm_variant == 128) {       // one product through the grouped kernels
            bf16k::GemmGroup g;
            g.n = 1; g.p[0] = p; g.blk0[0] = 0;
            const int t256 = g_gemm_variant == 256;
            const int nb = ((p.M + 255) / 256) * ((p.N + (t256 ? 255 : 127)) / (t256 ? 256 : 128)) * p.splitk;
            g.blk0[1] = (nb + 7) & ~7;
            if (t256) hipLaunchKernelGGL(bf16k::gemm_bf16_group256, dim3(g.blk0[1]), dim3(512), 0, s, g);
            else hipLaunchKernelGGL(bf16k::gemm_bf16_group, dim3(g.blk0[1]), dim3(512), 0, s, g);
            return hipGetLastError();
        }
        if (wide && !a && !b && g_gemm_variant >= 100 && g_gemm_variant < 116) {
            switch (g_gemm_variant - 100) {
#define LABL(x) case x: hipLaunchKernelGGL((bf16k::gemm_bf16<0, 0, 4, x>), grid, dim3(512), 0, s, p); break;
                LABL(1) LABL(2) LABL(3) LABL(8)
#undef LABL
                default: hipLaunchKernelGGL((bf16k::gemm_bf16<0, 0, 4, 0>), grid, dim3(512), 0, s, p);
            }
            return hipGetLastError();
        }
#endif
        if (p.a_rowmap) hipLaunchKernelGGL((bf16k::gemm_bf16<0, 0, 4, 0, 1>), grid, dim3(512), 0, s, p);
        else if (!a && !b) LB(0, 0);
        else if (!a && b) LB(0, 1);
        else if (a && !b) LB(1, 0);
        else LB(1, 1);
#undef LB
    } else {
        const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        dim3 grid(tiles * p.splitk, 1, 1), block(256, 1, 1);
#define LF(AL, BL) hipLaunchKernelGGL((f32k::gemm_f32<AL, BL>), grid, block, 0, s, p)
        if (!a && !b) LF(0, 0);
        else if (!a && b) LF(0, 1);
        else if (a && !b) LF(1, 0);
        else LF(1, 1);
#undef LF
    }
    return hipGetLastError();
}

// ---- folded first layer, fused backward (bf16)
bool afr_glyph_l1_bwd_fused_eligible(int dtype, int E, int N1, int vocab, int n_fonts) {
    return dtype == AFR_BF16 && E == bf16k::L1_E && N1 % 128 == 0 && N1 >= 256 && N1 <= 1024 && vocab + n_fonts <= 144 && !getenv("AFR_NO_L1_FUSED");
}
// column ranges per row block: enough blocks to cover the chip, ranges of whole 128-column sub-tiles
int afr_glyph_l1_bwd_fused_split(int B, int N1) {
    const int nrb = (B + bf16k::L1_R - 1) / bf16k::L1_R;
    int cs = 1;
    while (cs * 2 <= AFR_L1F_MAX_SPLIT && cs * 2 * nrb <= 256 && N1 % (cs * 2 * 128) == 0) cs *= 2;
    return cs;
}
int afr_glyph_l1_bwd_fused_blocks(int B, int N1) { return (B + bf16k::L1_R - 1) / bf16k::L1_R * afr_glyph_l1_bwd_fused_split(B, N1); }
long long afr_glyph_l1_bwd_fused_slab_floats(int B, int N1, int vocab, int n_fonts) {
    const int nc = N1 / afr_glyph_l1_bwd_fused_split(B, N1);
    return (long long)nc * bf16k::L1_E + nc + (long long)(vocab + n_fonts) * bf16k::L1_E;
}
hipError_t afr_launch_glyph_l1_bwd_fused(const void* d1, int ldd, const void* h0, int ldh, const void* W1T, const int64_t* x,
                                         const int64_t* font, int B, int N1, int vocab, int n_fonts, float* slabs, hipStream_t s,
                                         const int* h0_rowmap) {
    if (B <= 0) return hipSuccess;
    bf16k::L1BwdArgs a;
    a.d1 = (const bf16_t*)d1; a.h0 = (const bf16_t*)h0; a.W1T = (const bf16_t*)W1T; a.x = x; a.font = font; a.h0_rowmap = h0_rowmap;
    a.ldd = ldd; a.ldh = ldh; a.B = B; a.N1 = N1; a.vocab = vocab; a.n_fonts = n_fonts;
    a.CS = afr_glyph_l1_bwd_fused_split(B, N1); a.ncols = N1 / a.CS;
    a.slabs = slabs; a.slab_stride = afr_glyph_l1_bwd_fused_slab_floats(B, N1, vocab, n_fonts);
    a.o_b = a.ncols * bf16k::L1_E; a.o_tab = a.o_b + a.ncols;
    const size_t lds = (size_t)(a.ncols / 128 + 1) * bf16k::SUB + 2 * bf16k::L1_R * sizeof(int);
    static size_t set[16];
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && (dev < 0 || dev >= 16 || set[dev] < lds)) {
        hipError_t e = hipFuncSetAttribute((const void*)bf16k::glyph_l1_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) set[dev] = lds;
    }
    hipLaunchKernelGGL(bf16k::glyph_l1_bwd_fused_kernel, dim3(afr_glyph_l1_bwd_fused_blocks(B, N1)), dim3(512), lds, s, a);
    return hipGetLastError();
}
