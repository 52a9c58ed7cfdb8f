// glyph_fused.hip -- the launch-bound glyph nets (BASELINE configs[0] / [1]: embedding -> Linear(32,256) + ReLU ->
// Linear(256,256) -> clamp, 78 K parameters) as ONE kernel per training step plus the grouped slab reduction.
//
// (Round 3: a row block may be shared by cs = 2 or 4 workgroups, each owning P / cs OUTPUT COLUMNS -- the batch of these
// configs leaves most CUs idle (C1: 6 row blocks, C2: 64), and everything after the loss is linear in du: a workgroup computes
// fc1 for its rows in full, then its columns of u / loss / du, its rows of dW2 / db2, and its PART of dh1 = du . W2 (the
// reduction over its own columns only), from which dW1, db1 and the table rows follow as partial sums that the grouped reduce
// adds over all cs * row-block slabs in block order.)
// Every sample is independent up to the weight gradients, so a workgroup takes a block of R batch rows through the WHOLE
// step -- gather (reference model.py:136,167), fc1 + ReLU (:148,183), fc_output (:152,196), clamp + MSE + d(loss)/du
// (:156,268-270), and the backward of all of it (:309) -- with every activation of those rows resident in LDS, and
// leaves its partial parameter gradients in a slab laid out like the flat gradient buffer.  reduce_group_kernel then
// sums the slabs in block order (bitwise reproducible) and, on one GPU, applies AdamW in the same pass.  11-13 launches
// of latency-bound small-grid kernels become 2.
//
// All six products are 16x16-tile MFMA loops of one shape, C[i][j] = sum_k A(i,k) B(j,k) with BOTH operands k-contiguous:
//     pre1 = h0 . W1^T        A = h0  [R][E]      B = W1  [N1][E]            (global)
//     u    = h1 . W2^T        A = h1  [R][N1]     B = W2  [P][N1]            (global)
//     dW2  = du^T . h1        A = duT [P][R]      B = h1T [N1][R]            reduction over the block's rows
//     dh1  = du . W2          A = du  [R][P]      B = W2T [N1][P]            (global, transposed copy) / f32: gathered
//     dW1  = dpre1^T . h0     A = d1T [N1][R]     B = h0T [E][R]
//     dh0  = dpre1 . W1       A = d1  [R][N1]     B = W1T [E][N1]            (global, transposed copy) / f32: gathered
// so each activation is kept in LDS in both orientations (an accumulator tile holds 4 consecutive rows of a column per
// lane: the transposed copy is one vector store).  A lane's operand is 16 bytes: 4 floats feeding 4 exact-f32 MFMAs
// (v_mfma_f32_16x16x4_f32, parity mode: k-order permuted within 16, still one rounding per product) or 8 bf16 feeding one
// v_mfma_f32_16x16x32_bf16 (throughput mode).  Weights come straight from L2 (78 K parameters; every block reads them all).
#include "afr_common.h"
#include "../../include/afr.h"

namespace {
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <typename T> struct MM;
template <> struct MM<float> {
    static constexpr int KB = 16, R = 16, NTH = 256; // k per 16-byte operand step; batch rows per block; threads per block
    static __device__ __forceinline__ void mma(f32x4& acc, const i32x4 a, const i32x4 b) {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc, 0, 0, 0);
    }
};
template <> struct MM<bf16_t> {
    static constexpr int KB = 32, R = 64, NTH = 512;
    static __device__ __forceinline__ void mma(f32x4& acc, const i32x4 a, const i32x4 b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float v) { return (bf16_t)v; }

constexpr int PADB = 16;                             // every LDS row is padded by one 16-byte access (bank spread)

// 16 bytes of a k-contiguous operand: row `row`, k-block kb, lane quarter q
template <typename T>
__device__ __forceinline__ i32x4 ld16(const T* base, int ld, int row, int kb, int q) {
    return *reinterpret_cast<const i32x4*>(base + (size_t)row * ld + kb * MM<T>::KB + q * (MM<T>::KB / 4));
}
// the same operand slice gathered from a k-STRIDED f32 array W[k][x] (parity mode has no transposed weight copies):
// component c is W[kb*16 + 4q + c][x]
__device__ __forceinline__ i32x4 gather16(const float* W, int ld, int x, int kb, int q) {
    i32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = __builtin_bit_cast(int, W[(size_t)(kb * 16 + 4 * q + c) * ld + x]);
    return v;
}
// acc[m] (16 rows m*16.. of A x column tile nt) = sum over nkb k-blocks of A . B, for the column tiles nt = wave, wave + NW, ...
// A is in LDS; the B operand of (tile nt, k-block kb) comes from L2 through loadB(nt, kb).  Software pipelined by hand, a
// whole tile deep: while tile nt feeds the MFMAs, all of the wave's next tile's B operands (<= NKB x 16 bytes per lane) are already
// in flight (left to the compiler every load was waited for right before its use).  pre(nt) runs when a tile's MFMAs
// start: whatever its epilogue wants from memory (targets, bias) is issued there.  Plain loops, no index arithmetic: a
// first version that flattened (tile, k-block) pairs with divisions compiled to 5000 instructions per phase and ran
// instruction-bound.
// The B operands of a wave's FIRST tile may be requested by the caller ahead of time (first_tile, below: during the previous
// phase, so that their L2 latency -- 1.5-2 us of every phase otherwise -- is not on the critical path): `pref` != nullptr.
template <typename T> struct FirstTile { i32x4 v[256 / MM<T>::KB]; };
template <typename T, class LoadB>
__device__ __forceinline__ void first_tile(FirstTile<T>& ft, int ntiles, int nkb, int wave, LoadB loadB) {
    constexpr int NKB = 256 / MM<T>::KB;
    if (wave < ntiles) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) ft.v[kb] = loadB(wave, kb);
    }
}
template <typename T, int MT, int NW, class LoadB, class Pre, class Epi>
__device__ __forceinline__ void rows_times_global(const T* A, int lda, int ntiles, int nkb, int wave, int r, int q, LoadB loadB, Pre pre, Epi epi,
                                                  const FirstTile<T>* pref = nullptr) {
    constexpr int NKB = 256 / MM<T>::KB;             // K <= 256
    i32x4 cur[NKB], nxt[NKB];
    auto fetch = [&](i32x4 (&dst)[NKB], int nt) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) dst[kb] = loadB(nt, kb);
    };
    if (pref) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) cur[kb] = pref->v[kb];
    } else if (wave < ntiles) fetch(cur, wave);
    for (int nt = wave; nt < ntiles; nt += NW) {
        if (nt + NW < ntiles) fetch(nxt, nt + NW);
        pre(nt);
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb < nkb) {
#pragma unroll
                for (int m = 0; m < MT; ++m) MM<T>::mma(acc[m], ld16(A, lda, m * 16 + r, kb, q), cur[kb]);
            }
        }
        epi(nt, acc);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) cur[kb] = nxt[kb];
    }
}
// The same loop with the wave's (at most TPW) tiles unrolled: pre / epi get the tile's ordinal `it` as a compile-time value, so a
// phase can keep per-tile state it requested long before (the loss phase: every target of the wave's tiles, asked for at
// kernel entry) in registers.
template <typename T, int MT, int NW, int TPW, class LoadB, class Pre, class Epi>
__device__ __forceinline__ void rows_times_global_u(const T* A, int lda, int ntiles, int nkb, int wave, int r, int q, LoadB loadB, Pre pre, Epi epi,
                                                    const FirstTile<T>* pref = nullptr) {
    constexpr int NKB = 256 / MM<T>::KB;
    i32x4 cur[NKB], nxt[NKB];
    auto fetch = [&](i32x4 (&dst)[NKB], int nt) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkb) dst[kb] = loadB(nt, kb);
    };
    if (pref) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) cur[kb] = pref->v[kb];
    } else if (wave < ntiles) fetch(cur, wave);
#pragma unroll
    for (int it = 0; it < TPW; ++it) {
        const int nt = wave + it * NW;
        if (nt < ntiles) {
            if (nt + NW < ntiles) fetch(nxt, nt + NW);
            pre(nt, it);
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb < nkb) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) MM<T>::mma(acc[m], ld16(A, lda, m * 16 + r, kb, q), cur[kb]);
                }
            }
            epi(nt, it, acc);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) cur[kb] = nxt[kb];
        }
    }
}
}  // namespace


template <typename T, bool TU8>
__global__ __launch_bounds__(MM<T>::NTH) void glyph1_step_kernel(Glyph1Args a) {
    using M_ = MM<T>;
    constexpr int KB = M_::KB, R = M_::R, MT = R / 16, NT = M_::NTH, NW = NT / 64;
    constexpr int PAD = PADB / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int E = a.E, N1 = a.N1, P = a.P;
    const int ldE = E + PAD, ldN = N1 + PAD, ldP = P + PAD, ldR = R + PAD;
    // LDS carve: ids [2R] | (elements of T) h0 [R][E] | h0T [E][R] | h1 [R][N1] | h1T [N1][R] | du [R][P] | duT [P][R]; the
    // embedding phase reuses everything from du on as f32: dh0 [R][E] | tab [(vocab + n_fonts)][E] (the launch sizes the
    // allocation for whichever is longer)
    int* ids = reinterpret_cast<int*>(smem_);                 // [R] codes, [R] font ids
    int* fids = ids + R;
    float* lut = reinterpret_cast<float*>(smem_ + 2 * R * sizeof(int));     // [256]: k / 255.0f, the pixel values of helpers.py:121
    T* h0 = reinterpret_cast<T*>(smem_ + 2 * R * sizeof(int) + 256 * sizeof(float));
    T* h0T = h0 + R * ldE;
    T* h1 = h0T + E * ldR;
    T* h1T = h1 + R * ldN;
    T* du = h1T + N1 * ldR;
    T* duT = du + R * ldP;
#ifdef AFR_G1_DEBUG
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int cs = a.cs, rb = (int)blockIdx.x / cs, pc = (int)blockIdx.x - rb * cs;
    const int npt = P / 16 / cs, pt0 = pc * npt;       // this block's output-column tiles [pt0, pt0 + npt)
    const int b0 = rb * R;
    const int nb = min(R, a.B - b0);
    // every target this wave's loss tiles will need (P <= 256: at most TPW tiles of MT x 4 pixels per lane), requested now:
    // they arrive under the gather and fc1 (asked for when a tile's MFMAs start, each tile waited ~1.5 us for them)
    constexpr int TPW = 16 / NW;
    float tv[TPW][MT][4];
#pragma unroll
    for (int it = 0; it < TPW; ++it) {
        const int pcol = (pt0 + wave + it * NW) * 16 + r;
        if (wave + it * NW < npt) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t ti = (size_t)(b0 + min(m * 16 + 4 * q + i, nb - 1)) * P + pcol;
                    if constexpr (TU8) tv[it][m][i] = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const uint8_t*>(a.target)[ti]);
                    else tv[it][m][i] = reinterpret_cast<const float*>(a.target)[ti];
                }
        }
    }
#ifdef AFR_G1_DEBUG
    unsigned long long tstamp[12]; int nst = 0;
#define G1STAMP() do { __syncthreads(); tstamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G1STAMP() do { } while (0)
#endif
    G1STAMP();
    // (plain stores: streaming ones made this kernel 2 us shorter and the reduce, which then reads the slabs from HBM, 3 us longer)
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;

    const T* W1 = reinterpret_cast<const T*>(a.W1);
    const T* W2 = reinterpret_cast<const T*>(a.W2);
    auto loadW1 = [&](int nt, int kb) { return ld16(W1, E, nt * 16 + r, kb, q); };
    auto loadW2 = [&](int nt, int kb) { return ld16(W2, N1, nt * 16 + r, kb, q); };
    FirstTile<T> ftA, ftB;                               // two in flight at most: the next phase's, and the one being consumed
    constexpr bool PF = sizeof(T) == 2;                  // f32 operands are 4x the registers per tile: no room to run ahead
    if constexpr (PF) first_tile<T>(ftA, N1 / 16, E / KB, wave, loadW1);   // fc1's first tile: requested before the codes are even read
    // ---- codes (index check as glyph_embed_kernel: out of range sets the error word and is clamped) and the gather
    if (tid < R) {
        long long xi = tid < nb ? a.x[b0 + tid] : 0, fi = (tid < nb && a.n_fonts > 0 && a.font) ? a.font[b0 + tid] : 0;
        if (xi < 0 || xi >= a.vocab) { atomicOr(a.err, 1u); xi = min(max(xi, 0ll), (long long)a.vocab - 1); }
        if (a.n_fonts > 0 && (fi < 0 || fi >= a.n_fonts)) { atomicOr(a.err, 1u); fi = min(max(fi, 0ll), (long long)a.n_fonts - 1); }
        ids[tid] = (int)xi; fids[tid] = (int)fi;
    }
    if (TU8 && tid < 256) lut[tid] = (float)tid / 255.0f;    // one true division per value; the epilogue only looks up
    __syncthreads();
    for (int i = tid; i < R * E; i += NT) {
        const int row = i / E, e = i - row * E;
        float v = a.emb[(size_t)ids[row] * E + e];
        if (a.n_fonts > 0) v += a.femb[(size_t)fids[row] * E + e];
        const T w = cvt<T>(v);
        h0[row * ldE + e] = w;
        h0T[e * ldR + row] = w;
    }
    __syncthreads();

    // store one 16x16 accumulator tile (rows m0 + 4q + i, column n0 + r) in both orientations
    auto put_both = [&](T* rowm, int ldr, T* colm, int ldc, int m0, int n0, const float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rowm[(m0 + 4 * q + i) * ldr + n0 + r] = cvt<T>(v[i]);
        T* d = colm + (n0 + r) * ldc + m0 + 4 * q;
        if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(d) = (f32x4){v[0], v[1], v[2], v[3]};
        else *reinterpret_cast<bf16x4*>(d) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    };

    // ---- pre1 = h0 . W1^T + b1 ; h1 = relu(pre1)            wave w owns column tiles w, w+4, ...
    float bias_v = 0.f;
    auto loadW2o = [&](int nt, int kb) { return loadW2(pt0 + nt, kb); };      // (tile indices below are relative to pt0)
    if constexpr (PF) first_tile<T>(ftB, npt, N1 / KB, wave, loadW2o);   // fc_output's first tile arrives under fc1
    rows_times_global<T, MT, NW>(h0, ldE, N1 / 16, E / KB, wave, r, q, loadW1,
        [&](int nt) { bias_v = a.b1[nt * 16 + r]; },
        [&](int nt, const f32x4 (&acc)[MT]) {
            const float bias = bias_v;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[m][i] + bias, 0.f);
                put_both(h1, ldN, h1T, ldR, m * 16, nt * 16, v);
            }
        }, PF ? &ftA : nullptr);
    __syncthreads();

    G1STAMP();   // 1: gather + P1
    // ---- u = h1 . W2^T + b2 ; clamp, MSE, du (rows past the batch contribute nothing)
    float lsum = 0.f;
    const float g2 = 2.f * a.inv_n;
    rows_times_global_u<T, MT, NW, TPW>(h1, ldN, npt, N1 / KB, wave, r, q, loadW2o,
        [&](int nt, int) { bias_v = a.b2[(pt0 + nt) * 16 + r]; },
        [&](int nt, int it, const f32x4 (&acc)[MT]) {
            const float bias = bias_v;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m * 16 + 4 * q + i;
                    float u = acc[m][i] + bias;
                    if constexpr (sizeof(T) == 2) u = (float)(bf16_t)u;          // the value the unfused path stores
                    const float t = TU8 ? lut[__builtin_bit_cast(unsigned, tv[it][m][i])] : tv[it][m][i];
                    const float diff = fminf(fmaxf(u, 0.f), 1.f) - t;
                    const bool live = row < nb;
                    lsum += live ? diff * diff : 0.f;
                    v[i] = (live && u >= 0.f && u <= 1.f) ? g2 * diff : 0.f;
                }
                put_both(du, ldP, duT, ldR, m * 16, (pt0 + nt) * 16, v);
            }
        }, PF ? &ftB : nullptr);
    __syncthreads();
    G1STAMP();   // 2: P2 + loss
    // dh1's first W2^T tile is requested now and arrives under the dW2 phase (which reads LDS only)
    const T* W2T = reinterpret_cast<const T*>(a.W2T);
    const int kb0 = pt0 * 16 / KB, nkbp = npt * 16 / KB;          // this block's k-blocks of the reduction over P (npt * 16 is a multiple of KB)
    auto loadW2T = [&](int nt, int kb) {
        if constexpr (sizeof(T) == 4) return gather16(reinterpret_cast<const float*>(a.W2T), N1, nt * 16 + r, kb0 + kb, q);
        else return ld16(W2T, P, nt * 16 + r, kb0 + kb, q);
    };
    if constexpr (PF) first_tile<T>(ftA, N1 / 16, nkbp, wave, loadW2T);

    // ---- dW2[p][k] = sum_b du[b][p] h1[b][k]  (+ db2): wave w owns p tiles w, w+4, ...; 4 k tiles at a time
    // (operands swapped: the tile comes out as [k][p], so a lane holds 4 consecutive k of one row p -> one 16-byte store)
    for (int pt = pt0 + wave; pt < pt0 + npt; pt += NW) {
        for (int kc = 0; kc < N1 / 64; ++kc) {
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < R / KB; ++kb) {
                const i32x4 bv = ld16(duT, ldR, pt * 16 + r, kb, q);
#pragma unroll
                for (int j = 0; j < 4; ++j) M_::mma(acc[j], ld16(h1T, ldR, kc * 64 + j * 16 + r, kb, q), bv);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(slab + a.o_w2 + (size_t)(pt * 16 + r) * N1 + kc * 64 + j * 16 + 4 * q) = acc[j];
        }
    }
    for (int pp = pt0 * 16 + tid; pp < (pt0 + npt) * 16; pp += NT) {
        float s = 0.f;
        for (int b = 0; b < R; ++b) s += (float)duT[pp * ldR + b];
        slab[a.o_b2 + pp] = s;
    }
    __syncthreads();                     // every wave is done reading h1T (dW2) before dpre1 overwrites it below
    G1STAMP();   // 3: dW2 + db2

    // ---- dh1 = du . W2 (over this block's columns: a partial sum when cs > 1) ; dpre1 = dh1 * [h1 > 0], written over h1 / h1T
    // (each element is read and rewritten by its owner only)
    rows_times_global<T, MT, NW>(du + pt0 * 16, ldP, N1 / 16, nkbp, wave, r, q, loadW2T,
        [](int) {},
        [&](int nt, const f32x4 (&acc)[MT]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float h = (float)h1[(m * 16 + 4 * q + i) * ldN + nt * 16 + r];
                    v[i] = h > 0.f ? acc[m][i] : 0.f;
                }
                put_both(h1, ldN, h1T, ldR, m * 16, nt * 16, v);
            }
        }, PF ? &ftA : nullptr);
    __syncthreads();
    G1STAMP();   // 4: dh1

    // ---- dW1[n][e] = sum_b dpre1[b][n] h0[b][e]  (+ db1)
    for (int t = wave; t < (N1 / 16) * (E / 16); t += NW) {
        const int nt = t / (E / 16), et = t - nt * (E / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < R / KB; ++kb) M_::mma(acc, ld16(h0T, ldR, et * 16 + r, kb, q), ld16(h1T, ldR, nt * 16 + r, kb, q));
        *reinterpret_cast<f32x4*>(slab + a.o_w1 + (size_t)(nt * 16 + r) * E + et * 16 + 4 * q) = acc;       // [e][n] tile: 4 consecutive e per lane
    }
    for (int n = tid; n < N1; n += NT) {
        float s = 0.f;
        for (int b = 0; b < R; ++b) s += (float)h1T[n * ldR + b];
        slab[a.o_b1 + n] = s;
    }

    G1STAMP();   // 5: dW1 + db1
    // ---- dh0 = dpre1 . W1, kept transposed ([E][R], operand type) in the du area; then embedding_dense_backward as one more
    // product, dTab[v][e] = sum_b onehot[b][v] dh0[b][e] (A = dh0T rows e, B = onehotT rows v; the one-hot image is built in
    // LDS: row v has a 1 at every batch row that used table row v).  Fixed summation order: bitwise reproducible.  (A serial
    // walk over the block's rows, as glyph_embed_bwd_kernel does, cost 8 us of this kernel at 64 rows.)
    T* dh0T = du;                                             // [E][ldR]          (du, duT are dead: dh1, dW2, db2 are complete)
    T* ohT = du + E * ldR;                                    // [VT][ldR], VT = table rows rounded up to 16
    const int rows_tot = a.vocab + a.n_fonts, VT = (rows_tot + 15) & ~15;
    for (int i = tid; i < VT * ldR * (int)sizeof(T) / 16; i += NT) reinterpret_cast<i32x4*>(ohT)[i] = (i32x4){0, 0, 0, 0};
    rows_times_global<T, MT, NW>(h1, ldN, E / 16, N1 / KB, wave, r, q,
        [&](int et, int kb) {
            if constexpr (sizeof(T) == 4) return gather16(reinterpret_cast<const float*>(a.W1T), E, et * 16 + r, kb, q);
            else return ld16(reinterpret_cast<const T*>(a.W1T), N1, et * 16 + r, kb, q);
        },
        [](int) {},
        [&](int et, const f32x4 (&acc)[MT]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                T* d = dh0T + (et * 16 + r) * ldR + m * 16 + 4 * q;
                if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(d) = acc[m];
                else *reinterpret_cast<bf16x4*>(d) = (bf16x4){(bf16_t)acc[m][0], (bf16_t)acc[m][1], (bf16_t)acc[m][2], (bf16_t)acc[m][3]};
            }
        });
    __syncthreads();
    G1STAMP();   // 6: dh0
    if (tid < nb) {
        ohT[ids[tid] * ldR + tid] = cvt<T>(1.f);
        if (a.n_fonts > 0) ohT[(a.vocab + fids[tid]) * ldR + tid] = cvt<T>(1.f);
    }
    __syncthreads();
    for (int t = wave; t < (VT / 16) * (E / 16); t += NW) {
        const int vt = t / (E / 16), et = t - vt * (E / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < R / KB; ++kb) M_::mma(acc, ld16(dh0T, ldR, et * 16 + r, kb, q), ld16(ohT, ldR, vt * 16 + r, kb, q));
        const int v = vt * 16 + r;                            // [e][v] tile: this lane has 4 consecutive e of table row v
        if (v < a.vocab) *reinterpret_cast<f32x4*>(slab + a.o_emb + (size_t)v * E + et * 16 + 4 * q) = acc;
        else if (v < rows_tot) *reinterpret_cast<f32x4*>(slab + a.o_font + (size_t)(v - a.vocab) * E + et * 16 + 4 * q) = acc;
    }

    G1STAMP();   // 7: scatter + table store
#ifdef AFR_G1_DEBUG
    if (blockIdx.x == 0 && tid == 0) {
        printf("entry->first stamp %.2f us; phases (us):", (double)(tstamp[0] - t_entry) * 0.01);
        for (int i = 1; i < nst; ++i) printf(" %.2f", (double)(tstamp[i] - tstamp[i - 1]) * 0.01);
        printf("\n");
    }
#endif
    // ---- loss: block partial -> ticketed finish (fixed order)
    float* red = reinterpret_cast<float*>(h0);
    __syncthreads();
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    float bsum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) bsum += red[w];
    loss_block_finish(bsum, a.loss_partial, a.counter, a.loss_accum, a.inv_n, red + 16);
#ifdef AFR_G1_DEBUG
    if (blockIdx.x == 0 && tid == 0) printf("last stamp -> exit %.2f us; entry -> exit %.2f us\n", (double)(__builtin_amdgcn_s_memrealtime() - tstamp[nst - 1]) * 0.01,
                                            (double)(__builtin_amdgcn_s_memrealtime() - t_entry) * 0.01);
#endif
}

// W [N][K] f32 -> WT [K][N] bf16 (the transposed operand copies of the bf16 fused step)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* __restrict__ W, bf16_t* __restrict__ WT, int N, int K) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < (long long)N * K; i += (long long)gridDim.x * 256) {
        const int k = (int)(i / N), n = (int)(i - (long long)k * N);
        WT[i] = (bf16_t)W[(size_t)n * K + k];
    }
}

int afr_glyph1_rows(int dtype) { return dtype == AFR_BF16 ? MM<bf16_t>::R : MM<float>::R; }
// Blocks per row block: as many (4, 2, 1) as keep the launch within one round of the chip, with whole k-blocks of the
// dh1 reduction per block (P / cs a multiple of 32 bf16 / 16 f32 elements).  AFR_G1_CS overrides (kernel A/B measurements).
int afr_glyph1_colsplit(int dtype, int B, int P) {
    static const int force = getenv("AFR_G1_CS") ? atoi(getenv("AFR_G1_CS")) : 0;
    const int R = afr_glyph1_rows(dtype), kb = dtype == AFR_BF16 ? MM<bf16_t>::KB : MM<float>::KB, nrb = (B + R - 1) / R;
    int cs = force > 0 ? force : 4;
    while (cs > 1 && ((P / cs) % kb != 0 || (P / 16) % cs != 0 || (force <= 0 && nrb * cs > 256))) cs >>= 1;
    return cs;
}
int afr_glyph1_max_blocks(int dtype, int max_batch, int P) {
    const int R = afr_glyph1_rows(dtype), nrb = (max_batch + R - 1) / R;
    int m = nrb * afr_glyph1_colsplit(dtype, max_batch, P);
    // a smaller batch may split further: at most 4 blocks per row block, and (unless forced) never more than 256 blocks then
    const int alt = nrb * 4 < 256 ? nrb * 4 : 256;
    if (alt > m) m = alt;
    if (getenv("AFR_G1_CS")) m = nrb * 4;
    return m;
}
bool afr_glyph1_eligible(int E, int N1, int P, int vocab, int n_fonts) {
    return E % 32 == 0 && E <= 64 && N1 % 64 == 0 && N1 <= 256 && P % 64 == 0 && P <= 256 && vocab + n_fonts <= 264;
}
size_t afr_glyph1_lds_bytes(int dtype, int E, int N1, int P, int table_rows) {
    const size_t es = dtype == AFR_BF16 ? 2 : 4;
    const size_t R = afr_glyph1_rows(dtype), pad = PADB / es;
    const size_t head = 2 * R * sizeof(int) + 256 * sizeof(float) + ((size_t)R * (E + pad) + (size_t)E * (R + pad) + (size_t)R * (N1 + pad) + (size_t)N1 * (R + pad)) * es;
    const size_t tail = ((size_t)R * (P + pad) + (size_t)P * (R + pad)) * es;
    const size_t tail2 = ((size_t)E + (size_t)((table_rows + 15) & ~15)) * (R + pad) * es;   // dh0T + one-hot image, same place
    return head + (tail > tail2 ? tail : tail2);
}
hipError_t afr_launch_transpose_bf16(const float* W, bf16_t* WT, int N, int K, hipStream_t s) {
    const long long n = (long long)N * K;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256)), dim3(256), 0, s, W, WT, N, K);
    return hipGetLastError();
}
hipError_t afr_launch_glyph1_step(int dtype, const Glyph1Args& a, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    const int R = afr_glyph1_rows(dtype);
    const size_t lds = afr_glyph1_lds_bytes(dtype, a.E, a.N1, a.P, a.vocab + a.n_fonts);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static size_t set16[16], set32[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool u8 = a.tdtype == AFR_TARGET_U8;
    const void* kern = dtype == AFR_BF16 ? (u8 ? (const void*)glyph1_step_kernel<bf16_t, true> : (const void*)glyph1_step_kernel<bf16_t, false>)
                                         : (u8 ? (const void*)glyph1_step_kernel<float, true> : (const void*)glyph1_step_kernel<float, false>);
    size_t* done = (dtype == AFR_BF16 ? set16 : set32) + (u8 ? 0 : 8);          // devices 0..7 per (dtype, target type)
    if (lds > 48 * 1024 && (dev < 0 || dev >= 8 || done[dev] < lds)) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 8) done[dev] = lds;
    }
    if (a.cs < 1 || (a.P / 16) % a.cs) return hipErrorInvalidValue;
    const int nblk = (a.B + R - 1) / R * a.cs;
    const int NTL = dtype == AFR_BF16 ? MM<bf16_t>::NTH : MM<float>::NTH;
    if (dtype == AFR_BF16) {
        if (u8) hipLaunchKernelGGL((glyph1_step_kernel<bf16_t, true>), dim3(nblk), dim3(NTL), lds, s, a);
        else hipLaunchKernelGGL((glyph1_step_kernel<bf16_t, false>), dim3(nblk), dim3(NTL), lds, s, a);
    } else {
        if (u8) hipLaunchKernelGGL((glyph1_step_kernel<float, true>), dim3(nblk), dim3(NTL), lds, s, a);
        else hipLaunchKernelGGL((glyph1_step_kernel<float, false>), dim3(nblk), dim3(NTL), lds, s, a);
    }
    return hipGetLastError();
}
