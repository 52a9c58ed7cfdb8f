// sheet.hip -- the sheet model's per-string front end, fused into one kernel per direction.
//
// Forward (reference model.py:167-193): embedding gather -> dropout -> + learned positions -> packed in-proj ->
// 4-head 100x100 softmax attention (dropout on the probabilities) -> out-proj -> residual + LayerNorm ->
// fc1 + ReLU + dropout -> flattened (zero-padded) feature row z[b][max_length*64], the A operand of fc_output.
// Backward (model.py:309) rebuilds that chain in LDS -- the attention output and softmax row statistics come from the
// training forward (16 KB per string), everything else is recomputed from the same counter-hash dropout stream -- and
// then walks it in reverse (a sample's whole state is < 150 KB of LDS).
//
// One 1024-thread workgroup owns one string at a time and loops over strings (persistent grid, <= 256 blocks):
// weights are loaded into LDS once.  The small matmuls (in-proj, out-proj, fc1 and their data/weight gradients) run on
// the matrix cores straight from LDS (exact-f32 v_mfma_f32_16x16x4_f32); the 100x100 attention stays on the VALU.
// Every reduction has a single owner thread and a fixed order -- no atomics:
//   * attention backward is split by ROW (delta, dq) and then by COLUMN (dk, dv), each recomputing the scores it
//     needs, so dv/dk need no scatter;
//   * the small parameter gradients accumulate across the block's strings in registers / MFMA accumulators; the
//     embedding gradient follows each code's occurrence chain and is added to the block's slab in L2; every block
//     leaves one partial slab, summed in block order by the grouped reduce.
// E=32, 4 heads of 8, fc1 width 64 are compile-time (the reference hard-codes them: model.py:79,81,148).
#include "afr_common.h"
#include "../../include/afr.h"

namespace {
constexpr int E = 32, H = 4, D = 8, F = 64, QKV = 96;
constexpr int SE = 33, SQ = 100, SD = 36;       // LDS row strides: SE odd (conflict-free column walks); SQ, SD multiples of 4
                                                // so that a head's 8 q/k/v/dO values are two aligned 16-byte LDS reads
#ifndef AFR_SHEET_NT
#define AFR_SHEET_NT 1024
#endif
constexpr int NT = AFR_SHEET_NT;                // threads per workgroup: many waves hide the LDS/FMA latencies of the serial phases
constexpr int NG = NT / 32;                     // 32-lane column groups (accumulator ownership)
static_assert(NT == 1024, "the weight-gradient tile ownership below assumes 16 waves per workgroup");
constexpr int W_FLOATS = QKV * SE + QKV + E * SE + E + E + E + F * SE + F;   // weights block
// LDS buffers sit at fixed offsets sized for the longest supported string (LMAX): a block owns its CU's LDS anyway, and
// compile-time addresses keep ~15 pointers out of the register file (the backward kernel is register-bound at 128 VGPRs).
constexpr int LMAX = 120;
static_assert(LMAX <= 128, "the saved attention-dropout keep bits hold 128 keys per (head, row): 4 words, word (j&1)*2 + (j>>6)");
constexpr int al4c(int n) { return (n + 3) & ~3; }
constexpr int O_E = W_FLOATS, O_BIG = O_E + al4c(LMAX * SE), O_O = O_BIG + LMAX * SQ, O_XH = O_O + al4c(LMAX * SE),
              O_DN = O_XH + LMAX * SD, O_RSTD = O_DN + al4c(LMAX * SE), O_SMAX = O_RSTD + al4c(LMAX), O_SINV = O_SMAX + H * LMAX,
              O_SDEL = O_SINV + H * LMAX, O_REDA = O_SDEL + H * LMAX, O_REDB = O_REDA + 256, O_TOK = O_REDB + 256,
              O_NXT = O_TOK + LMAX, O_MB = O_NXT + LMAX, BWD_FLOATS = O_MB + 4 * H * LMAX;
// forward: e | qkv | o | r | n | rstd | smax | sinv | tok | attention keep bits
constexpr int F_E = W_FLOATS, F_QKV = F_E + al4c(LMAX * SE), F_O = F_QKV + LMAX * SQ, F_R = F_O + al4c(LMAX * SE),
              F_N = F_R + al4c(LMAX * SE), F_RSTD = F_N + al4c(LMAX * SE), F_SMAX = F_RSTD + al4c(LMAX), F_SINV = F_SMAX + H * LMAX,
              F_TOK = F_SINV + H * LMAX, F_MB = F_TOK + LMAX, FWD_FLOATS = F_MB + 4 * H * LMAX;
static_assert(BWD_FLOATS * 4 <= 160 * 1024 && FWD_FLOATS * 4 <= 160 * 1024, "one string's state must fit the CU's LDS");

struct Wts { float *Win, *bin, *Wo, *bo, *lg, *lb, *W1, *b1; };

__device__ __forceinline__ Wts carve_weights(float* sm) {
    Wts w;
    w.Win = sm; w.bin = w.Win + QKV * SE; w.Wo = w.bin + QKV; w.bo = w.Wo + E * SE;
    w.lg = w.bo + E; w.lb = w.lg + E; w.W1 = w.lb + E; w.b1 = w.W1 + F * SE;
    return w;
}
__device__ __forceinline__ void load_weights(const Wts& w, const SheetParams& P, int tid) {
    for (int i = tid; i < QKV * E; i += NT) w.Win[(i >> 5) * SE + (i & 31)] = P.w_in[i];
    for (int i = tid; i < E * E; i += NT) w.Wo[(i >> 5) * SE + (i & 31)] = P.w_o[i];
    for (int i = tid; i < F * E; i += NT) w.W1[(i >> 5) * SE + (i & 31)] = P.w1[i];
    if (tid < QKV) w.bin[tid] = P.b_in[tid];
    if (tid < E) { w.bo[tid] = P.b_o[tid]; w.lg[tid] = P.ln_g[tid]; w.lb[tid] = P.ln_b[tid]; }
    if (tid < F) w.b1[tid] = P.b1[tid];
}

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float dot8(const float (&a)[8], const float (&b)[8]) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) s = fmaf(a[d], b[d], s);
    return s;
}
__device__ __forceinline__ int al4(int n) { return (n + 3) & ~3; }

// Small LDS-resident matmul on the matrix cores: out(m,n) = sum_k A(m,k) * B(n,k), m<M, n<N, K a multiple of 4, with
// A(m,k) = A[m*sa_m + k*sa_k] and B(n,k) = B[n*sb_n + k*sb_k] (any orientation: LDS is indexed freely).  16x16 output
// tiles are dealt round-robin to the block's waves; each is a chain of exact-f32 v_mfma_f32_16x16x4_f32 (bitwise a
// k-ordered fma chain).  Rows/columns past the edge are clamped on load (an MFMA output row depends only on its own
// A row) and dropped by the epilogue bounds.  epi(m, n, value) is called once per valid output element.
template <class Epi>
__device__ __forceinline__ void lds_mma(const float* A, int sa_m, int sa_k, const float* B, int sb_n, int sb_k, int M, int N,
                                        int K, int tid, Epi epi) {
    const int wave = tid >> 6, lane = tid & 63;
    const int tn = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn;
    const int kq = lane >> 4, l15 = lane & 15;
    for (int t = wave; t < tiles; t += NT / 64) {
        const int m0 = (t / tn) << 4, n0 = (t % tn) << 4;
        const float* ap = A + min(m0 + l15, M - 1) * sa_m + kq * sa_k;
        const float* bp = B + min(n0 + l15, N - 1) * sb_n + kq * sb_k;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k * sa_k], bp[k * sb_k], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * kq + r, n = n0 + l15;
            if (m < M && n < N) epi(m, n, acc[r]);
        }
    }
}

// One persistent 16x16 accumulator tile: acc += sum_{k<K} A(m0+i, k) * B(n0+j, k), any K (tail guarded).  Used for the
// small weight gradients, whose accumulators stay in MFMA registers across all strings a block processes.
__device__ __forceinline__ void lds_mma_tile(f32x4& acc, const float* A, int sa_m, int sa_k, const float* B, int sb_n, int sb_k,
                                             int m0, int n0, int K, int lane) {
    const int kq = lane >> 4, l15 = lane & 15;
    const float* ap = A + (m0 + l15) * sa_m + kq * sa_k;
    const float* bp = B + (n0 + l15) * sb_n + kq * sb_k;
    const int K4 = K & ~3;
    for (int k = 0; k < K4; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k * sa_k], bp[k * sb_k], acc, 0, 0, 0);
    if (K4 < K) {
        const bool ok = K4 + kq < K;
        const float a = ok ? ap[K4 * sa_k] : 0.f, bq = ok ? bp[K4 * sb_k] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
    }
}

// ---- shared forward pieces (used by both kernels so that backward's recomputation is bit-identical) ----------
__device__ __forceinline__ void ph_tokens(int* tok, const int64_t* x, int ldx, int b, int L, int vocab, uint32_t* err, int tid) {
    if (tid < L) {
        long long v = x[(size_t)b * ldx + tid];
        if (v < 0 || v >= vocab) { if (err) atomicOr(err, 1u); v = v < 0 ? 0 : vocab - 1; }
        tok[tid] = (int)v;
    }
}
// e = dropout(Emb[x]) + P[:L]                                                  (model.py:167-172)
__device__ __forceinline__ void ph_embed(float* e, const int* tok, const SheetParams& P, const SheetDrop& dr, int b, int L, int tid) {
    for (int i = tid; i < L * E; i += NT) {
        const int l = i >> 5, c = i & 31;
        float v = P.emb[tok[l] * E + c];
        if (dr.dbg_e0) dr.dbg_e0[(size_t)b * L * E + i] = v;
        if (dr.training) v = afr_keep((uint64_t)b * L * E + i, dr.key_e, dr.thr_e) ? v * dr.sc_e : 0.f;
        e[l * SE + c] = v + P.pos[i];
    }
}
// qkv = e . W_in^T + b_in                                                       (packed in-proj of nn.MultiheadAttention)
__device__ __forceinline__ void ph_inproj(float* qkv, const float* e, const Wts& w, int L, int tid) {
    lds_mma(e, SE, 1, w.Win, SE, 1, L, QKV, E, tid, [&](int m, int n, float v) { qkv[m * SQ + n] = v + w.bin[n]; });
}
__device__ __forceinline__ float attn_mask(const SheetDrop& dr, int b, int h, int i, int j, int L) {
    if (!dr.training) return 1.f;
    const uint64_t idx = (((uint64_t)b * H + h) * L + i) * L + j;
    return afr_keep(idx, dr.key_a, dr.thr_a) ? dr.sc_a : 0.f;
}
// o = concat_h( dropout(softmax((q/sqrt(D)) k^T)) v )      two adjacent lanes per (head, query row), keys split even/odd
// mbits (optional, training only): the dropout keep decisions, one bit per probability, stored as the lanes produce
// them: row (h,i), key j = 2t + part -> bit t&31 of mbits[((h*L+i)*2 + part)*2 + (t>>5)].  The backward passes read
// the bit instead of hashing the element's counter again.
__device__ __forceinline__ void ph_attention(float* o, const float* qkv, const SheetDrop& dr, int b, int L, int tid,
                                             float* smax_out = nullptr, float* sinv_out = nullptr, uint32_t* mbits = nullptr) {
    const float scale = 0.35355339059327373f;    // sqrt(1/8), applied to q as torch does
    for (int rr = tid; rr < 2 * H * L; rr += NT) {
        const int r = rr >> 1, part = rr & 1;
        const int h = r / L, i = r - h * L;
        float q[D], kv[D];
        ld8(qkv + i * SQ + h * D, q);
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] *= scale;
        float mx = -INFINITY;
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
        for (int j = part; j < L; j += 2) {
            ld8(qkv + j * SQ + E + h * D, kv);
            mx = fmaxf(mx, dot8(q, kv));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        float sum = 0.f, acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = 0.f;
        // keep bits of this lane's keys (j = 2t + part): bit t&31 of wb[t>>5].  Two plain loops (keys below / from 64) so
        // that each unrolls by two like the loop without the bits; no cross-lane operation inside them.
        uint32_t wb[2] = {0u, 0u};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t bit = 1u, acc_bits = 0u;
            const int jend = half ? L : min(L, 64);
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
            for (int j = 64 * half + part; j < jend; j += 2) {
                ld8(qkv + j * SQ + E + h * D, kv);
                const float p = __expf(dot8(q, kv) - mx);
                sum += p;
                const float am = attn_mask(dr, b, h, i, j, L);
                acc_bits |= am != 0.f ? bit : 0u;
                bit <<= 1;
                const float pm = p * am;
                ld8(qkv + j * SQ + 2 * E + h * D, kv);
#pragma unroll
                for (int d = 0; d < D; ++d) acc[d] = fmaf(pm, kv[d], acc[d]);
            }
            wb[half] = acc_bits;
        }
        if (mbits) *reinterpret_cast<uint2*>(mbits + rr * 2) = make_uint2(wb[0], wb[1]);
        sum += __shfl_xor(sum, 1, 64);
        const float inv = 1.f / sum;
        if (smax_out && part == 0) { smax_out[r] = mx; sinv_out[r] = inv; }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float a = acc[d] + __shfl_xor(acc[d], 1, 64);
            if (part == 0) o[i * SE + h * D + d] = a * inv;
        }
    }
}
// r = e + o . W_o^T + b_o                                                       (model.py:176-180)
__device__ __forceinline__ void ph_outproj_res(float* r, const float* e, const float* o, const Wts& w, int L, int tid) {
    lds_mma(o, SE, 1, w.Wo, SE, 1, L, E, E, tid, [&](int m, int n, float v) { r[m * SE + n] = e[m * SE + n] + v + w.bo[n]; });
}
// LayerNorm over the 32 channels, biased variance: xh <- (r-mu)*rstd in place, n <- xh*gamma+beta, rstd kept
__device__ __forceinline__ float sum8(float v) {           // sum over the 8 adjacent lanes that share a row
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ void ph_layernorm(float* xh, float* n, float* rstd, const Wts& w, int L, float eps, int tid) {
    // 8 lanes per row, 4 channels each (L <= 120 rows fit the block's 1024 threads)
    const int l = tid >> 3, c0 = (tid & 7) * 4;
    if (l < L) {
        float* row = xh + l * SE + c0;
        float v[4] = {row[0], row[1], row[2], row[3]};
        const float mu = sum8((v[0] + v[1]) + (v[2] + v[3])) * (1.f / E);
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] -= mu; var = fmaf(v[j], v[j], var); }
        var = sum8(var) * (1.f / E);
        const float rs = 1.f / sqrtf(var + eps);
        if ((tid & 7) == 0) rstd[l] = rs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = v[j] * rs;
            row[j] = xv;
            n[l * SE + c0 + j] = fmaf(xv, w.lg[c0 + j], w.lb[c0 + j]);
        }
    }
}
__device__ __forceinline__ float fc_mask(const SheetDrop& dr, int b, int L, int i) {
    if (!dr.training) return 1.f;
    return afr_keep((uint64_t)b * L * F + i, dr.key_f, dr.thr_f) ? dr.sc_f : 0.f;
}
// ------------------------------------------------------------------------------------------ forward kernel
template <typename T>
__global__ __launch_bounds__(NT) void sheet_fwd_kernel(SheetDims dm, SheetParams P, SheetDrop dr, const int64_t* __restrict__ x,
                                                        int ldx, int B, T* __restrict__ z, float eps, uint32_t* err) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid0 = threadIdx.x, L = dm.L;
    const Wts w = carve_weights(sm);
    float* const e = sm + F_E;
    float* const qkv = sm + F_QKV;
    float* const o = sm + F_O;
    float* const r = sm + F_R;
    float* const n = sm + F_N;
    float* const rstd = sm + F_RSTD;
    float* const smax = sm + F_SMAX;   // [4L]
    float* const sinv = sm + F_SINV;   // [4L]
    int* const tok = reinterpret_cast<int*>(sm + F_TOK);
    uint32_t* const mb = reinterpret_cast<uint32_t*>(sm + F_MB);   // [4L][4] attention-dropout keep bits
    load_weights(w, P, tid0);
    const size_t Kz = (size_t)dm.Lmax * F;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int tid = tid0;                                             // opaque per iteration: see sheet_bwd_kernel
        asm volatile("" : "+v"(tid));
        ph_tokens(tok, x, ldx, b, L, dm.vocab, err, tid);
        __syncthreads();
        ph_embed(e, tok, P, dr, b, L, tid);
        __syncthreads();
        ph_inproj(qkv, e, w, L, tid);
        __syncthreads();
        ph_attention(o, qkv, dr, b, L, tid, dr.save ? smax : nullptr, sinv, (dr.save && dr.training) ? mb : nullptr);
        __syncthreads();
        if (dr.save) {                                              // keep o and the softmax statistics for backward
            float* sv = dr.save + (size_t)b * L * AFR_SHEET_SAVE_PER_POS;
            for (int i = tid; i < L * E; i += NT) sv[i] = o[(i >> 5) * SE + (i & 31)];
            for (int i = tid; i < H * L; i += NT) { sv[L * E + i] = smax[i]; sv[L * E + H * L + i] = sinv[i]; }
            if (dr.training)
                for (int i = tid; i < 4 * H * L; i += NT) reinterpret_cast<uint32_t*>(sv)[L * 40 + i] = mb[i];
        }
        ph_outproj_res(r, e, o, w, L, tid);
        __syncthreads();
        ph_layernorm(r, n, rstd, w, L, eps, tid);
        __syncthreads();
        T* zr = z + (size_t)b * Kz;
        float* fbuf = qkv;                                          // qkv is dead: [L][64] ReLU outputs, then a coalesced copy-out
        lds_mma(n, SE, 1, w.W1, SE, 1, L, F, E, tid, [&](int m, int j, float v) { fbuf[m * F + j] = fmaxf(v + w.b1[j], 0.f); });
        __syncthreads();
        for (int i = tid; i < L * F; i += NT) zr[i] = (T)(fbuf[i] * fc_mask(dr, b, L, i));      // fc1 + ReLU + dropout, model.py:183-184
        for (size_t i = (size_t)L * F + tid; i < Kz; i += NT) zr[i] = (T)0.f;   // zero-pad branch, model.py:190-193
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------- backward kernel
__device__ __forceinline__ float dqkv_at(const float* dq, const float* big, int l, int j) {
    return j < E ? dq[l * SE + j] : big[l * SQ + j];
}

// Phase timing for kernel development: build with -DAFR_SHEET_TIMING and block 0 prints per-phase totals (10 ns ticks).
#ifdef AFR_SHEET_TIMING
#define TMARK(k) do { __syncthreads(); if (threadIdx.x == 0) { const long long t_ = wall_clock64(); tacc[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define TMARK(k) do { } while (0)
#endif
// partial column sums of an [L][ncols] LDS array: thread t < ncols*P owns column t%ncols and rows t/ncols, +P, ... ; the
// P partials per column land in red[part*ncols + c] and are added by the column's owner after the next barrier
template <class F>
__device__ __forceinline__ void col_partials(float* red, int t, int ncols, int P, int L, F at) {
    if (t < 0 || t >= ncols * P) return;
    const int c = t % ncols, part = t / ncols;
    float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
    for (int l = part; l < L; l += P) a += at(l, c);
    red[part * ncols + c] = a;
}
__device__ __forceinline__ float col_total(const float* red, int c, int ncols, int P) {
    float a = 0.f;
    for (int q = 0; q < P; ++q) a += red[q * ncols + c];
    return a;
}

template <typename T>
__global__ __launch_bounds__(NT) void sheet_bwd_kernel(SheetDims dm, SheetParams P, SheetDrop dr, const int64_t* __restrict__ x,
                                                        int ldx, int B, const T* __restrict__ dz, float eps,
                                                        float* __restrict__ slabs, SheetSlabOff so) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid0 = threadIdx.x, L = dm.L;
    const Wts w = carve_weights(sm);
    float* const e = sm + O_E;         // [L][33]   whole sample
    float* const big = sm + O_BIG;     // [L][100]  qkv  | later n [L][33] + df [L][64] | later qkv again -> dk,dv in place
    float* const o = sm + O_O;         // [L][33]   o    | later dq
    float* const xh = sm + O_XH;       // [L][36]   r -> xhat (stride 33) | later dO, grad wrt attention output (stride 36)
    float* const dn = sm + O_DN;       // [L][33]   dn -> dr -> de
    float* const rstd = sm + O_RSTD;   // [L]
    float* const smax = sm + O_SMAX;   // [4L]
    float* const sinv = sm + O_SINV;   // [4L]
    float* const sdel = sm + O_SDEL;   // [4L]
    float* const redA = sm + O_REDA;   // [256] partial column sums (bias gradients), two buffers used alternately
    float* const redB = sm + O_REDB;
    int* const tok = reinterpret_cast<int*>(sm + O_TOK);   // [L]
    int* const nxt = reinterpret_cast<int*>(sm + O_NXT);   // [L] occurrence chain: low 16 bits = 1 + next position with the
                                       //     same code (0: none), bit 16 = this is the code's first occurrence in the string
    uint32_t* const mb = reinterpret_cast<uint32_t*>(sm + O_MB);   // [4L][4] attention-dropout keep bits (from the forward)
    float* const nbuf = big;               // n  [L][33]
    float* const df = big + LMAX * SE;     // df [L][64]
    load_weights(w, P, tid0);

    constexpr int KPOS = (120 + NG - 1) / NG;         // rows of dP (and (row, channel) items of a string) per thread
    float aPos[KPOS];
    // weight-gradient tiles (16x16, MFMA accumulators, persistent across strings): 8 of dW1 [64x32], 4 of dWo [32x32],
    // 12 of dWin [96x32].  Wave w owns tile w (accA); waves 0..7 also own dWin tile w+4 (accB).
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    float a_b1 = 0.f, a_bo = 0.f, a_g = 0.f, a_b = 0.f, a_bin = 0.f;
#pragma unroll
    for (int k = 0; k < KPOS; ++k) aPos[k] = 0.f;
    const float scale = 0.35355339059327373f;
    const float keep_sc = dr.training ? dr.sc_a : 1.f;      // value of a kept attention probability's dropout mask
    const size_t Kz = (size_t)dm.Lmax * F;
    float* Sblk = slabs + (size_t)blockIdx.x * so.total;       // this block's partial slab
    // The block zeroes its own slab (55 KB, L2): the embedding rows are accumulated into it string by string, rows of dP
    // beyond a short input and the padding between tensors must read as zero in the grouped reduce.  (This used to be a
    // 14 MB host-side memset launch in front of every backward.)  The first accumulation is many barriers away.
    for (int i = tid0; i < so.total / 4; i += NT) reinterpret_cast<float4*>(Sblk)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool saved = dr.save != nullptr;
#ifdef AFR_SHEET_TIMING
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#endif

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // Every per-thread LDS address below depends only on the thread id and L, i.e. is invariant across this loop; left
        // alone the compiler hoists all of them out of it and spills >100 registers around every phase.  An opaque copy of
        // the thread id per iteration keeps those values local to the phase that uses them.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int c32 = tid & 31, g8 = tid >> 5, wave = tid >> 6, lane = tid & 63;     // g8: column group 0..NG-1
        TMARK(11);
        // ---- codes, then the embedded string; o and the softmax statistics come from the training forward
        ph_tokens(tok, x, ldx, b, L, dm.vocab, nullptr, tid);
        if (saved) {
            const float* sv = dr.save + (size_t)b * L * AFR_SHEET_SAVE_PER_POS;
            for (int i = tid; i < L * E; i += NT) o[(i >> 5) * SE + (i & 31)] = sv[i];
            for (int i = tid; i < H * L; i += NT) { smax[i] = sv[L * E + i]; sinv[i] = sv[L * E + H * L + i]; }
            if (dr.training)
                for (int i = tid; i < 4 * H * L; i += NT) mb[i] = reinterpret_cast<const uint32_t*>(sv)[L * 40 + i];
        }
        __syncthreads();
        ph_embed(e, tok, P, dr, b, L, tid);
        {   // occurrence chains of the codes (for the embedding gradient): 8 lanes per position
            const int l = tid >> 3, k = tid & 7;
            if (l < L) {
                const int v = tok[l];
                int nx = 0x7fff, pv = 0;
                for (int m = l + 1 + k; m < L; m += 8) if (tok[m] == v) { nx = m; break; }
                for (int m = l - 1 - k; m >= 0; m -= 8) if (tok[m] == v) { pv = 1; break; }
                nx = min(nx, __shfl_xor(nx, 1, 64)); nx = min(nx, __shfl_xor(nx, 2, 64)); nx = min(nx, __shfl_xor(nx, 4, 64));
                pv |= __shfl_xor(pv, 1, 64); pv |= __shfl_xor(pv, 2, 64); pv |= __shfl_xor(pv, 4, 64);
                if (k == 0) nxt[l] = (nx == 0x7fff ? 0 : nx + 1) | (pv ? 0 : 1 << 16);
            }
        }
        __syncthreads();
        if (!saved) {                                              // no training forward ran: recompute o and the statistics
            ph_inproj(big, e, w, L, tid);
            __syncthreads();
            ph_attention(o, big, dr, b, L, tid, smax, sinv, dr.training ? mb : nullptr);
            __syncthreads();
        }
        TMARK(0);
        ph_outproj_res(xh, e, o, w, L, tid);
        __syncthreads();
        ph_layernorm(xh, nbuf, rstd, w, L, eps, tid);        // qkv dead: n overwrites the front of `big`
        __syncthreads();
        TMARK(1);
        // ---- df = dz * dropout-mask * [pre>0]
        const T* dzr = dz + (size_t)b * Kz;
        for (int i = tid; i < L * F; i += NT) df[i] = (float)dzr[i] * fc_mask(dr, b, L, i);
        __syncthreads();
        lds_mma(nbuf, SE, 1, w.W1, SE, 1, L, F, E, tid, [&](int m, int j, float v) { if (!(v + w.b1[j] > 0.f)) df[m * F + j] = 0.f; });
        __syncthreads();
        TMARK(2);
        // ---- dW1 += df^T n (persistent MFMA tiles, waves 0-7) ; db1 partials (waves 12-15) ; dn = df . W1
        if (wave < 8) lds_mma_tile(accA, df, 1, F, nbuf, 1, SE, 16 * (wave >> 1), 16 * (wave & 1), L, lane);
        col_partials(redA, tid - 768, F, 4, L, [&](int l, int c) { return df[l * F + c]; });
        lds_mma(df, F, 1, w.W1, 1, SE, L, E, F, tid, [&](int m, int c, float v) { dn[m * SE + c] = v; });
        __syncthreads();
        TMARK(3);
        // ---- LayerNorm backward: dgamma, dbeta partials (4 row groups x 64 columns), then dr in place (8 lanes per row)
        if (tid < F) a_b1 += col_total(redA, tid, F, 4);
        col_partials(redB, tid, 2 * E, 4, L, [&](int l, int c) {
            return c < E ? dn[l * SE + c] * xh[l * SE + c] : dn[l * SE + c - E]; });
        __syncthreads();
        if (tid < E) a_g += col_total(redB, tid, 2 * E, 4);
        else if (tid < 2 * E) a_b += col_total(redB, tid, 2 * E, 4);
        {
            const int l = tid >> 3, c0 = (tid & 7) * 4;
            if (l < L) {
                float* row = dn + l * SE + c0;
                const float* xr = xh + l * SE + c0;
                float gv[4], xv[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { gv[j] = row[j] * w.lg[c0 + j]; xv[j] = xr[j]; m1 += gv[j]; m2 = fmaf(gv[j], xv[j], m2); }
                m1 = sum8(m1) * (1.f / E); m2 = sum8(m2) * (1.f / E);
                const float rs = rstd[l];
#pragma unroll
                for (int j = 0; j < 4; ++j) row[j] = (gv[j] - m1 - xv[j] * m2) * rs;
            }
        }
        __syncthreads();
        TMARK(4);
        // ---- out-proj backward: dWo += dr^T o (waves 8-11) ; dbo partials (waves 0-1) ; dO = dr . Wo  (over xhat, which is dead)
        if (wave >= 8 && wave < 12) lds_mma_tile(accA, dn, 1, SE, o, 1, SE, 16 * ((wave - 8) >> 1), 16 * (wave & 1), L, lane);
        col_partials(redA, tid, E, 4, L, [&](int l, int c) { return dn[l * SE + c]; });
        lds_mma(dn, SE, 1, w.Wo, 1, SE, L, E, E, tid, [&](int m, int c, float v) { xh[m * SD + c] = v; });
        __syncthreads();                                     // n, df dead; o dead after the dWo loop above
        TMARK(5);
        if (tid < E) a_bo += col_total(redA, tid, E, 4);
        ph_inproj(big, e, w, L, tid);                        // recompute qkv
        __syncthreads();
        TMARK(6);
        // ---- attention backward, by ROW: delta = sum_j dA.A, dq      (dq -> `o` buffer)
        //      two adjacent lanes per (head, query row): keys split even/odd, combined with xor-1 shuffles
        for (int rr = tid; rr < 2 * H * L; rr += NT) {
            const int r = rr >> 1, part = rr & 1;
            const int h = r / L, i = r - h * L;
            float q[D], dO[D], kk[D], vv[D];
            ld8(big + i * SQ + h * D, q);
            ld8(xh + i * SD + h * D, dO);
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] *= scale;
            const float mx = smax[r];
            // one pass for everything that depends on the softmax row: with p~ = exp(s - max) (unnormalised),
            //   sum = S p~ ,  num = S p~ dA ,  T1 = S p~ dA k_j ,  T2 = S p~ k_j     (dA = (dO.v_j) * dropout mask)
            // then  delta = num/sum  and  dq = scale * (T1 - delta * T2) / sum  ==  scale * S_j A_ij (dA_ij - delta) k_j
            float sum = 0.f, num = 0.f, t1[D], t2[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { t1[d] = 0.f; t2[d] = 0.f; }
            const uint2 mw = dr.training ? *reinterpret_cast<const uint2*>(mb + rr * 2) : make_uint2(~0u, ~0u);   // this lane's keys
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // keys below / from 64: one word of keep bits each
                uint32_t bits = half ? mw.y : mw.x;
                const int jend = half ? L : min(L, 64);
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
                for (int j = 64 * half + part; j < jend; j += 2) {
                    ld8(big + j * SQ + E + h * D, kk);
                    ld8(big + j * SQ + 2 * E + h * D, vv);
                    const float p = __expf(dot8(q, kk) - mx);
                    const float pa = p * (dot8(dO, vv) * ((bits & 1u) ? keep_sc : 0.f));
                    bits >>= 1;
                    sum += p;
                    num += pa;
#pragma unroll
                    for (int d = 0; d < D; ++d) { t1[d] = fmaf(pa, kk[d], t1[d]); t2[d] = fmaf(p, kk[d], t2[d]); }
                }
            }
            sum += __shfl_xor(sum, 1, 64);
            num += __shfl_xor(num, 1, 64);
            const float inv = 1.f / sum, delta = num * inv;
            if (part == 0) { sinv[r] = inv; sdel[r] = delta; }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float a1 = t1[d] + __shfl_xor(t1[d], 1, 64);
                const float a2 = t2[d] + __shfl_xor(t2[d], 1, 64);
                if (part == 0) o[i * SE + h * D + d] = (a1 - delta * a2) * inv * scale;
            }
        }
        __syncthreads();
        TMARK(7);
        // ---- attention backward, by COLUMN: dk_j, dv_j; two lanes per (head, key), queries split even/odd.  The pair
        //      reads k_j, v_j into registers first and only then (after the shuffles) lane 0 overwrites them in place.
        for (int rr = tid; rr < 2 * H * L; rr += NT) {
            const int r = rr >> 1, part = rr & 1;
            const int h = r / L, j = r - h * L;
            float kk[D], vv[D], dk[D], dv[D], qs[D], dO[D];
            ld8(big + j * SQ + E + h * D, kk);
            ld8(big + j * SQ + 2 * E + h * D, vv);
            const int mword = (j & 1) * 2 + (j >> 6), mbit = (j >> 1) & 31;     // where key j's keep bit sits in a row's 4 words
#pragma unroll
            for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
            for (int i = part; i < L; i += 2) {
                ld8(big + i * SQ + h * D, qs);
                ld8(xh + i * SD + h * D, dO);
#pragma unroll
                for (int d = 0; d < D; ++d) qs[d] *= scale;
                const float p = __expf(dot8(qs, kk) - smax[h * L + i]) * sinv[h * L + i];
                const uint32_t bits = dr.training ? mb[(h * L + i) * 4 + mword] : 0xffffffffu;
                const float m = ((bits >> mbit) & 1u) ? keep_sc : 0.f;
                const float dS = p * (dot8(dO, vv) * m - sdel[h * L + i]);
                const float pm = p * m;
#pragma unroll
                for (int d = 0; d < D; ++d) { dk[d] = fmaf(dS, qs[d], dk[d]); dv[d] = fmaf(pm, dO[d], dv[d]); }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float a1 = dk[d] + __shfl_xor(dk[d], 1, 64);
                const float a2 = dv[d] + __shfl_xor(dv[d], 1, 64);
                if (part == 0) { big[j * SQ + E + h * D + d] = a1; big[j * SQ + 2 * E + h * D + d] = a2; }
            }
        }
        __syncthreads();
        TMARK(8);
        // ---- in-proj backward: dWin += dqkv^T e ; dbin partials (waves 8-11) ; de = dr + dqkv . Win   (de in place over dr)
        //      dqkv lives in two places: dq in `o` (columns 0..31), dk|dv in `big` (columns 32..95).
        //      The slab words this string's embedding gradient will be added to are fetched now and used after the barrier.
        float olde[KPOS];
#pragma unroll
        for (int k = 0; k < KPOS; ++k) {
            const int i = tid + k * NT;
            olde[k] = 0.f;
            if (i < L * E && (nxt[i >> 5] >> 16)) olde[k] = Sblk[so.emb + tok[i >> 5] * E + (i & 31)];
        }
        if (wave >= 12) {                                   // dWin tiles 0..3: rows j = 0..31 come from dq
            const int id = wave - 12;
            lds_mma_tile(accA, o, 1, SE, e, 1, SE, 16 * (id >> 1), 16 * (id & 1), L, lane);
        } else if (wave < 8) {                              // dWin tiles 4..11: rows j = 32..95 come from dk|dv
            const int id = wave + 4;
            lds_mma_tile(accB, big, 1, SQ, e, 1, SE, 16 * (id >> 1), 16 * (id & 1), L, lane);
        }
        col_partials(redB, tid - 512, QKV, 2, L, [&](int l, int c) { return dqkv_at(o, big, l, c); });
        lds_mma(o, SE, 1, w.Win, 1, SE, L, E, E, tid, [&](int m, int c, float v) { dn[m * SE + c] += v; });
        lds_mma(big + E, SQ, 1, w.Win + E * SE, 1, SE, L, E, 2 * E, tid, [&](int m, int c, float v) { dn[m * SE + c] += v; });
        __syncthreads();
        TMARK(9);
        // ---- dP += de ; dEmb[code] += sum over the code's occurrences of de * embed-dropout-mask.  The (first occurrence,
        //      channel) item owns the sum, walks the chain in position order and adds it to the block's slab in HBM/L2.
        if (tid < QKV) a_bin += col_total(redB, tid, QKV, 2);
#pragma unroll
        for (int k = 0; k < KPOS; ++k) { const int l = g8 + NG * k; if (l < L) aPos[k] += dn[l * SE + c32]; }
#pragma unroll
        for (int k = 0; k < KPOS; ++k) {
            const int i = tid + k * NT;
            if (i < L * E && (nxt[i >> 5] >> 16)) {
                const int c = i & 31;
                float sum = 0.f;
                for (int m = i >> 5;;) {
                    float mk = 1.f;
                    if (dr.training) mk = afr_keep((uint64_t)b * L * E + m * E + c, dr.key_e, dr.thr_e) ? dr.sc_e : 0.f;
                    sum += dn[m * SE + c] * mk;
                    const int nn = nxt[m] & 0xffff;
                    if (!nn) break;
                    m = nn - 1;
                }
                Sblk[so.emb + tok[i >> 5] * E + c] = olde[k] + sum;
            }
        }
        __syncthreads();
    }
    TMARK(10);
#ifdef AFR_SHEET_TIMING
    if (blockIdx.x == 0 && tid0 == 0) { printf("sheet_bwd phases (x10ns):"); for (int k = 0; k < 12; ++k) printf(" %lld", tacc[k]); printf("\n"); }
#endif
    // ---- one partial slab per block, laid out like the flat parameter buffer (pads were zeroed by the host memset)
    const int tid = tid0, c32 = tid & 31, g8 = tid >> 5, wave = tid >> 6, lane = tid & 63;
    float* S = Sblk;
#pragma unroll
    for (int k = 0; k < KPOS; ++k) { const int l = g8 + NG * k; if (l < L) S[so.pos + l * E + c32] = aPos[k]; }
    {   // MFMA D layout: row = 4*(lane>>4) + r, col = lane&15
        const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (wave < 8) {
                S[so.w1 + (16 * (wave >> 1) + 4 * kq + r) * E + 16 * (wave & 1) + l15] = accA[r];
                const int id = wave + 4;
                S[so.win + (16 * (id >> 1) + 4 * kq + r) * E + 16 * (id & 1) + l15] = accB[r];
            } else if (wave < 12) {
                S[so.wo + (16 * ((wave - 8) >> 1) + 4 * kq + r) * E + 16 * (wave & 1) + l15] = accA[r];
            } else {
                const int id = wave - 12;
                S[so.win + (16 * (id >> 1) + 4 * kq + r) * E + 16 * (id & 1) + l15] = accA[r];
            }
        }
    }
    if (tid < QKV) S[so.bin + tid] = a_bin;
    if (tid < E) { S[so.bo + tid] = a_bo; S[so.g + tid] = a_g; }
    else if (tid < 2 * E) S[so.b + tid - E] = a_b;
    if (tid < F) S[so.b1 + tid] = a_b1;
}
}  // namespace

int afr_sheet_blocks(int B) { return B < 256 ? B : 256; }
static inline int al4h(int n) { return (n + 3) & ~3; }
size_t afr_sheet_fwd_lds_bytes(const SheetDims& d) { return d.L <= LMAX ? (size_t)FWD_FLOATS * sizeof(float) : (size_t)-1; }
size_t afr_sheet_bwd_lds_bytes(const SheetDims& d) { return d.L <= LMAX ? (size_t)BWD_FLOATS * sizeof(float) : (size_t)-1; }

// the opt-in to > 64 KiB of dynamic LDS is a property of the (device, kernel) pair: set once, not per launch
template <typename K>
static hipError_t lds_optin_once(K kernel, size_t lds, bool (&done)[16]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16 || !done[dev]) {
        if ((e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        if (dev >= 0 && dev < 16) done[dev] = true;
    }
    return hipSuccess;
}

hipError_t afr_launch_sheet_fwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr, const int64_t* x,
                                int ldx, int B, void* z, float ln_eps, uint32_t* err_flag, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t lds = afr_sheet_fwd_lds_bytes(d);
    dim3 g(afr_sheet_blocks(B)), blk(NT);
    hipError_t e;
    static bool opt16[16], opt32[16];
    if (act_dtype == AFR_BF16) {
        if ((e = lds_optin_once(sheet_fwd_kernel<bf16_t>, lds, opt16)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_fwd_kernel<bf16_t>, g, blk, lds, s, d, P, dr, x, ldx, B, (bf16_t*)z, ln_eps, err_flag);
    } else {
        if ((e = lds_optin_once(sheet_fwd_kernel<float>, lds, opt32)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_fwd_kernel<float>, g, blk, lds, s, d, P, dr, x, ldx, B, (float*)z, ln_eps, err_flag);
    }
    return hipGetLastError();
}

hipError_t afr_launch_sheet_bwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr, const int64_t* x,
                                int ldx, int B, const void* dz, float ln_eps, float* slabs, const SheetSlabOff& so, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t lds = afr_sheet_bwd_lds_bytes(d);
    const int nb = afr_sheet_blocks(B);
    if (so.total & 3) return hipErrorInvalidValue;             // the kernel zeroes its slab 16 bytes at a time
    hipError_t e;
    static bool opt16[16], opt32[16];
    dim3 g(nb), blk(NT);
    if (act_dtype == AFR_BF16) {
        if ((e = lds_optin_once(sheet_bwd_kernel<bf16_t>, lds, opt16)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_bwd_kernel<bf16_t>, g, blk, lds, s, d, P, dr, x, ldx, B, (const bf16_t*)dz, ln_eps, slabs, so);
    } else {
        if ((e = lds_optin_once(sheet_bwd_kernel<float>, lds, opt32)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_bwd_kernel<float>, g, blk, lds, s, d, P, dr, x, ldx, B, (const float*)dz, ln_eps, slabs, so);
    }
    return hipGetLastError();
}
