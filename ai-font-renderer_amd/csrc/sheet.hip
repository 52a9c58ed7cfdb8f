// sheet.hip -- the sheet model's per-string front end, fused into one kernel per direction.
//
// Forward (reference model.py:167-193): embedding gather -> dropout -> + learned positions -> packed in-proj ->
// 4-head 100x100 softmax attention (dropout on the probabilities) -> out-proj -> residual + LayerNorm ->
// fc1 + ReLU + dropout -> flattened (zero-padded) feature row z[b][max_length*64], the A operand of fc_output.
// Backward (model.py:309) recomputes that chain in LDS from the same counter-hash dropout stream instead of
// saving activations (a sample's whole state is <100 KB of LDS), then walks it in reverse.
//
// One 1024-thread workgroup owns one string at a time and loops over strings (persistent grid, <= 256 blocks):
// weights are loaded into LDS once.  The small matmuls (in-proj, out-proj, fc1 and their data/weight gradients) run on
// the matrix cores straight from LDS (exact-f32 v_mfma_f32_16x16x4_f32); the 100x100 attention stays on the VALU.  Every reduction has a single owner thread and a fixed order -- no atomics:
//   * attention backward is split by ROW (softmax statistics, delta, dq) and then by COLUMN (dk, dv), each
//     recomputing the scores it needs, so dv/dk need no scatter;
//   * the 10 small parameter gradients accumulate across the block's strings in registers (dEmb in LDS, row v
//     owned by thread slot v%32) and leave as one partial slab per block, summed in block order by reduce_slabs.
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
__device__ __forceinline__ void ph_attention(float* o, const float* qkv, const SheetDrop& dr, int b, int L, int tid,
                                             float* smax_out = nullptr, float* sinv_out = nullptr) {
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
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
        for (int j = part; j < L; j += 2) {
            ld8(qkv + j * SQ + E + h * D, kv);
            const float p = __expf(dot8(q, kv) - mx);
            sum += p;
            const float pm = p * attn_mask(dr, b, h, i, j, L);
            ld8(qkv + j * SQ + 2 * E + h * D, kv);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(pm, kv[d], acc[d]);
        }
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
__device__ __forceinline__ void ph_layernorm(float* xh, float* n, float* rstd, const Wts& w, int L, float eps, int tid) {
    if (tid < L) {
        float* row = xh + tid * SE;
        float mu = 0.f;
#pragma unroll 8
        for (int c = 0; c < E; ++c) mu += row[c];
        mu *= (1.f / E);
        float var = 0.f;
#pragma unroll 8
        for (int c = 0; c < E; ++c) { const float dlt = row[c] - mu; var = fmaf(dlt, dlt, var); }
        var *= (1.f / E);
        const float rs = 1.f / sqrtf(var + eps);
        rstd[tid] = rs;
#pragma unroll 8
        for (int c = 0; c < E; ++c) {
            const float xv = (row[c] - mu) * rs;
            row[c] = xv;
            n[tid * SE + c] = fmaf(xv, w.lg[c], w.lb[c]);
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
    const int tid = threadIdx.x, L = dm.L;
    const Wts w = carve_weights(sm);
    float* e = sm + W_FLOATS;
    float* qkv = e + al4(L * SE);
    float* o = qkv + L * SQ;
    float* r = o + al4(L * SE);
    float* n = r + al4(L * SE);
    float* rstd = n + al4(L * SE);
    float* smax = rstd + al4(L);       // [4L]
    float* sinv = smax + H * L;        // [4L]
    int* tok = reinterpret_cast<int*>(sinv + H * L);
    load_weights(w, P, tid);
    const size_t Kz = (size_t)dm.Lmax * F;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        ph_tokens(tok, x, ldx, b, L, dm.vocab, err, tid);
        __syncthreads();
        ph_embed(e, tok, P, dr, b, L, tid);
        __syncthreads();
        ph_inproj(qkv, e, w, L, tid);
        __syncthreads();
        ph_attention(o, qkv, dr, b, L, tid, dr.save ? smax : nullptr, sinv);
        __syncthreads();
        if (dr.save) {                                              // keep o and the softmax statistics for backward
            float* sv = dr.save + (size_t)b * L * 40;
            for (int i = tid; i < L * E; i += NT) sv[i] = o[(i >> 5) * SE + (i & 31)];
            for (int i = tid; i < H * L; i += NT) { sv[L * E + i] = smax[i]; sv[L * E + H * L + i] = sinv[i]; }
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

template <typename T>
__global__ __launch_bounds__(NT) void sheet_bwd_kernel(SheetDims dm, SheetParams P, SheetDrop dr, const int64_t* __restrict__ x,
                                                        int ldx, int B, const T* __restrict__ dz, float eps,
                                                        float* __restrict__ slabs, SheetSlabOff so) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, L = dm.L;
    const Wts w = carve_weights(sm);
    float* e = sm + W_FLOATS;          // [L][33]   whole sample
    float* big = e + al4(L * SE);      // [L][100]  qkv  | later n [L][33] + df [L][64] | later qkv again -> dk,dv in place
    float* o = big + L * SQ;           // [L][33]   o    | later dq
    float* xh = o + al4(L * SE);       // [L][36]   r -> xhat (stride 33) | later dO, grad wrt attention output (stride 36)
    float* dn = xh + L * SD;           // [L][33]   dn -> dr -> de
    float* rstd = dn + al4(L * SE);    // [L]
    float* smax = rstd + al4(L);       // [4L]
    float* sinv = smax + H * L;        // [4L]
    float* sdel = sinv + H * L;        // [4L]
    float* demb = sdel + H * L;        // [vocab][32]
    int* tok = reinterpret_cast<int*>(demb + dm.vocab * E);
    float* nbuf = big;                 // n  [L][33]
    float* df = big + L * SE;          // df [L][64]
    load_weights(w, P, tid);
    for (int i = tid; i < dm.vocab * E; i += NT) demb[i] = 0.f;

    const int c32 = tid & 31, g8 = tid >> 5;          // g8: column group 0..NG-1
    constexpr int KPOS = (120 + NG - 1) / NG;
    float aPos[KPOS];
    // weight-gradient tiles (16x16, MFMA accumulators, persistent across strings): 8 of dW1 [64x32], 4 of dWo [32x32],
    // 12 of dWin [96x32].  Wave w owns tile w (accA); waves 0..7 also own dWin tile w+4 (accB).
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    const int wave = tid >> 6, lane = tid & 63;
    float a_b1 = 0.f, a_bo = 0.f, a_g = 0.f, a_b = 0.f, a_bin = 0.f;
#pragma unroll
    for (int k = 0; k < KPOS; ++k) aPos[k] = 0.f;
    const float scale = 0.35355339059327373f;
    const size_t Kz = (size_t)dm.Lmax * F;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- recompute the forward chain
        ph_tokens(tok, x, ldx, b, L, dm.vocab, nullptr, tid);
        __syncthreads();
        ph_embed(e, tok, P, dr, b, L, tid);
        __syncthreads();
        const bool saved = dr.save != nullptr;
        if (saved) {                                               // o and the softmax statistics come from the forward
            const float* sv = dr.save + (size_t)b * L * 40;
            for (int i = tid; i < L * E; i += NT) o[(i >> 5) * SE + (i & 31)] = sv[i];
            for (int i = tid; i < H * L; i += NT) { smax[i] = sv[L * E + i]; sinv[i] = sv[L * E + H * L + i]; }
        } else {
            ph_inproj(big, e, w, L, tid);
            __syncthreads();
            ph_attention(o, big, dr, b, L, tid);
        }
        __syncthreads();
        ph_outproj_res(xh, e, o, w, L, tid);
        __syncthreads();
        ph_layernorm(xh, nbuf, rstd, w, L, eps, tid);        // qkv dead: n overwrites the front of `big`
        __syncthreads();
        // ---- df = dz * dropout-mask * [pre>0]
        const T* dzr = dz + (size_t)b * Kz;
        for (int i = tid; i < L * F; i += NT) df[i] = (float)dzr[i] * fc_mask(dr, b, L, i);
        __syncthreads();
        lds_mma(nbuf, SE, 1, w.W1, SE, 1, L, F, E, tid, [&](int m, int j, float v) { if (!(v + w.b1[j] > 0.f)) df[m * F + j] = 0.f; });
        __syncthreads();
        // ---- dW1 += df^T n (persistent MFMA tiles) ; db1 += sum df ; dn = df . W1
        if (wave < 8) lds_mma_tile(accA, df, 1, F, nbuf, 1, SE, 16 * (wave >> 1), 16 * (wave & 1), L, lane);
        if (tid < F) { float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
            for (int l = 0; l < L; ++l) a += df[l * F + tid]; a_b1 += a; }
        lds_mma(df, F, 1, w.W1, 1, SE, L, E, F, tid, [&](int m, int c, float v) { dn[m * SE + c] = v; });
        __syncthreads();
        // ---- LayerNorm backward: dgamma, dbeta (column owners), then dr in place (row owners)
        if (tid < E) { float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
            for (int l = 0; l < L; ++l) a = fmaf(dn[l * SE + tid], xh[l * SE + tid], a); a_g += a; }
        else if (tid < 2 * E) { const int c = tid - E; float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
            for (int l = 0; l < L; ++l) a += dn[l * SE + c]; a_b += a; }
        __syncthreads();
        if (tid < L) {
            float* row = dn + tid * SE;
            const float* xr = xh + tid * SE;
            float m1 = 0.f, m2 = 0.f;
#pragma unroll 8
            for (int c = 0; c < E; ++c) { const float gv = row[c] * w.lg[c]; m1 += gv; m2 = fmaf(gv, xr[c], m2); }
            m1 *= (1.f / E); m2 *= (1.f / E);
            const float rs = rstd[tid];
#pragma unroll 8
            for (int c = 0; c < E; ++c) row[c] = (row[c] * w.lg[c] - m1 - xr[c] * m2) * rs;
        }
        __syncthreads();
        // ---- out-proj backward: dWo += dr^T o ; dbo += sum dr ; dO = dr . Wo  (written over xhat, which is dead)
        if (wave >= 8 && wave < 12) lds_mma_tile(accA, dn, 1, SE, o, 1, SE, 16 * ((wave - 8) >> 1), 16 * (wave & 1), L, lane);
        if (tid < E) { float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
            for (int l = 0; l < L; ++l) a += dn[l * SE + tid]; a_bo += a; }
        lds_mma(dn, SE, 1, w.Wo, 1, SE, L, E, E, tid, [&](int m, int c, float v) { xh[m * SD + c] = v; });
        __syncthreads();                                     // n, df dead; o dead after the dWo loop above
        ph_inproj(big, e, w, L, tid);                        // recompute qkv
        __syncthreads();
        // ---- attention backward, by ROW: softmax stats, delta = sum_j dA.A, dq      (dq -> `o` buffer)
        //      two adjacent lanes per (head, query row): keys split even/odd, combined with xor-1 shuffles
        for (int rr = tid; rr < 2 * H * L; rr += NT) {
            const int r = rr >> 1, part = rr & 1;
            const int h = r / L, i = r - h * L;
            float q[D], dO[D], kk[D], vv[D];
            ld8(big + i * SQ + h * D, q);
            ld8(xh + i * SD + h * D, dO);
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] *= scale;
            float mx;
            if (saved) {
                mx = smax[r];
            } else {
                mx = -INFINITY;
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
                for (int j = part; j < L; j += 2) {
                    ld8(big + j * SQ + E + h * D, kk);
                    mx = fmaxf(mx, dot8(q, kk));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            }
            // one pass for everything that depends on the softmax row: with p~ = exp(s - max) (unnormalised),
            //   sum = S p~ ,  num = S p~ dA ,  T1 = S p~ dA k_j ,  T2 = S p~ k_j     (dA = (dO.v_j) * dropout mask)
            // then  delta = num/sum  and  dq = scale * (T1 - delta * T2) / sum  ==  scale * S_j A_ij (dA_ij - delta) k_j
            float sum = 0.f, num = 0.f, t1[D], t2[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { t1[d] = 0.f; t2[d] = 0.f; }
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
            for (int j = part; j < L; j += 2) {
                ld8(big + j * SQ + E + h * D, kk);
                ld8(big + j * SQ + 2 * E + h * D, vv);
                const float p = __expf(dot8(q, kk) - mx);
                const float pa = p * (dot8(dO, vv) * attn_mask(dr, b, h, i, j, L));
                sum += p;
                num += pa;
#pragma unroll
                for (int d = 0; d < D; ++d) { t1[d] = fmaf(pa, kk[d], t1[d]); t2[d] = fmaf(p, kk[d], t2[d]); }
            }
            sum += __shfl_xor(sum, 1, 64);
            num += __shfl_xor(num, 1, 64);
            const float inv = 1.f / sum, delta = num * inv;
            if (part == 0) { smax[r] = mx; sinv[r] = inv; sdel[r] = delta; }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float a1 = t1[d] + __shfl_xor(t1[d], 1, 64);
                const float a2 = t2[d] + __shfl_xor(t2[d], 1, 64);
                if (part == 0) o[i * SE + h * D + d] = (a1 - delta * a2) * inv * scale;
            }
        }
        __syncthreads();
        // ---- attention backward, by COLUMN: dk_j, dv_j; two lanes per (head, key), queries split even/odd.  The pair
        //      reads k_j, v_j into registers first and only then (after the shuffles) lane 0 overwrites them in place.
        for (int rr = tid; rr < 2 * H * L; rr += NT) {
            const int r = rr >> 1, part = rr & 1;
            const int h = r / L, j = r - h * L;
            float kk[D], vv[D], dk[D], dv[D], qs[D], dO[D];
            ld8(big + j * SQ + E + h * D, kk);
            ld8(big + j * SQ + 2 * E + h * D, vv);
#pragma unroll
            for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma clang loop unroll_count(2) vectorize(disable) interleave(disable)
            for (int i = part; i < L; i += 2) {
                ld8(big + i * SQ + h * D, qs);
                ld8(xh + i * SD + h * D, dO);
#pragma unroll
                for (int d = 0; d < D; ++d) qs[d] *= scale;
                const float p = __expf(dot8(qs, kk) - smax[h * L + i]) * sinv[h * L + i];
                const float m = attn_mask(dr, b, h, i, j, L);
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
        // ---- in-proj backward: dWin += dqkv^T e ; dbin += sum dqkv ; de = dr + dqkv . Win   (de in place over dr)
        //      dqkv lives in two places: dq in `o` (columns 0..31), dk|dv in `big` (columns 32..95)
        if (wave >= 12) {                                   // dWin tiles 0..3: rows j = 0..31 come from dq
            const int id = wave - 12;
            lds_mma_tile(accA, o, 1, SE, e, 1, SE, 16 * (id >> 1), 16 * (id & 1), L, lane);
        } else if (wave < 8) {                              // dWin tiles 4..11: rows j = 32..95 come from dk|dv
            const int id = wave + 4;
            lds_mma_tile(accB, big, 1, SQ, e, 1, SE, 16 * (id >> 1), 16 * (id & 1), L, lane);
        }
        if (tid < QKV) { float a = 0.f;
_Pragma("clang loop unroll_count(4) vectorize(disable) interleave(disable)")
            for (int l = 0; l < L; ++l) a += dqkv_at(o, big, l, tid); a_bin += a; }
        lds_mma(o, SE, 1, w.Win, 1, SE, L, E, E, tid, [&](int m, int c, float v) { dn[m * SE + c] += v; });
        lds_mma(big + E, SQ, 1, w.Win + E * SE, 1, SE, L, E, 2 * E, tid, [&](int m, int c, float v) { dn[m * SE + c] += v; });
        __syncthreads();
        // ---- dP += de ; dEmb[tok] += de * embed-dropout-mask       (row v of dEmb owned by thread slot v%NG)
#pragma unroll
        for (int k = 0; k < KPOS; ++k) { const int l = g8 + NG * k; if (l < L) aPos[k] += dn[l * SE + c32]; }
#pragma clang loop unroll_count(4) vectorize(disable) interleave(disable)
        for (int l = 0; l < L; ++l) {
            const int v = tok[l];
            if ((v & (NG - 1)) == g8) {
                float m = 1.f;
                if (dr.training) m = afr_keep((uint64_t)b * L * E + l * E + c32, dr.key_e, dr.thr_e) ? dr.sc_e : 0.f;
                demb[v * E + c32] += dn[l * SE + c32] * m;
            }
        }
        __syncthreads();
    }
    // ---- one partial slab per block, laid out like the flat parameter buffer (pads were zeroed by the host memset)
    float* S = slabs + (size_t)blockIdx.x * so.total;
#pragma unroll
    for (int k = 0; k < KPOS; ++k) { const int l = g8 + NG * k; if (l < L) S[so.pos + l * E + c32] = aPos[k]; }
    for (int i = tid; i < dm.vocab * E; i += NT) S[so.emb + i] = demb[i];
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
size_t afr_sheet_fwd_lds_bytes(const SheetDims& d) {
    return (size_t)(W_FLOATS + 4 * al4h(d.L * SE) + d.L * SQ + 2 * al4h(d.L) + 2 * H * d.L) * sizeof(float);
}
size_t afr_sheet_bwd_lds_bytes(const SheetDims& d) {
    return (size_t)(W_FLOATS + 3 * al4h(d.L * SE) + d.L * SD + d.L * SQ + al4h(d.L) + 3 * H * d.L + d.vocab * E + d.L) * sizeof(float);
}

hipError_t afr_launch_sheet_fwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr, const int64_t* x,
                                int ldx, int B, void* z, float ln_eps, uint32_t* err_flag, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t lds = afr_sheet_fwd_lds_bytes(d);
    dim3 g(afr_sheet_blocks(B)), blk(NT);
    hipError_t e;
    if (act_dtype == AFR_BF16) {
        if ((e = hipFuncSetAttribute((const void*)sheet_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_fwd_kernel<bf16_t>, g, blk, lds, s, d, P, dr, x, ldx, B, (bf16_t*)z, ln_eps, err_flag);
    } else {
        if ((e = hipFuncSetAttribute((const void*)sheet_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_fwd_kernel<float>, g, blk, lds, s, d, P, dr, x, ldx, B, (float*)z, ln_eps, err_flag);
    }
    return hipGetLastError();
}

hipError_t afr_launch_sheet_bwd(int act_dtype, const SheetDims& d, const SheetParams& P, const SheetDrop& dr, const int64_t* x,
                                int ldx, int B, const void* dz, float ln_eps, float* slabs, const SheetSlabOff& so, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const size_t lds = afr_sheet_bwd_lds_bytes(d);
    const int nb = afr_sheet_blocks(B);
    hipError_t e = hipMemsetAsync(slabs, 0, (size_t)nb * so.total * sizeof(float), s);
    if (e != hipSuccess) return e;
    dim3 g(nb), blk(NT);
    if (act_dtype == AFR_BF16) {
        if ((e = hipFuncSetAttribute((const void*)sheet_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_bwd_kernel<bf16_t>, g, blk, lds, s, d, P, dr, x, ldx, B, (const bf16_t*)dz, ln_eps, slabs, so);
    } else {
        if ((e = hipFuncSetAttribute((const void*)sheet_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(sheet_bwd_kernel<float>, g, blk, lds, s, d, P, dr, x, ldx, B, (const float*)dz, ln_eps, slabs, so);
    }
    return hipGetLastError();
}
