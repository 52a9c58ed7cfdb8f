// afr_api.hip -- the C ABI of libafr.so (include/afr.h): plan construction, workspace carve-up and the
// launch sequences of forward / loss / backward / AdamW for both model families.
#include "afr_common.h"
#include "../../include/afr.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

// --------------------------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(AFR_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every entry point that enqueues work makes the device that owns the caller's buffers current for the duration of the
// call (and restores the previous one): kernels are launched on the caller's stream, which belongs to that device, so a
// process whose current device is another GPU (LOCAL_RANK != 0 without a set_device) still launches where the data lives.
static int device_of(const void* ptr) {
    hipPointerAttribute_t a;
    if (!ptr || hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return a.device;
}
struct DevGuard {
    int prev = -1; bool switched = false;
    explicit DevGuard(int dev) {
        if (dev < 0 || hipGetDevice(&prev) != hipSuccess || prev == dev) return;
        switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" int afr_version(void) { return AFR_VERSION; }
extern "C" const char* afr_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ plan
struct Tensor {
    std::string name;
    int64_t off = 0, numel = 0;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
};

struct ProfRec { int tag; hipEvent_t a, b; double flops, bytes; };

struct afr_plan {
    afr_config cfg;
    std::vector<Tensor> params;
    int64_t total = 0;
    int act_bytes = 4;
    // bound buffers
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
    char* ws = nullptr;
    size_t ws_bytes = 0, ws_need = 0;
    int device = -1;               // the GPU that owns the bound buffers (afr_bind)
    // products collected for ONE grouped launch (a layer's weight gradient + input gradient): see flush_gemms
    bool defer = false;
    int pend_tile256 = 0;
    std::vector<GemmParams> pend;
    std::vector<std::string> pend_tag;
    double pend_flops = 0.0, pend_bytes = 0.0;
    // workspace offsets (bytes)
    size_t o_shadow = 0, o_err = 0, o_loss = 0, o_u = 0, o_z = 0, o_dz = 0, o_slab_e = 0, o_save = 0;
    std::vector<size_t> o_act;     // glyph: activations h0..h_nh
    size_t o_d[2] = {0, 0};        // glyph: ping-pong d buffers
    size_t o_table = 0;            // glyph: [Emb; Font] . W1^T, the first Linear folded through the tables
    size_t o_dw1 = 0;              // glyph: compact dW1 [N1][E] extracted from the widened weight-gradient slabs
    int k0 = 0;                    // glyph: columns of h0' = [h0 | one-hot] when the first layer is folded, else 0
    // glyph, one small hidden layer (BASELINE C1 / C2): the whole step as ONE fused kernel + the grouped reduce (glyph_fused.hip)
    bool fused1 = false;
    bool wT_valid = false;         // bf16: the transposed operand copies W1T / W2T match the current parameters
    size_t o_w1t = 0, o_w2t = 0, o_slab1 = 0;
    bool l1f = false; size_t o_l1f = 0;   // glyph, bf16: folded first layer's backward in one kernel (slabs [blocks][dW1|db1|dTab])
    // glyph, bf16 training steps: the first layer's output as a COMBINATION table (elementwise.hip glyph_combo_kernel): h1 / h0
    // are not materialised per glyph; the products that consume them gather table rows through cidx while staging
    bool combo_ok = false, combo_on = false; size_t o_h1c = 0, o_h0c = 0, o_cidx = 0; int h1c_ld = 0;
    // glyph, bf16 training steps: hidden activations written by a GEMM epilogue also leave their ReLU mask as bits
    // (o_mbits[i] for the output of layer i, 0 = none); the next layer's input-gradient product reads those instead of the activation
    std::vector<size_t> o_mbits; bool mbits_on = false;
    size_t o_fix = 0, o_fixcnt = 0; bool have_fix = false;   // in-launch split-K: slice parking area + per-tile arrival counters
    // bf16 glyph nets: TWO weight shadows.  A fused optimizer step writes every tensor's new bf16 copy into the one that is
    // not being read (a layer's weight-gradient workgroups update the weights while the same launch's input-gradient
    // workgroups still read them), and the roles swap when the step is complete.
    size_t o_shadow2 = 0; int shadow_cur = 0;
    // hyper-parameters of the optimizer step in progress (afr_train_step): set while backward runs, so that weight-gradient
    // products with a cooperative split-K tail apply AdamW themselves; adam_done lists the tensors they have updated
    bool step_on = false; float st_decay = 1.f, st_b1 = 0.f, st_b2 = 0.f, st_eps = 0.f, st_step = 0.f, st_rsqrt_bc2 = 1.f;
    std::vector<int64_t> adam_done;
    // pixel-token transformer (AFR_KIND_PIXEL): per-block parameter offsets and the forward's workspace
    struct PixBlock { int64_t ln1g, ln1b, win, bin, wo, bo, ln2g, ln2b, w1, b1, w2, b2; };
    std::vector<PixBlock> pix;
    int64_t px_pos = 0, px_emb = 0, px_font = -1, px_lnfg = 0, px_lnfb = 0, px_wout = 0, px_bout = 0;
    struct PixSave { size_t hin, h1, n1, q, o, n2, kv, f, ln1p, ln2p, fbits; };      // per block: what its backward needs + its LayerNorm partial slabs
    std::vector<PixSave> pxs;
    size_t o_ctx = 0, o_a = 0, o_hf = 0, o_dh = 0, o_dht = 0, o_df = 0, o_dn = 0, o_dq = 0, o_dkvp = 0, o_dkv = 0, o_dkvt = 0, o_dctxt = 0,
           o_dctx = 0, o_headp = 0;
    // glyph layer table
    struct Layer { int N, K; int64_t w_off, b_off; int sk = 1; size_t o_slab_w = 0, o_slab_b = 0;
                   size_t o_cnt = 0; int n_cnt = 0; unsigned coop_epoch = 0; };   // cooperative split-K: per-tile arrival counters, launches so far
    std::vector<Layer> layers;
    std::vector<Layer> pxl;          // the 5 Linears of every block as weight-gradient descriptors: q, kv, out-proj, fc1, fc2
    int64_t emb_off = 0, font_off = 0;
    // sheet offsets
    int64_t s_pos = 0, s_emb = 0, s_win = 0, s_bin = 0, s_wo = 0, s_bo = 0, s_g = 0, s_b = 0, s_w1 = 0, s_b1 = 0, s_wout = 0, s_bout = 0;
    // state left by the last forward
    const int64_t* last_x = nullptr;
    const int64_t* last_font = nullptr;
    int last_B = 0, last_L = 0, last_ldx = 0, last_training = 0;
    uint64_t last_step = 0;
    bool have_du = false;
    int next_stage = 0;
    // profiling
    int prof_mode = 0;      // 0 off, 1 every launch, 2 only prof_only, 3 every 4th launch of prof_only
    unsigned prof_seen = 0; // launches of prof_only met in mode 3
    std::string prof_only_sym;   // mode 2: the kernel symbol to keep timing
    double prof_overhead_ms = -1.0;   // event-bracket overhead, measured on first use (prof_calibrate)
    std::vector<ProfRec> prof;
    std::vector<std::string> prof_tags;
    std::vector<hipEvent_t> ev_pool;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void add_param(afr_plan* p, const char* name, std::initializer_list<int64_t> shape) {
    Tensor t;
    t.name = name;
    t.ndim = (int)shape.size();
    int64_t n = 1;
    int i = 0;
    for (int64_t s : shape) { t.shape[i++] = s; n *= s; }
    t.numel = n;
    t.off = p->total;
    p->total += (n + 63) / 64 * 64;
    p->params.push_back(t);
}
static int64_t off_of(const afr_plan* p, const char* name) {
    for (const Tensor& t : p->params)
        if (t.name == name) return t.off;
    return -1;
}

static_assert(AFR_RT_MAXSEG >= 2 * AFR_MAX_HIDDEN + 2 * AFR_L1F_MAX_SPLIT + 4, "a backward pass of the deepest glyph net must fit one grouped reduce");
#ifndef AFR_SPLITK_TARGET
#define AFR_SPLITK_TARGET 512
#endif
static int choose_splitk(int M, int N, int K) {
    // a small weight gradient reduced over very many rows (the pixel transformer's: 131072 token rows into 2048 x 512): as many
    // K-slices as make the 256x256 tiles fill the chip exactly once -- the launcher then takes the 256x256 body for it
    // (gemm.hip bf16_use_body256; measured 323 -> 292 us and 287 -> 246 us against the best split of the 256x128 ring)
    const int t256 = ((M + 255) / 256) * ((N + 255) / 256);
    if (K >= 32768 && t256 >= 8 && t256 <= 64 && 256 % t256 == 0 && K / (256 / t256) >= 1024) return 256 / t256;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int s = (AFR_SPLITK_TARGET + tiles - 1) / tiles;
    const int maxs = K / 256 > 0 ? K / 256 : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

extern "C" int afr_plan_create(const afr_config* c, afr_plan** out) {
    if (!c || !out) return fail(AFR_EINVAL, "null argument");
    if (c->dtype != AFR_F32 && c->dtype != AFR_BF16) return fail(AFR_EINVAL, "dtype must be AFR_F32 or AFR_BF16");
    if (c->max_batch <= 0) return fail(AFR_EINVAL, "max_batch must be positive");
    if (c->vocab <= 0 || c->embed_dim <= 0 || c->out_h <= 0 || c->out_w <= 0) return fail(AFR_EINVAL, "bad shape");
    const int Pix = c->out_h * c->out_w;
    if (Pix % 8) return fail(AFR_EUNSUPPORTED, "out_h*out_w must be a multiple of 8 (got %d)", Pix);
    afr_plan* p = new afr_plan();
    p->cfg = *c;
    p->act_bytes = c->dtype == AFR_BF16 ? 2 : 4;
    const int E = c->embed_dim;
    const size_t B = (size_t)c->max_batch;
    const size_t ab = (size_t)p->act_bytes;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };

    if (c->kind == AFR_KIND_SHEET) {
        if (E != 32 || c->heads != 4 || c->fc_dim != 64)
            { delete p; return fail(AFR_EUNSUPPORTED, "sheet front end is built for embed_dim 32, 4 heads, fc_dim 64 (model.py:79,81,148)"); }
        if (c->max_length <= 0 || c->max_length > 120) { delete p; return fail(AFR_EUNSUPPORTED, "max_length must be in 1..120 (one string's state must fit 160 KiB of LDS)"); }
        const int L = c->max_length, F = c->fc_dim;
        add_param(p, "positional_encoding", {L, E});
        add_param(p, "embedding.weight", {c->vocab, E});
        add_param(p, "attention.in_proj_weight", {3 * E, E});
        add_param(p, "attention.in_proj_bias", {3 * E});
        add_param(p, "attention.out_proj.weight", {E, E});
        add_param(p, "attention.out_proj.bias", {E});
        add_param(p, "layer_norm.weight", {E});
        add_param(p, "layer_norm.bias", {E});
        add_param(p, "fc1.weight", {F, E});
        add_param(p, "fc1.bias", {F});
        add_param(p, "fc_output.weight", {Pix, (int64_t)L * F});
        add_param(p, "fc_output.bias", {Pix});
        p->s_pos = off_of(p, "positional_encoding"); p->s_emb = off_of(p, "embedding.weight");
        p->s_win = off_of(p, "attention.in_proj_weight"); p->s_bin = off_of(p, "attention.in_proj_bias");
        p->s_wo = off_of(p, "attention.out_proj.weight"); p->s_bo = off_of(p, "attention.out_proj.bias");
        p->s_g = off_of(p, "layer_norm.weight"); p->s_b = off_of(p, "layer_norm.bias");
        p->s_w1 = off_of(p, "fc1.weight"); p->s_b1 = off_of(p, "fc1.bias");
        p->s_wout = off_of(p, "fc_output.weight"); p->s_bout = off_of(p, "fc_output.bias");
        const size_t Kz = (size_t)L * F;
        if (c->dtype == AFR_BF16) p->o_shadow = carve((size_t)p->total * 2);
        p->o_err = carve(256);
        p->o_loss = carve(((size_t)((B + 127) / 128) * ((Pix + 127) / 128) + 1040) * sizeof(float));
        p->o_z = carve(B * Kz * ab);
        p->o_u = carve(B * Pix * ab);
        p->o_dz = carve(B * Kz * ab);
        afr_plan::Layer ly; ly.N = Pix; ly.K = (int)Kz; ly.w_off = p->s_wout; ly.b_off = p->s_bout;
        ly.sk = choose_splitk(Pix, (int)Kz, (int)B);
        if (ly.sk > 1) { ly.o_slab_w = carve((size_t)ly.sk * Pix * Kz * sizeof(float)); ly.o_slab_b = carve((size_t)ly.sk * Pix * sizeof(float)); }
        p->layers.push_back(ly);
        p->o_slab_e = carve((size_t)afr_sheet_blocks((int)B) * (size_t)p->s_wout * sizeof(float));
        p->o_save = carve(B * (size_t)L * AFR_SHEET_SAVE_PER_POS * sizeof(float));
        if (c->dtype == AFR_BF16 && (c->reserved & 8)) {   // opt-in: fc_output's input-gradient product as in-launch split-K
            int hd, sk;
            if (afr_gemm_fix_plan((int)B, Pix, (int)Kz, &hd, &sk) || afr_gemm_fix_plan((int)B, (int)Kz, Pix, &hd, &sk)) {
                p->o_fix = carve(AFR_FIX_WS_BYTES); p->o_fixcnt = carve(AFR_FIX_MAX_SLICES * sizeof(unsigned)); p->have_fix = true;
            }
        }
    } else if (c->kind == AFR_KIND_GLYPH) {
        if (c->n_hidden < 0 || c->n_hidden > AFR_MAX_HIDDEN) { delete p; return fail(AFR_EINVAL, "n_hidden out of range"); }
        if (E % 8) { delete p; return fail(AFR_EUNSUPPORTED, "embed_dim must be a multiple of 8"); }
        for (int i = 0; i < c->n_hidden; ++i)
            if (c->hidden[i] <= 0 || c->hidden[i] % 8) { delete p; return fail(AFR_EUNSUPPORTED, "hidden widths must be positive multiples of 8"); }
        add_param(p, "embedding.weight", {c->vocab, E});
        if (c->n_fonts > 0) add_param(p, "font_embedding.weight", {c->n_fonts, E});
        int k = E;
        char nm[64];
        for (int i = 0; i < c->n_hidden; ++i) {
            snprintf(nm, sizeof nm, "fc%d.weight", i + 1);
            add_param(p, nm, {c->hidden[i], k});
            afr_plan::Layer ly; ly.N = c->hidden[i]; ly.K = k; ly.w_off = p->params.back().off;
            snprintf(nm, sizeof nm, "fc%d.bias", i + 1);
            add_param(p, nm, {c->hidden[i]});
            ly.b_off = p->params.back().off;
            p->layers.push_back(ly);
            k = c->hidden[i];
        }
        add_param(p, "fc_output.weight", {Pix, k});
        afr_plan::Layer ly; ly.N = Pix; ly.K = k; ly.w_off = p->params.back().off;
        add_param(p, "fc_output.bias", {Pix});
        ly.b_off = p->params.back().off;
        p->layers.push_back(ly);
        p->emb_off = off_of(p, "embedding.weight");
        p->font_off = c->n_fonts > 0 ? off_of(p, "font_embedding.weight") : -1;
        if (c->dtype == AFR_BF16) { p->o_shadow = carve((size_t)p->total * 2); p->o_shadow2 = carve((size_t)p->total * 2); }
        p->o_err = carve(256);
        p->o_loss = carve(((size_t)((B + 127) / 128) * ((Pix + 127) / 128) + 1040) * sizeof(float));
        size_t maxw = (size_t)E, maxn = 0;
        p->k0 = c->n_hidden > 0 ? afr_glyph_k0(E, c->vocab, c->n_fonts) : 0;
        if (p->k0 > 512 || E > 128) p->k0 = 0;     // beyond what the folded-layer kernels stage per block: plain embedding + GEMM path
        p->o_act.push_back(carve(B * (size_t)(p->k0 ? p->k0 : E) * ab));
        for (int i = 0; i < c->n_hidden; ++i) {
            p->o_act.push_back(carve(B * (size_t)c->hidden[i] * ab));
            if ((size_t)c->hidden[i] > maxw) maxw = (size_t)c->hidden[i];
        }
        p->o_mbits.assign(c->n_hidden + 1, 0);
        if (c->dtype == AFR_BF16 && !(c->reserved & 128))
            for (int i = 1; i < c->n_hidden; ++i)          // layer 0's output comes from the table / gather kernels, not a GEMM
                if (c->hidden[i] % 8 == 0) p->o_mbits[i] = carve(B * (size_t)(c->hidden[i] / 8));
        p->o_u = carve(B * Pix * ab);
        p->o_d[0] = carve(B * maxw * ab);
        p->o_d[1] = carve(B * maxw * ab);
        bool first = true;
        for (auto& l : p->layers) {
            // the folded first layer's weight-gradient GEMM is K0 wide and always lands in slabs (even a single one)
            const int kw = (first && p->k0) ? p->k0 : l.K;
            l.sk = choose_splitk(l.N, kw, (int)B);
            // (slab space in whole 256x256 tiles: the cooperative split-K parks register images of full tiles there)
            const size_t wt = (size_t)((l.N + 255) / 256) * ((kw + 255) / 256);
            if (l.sk > 1 || (first && p->k0)) { l.o_slab_w = carve((size_t)l.sk * wt * 65536 * sizeof(float)); l.o_slab_b = carve((size_t)l.sk * l.N * sizeof(float)); }
            l.n_cnt = (int)wt; l.o_cnt = carve(wt * sizeof(unsigned));
            if ((size_t)l.N > maxn) maxn = (size_t)l.N;
            first = false;
        }
        size_t eb = (size_t)afr_embed_bwd_blocks((int)B);
        if (p->k0) {
            const size_t lb = (size_t)afr_glyph_l1_bwd_blocks(c->hidden[0]);
            if (lb > eb) eb = lb;
            p->o_table = carve((size_t)(c->vocab + c->n_fonts) * c->hidden[0] * sizeof(float));
            p->o_dw1 = carve((size_t)c->hidden[0] * E * sizeof(float));
            if (afr_glyph_l1_bwd_fused_eligible(c->dtype, E, c->hidden[0], c->vocab, c->n_fonts)) {
                p->l1f = true;
                // worst case over batch sizes <= max_batch: every 64-glyph block with the full-width slab, or 256 blocks of the narrowest
                p->o_l1f = carve(((size_t)((B + 63) / 64) + 256) * (size_t)((size_t)c->hidden[0] * E + c->hidden[0] + (size_t)(c->vocab + c->n_fonts) * E) * sizeof(float));
                p->o_w1t = carve((size_t)c->hidden[0] * E * 2);
            }
        }
        p->o_slab_e = carve(eb * (size_t)(c->vocab + c->n_fonts) * E * sizeof(float));
        {
            const size_t ncombo = (size_t)c->vocab * (c->n_fonts > 0 ? c->n_fonts : 1);
            if (c->dtype == AFR_BF16 && p->l1f && c->n_hidden >= 2 && ncombo <= 1024 && !(c->reserved & (64 | 16))) {
                p->combo_ok = true;
                p->h1c_ld = c->hidden[0] + (getenv("AFR_H1C_PAD") ? atoi(getenv("AFR_H1C_PAD")) : 0);      // kernel A/B measurements
                p->o_h1c = carve(ncombo * p->h1c_ld * 2); p->o_h0c = carve(ncombo * E * 2); p->o_cidx = carve(B * sizeof(int));
            }
        }
        if (c->n_hidden == 1 && afr_glyph1_eligible(E, c->hidden[0], Pix, c->vocab, c->n_fonts) &&
            afr_glyph1_lds_bytes(c->dtype, E, c->hidden[0], Pix, c->vocab + c->n_fonts) <= 160 * 1024) {
            p->fused1 = true;
            const size_t nblk = (size_t)afr_glyph1_max_blocks(c->dtype, (int)B, Pix);      // row blocks x column split
            p->o_slab1 = carve(nblk * (size_t)p->total * sizeof(float));
            p->o_loss = carve((1040 + nblk + 64) * sizeof(float));           // room for one loss partial per block
            if (c->dtype == AFR_BF16) {
                if (!p->l1f) p->o_w1t = carve((size_t)c->hidden[0] * E * 2);
                p->o_w2t = carve((size_t)Pix * c->hidden[0] * 2);
            }
        }
    } else if (c->kind == AFR_KIND_PIXEL) {
        const int d = E, ff = c->fc_dim, T = Pix, nf = c->n_fonts > 0 ? c->n_fonts : 0, C = nf > 0 ? 2 : 1;
        if (d > 512 || d % 64 || c->heads * 64 != d || ff <= 0 || ff % 8 || c->n_hidden < 1 || c->n_hidden > AFR_MAX_HIDDEN)
            { delete p; return fail(AFR_EUNSUPPORTED, "pixel transformer: d_model = 64 * heads <= 512, ff a multiple of 8, 1..%d blocks", AFR_MAX_HIDDEN); }
        add_param(p, "positional_encoding", {T, d});
        add_param(p, "embedding.weight", {c->vocab, d});
        if (nf > 0) add_param(p, "font_embedding.weight", {nf, d});
        char nm[64];
        for (int l = 0; l < c->n_hidden; ++l) {
            afr_plan::PixBlock b;
            auto addp = [&](const char* suffix, std::initializer_list<int64_t> shape) { snprintf(nm, sizeof nm, "layers.%d.%s", l, suffix); add_param(p, nm, shape); return p->params.back().off; };
            b.ln1g = addp("ln1.weight", {d}); b.ln1b = addp("ln1.bias", {d});
            b.win = addp("attn.in_proj_weight", {3 * d, d}); b.bin = addp("attn.in_proj_bias", {3 * d});
            b.wo = addp("attn.out_proj.weight", {d, d}); b.bo = addp("attn.out_proj.bias", {d});
            b.ln2g = addp("ln2.weight", {d}); b.ln2b = addp("ln2.bias", {d});
            b.w1 = addp("fc1.weight", {ff, d}); b.b1 = addp("fc1.bias", {ff});
            b.w2 = addp("fc2.weight", {d, ff}); b.b2 = addp("fc2.bias", {d});
            p->pix.push_back(b);
        }
        add_param(p, "ln_f.weight", {d}); add_param(p, "ln_f.bias", {d});
        add_param(p, "fc_output.weight", {1, d}); add_param(p, "fc_output.bias", {1});
        p->px_pos = off_of(p, "positional_encoding"); p->px_emb = off_of(p, "embedding.weight");
        p->px_font = nf > 0 ? off_of(p, "font_embedding.weight") : -1;
        p->px_lnfg = off_of(p, "ln_f.weight"); p->px_lnfb = off_of(p, "ln_f.bias");
        p->px_wout = off_of(p, "fc_output.weight"); p->px_bout = off_of(p, "fc_output.bias");
        const size_t rows = B * (size_t)T;
        if (c->dtype == AFR_BF16 && (rows * (size_t)(ff > d ? ff : d) * 2 >= (1ull << 31)))
            { delete p; return fail(AFR_EUNSUPPORTED, "max_batch %d x %d tokens: an activation operand would reach 2 GiB in bf16", c->max_batch, T); }
        if (c->dtype == AFR_BF16) p->o_shadow = carve((size_t)p->total * 2);
        p->o_err = carve(256);
        p->o_loss = carve((1040 + 1040) * sizeof(float));
        p->o_ctx = carve(B * C * d * ab);
        const size_t nbp = (size_t)afr_pixel_bwd_blocks((long long)rows);
        for (int l = 0; l < c->n_hidden; ++l) {
            afr_plan::PixSave sv;
            sv.hin = carve(rows * d * sizeof(float)); sv.h1 = carve(rows * d * sizeof(float));
            sv.n1 = carve(rows * d * ab); sv.q = carve(rows * d * ab); sv.o = carve(rows * d * ab); sv.n2 = carve(rows * d * ab);
            sv.kv = carve(B * C * 2 * d * ab); sv.f = carve(rows * ff * ab);
            sv.fbits = c->dtype == AFR_BF16 ? carve(rows * (size_t)(ff / 8)) : 0;      // ReLU gate of F, one byte per 8 columns (bf16 mode)
            sv.ln1p = carve(nbp * 2 * d * sizeof(float)); sv.ln2p = carve(nbp * 2 * d * sizeof(float));
            p->pxs.push_back(sv);
            const afr_plan::PixBlock& b = p->pix[l];
            const struct { int N, K; int64_t w, bo; long long red; } lin[5] = {
                {d, d, b.win, b.bin, (long long)rows}, {2 * d, d, b.win + (int64_t)d * d, b.bin + d, (long long)(B * C)}, {d, d, b.wo, b.bo, (long long)rows},
                {ff, d, b.w1, b.b1, (long long)rows}, {d, ff, b.w2, b.b2, (long long)rows}};
            for (const auto& q_ : lin) {
                afr_plan::Layer ly; ly.N = q_.N; ly.K = q_.K; ly.w_off = q_.w; ly.b_off = q_.bo;
                ly.sk = choose_splitk(q_.N, q_.K, (int)(q_.red > 0x7fffffff ? 0x7fffffff : q_.red));
                if (ly.sk > 1) { ly.o_slab_w = carve((size_t)ly.sk * q_.N * q_.K * sizeof(float)); ly.o_slab_b = carve((size_t)ly.sk * q_.N * sizeof(float)); }
                p->pxl.push_back(ly);
            }
        }
        p->o_a = carve(rows * d * ab); p->o_hf = carve(rows * d * sizeof(float));
        p->o_u = carve(rows * sizeof(float));
        p->o_dh = carve(rows * d * sizeof(float)); p->o_dht = c->dtype == AFR_BF16 ? carve(rows * d * ab) : 0;
        p->o_df = carve(rows * ff * ab); p->o_dn = carve(rows * d * ab); p->o_dq = carve(rows * d * ab);
        const size_t chunks = ((size_t)T + afr_pixel_attn_chunk(T) - 1) / afr_pixel_attn_chunk(T);
        p->o_dkvp = carve(B * chunks * 4 * d * sizeof(float)); p->o_dkv = carve(B * 4 * d * sizeof(float)); p->o_dkvt = carve(B * C * 2 * d * ab);
        p->o_dctxt = carve(B * C * d * ab); p->o_dctx = carve(B * C * d * sizeof(float));
        p->o_headp = carve(nbp * 4 * d * sizeof(float));
    } else {
        delete p;
        return fail(AFR_EINVAL, "unknown model kind %d", c->kind);
    }
    if (c->dtype == AFR_BF16) {
        // every bf16 GEMM operand (activations [max_batch][width], weight shadows) is addressed with 32-bit byte offsets
        long long widest = Pix;
        for (const auto& l : p->layers) { if (l.N > widest) widest = l.N; if (l.K > widest) widest = l.K; }
        if ((long long)c->max_batch * widest * 2 >= (1ll << 31)) {
            const long long lim = ((1ll << 31) - 1) / (widest * 2);
            delete p;
            return fail(AFR_EUNSUPPORTED, "max_batch %d too large for bf16 mode: an activation operand would reach 2 GiB (limit %lld rows of %lld)", c->max_batch, lim, widest);
        }
        for (const auto& l : p->layers)
            if ((long long)l.N * l.K * 2 >= (1ll << 31)) { delete p; return fail(AFR_EUNSUPPORTED, "a %d x %d weight is 2 GiB or more in bf16", l.N, l.K); }
    }
    p->ws_need = off;
    *out = p;
    return AFR_OK;
}

extern "C" int afr_plan_destroy(afr_plan* p) {
    if (!p) return AFR_OK;
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    for (auto& r : p->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    delete p;
    return AFR_OK;
}

extern "C" int64_t afr_param_elems(const afr_plan* p) { return p ? p->total : 0; }
extern "C" int afr_param_count(const afr_plan* p) { return p ? (int)p->params.size() : 0; }
extern "C" int afr_param_info(const afr_plan* p, int i, char* name, int cap, int64_t* offset, int64_t* numel,
                              int32_t* ndim, int64_t shape[4]) {
    if (!p || i < 0 || i >= (int)p->params.size()) return fail(AFR_EINVAL, "parameter index out of range");
    const Tensor& t = p->params[i];
    if (name && cap > 0) { strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (offset) *offset = t.off;
    if (numel) *numel = t.numel;
    if (ndim) *ndim = t.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = t.shape[k];
    return AFR_OK;
}
extern "C" size_t afr_workspace_bytes(const afr_plan* p) { return p ? p->ws_need : 0; }

extern "C" int afr_bind(afr_plan* p, float* params, float* grads, float* m, float* v, void* ws, size_t ws_bytes) {
    if (!p || !params || !ws) return fail(AFR_EINVAL, "params and workspace are required");
    if (ws_bytes < p->ws_need) return fail(AFR_EINVAL, "workspace too small: %zu < %zu", ws_bytes, p->ws_need);
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ws) & 255)
        return fail(AFR_EINVAL, "buffers must be 256-byte aligned");
    const int dev = device_of(params);
    for (const void* q : {(const void*)grads, (const void*)m, (const void*)v, (const void*)ws}) {
        const int d = device_of(q);
        if (q && d >= 0 && dev >= 0 && d != dev) return fail(AFR_EINVAL, "buffers live on different devices (%d and %d)", dev, d);
    }
    p->device = dev;
    p->P = params; p->G = grads; p->M = m; p->V = v;
    p->ws = (char*)ws; p->ws_bytes = ws_bytes;
    p->have_du = false;
    p->wT_valid = false;
    p->shadow_cur = 0; p->step_on = false; p->adam_done.clear();
    for (auto& l : p->layers) {                          // cooperative split-K: arrival counters and launch count start at zero
        l.coop_epoch = 0;
        if (l.n_cnt) { DevGuard dg(dev); HIPCHK(hipMemset(p->ws + l.o_cnt, 0, (size_t)l.n_cnt * sizeof(unsigned))); }
    }
    if (p->have_fix) {                                   // arrival counters start at zero; the kernels re-arm them
        DevGuard dg(dev);
        HIPCHK(hipMemset(p->ws + p->o_fixcnt, 0, AFR_FIX_MAX_SLICES * sizeof(unsigned)));
    }
    return AFR_OK;
}

// ---------------------------------------------------------------------------------- profiling
static hipEvent_t ev_get(afr_plan* p) {
    if (!p->ev_pool.empty()) { hipEvent_t e = p->ev_pool.back(); p->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
static int tag_id(afr_plan* p, const char* tag) {
    for (size_t i = 0; i < p->prof_tags.size(); ++i)
        if (p->prof_tags[i] == tag) return (int)i;
    p->prof_tags.push_back(tag);
    return (int)p->prof_tags.size() - 1;
}
// What an (event, launch, event) bracket adds to the kernel's own duration.  Two events recorded back to back with
// nothing in between are ~4.5 us apart on this part (two command-processor packets); a bracketed launch pays for ONE of
// them beyond the kernel (the closing record), i.e. half that interval: measured against rocprofv3's kernel durations the
// raw brackets read 2.6 us high, the full interval subtracted 1.9 us low, half of it within 0.5 us.  Measured once per
// plan, the first time profiling is on, and subtracted from every bracket.
static void prof_calibrate(afr_plan* p, hipStream_t s) {
    p->prof_overhead_ms = 0.0;
    float best = 1e9f;
    for (int i = 0; i < 16; ++i) {
        hipEvent_t a = ev_get(p), b = ev_get(p);
        if (hipEventRecord(a, s) != hipSuccess || hipEventRecord(b, s) != hipSuccess || hipEventSynchronize(b) != hipSuccess) return;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms < best) best = ms;
        p->ev_pool.push_back(a); p->ev_pool.push_back(b);
    }
    if (best < 1e8f) p->prof_overhead_ms = 0.5 * best;
}
// "gemm_bf16<1,1,4>[1024x1024x8192]" -> "gemm_bf16<1,1,4>": the symbol rocprofv3 reports; launches of one symbol with
// different shapes are recorded apart (per-shape table) and summed per symbol when the dominant KERNEL is picked
static std::string symbol_of(const std::string& tag) { return tag.substr(0, tag.find('[')); }
struct ProfScope {
    afr_plan* p; hipStream_t s; ProfRec r; bool on;
    ProfScope(afr_plan* p_, hipStream_t s_, const char* tag, double flops, double bytes) : p(p_), s(s_), on(p_->prof_mode != 0) {
        if (!on) return;
        if (p->prof_overhead_ms < 0.0) prof_calibrate(p, s);
        r.tag = tag_id(p, tag);
        if (p->prof_mode >= 2 && symbol_of(p->prof_tags[r.tag]) != p->prof_only_sym) { on = false; return; }
        if (p->prof_mode == 3 && (p->prof_seen++ & 3u)) { on = false; return; }     // a sample: every 4th launch
        r.flops = flops; r.bytes = bytes; r.a = ev_get(p); r.b = ev_get(p);
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, s); p->prof.push_back(r); } }
};
static int prof_totals(afr_plan* p, std::vector<double>& tot, std::vector<double>& fl, std::vector<double>& by,
                       std::vector<int64_t>& cnt) {
    const size_t n = p->prof_tags.size();
    tot.assign(n, 0.0); fl.assign(n, 0.0); by.assign(n, 0.0); cnt.assign(n, 0);
    for (auto& r : p->prof) {
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        ms = ms > (float)p->prof_overhead_ms ? ms - (float)p->prof_overhead_ms : 0.f;
        tot[r.tag] += ms; fl[r.tag] += r.flops; by[r.tag] += r.bytes; cnt[r.tag]++;
    }
    return AFR_OK;
}
// per-symbol sums of a per-tag table
struct SymAgg { std::string sym; double tot = 0, fl = 0, by = 0; int64_t cnt = 0; };
static std::vector<SymAgg> prof_by_symbol(afr_plan* p, const std::vector<double>& tot, const std::vector<double>& fl,
                                          const std::vector<double>& by, const std::vector<int64_t>& cnt) {
    std::vector<SymAgg> out;
    for (size_t i = 0; i < tot.size(); ++i) {
        if (!cnt[i]) continue;
        const std::string sy = symbol_of(p->prof_tags[i]);
        size_t k = 0;
        while (k < out.size() && out[k].sym != sy) ++k;
        if (k == out.size()) { out.emplace_back(); out[k].sym = sy; }
        out[k].tot += tot[i]; out[k].fl += fl[i]; out[k].by += by[i]; out[k].cnt += cnt[i];
    }
    return out;
}
extern "C" int afr_profile_dominant(afr_plan* p, int mode) {
    if (!p) return fail(AFR_EINVAL, "null plan");
    if (mode == 2 || mode == 3) {     // keep timing only the kernel (symbol) that dominated the launches recorded so far
        if (p->prof.empty() && p->prof_only_sym.empty())
            return fail(AFR_ESTATE, "mode 2 needs a mode-1 recording to pick the dominant kernel from");
        if (!p->prof.empty() && p->prof_mode == 1) {
            std::vector<double> tot, fl, by; std::vector<int64_t> cnt;
            int rc = prof_totals(p, tot, fl, by, cnt);
            if (rc) return rc;
            const std::vector<SymAgg> ag = prof_by_symbol(p, tot, fl, by, cnt);
            size_t best = 0;
            for (size_t i = 1; i < ag.size(); ++i) if (ag[i].tot > ag[best].tot) best = i;
            p->prof_only_sym = ag[best].sym;
        }
    }
    for (auto& r : p->prof) { p->ev_pool.push_back(r.a); p->ev_pool.push_back(r.b); }
    p->prof.clear();
    p->prof_mode = mode;
    p->prof_seen = 0;
    return AFR_OK;
}
extern "C" int afr_profile_read(afr_plan* p, char* name, int cap, double* avg_ms, int64_t* launches, double* flops,
                                double* bytes) {
    if (!p) return fail(AFR_EINVAL, "null plan");
    if (p->prof.empty()) return fail(AFR_ESTATE, "no profiled launches recorded");
    std::vector<double> tot, fl, by; std::vector<int64_t> cnt;
    int rc = prof_totals(p, tot, fl, by, cnt);
    if (rc) return rc;
    const std::vector<SymAgg> ag = prof_by_symbol(p, tot, fl, by, cnt);
    size_t best = 0;
    for (size_t i = 1; i < ag.size(); ++i) if (ag[i].tot > ag[best].tot) best = i;
    if (name && cap > 0) { strncpy(name, ag[best].sym.c_str(), cap - 1); name[cap - 1] = 0; }
    if (avg_ms) *avg_ms = ag[best].tot / (double)ag[best].cnt;
    if (launches) *launches = ag[best].cnt;
    if (flops) *flops = ag[best].fl / (double)ag[best].cnt;
    if (bytes) *bytes = ag[best].by / (double)ag[best].cnt;
    return AFR_OK;
}

extern "C" int afr_profile_dump(afr_plan* p, char* buf, int cap) {
    if (!p || !buf || cap <= 0) return fail(AFR_EINVAL, "bad arguments");
    std::vector<double> tot, fl, by; std::vector<int64_t> cnt;
    int rc = prof_totals(p, tot, fl, by, cnt);
    if (rc) return rc;
    int n = 0;
    buf[0] = 0;
    for (size_t i = 0; i < tot.size() && n < cap - 1; ++i) {
        if (!cnt[i]) continue;
        n += snprintf(buf + n, cap - n, "%s\t%lld\t%.6f\t%.6f\t%.6g\t%.6g\n", p->prof_tags[i].c_str(), (long long)cnt[i], tot[i],
                      tot[i] / cnt[i], fl[i] / cnt[i], by[i] / cnt[i]);
    }
    return AFR_OK;
}

// ------------------------------------------------------------------------------------ helpers
struct FusedLoss { const void* target; int tdtype; int64_t mean_elems; float* loss_accum; };
struct FusedAdam { float *p, *m, *v; bf16_t* shadow; float decay, b1, b2, eps, step_size, rsqrt_bc2; };
struct CoopArgs { float* ws; unsigned* cnt; unsigned target; };
struct RowMaps { const int* a; const int* b; const int* aux;        // GemmParams::a_rowmap / b_rowmap / aux_rowmap
                 unsigned char* mask_out = nullptr; const unsigned char* mask_in = nullptr; int ldmask = 0; };   // ... mask_out / mask_in
// the bf16 weight shadow the GEMMs read / the one a fused optimizer step writes (the same buffer unless the plan has two)
static inline bf16_t* shadow_rd(const afr_plan* p) {
    if (p->cfg.dtype != AFR_BF16) return nullptr;
    return (bf16_t*)(p->ws + ((p->shadow_cur && p->o_shadow2) ? p->o_shadow2 : p->o_shadow));
}
static inline bf16_t* shadow_wr(const afr_plan* p) {
    if (p->cfg.dtype != AFR_BF16) return nullptr;
    if (!p->o_shadow2) return (bf16_t*)(p->ws + p->o_shadow);
    return (bf16_t*)(p->ws + (p->shadow_cur ? p->o_shadow : p->o_shadow2));
}
static inline const void* weight_ptr(const afr_plan* p, int64_t off) {
    if (p->cfg.dtype == AFR_BF16) return shadow_rd(p) + off;
    return p->P + off;
}
static int run_gemm(afr_plan* p, hipStream_t s, int flags, const void* A, const void* B, void* C, const float* bias,
                    const void* aux, int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int splitk,
                    long long slab_stride, float* colsum = nullptr, long long colsum_stride = 0, const FusedLoss* fl = nullptr,
                    const FusedAdam* fa = nullptr, const CoopArgs* coop = nullptr, const RowMaps* rm = nullptr) {
    if (p->cfg.dtype == AFR_BF16) {
        const bool ak = flags & AFR_GEMM_A_KSTRIDED, bk = flags & AFR_GEMM_B_KSTRIDED;
        const long long ea = (long long)(ak ? K : M) * lda * 2, ebb = (long long)(bk ? K : N) * ldb * 2;
        if (ea >= (1ll << 31) || ebb >= (1ll << 31)) return fail(AFR_EUNSUPPORTED, "bf16 GEMM operand of 2 GiB or more (%lld / %lld bytes)", ea, ebb);
    }
    GemmParams g;
    if (fa) {
        g.ad_p = fa->p; g.ad_m = fa->m; g.ad_v = fa->v; g.ad_shadow = fa->shadow;
        g.ad_decay = fa->decay; g.ad_b1 = fa->b1; g.ad_b2 = fa->b2; g.ad_eps = fa->eps; g.ad_step = fa->step_size; g.ad_rsqrt_bc2 = fa->rsqrt_bc2;
    }
    g.colsum = colsum; g.colsum_stride = colsum_stride;
    if (rm) { g.a_rowmap = rm->a; g.b_rowmap = rm->b; g.aux_rowmap = rm->aux; g.mask_out = rm->mask_out; g.mask_in = rm->mask_in; g.ldmask = rm->ldmask; }
    if (coop) { g.coop_ws = coop->ws; g.coop_cnt = coop->cnt; g.coop_target = coop->target; g.err = (uint32_t*)(p->ws + p->o_err); }
    if (fl) {
        float* scratch = (float*)(p->ws + p->o_loss);
        g.mse_target = fl->target; g.mse_target_dtype = fl->tdtype; g.mse_inv_n = (float)(1.0 / (double)fl->mean_elems);
        g.mse_partial = scratch + 1040; g.mse_counter = reinterpret_cast<unsigned*>(scratch + 1032); g.mse_loss_accum = fl->loss_accum;
    }
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
    g.flags = flags; g.splitk = splitk; g.slab_stride = slab_stride;
    const double eb = p->cfg.dtype == AFR_BF16 ? 2.0 : 4.0;
    const double ob = (flags & AFR_GEMM_OUT_BF16) ? 2.0 : 4.0;
    // algorithmic bytes: operands once + the product once (split-K partial slabs are an implementation choice, not
    // algorithmic output); with the fused optimizer the output is p,m,v read + p,m,v(,shadow) written
    const double out_bytes = fa ? (double)M * N * (24.0 + (fa->shadow ? 2.0 : 0.0)) : (splitk > 1 ? 4.0 : ob) * (double)M * N;
    char tag[96];
    const double fl_ = 2.0 * M * (double)N * K, by_ = eb * ((double)M * K + (double)N * K) + out_bytes;
    int hd = 0, fsk = 1;
    if (p->have_fix && p->cfg.dtype == AFR_BF16 && splitk == 1 && !fl && !fa && !colsum && (p->cfg.reserved & 8) &&
        afr_gemm_fix_plan(M, N, K, &hd, &fsk)) {
        g.head_tiles = hd; g.splitk = fsk;
        g.fix_ws = (float*)(p->ws + p->o_fix); g.fix_cnt = (unsigned*)(p->ws + p->o_fixcnt);
        snprintf(tag, sizeof tag, "gemm_bf16_group256[%dx%dx%d]", M, N, K);
        ProfScope ps(p, s, tag, fl_, by_);
        HIPCHK(afr_launch_gemm_fix(g, s));
        return AFR_OK;
    }
    {
        const char* kn = afr_gemm_kernel_name(p->cfg.dtype, g);
        if (strcmp(kn, "gemm_bf16_group256") == 0)     // a plain product on the 256x256 body: the operand orientation follows the shape
            snprintf(tag, sizeof tag, "%s[%dx%dx%d]<%d,%d>", kn, M, N, K, (flags & AFR_GEMM_A_KSTRIDED) ? 1 : 0, (flags & AFR_GEMM_B_KSTRIDED) ? 1 : 0);
        else snprintf(tag, sizeof tag, "%s[%dx%dx%d]", kn, M, N, K);
    }
    if (coop && !(p->defer && p->pend_tile256 && p->pend.empty() && afr_gemm_groupable(p->cfg.dtype, g)))
        return fail(AFR_ESTATE, "cooperative split-K product outside a 256x256 grouped launch");
    if (rm && rm->b && !(p->defer && p->pend_tile256 && afr_gemm_groupable(p->cfg.dtype, g)))
        return fail(AFR_ESTATE, "gathered k-strided operand outside a 256x256 grouped launch");
    if (p->defer && afr_gemm_groupable(p->cfg.dtype, g) && p->pend.size() < 4) {
        p->pend.push_back(g); p->pend_tag.push_back(tag); p->pend_flops += fl_; p->pend_bytes += by_;
        return AFR_OK;
    }
    ProfScope ps(p, s, tag, fl_, by_);
    HIPCHK(afr_launch_gemm(p->cfg.dtype, g, s));
    return AFR_OK;
}
// launch what run_gemm collected while p->defer was set: one grouped launch (or the plain one when only one qualified)
static int flush_gemms(afr_plan* p, hipStream_t s) {
    p->defer = false;
    if (p->pend.empty()) return AFR_OK;
    std::string tag = p->pend.size() > 1 ? (p->pend_tile256 ? "gemm_bf16_group256" : "gemm_bf16_group") : p->pend_tag[0];
    if (p->pend.size() > 1) {
        tag += "[";
        for (size_t i = 0; i < p->pend_tag.size(); ++i) tag += (i ? "+" : "") + p->pend_tag[i].substr(p->pend_tag[i].find('[') + 1, p->pend_tag[i].find(']') - p->pend_tag[i].find('[') - 1);
        tag += "]";
    }
    hipError_t e;
    {
        ProfScope ps(p, s, tag.c_str(), p->pend_flops, p->pend_bytes);
        e = afr_launch_gemm_group(p->cfg.dtype, p->pend.data(), (int)p->pend.size(), p->pend_tile256, s);
    }
    p->pend.clear(); p->pend_tag.clear(); p->pend_flops = p->pend_bytes = 0.0;
    if (e != hipSuccess) return fail(AFR_EHIP, "grouped GEMM launch: %s", hipGetErrorString(e));
    return AFR_OK;
}
// dW[N][K] = dy[B][N]^T . a[B][K] and db[N] = sum_b dy, reduced over the batch in ONE GEMM launch (the bias gradient
// is the column sum of the A tiles the kernel already stages).  Small outputs use split-K partial slabs, summed later
// by the grouped reduce; large ones (fc_output of the sheet model) write the gradient buffer directly.
static int run_dw(afr_plan* p, hipStream_t s, afr_plan::Layer& l, const void* dy, const void* a, int Bn, RTable& rt, int sk_want = 0,
                  bool coop = false, const int* a_rows = nullptr, int a_ld = 0) {
    const RowMaps rmv{nullptr, a_rows, nullptr};
    const RowMaps* rm = a_rows ? &rmv : nullptr;          // the layer's input rows are gathered from a table (k-strided B operand)
    if (rm && !coop) return fail(AFR_ESTATE, "gathered weight-gradient operand needs the cooperative 256x256 launch");
    const int fl = AFR_GEMM_A_KSTRIDED | AFR_GEMM_B_KSTRIDED;
    const int N = l.N, K = l.K;
    int sk = sk_want > 0 ? sk_want : choose_splitk(N, K, Bn);
    if (sk > l.sk) sk = l.sk;
    if (sk == 1) return run_gemm(p, s, fl, dy, a, p->G + l.w_off, nullptr, nullptr, N, K, Bn, N, K, K, 0, 1, 0, p->G + l.b_off, 0);
    float* sw = (float*)(p->ws + l.o_slab_w);
    float* sb = (float*)(p->ws + l.o_slab_b);
    if (coop) {
        // cooperative split-K: the slices meet inside the launch; the product leaves as the finished gradient, or -- during a
        // fused optimizer step -- as the AdamW update of this weight (its new bf16 copy goes to the write shadow, which the
        // launch's input-gradient workgroups do not read)
        CoopArgs ca{sw, (unsigned*)(p->ws + l.o_cnt), ++l.coop_epoch * (unsigned)sk};
        FusedAdam fa{p->P + l.w_off, p->M + l.w_off, p->V + l.w_off, shadow_wr(p) ? shadow_wr(p) + l.w_off : nullptr,
                     p->st_decay, p->st_b1, p->st_b2, p->st_eps, p->st_step, p->st_rsqrt_bc2};
        int rc = run_gemm(p, s, fl, dy, a, p->G + l.w_off, nullptr, nullptr, N, K, Bn, N, a_rows ? a_ld : K, K, 0, sk, 0, sb, N, nullptr,
                          p->step_on ? &fa : nullptr, &ca, rm);
        if (rc) return rc;
        if (p->step_on) p->adam_done.push_back(l.w_off);
        afr_rtable_add(rt, p->G + l.b_off, sb, sk, N, N);
        return AFR_OK;
    }
    int rc = run_gemm(p, s, fl, dy, a, sw, nullptr, nullptr, N, K, Bn, N, K, K, 0, sk, (long long)N * K, sb, N);
    if (rc) return rc;
    afr_rtable_add(rt, p->G + l.w_off, sw, sk, (long long)N * K, (long long)N * K);
    afr_rtable_add(rt, p->G + l.b_off, sb, sk, N, N);
    return AFR_OK;
}
static int run_reduce_group(afr_plan* p, hipStream_t s, const RTable& rt) {
    double bytes = 0;
    for (int i = 0; i < rt.nseg; ++i) bytes += 16.0 * rt.seg[i].n4 * (rt.seg[i].nslabs + 1);
    ProfScope ps(p, s, "reduce_group", 0.0, bytes);
    HIPCHK(afr_launch_reduce_group(rt, s));
    return AFR_OK;
}

extern "C" int afr_sync_params(afr_plan* p, void* stream) {
    if (!p || !p->P) return fail(AFR_ESTATE, "plan has no bound parameters");
    DevGuard dg(p->device);
    p->wT_valid = false;
    if (p->cfg.dtype != AFR_BF16) return AFR_OK;
    HIPCHK(afr_launch_f32_to_bf16(p->P, shadow_rd(p), p->total, (hipStream_t)stream));
    return AFR_OK;
}

static SheetDrop make_drop(const afr_plan* p, int training, uint64_t step) {
    SheetDrop d;
    const afr_config& c = p->cfg;
    d.training = training;
    d.key_e = afr_dropout_key(c.seed, step, AFR_STREAM_EMBED, (uint64_t)c.rank);
    d.key_a = afr_dropout_key(c.seed, step, AFR_STREAM_ATTN, (uint64_t)c.rank);
    d.key_f = afr_dropout_key(c.seed, step, AFR_STREAM_FC, (uint64_t)c.rank);
    d.thr_e = afr_keep_threshold(1.f - c.p_embed);
    d.thr_a = afr_keep_threshold(1.f - c.p_attn);
    d.thr_f = afr_keep_threshold(1.f - c.p_fc);
    d.sc_e = 1.f / (1.f - c.p_embed);
    d.sc_a = 1.f / (1.f - c.p_attn);
    d.sc_f = 1.f / (1.f - c.p_fc);
    d.save = training ? (float*)(p->ws + p->o_save) : nullptr;     // only a training forward leaves o + softmax stats behind
    return d;
}
static SheetParams sheet_params(const afr_plan* p) {
    SheetParams sp;
    sp.pos = p->P + p->s_pos; sp.emb = p->P + p->s_emb; sp.w_in = p->P + p->s_win; sp.b_in = p->P + p->s_bin;
    sp.w_o = p->P + p->s_wo; sp.b_o = p->P + p->s_bo; sp.ln_g = p->P + p->s_g; sp.ln_b = p->P + p->s_b;
    sp.w1 = p->P + p->s_w1; sp.b1 = p->P + p->s_b1;
    return sp;
}

// How a glyph layer's gradient pair (dW + dX) leaves at batch B: as ONE grouped launch (sk_group > 0) on 256x256 tiles
// (tile256) with the weight gradient's slices meeting inside the launch (coop), or as separate launches (all zero).
struct PairPlan { int sk_group = 0, tile256 = 0; bool coop = false; };
static PairPlan pair_plan_for(const afr_plan* p, const afr_plan::Layer& l, int B) {
    PairPlan pp;
    const afr_config& c = p->cfg;
    if (c.dtype != AFR_BF16 || (c.reserved & 2)) return pp;
    const long long dx_tiles = (long long)((B + 255) / 256) * ((l.K + 127) / 128);
    if (dx_tiles >= 232 && l.N >= 256) afr_gemm_pair_plan(B, l.N, l.K, &pp.tile256, &pp.sk_group);
    if (pp.sk_group > l.sk) { pp.sk_group = 0; pp.tile256 = 0; }          // the plan's slab space bounds the split
    // with 256x256 tiles the weight gradient's split-K slices are summed inside the launch (cooperative split-K): no slabs
    // for the grouped reduce to re-read; config.reserved bit 5 keeps the slab path (A/B measurements, parity cross-checks)
    pp.coop = pp.tile256 && (pp.sk_group == 2 || pp.sk_group == 4 || pp.sk_group == 8) && !(c.reserved & 32) &&
              (long long)((l.N + 255) / 256) * ((l.K + 255) / 256) * pp.sk_group <= 256;
    return pp;
}
// A training forward at batch B runs the first layer as a combination table when every consumer of h1 / h0 can gather: the
// second layer's forward on the 256x128 ring kernel, its gradient pair as a cooperative 256x256 launch, the fused first-layer
// backward.
static bool combo_for(const afr_plan* p, int B) {
    if (!p->combo_ok || p->layers.size() < 3) return false;
    const auto& l2 = p->layers[1];
    return afr_gemm_wide_ok(B, l2.N, l2.K) && pair_plan_for(p, l2, B).coop;
}

// ------------------------------------------------------------------------------------- forward
static int forward_impl(afr_plan* p, const int64_t* x, const int64_t* font, int B, int L, float* y, int training,
                        uint64_t step, void* stream, const FusedLoss* fl) {
    if (!p || !p->P) return fail(AFR_ESTATE, "plan has no bound parameters");
    DevGuard dg(p->device);
    if (!x) return fail(AFR_EINVAL, "x is null");
    if (B <= 0 || B > p->cfg.max_batch) return fail(AFR_EINVAL, "batch %d outside 1..max_batch=%d", B, p->cfg.max_batch);
    hipStream_t s = (hipStream_t)stream;
    const afr_config& c = p->cfg;
    const int Pix = c.out_h * c.out_w;
    const int ob = c.dtype == AFR_BF16 ? AFR_GEMM_OUT_BF16 : 0;
    uint32_t* err = (uint32_t*)(p->ws + p->o_err);
    void* u = p->ws + p->o_u;
    if (c.kind == AFR_KIND_SHEET) {
        if (L <= 0) return fail(AFR_EINVAL, "sequence length must be positive");
        const int Lc = L < c.max_length ? L : c.max_length;               // model.py:163-164
        SheetDims d{Lc, c.max_length, c.embed_dim, c.heads, c.fc_dim, c.vocab};
        void* z = p->ws + p->o_z;
        {
            ProfScope ps(p, s, "sheet_fwd", 2.5e6 * B, 0.0);        // ~2.5 MFLOP of f32 VALU work per string
            HIPCHK(afr_launch_sheet_fwd(c.dtype, d, sheet_params(p), make_drop(p, training, step), x, L, B, z, c.ln_eps, err, s));
        }
        const int Kz = c.max_length * c.fc_dim;
        int rc = run_gemm(p, s, AFR_GEMM_BIAS | ob, z, weight_ptr(p, p->s_wout), u, p->P + p->s_bout, nullptr, B, Pix, Kz,
                          Kz, Kz, Pix, 0, 1, 0, nullptr, 0, fl);
        if (rc) return rc;
        p->last_L = Lc;
        p->last_ldx = L;
    } else if (c.kind == AFR_KIND_PIXEL) {
        // BASELINE configs[4] (oracle.pixel_forward).  Token-wise kernels in pixel.hip, every Linear on the GEMM
        // kernels; the residual stream stays f32; every block keeps what its backward needs (its two residual inputs, LayerNorm
        // outputs, q, k|v, attention output, ReLU output).  No dropout in this model: `training` changes nothing.
        if (c.n_fonts > 0 && !font) return fail(AFR_EINVAL, "font ids are required when n_fonts > 0");
        const int d = c.embed_dim, ff = c.fc_dim, T = Pix, C = c.n_fonts > 0 ? 2 : 1;
        const long long rows = (long long)B * T;
        void *ctx = p->ws + p->o_ctx, *a = p->ws + p->o_a;
        {
            ProfScope ps(p, s, "pixel_ctx", 0.0, 0.0);
            HIPCHK(afr_launch_pixel_ctx(c.dtype, p->P + p->px_emb, p->px_font >= 0 ? p->P + p->px_font : nullptr, x, font, B, d, c.vocab, c.n_fonts, ctx, err, s));
        }
        int rc;
        for (int l = 0; l < c.n_hidden; ++l) {
            const afr_plan::PixBlock& b = p->pix[l];
            const afr_plan::PixSave& sv = p->pxs[l];
            float *hin = (float*)(p->ws + sv.hin), *h1 = (float*)(p->ws + sv.h1);
            void *n1 = p->ws + sv.n1, *q = p->ws + sv.q, *o = p->ws + sv.o, *n2 = p->ws + sv.n2, *kv = p->ws + sv.kv, *f = p->ws + sv.f;
            {
                ProfScope ps(p, s, "pixel_add_ln", 0.0, (double)rows * d * (8.0 + p->act_bytes));
                HIPCHK(afr_launch_pixel_add_ln(c.dtype, l == 0 ? nullptr : (const float*)(p->ws + p->pxs[l - 1].h1), hin, l == 0 ? p->P + p->px_pos : nullptr,
                                               l == 0 ? nullptr : a, p->P + b.ln1g, p->P + b.ln1b, n1, rows, T, d, c.ln_eps, s));
            }
            // packed in-projection (model.py:144): rows [0, d) of in_proj_weight make q from the pixel tokens, rows [d, 3d) k | v from the context
            if ((rc = run_gemm(p, s, AFR_GEMM_BIAS | ob, n1, weight_ptr(p, b.win), q, p->P + b.bin, nullptr, (int)rows, d, d, d, d, d, 0, 1, 0))) return rc;
            if ((rc = run_gemm(p, s, AFR_GEMM_BIAS | ob, ctx, weight_ptr(p, b.win + (int64_t)d * d), kv, p->P + b.bin + d, nullptr, B * C, 2 * d, d, d, d, 2 * d, 0, 1, 0))) return rc;
            {
                ProfScope ps(p, s, "pixel_attn", 0.0, (double)rows * d * 2.0 * p->act_bytes);
                HIPCHK(afr_launch_pixel_attn(c.dtype, q, kv, o, rows, T, d, c.heads, C, s));
            }
            if ((rc = run_gemm(p, s, AFR_GEMM_BIAS | ob, o, weight_ptr(p, b.wo), a, p->P + b.bo, nullptr, (int)rows, d, d, d, d, d, 0, 1, 0))) return rc;
            {
                ProfScope ps(p, s, "pixel_add_ln", 0.0, (double)rows * d * (8.0 + 2.0 * p->act_bytes));
                HIPCHK(afr_launch_pixel_add_ln(c.dtype, hin, h1, nullptr, a, p->P + b.ln2g, p->P + b.ln2b, n2, rows, T, d, c.ln_eps, s));
            }
            // (bf16 mode: the epilogue also leaves the ReLU gate of the stored values as bits for the backward's input-gradient product)
            RowMaps rmf{nullptr, nullptr, nullptr};
            if (sv.fbits) { rmf.mask_out = (unsigned char*)(p->ws + sv.fbits); rmf.ldmask = ff / 8; }
            if ((rc = run_gemm(p, s, AFR_GEMM_BIAS | AFR_GEMM_RELU | ob, n2, weight_ptr(p, b.w1), f, p->P + b.b1, nullptr, (int)rows, ff, d, d, d, ff, 0, 1, 0,
                               nullptr, 0, nullptr, nullptr, nullptr, sv.fbits ? &rmf : nullptr))) return rc;
            if ((rc = run_gemm(p, s, AFR_GEMM_BIAS | ob, f, weight_ptr(p, b.w2), a, p->P + b.b2, nullptr, (int)rows, d, ff, ff, ff, d, 0, 1, 0))) return rc;
        }
        {
            ProfScope ps(p, s, "pixel_head", 0.0, (double)rows * d * (8.0 + p->act_bytes));
            HIPCHK(afr_launch_pixel_head(c.dtype, (const float*)(p->ws + p->pxs.back().h1), (float*)(p->ws + p->o_hf), a, p->P + p->px_lnfg, p->P + p->px_lnfb,
                                         p->P + p->px_wout, p->P + p->px_bout, (float*)u, y, rows, d, c.ln_eps, s));
        }
        p->last_x = x; p->last_font = font; p->last_B = B; p->last_L = 1; p->last_training = training; p->last_step = step;
        p->next_stage = 0; p->combo_on = false; p->mbits_on = false;
        p->have_du = false;
        if (fl) {        // the loss on the f32 pre-clamp output (model.py:156,268-270): du in place over u
            ProfScope ps(p, s, "mse_grad", 0.0, (double)rows * 9.0);
            HIPCHK(afr_launch_mse_grad(AFR_F32, u, fl->target, fl->tdtype, u, B, Pix, fl->mean_elems, fl->loss_accum, (float*)(p->ws + p->o_loss), s));
            p->have_du = true;
        }
        return AFR_OK;
    } else {
        if (c.n_fonts > 0 && !font) return fail(AFR_EINVAL, "font ids are required when n_fonts > 0");
        void* h = p->ws + p->o_act[0];
        const int nl = (int)p->layers.size();
        const float* femb = c.n_fonts > 0 ? p->P + p->font_off : nullptr;
        int first = 0;
        const bool combo = fl != nullptr && p->k0 && combo_for(p, B);
        const RowMaps rma{(const int*)(p->ws + p->o_cidx), nullptr, nullptr};
        if (combo) {
            // training step: the first layer as a combination table; the second layer gathers its input rows from it
            const auto& l = p->layers[0];
            ProfScope ps(p, s, "glyph_l1_combo", 0.0, (double)c.vocab * (c.n_fonts > 0 ? c.n_fonts : 1) * l.N * 2.0);
            HIPCHK(afr_launch_glyph_combo(p->P + p->emb_off, femb, p->P + l.w_off, p->P + l.b_off, x, font, B, c.embed_dim, l.N, c.vocab,
                                          c.n_fonts, (float*)(p->ws + p->o_table), p->ws + p->o_h1c, p->h1c_ld, p->ws + p->o_h0c,
                                          (int*)(p->ws + p->o_cidx), err, s, p->ws + p->o_w1t));
            h = p->ws + p->o_h1c;
            first = 1;
        } else if (p->k0) {
            // hidden layer 1 as a table gather (see glyph_table_kernel); also leaves h0 for the backward dW GEMM
            const auto& l = p->layers[0];
            ProfScope ps(p, s, "glyph_l1_fwd", 0.0, (double)B * l.N * p->act_bytes);
            HIPCHK(afr_launch_glyph_l1_fwd(c.dtype, p->P + p->emb_off, femb, p->P + l.w_off, p->P + l.b_off, x, font, B, c.embed_dim,
                                           l.N, c.vocab, c.n_fonts, (float*)(p->ws + p->o_table), h, p->ws + p->o_act[1], err, s,
                                           p->l1f ? (void*)(p->ws + p->o_w1t) : nullptr));
            h = p->ws + p->o_act[1];
            first = 1;
        } else {
            ProfScope ps(p, s, "glyph_embed", 0.0, 0.0);
            HIPCHK(afr_launch_glyph_embed(c.dtype, p->P + p->emb_off, femb, x, font, B, c.embed_dim, c.vocab, c.n_fonts, h, err, s));
        }
        const bool bits = fl != nullptr && c.dtype == AFR_BF16;
        for (int i = first; i < nl; ++i) {
            const auto& l = p->layers[i];
            const bool last = (i == nl - 1);
            void* outp = last ? u : (void*)(p->ws + p->o_act[i + 1]);
            RowMaps rmi{(combo && i == 1) ? rma.a : nullptr, nullptr, nullptr};
            if (bits && !last && p->o_mbits[i]) { rmi.mask_out = (unsigned char*)(p->ws + p->o_mbits[i]); rmi.ldmask = l.N / 8; }
            int rc = run_gemm(p, s, AFR_GEMM_BIAS | (last ? 0 : AFR_GEMM_RELU) | ob, h, weight_ptr(p, l.w_off), outp,
                              p->P + l.b_off, nullptr, B, l.N, l.K, (combo && i == 1) ? p->h1c_ld : l.K, l.K, l.N, 0, 1, 0, nullptr, 0,
                              last ? fl : nullptr, nullptr, nullptr, (rmi.a || rmi.mask_out) ? &rmi : nullptr);
            if (rc) return rc;
            h = outp;
        }
        p->last_L = 1;
    }
    if (y) {
        ProfScope ps(p, s, "clamp_out", 0.0, 0.0);
        HIPCHK(afr_launch_clamp_out(c.dtype, u, y, (long long)B * Pix, s));
    }
    p->last_x = x; p->last_font = font; p->last_B = B; p->last_training = training; p->last_step = step;
    p->next_stage = 0;
    p->combo_on = c.kind == AFR_KIND_GLYPH && fl != nullptr && p->k0 && combo_for(p, B);
    p->mbits_on = c.kind == AFR_KIND_GLYPH && fl != nullptr && c.dtype == AFR_BF16;
    p->have_du = fl != nullptr;      // with the loss fused into the last layer's epilogue the buffer already holds du
    return AFR_OK;
}
extern "C" int afr_forward(afr_plan* p, const int64_t* x, const int64_t* font, int B, int L, float* y, int training,
                           uint64_t step, void* stream) {
    return forward_impl(p, x, font, B, L, y, training, step, stream, nullptr);
}

// --------------------------------------------------------------------------------- loss + grad
extern "C" int afr_loss_grad(afr_plan* p, const void* target, int tdtype, int B, int64_t mean_elems, float* loss_accum,
                             void* stream) {
    if (!p || !p->P) return fail(AFR_ESTATE, "plan has no bound parameters");
    DevGuard dg(p->device);
    if (!target || !loss_accum) return fail(AFR_EINVAL, "target and loss_accum are required");
    if (B != p->last_B) return fail(AFR_ESTATE, "loss_grad batch %d does not match the last forward (%d)", B, p->last_B);
    if (tdtype != AFR_TARGET_U8 && tdtype != AFR_TARGET_F32) return fail(AFR_EINVAL, "bad target dtype");
    if (mean_elems <= 0) return fail(AFR_EINVAL, "mean_elems must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int Pix = p->cfg.out_h * p->cfg.out_w;
    void* u = p->ws + p->o_u;
    const double tb = tdtype == AFR_TARGET_U8 ? 1.0 : 4.0;
    ProfScope ps(p, s, "mse_grad", 0.0, (double)B * Pix * (2.0 * p->act_bytes + tb));
    HIPCHK(afr_launch_mse_grad(p->cfg.kind == AFR_KIND_PIXEL ? AFR_F32 : p->cfg.dtype, u, target, tdtype, u, B, Pix, mean_elems, loss_accum,
                               (float*)(p->ws + p->o_loss), s));      // (the pixel transformer's pre-clamp output is f32 in both modes)
    p->have_du = true;
    return AFR_OK;
}

extern "C" int afr_set_output_grad(afr_plan* p, const float* dy, int B, void* stream) {
    if (!p || !p->P) return fail(AFR_ESTATE, "plan has no bound parameters");
    DevGuard dg(p->device);
    if (!dy) return fail(AFR_EINVAL, "dy is null");
    if (B != p->last_B) return fail(AFR_ESTATE, "batch %d does not match the last forward (%d)", B, p->last_B);
    HIPCHK(afr_launch_clamp_bwd(p->cfg.kind == AFR_KIND_PIXEL ? AFR_F32 : p->cfg.dtype, p->ws + p->o_u, dy, (long long)B * p->cfg.out_h * p->cfg.out_w, (hipStream_t)stream));
    p->have_du = true;
    return AFR_OK;
}

// ------------------------------------------------------------------------------------ backward
// Backward runs in STAGES, last layer first; each stage finishes a contiguous range of the flat gradient buffer
// (its own slab reduction included), so a data-parallel caller can start the all-reduce of that range while the
// next stage computes.  Glyph: one stage per Linear (the first layer's stage also does the embedding tables).
// Sheet: stage 0 = fc_output (dW + db), stage 1 = dz GEMM + fused front-end backward.  Pixel transformer: one stage per block.

// Backward of the pixel-token transformer (reverse of forward_impl's AFR_KIND_PIXEL branch; oracle.pixel_backward): du (f32,
// left in the u buffer by the loss) -> every parameter gradient.  Linears: the dW (+ fused db) and dX GEMMs of gemm.hip;
// token-wise reverses: pixel.hip.  Slab-produced gradients (split-K dW, bias partials, LayerNorm / head partials) are registered
// for a grouped reduce per block; the rest is written directly.
// One STAGE per block, last block first (stage 0 also runs the head's reverse, the last stage the positional table and the
// embedding rows): each finishes the contiguous range of the flat gradient buffer that holds its block's tensors (stage 0: from
// the last block to the end; the last stage: from the start to the second block), so a data-parallel caller can reduce it while the
// next stage computes.
static int pixel_backward(afr_plan* p, hipStream_t s, int stage, int64_t* g_off, int64_t* g_len) {
    const afr_config& c = p->cfg;
    RTable rt;
    rt.nseg = 0; rt.nblocks = 0; rt.adam = 0;
    const int B = p->last_B, d = c.embed_dim, ff = c.fc_dim, T = c.out_h * c.out_w, C = c.n_fonts > 0 ? 2 : 1;
    const long long rows = (long long)B * T;
    const int ob = c.dtype == AFR_BF16 ? AFR_GEMM_OUT_BF16 : 0;
    const int nbp = afr_pixel_bwd_blocks(rows);
    float* du = (float*)(p->ws + p->o_u);
    float* dh = (float*)(p->ws + p->o_dh);
    void* dhT = c.dtype == AFR_BF16 ? (void*)(p->ws + p->o_dht) : (void*)dh;      // the GEMM-operand copy of dh (f32 mode: dh itself)
    void *dfb = p->ws + p->o_df, *dn = p->ws + p->o_dn, *dq = p->ws + p->o_dq, *ctx = p->ws + p->o_ctx;
    int rc;
    const int nl = c.n_hidden, l = nl - 1 - stage;
    if (stage == 0) {
        float* hp = (float*)(p->ws + p->o_headp);
        ProfScope ps(p, s, "pixel_head_bwd", 0.0, (double)rows * d * 12.0);
        HIPCHK(afr_launch_pixel_head_bwd(c.dtype, du, (const float*)(p->ws + p->o_hf), p->P + p->px_lnfg, p->P + p->px_lnfb, p->P + p->px_wout, dh,
                                         c.dtype == AFR_BF16 ? dhT : nullptr, hp, rows, d, c.ln_eps, s));
        afr_rtable_add(rt, p->G + p->px_lnfg, hp, nbp, 4ll * d, d);
        afr_rtable_add(rt, p->G + p->px_lnfb, hp + d, nbp, 4ll * d, d);
        afr_rtable_add(rt, p->G + p->px_wout, hp + 2 * d, nbp, 4ll * d, d);
        afr_rtable_add(rt, p->G + p->px_bout, hp + 3 * d, nbp, 4ll * d, 4);       // (element 0 is db_out; the 3 after it are zero: a 64-aligned tensor)
    }
    {
        const afr_plan::PixBlock& b = p->pix[l];
        const afr_plan::PixSave& sv = p->pxs[l];
        afr_plan::Layer* L5 = &p->pxl[(size_t)l * 5];                 // q, kv, out-proj, fc1, fc2
        // ---- MLP:  h_out = h1 + fc2(relu(fc1(LN2(h1))))
        if ((rc = run_dw(p, s, L5[4], dhT, p->ws + sv.f, (int)rows, rt))) return rc;
        RowMaps rmf{nullptr, nullptr, nullptr};
        if (sv.fbits) { rmf.mask_in = (const unsigned char*)(p->ws + sv.fbits); rmf.ldmask = ff / 8; }
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | AFR_GEMM_RELU_MASK | ob, dhT, weight_ptr(p, b.w2), dfb, nullptr, p->ws + sv.f, (int)rows, ff, d, d, ff, ff, ff, 1, 0,
                           nullptr, 0, nullptr, nullptr, nullptr, sv.fbits ? &rmf : nullptr))) return rc;
        if ((rc = run_dw(p, s, L5[3], dfb, p->ws + sv.n2, (int)rows, rt))) return rc;
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, dfb, weight_ptr(p, b.w1), dn, nullptr, nullptr, (int)rows, d, ff, ff, d, d, 0, 1, 0))) return rc;
        {
            float* lp = (float*)(p->ws + sv.ln2p);
            ProfScope ps(p, s, "pixel_ln_bwd", 0.0, (double)rows * d * (12.0 + 2.0 * p->act_bytes));
            HIPCHK(afr_launch_pixel_ln_bwd(c.dtype, dn, (const float*)(p->ws + sv.h1), p->P + b.ln2g, dh, c.dtype == AFR_BF16 ? dhT : nullptr, lp, rows, d, c.ln_eps, s));
            afr_rtable_add(rt, p->G + b.ln2g, lp, nbp, 2ll * d, d);
            afr_rtable_add(rt, p->G + b.ln2b, lp + d, nbp, 2ll * d, d);
        }
        // ---- attention:  h1 = hin + out_proj(softmax(q k^T) v)
        if ((rc = run_dw(p, s, L5[2], dhT, p->ws + sv.o, (int)rows, rt))) return rc;
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, dhT, weight_ptr(p, b.wo), dn, nullptr, nullptr, (int)rows, d, d, d, d, d, 0, 1, 0))) return rc;
        const int chunk = afr_pixel_attn_chunk(T), chunks = (T + chunk - 1) / chunk;
        float *dkvp = (float*)(p->ws + p->o_dkvp), *dkv = (float*)(p->ws + p->o_dkv);
        {
            ProfScope ps(p, s, "pixel_attn_bwd", 0.0, (double)rows * d * 3.0 * p->act_bytes);
            HIPCHK(afr_launch_pixel_attn_bwd(c.dtype, dn, p->ws + sv.q, p->ws + sv.kv, dq, dkvp, B, T, d, C, s));
        }
        // dk | dv of every context token: the chunk slabs summed in chunk order (one chunk: the kernel's output is the sum)
        if (chunks > 1) {
            ProfScope ps(p, s, "reduce", 0.0, (double)B * 4 * d * 4.0 * (chunks + 1));
            HIPCHK(afr_launch_reduce(dkv, dkvp, chunks, (long long)B * 4 * d, (long long)B * 4 * d, 1.f, 0, s));
        }
        // the GEMM operand [B*C][2d] in the activation dtype: row c of sample b = [dk_c | dv_c] (C = 1: the first 2d of the 4d)
        void* dkvT = p->ws + p->o_dkvt;
        HIPCHK(afr_launch_pixel_cast(c.dtype, dkvT, chunks > 1 ? dkv : dkvp, B, C * 2 * d, 4 * d, s));
        if ((rc = run_dw(p, s, L5[0], dq, p->ws + sv.n1, (int)rows, rt))) return rc;
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, dq, weight_ptr(p, b.win), dn, nullptr, nullptr, (int)rows, d, d, d, d, d, 0, 1, 0))) return rc;
        if ((rc = run_dw(p, s, L5[1], dkvT, ctx, B * C, rt))) return rc;
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, dkvT, weight_ptr(p, b.win + (int64_t)d * d), p->ws + p->o_dctxt, nullptr, nullptr, B * C, d, 2 * d, 2 * d, d, d, 0, 1, 0))) return rc;
        HIPCHK(afr_launch_pixel_accum(c.dtype, (float*)(p->ws + p->o_dctx), p->ws + p->o_dctxt, (long long)B * C * d, l == c.n_hidden - 1, s));
        {
            float* lp = (float*)(p->ws + sv.ln1p);
            ProfScope ps(p, s, "pixel_ln_bwd", 0.0, (double)rows * d * (12.0 + 2.0 * p->act_bytes));
            HIPCHK(afr_launch_pixel_ln_bwd(c.dtype, dn, (const float*)(p->ws + sv.hin), p->P + b.ln1g, dh, c.dtype == AFR_BF16 ? dhT : nullptr, lp, rows, d, c.ln_eps, s));
            afr_rtable_add(rt, p->G + b.ln1g, lp, nbp, 2ll * d, d);
            afr_rtable_add(rt, p->G + b.ln1b, lp + d, nbp, 2ll * d, d);
        }
        // a block registers 14 slab sets (5 weights, 5 biases, 4 LayerNorm vectors; + the head's 4 with the last block): summed per
        // block, since the grouped reduce takes 32 segments
        if ((rc = run_reduce_group(p, s, rt))) return rc;
        rt.nseg = 0; rt.nblocks = 0;
    }
    const int64_t lo = l == 0 ? 0 : p->pix[l].ln1g, hi = stage == 0 ? p->total : p->pix[l + 1].ln1g;
    if (g_off) *g_off = lo;
    if (g_len) *g_len = hi - lo;
    if (l > 0) return AFR_OK;
    // positional table: the sum over the batch of the gradient of the residual stream's first value (model.py:140-141 idiom)
    HIPCHK(afr_launch_reduce(p->G + p->px_pos, dh, B, (long long)T * d, (long long)T * d, 1.f, 0, s));
    HIPCHK(afr_launch_pixel_ctx_bwd((const float*)(p->ws + p->o_dctx), p->last_x, p->last_font, B, d, c.vocab, c.n_fonts, p->G + p->px_emb,
                                    p->px_font >= 0 ? p->G + p->px_font : nullptr, s));
    return AFR_OK;
}

static int backward_stage_impl(afr_plan* p, int stage, int64_t* g_off, int64_t* g_len, hipStream_t s, RTable* shared_rt) {
    const afr_config& c = p->cfg;
    const int B = p->last_B, Pix = c.out_h * c.out_w;
    const int ob = c.dtype == AFR_BF16 ? AFR_GEMM_OUT_BF16 : 0;
    void* du = p->ws + p->o_u;
    int rc;
    // slab reductions: flushed per stage (so the stage's gradient range is final), or deferred to ONE grouped launch
    // at the end of a monolithic afr_backward (shared_rt)
    RTable local_rt;
    local_rt.nseg = 0; local_rt.nblocks = 0; local_rt.adam = 0;
    RTable& rt = shared_rt ? *shared_rt : local_rt;
    auto flush = [&]() -> int { return shared_rt ? AFR_OK : run_reduce_group(p, s, rt); };
    if (c.kind == AFR_KIND_PIXEL) return pixel_backward(p, s, stage, g_off, g_len);     // (its slab reductions are flushed per stage inside)
    if (c.kind == AFR_KIND_SHEET) {
        const int Kz = c.max_length * c.fc_dim;
        void* z = p->ws + p->o_z;
        void* dz = p->ws + p->o_dz;
        if (stage == 0) {
            if ((rc = run_dw(p, s, p->layers[0], du, z, B, rt))) return rc;
            if ((rc = flush())) return rc;
            if (g_off) *g_off = p->s_wout;
            if (g_len) *g_len = p->total - p->s_wout;
            return AFR_OK;
        }
        if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, du, weight_ptr(p, p->s_wout), dz, nullptr, nullptr, B, Kz, Pix, Pix,
                           Kz, Kz, 0, 1, 0))) return rc;
        SheetDims d{p->last_L, c.max_length, c.embed_dim, c.heads, c.fc_dim, c.vocab};
        float* slabs = (float*)(p->ws + p->o_slab_e);
        // a slab is laid out like the flat buffer's first s_wout floats: the 10 small tensors (pos .. fc1.bias)
        SheetSlabOff so{(int)p->s_pos, (int)p->s_emb, (int)p->s_win, (int)p->s_bin, (int)p->s_wo, (int)p->s_bo, (int)p->s_g,
                        (int)p->s_b, (int)p->s_w1, (int)p->s_b1, (int)p->s_wout};
        {
            ProfScope ps(p, s, "sheet_bwd", 7.0e6 * B, 0.0);       // partial recompute + reverse: ~7 MFLOP of f32 work per string
            HIPCHK(afr_launch_sheet_bwd(c.dtype, d, sheet_params(p), make_drop(p, p->last_training, p->last_step), p->last_x,
                                        p->last_ldx, B, dz, c.ln_eps, slabs, so, s));
        }
        afr_rtable_add(rt, p->G, slabs, afr_sheet_blocks(B), (long long)so.total, (long long)so.total);
        if ((rc = flush())) return rc;
        if (g_off) *g_off = 0;
        if (g_len) *g_len = p->s_wout;
        return AFR_OK;
    }
    const int nl = (int)p->layers.size();
    const int i = nl - 1 - stage;
    auto& l = p->layers[i];
    const void* a = p->ws + p->o_act[i];
    // d(loss)/d(output of layer i): du for the last layer, else the ping-pong buffer the previous stage wrote
    const void* dy = stage == 0 ? du : (const void*)(p->ws + p->o_d[(stage - 1) & 1]);
    if (i == 0 && p->k0) {
        // folded first layer (glyph_l1_bwd_kernel): ONE weight-gradient GEMM against h0' = [h0 | one-hot] yields dW1, db1
        // and the per-table-row segment sums; a small kernel turns those into dEmb / dFont.  No input-gradient GEMM,
        // no scatter-add.
        const int K0 = p->k0, E = c.embed_dim;
        if (p->l1f && !(c.reserved & 16)) {
            // throughput mode: one kernel per 64 glyphs (gemm.hip: glyph_l1_bwd_fused_kernel) -> [dW1 | db1 | dTab] slabs
            float* sl = (float*)(p->ws + p->o_l1f);
            const int CS = afr_glyph_l1_bwd_fused_split(B, l.N), nb = afr_glyph_l1_bwd_fused_blocks(B, l.N), nc = l.N / CS;
            const long long st = afr_glyph_l1_bwd_fused_slab_floats(B, l.N, c.vocab, c.n_fonts);
            {
                ProfScope ps(p, s, "glyph_l1_bwd_fused", 2.0 * B * l.N * (2.0 * E + 1.0), (double)B * l.N * 2.0 + (double)nb * st * 4.0);
                if (p->combo_on) HIPCHK(afr_launch_glyph_l1_bwd_fused(dy, l.N, p->ws + p->o_h0c, E, p->ws + p->o_w1t, p->last_x, p->last_font, B, l.N,
                                                                      c.vocab, c.n_fonts, sl, s, (const int*)(p->ws + p->o_cidx)));
                else HIPCHK(afr_launch_glyph_l1_bwd_fused(dy, l.N, a, K0, p->ws + p->o_w1t, p->last_x, p->last_font, B, l.N, c.vocab, c.n_fonts, sl, s));
            }
            // block = (row block, column range): range cs's slabs are blocks cs, cs + CS, ...; every block has a dTab partial
            for (int cs = 0; cs < CS; ++cs) {
                afr_rtable_add(rt, p->G + l.w_off + (size_t)cs * nc * E, sl + (size_t)cs * st, nb / CS, st * CS, (long long)nc * E);
                afr_rtable_add(rt, p->G + l.b_off + (size_t)cs * nc, sl + (size_t)cs * st + (size_t)nc * E, nb / CS, st * CS, nc);
            }
            afr_rtable_add(rt, p->G + p->emb_off, sl + (size_t)nc * E + nc, nb, st, (long long)c.vocab * E);
            if (c.n_fonts > 0) afr_rtable_add(rt, p->G + p->font_off, sl + (size_t)nc * E + nc + (size_t)c.vocab * E, nb, st, (long long)c.n_fonts * E);
            if ((rc = flush())) return rc;
            if (g_off) *g_off = 0;
            if (g_len) *g_len = l.b_off + (l.N + 63) / 64 * 64;
            return AFR_OK;
        }
        int sk = choose_splitk(l.N, K0, B);
        if (sk > l.sk) sk = l.sk;
        float* sw = (float*)(p->ws + l.o_slab_w);
        float* sb = (float*)(p->ws + l.o_slab_b);
        const long long stride = (long long)l.N * K0;
        const int gfl = AFR_GEMM_A_KSTRIDED | AFR_GEMM_B_KSTRIDED;
        if (sk == 1) rc = run_gemm(p, s, gfl, dy, a, sw, nullptr, nullptr, l.N, K0, B, l.N, K0, K0, 0, 1, 0, p->G + l.b_off, 0);
        else rc = run_gemm(p, s, gfl, dy, a, sw, nullptr, nullptr, l.N, K0, B, l.N, K0, K0, 0, sk, stride, sb, l.N);
        if (rc) return rc;
        if (sk > 1) afr_rtable_add(rt, p->G + l.b_off, sb, sk, l.N, l.N);
        float* dw1 = (float*)(p->ws + p->o_dw1);
        float* part = (float*)(p->ws + p->o_slab_e);
        {
            ProfScope ps(p, s, "glyph_l1_bwd", 0.0, (double)sk * l.N * K0 * 4.0);
            HIPCHK(afr_launch_glyph_l1_bwd(sw, sk, stride, p->P + l.w_off, l.N, E, c.vocab, c.n_fonts, dw1, part, s));
        }
        const int nb = afr_glyph_l1_bwd_blocks(l.N);
        const long long pstride = (long long)(c.vocab + c.n_fonts) * E;
        afr_rtable_add(rt, p->G + l.w_off, dw1, 1, 0, (long long)l.N * E);
        afr_rtable_add(rt, p->G + p->emb_off, part, nb, pstride, (long long)c.vocab * E);
        if (c.n_fonts > 0) afr_rtable_add(rt, p->G + p->font_off, part + (size_t)c.vocab * E, nb, pstride, (long long)c.n_fonts * E);
        if ((rc = flush())) return rc;
        if (g_off) *g_off = 0;
        if (g_len) *g_len = l.b_off + (l.N + 63) / 64 * 64;
        return AFR_OK;
    }
    // A layer's two gradient products both consume dy and are independent: when the input-gradient product alone fills the
    // chip with 256x128 tiles they go out as ONE grouped launch, the weight gradient first and split so that one of its
    // blocks runs twice the K-tiles of an input-gradient block (half the slabs of the stand-alone choice; a CU draws
    // either one long block or two short ones).
    const PairPlan pp = pair_plan_for(p, l, B);
    const int sk_group = pp.sk_group, tile256 = pp.tile256;
    const bool coop = pp.coop;
    // the layer whose input is the first layer's output: after a combination-table forward its rows are gathered from H1c
    const bool gath = p->combo_on && i == 1;
    const int* cidx = gath ? (const int*)(p->ws + p->o_cidx) : nullptr;
    if (gath) {
        if (!coop) return fail(AFR_ESTATE, "combination-table forward without a cooperative gradient pair (batch changed?)");
        a = p->ws + p->o_h1c;
    }
    p->pend_tile256 = tile256;
    p->defer = sk_group > 0;
    if ((rc = run_dw(p, s, l, dy, a, B, rt, sk_group, coop, cidx, p->h1c_ld))) { p->defer = false; p->pend.clear(); p->pend_tag.clear(); p->pend_flops = p->pend_bytes = 0.0; return rc; }
    void* dx = p->ws + p->o_d[stage & 1];
    const int fl = AFR_GEMM_B_KSTRIDED | ob | (i > 0 ? AFR_GEMM_RELU_MASK : 0);
    RowMaps rmx{nullptr, nullptr, cidx};
    const bool use_bits = p->mbits_on && i >= 2 && p->o_mbits[i - 1] && (fl & AFR_GEMM_OUT_BF16);
    if (use_bits) { rmx.mask_in = (const unsigned char*)(p->ws + p->o_mbits[i - 1]); rmx.ldmask = l.K / 8; }
    if ((rc = run_gemm(p, s, fl, dy, weight_ptr(p, l.w_off), dx, nullptr, i > 0 ? a : nullptr, B, l.K, l.N, l.N, l.K, l.K,
                       gath ? p->h1c_ld : l.K, 1, 0, nullptr, 0, nullptr, nullptr, nullptr, (gath || use_bits) ? &rmx : nullptr))) { p->defer = false; p->pend.clear(); p->pend_tag.clear(); p->pend_flops = p->pend_bytes = 0.0; return rc; }
    if ((rc = flush_gemms(p, s))) return rc;
    const int64_t end = l.b_off + (l.N + 63) / 64 * 64;
    if (i > 0) {
        if ((rc = flush())) return rc;
        if (g_off) *g_off = l.w_off;
        if (g_len) *g_len = end - l.w_off;
        return AFR_OK;
    }
    float* slabs = (float*)(p->ws + p->o_slab_e);
    const int blocks = afr_embed_bwd_blocks(B);
    const long long rows = c.vocab + c.n_fonts;
    {
        ProfScope ps(p, s, "glyph_embed_bwd", 0.0, 0.0);
        HIPCHK(afr_launch_glyph_embed_bwd(c.dtype, dx, p->last_x, p->last_font, B, c.embed_dim, c.vocab, c.n_fonts, slabs, s));
    }
    const long long stride = rows * c.embed_dim;
    afr_rtable_add(rt, p->G + p->emb_off, slabs, blocks, stride, (long long)c.vocab * c.embed_dim);
    if (c.n_fonts > 0)
        afr_rtable_add(rt, p->G + p->font_off, slabs + (size_t)c.vocab * c.embed_dim, blocks, stride, (long long)c.n_fonts * c.embed_dim);
    if ((rc = flush())) return rc;
    if (g_off) *g_off = 0;
    if (g_len) *g_len = end;
    return AFR_OK;
}

extern "C" int afr_backward_stages(const afr_plan* p) {
    if (!p) return 0;
    if (p->cfg.kind == AFR_KIND_PIXEL) return p->cfg.n_hidden;
    return p->cfg.kind == AFR_KIND_SHEET ? 2 : (int)p->layers.size();
}

extern "C" int afr_backward_stage(afr_plan* p, int stage, int64_t* grad_offset, int64_t* grad_elems, void* stream) {
    if (!p || !p->P || !p->G) return fail(AFR_ESTATE, "plan has no bound parameter/gradient buffers");
    DevGuard dg(p->device);
    const int n = afr_backward_stages(p);
    if (stage < 0 || stage >= n) return fail(AFR_EINVAL, "stage %d outside 0..%d", stage, n - 1);
    if (stage == 0 && !p->have_du) return fail(AFR_ESTATE, "backward needs a forward + loss first");
    if (stage != p->next_stage) return fail(AFR_ESTATE, "stages must run in order: expected %d, got %d", p->next_stage, stage);
    int rc = backward_stage_impl(p, stage, grad_offset, grad_elems, (hipStream_t)stream, nullptr);
    if (rc) return rc;
    p->next_stage = stage + 1 == n ? 0 : stage + 1;
    if (stage + 1 == n) p->have_du = false;
    return AFR_OK;
}

extern "C" int afr_backward(afr_plan* p, void* stream) {
    if (!p || !p->P || !p->G) return fail(AFR_ESTATE, "plan has no bound parameter/gradient buffers");
    DevGuard dg(p->device);
    if (!p->have_du) return fail(AFR_ESTATE, "afr_backward needs afr_forward + afr_loss_grad first");
    const int n = afr_backward_stages(p);
    RTable rt;
    rt.nseg = 0; rt.nblocks = 0; rt.adam = 0;
    for (int st = 0; st < n; ++st) {
        int rc = backward_stage_impl(p, st, nullptr, nullptr, (hipStream_t)stream, &rt);
        if (rc) return rc;
    }
    int rc = run_reduce_group(p, (hipStream_t)stream, rt);
    if (rc) return rc;
    p->next_stage = 0;
    p->have_du = false;
    return AFR_OK;
}

// --------------------------------------------------------------------------------------- AdamW
extern "C" int afr_adamw_step(afr_plan* p, float lr, float b1, float b2, float eps, float wd, int64_t t, float gscale,
                              void* stream) {
    if (!p || !p->P || !p->G || !p->M || !p->V) return fail(AFR_ESTATE, "AdamW needs params, grads and both moments bound");
    DevGuard dg(p->device);
    if (t < 1) return fail(AFR_EINVAL, "t starts at 1");
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t));
    const float bc2 = (float)(1.0 - std::pow((double)b2, (double)t));
    bf16_t* shadow = shadow_rd(p);        // nothing reads the weights concurrently: updated in place
    ProfScope ps(p, s, "adamw", 0.0, (double)p->total * (shadow ? 30.0 : 28.0));
    HIPCHK(afr_launch_adamw(p->P, p->G, p->M, p->V, shadow, p->total, lr, b1, b2, eps, wd, bc1, bc2, gscale, s));
    p->wT_valid = false;
    return AFR_OK;
}

// Single-GPU optimiser step fused into the grouped slab reduction: every tensor whose gradient was produced as partial
// slabs (split-K dW, bias partials, embedding partials, the sheet model's small tensors) is updated in the kernel that
// sums its slabs -- the summed gradient is never stored; tensors whose gradient a GEMM wrote directly get the plain kernel.
static int reduce_and_step(afr_plan* p, hipStream_t s, RTable& rt, float lr, float b1, float b2, float eps, float wd, int64_t t,
                           int64_t skip_off = -1) {
    const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t));
    const float bc2 = (float)(1.0 - std::pow((double)b2, (double)t));
    bf16_t* shadow = shadow_wr(p);        // every tensor's new bf16 copy goes to the write shadow; the roles swap below
    rt.adam = 1; rt.ad_decay = 1.f - lr * wd; rt.ad_b1 = b1; rt.ad_b2 = b2; rt.ad_eps = eps; rt.ad_step = lr / bc1;
    rt.ad_rsqrt_bc2 = (float)(1.0 / std::sqrt((double)bc2));
    rt.gbase = p->G; rt.P = p->P; rt.M = p->M; rt.V = p->V; rt.shadow = shadow;
    p->wT_valid = false;
    int rc = run_reduce_group(p, s, rt);
    if (rc) return rc;
    for (const Tensor& tn : p->params) {
        if (tn.off == skip_off) continue;
        bool done = false;                 // updated inside its weight-gradient GEMM (cooperative split-K tail)
        for (int64_t o : p->adam_done) done = done || o == tn.off;
        if (done) continue;
        // covered by the (disjoint) segments of the grouped reduce -- possibly several per tensor (column ranges)?
        int64_t cov = 0;
        for (int i = 0; i < rt.nseg; ++i) {
            const int64_t so = rt.seg[i].dst - p->G, se = so + rt.seg[i].n4 * 4;
            const int64_t lo = so > tn.off ? so : tn.off, hi = se < tn.off + tn.numel ? se : tn.off + tn.numel;
            if (hi > lo) cov += hi - lo;
        }
        if (cov >= tn.numel) continue;
        const int64_t n = (tn.numel + 63) / 64 * 64;
        ProfScope ps(p, s, "adamw", 0.0, (double)n * 28.0);
        HIPCHK(afr_launch_adamw(p->P + tn.off, p->G + tn.off, p->M + tn.off, p->V + tn.off, shadow ? shadow + tn.off : nullptr, n, lr, b1,
                                b2, eps, wd, bc1, bc2, 1.f, s));
    }
    if (p->o_shadow2) p->shadow_cur ^= 1;   // every tensor has been rewritten: the write shadow is the current one now
    p->adam_done.clear();
    return AFR_OK;
}

// Single-GPU sheet step with the optimizer fused into the weight-gradient GEMM: fc_output.weight (99.98 % of the
// parameters) gets its AdamW update in the epilogue of dW = du^T.z, tile by tile, so its gradient is never written to
// or re-read from HBM (-8 bytes/parameter/step) and the update traffic overlaps other tiles' MFMA work.  dz = du.W
// runs FIRST because it must see the pre-update weights.  The small tensors take the ordinary AdamW kernel.
static bool fused_step_eligible(const afr_plan* p, int B) {
    return p->cfg.kind == AFR_KIND_SHEET && p->M && p->V && choose_splitk(p->layers[0].N, p->layers[0].K, B) == 1;
}
static int sheet_fused_step(afr_plan* p, hipStream_t s, float lr, float b1, float b2, float eps, float wd, int64_t t) {
    const afr_config& c = p->cfg;
    const int B = p->last_B, Pix = c.out_h * c.out_w, Kz = c.max_length * c.fc_dim;
    const int ob = c.dtype == AFR_BF16 ? AFR_GEMM_OUT_BF16 : 0;
    void* du = p->ws + p->o_u;
    void* z = p->ws + p->o_z;
    void* dz = p->ws + p->o_dz;
    bf16_t* shadow = shadow_rd(p);        // the sheet model keeps ONE shadow: its input-gradient product runs before the update
    const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t));
    const float bc2 = (float)(1.0 - std::pow((double)b2, (double)t));
    int rc;
    if ((rc = run_gemm(p, s, AFR_GEMM_B_KSTRIDED | ob, du, weight_ptr(p, p->s_wout), dz, nullptr, nullptr, B, Kz, Pix, Pix,
                       Kz, Kz, 0, 1, 0))) return rc;
    FusedAdam fa{p->P + p->s_wout, p->M + p->s_wout, p->V + p->s_wout, shadow ? shadow + p->s_wout : nullptr,
                 1.f - lr * wd, b1, b2, eps, lr / bc1, (float)(1.0 / std::sqrt((double)bc2))};
    if ((rc = run_gemm(p, s, AFR_GEMM_A_KSTRIDED | AFR_GEMM_B_KSTRIDED, du, z, p->G + p->s_wout, nullptr, nullptr, Pix, Kz, B, Pix, Kz,
                       Kz, 0, 1, 0, p->G + p->s_bout, 0, nullptr, &fa))) return rc;
    SheetDims d{p->last_L, c.max_length, c.embed_dim, c.heads, c.fc_dim, c.vocab};
    float* slabs = (float*)(p->ws + p->o_slab_e);
    SheetSlabOff so{(int)p->s_pos, (int)p->s_emb, (int)p->s_win, (int)p->s_bin, (int)p->s_wo, (int)p->s_bo, (int)p->s_g,
                    (int)p->s_b, (int)p->s_w1, (int)p->s_b1, (int)p->s_wout};
    {
        ProfScope ps(p, s, "sheet_bwd", 9.0e6 * B, 0.0);
        HIPCHK(afr_launch_sheet_bwd(c.dtype, d, sheet_params(p), make_drop(p, p->last_training, p->last_step), p->last_x,
                                    p->last_ldx, B, dz, c.ln_eps, slabs, so, s));
    }
    RTable rt;
    rt.nseg = 0; rt.nblocks = 0; rt.adam = 0;
    afr_rtable_add(rt, p->G, slabs, afr_sheet_blocks(B), (long long)so.total, (long long)so.total);
    // the ten small tensors are updated inside the slab reduction; fc_output.bias by the plain kernel; fc_output.weight
    // was updated in the dW GEMM above (skipped here)
    if ((rc = reduce_and_step(p, s, rt, lr, b1, b2, eps, wd, t, p->s_wout))) return rc;
    p->have_du = false;
    p->next_stage = 0;
    return AFR_OK;
}

// One fused launch for the whole forward + loss + backward of a small one-hidden-layer glyph net (glyph_fused.hip); the
// per-block partial gradients it leaves are registered in `rt` for the grouped reduce (with or without AdamW).
static int glyph1_fused(afr_plan* p, const int64_t* x, const int64_t* font, const void* target, int tdtype, int B,
                        int64_t mean_elems, float* loss_accum, hipStream_t s, RTable& rt) {
    const afr_config& c = p->cfg;
    if (!x) return fail(AFR_EINVAL, "x is null");
    if (B <= 0 || B > c.max_batch) return fail(AFR_EINVAL, "batch %d outside 1..max_batch=%d", B, c.max_batch);
    if (c.n_fonts > 0 && !font) return fail(AFR_EINVAL, "font ids are required when n_fonts > 0");
    const auto& l1 = p->layers[0];
    const auto& l2 = p->layers[1];
    const int E = c.embed_dim, N1 = l1.N, Pix = l2.N;
    const bool b16 = c.dtype == AFR_BF16;
    if (b16 && !p->wT_valid) {
        ProfScope ps(p, s, "glyph1_transpose", 0.0, 0.0);
        HIPCHK(afr_launch_transpose_bf16(p->P + l1.w_off, (bf16_t*)(p->ws + p->o_w1t), N1, E, s));
        HIPCHK(afr_launch_transpose_bf16(p->P + l2.w_off, (bf16_t*)(p->ws + p->o_w2t), Pix, N1, s));
        p->wT_valid = true;
    }
    Glyph1Args a;
    a.x = x; a.font = font; a.target = target; a.tdtype = tdtype;
    a.B = B; a.E = E; a.N1 = N1; a.P = Pix; a.vocab = c.vocab; a.n_fonts = c.n_fonts;
    a.emb = p->P + p->emb_off; a.femb = c.n_fonts > 0 ? p->P + p->font_off : nullptr;
    a.b1 = p->P + l1.b_off; a.b2 = p->P + l2.b_off;
    a.W1 = weight_ptr(p, l1.w_off); a.W2 = weight_ptr(p, l2.w_off);
    a.W1T = b16 ? (const void*)(p->ws + p->o_w1t) : (const void*)(p->P + l1.w_off);
    a.W2T = b16 ? (const void*)(p->ws + p->o_w2t) : (const void*)(p->P + l2.w_off);
    a.slabs = (float*)(p->ws + p->o_slab1); a.slab_stride = p->total;
    a.o_emb = p->emb_off; a.o_font = c.n_fonts > 0 ? p->font_off : 0; a.o_w1 = l1.w_off; a.o_b1 = l1.b_off; a.o_w2 = l2.w_off; a.o_b2 = l2.b_off;
    float* scratch = (float*)(p->ws + p->o_loss);
    a.inv_n = (float)(1.0 / (double)mean_elems); a.loss_partial = scratch + 1040; a.counter = reinterpret_cast<unsigned*>(scratch + 1032);
    a.loss_accum = loss_accum; a.err = (uint32_t*)(p->ws + p->o_err);
    const int R = afr_glyph1_rows(c.dtype), nrb = (B + R - 1) / R, cs = afr_glyph1_colsplit(c.dtype, B, Pix), nblk = nrb * cs;
    a.cs = cs;
    {
        const double fl = 6.0 * B * ((double)E * N1 + (double)N1 * Pix);
        ProfScope ps(p, s, b16 ? "glyph1_step<bf16>" : "glyph1_step<f32>", fl, (double)nblk * p->total * 4.0 + (double)B * Pix);
        HIPCHK(afr_launch_glyph1_step(c.dtype, a, s));
    }
    for (const Tensor& tn : p->params) {
        if (cs > 1 && (tn.off == l2.w_off || tn.off == l2.b_off)) {
            // fc_output's rows are split over the cs blocks of a row block: block (rb, pc) holds rows [pc, pc + 1) * Pix / cs, so
            // each row range is its own segment over the slabs rb * cs + pc
            const long long per = tn.off == l2.w_off ? (long long)(Pix / cs) * N1 : Pix / cs;
            for (int pc = 0; pc < cs; ++pc) {
                afr_rtable_add(rt, p->G + tn.off + pc * per, a.slabs + (size_t)pc * p->total + tn.off + pc * per, nrb, (long long)cs * p->total, per);
                if (b16 && tn.off == l2.w_off && rt.nseg > 0 && !rt.overflow) {
                    RSeg& sg = rt.seg[rt.nseg - 1];
                    sg.shT = (bf16_t*)(p->ws + p->o_w2t) + (size_t)pc * (Pix / cs); sg.tN = Pix; sg.tK = N1;      // columns pc * Pix / cs .. of W2^T [N1][Pix]
                }
            }
            continue;
        }
        afr_rtable_add(rt, p->G + tn.off, a.slabs + tn.off, nblk, p->total, (tn.numel + 3) / 4 * 4);
        // when the optimizer runs inside this reduce, it also keeps the transposed operand copies current
        if (b16 && rt.nseg > 0 && !rt.overflow) {
            RSeg& sg = rt.seg[rt.nseg - 1];
            if (tn.off == l1.w_off) { sg.shT = (bf16_t*)(p->ws + p->o_w1t); sg.tN = N1; sg.tK = E; }
            if (tn.off == l2.w_off) { sg.shT = (bf16_t*)(p->ws + p->o_w2t); sg.tN = Pix; sg.tK = N1; }
        }
    }
    p->last_x = x; p->last_font = font; p->last_B = B; p->last_L = 1; p->last_training = 1;
    p->have_du = false; p->next_stage = 0;
    return AFR_OK;
}

extern "C" int afr_forward_loss(afr_plan* p, const int64_t* x, const int64_t* font, const void* target, int tdtype, int B, int L,
                                int64_t mean_elems, float* loss_accum, uint64_t step, void* stream) {
    if (!target || !loss_accum) return fail(AFR_EINVAL, "target and loss_accum are required");
    if (tdtype != AFR_TARGET_U8 && tdtype != AFR_TARGET_F32) return fail(AFR_EINVAL, "bad target dtype");
    if (mean_elems <= 0) return fail(AFR_EINVAL, "mean_elems must be positive");
    FusedLoss fl{target, tdtype, mean_elems, loss_accum};
    return forward_impl(p, x, font, B, L, nullptr, 1, step, stream, &fl);
}

extern "C" int afr_train_step(afr_plan* p, const int64_t* x, const int64_t* font, const void* target, int tdtype, int B,
                              int L, int64_t mean_elems, float* loss_accum, uint64_t step, int do_step, float lr, float b1,
                              float b2, float eps, float wd, int64_t t, void* stream) {
    int rc;
    if (!target || !loss_accum) return fail(AFR_EINVAL, "target and loss_accum are required");
    if (tdtype != AFR_TARGET_U8 && tdtype != AFR_TARGET_F32) return fail(AFR_EINVAL, "bad target dtype");
    if (mean_elems <= 0) return fail(AFR_EINVAL, "mean_elems must be positive");
    if (!p || !p->P) return fail(AFR_ESTATE, "plan has no bound parameters");
    DevGuard dg(p->device);
    if (p->fused1 && !(p->cfg.reserved & 4)) {
        // small glyph net: forward + loss + backward in ONE launch, then the grouped reduce (with AdamW when stepping here)
        if (!p->G) return fail(AFR_ESTATE, "plan has no bound gradient buffer");
        RTable rt;
        if ((rc = glyph1_fused(p, x, font, target, tdtype, B, mean_elems, loss_accum, (hipStream_t)stream, rt))) return rc;
        if (do_step && p->M && p->V && !(p->cfg.reserved & 1)) {
            if (t < 1) return fail(AFR_EINVAL, "t starts at 1");
            rc = reduce_and_step(p, (hipStream_t)stream, rt, lr, b1, b2, eps, wd, t);
            p->wT_valid = rc == AFR_OK;                 // the reduce's AdamW wrote W1T / W2T beside the shadow
            return rc;
        }
        if ((rc = run_reduce_group(p, (hipStream_t)stream, rt))) return rc;
        if (do_step && (rc = afr_adamw_step(p, lr, b1, b2, eps, wd, t, 1.f, stream))) return rc;
        return AFR_OK;
    }
    // the loss and its gradient are computed in the epilogue of the last forward GEMM: u never touches HBM
    FusedLoss fl{target, tdtype, mean_elems, loss_accum};
    if ((rc = forward_impl(p, x, font, B, L, nullptr, 1, step, stream, &fl))) return rc;
    if (do_step && fused_step_eligible(p, B) && !(p->cfg.reserved & 1)) {
        if (t < 1) return fail(AFR_EINVAL, "t starts at 1");
        return sheet_fused_step(p, (hipStream_t)stream, lr, b1, b2, eps, wd, t);
    }
    if (do_step && p->M && p->V && !(p->cfg.reserved & 1) && p->cfg.kind != AFR_KIND_PIXEL) {
        if (t < 1) return fail(AFR_EINVAL, "t starts at 1");
        const int n = afr_backward_stages(p);
        RTable rt;
        rt.nseg = 0; rt.nblocks = 0; rt.adam = 0;
        {   // weight-gradient products with a cooperative split-K tail apply this step's AdamW themselves
            const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t)), bc2 = (float)(1.0 - std::pow((double)b2, (double)t));
            p->step_on = true; p->adam_done.clear();
            p->st_decay = 1.f - lr * wd; p->st_b1 = b1; p->st_b2 = b2; p->st_eps = eps; p->st_step = lr / bc1;
            p->st_rsqrt_bc2 = (float)(1.0 / std::sqrt((double)bc2));
        }
        for (int st = 0; st < n; ++st)
            if ((rc = backward_stage_impl(p, st, nullptr, nullptr, (hipStream_t)stream, &rt))) { p->step_on = false; return rc; }
        p->step_on = false;
        p->next_stage = 0;
        p->have_du = false;
        return reduce_and_step(p, (hipStream_t)stream, rt, lr, b1, b2, eps, wd, t);
    }
    if ((rc = afr_backward(p, stream))) return rc;
    if (do_step && (rc = afr_adamw_step(p, lr, b1, b2, eps, wd, t, 1.f, stream))) return rc;
    return AFR_OK;
}

extern "C" int afr_error_flags(afr_plan* p, void* stream, uint32_t* out) {
    if (!p || !p->ws || !out) return fail(AFR_EINVAL, "plan must be bound and out non-null");
    DevGuard dg(p->device);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(out, p->ws + p->o_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (*out) HIPCHK(hipMemsetAsync(p->ws + p->o_err, 0, sizeof(uint32_t), s));   // read-and-clear
    return AFR_OK;
}

extern "C" int afr_debug_copy(afr_plan* p, int which, void* dst, size_t cap, size_t* bytes_out, void* stream) {
    if (!p || !p->ws || !dst) return fail(AFR_EINVAL, "plan must be bound and dst non-null");
    DevGuard dg(p->device);
    if (p->last_B <= 0) return fail(AFR_ESTATE, "no forward has run yet");
    const afr_config& c = p->cfg;
    const size_t B = (size_t)p->last_B, ab = (size_t)p->act_bytes;
    size_t off = 0, bytes = 0;
    if (which == AFR_BUF_U) { off = p->o_u; bytes = B * c.out_h * c.out_w * ab; }
    else if (which == AFR_BUF_Z && c.kind == AFR_KIND_SHEET) { off = p->o_z; bytes = B * c.max_length * c.fc_dim * ab; }
    else if (which == AFR_BUF_DZ && c.kind == AFR_KIND_SHEET) { off = p->o_dz; bytes = B * c.max_length * c.fc_dim * ab; }
    else if ((which == AFR_BUF_W1T || which == AFR_BUF_W2T) && p->fused1 && c.dtype == AFR_BF16) {
        const int N1 = p->layers[0].N, Pix = p->layers[1].N;
        off = which == AFR_BUF_W1T ? p->o_w1t : p->o_w2t;
        bytes = (size_t)(which == AFR_BUF_W1T ? c.embed_dim * N1 : N1 * Pix) * 2;
    }
    else if (which >= AFR_BUF_ACT && c.kind == AFR_KIND_GLYPH && which - AFR_BUF_ACT < (int)p->o_act.size()) {
        const int i = which - AFR_BUF_ACT;
        off = p->o_act[i];
        bytes = B * (size_t)(i == 0 ? (p->k0 ? p->k0 : c.embed_dim) : c.hidden[i - 1]) * ab;
    } else return fail(AFR_EINVAL, "no such buffer %d for this model kind", which);
    if (bytes > cap) return fail(AFR_EINVAL, "destination too small: %zu < %zu", cap, bytes);
    HIPCHK(hipMemcpyAsync(dst, p->ws + off, bytes, hipMemcpyDefault, (hipStream_t)stream));
    if (bytes_out) *bytes_out = bytes;
    return AFR_OK;
}

extern "C" int afr_debug_sheet_gather(afr_plan* p, const int64_t* x, int B, int L, float* e0, void* stream) {
    if (!p || !p->P || !p->ws) return fail(AFR_ESTATE, "plan has no bound parameters");
    if (p->cfg.kind != AFR_KIND_SHEET) return fail(AFR_EINVAL, "the sheet model's gather: plan is not AFR_KIND_SHEET");
    if (!x || !e0) return fail(AFR_EINVAL, "x and e0 are required");
    if (B <= 0 || B > p->cfg.max_batch || L <= 0) return fail(AFR_EINVAL, "batch %d / length %d out of range", B, L);
    DevGuard dg(p->device);
    const afr_config& c = p->cfg;
    const int Lc = L < c.max_length ? L : c.max_length;
    SheetDims d{Lc, c.max_length, c.embed_dim, c.heads, c.fc_dim, c.vocab};
    SheetDrop dr = make_drop(p, 0, 0);
    dr.dbg_e0 = e0;
    HIPCHK(afr_launch_sheet_fwd(c.dtype, d, sheet_params(p), dr, x, L, B, p->ws + p->o_z, c.ln_eps, (uint32_t*)(p->ws + p->o_err),
                                (hipStream_t)stream));
    p->have_du = false;
    return AFR_OK;
}

// --------------------------------------------------------------------------- single-kernel ops
extern "C" int afr_op_gemm(int dtype, int flags, const void* A, const void* B, void* C, const float* bias, const void* aux,
                           int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int splitk, void* stream) {
    if (!A || !B || !C) return fail(AFR_EINVAL, "null operand");
    if (M <= 0 || N <= 0 || K <= 0 || splitk < 1) return fail(AFR_EINVAL, "bad GEMM extents");
    const int v = dtype == AFR_BF16 ? 8 : 4;
    const bool ak = flags & AFR_GEMM_A_KSTRIDED, bk = flags & AFR_GEMM_B_KSTRIDED;
    if ((ak ? M : K) % v || (bk ? N : K) % v || lda % v || ldb % v || N % v || ldc % v)
        return fail(AFR_EUNSUPPORTED, "contiguous extents and leading dimensions must be multiples of %d", v);
    if (splitk > 1 && (flags & (AFR_GEMM_BIAS | AFR_GEMM_RELU | AFR_GEMM_RELU_MASK | AFR_GEMM_OUT_BF16)))
        return fail(AFR_EINVAL, "split-K output is plain f32 partial slabs");
    if (dtype == AFR_BF16) {      // the LDS-DMA path addresses an operand with 32-bit byte offsets: 2 GiB per operand
        const long long ea = (long long)(ak ? K : M) * lda * 2, ebb = (long long)(bk ? K : N) * ldb * 2;
        if (ea >= (1ll << 31) || ebb >= (1ll << 31))
            return fail(AFR_EUNSUPPORTED, "a bf16 GEMM operand must be smaller than 2 GiB (A %lld bytes, B %lld bytes)", ea, ebb);
    }
    DevGuard dg(device_of(C));
    GemmParams g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.flags = flags; g.splitk = splitk;
    g.slab_stride = (long long)M * ldc;
    HIPCHK(afr_launch_gemm(dtype, g, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" size_t afr_op_gemm_fix_workspace_bytes(int M, int N, int head_tiles, int splitk) {
    const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
    const long long tail = tiles - (head_tiles < tiles ? (head_tiles < 0 ? 0 : head_tiles) : tiles);
    return (size_t)(tail * splitk) * AFR_FIX_SLICE_BYTES + (size_t)(tail > 0 ? tail : 1) * sizeof(unsigned);
}
extern "C" int afr_op_gemm_fix(int flags, const void* A, const void* B, void* C, const float* bias, const void* aux,
                               int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int head_tiles, int splitk,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !C || !workspace) return fail(AFR_EINVAL, "null operand");
    if (M <= 0 || N <= 0 || K <= 0 || splitk < 1 || splitk > 64 || head_tiles < 0) return fail(AFR_EINVAL, "bad GEMM extents");
    const bool ak = flags & AFR_GEMM_A_KSTRIDED, bk = flags & AFR_GEMM_B_KSTRIDED;
    if ((ak ? M : K) % 8 || (bk ? N : K) % 8 || lda % 8 || ldb % 8 || N % 8 || ldc % 8)
        return fail(AFR_EUNSUPPORTED, "contiguous extents and leading dimensions must be multiples of 8");
    const long long ea = (long long)(ak ? K : M) * lda * 2, ebb = (long long)(bk ? K : N) * ldb * 2;
    if (ea >= (1ll << 31) || ebb >= (1ll << 31))
        return fail(AFR_EUNSUPPORTED, "a bf16 GEMM operand must be smaller than 2 GiB (A %lld bytes, B %lld bytes)", ea, ebb);
    const size_t need = afr_op_gemm_fix_workspace_bytes(M, N, head_tiles, splitk);
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return fail(AFR_EINVAL, "workspace too small or misaligned: %zu < %zu", workspace_bytes, need);
    const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
    const long long tail = tiles - (head_tiles < tiles ? head_tiles : tiles);
    DevGuard dg(device_of(C));
    GemmParams g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.flags = flags; g.splitk = splitk;
    g.head_tiles = head_tiles;
    g.fix_ws = (float*)workspace;
    g.fix_cnt = (unsigned*)((char*)workspace + (size_t)(tail * splitk) * AFR_FIX_SLICE_BYTES);
    HIPCHK(afr_launch_gemm_fix(g, (hipStream_t)stream));
    return AFR_OK;
}
static bool pair_plan(int B, int N, int K, int* sk) {
    int tile256 = 0;
    *sk = 0;
    if (B <= 0 || N < 256 || K <= 0) return false;
    if ((long long)((B + 255) / 256) * ((K + 127) / 128) < 232) return false;          // as backward_stage_impl decides
    afr_gemm_pair_plan(B, N, K, &tile256, sk);
    return tile256 && (*sk == 2 || *sk == 4 || *sk == 8) && (long long)((N + 255) / 256) * ((K + 255) / 256) * *sk <= 256;
}
extern "C" int afr_op_gemm_pair_plan(int B, int N, int K, int* splitk, size_t* workspace_bytes) {
    int sk;
    if (!pair_plan(B, N, K, &sk)) return fail(AFR_EUNSUPPORTED, "%d x %d x %d does not run as a cooperative 256x256 pair", B, N, K);
    const size_t tiles = (size_t)((N + 255) / 256) * ((K + 255) / 256);
    if (splitk) *splitk = sk;
    if (workspace_bytes) *workspace_bytes = tiles * sk * AFR_FIX_SLICE_BYTES + align_up(tiles * sizeof(unsigned), 256);
    return AFR_OK;
}
extern "C" int afr_op_gemm_pair(const void* dy, const void* x, const void* W, const void* aux, float* dW, float* db_part, void* dX,
                                int B, int N, int K, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !x || !W || !dW || !db_part || !dX || !workspace) return fail(AFR_EINVAL, "null operand");
    int sk; size_t need;
    int rc = afr_op_gemm_pair_plan(B, N, K, &sk, &need);
    if (rc) return rc;
    if (N % 8 || K % 8) return fail(AFR_EUNSUPPORTED, "N and K must be multiples of 8");
    if (workspace_bytes < need || ((uintptr_t)workspace & 255)) return fail(AFR_EINVAL, "workspace too small or misaligned: %zu < %zu", workspace_bytes, need);
    if ((long long)B * N * 2 >= (1ll << 31) || (long long)B * K * 2 >= (1ll << 31) || (long long)N * K * 2 >= (1ll << 31))
        return fail(AFR_EUNSUPPORTED, "a bf16 GEMM operand must be smaller than 2 GiB");
    DevGuard dg(device_of(dW));
    hipStream_t s = (hipStream_t)stream;
    const size_t tiles = (size_t)((N + 255) / 256) * ((K + 255) / 256);
    unsigned* cnt = (unsigned*)((char*)workspace + tiles * sk * AFR_FIX_SLICE_BYTES);
    HIPCHK(hipMemsetAsync(cnt, 0, align_up(tiles * sizeof(unsigned), 256), s));
    GemmParams g[2];
    g[0].A = dy; g[0].B = x; g[0].C = dW; g[0].M = N; g[0].N = K; g[0].K = B; g[0].lda = N; g[0].ldb = K; g[0].ldc = K; g[0].ldaux = 0;
    g[0].bias = nullptr; g[0].aux = nullptr; g[0].flags = AFR_GEMM_A_KSTRIDED | AFR_GEMM_B_KSTRIDED; g[0].splitk = sk; g[0].slab_stride = 0;
    g[0].colsum = db_part; g[0].colsum_stride = N;
    g[0].coop_ws = (float*)workspace; g[0].coop_cnt = cnt; g[0].coop_target = (unsigned)sk; g[0].err = nullptr;
    g[1].A = dy; g[1].B = W; g[1].C = dX; g[1].M = B; g[1].N = K; g[1].K = N; g[1].lda = N; g[1].ldb = K; g[1].ldc = K; g[1].ldaux = K;
    g[1].bias = nullptr; g[1].aux = aux; g[1].flags = AFR_GEMM_B_KSTRIDED | AFR_GEMM_OUT_BF16 | (aux ? AFR_GEMM_RELU_MASK : 0);
    g[1].splitk = 1; g[1].slab_stride = 0;
    HIPCHK(afr_launch_gemm_group(AFR_BF16, g, 2, 1, s));
    return AFR_OK;
}
extern "C" int afr_op_reduce(float* dst, const float* slabs, int nslabs, int64_t stride, int64_t n, float scale, int acc,
                             void* stream) {
    if (!dst || !slabs || nslabs < 1) return fail(AFR_EINVAL, "bad reduce arguments");
    DevGuard dg(device_of(dst));
    HIPCHK(afr_launch_reduce(dst, slabs, nslabs, stride, n, scale, acc, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_reduce_group(int nseg, float* const* dst, const float* const* slabs, const int* nslabs, const int64_t* stride,
                                   const int64_t* n, void* stream) {
    if (nseg < 0 || (nseg > 0 && (!dst || !slabs || !nslabs || !stride || !n))) return fail(AFR_EINVAL, "bad grouped-reduce arguments");
    if (nseg > AFR_RT_MAXSEG) return fail(AFR_EINVAL, "a grouped reduce takes at most %d segments (got %d)", AFR_RT_MAXSEG, nseg);
    RTable rt;
    for (int i = 0; i < nseg; ++i) {
        if (!dst[i] || !slabs[i] || nslabs[i] < 1 || n[i] < 0 || (n[i] & 3)) return fail(AFR_EINVAL, "segment %d: null pointer, no slabs or length not a multiple of 4", i);
        afr_rtable_add(rt, dst[i], slabs[i], nslabs[i], stride[i], n[i]);
    }
    if (nseg == 0) return AFR_OK;
    DevGuard dg(device_of(dst[0]));
    HIPCHK(afr_launch_reduce_group(rt, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_adamw(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float b1,
                            float b2, float eps, float wd, int64_t t, float gscale, void* stream) {
    if (!p || !g || !m || !v || t < 1) return fail(AFR_EINVAL, "bad AdamW arguments");
    DevGuard dg(device_of(p));
    const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t));
    const float bc2 = (float)(1.0 - std::pow((double)b2, (double)t));
    HIPCHK(afr_launch_adamw(p, g, m, v, (bf16_t*)shadow, n, lr, b1, b2, eps, wd, bc1, bc2, gscale, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_mse_grad(int act_dtype, const void* u, const void* target, int tdtype, void* du, int64_t rows,
                               int64_t cols, int64_t mean_elems, float* loss_accum, float* scratch, void* stream) {
    if (!u || !target || !du || !loss_accum || !scratch) return fail(AFR_EINVAL, "null argument");
    DevGuard dg(device_of(du));
    HIPCHK(afr_launch_mse_grad(act_dtype, u, target, tdtype, du, rows, cols, mean_elems, loss_accum, scratch,
                               (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    DevGuard dg(device_of(dst));
    HIPCHK(afr_launch_f32_to_bf16(src, (bf16_t*)dst, n, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_f32_to_fp8(const float* src, void* dst, int64_t n, float scale, void* stream) {
    if (!src || !dst || n < 0 || !(scale > 0.f)) return fail(AFR_EINVAL, "bad f32 -> fp8 arguments");
    DevGuard dg(device_of(dst));
    HIPCHK(afr_launch_f32_to_fp8(src, (unsigned char*)dst, n, 1.f / scale, (hipStream_t)stream));
    return AFR_OK;
}
extern "C" int afr_op_gemm_fp8(int flags, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda, int ldb,
                               int ldc, float scale_ab, void* stream) {
    if (!A || !B || !C) return fail(AFR_EINVAL, "null operand");
    if (M <= 0 || N <= 0 || K <= 0) return fail(AFR_EINVAL, "bad GEMM extents");
    if (flags & ~(AFR_GEMM_BIAS | AFR_GEMM_RELU | AFR_GEMM_OUT_BF16)) return fail(AFR_EUNSUPPORTED, "fp8 products take bias / relu / bf16-output only (both operands k-contiguous)");
    if ((flags & AFR_GEMM_BIAS) && !bias) return fail(AFR_EINVAL, "bias flag without a bias");
    if (K % 16 || lda % 16 || ldb % 16 || N % 8 || ldc % 8) return fail(AFR_EUNSUPPORTED, "K, lda, ldb must be multiples of 16; N, ldc of 8");
    if ((long long)M * lda >= (1ll << 31) || (long long)N * ldb >= (1ll << 31)) return fail(AFR_EUNSUPPORTED, "an fp8 GEMM operand must be smaller than 2 GiB");
    DevGuard dg(device_of(C));
    GemmParams g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = 0;
    g.flags = flags; g.splitk = 1; g.slab_stride = 0; g.out_scale = scale_ab;
    HIPCHK(afr_launch_gemm_fp8(g, (hipStream_t)stream));
    return AFR_OK;
}
