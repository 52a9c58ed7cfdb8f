"""GPU: the kernels bench.py times, at the shapes bench.py times them, against the CPU oracle / fp64.

The bf16 launcher picks the 256x128 ring kernel only for grids that fill the chip (gemm.hip bf16_use_wide); the small
shapes of test_gpu_ops / test_gpu_models never reach it.  Everything here does: the full C3 batch (8192 glyphs, bf16,
fused loss, split-K weight gradients with fused bias gradients, grouped reduce, fused AdamW), C2 at its batch of 4096,
afr_op_gemm in all four operand orientations with every epilogue, and the bit-exact embedding gather of the north star."""
import itertools

import numpy as np
import pytest
import torch

from .util import GlyphConfig, engine_rounding, fused1_eligible, glyph_inputs, maxabs, oracle, rnd_du, synth, tparams

pytestmark = pytest.mark.gpu


def _engine(cfg, dtype="f32", max_batch=64, **kw):
    from ai_font_renderer_amd.engine import Engine
    eng = Engine(cfg, dtype=dtype, max_batch=max_batch, **kw)
    eng.load_params(synth.make_params(cfg))
    return eng


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return maxabs(got, ref) / max(1e-7, float(np.abs(ref).max()))


def _rand(tid, shape, bound=1.0):
    return torch.from_numpy(synth.hash_uniform(tid, shape, bound))


def _b16(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ----------------------------------------------------------------------------- whole steps at the benchmarked batch
def _oracle_step_with_engine_masks(cfg, dtype, eng, x, font, tu8):
    """Oracle gradients for the engine's own ReLU / clamp masks (read back through afr_debug_copy), plus the check that
    those masks differ from the oracle's own only where the oracle's pre-activation is within rounding of its threshold
    (gradients are discontinuous there; see test_gpu_models.test_glyph_c3_shape_bf16_vs_oracle_and_f32_engine)."""
    B = x.shape[0]
    xt, ft = torch.from_numpy(x), torch.from_numpy(font)
    rnd = engine_rounding(cfg, dtype)
    eng.forward(xt, ft if cfg.n_fonts else None, want_output=False)
    u_eng = eng.debug_read("u").view(B, -1).cpu()
    rmasks = [eng.debug_read("act", i + 1).view(B, -1).cpu() > 0 for i in range(len(cfg.hidden))]
    cmask = (u_eng >= 0) & (u_eng <= 1)
    P = tparams(cfg)
    _, own = oracle.glyph_forward(P, xt, ft, cfg, rnd=rnd)
    eps = 2e-5 if dtype == "f32" else 2e-2
    for i, m in enumerate(rmasks):
        bad = m != (own["pres"][i] > 0)
        assert bad.float().mean() < 1e-3 and (own["pres"][i][bad].abs() < eps).all(), (dtype, i)
    badc = cmask != ((own["u"] >= 0) & (own["u"] <= 1))
    assert badc.float().mean() < 1e-3 and (torch.minimum(own["u"][badc].abs(), (own["u"][badc] - 1).abs()) < eps).all()
    _, cache = oracle.glyph_forward(P, xt, ft, cfg, rnd=rnd, relu_masks=rmasks)
    assert maxabs(u_eng.numpy(), cache["u"].numpy()) < (2e-5 if dtype == "f32" else 3e-2)
    tgt = torch.from_numpy(tu8.astype(np.float32) / 255.0)
    lref, du = oracle.mse_loss_grad(cache["u"], tgt, clamp_mask=cmask)
    return P, float(lref), oracle.glyph_backward(P, cache, rnd_du(rnd, du), cfg, rnd=rnd), u_eng


@pytest.mark.parametrize("workload,dtype,tol", [("c3", "bf16", 3e-2), ("c2", "bf16", 3e-2), ("c3", "f32", 1e-4), ("c2", "f32", 1e-4),
                                                ("c1", "f32", 1e-4), ("c1", "bf16", 3e-2)])
def test_benchmarked_batch_full_step_vs_oracle(workload, dtype, tol):
    """BASELINE configs[2] (C3: 8192 glyphs, 1024-wide, font ids) and configs[1] (C2: 4096 glyphs, hidden 256) exactly as
    bench.py runs them: afr_train_step with the loss fused into the last GEMM's epilogue, split-K weight-gradient GEMMs
    with fused bias gradients, the grouped slab reduction -- every gradient against the oracle; then the fused
    optimizer step against the oracle's AdamW on those gradients."""
    from ai_font_renderer_amd.config import WORKLOADS
    cfg, B = WORKLOADS[workload]["cfg"], WORKLOADS[workload]["batch"]
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font) if cfg.n_fonts else None, torch.from_numpy(tu8)
    eng = _engine(cfg, dtype=dtype, max_batch=B)
    if fused1_eligible(cfg):
        # afr_train_step runs this net as ONE fused kernel: no activation ever leaves LDS, so there are no engine masks to
        # read back.  Both sides round at the same points (util.engine_rounding), so their masks differ only where an
        # accumulation-order difference crosses a threshold: rare enough for the tolerance.
        rnd = engine_rounding(cfg, dtype, train_step=True)
        P = tparams(cfg)
        _, cache = oracle.glyph_forward(P, xt, torch.from_numpy(font), cfg, rnd=rnd)
        lref, du = oracle.mse_loss_grad(cache["u"], torch.from_numpy(tu8.astype(np.float32) / 255.0))
        lref, Gref = float(lref), oracle.glyph_backward(P, cache, rnd_du(rnd, du), cfg, rnd=rnd)
    else:
        P, lref, Gref, _ = _oracle_step_with_engine_masks(cfg, dtype, eng, x, font, tu8)
    eng.read_loss()
    eng.train_step(xt, tt, font=ft, do_step=False)              # the bench's kernels, gradients materialised
    assert abs(eng.read_loss() - lref) < tol * lref
    for k, g in eng.grads.items():
        assert _rel(g.cpu().numpy(), Gref[k].numpy()) < tol, (workload, dtype, k)
    # the step bench.py times (AdamW inside the slab reduction): t = 1, so every update is -lr * g / (|g| + eps) - decay.
    # Compared with the oracle's AdamW on the ORACLE's gradients; entries whose gradient is ~0 may move the other way.
    g_eng = {k: v.clone() for k, v in eng.grads.items()}
    eng.train_step(xt, tt, font=ft, do_step=True)
    eng.read_loss()
    for k, v in eng.state_dict().items():
        p1, _, _ = oracle.adamw_step(P[k], Gref[k], torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1)
        diff = (v.cpu() - p1).abs()
        small = Gref[k].abs() < tol * Gref[k].abs().max()       # sign of the update undetermined within the tolerance
        assert float(diff[~small].max() if (~small).any() else 0.0) < 2e-4, (workload, dtype, k)
        assert float(diff.max()) <= 2.1e-3, k                   # never more than one full +-lr flip
    # and, to rounding, what the stand-alone AdamW kernel makes of the ENGINE's own gradients
    eng2 = _engine(cfg, dtype=dtype, max_batch=B)
    for k in g_eng:
        eng2.grads[k].copy_(g_eng[k])
    eng2.adamw_step()
    for k, v in eng.state_dict().items():
        d = (v - eng2.params[k]).abs()
        assert float(d.max()) <= 3e-6 * max(1.0, float(eng2.params[k].abs().max())), (workload, dtype, k)


def test_c3_fused_loss_epilogue_equals_unfused_loss_kernel():
    """The last forward GEMM of a C3 training step computes clamp / MSE / du in its epilogue (wide kernel, 256 blocks with
    the ticketed loss reduction).  Same inputs through afr_forward + afr_loss_grad: du bit for bit, loss to f32 rounding;
    and both against the oracle's mse_loss_grad on the engine's own u."""
    from ai_font_renderer_amd.config import WORKLOADS
    cfg, B = WORKLOADS["c3"]["cfg"], 8192
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font), torch.from_numpy(tu8)
    eng = _engine(cfg, dtype="bf16", max_batch=B)
    eng.forward(xt, ft, training=True, want_output=False)
    u = eng.debug_read("u").view(B, -1).cpu()
    eng.loss_grad(tt)
    du_unfused, l_unfused = eng.debug_read("u").view(B, -1).cpu(), eng.read_loss()
    eng.forward_loss(xt, tt, font=ft)
    du_fused, l_fused = eng.debug_read("u").view(B, -1).cpu(), eng.read_loss()
    assert torch.equal(du_fused, du_unfused)
    assert abs(l_fused - l_unfused) <= 2e-6 * l_unfused
    lref, du_ref = oracle.mse_loss_grad(u.double(), torch.from_numpy(tu8.astype(np.float64) / 255.0))
    assert abs(l_fused - float(lref)) < 1e-5 * float(lref)
    assert float((du_fused.double() - du_ref).abs().max()) <= 1e-2 * float(du_ref.abs().max())      # du is stored as bf16
    assert float(du_fused[(u < 0) | (u > 1)].abs().max()) == 0.0


# ----------------------------------------------------------------------------- the wide GEMM kernel through afr_op_gemm
WIDE_SHAPES = [(8192, 1024, 1024), (4000, 2040, 520)]       # grids of 256 tiles of 256x128; the second ragged in M, N and K


@pytest.mark.parametrize("ak,bk", list(itertools.product([False, True], repeat=2)))
def test_wide_bf16_gemm_all_orientations_vs_fp64(ak, bk):
    from .gpu_util import gemm
    for si, (M, N, K) in enumerate(WIDE_SHAPES):
        A, B = _b16(_rand(110 + si, (M, K))), _b16(_rand(120 + si, (N, K), 0.25))
        ref = A.double() @ B.double().t()
        scale = float(ref.abs().max())
        got = gemm("bf16", A, B, ak, bk)
        assert float((got.double() - ref).abs().max()) < 1e-5 * scale, (ak, bk, M, N, K)     # f32 accumulation of exact products
        got = gemm("bf16", A, B, ak, bk, out_bf16=True)
        assert float((got.double() - ref).abs().max()) < 1e-2 * scale, (ak, bk, M, N, K)


def test_wide_bf16_gemm_epilogues_and_splitk():
    from .gpu_util import gemm
    M, N, K = 8192, 1024, 1024
    A, B = _b16(_rand(131, (M, K))), _b16(_rand(132, (N, K), 0.2))
    bias, aux = _rand(133, (N,)), _b16(_rand(134, (M, N)))
    ref = A.double() @ B.double().t()
    scale = float(ref.abs().max())
    got = gemm("bf16", A, B, bias=bias, relu=True, out_bf16=True)                               # forward layer
    assert float((got.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-2 * scale
    got = gemm("bf16", A, B, bias=bias, relu=True)
    assert float((got.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-5 * scale
    got = gemm("bf16", A, B, b_kstrided=True, aux=aux, out_bf16=True)                           # input gradient with ReLU mask
    assert float((got.double() - ref * (aux > 0)).abs().max()) < 1e-2 * scale
    got = gemm("bf16", A, B, b_kstrided=True, aux=aux)
    assert float((got.double() - ref * (aux > 0)).abs().max()) < 1e-5 * scale
    # weight gradient: 1024 x 1024 output reduced over 8192, split-K 8 (256 blocks) and 3 (ragged split)
    A2, B2 = _b16(_rand(141, (1024, 8192), 0.05)), _b16(_rand(142, (1024, 8192)))
    ref2 = A2.double() @ B2.double().t()
    for sk in (8, 3):
        got = gemm("bf16", A2, B2, a_kstrided=True, b_kstrided=True, splitk=sk)
        assert float((got.double() - ref2).abs().max()) < 2e-5 * float(ref2.abs().max()), sk


@pytest.mark.parametrize("M,N,K,head,sk,ak,bk", [
    (1024, 19200, 6400, 256, 5, False, False),     # R0 fc_output forward: one full round whole + 44 tail tiles cut 5 ways
    (1024, 6400, 19200, 0, 2, False, True),        # R0 fc_output input gradient: 100 tiles cut 2 ways
    (520, 776, 2120, 1, 3, False, False),          # ragged tiles, K not a multiple of the slice length, one head tile
    (520, 776, 2120, 0, 4, True, True),            # both operands k-strided
    (304, 520, 4096, 6, 2, True, False),           # head_tiles == tiles: nothing is split
])
def test_in_launch_splitk_gemm(M, N, K, head, sk, ak, bk):
    """afr_op_gemm_fix (the form afr_train_step uses for the sheet model's fc_output products): every epilogue against
    fp64, the arrival counters left at zero, and bitwise equal results from launch to launch (the slices are added in
    slice order whichever workgroup arrives last)."""
    from .gpu_util import gemm_fix
    A, B = _b16(_rand(151, (M, K))), _b16(_rand(152, (N, K), 0.2))
    bias, aux = _rand(153, (N,)), _b16(_rand(154, (M, N)))
    ref = A.double() @ B.double().t()
    scale = float(ref.abs().max())
    outs, cnt = gemm_fix(A, B, head, sk, a_kstrided=ak, b_kstrided=bk, repeats=3)
    assert float((outs[0].double() - ref).abs().max()) < 2e-5 * scale
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert int(cnt.abs().max()) == 0
    outs, _ = gemm_fix(A, B, head, sk, a_kstrided=ak, b_kstrided=bk, bias=bias, relu=True, out_bf16=True)
    assert float((outs[0].double() - torch.relu(ref + bias.double())).abs().max()) < 1e-2 * scale
    outs, _ = gemm_fix(A, B, head, sk, a_kstrided=ak, b_kstrided=bk, aux=aux, out_bf16=True)
    assert float((outs[0].double() - ref * (aux > 0)).abs().max()) < 1e-2 * scale


@pytest.mark.parametrize("B,N,K,sk_want", [
    (8192, 1024, 1024, 8),        # C3's hidden layers: 128 input-gradient tiles + 16 weight-gradient tiles x 8 slices
    (8192, 2048, 1024, 4),        # 32 x 4 slices
    (8192, 4096, 1024, 2),        # 64 x 2 slices
    (8000, 1000, 1016, 8),        # ragged in every extent; the last K-slice is short
])
def test_grouped_gradient_pair_vs_fp64(B, N, K, sk_want):
    """afr_op_gemm_pair = the launch that takes 45 % of a C3 step (gemm_bf16_group256: a layer's dW with cooperative
    split-K + dX with the ReLU mask, one grid): dW and db against fp64 at f32-accumulation accuracy, dX at bf16 output
    accuracy, bitwise equal from launch to launch, and dX bitwise equal to the stand-alone afr_op_gemm product."""
    from .gpu_util import gemm, gemm_pair
    dy, x = _b16(_rand(161, (B, N), 0.05)), _b16(_rand(162, (B, K)))
    W, aux = _b16(_rand(163, (N, K), 0.2)), _b16(_rand(164, (B, K)))
    outs, sk = gemm_pair(dy, x, W, aux, repeats=3)
    assert sk == sk_want
    dW, db, dX = outs[0]
    ref_w = dy.double().t() @ x.double()
    assert float((dW.double() - ref_w).abs().max()) < 1e-5 * float(ref_w.abs().max())
    ref_b = dy.double().sum(0)
    assert float((db.double() - ref_b).abs().max()) < 1e-5 * float(ref_b.abs().max())
    ref_x = (dy.double() @ W.double()) * (aux > 0)
    assert float((dX.double() - ref_x).abs().max()) < 1e-2 * float(ref_x.abs().max())
    for o in outs[1:]:
        assert torch.equal(o[0], dW) and torch.equal(o[1], db) and torch.equal(o[2], dX)
    solo = gemm("bf16", dy, W.t().contiguous(), b_kstrided=True, aux=aux, out_bf16=True)       # B(n,k) = W[k][n]
    assert float((solo - dX).abs().max()) <= 2e-2 * float(ref_x.abs().max())                    # at most one bf16 ulp apart
    dXn = gemm_pair(dy, x, W, None)[0][0][2]
    assert float((dXn.double() - dy.double() @ W.double()).abs().max()) < 1e-2 * float(ref_x.abs().max())


def test_c3_cooperative_splitk_equals_slabs_and_separate_launches():
    """C3 at its batch, bf16: the step's grouped backward launches with the cooperative split-K tail (default), with
    split-K slabs + grouped reduce (config.reserved bit 5) and as one launch per product (bit 1).  Gradients: default ==
    slabs bit for bit (same slices, same order); against separate launches to f32 accumulation-order rounding.  Then three
    optimizer steps: AdamW inside the weight-gradient GEMM == AdamW inside the slab reduction, bit for bit."""
    from ai_font_renderer_amd.config import WORKLOADS
    cfg, B = WORKLOADS["c3"]["cfg"], 8192
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font), torch.from_numpy(tu8)
    engs = {fl: _engine(cfg, dtype="bf16", max_batch=B, flags=fl) for fl in (0, 32, 2)}
    for e in engs.values():
        e.train_step(xt, tt, font=ft, do_step=False)
        e.read_loss()
    for k in engs[0].grads:
        assert torch.equal(engs[0].grads[k], engs[32].grads[k]), k
        ref = engs[2].grads[k]
        assert float((engs[0].grads[k] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), k
    for _ in range(3):
        for fl in (0, 32):
            engs[fl].train_step(xt, tt, font=ft)
    assert abs(engs[0].read_loss() - engs[32].read_loss()) == 0.0
    assert torch.equal(engs[0].flat_params, engs[32].flat_params)
    assert torch.equal(engs[0].exp_avg, engs[32].exp_avg) and torch.equal(engs[0].exp_avg_sq, engs[32].exp_avg_sq)
    for e in engs.values():
        assert e.error_flags() == 0
    # the bf16 copies the next forward reads are those of the updated masters
    for fl in (0, 32):
        y = engs[fl].forward(xt[:512], ft[:512])
        e2 = _engine(cfg, dtype="bf16", max_batch=512)
        e2.load_params({k: v.cpu() for k, v in engs[fl].state_dict().items()})
        assert torch.equal(y, e2.forward(xt[:512], ft[:512])), fl


@pytest.mark.parametrize("B", [8192, 8008])
def test_c3_combination_table_first_layer_equals_materialised_h1(B):
    """A C3 training step does not materialise the first layer's output per glyph: h1 / h0 live as a (character, font)
    combination table and the second layer's forward, its weight gradient, its ReLU mask and the first layer's backward
    gather table rows while staging (default).  config.reserved bit 6 = the gather kernel + dense h1 [B][1024].  Same values,
    same products, same summation order: gradients and three optimizer steps agree bit for bit -- at the bench's batch
    (every workgroup on the fast addressing path) and at a ragged one (last row block and last K-slice on the slow path)."""
    from ai_font_renderer_amd.config import WORKLOADS
    cfg = WORKLOADS["c3"]["cfg"]
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font), torch.from_numpy(tu8)
    a = _engine(cfg, dtype="bf16", max_batch=B)
    a.train_step(xt, tt, font=ft, do_step=False)
    la, ga = a.read_loss(), {k: v.clone() for k, v in a.grads.items()}
    for _ in range(3):
        a.train_step(xt, tt, font=ft)
    la3 = a.read_loss()
    # bit 6: dense h1; bit 7: ReLU masks read from the stored activations instead of the forward epilogues' bit masks
    for fl in (64, 128, 64 | 128):
        b = _engine(cfg, dtype="bf16", max_batch=B, flags=fl)
        b.train_step(xt, tt, font=ft, do_step=False)
        assert la == b.read_loss(), fl
        for k in ga:
            assert torch.equal(ga[k], b.grads[k]), (fl, k)
        for _ in range(3):
            b.train_step(xt, tt, font=ft)
        assert la3 == b.read_loss(), fl
        assert torch.equal(a.flat_params, b.flat_params), fl
        assert b.error_flags() == 0
        del b
    assert a.error_flags() == 0
    # an out-of-range code is flagged by the combination path as by the gather kernel (the reference raises IndexError)
    xbad = xt.clone()
    xbad[17] = 128
    a.train_step(xbad, tt, font=ft, do_step=False)
    assert a.error_flags() & 1


@pytest.mark.parametrize("cfgkw,B,flags", [
    (dict(hidden=(1024, 1024), out_h=32, out_w=32, n_fonts=2), 8192, 0),     # C3: 128 row blocks x 2 column ranges
    (dict(hidden=(1024, 512), out_h=16, out_w=16, n_fonts=2), 1000, 0),      # ragged last block, 4 column ranges
    (dict(hidden=(256,), out_h=16, out_w=16, n_fonts=0), 4096, 4),           # C2's net through the per-layer kernels, no fonts
])
def test_fused_first_layer_backward_matches_the_gemm_path(cfgkw, B, flags):
    """glyph_l1_bwd_fused_kernel (bf16 mode) against the path it replaces (config.reserved bit 4: widened weight-gradient
    GEMM + post-pass): dW1 / db1 take the same bf16 operands (f32 sums in another order); the embedding-row gradients
    differ by the bf16 rounding of W1 and dh0 on the fused side.  Bitwise reproducible."""
    cfg = GlyphConfig(**cfgkw)
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, tt = torch.from_numpy(x), torch.from_numpy(tu8)
    ft = torch.from_numpy(font) if cfg.n_fonts else None
    got = {}
    for fl in (flags, flags | 16):
        eng = _engine(cfg, dtype="bf16", max_batch=B, flags=fl)
        eng.train_step(xt, tt, font=ft, do_step=False)
        g = {k: v.clone() for k, v in eng.grads.items()}
        eng.train_step(xt, tt, font=ft, do_step=False)
        assert all(torch.equal(g[k], v) for k, v in eng.grads.items()), fl
        got[fl] = g
    for k, ref in got[flags | 16].items():
        rel = float((got[flags][k] - ref).norm() / ref.norm().clamp_min(1e-20))
        assert rel < (2e-2 if "embedding" in k else 2e-3), (k, rel)


# ----------------------------------------------------------------------------- bit-exact gather (north star)
@pytest.mark.parametrize("cfgkw,B,xmax", [
    (dict(hidden=(1024, 1024), out_h=32, out_w=32, n_fonts=2), 8192, None),       # C3: folded first layer (table gather)
    (dict(hidden=(256,), out_h=16, out_w=16, n_fonts=0), 4096, None),            # C1/C2 net
    (dict(hidden=(32,), out_h=4, out_w=4, n_fonts=2, vocab=600), 700, 600),      # too wide to fold: glyph_embed_kernel
    (dict(hidden=(), out_h=4, out_w=6, n_fonts=2), 64, None),                    # no hidden layer: glyph_embed_kernel
])
def test_embedding_gather_is_bit_exact(cfgkw, B, xmax):
    """nn.Embedding lookup (model.py:136,167): h0[b] == Emb[x_b] (+ Font[f_b]) bit for bit in f32 mode; the folded path's
    one-hot columns are exactly {0,1} at exactly the two table rows a glyph uses."""
    cfg = GlyphConfig(**cfgkw)
    x, font, _ = glyph_inputs(cfg, B)
    if xmax is not None:
        x = (np.arange(B, dtype=np.int64) * 7) % xmax
    eng = _engine(cfg, dtype="f32", max_batch=B)
    eng.forward(torch.from_numpy(x), torch.from_numpy(font) if cfg.n_fonts else None, want_output=False)
    E = cfg.embed_dim
    h0 = eng.debug_read("act", 0).view(B, -1).cpu()
    P = tparams(cfg)
    want = P["embedding.weight"][torch.from_numpy(x)]
    if cfg.n_fonts:
        want = want + P["font_embedding.weight"][torch.from_numpy(font)]
    assert torch.equal(h0[:, :E], want)
    if h0.shape[1] > E:                                   # folded: [h0 | one-hot(x) | one-hot(vocab + f) | zero pad]
        oh = torch.zeros(B, h0.shape[1] - E)
        oh[torch.arange(B), torch.from_numpy(x)] = 1.0
        if cfg.n_fonts:
            oh[torch.arange(B), cfg.vocab + torch.from_numpy(font)] = 1.0
        assert torch.equal(h0[:, E:], oh)
    assert eng.error_flags() == 0
    # bf16 mode stores the same sum rounded once to bf16
    eng16 = _engine(cfg, dtype="bf16", max_batch=B)
    eng16.forward(torch.from_numpy(x), torch.from_numpy(font) if cfg.n_fonts else None, want_output=False)
    h16 = eng16.debug_read("act", 0).view(B, -1).cpu()
    assert torch.equal(h16[:, :E], want.to(torch.bfloat16).float())


def test_golden_glyph_fixtures_through_the_engine():
    """The reference-pinned glyph fixtures (tests/golden/glyph_ref1.npz: the reference class at max_length=1;
    glyph_twin.npz: torch.nn twin, 3 AdamW steps) replayed through the C ABI in f32 mode."""
    from .util import load
    from ai_font_renderer_amd.config import WORKLOADS
    from ai_font_renderer_amd.engine import Engine
    fx = load("glyph_ref1.npz")
    h, w = fx["y"].shape[1:]
    cfg = GlyphConfig(hidden=(64,), out_h=h, out_w=w, embed_dim=32, vocab=128, n_fonts=0)
    eng = Engine(cfg, dtype="f32", max_batch=256)
    P = {"embedding.weight": fx["table"]}
    for k in ("fc1.weight", "fc1.bias", "fc_output.weight", "fc_output.bias"):
        P[k] = fx["param/" + k]
    eng.load_params(P)
    x, t = torch.from_numpy(fx["x"]), torch.from_numpy(fx["target_u8"])
    assert maxabs(eng.forward(x).cpu().numpy(), fx["y"]) < 2e-5
    eng.train_step(x, t)
    assert abs(eng.read_loss() - float(fx["loss"])) < 3e-6
    for k in ("fc1.weight", "fc1.bias", "fc_output.weight", "fc_output.bias"):
        assert maxabs(eng.params[k].cpu().numpy(), fx["step1/" + k]) < 2e-5, k
    tw = load("glyph_twin.npz")
    for tag, cfg, B in (("small", GlyphConfig(hidden=(48, 40), out_h=4, out_w=6, n_fonts=2), 300), ("c1", WORKLOADS["c1"]["cfg"], 95)):
        x, font, tu8 = glyph_inputs(cfg, B)
        xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font) if cfg.n_fonts else None, torch.from_numpy(tu8)
        eng = _engine(cfg, max_batch=B)
        assert maxabs(eng.forward(xt, ft).cpu().numpy(), tw[f"{tag}/eval_y"]) < 2e-5
        eng.train_step(xt, tt, font=ft, do_step=False)
        eng.read_loss()
        for k, g in eng.grads.items():
            if f"{tag}/grad/{k}" in tw:
                assert _rel(g.cpu().numpy(), tw[f"{tag}/grad/{k}"]) < 1e-4, (tag, k)
        for i in range(3):
            eng.train_step(xt, tt, font=ft)
            assert abs(eng.read_loss() - float(tw[f"{tag}/losses"][i])) < 3e-6, (tag, i)
        for k, v in eng.state_dict().items():
            assert maxabs(v.cpu().numpy(), tw[f"{tag}/param3/{k}"]) < 2e-5, (tag, k)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfgkw,B", [
    (dict(hidden=(256,), out_h=16, out_w=16), 95),                          # C1: one ragged 16-/64-row block at the end
    (dict(hidden=(64,), out_h=8, out_w=8, n_fonts=2), 300),                 # smallest shapes the fused kernel takes, with fonts
    (dict(hidden=(192,), out_h=8, out_w=16, n_fonts=1, embed_dim=64), 130),  # wider embedding, widths that are not powers of 2
])
def test_fused_small_net_step_equals_the_per_layer_kernels(cfgkw, B, dtype):
    """afr_train_step on a small one-hidden-layer glyph net is ONE fused kernel + the grouped reduce (glyph_fused.hip).
    f32: the same step through the generic per-layer kernels (config flag bit 2) -- gradients, loss and three optimizer
    steps agree to accumulation-order rounding.  bf16: the two paths round at different points (the fused kernel takes
    bf16 operands in every product, the generic path evaluates fc1 from f32 tables), so each is held to the oracle rounded
    at ITS points instead."""
    cfg = GlyphConfig(**cfgkw)
    assert fused1_eligible(cfg)
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, ft, tt = torch.from_numpy(x), torch.from_numpy(font) if cfg.n_fonts else None, torch.from_numpy(tu8)
    a = _engine(cfg, dtype=dtype, max_batch=B)
    a.train_step(xt, tt, font=ft, do_step=False)
    la = a.read_loss()
    if dtype == "f32":
        b = _engine(cfg, dtype=dtype, max_batch=B, flags=4)
        b.train_step(xt, tt, font=ft, do_step=False)
        lb = b.read_loss()
        assert abs(la - lb) <= 2e-6 * lb
        for k in a.grads:
            assert _rel(a.grads[k].cpu().numpy(), b.grads[k].cpu().numpy()) < 2e-5, k
        for _ in range(3):
            a.train_step(xt, tt, font=ft)
            b.train_step(xt, tt, font=ft)
        assert abs(a.read_loss() - b.read_loss()) < 1e-5 * 3
        for k in a.params:
            assert float((a.params[k] - b.params[k]).abs().max()) < 2e-5, k
    else:
        rnd = engine_rounding(cfg, dtype, train_step=True)
        P = tparams(cfg)
        _, cache = oracle.glyph_forward(P, xt, torch.from_numpy(font), cfg, rnd=rnd)
        lref, du = oracle.mse_loss_grad(cache["u"], torch.from_numpy(tu8.astype(np.float32) / 255.0))
        Gref = oracle.glyph_backward(P, cache, rnd_du(rnd, du), cfg, rnd=rnd)
        assert abs(la - float(lref)) < 3e-2 * float(lref)
        for k in a.grads:
            assert _rel(a.grads[k].cpu().numpy(), Gref[k].numpy()) < 3e-2, k
    assert a.error_flags() == 0
    # run-to-run bitwise reproducibility of the fused step (slabs are summed in block order)
    a.train_step(xt, tt, font=ft, do_step=False)
    g1 = a.flat_grads.clone()
    a.train_step(xt, tt, font=ft, do_step=False)
    assert torch.equal(g1, a.flat_grads)


@pytest.mark.parametrize("workload", ["c2", "c1"])
def test_fused_small_net_bf16_transposed_copies_stay_current(workload):
    """bf16 fused small-net step (C1 / C2, the benchmarked path): the transposed operand copies W1^T / W2^T that the NEXT
    step's dh1 = du . W2^T and dh0 = dpre1 . W1^T read are written by the grouped reduce's AdamW epilogue (shT scatter) and
    cached behind wT_valid.  Three fused steps against three steps with the optimizer un-fused (config.reserved bit 0:
    wT_valid = false, a fresh transpose kernel every step): parameters, moments and the last step's gradients bit for bit;
    and the copies read back through afr_debug_copy equal transpose(bf16(P))."""
    from ai_font_renderer_amd.config import WORKLOADS
    cfg, B = WORKLOADS[workload]["cfg"], WORKLOADS[workload]["batch"]
    x, font, tu8 = glyph_inputs(cfg, B)
    xt, tt = torch.from_numpy(x), torch.from_numpy(tu8)
    a, b = _engine(cfg, dtype="bf16", max_batch=B), _engine(cfg, dtype="bf16", max_batch=B, flags=1)
    for _ in range(3):
        a.train_step(xt, tt)
        b.train_step(xt, tt)
    assert a.read_loss() == b.read_loss()
    assert torch.equal(a.flat_params, b.flat_params)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    N1, P = cfg.hidden[0], cfg.pixels
    for eng in (a, b):
        w1t = eng.debug_read("w1t").view(cfg.embed_dim, N1).cpu()
        w2t = eng.debug_read("w2t").view(N1, P).cpu()
        if eng is b:                               # un-fused: the copies are those the LAST step's transpose made, one AdamW behind
            continue
        assert torch.equal(w1t, eng.params["fc1.weight"].cpu().to(torch.bfloat16).float().t())
        assert torch.equal(w2t, eng.params["fc_output.weight"].cpu().to(torch.bfloat16).float().t())
    a.train_step(xt, tt, do_step=False)             # a fourth step's gradients read the copies the third step's reduce wrote
    b.train_step(xt, tt, do_step=False)
    assert torch.equal(a.flat_grads, b.flat_grads)
