"""GPU: edge cases and full-size properties of the hot path (inputs the reference accepts but never tests)."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from .util import MINI, R0, engine_rounding, maxabs, oracle, rnd_du, synth, tmasks, tparams

pytestmark = pytest.mark.gpu


def _engine(cfg, dtype="f32", max_batch=64, **kw):
    from ai_font_renderer_amd.engine import Engine
    eng = Engine(cfg, dtype=dtype, max_batch=max_batch, **kw)
    eng.load_params(synth.make_params(cfg))
    return eng


def _oracle_step(cfg, x, tgt, masks=None):
    P = tparams(cfg)
    y, cache = oracle.sheet_forward(P, torch.from_numpy(x), cfg, masks)
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    return y, float(loss), oracle.sheet_backward(P, cache, du, cfg)


@pytest.mark.parametrize("B,L", [(1, 10), (3, 1), (2, 10)])
def test_single_sample_single_char_and_all_padding(B, L):
    """B=1 is how the reference renders (helpers.py:61-64); L=1 leaves 9/10 of the features zero-padded; an all-zero
    row is the all-padding string (pad code 0 is an ordinary token)."""
    x = np.zeros((B, L), dtype=np.int64)
    x[0, :] = [65 + (i % 26) for i in range(L)]
    if B > 1:
        x[1, :] = 127                                            # highest legal code
    eng = _engine(MINI, seed=3)
    tu8 = synth.synth_sheet_targets(B, 8, 24, tensor_id=970)
    tgt = torch.from_numpy(tu8.astype(np.float32) / 255.0)
    y = eng.forward(torch.from_numpy(x)).cpu().numpy()
    yref, _, _ = _oracle_step(MINI, x, tgt)
    assert maxabs(y, yref.numpy()) < 2e-5
    assert eng.error_flags() == 0
    eng.train_step(torch.from_numpy(x), torch.from_numpy(tu8), step=2, do_step=False)
    masks = tmasks(synth.sheet_dropout_masks(MINI, B, min(L, 10), seed=3, step=2))
    _, lref, Gref = _oracle_step(MINI, x, tgt, masks)
    assert abs(eng.read_loss() - lref) < 1e-5 * max(lref, 1e-3)
    for k, g in eng.grads.items():
        ref = Gref[k].numpy()
        assert maxabs(g.cpu().numpy(), ref) <= 1e-4 * max(1e-7, float(np.abs(ref).max())), k


def test_longest_supported_sequence_and_its_limit():
    """max_length 120 is the largest string whose backward state fits one CU's 160 KiB of LDS; 121 is refused."""
    from ai_font_renderer_amd import _lib
    from ai_font_renderer_amd.engine import Engine
    from .util import SheetConfig
    cfg = SheetConfig(max_length=120, sheet_h=8, sheet_w=16)
    x = synth.encode_strings([s * 2 for s in synth.dataset_strings(5)], 120)
    tu8 = synth.synth_sheet_targets(5, 8, 16, tensor_id=971)
    eng = _engine(cfg, seed=9)
    eng.train_step(torch.from_numpy(x), torch.from_numpy(tu8), step=1, do_step=False)
    masks = tmasks(synth.sheet_dropout_masks(cfg, 5, 120, seed=9, step=1))
    _, lref, Gref = _oracle_step(cfg, x, torch.from_numpy(tu8.astype(np.float32) / 255.0), masks)
    assert abs(eng.read_loss() - lref) < 1e-5 * lref
    for k, g in eng.grads.items():
        ref = Gref[k].numpy()
        assert maxabs(g.cpu().numpy(), ref) <= 1e-4 * float(np.abs(ref).max()), k
    with pytest.raises(_lib.AfrError):
        Engine(SheetConfig(max_length=121, sheet_h=8, sheet_w=16), max_batch=4)


def test_uint8_and_float32_targets_give_identical_results():
    """helpers.load_string_dataset hands float32 k/255 targets (helpers.py:121); the loop keeps uint8 in HBM."""
    cfg = replace(MINI, p_embed=0.0, p_attn=0.0, p_fc=0.0)
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(6), 10))
    tu8 = torch.from_numpy(synth.synth_sheet_targets(6, 8, 24, tensor_id=972))
    tf = tu8.to(torch.float32) / 255.0
    eng = _engine(cfg)
    out = []
    for t in (tu8, tf):
        for fused in (True, False):
            if fused:
                eng.train_step(x, t, do_step=False)
            else:
                eng.forward(x, training=True, want_output=False)
                eng.loss_grad(t)
                eng.backward()
            out.append((eng.read_loss(), eng.flat_grads.clone()))
    for l, g in out[1:]:                       # the loss is summed in a different (still fixed) order when it is fused
        assert abs(l - out[0][0]) <= 2e-7 * out[0][0] and torch.equal(g, out[0][1])


def test_eval_forward_is_deterministic_and_ignores_the_dropout_stream():
    eng = _engine(MINI)
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(7), 10))
    a = eng.forward(x, training=False, step=0).clone()
    b = eng.forward(x, training=False, step=123).clone()
    c = eng.forward(x, training=True, step=123).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0


def test_reference_batch_r0_full_size_linearity_and_tail_batches():
    """The shipped model at the reference's GPU batch size (1024 sheets, model.py:409) in f32: the batch gradient equals
    the sum of the gradients of a 832 + 192 split (192 is the reference's last training batch) when both
    halves use the global mean denominator; and the step is bitwise reproducible."""
    cfg = replace(R0, p_embed=0.0, p_attn=0.0, p_fc=0.0)
    B = 1024
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(B), 100)).cuda()
    t = torch.from_numpy(synth.synth_sheet_targets(B, 80, 240, tensor_id=973)).cuda()
    eng = _engine(cfg, max_batch=B, with_optimizer=False)
    eng.train_step(x, t, do_step=False)
    full, lfull = eng.flat_grads.clone(), eng.read_loss()
    eng.train_step(x, t, do_step=False)
    assert torch.equal(eng.flat_grads, full) and eng.read_loss() == lfull
    acc, lsum = torch.zeros_like(full), 0.0
    for sl in (slice(0, 832), slice(832, B)):
        eng.train_step(x[sl], t[sl], do_step=False, mean_elems=B * cfg.pixels)
        acc += eng.flat_grads
        lsum += eng.read_loss()
    assert abs(lsum - lfull) < 1e-5 * lfull
    assert float((acc - full).abs().max()) < 1e-4 * float(full.abs().max())


def test_reference_batch_r0_full_size_bf16_with_dropout_tracks_the_f32_engine():
    """Throughput mode at the reference's full batch (1024 sheets, all three dropouts active): bitwise reproducible, and
    loss / gradients stay within bf16 rounding of the exact-f32 engine run on the same counter-hash masks."""
    B = 1024
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(B), 100)).cuda()
    t = torch.from_numpy(synth.synth_sheet_targets(B, 80, 240, tensor_id=974)).cuda()
    res = {}
    for dtype in ("f32", "bf16"):
        eng = _engine(R0, dtype=dtype, max_batch=B, with_optimizer=False)
        eng.train_step(x, t, step=7, do_step=False)
        g, l = eng.flat_grads.clone(), eng.read_loss()
        eng.train_step(x, t, step=7, do_step=False)
        assert torch.equal(eng.flat_grads, g) and eng.read_loss() == l, dtype
        res[dtype] = (l, {k: v.clone() for k, v in eng.grads.items()})
        del eng
        torch.cuda.empty_cache()
    assert abs(res["bf16"][0] - res["f32"][0]) < 2e-3 * res["f32"][0]
    for k, ref in res["f32"][1].items():
        got = res["bf16"][1][k]
        rel = float((got - ref).norm() / ref.norm().clamp_min(1e-12))
        assert rel < 5e-2, (k, rel)


def test_r0_input_gradient_as_in_launch_splitk_matches_the_default_kernels():
    """config.reserved bit 3: fc_output's input-gradient product on 256x256 tiles with the split-K sum inside the launch.
    Same bf16 operands and f32 accumulation, another summation order: gradients equal the default path's to f32 rounding
    (the front end's gradients come through a bf16 dz, so to a bf16 ulp there), and the step stays bitwise reproducible."""
    B = 512
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(B), 100)).cuda()
    t = torch.from_numpy(synth.synth_sheet_targets(B, 80, 240, tensor_id=975)).cuda()
    res = {}
    for flags in (0, 8):
        eng = _engine(R0, dtype="bf16", max_batch=B, with_optimizer=False, flags=flags)
        eng.train_step(x, t, step=3, do_step=False)
        g, l = eng.flat_grads.clone(), eng.read_loss()
        eng.train_step(x, t, step=3, do_step=False)
        assert torch.equal(eng.flat_grads, g) and eng.read_loss() == l, flags
        res[flags] = (l, {k: v.clone() for k, v in eng.grads.items()})
        del eng
        torch.cuda.empty_cache()
    assert res[0][0] == res[8][0]                         # the forward is the same launch
    for k, ref in res[0][1].items():
        rel = float((res[8][1][k] - ref).norm() / ref.norm().clamp_min(1e-12))
        assert rel < 4e-3, (k, rel)
    assert torch.equal(res[0][1]["fc_output.weight"], res[8][1]["fc_output.weight"])     # not downstream of dz


def test_fused_optimizer_step_equals_unfused_step():
    """afr_train_step fuses AdamW of fc_output.weight into its dW GEMM; the result must equal backward + afr_adamw_step."""
    from .util import SheetConfig
    cfg = SheetConfig(max_length=24, sheet_h=16, sheet_w=40)
    x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(40), 24))
    t = torch.from_numpy(synth.synth_sheet_targets(40, 16, 40, tensor_id=974))
    for dtype in ("f32", "bf16"):
        a, b = _engine(cfg, dtype=dtype), _engine(cfg, dtype=dtype)
        for step in (1, 2, 3):
            a.train_step(x, t, step=step)                                   # fused
            b.train_step(x, t, step=step, do_step=False)                    # materialised gradients ...
            b.adamw_step()                                                  # ... then the stand-alone AdamW kernel
        assert abs(a.read_loss() - b.read_loss()) <= 1e-6 * b.read_loss() + 1e-7
        E = cfg.embed_dim
        for k in a.params:
            d = (a.params[k] - b.params[k]).abs()
            if k == "attention.in_proj_bias":      # k-bias: analytically zero gradient, Adam amplifies rounding noise
                d = torch.cat([d[:E], d[2 * E:]])
            # the two kernels may contract the update's multiply-adds differently: last-bit differences only
            assert float(d.max()) <= 3e-6 * max(1.0, float(b.params[k].abs().max())), (dtype, k)


# ----------------------------------------------------------------------------- glyph model: first-layer fold edge shapes
def _glyph_check(cfg, B, dtype="f32", tol=1e-4, ytol=2e-5, xmax=None):
    from .util import glyph_inputs
    x, font, tu8 = glyph_inputs(cfg, B)
    if xmax is not None:                                     # spread the codes over the whole vocabulary, repeats included
        x = (np.arange(B, dtype=np.int64) * 7) % xmax
    eng = _engine(cfg, dtype=dtype, max_batch=max(B, 8))
    P = tparams(cfg)
    rnd = engine_rounding(cfg, dtype)
    xt, ft = torch.from_numpy(x), torch.from_numpy(font)
    y = eng.forward(xt, ft if cfg.n_fonts else None).cpu().numpy()
    yref, cache = oracle.glyph_forward(P, xt, ft, cfg, rnd=rnd)
    assert maxabs(y, yref.numpy()) < ytol
    lref, du = oracle.mse_loss_grad(cache["u"], torch.from_numpy(tu8.astype(np.float32) / 255.0))
    Gref = oracle.glyph_backward(P, cache, rnd_du(rnd, du), cfg, rnd=rnd)
    eng.train_step(xt, torch.from_numpy(tu8), font=ft if cfg.n_fonts else None, do_step=False)
    assert abs(eng.read_loss() - float(lref)) < tol * float(lref)
    for k, g in eng.grads.items():
        ref = Gref[k].numpy()
        assert maxabs(g.cpu().numpy(), ref) <= tol * max(1e-7, float(np.abs(ref).max())), k


@pytest.mark.parametrize("cfg,B,xmax", [
    (dict(hidden=(48,), out_h=4, out_w=6, n_fonts=0), 300, None),                 # no font table
    (dict(hidden=(40, 24), out_h=4, out_w=4, n_fonts=3, vocab=100), 257, 100),    # vocab not a multiple of 8, every code used
    (dict(hidden=(264,), out_h=4, out_w=6, n_fonts=1, embed_dim=64), 33, None),   # wider embedding, fc1 wider than one table tile
    (dict(hidden=(24,), out_h=2, out_w=4, n_fonts=1, embed_dim=128), 40, None),   # widest embedding the table kernel stages (136 KB of LDS)
    (dict(hidden=(16,), out_h=2, out_w=4, n_fonts=2), 1, None),                   # one glyph
    (dict(hidden=(32,), out_h=4, out_w=4, n_fonts=2, vocab=600), 700, 600),       # table too wide to fold: plain gather + GEMM path
    (dict(hidden=(), out_h=4, out_w=6, n_fonts=2), 64, None),                     # no hidden layer: embedding -> output
])
def test_glyph_first_layer_fold_edge_shapes(cfg, B, xmax):
    """The glyph model's first Linear runs as a table gather (forward) and a one-hot-widened weight-gradient GEMM
    (backward); shapes around its tile and staging sizes, and the two configurations that must NOT take that path."""
    from .util import GlyphConfig
    _glyph_check(GlyphConfig(**cfg), B, xmax=xmax)


def test_glyph_fold_bf16_every_code_repeated():
    from .util import GlyphConfig
    _glyph_check(GlyphConfig(hidden=(64, 48), out_h=4, out_w=8, n_fonts=2), 1000, dtype="bf16", tol=3e-2, ytol=3e-2, xmax=128)


def test_grouped_reduce_op_sums_each_segment_in_order():
    """afr_op_reduce_group: several slab reductions in one launch (shallow and deep segments), each == the ordered sum."""
    import ctypes as C
    from ai_font_renderer_amd import _lib
    lib = _lib.lib()
    shapes = [(3, 4096), (40, 256), (8, 1000)]                 # (slabs, elements): 40 slabs takes the deep (4-wave) path
    srcs = [torch.from_numpy(synth.hash_uniform(980 + i, (s, n), 1.0)).cuda() for i, (s, n) in enumerate(shapes)]
    dsts = [torch.full((n,), float("nan"), device="cuda") for _, n in shapes]
    k = len(shapes)
    rc = lib.afr_op_reduce_group(k, (C.c_void_p * k)(*[d.data_ptr() for d in dsts]), (C.c_void_p * k)(*[s.data_ptr() for s in srcs]),
                                 (C.c_int * k)(*[s for s, _ in shapes]), (C.c_int64 * k)(*[n for _, n in shapes]),
                                 (C.c_int64 * k)(*[n for _, n in shapes]), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.afr_last_error()
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert float((d.cpu().double() - s.cpu().double().sum(0)).abs().max()) < 1e-5
