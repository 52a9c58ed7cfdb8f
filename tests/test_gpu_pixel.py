"""GPU: BASELINE configs[4]'s per-pixel-token transformer (config.PixelConfig) -- forward, backward and the AdamW
step through the C ABI (AFR_KIND_PIXEL: token-wise kernels of csrc/pixel.hip + the GEMM kernels) against the torch.nn twin
fixture (tests/golden/pixel_twin.npz, C5-mini) and the CPU oracle.  The reference has no such model: parity is pinned to torch.nn."""
import numpy as np
import pytest
import torch

from ai_font_renderer_amd import _lib
from .util import load, maxabs, oracle, synth, tparams

pytestmark = pytest.mark.gpu


def test_c5_mini_forward_matches_the_torch_nn_twin():
    from ai_font_renderer_amd.config import C5_MINI as cfg
    from ai_font_renderer_amd.engine import Engine
    fx = load("pixel_twin.npz")
    x, font = torch.from_numpy(fx["x"]), torch.from_numpy(fx["font"])
    eng = Engine(cfg, dtype="f32", max_batch=32, with_optimizer=False)
    eng.load_params(synth.make_params(cfg))
    assert [n for n, _, _, _ in eng.layout] == [k for k, _ in cfg.param_shapes()]      # the twin's state_dict order
    y = eng.forward(x, font).cpu().numpy()
    assert y.shape == fx["eval_y"].shape
    assert maxabs(y, fx["eval_y"]) < 2e-5                      # f32 mode: the 1e-4 bitmap bar of the north star, with margin
    assert eng.error_flags() == 0
    # ragged batch, repeated codes, a batch that is not a multiple of anything
    xs, fs = x[:13], font[:13]
    assert maxabs(eng.forward(xs, fs).cpu().numpy(), fx["eval_y"][:13]) < 2e-5
    # throughput mode: bf16 operands in every Linear, f32 residual stream and LayerNorm statistics
    e16 = Engine(cfg, dtype="bf16", max_batch=32, with_optimizer=False)
    e16.load_params(synth.make_params(cfg))
    d = maxabs(e16.forward(x, font).cpu().numpy(), fx["eval_y"])
    print(f"C5-mini bf16 forward: max-abs bitmap diff to the f32 twin {d:.3e}")
    assert d < 2.5e-2
    # an out-of-range code is flagged
    xb = x.clone()
    xb[3] = 128
    eng.forward(xb, font)
    assert eng.error_flags() & 1


def test_c5_no_fonts_and_other_widths_vs_the_oracle():
    from ai_font_renderer_amd.config import PixelConfig
    from ai_font_renderer_amd.engine import Engine
    for cfg, B in ((PixelConfig(out_h=4, out_w=6, d_model=128, heads=2, layers=2, ff_dim=200, n_fonts=0), 7),
                   (PixelConfig(out_h=16, out_w=16, d_model=256, heads=4, layers=1, ff_dim=512, n_fonts=3), 5)):
        eng = Engine(cfg, dtype="f32", max_batch=B, with_optimizer=False)
        eng.load_params(synth.make_params(cfg))
        x = torch.from_numpy((32 + (np.arange(B) * 11) % 95).astype(np.int64))
        font = torch.from_numpy((np.arange(B) % max(cfg.n_fonts, 1)).astype(np.int64))
        y = eng.forward(x, font if cfg.n_fonts else None).cpu()
        yref, _ = oracle.pixel_forward(tparams(cfg), x, font, cfg)
        assert float((y - yref).abs().max()) < 2e-5, cfg


def _check_against_twin(fx, cfg, prefix, T, tol, floor=0.0):
    """T: name -> numpy array, compared with the fixture's full tensors / row sums, column sums and samples (the oracle test's rule)."""
    n = 0
    for k, _ in cfg.param_shapes():
        got = T[k]
        if prefix + k in fx:
            ref = fx[prefix + k]
            assert maxabs(got, ref) <= max(tol * max(float(np.abs(ref).max()), 1e-12), floor), (prefix, k)
        else:
            g2 = got.reshape(got.shape[0], -1)
            for part, val, sc in (("rowsum", g2.sum(1), np.abs(g2).sum(1).max()), ("colsum", g2.sum(0), np.abs(g2).sum(0).max()),
                                  ("samples", got.reshape(-1)[fx[prefix + k + "/idx"]], np.abs(got).max())):
                ref = fx[f"{prefix}{k}/{part}"]
                assert maxabs(val, ref) <= max(tol * max(float(sc), 1e-12), floor * (1 if part == "samples" else len(got.reshape(-1)) ** 0.5)), (prefix, k, part)
        n += 1
    return n


def test_c5_mini_backward_and_adamw_trajectory_match_the_torch_nn_twin():
    """f32 mode: loss, EVERY gradient and a 3-step AdamW trajectory of the HIP path against the torch.nn twin (autograd +
    torch.optim.AdamW; make_golden.py pixel_twin), held to the bounds the CPU oracle is held to (test_oracle_golden.py)."""
    from ai_font_renderer_amd.config import C5_MINI as cfg
    from ai_font_renderer_amd.engine import Engine
    fx = load("pixel_twin.npz")
    x, font, tgt = torch.from_numpy(fx["x"]), torch.from_numpy(fx["font"]), torch.from_numpy(fx["target_u8"])
    lr = float(fx["lr"])
    eng = Engine(cfg, dtype="f32", max_batch=32)
    eng.load_params(synth.make_params(cfg))
    assert eng.backward_stages == cfg.layers
    # forward -> loss -> backward through the separate entry points
    eng.forward(x, font, training=True, want_output=False)
    eng.loss_grad(tgt)
    eng.backward()
    assert abs(eng.read_loss() - float(fx["losses"][0])) < 1e-6
    G = {n: eng.grads[n].cpu().numpy().copy() for n, _ in cfg.param_shapes()}
    # (the f32 twin itself sits 2.3e-3 of the largest entry from the same model evaluated in fp64 -- positional table and first
    # block, where a ReLU gate that is within rounding of zero flips -- so it is held to 4e-3; the explicit fp64 oracle, which
    # equals autograd to 4e-16 in fp64, is the tight reference: measured 1.6e-6)
    assert _check_against_twin(fx, cfg, "grad/", G, 4e-3) == len(cfg.param_shapes())
    P64 = {k: v.double() for k, v in tparams(cfg).items()}
    _, c64 = oracle.pixel_forward(P64, x, font, cfg)
    _, du64 = oracle.mse_loss_grad(c64["u"], tgt.double() / 255.0)
    G64 = oracle.pixel_backward(P64, c64, du64, cfg)
    for n, _ in cfg.param_shapes():
        ref = G64[n].numpy()
        assert maxabs(G[n], ref) <= 1e-5 * float(np.abs(ref).max()), n
    # the one-call step without the optimizer leaves the same gradients, bit for bit (same kernels, same order)
    eng.train_step(x, tgt, font=font, do_step=False)
    eng.read_loss()
    for n, _ in cfg.param_shapes():
        assert np.array_equal(eng.grads[n].cpu().numpy(), G[n]), n
    # three optimizer steps
    for t in (1, 2, 3):
        eng.train_step(x, tgt, font=font, lr=lr)
        assert abs(eng.read_loss() - float(fx["losses"][t - 1])) < 5e-6, t
    P = {n: eng.params[n].cpu().numpy() for n, _ in cfg.param_shapes()}
    _check_against_twin(fx, cfg, "param3/", P, 2e-5, floor=3.2 * lr)
    assert eng.error_flags() == 0


def test_c5_backward_other_shapes_and_bf16_vs_the_oracle():
    """Other widths / no font table / a ragged batch in f32 against the oracle's explicit backward; bf16 mode (bf16 GEMM operands,
    f32 residual stream and statistics) against the f32 oracle with the loose bound mixed precision allows."""
    from ai_font_renderer_amd.config import C5_MINI, PixelConfig
    from ai_font_renderer_amd.engine import Engine
    cases = ((PixelConfig(out_h=4, out_w=6, d_model=128, heads=2, layers=2, ff_dim=200, n_fonts=0), 7, "f32"),
             (PixelConfig(out_h=16, out_w=32, d_model=256, heads=4, layers=1, ff_dim=512, n_fonts=3), 5, "f32"),      # 512 tokens: two attention-backward chunks
             (C5_MINI, 24, "bf16"))
    for cfg, B, dt in cases:
        eng = Engine(cfg, dtype=dt, max_batch=B)
        eng.load_params(synth.make_params(cfg))
        rng = np.random.default_rng(5)
        x = torch.from_numpy((32 + (np.arange(B) * 11) % 95).astype(np.int64))
        font = torch.from_numpy((np.arange(B) % max(cfg.n_fonts, 1)).astype(np.int64))
        tgt = torch.from_numpy(rng.integers(0, 256, (B, cfg.out_h, cfg.out_w), dtype=np.uint8))
        eng.train_step(x, tgt, font=font if cfg.n_fonts else None, do_step=False)
        P = tparams(cfg)
        y, cache = oracle.pixel_forward(P, x, font, cfg)
        loss, du = oracle.mse_loss_grad(cache["u"], tgt.float() / 255.0)
        G = oracle.pixel_backward(P, cache, du, cfg)
        # f32: largest deviation against the tensor's largest entry.  bf16: every Linear rounds its operands to 8 bits of mantissa,
        # so single entries of a deep gradient move by several percent of the largest one; the tensor as a whole is held to a
        # relative Frobenius error (measured 0.9e-2 .. 3.4e-2 for the blocks and the head, whose du already carries the forward's 9e-3
        # bitmap deviation; 5.3e-2 at the positional table and the first LayerNorm, the end of the chain) and single entries to 15 % of the largest
        tol_l, tol_g = (1e-6, 2e-3) if dt == "f32" else (2e-3, 0.15)
        assert abs(eng.read_loss() - float(loss)) < tol_l * max(1.0, float(loss)), (cfg, dt)
        worst = 0.0
        for n, _ in cfg.param_shapes():
            ref = G[n].numpy()
            got = eng.grads[n].cpu().numpy()
            sc = max(float(np.abs(ref).max()), 1e-12)
            assert maxabs(got, ref) <= tol_g * sc, (dt, n, maxabs(got, ref) / sc)
            if dt == "bf16":
                fro = float(np.linalg.norm((got - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-30))
                worst = max(worst, fro)
                print(f"  {n:36s} {fro:.2e}")
                assert fro < 8e-2, (n, fro)
        if dt == "bf16":
            print(f"C5-mini bf16 gradients: worst relative Frobenius error to the f32 oracle {worst:.2e}")
        assert eng.error_flags() == 0




def test_gradient_accumulation_over_micro_batches_equals_the_whole_batch():
    """Engine(micro_batch=m): a batch of B > m samples runs as micro-steps of m (the last one shorter) whose gradients are summed
    -- how BASELINE configs[4]'s 2048 glyphs per GPU fit.  Loss and gradients equal the one-pass step of the whole batch up to
    the order of the f32 sums; two optimizer steps stay together."""
    from ai_font_renderer_amd.config import C5_MINI as cfg
    from ai_font_renderer_amd.engine import Engine
    B = 27
    rng = np.random.default_rng(9)
    x = torch.from_numpy((32 + (np.arange(B) * 11) % 95).astype(np.int64))
    font = torch.from_numpy((np.arange(B) % 2).astype(np.int64))
    tgt = torch.from_numpy(rng.integers(0, 256, (B, cfg.out_h, cfg.out_w), dtype=np.uint8))
    whole = Engine(cfg, dtype="f32", max_batch=B)
    acc = Engine(cfg, dtype="f32", max_batch=B, micro_batch=8)
    assert acc.max_batch == 8
    for e in (whole, acc):
        e.load_params(synth.make_params(cfg))
        e.train_step(x, tgt, font=font, do_step=False)
    lw, la = whole.read_loss(), acc.read_loss()
    assert abs(lw - la) < 1e-6 * lw
    for n, _ in cfg.param_shapes():
        ref = whole.grads[n].cpu().numpy()
        assert maxabs(acc.grads[n].cpu().numpy(), ref) <= 2e-5 * max(float(np.abs(ref).max()), 1e-12), n
    for e in (whole, acc):
        for _ in range(2):
            e.train_step(x, tgt, font=font, lr=1e-5)
    assert abs(whole.read_loss() - acc.read_loss()) < 1e-5
    assert acc.t == whole.t == 2
    # inference forward of the large batch, micro_batch rows at a time
    assert maxabs(acc.forward(x, font).cpu().numpy(), whole.forward(x, font).cpu().numpy()) < 2e-5


def test_full_width_block_with_32768_token_rows_takes_the_256x256_body_for_its_large_products():
    """One block of the FULL-WIDTH model (4096 pixel tokens, d_model 512, ff 2048) on 8 glyphs = 32768 token rows: the forward /
    input-gradient products of the MLP (1024 tiles of 256x256) and its weight gradients (16 K-slices of 2048 rows, bias gradient as
    the staged tiles' column sums) run on the 256x256 body here, as in the benchmarked step -- the miniature shapes of the
    other tests never reach those routes.  bf16 engine against the f32 oracle: every gradient within the mixed-precision bound;
    a missing K-slice or a misplaced tile would be an error of order one."""
    from ai_font_renderer_amd.config import PixelConfig
    from ai_font_renderer_amd.engine import Engine
    cfg = PixelConfig(layers=1)
    B = 8
    rng = np.random.default_rng(11)
    x = torch.from_numpy((40 + (np.arange(B) * 7) % 80).astype(np.int64))
    font = torch.from_numpy((np.arange(B) % 2).astype(np.int64))
    tgt = torch.from_numpy(rng.integers(0, 256, (B, cfg.out_h, cfg.out_w), dtype=np.uint8))
    eng = Engine(cfg, dtype="bf16", max_batch=B)
    eng.load_params(synth.make_params(cfg))
    eng.profile(1)
    eng.train_step(x, tgt, font=font, do_step=False)
    names = [r["kernel"] for r in eng.profile_table()]
    eng.profile(0)
    rows = B * cfg.tokens
    # (the 512-column products have 256 tiles here and stay on the ring kernel: their body route is covered at op level, tests/test_gpu_ops.py)
    for shape in (f"[{rows}x2048x512]<0,0>", f"[{rows}x2048x512]<0,1>", f"[2048x512x{rows}]<1,1>", f"[512x2048x{rows}]<1,1>"):
        assert "gemm_bf16_group256" + shape in names, (shape, names)
    P = tparams(cfg)
    _, cache = oracle.pixel_forward(P, x, font, cfg)
    loss, du = oracle.mse_loss_grad(cache["u"], tgt.float() / 255.0)
    G = oracle.pixel_backward(P, cache, du, cfg)
    assert abs(eng.read_loss() - float(loss)) < 3e-3 * float(loss)
    for n, _ in cfg.param_shapes():
        ref, got = G[n].numpy(), eng.grads[n].cpu().numpy()
        fro = float(np.linalg.norm((got - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-30))
        print(f"  {n:36s} {fro:.2e}")
        # measured: weight gradients of the routed products 1.1-1.5e-2 (bound 3e-2); bias gradients (column sums of the bf16 operand
        # tiles: signed values that largely cancel over 32768 rows) up to 5.1e-2; the positional table and the first LayerNorm, at the
        # end of the chain and summed over 8 glyphs only, 1.0e-1 and 6.2e-2
        bound = 1.5e-1 if n in ("positional_encoding", "layers.0.ln1.weight") else 8e-2 if n.endswith("bias") else 3e-2
        assert fro < bound, (n, fro)
    assert eng.error_flags() == 0


def test_pixel_plan_limits_and_caller_side_loss():
    """Shapes the plan refuses say so (never a silent wrong answer); the narrowest supported model (d_model 64, one head, one
    block, 8 pixel tokens, one context token) runs; a caller-side loss enters through set_output_grad exactly like the fused MSE."""
    from ai_font_renderer_amd.config import PixelConfig
    from ai_font_renderer_amd.engine import Engine
    for bad in (PixelConfig(d_model=576, heads=9), PixelConfig(d_model=256, heads=2), PixelConfig(ff_dim=100), PixelConfig(out_h=3, out_w=3)):
        with pytest.raises(_lib.AfrError):
            Engine(bad, dtype="f32", max_batch=2)
    cfg = PixelConfig(out_h=2, out_w=4, d_model=64, heads=1, layers=1, ff_dim=8, n_fonts=0)
    B = 3
    x = torch.tensor([33, 90, 126])
    tgt = torch.from_numpy(synth.hash_u8(934, (B, 2, 4)))
    eng = Engine(cfg, dtype="f32", max_batch=B)
    eng.load_params(synth.make_params(cfg))
    eng.train_step(x, tgt, do_step=False)
    P = tparams(cfg)
    y, cache = oracle.pixel_forward(P, x, None, cfg)
    loss, du = oracle.mse_loss_grad(cache["u"], tgt.float() / 255.0)
    G = oracle.pixel_backward(P, cache, du, cfg)
    assert abs(eng.read_loss() - float(loss)) < 1e-6
    for n, _ in cfg.param_shapes():
        ref = G[n].numpy()
        assert maxabs(eng.grads[n].cpu().numpy(), ref) <= 2e-4 * max(float(np.abs(ref).max()), 1e-12), n
    fused = {n: eng.grads[n].clone() for n, _ in cfg.param_shapes()}
    # the same gradient from a loss the caller differentiates itself: dy = d(mse)/d(clamped output)
    yy = eng.forward(x, training=True)
    dy = (2.0 * (yy - tgt.cuda().float() / 255.0) / yy.numel()).reshape(B, -1)
    eng.set_output_grad(dy)
    eng.backward()
    for n, _ in cfg.param_shapes():
        assert maxabs(eng.grads[n].cpu().numpy(), fused[n].cpu().numpy()) <= 1e-6 * max(float(fused[n].abs().max()), 1e-12), n
    assert eng.error_flags() == 0


def test_c5_full_size_forward_and_gradients_match_the_torch_nn_twin():
    """The FULL-SIZE model (4096 pixel tokens, 4 blocks) on the three glyphs of tests/golden/pixel_twin_full.npz, f32 mode: eval
    bitmaps, loss and every gradient of the HIP path against the torch.nn twin directly (16 chunks of the attention backward's
    key/value sums, every product many tiles tall)."""
    from ai_font_renderer_amd.config import C5 as cfg
    from ai_font_renderer_amd.engine import Engine
    fx = load("pixel_twin_full.npz")
    x, font, tgt = torch.from_numpy(fx["x"]), torch.from_numpy(fx["font"]), torch.from_numpy(fx["target_u8"])
    eng = Engine(cfg, dtype="f32", max_batch=3)
    eng.load_params(synth.make_params(cfg))
    assert maxabs(eng.forward(x, font).cpu().numpy(), fx["eval_y"]) < 2e-5
    eng.train_step(x, tgt, font=font, do_step=False)
    assert abs(eng.read_loss() - float(fx["loss"])) < 1e-6
    G = {n: eng.grads[n].cpu().numpy() for n, _ in cfg.param_shapes()}
    fxg = {k: fx[k] for k in fx.files}
    assert _check_against_twin(fxg, cfg, "grad/", G, 4e-3) == len(cfg.param_shapes())
    assert eng.error_flags() == 0
