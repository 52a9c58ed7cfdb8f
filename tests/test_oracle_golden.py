"""CPU: pin the oracle (oracle/afr_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import numpy as np
import torch

from .util import MINI, R0, load, maxabs, oracle, synth, tmasks, tparams


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_mini_eval_all_length_branches():
    fx = load("sheet_mini.npz")
    P = tparams(MINI)
    for key in ("10", "6", "14"):          # exact, zero-pad (model.py:190-193), truncate (model.py:163-164)
        y, _ = oracle.sheet_forward(P, _t(fx["x" + key]), MINI)
        assert maxabs(y.numpy(), fx["eval_y" + key]) < 2e-6, key


def _check_grads(fx, prefix, G, tol):
    for k, g in G.items():
        ref = fx[prefix + k]
        scale = max(1e-6, float(np.abs(ref).max()))
        assert maxabs(g.numpy(), ref) / scale < tol, (k, maxabs(g.numpy(), ref), scale)


def test_mini_train_grads_no_dropout():
    fx = load("sheet_mini.npz")
    P = tparams(MINI)
    tgt = _t(fx["target_u8"].astype(np.float32) / 255.0)
    for key, pre in (("x10", "nodrop"), ("x6", "nodrop6")):
        _, cache = oracle.sheet_forward(P, _t(fx[key]), MINI)
        loss, du = oracle.mse_loss_grad(cache["u"], tgt)
        assert abs(float(loss) - float(fx[pre + "_loss"])) < 1e-6
        _check_grads(fx, pre + "_grad/", oracle.sheet_backward(P, cache, du, MINI), 2e-5)


def test_mini_train_grads_injected_dropout():
    """The three dropouts (model.py:137,144,149) with the counter-hash masks injected into the
    reference's F.dropout: same masks, same scaling, same order."""
    fx = load("sheet_mini.npz")
    P = tparams(MINI)
    tgt = _t(fx["target_u8"].astype(np.float32) / 255.0)
    masks = tmasks(synth.sheet_dropout_masks(MINI, 5, 10, seed=42, step=7))
    y, cache = oracle.sheet_forward(P, _t(fx["x10"]), MINI, masks)
    assert maxabs(y.numpy(), fx["drop_y"]) < 2e-6
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    assert abs(float(loss) - float(fx["drop_loss"])) < 1e-6
    _check_grads(fx, "drop_grad/", oracle.sheet_backward(P, cache, du, MINI), 2e-5)


def test_mini_three_adamw_steps():
    fx = load("sheet_mini.npz")
    P = tparams(MINI)
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    tgt = _t(fx["target_u8"].astype(np.float32) / 255.0)
    x = _t(fx["x10"])
    for t in (1, 2, 3):
        loss, _, P, M, V = oracle.train_step(P, M, V, t, x, tgt, MINI)
        assert abs(float(loss) - float(fx["adamw_losses"][t - 1])) < 2e-6
    E = MINI.embed_dim
    for k in P:
        got, ref = P[k].numpy(), fx["adamw_param/" + k]
        if k == "attention.in_proj_bias":
            # d(loss)/d(k-bias) is analytically 0 (softmax is shift invariant), so its gradient is
            # rounding noise and Adam turns noise into +-lr steps: only bounded, never reproducible.
            assert maxabs(got[E:2 * E], ref[E:2 * E]) < 3 * 1e-3 * 1.01
            got, ref = np.delete(got, np.s_[E:2 * E]), np.delete(ref, np.s_[E:2 * E])
        assert maxabs(got, ref) < 5e-6, k


def test_r0_test_strings_eval():
    """The shipped configuration on the reference's 15 test_strings (model.py:111-127)."""
    fx = load("sheet_r0.npz")
    P = tparams(R0)
    y, _ = oracle.sheet_forward(P, _t(fx["test_x"]), R0)
    assert maxabs(y.numpy(), fx["test_eval_y"]) < 5e-6


def test_r0_train_step_grads():
    fx = load("sheet_r0.npz")
    P = tparams(R0)
    tgt = _t(synth.synth_sheet_targets(8, 80, 240, tensor_id=902).astype(np.float32) / 255.0)
    _, cache = oracle.sheet_forward(P, _t(fx["train_x"]), R0)
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    assert abs(float(loss) - float(fx["train_loss"])) < 2e-6
    G = oracle.sheet_backward(P, cache, du, R0)
    gw = G.pop("fc_output.weight").numpy()
    _check_grads(fx, "train_grad/", G, 5e-5)
    assert maxabs(gw.sum(1), fx["train_gradW_rowsum"]) / np.abs(fx["train_gradW_rowsum"]).max() < 5e-5
    assert maxabs(gw.sum(0), fx["train_gradW_colsum"]) / np.abs(fx["train_gradW_colsum"]).max() < 5e-5
    s = gw.reshape(-1)[fx["train_gradW_idx"]]
    assert maxabs(s, fx["train_gradW_samples"]) / np.abs(fx["train_gradW_samples"]).max() < 5e-5


def test_oracle_backward_matches_autograd_fp64():
    """Independent of the fixtures: the explicit backward equals autograd of the explicit forward."""
    cfg = MINI
    P = {k: v.double().requires_grad_(True) for k, v in tparams(cfg).items()}
    x = _t(synth.encode_strings(["ABCDEFGH", "X Y Z"], 10))[:, :8]
    masks = tmasks(synth.sheet_dropout_masks(cfg, 2, 8, seed=1, step=3))
    tgt = torch.rand(2, 8, 24, dtype=torch.float64)
    y, cache = oracle.sheet_forward(P, x, cfg, masks)
    loss = ((y - tgt) ** 2).mean()
    loss.backward()
    with torch.no_grad():
        l2, du = oracle.mse_loss_grad(cache["u"], tgt)
        G = oracle.sheet_backward(P, cache, du, cfg)
    assert abs(float(l2) - float(loss)) < 1e-12
    for k in P:
        assert maxabs(G[k].numpy(), P[k].grad.numpy()) < 1e-10, k


def test_glyph_oracle_backward_matches_autograd_fp64():
    from .util import GlyphConfig, glyph_inputs
    cfg = GlyphConfig(hidden=(48, 40), out_h=4, out_w=6, n_fonts=2)
    P = {k: v.double().requires_grad_(True) for k, v in tparams(cfg).items()}
    x, font, t = glyph_inputs(cfg, 200)
    x, font = _t(x), _t(font)
    tgt = _t(t.astype(np.float64) / 255.0)
    y, cache = oracle.glyph_forward(P, x, font, cfg)
    loss = ((y - tgt) ** 2).mean()
    loss.backward()
    with torch.no_grad():
        l2, du = oracle.mse_loss_grad(cache["u"], tgt)
        G = oracle.glyph_backward(P, cache, du, cfg)
    assert abs(float(l2) - float(loss)) < 1e-12
    for k in P:
        assert maxabs(G[k].numpy(), P[k].grad.numpy()) < 1e-10, k


# ----------------------------------------------------------------------------- glyph MLP oracle: pinned, not self-checked
def _glyph_ref1():
    from .util import GlyphConfig
    fx = load("glyph_ref1.npz")
    h, w = fx["y"].shape[1:]
    cfg = GlyphConfig(hidden=(64,), out_h=h, out_w=w, embed_dim=32, vocab=128, n_fonts=0)
    P = {"embedding.weight": _t(fx["table"])}
    for k in ("fc1.weight", "fc1.bias", "fc_output.weight", "fc_output.bias"):
        P[k] = _t(fx["param/" + k])
    return fx, cfg, P


def test_glyph_oracle_matches_reference_class_at_max_length_1():
    """the reference's AttentionFontRenderer(max_length=1) ends in exactly the glyph network
    (fc1+ReLU -> fc_output -> clamp, model.py:148,152-156,183,196-202) fed by a per-code vector; the fixture holds that
    table and the REFERENCE's outputs, loss, autograd gradients and tensors after one step of its AdamW."""
    fx, cfg, P = _glyph_ref1()
    x = _t(fx["x"])
    tgt = _t(fx["target_u8"].astype(np.float32) / 255.0)
    y, cache = oracle.glyph_forward(P, x, None, cfg)
    assert maxabs(y.numpy(), fx["y"]) < 2e-6
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    G = oracle.glyph_backward(P, cache, du, cfg)
    for k in ("fc1.weight", "fc1.bias", "fc_output.weight", "fc_output.bias"):
        ref = fx["grad/" + k]
        assert maxabs(G[k].numpy(), ref) <= 2e-5 * float(np.abs(ref).max()), k
        p1, _, _ = oracle.adamw_step(P[k], G[k], torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1)
        assert maxabs(p1.numpy(), fx["step1/" + k]) < 2e-6, k


def _twin_case(tag):
    from .util import GlyphConfig, glyph_inputs
    from ai_font_renderer_amd.config import WORKLOADS
    cfg, B = (GlyphConfig(hidden=(48, 40), out_h=4, out_w=6, n_fonts=2), 300) if tag == "small" else (WORKLOADS["c1"]["cfg"], 95)
    x, font, tu8 = glyph_inputs(cfg, B)
    return cfg, _t(x), _t(font), _t(tu8.astype(np.float32) / 255.0)


def test_glyph_oracle_matches_torch_nn_twin_fonts_and_two_hidden_layers():
    """Shapes the reference class cannot express (font table, deeper stack, BASELINE C1): goldens from a torch.nn-composed
    twin that make_golden.py checks against the imported reference on the overlapping parameterisation."""
    fx = load("glyph_twin.npz")
    for tag in ("small", "c1"):
        cfg, x, font, tgt = _twin_case(tag)
        P = tparams(cfg)
        y, cache = oracle.glyph_forward(P, x, font, cfg)
        assert maxabs(y.numpy(), fx[f"{tag}/eval_y"]) < 2e-6, tag
        loss, du = oracle.mse_loss_grad(cache["u"], tgt)
        assert abs(float(loss) - float(fx[f"{tag}/losses"][0])) < 1e-6
        G = oracle.glyph_backward(P, cache, du, cfg)
        n = 0
        for k in P:
            if f"{tag}/grad/{k}" in fx:
                ref = fx[f"{tag}/grad/{k}"]
                assert maxabs(G[k].numpy(), ref) <= 2e-5 * float(np.abs(ref).max()), (tag, k)
                n += 1
        assert n >= 5
        M = {k: torch.zeros_like(v) for k, v in P.items()}
        V = {k: torch.zeros_like(v) for k, v in P.items()}
        for t in (1, 2, 3):
            loss, _, P, M, V = oracle.train_step(P, M, V, t, x, tgt, cfg, font=font)
            assert abs(float(loss) - float(fx[f"{tag}/losses"][t - 1])) < 2e-6, (tag, t)
        for k in P:
            assert maxabs(P[k].numpy(), fx[f"{tag}/param3/{k}"]) < 5e-6, (tag, k)


def test_text_generator_matches_survey_strings():
    """generate_font.ts:164-199 restated; first strings for seeds 42..44."""
    assert synth.lcg_text(42) == "P JAL WZ MQWPCDYYX EOGYVE MBANVV"
    assert synth.lcg_text(43) == "GG U AJBHEQVVO ZFU TFI G PHRPSUL"
    assert synth.lcg_text(44) == "YHS IYXCTW TBALZN YHXKESJ CHFW BM"
    lens = [len(synth.lcg_text(42 + i)) for i in range(2000)]
    assert min(lens) >= 10 and max(lens) <= 100


def test_inplace_adamw_of_the_cpu_baseline_equals_the_functional_form():
    """bench.py's cpu_baseline times oracle.train_step(inplace=True); its AdamW must be the same update as adamw_step (which the
    goldens pin against torch.optim.AdamW) -- equal to f32 rounding (lerp_/addcmul_/addcdiv_ fuse differently)."""
    import torch
    g = torch.Generator().manual_seed(3)
    p, gr = torch.randn(4096, generator=g), torch.randn(4096, generator=g) * 0.1
    m, v = torch.randn(4096, generator=g) * 0.01, torch.rand(4096, generator=g) * 1e-3
    for t in (1, 2, 7):
        want = oracle.adamw_step(p, gr, m, v, t)
        got = oracle.adamw_step_(p.clone(), gr, m.clone(), v.clone(), t)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 2e-7 * float(b.abs().max()) + 1e-9


def test_pixel_transformer_oracle_matches_the_torch_nn_twin():
    """BASELINE configs[4] (per-pixel-token transformer; config.PixelConfig) has no class in the reference: PARITY
    UNPINNED BY THE REFERENCE, pinned to a torch.nn composition of the reference's own layer idioms (tests/golden/
    pixel_twin.npz, make_golden.py pixel_twin: C5-mini = 64 pixel tokens, d_model 512, 8 heads, 4 layers, ff 2048, batch 32).
    oracle.pixel_forward / pixel_backward (explicit algebra, no torch.nn, no autograd) against its eval bitmaps, loss, every
    gradient (small tensors in full; the large ones by row sums, column sums and 2048 samples) and a 3-step AdamW trajectory."""
    from ai_font_renderer_amd.config import C5_MINI as cfg
    fx = load("pixel_twin.npz")
    x, font = torch.from_numpy(fx["x"]), torch.from_numpy(fx["font"])
    tgt = torch.from_numpy(fx["target_u8"].astype(np.float32) / 255.0)
    P = tparams(cfg)
    assert list(P) == [k for k, _ in cfg.param_shapes()]
    y, cache = oracle.pixel_forward(P, x, font, cfg)
    assert maxabs(y.numpy(), fx["eval_y"]) < 1e-5
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    assert abs(float(loss) - float(fx["losses"][0])) < 1e-6
    G = oracle.pixel_backward(P, cache, du, cfg)

    def check(prefix, T, tol, floor=0.0):
        n = 0
        for k in P:
            got = T[k].numpy()
            if prefix + k in fx:
                ref = fx[prefix + k]
                assert maxabs(got, ref) <= max(tol * max(float(np.abs(ref).max()), 1e-12), floor), (prefix, k)
            else:
                g2 = got.reshape(got.shape[0], -1)
                # (a sum is held to the size of what it adds up: some of these sums are analytically zero -- a LayerNorm follows)
                for part, val, sc in (("rowsum", g2.sum(1), np.abs(g2).sum(1).max()), ("colsum", g2.sum(0), np.abs(g2).sum(0).max()),
                                      ("samples", got.reshape(-1)[fx[prefix + k + "/idx"]], np.abs(got).max())):
                    ref = fx[f"{prefix}{k}/{part}"]
                    assert maxabs(val, ref) <= max(tol * max(float(sc), 1e-12), floor * (1 if part == "samples" else len(got.reshape(-1)) ** 0.5)), (prefix, k, part)
            n += 1
        return n

    # (f32 on both sides, two summation orders, four LayerNorm backward passes of cancelling terms down to the positional table:
    # measured agreement 5e-4 of a tensor's largest entry; in fp64 the oracle's backward equals autograd to 4e-16)
    assert check("grad/", G, 2e-3) == len(cfg.param_shapes())
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    for t in (1, 2, 3):
        loss, _, P, M, V = oracle.train_step(P, M, V, t, x, tgt, cfg, font=font, lr=float(fx["lr"]))
        assert abs(float(loss) - float(fx["losses"][t - 1])) < 5e-6, t
    # entries whose gradient is analytically zero (the key projection's bias and what hangs on it) get Adam steps of +-lr from
    # rounding noise, in either direction on either side: three steps of lr is the floor of any parameter comparison
    check("param3/", P, 2e-5, floor=3.2 * float(fx["lr"]))


def test_pixel_transformer_oracle_matches_the_full_size_twin():
    """The full-size configuration (4096 pixel tokens per glyph; tests/golden/pixel_twin_full.npz, make_golden.py pixel_twin_full,
    three glyphs): eval bitmaps, loss and every gradient of the torch.nn twin against the explicit oracle."""
    from ai_font_renderer_amd.config import C5 as cfg
    fx = load("pixel_twin_full.npz")
    x, font = torch.from_numpy(fx["x"]), torch.from_numpy(fx["font"])
    tgt = torch.from_numpy(fx["target_u8"].astype(np.float32) / 255.0)
    P = tparams(cfg)
    y, cache = oracle.pixel_forward(P, x, font, cfg)
    assert maxabs(y.numpy(), fx["eval_y"]) < 1e-5
    loss, du = oracle.mse_loss_grad(cache["u"], tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    G = oracle.pixel_backward(P, cache, du, cfg)
    for k in P:
        got = G[k].numpy()
        if "grad/" + k in fx:
            ref = fx["grad/" + k]
            assert maxabs(got, ref) <= 4e-3 * max(float(np.abs(ref).max()), 1e-12), k
        else:
            g2 = got.reshape(got.shape[0], -1)
            for part, val, sc in (("rowsum", g2.sum(1), np.abs(g2).sum(1).max()), ("colsum", g2.sum(0), np.abs(g2).sum(0).max()),
                                  ("samples", got.reshape(-1)[fx["grad/" + k + "/idx"]], np.abs(got).max())):
                assert maxabs(val, fx[f"grad/{k}/{part}"]) <= 4e-3 * max(float(sc), 1e-12), (k, part)
