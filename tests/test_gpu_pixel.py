"""GPU: BASELINE configs[4]'s per-pixel-token transformer (config.PixelConfig) -- the FORWARD path through the C
ABI (AFR_KIND_PIXEL: token-wise kernels of csrc/pixel.hip + the GEMM kernels) against the torch.nn twin fixture
(tests/golden/pixel_twin.npz, C5-mini) and the CPU oracle.  The reference has no such model: parity is pinned to torch.nn.
The training entry points of this kind are refused (no backward yet)."""
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


def test_c5_training_entry_points_are_refused_not_faked():
    from ai_font_renderer_amd.config import C5_MINI as cfg
    from ai_font_renderer_amd.engine import Engine
    eng = Engine(cfg, dtype="f32", max_batch=4)
    eng.load_params(synth.make_params(cfg))
    x, font = torch.tensor([40, 41, 42, 43]), torch.tensor([0, 1, 0, 1])
    t = torch.zeros(4, cfg.out_h, cfg.out_w, dtype=torch.uint8)
    with pytest.raises(_lib.AfrError, match="no training path"):
        eng.train_step(x, t, font=font)
    with pytest.raises(_lib.AfrError, match="no training path"):
        eng.forward_loss(x, t, font=font)
