#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):
    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Fixtures hold inputs and expected outputs only.  Weights are regenerated from the counter hash
(ai_font_renderer_amd.synth.make_params) and loaded with load_state_dict, so no weight file is
stored.  What is captured:
  sheet_mini.npz  mini sheet model (8x24 sheet, max_length 10): eval outputs for L=10, L=6 (pad
                  branch, model.py:190-193) and L=14 (truncate branch, model.py:163-164); train-mode
                  loss + all 12 grads with dropout off and with injected dropout masks; 3 AdamW steps
  sheet_r0.npz    the shipped model (80x240, L=100): eval outputs of the 15 test_strings
                  (model.py:111-127) and a B=8 train-mode step: loss, 11 small grads in full,
                  fc_output.weight.grad as row sums, column sums and 4096 hashed samples
  glyph_ref1.npz  the reference class at max_length=1 (attention over one token = out_proj(v_proj(e))):
                  its tail LayerNorm-output -> fc1+ReLU -> fc_output -> clamp IS the glyph MLP with hidden=(64,).  Stored:
                  the per-code table n[c] the reference's own front end produces (fed to the glyph net as its embedding
                  table), outputs, loss, the fc1 / fc_output gradients from the reference's autograd and those tensors
                  after one AdamW step of the reference's optimizer
  glyph_twin.npz  a torch.nn-composed twin (nn.Embedding [+ font nn.Embedding], nn.Linear/ReLU stack, clamp, F.mse_loss,
                  optim.AdamW with the reference's hyper-parameters, model.py:136,148,152-156,268-273) checked IN THIS
                  SCRIPT against the imported reference on that overlapping parameterisation, then run on the shapes the
                  reference class cannot express: a small net with a font table and two hidden layers (eval output, loss,
                  all gradients, 3 AdamW steps) and BASELINE C1 (16x16, hidden 256, batch 95: losses + parameters after 3
                  steps + step-1 gradients)
  glyph_bitmaps.npz  the rasterised targets of the BASELINE glyph configs: FiraCode-Retina 95 printable ASCII at 16x16 (C1/C2) and
                  FiraCode-Retina + Montserrat-Regular at 32x32 (C3/C4), drawn with PIL/FreeType by
                  ai_font_renderer_amd.datagen.render_glyphs from the two TTFs the reference ships (data, read only here)
  train_loop.npz  the reference's own train_attention_model (model.py:209-384) on an 80-sheet mini dataset (8x24 sheets, 10
                  characters, batch 16, dropout off, plateau patience 1, early-stop patience 4, <= 16 epochs; run "a" LR 0.004, run "b" LR 0.03):
                  per-epoch validation loss and learning rate as its scheduler saw them, the train losses it printed, its
                  training_results.txt, and the parameters it ended with
  cpu_step_times.json  (on request: `make_golden.py cpu_step_times`) step time of the reference's own training step against the
                  oracle's on the same R0 shapes in this container: the provenance of bench.py's cpu_baseline "port"
  helpers.npz     binary_array_to_image truncation (helpers.py:33), image_to_binary_array (helpers.py:121)
                  on a 24-bit top-down BMP written per generate_font.ts:6-62
"""
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import model as ref  # noqa: E402  (the reference's model.py)
import helpers as ref_helpers  # noqa: E402

from ai_font_renderer_amd import synth  # noqa: E402
from ai_font_renderer_amd.config import GlyphConfig, SheetConfig, WORKLOADS  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def build_ref(cfg):
    ref.SHEET_HEIGHT, ref.SHEET_WIDTH = cfg.sheet_h, cfg.sheet_w
    m = ref.AttentionFontRenderer(max_length=cfg.max_length)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()}
    m.load_state_dict(sd)
    return m


class InjectedDropout:
    """Replace F.dropout (used by nn.Dropout and by multi_head_attention_forward) with supplied
    keep masks, looked up by tensor shape."""

    def __init__(self, masks_by_shape):
        self.masks = masks_by_shape
        self.calls = 0

    def __enter__(self):
        self.orig = F.dropout

        def fake(input, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return input
            mask = self.masks[tuple(input.shape)]
            self.calls += 1
            return input * (mask.to(input.dtype) * (1.0 / (1.0 - p)))

        F.dropout = fake
        torch.nn.functional.dropout = fake
        return self

    def __exit__(self, *a):
        F.dropout = self.orig
        torch.nn.functional.dropout = self.orig


def grads_of(m):
    return {k: p.grad.detach().clone().numpy() for k, p in m.named_parameters()}


def train_grads(m, x, t, masks=None, cfg=None):
    m.train()
    m.zero_grad()
    if masks is None:
        m.embedding_dropout.p = 0.0
        m.dropout1.p = 0.0
        m.attention.dropout = 0.0
        out = m(x)
    else:
        m.embedding_dropout.p = cfg.p_embed
        m.dropout1.p = cfg.p_fc
        m.attention.dropout = cfg.p_attn
        B, L = x.shape[0], min(x.shape[1], cfg.max_length)
        by_shape = {
            (B, L, cfg.embed_dim): torch.from_numpy(masks["embed"]),
            (B * cfg.heads, L, L): torch.from_numpy(masks["attn"]).reshape(B * cfg.heads, L, L),
            (B, L, cfg.fc_dim): torch.from_numpy(masks["fc"]),
        }
        with InjectedDropout(by_shape) as inj:
            out = m(x)
        assert inj.calls == 3, inj.calls
    loss = F.mse_loss(out, t.view(out.shape))
    loss.backward()
    return float(loss.item()), grads_of(m), out.detach().numpy()


def mini():
    cfg = SheetConfig(max_length=10, sheet_h=8, sheet_w=24)
    m = build_ref(cfg)
    strings = ["HELLO WORL", "AB CD", "ZZZZZZZZZZ", "Q W E R T ", "  MIX  UP "]
    x10 = torch.from_numpy(synth.encode_strings(strings, 10))
    x6 = x10[:, :6].contiguous()
    x14 = torch.cat([x10, x10[:, :4]], dim=1).contiguous()
    tgt = torch.from_numpy(synth.synth_sheet_targets(5, 8, 24, tensor_id=901).astype(np.float32) / 255.0)
    fx = dict(x10=x10.numpy(), x6=x6.numpy(), x14=x14.numpy(), target_u8=synth.synth_sheet_targets(5, 8, 24, tensor_id=901))
    m.eval()
    with torch.no_grad():
        fx["eval_y10"] = m(x10).numpy()
        fx["eval_y6"] = m(x6).numpy()
        fx["eval_y14"] = m(x14).numpy()
    loss, g, _ = train_grads(m, x10, tgt)
    fx["nodrop_loss"] = np.float32(loss)
    for k, v in g.items():
        fx["nodrop_grad/" + k] = v
    loss6, g6, _ = train_grads(m, x6, tgt)
    fx["nodrop6_loss"] = np.float32(loss6)
    for k, v in g6.items():
        fx["nodrop6_grad/" + k] = v
    masks = synth.sheet_dropout_masks(cfg, 5, 10, seed=42, step=7)
    lossd, gd, outd = train_grads(m, x10, tgt, masks=masks, cfg=cfg)
    fx["drop_loss"] = np.float32(lossd)
    fx["drop_y"] = outd
    for k, v in gd.items():
        fx["drop_grad/" + k] = v
    # 3 AdamW steps, dropout off (model.py:273 hyper-parameters)
    m = build_ref(cfg)
    m.train()
    m.embedding_dropout.p = 0.0
    m.dropout1.p = 0.0
    m.attention.dropout = 0.0
    opt = torch.optim.AdamW(m.parameters(), lr=ref.LEARNING_RATE, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m(x10)
        loss = F.mse_loss(out, tgt.view(out.shape))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    fx["adamw_losses"] = np.array(losses, dtype=np.float32)
    for k, v in m.state_dict().items():
        fx["adamw_param/" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "sheet_mini.npz"), **fx)
    print("sheet_mini.npz", {k: np.asarray(v).shape for k, v in list(fx.items())[:8]})


def r0():
    cfg = SheetConfig()
    m = build_ref(cfg)
    x = torch.from_numpy(synth.encode_strings(ref.test_strings, cfg.max_length))   # helpers.py:57-59
    fx = dict(test_x=x.numpy())
    m.eval()
    with torch.no_grad():
        fx["test_eval_y"] = m(x).numpy()
    strings = synth.dataset_strings(8)
    xb = torch.from_numpy(synth.encode_strings(strings, cfg.max_length))
    tu8 = synth.synth_sheet_targets(8, 80, 240, tensor_id=902)
    tgt = torch.from_numpy(tu8.astype(np.float32) / 255.0)
    loss, g, _ = train_grads(m, xb, tgt)
    fx["train_x"] = xb.numpy()
    fx["train_loss"] = np.float32(loss)
    for k, v in g.items():
        if k == "fc_output.weight":
            fx["train_gradW_rowsum"] = v.sum(1)
            fx["train_gradW_colsum"] = v.sum(0)
            idx = (synth._counter(77, 4096, 42) % np.uint64(v.size)).astype(np.int64)
            fx["train_gradW_idx"] = idx
            fx["train_gradW_samples"] = v.reshape(-1)[idx]
        else:
            fx["train_grad/" + k] = v
    np.savez_compressed(os.path.join(OUT, "sheet_r0.npz"), **fx)
    print("sheet_r0.npz eval", fx["test_eval_y"].shape, "loss", loss)


class GlyphTwin(torch.nn.Module):
    """The glyph MLP composed from the torch.nn modules the reference uses for the same idioms (model.py:136,148,
    152-156): nn.Embedding gather [+ a second nn.Embedding for the font id], nn.Linear + ReLU per hidden layer,
    nn.Linear + clamp(0,1) output.  Parameter names = GlyphConfig.param_shapes()."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embedding = torch.nn.Embedding(cfg.vocab, cfg.embed_dim)
        if cfg.n_fonts > 0:
            self.font_embedding = torch.nn.Embedding(cfg.n_fonts, cfg.embed_dim)
        k = cfg.embed_dim
        for i, h in enumerate(cfg.hidden):
            setattr(self, f"fc{i + 1}", torch.nn.Linear(k, h))
            k = h
        self.fc_output = torch.nn.Linear(k, cfg.pixels)

    def forward(self, x, font=None):
        h = self.embedding(x)
        if self.cfg.n_fonts > 0:
            h = h + self.font_embedding(font)
        for i in range(len(self.cfg.hidden)):
            h = torch.relu(getattr(self, f"fc{i + 1}")(h))
        return torch.clamp(self.fc_output(h), 0, 1).view(-1, self.cfg.out_h, self.cfg.out_w)


def glyph_inputs(cfg, B, seed=0):            # same recipe as tests/util.glyph_inputs
    i = np.arange(B)
    x = (32 + (i % 95)).astype(np.int64)
    font = ((i // 95) % max(cfg.n_fonts, 1)).astype(np.int64)
    return x, font, synth.hash_u8(910 + seed, (B, cfg.out_h, cfg.out_w))


def glyph_ref1():
    """Reference AttentionFontRenderer(max_length=1) vs the glyph twin with hidden=(64,)."""
    h, w = 8, 12
    scfg = SheetConfig(max_length=1, sheet_h=h, sheet_w=w)
    m = build_ref(scfg)
    B = 200
    x = torch.from_numpy((32 + (np.arange(B) * 7) % 95).astype(np.int64)).view(B, 1)
    tu8 = synth.hash_u8(915, (B, h, w))
    tgt = torch.from_numpy(tu8.astype(np.float32) / 255.0)
    # the reference's own front end, per code: n[c] = LayerNorm(e + out_proj(v_proj(e))), e = Emb[c] + pos[0]
    m.eval()
    with torch.no_grad():
        codes = torch.arange(128).view(128, 1)
        e = m.embedding(codes) + m.positional_encoding[:1].unsqueeze(0)
        a, _ = m.attention(e.transpose(0, 1), e.transpose(0, 1), e.transpose(0, 1))
        ntab = m.layer_norm(e + a.transpose(0, 1)).reshape(128, -1).clone()
        y_eval = m(x).numpy()
    loss, g, y_train = train_grads(m, x, tgt)                # dropout p = 0: train mode == eval arithmetic
    assert np.abs(y_train - y_eval).max() == 0.0
    gcfg = GlyphConfig(hidden=(64,), out_h=h, out_w=w, embed_dim=32, vocab=128, n_fonts=0)
    tw = GlyphTwin(gcfg)
    sd = {"embedding.weight": ntab, "fc1.weight": m.fc1.weight.detach().clone(), "fc1.bias": m.fc1.bias.detach().clone(),
          "fc_output.weight": m.fc_output.weight.detach().clone(), "fc_output.bias": m.fc_output.bias.detach().clone()}
    tw.load_state_dict(sd)
    out = tw(x.view(-1))
    l2 = F.mse_loss(out, tgt)
    l2.backward()
    assert np.abs(out.detach().numpy() - y_eval).max() < 1e-6, np.abs(out.detach().numpy() - y_eval).max()
    assert abs(float(l2) - loss) < 1e-7
    shared = ("fc1.weight", "fc1.bias", "fc_output.weight", "fc_output.bias")
    for k in shared:
        tg = dict(tw.named_parameters())[k].grad.numpy()
        assert np.abs(tg - g[k]).max() <= 2e-6 * max(1e-6, np.abs(g[k]).max()), k
    # one step of the reference's optimizer on the reference (model.py:273)
    opt = torch.optim.AdamW(m.parameters(), lr=ref.LEARNING_RATE, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))
    opt.step()
    fx = dict(x=x.view(-1).numpy(), target_u8=tu8, table=ntab.numpy(), y=y_eval, loss=np.float32(loss))
    for k in shared:
        fx["param/" + k] = sd[k].numpy()
        fx["grad/" + k] = g[k]
        fx["step1/" + k] = dict(m.named_parameters())[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "glyph_ref1.npz"), **fx)
    print("glyph_ref1.npz loss", loss, "(twin == reference on the shared tensors)")


def glyph_twin():
    fx = {}
    for tag, cfg, B, full in (("small", GlyphConfig(hidden=(48, 40), out_h=4, out_w=6, n_fonts=2), 300, True),
                              ("c1", WORKLOADS["c1"]["cfg"], 95, False)):
        x, font, tu8 = glyph_inputs(cfg, B)
        xt, ft = torch.from_numpy(x), torch.from_numpy(font)
        tgt = torch.from_numpy(tu8.astype(np.float32) / 255.0)
        tw = GlyphTwin(cfg)
        tw.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()})
        with torch.no_grad():
            fx[f"{tag}/eval_y"] = tw(xt, ft).numpy()
        opt = torch.optim.AdamW(tw.parameters(), lr=ref.LEARNING_RATE, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))
        losses = []
        for step in range(3):
            opt.zero_grad()
            loss = F.mse_loss(tw(xt, ft), tgt)
            loss.backward()
            if step == 0:
                for k, p in tw.named_parameters():
                    if full or p.numel() <= 70000:
                        fx[f"{tag}/grad/{k}"] = p.grad.detach().clone().numpy()
            opt.step()
            losses.append(loss.item())
        fx[f"{tag}/losses"] = np.array(losses, dtype=np.float32)
        for k, p in tw.named_parameters():
            fx[f"{tag}/param3/{k}"] = p.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "glyph_twin.npz"), **fx)
    print("glyph_twin.npz", {k: fx[k].shape for k in list(fx)[:6]})


class PixelTwin(torch.nn.Module):
    """torch.nn composition of config.PixelConfig (BASELINE configs[4] as DESIGN.md 8 defines it) from the reference's layer
    idioms: nn.Embedding (model.py:136), a learned positional nn.Parameter (:140-141), nn.MultiheadAttention (:144), nn.LayerNorm
    (:145), nn.Linear + relu (:148,183), nn.Linear + clamp (:152-156).  The reference itself has no such class."""

    class Block(torch.nn.Module):
        def __init__(self, d, heads, ff, eps):
            super().__init__()
            self.ln1 = torch.nn.LayerNorm(d, eps=eps)
            self.attn = torch.nn.MultiheadAttention(embed_dim=d, num_heads=heads, batch_first=True)
            self.ln2 = torch.nn.LayerNorm(d, eps=eps)
            self.fc1 = torch.nn.Linear(d, ff)
            self.fc2 = torch.nn.Linear(ff, d)

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.d_model
        self.embedding = torch.nn.Embedding(cfg.vocab, d)
        if cfg.n_fonts > 0:
            self.font_embedding = torch.nn.Embedding(cfg.n_fonts, d)
        self.positional_encoding = torch.nn.Parameter(torch.zeros(cfg.tokens, d))
        self.layers = torch.nn.ModuleList([PixelTwin.Block(d, cfg.heads, cfg.ff_dim, cfg.ln_eps) for _ in range(cfg.layers)])
        self.ln_f = torch.nn.LayerNorm(d, eps=cfg.ln_eps)
        self.fc_output = torch.nn.Linear(d, 1)

    def forward(self, x, font=None):
        B = x.shape[0]
        ctx = self.embedding(x).unsqueeze(1)
        if self.cfg.n_fonts > 0:
            ctx = torch.cat([ctx, self.font_embedding(font).unsqueeze(1)], 1)
        h = self.positional_encoding.unsqueeze(0).expand(B, -1, -1)
        for blk in self.layers:
            a, _ = blk.attn(blk.ln1(h), ctx, ctx)                     # (out, weights) as the reference calls it, model.py:176
            h = h + a
            h = h + blk.fc2(F.relu(blk.fc1(blk.ln2(h))))
        u = self.fc_output(self.ln_f(h)).squeeze(-1)
        return torch.clamp(u, 0, 1).view(B, self.cfg.out_h, self.cfg.out_w)


def _summary(fx, key, a, idx_seed):
    """Full tensor when small, else row sums, column sums and 2048 hashed samples (as sheet_r0's fc_output.weight.grad)."""
    a = np.asarray(a)
    if a.size <= 70000:
        fx[key] = a
        return
    a2 = a.reshape(a.shape[0], -1)
    idx = (synth._counter(idx_seed, 2048, 42) % np.uint64(a.size)).astype(np.int64)
    fx[key + "/rowsum"], fx[key + "/colsum"], fx[key + "/idx"], fx[key + "/samples"] = a2.sum(1), a2.sum(0), idx, a.reshape(-1)[idx]


def pixel_twin():
    """C5-mini (8x8 = 64 pixel tokens, d_model 512, 8 heads, 4 layers, ff 2048, batch 32): eval bitmaps, loss, every gradient
    and a 3-step AdamW trajectory of the torch.nn twin -- what oracle.pixel_forward / pixel_backward are pinned to."""
    from ai_font_renderer_amd.config import C5_MINI as cfg
    B = 32
    i = np.arange(B)
    x = (32 + (i * 7) % 95).astype(np.int64)
    font = ((i // 3) % cfg.n_fonts).astype(np.int64)
    tu8 = synth.hash_u8(960, (B, cfg.out_h, cfg.out_w))
    xt, ft, tgt = torch.from_numpy(x), torch.from_numpy(font), torch.from_numpy(tu8.astype(np.float32) / 255.0)
    tw = PixelTwin(cfg)
    tw.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()})
    assert [k for k, _ in tw.named_parameters()] == [k for k, _ in cfg.param_shapes()]     # state_dict order == the config's
    fx = dict(x=x, font=font, target_u8=tu8)
    tw.eval()
    with torch.no_grad():
        fx["eval_y"] = tw(xt, ft).numpy()
    tw.train()
    # (the reference's rate, 1e-3, saturates this 12.7 M-parameter net's clamp in ONE step -- loss 0.283 -> 0.3348 and constant
    # from then on, a dead trajectory that pins nothing; at 1e-4 the loss still rises; 1e-5 descends: 0.283 -> 0.266 -> 0.242)
    lr = 0.01 * ref.LEARNING_RATE
    fx["lr"] = np.float64(lr)
    opt = torch.optim.AdamW(tw.parameters(), lr=lr, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = F.mse_loss(tw(xt, ft), tgt)
        loss.backward()
        if step == 0:
            for n_, (k, p) in enumerate(tw.named_parameters()):
                _summary(fx, "grad/" + k, p.grad.detach().numpy(), 7000 + n_)
        opt.step()
        losses.append(loss.item())
    fx["losses"] = np.array(losses, dtype=np.float32)
    for n_, (k, p) in enumerate(tw.named_parameters()):
        _summary(fx, "param3/" + k, p.detach().numpy(), 7000 + n_)
    assert len(set(losses)) == 3 and losses[2] < losses[0], losses
    np.savez_compressed(os.path.join(OUT, "pixel_twin.npz"), **fx)
    print("pixel_twin.npz losses", losses, "clamped-inside fraction", float(((fx["eval_y"] > 0) & (fx["eval_y"] < 1)).mean()))


def pixel_twin_full():
    """The FULL-SIZE configuration of BASELINE configs[4] (64x64 = 4096 pixel tokens, d_model 512, 8 heads, 4 blocks, ff 2048) on
    three glyphs: eval bitmaps, training loss and every gradient of the torch.nn twin (small tensors in full, the large ones by
    row sums, column sums and 2048 samples).  Pins what the miniature cannot reach: 4096 tokens per glyph, i.e. 16 chunks of the
    attention backward's key/value sums, token counts beyond one tile of every product."""
    from ai_font_renderer_amd.config import C5 as cfg
    B = 3
    x = np.array([40, 77, 105], dtype=np.int64)
    font = np.array([0, 1, 1], dtype=np.int64)
    tu8 = synth.hash_u8(961, (B, cfg.out_h, cfg.out_w))
    xt, ft, tgt = torch.from_numpy(x), torch.from_numpy(font), torch.from_numpy(tu8.astype(np.float32) / 255.0)
    tw = PixelTwin(cfg)
    tw.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()})
    fx = dict(x=x, font=font, target_u8=tu8)
    tw.eval()
    with torch.no_grad():
        fx["eval_y"] = tw(xt, ft).numpy()
    tw.train()
    loss = F.mse_loss(tw(xt, ft), tgt)
    loss.backward()
    for n_, (k, p) in enumerate(tw.named_parameters()):
        _summary(fx, "grad/" + k, p.grad.detach().numpy(), 7100 + n_)
    fx["loss"] = np.float32(loss.item())
    np.savez_compressed(os.path.join(OUT, "pixel_twin_full.npz"), **fx)
    print("pixel_twin_full.npz loss", loss.item(), "clamped-inside fraction", float(((fx["eval_y"] > 0) & (fx["eval_y"] < 1)).mean()))


def glyph_bitmaps():
    from ai_font_renderer_amd import datagen
    fonts = ["/root/reference/FiraCode-Retina.ttf", "/root/reference/Montserrat-Regular.ttf"]
    g16 = datagen.render_glyphs(fonts[:1], 16)            # [1, 95, 16, 16]
    g32 = datagen.render_glyphs(fonts, 32)                # [2, 95, 32, 32]
    np.savez_compressed(os.path.join(OUT, "glyph_bitmaps.npz"), fira16=g16[0], fira_mont32=g32)
    print("glyph_bitmaps.npz", g16.shape, g32.shape, "dark fraction", float((g32 < 128).mean()))


def train_loop():
    """Three runs of the reference's training driver.  "a" (LR 0.004) learns steadily for 16 epochs and never changes its
    rate; "b" (LR 0.03) saturates into a dead clamp after one epoch (constant validation loss), so the plateau scheduler
    cuts the rate and early stopping ends the run: both pin the bookkeeping.  "c" (LR 0.016, scheduler patience 0, early-stop
    patience 6, 10 epochs) is the one in which ReduceLROnPlateau fires WHILE THE LOSS STILL MOVES: the validation loss falls
    from 0.35 to 0.085 with two rate cuts on the way (0.016 -> 0.0112 after epoch 4, -> 0.00784 after epoch 8), each followed
    by epochs of further improvement (0.203 -> 0.134 -> 0.115 -> 0.093; 0.097 -> 0.087 -> 0.085), so gradients scaled by a
    freshly cut rate enter the pinned trajectory.  (Ten epochs: beyond that the two f32 implementations' rounding
    differences, amplified by the oscillating stretch at this rate, decide a later plateau comparison differently.)"""
    fx = {}
    for tag, lr, sp, ep, ne in (("a", 0.004, 1, 4, 16), ("b", 0.03, 1, 4, 16), ("c", 0.016, 0, 6, 10)):
        for k, v in _train_loop_run(lr, sp, ep, ne).items():
            if tag == "b" and k.startswith("final/"):
                continue                                    # a saturated run's parameters say nothing; its bookkeeping does
            fx[f"{tag}/{k}"] = v
        fx[f"{tag}/patience"] = np.array([sp, ep, ne], dtype=np.int64)      # scheduler patience, early-stop patience, NUM_EPOCHS
    v, l = fx["c/val_losses"], fx["c/lrs"]
    cuts = [i for i in range(1, len(l)) if l[i] < l[i - 1]]
    assert len(cuts) >= 2 and all(abs(v[i + 1] / v[i] - 1) > 1e-2 for i in cuts if i + 1 < len(v)), "run c must cut the rate while the loss moves"
    np.savez_compressed(os.path.join(OUT, "train_loop.npz"), **fx)


def _train_loop_run(lr_, sched_patience=1, stop_patience=4, epochs=16):
    """The reference's training driver itself (model.py:209-384): data split, loaders, AdamW, ReduceLROnPlateau, early
    stopping, artefacts -- on a dataset small enough to run here, with its own constants turned down."""
    import contextlib
    cfg = SheetConfig(max_length=10, sheet_h=8, sheet_w=24)
    N, BS = 80, 16
    saved = {k: getattr(ref, k) for k in ("NUM_EPOCHS", "LEARNING_RATE", "SCHEDULER_PATIENCE", "EARLY_STOPPING_PATIENCE", "OUTPUT_DIR",
                                          "SHEET_HEIGHT", "SHEET_WIDTH", "MAX_CHARS_PER_SHEET")}
    tmp = tempfile.mkdtemp()
    try:
        ref.NUM_EPOCHS, ref.LEARNING_RATE, ref.SCHEDULER_PATIENCE, ref.EARLY_STOPPING_PATIENCE = epochs, lr_, sched_patience, stop_patience
        ref.OUTPUT_DIR, ref.MAX_CHARS_PER_SHEET = os.path.join(tmp, "out"), 10
        m = build_ref(cfg)                                       # also sets ref.SHEET_HEIGHT / WIDTH
        m.embedding_dropout.p = 0.0
        m.dropout1.p = 0.0
        m.attention.dropout = 0.0
        x = torch.from_numpy(synth.encode_strings(synth.dataset_strings(N), 10))
        # a LEARNABLE synthetic target (hash noise would saturate at its variance within an epoch): the pixel at (h, w) shows
        # the code under it -- character (10 w) // 24 of the string -- as a grey level, shifted by the row
        xs = x.numpy()
        tu8 = ((xs[:, (np.arange(24) * 10) // 24][:, None, :] * 7 + np.arange(8)[None, :, None] * 16) % 256).astype(np.uint8)
        ds = torch.utils.data.TensorDataset(x, torch.from_numpy(tu8.astype(np.float32) / 255.0))
        log = []
        Sched = torch.optim.lr_scheduler.ReduceLROnPlateau
        orig_step = Sched.step

        def step(self, metrics, *a, **k):
            r = orig_step(self, metrics, *a, **k)
            log.append((float(metrics), float(self.optimizer.param_groups[0]["lr"])))
            return r

        Sched.step = step
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ref.train_attention_model(m, ds, BS)
        finally:
            Sched.step = orig_step
        printed = {}
        for line in buf.getvalue().splitlines():
            if line.startswith("Epoch ") and "Train Loss:" in line:
                ep = int(line.split(",")[0].split()[1])
                printed[ep] = float(line.split("Train Loss:")[1].split(",")[0])
        results = open(os.path.join(ref.OUTPUT_DIR, "training_results.txt")).read().splitlines()
        results = [ln for ln in results if not ln.startswith("training_completed")]
        fx = dict(x=x.numpy(), target_u8=tu8, val_losses=np.array([v for v, _ in log], dtype=np.float32),
                  lrs=np.array([lr for _, lr in log], dtype=np.float64),
                  printed_epochs=np.array(sorted(printed), dtype=np.int64),
                  printed_train_losses=np.array([printed[e] for e in sorted(printed)], dtype=np.float32),
                  results=np.array("\n".join(results)), stdout=np.array(buf.getvalue()))
        for k, v in m.state_dict().items():
            fx["final/" + k] = v.detach().numpy()
        print("train_loop run lr", lr_, "epochs", len(log), "val", [round(v, 5) for v, _ in log], "lr", [lr for _, lr in log])
        print("\n".join(results))
        return fx
    finally:
        for k, v in saved.items():
            setattr(ref, k, v)


def cpu_step_times():
    """BASELINE.md 3 / bench.py cpu_baseline provenance: the oracle's train step (what bench.py times on the GPU box's host
    cores as "kind": "port") against the REFERENCE's own step (model.train(); forward; F.mse_loss; backward; AdamW.step --
    model.py:291-311) on the same shapes, in this container.  Writes tests/golden/cpu_step_times.json."""
    import json
    import time
    sys.path.insert(0, os.path.join(ROOT))
    from oracle import afr_oracle as oracle
    out = {"torch": torch.__version__, "threads": torch.get_num_threads(), "cases": []}
    for B in (64, 256):
        cfg = WORKLOADS["r0"]["cfg"]
        strings = synth.dataset_strings(B)
        x = torch.from_numpy(synth.encode_strings(strings, cfg.max_length))
        tgt = torch.from_numpy(synth.synth_sheet_targets(B, cfg.sheet_h, cfg.sheet_w, tensor_id=955).astype(np.float32) / 255.0)
        m = build_ref(cfg)
        m.train()
        opt = torch.optim.AdamW(m.parameters(), lr=ref.LEARNING_RATE, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))

        def ref_step():
            opt.zero_grad()
            o = m(x)
            loss = F.mse_loss(o, tgt.view(o.shape))
            loss.backward()
            opt.step()

        def timed(fn, n):
            fn()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            return float(np.min(ts))                  # the container is shared: the fastest of n is the least disturbed
        t_ref = timed(ref_step, 7)
        m.embedding_dropout.p = 0.0; m.dropout1.p = 0.0; m.attention.dropout = 0.0
        t_ref_nodrop = timed(ref_step, 7)
        del m, opt
        P = {k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()}
        M = {k: torch.zeros_like(v) for k, v in P.items()}
        V = {k: torch.zeros_like(v) for k, v in P.items()}
        masks = {k: torch.from_numpy(v) for k, v in synth.sheet_dropout_masks(cfg, B, cfg.max_length, 42, 1).items()}
        state = [P, M, V, 0]

        def ora_step():
            state[3] += 1
            _, _, state[0], state[1], state[2] = oracle.train_step(state[0], state[1], state[2], state[3], x, tgt, cfg, masks=masks, inplace=True)
        t_ora = timed(ora_step, 7)
        out["cases"].append({"workload": "r0", "batch": B, "reference_step_ms": t_ref * 1e3, "reference_step_no_dropout_ms": t_ref_nodrop * 1e3,
                             "oracle_step_ms": t_ora * 1e3, "oracle_over_reference": t_ora / t_ref})
        print(out["cases"][-1])
    # C3 (no class in the reference): the torch.nn twin above stands for "the reference's way of writing it" (nn.Embedding,
    # nn.Linear, F.mse_loss, autograd, optim.AdamW)
    cfg = WORKLOADS["c3"]["cfg"]
    B = WORKLOADS["c3"]["batch"]
    xg, fg, tg = glyph_inputs(cfg, B)
    xg, fg = torch.from_numpy(xg), torch.from_numpy(fg)
    tgt = torch.from_numpy(tg.astype(np.float32) / 255.0)
    tw = GlyphTwin(cfg)
    tw.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()})
    opt = torch.optim.AdamW(tw.parameters(), lr=ref.LEARNING_RATE, weight_decay=ref.WEIGHT_DECAY, betas=(0.9, 0.99))

    def twin_step():
        opt.zero_grad()
        o = tw(xg, fg)
        F.mse_loss(o, tgt.view(o.shape)).backward()
        opt.step()

    def timed2(fn, n):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.min(ts))
    t_tw = timed2(twin_step, 9)
    P = {k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    st = [0]

    def ora3():
        st[0] += 1
        oracle.train_step(P, M, V, st[0], xg, tgt.view(B, -1), cfg, font=fg, inplace=True)
    t_o3 = timed2(ora3, 9)
    out["cases"].append({"workload": "c3", "batch": B, "reference_step_ms": None, "torch_nn_twin_step_ms": t_tw * 1e3,
                         "oracle_step_ms": t_o3 * 1e3, "oracle_over_twin": t_o3 / t_tw})
    print(out["cases"][-1])
    json.dump(out, open(os.path.join(OUT, "cpu_step_times.json"), "w"), indent=1)


def bmp24_topdown(rgb):
    """24-bit BGR top-down BMP bytes in the layout generate_font.ts:6-62 writes."""
    import struct
    h, w, _ = rgb.shape
    row = (w * 3 + 3) // 4 * 4
    data = bytearray()
    for y in range(h):
        line = bytearray(rgb[y, :, ::-1].tobytes())
        line += b"\0" * (row - w * 3)
        data += line
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54)
    dib = struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 24, 0, len(data), 0, 0, 0, 0)
    return hdr + dib + bytes(data)


def helpers_fx():
    a = synth.hash_u01(903, 80 * 240).astype(np.float32).reshape(80, 240)
    a[0, :5] = [0.0, 1.0, 0.999, 0.5, 254.5 / 255.0]
    img = ref_helpers.binary_array_to_image(a)                    # helpers.py:20-44
    u8 = np.array(img)
    gray = synth.hash_u8(904, (80, 240))
    rgb = np.stack([gray, gray, gray], axis=-1)
    rgb2 = synth.hash_u8(905, (80, 240, 3))                       # coloured pixels go through PIL 'L' luma
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "1.bmp"), os.path.join(d, "2.bmp")
        open(p1, "wb").write(bmp24_topdown(rgb))
        open(p2, "wb").write(bmp24_topdown(rgb2))
        f1 = ref_helpers.image_to_binary_array(p1)                 # helpers.py:107-123
        f2 = ref_helpers.image_to_binary_array(p2)
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), arr=a, arr_u8=u8, gray=gray, gray_f32=f1, rgb=rgb2, rgb_f32=f2)
    print("helpers.npz")


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (glyph_ref1, glyph_twin, pixel_twin, pixel_twin_full, glyph_bitmaps, train_loop, mini, helpers_fx, r0):
        if not only or fn.__name__ in only:
            fn()
    if "cpu_step_times" in only:              # timing, not a parity fixture: only on request
        cpu_step_times()
