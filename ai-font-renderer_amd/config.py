"""Model-shape descriptions for the two families the hot path serves.

SheetConfig  -- the reference's AttentionFontRenderer (model.py:129-204); defaults are its module
                constants (model.py:64-66,79-81,148-149).  "R0" in SURVEY.md.
GlyphConfig  -- BASELINE.json's per-glyph MLP configs C1-C4 (embedding [+font embedding] ->
                Linear/ReLU stack -> Linear -> clamp), built from the same layer idioms
                (model.py:136,148,152-156; learnings.md:3).
Both list their parameters in state_dict order: that order is the checkpoint contract and the
layout of the flat parameter/gradient/moment buffers the C ABI works on.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

ALIGN_ELEMS = 64  # every tensor starts on a 256-byte boundary inside the flat fp32 buffers


@dataclass(frozen=True)
class SheetConfig:
    max_length: int = 100
    embed_dim: int = 32
    heads: int = 4
    fc_dim: int = 64
    sheet_h: int = 80
    sheet_w: int = 240
    vocab: int = 128
    p_embed: float = 0.2
    p_attn: float = 0.2
    p_fc: float = 0.25
    ln_eps: float = 1e-5
    kind: str = "sheet"

    @property
    def pixels(self):
        return self.sheet_h * self.sheet_w

    @property
    def flat_dim(self):
        return self.max_length * self.fc_dim

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        E, F = self.embed_dim, self.fc_dim
        return [
            ("positional_encoding", (self.max_length, E)),
            ("embedding.weight", (self.vocab, E)),
            ("attention.in_proj_weight", (3 * E, E)),
            ("attention.in_proj_bias", (3 * E,)),
            ("attention.out_proj.weight", (E, E)),
            ("attention.out_proj.bias", (E,)),
            ("layer_norm.weight", (E,)),
            ("layer_norm.bias", (E,)),
            ("fc1.weight", (F, E)),
            ("fc1.bias", (F,)),
            ("fc_output.weight", (self.pixels, self.flat_dim)),
            ("fc_output.bias", (self.pixels,)),
        ]


@dataclass(frozen=True)
class GlyphConfig:
    hidden: Tuple[int, ...] = (256,)
    out_h: int = 16
    out_w: int = 16
    embed_dim: int = 32
    vocab: int = 128
    n_fonts: int = 0
    kind: str = "glyph"

    @property
    def pixels(self):
        return self.out_h * self.out_w

    def layer_dims(self) -> List[Tuple[int, int]]:
        dims, k = [], self.embed_dim
        for h in self.hidden:
            dims.append((h, k))
            k = h
        dims.append((self.pixels, k))
        return dims

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = [("embedding.weight", (self.vocab, self.embed_dim))]
        if self.n_fonts > 0:
            out.append(("font_embedding.weight", (self.n_fonts, self.embed_dim)))
        dims = self.layer_dims()
        for i, (n, k) in enumerate(dims[:-1]):
            out += [(f"fc{i + 1}.weight", (n, k)), (f"fc{i + 1}.bias", (n,))]
        n, k = dims[-1]
        out += [("fc_output.weight", (n, k)), ("fc_output.bias", (n,))]
        return out


@dataclass(frozen=True)
class PixelConfig:
    """BASELINE.json configs[4] ("64x64 bitmap, 4-layer transformer (d_model=512, per-pixel tokens), fp8"): no class in the
    reference.  Definition fixed in DESIGN.md 8: one token per output pixel, initialised from a learned
    positional table (the reference's positional_encoding idiom, model.py:140-141); every block is pre-LayerNorm
    cross-attention of the pixel tokens (queries) to the glyph's CONTEXT tokens [Emb[char], FontEmb[font]] (keys/values;
    nn.MultiheadAttention with the packed in-projection of model.py:144) followed by a Linear-ReLU-Linear MLP (model.py:148
    idiom), both residual; a final LayerNorm and a Linear d_model -> 1 per token give the pixel, clamped to [0,1]
    (model.py:152-156).  Full self-attention over 4096 pixel tokens would cost 8x the FLOPs of everything else together."""
    out_h: int = 64
    out_w: int = 64
    d_model: int = 512
    heads: int = 8
    layers: int = 4
    ff_dim: int = 2048
    vocab: int = 128
    n_fonts: int = 2
    ln_eps: float = 1e-5
    kind: str = "pixel"

    @property
    def pixels(self):
        return self.out_h * self.out_w

    @property
    def tokens(self):
        return self.out_h * self.out_w

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        d, f = self.d_model, self.ff_dim
        out = [("positional_encoding", (self.tokens, d)), ("embedding.weight", (self.vocab, d))]      # a module's own parameters come first
        if self.n_fonts > 0:
            out.append(("font_embedding.weight", (self.n_fonts, d)))
        for l in range(self.layers):
            q = f"layers.{l}."
            out += [(q + "ln1.weight", (d,)), (q + "ln1.bias", (d,)),
                    (q + "attn.in_proj_weight", (3 * d, d)), (q + "attn.in_proj_bias", (3 * d,)),
                    (q + "attn.out_proj.weight", (d, d)), (q + "attn.out_proj.bias", (d,)),
                    (q + "ln2.weight", (d,)), (q + "ln2.bias", (d,)),
                    (q + "fc1.weight", (f, d)), (q + "fc1.bias", (f,)), (q + "fc2.weight", (d, f)), (q + "fc2.bias", (d,))]
        out += [("ln_f.weight", (d,)), ("ln_f.bias", (d,)), ("fc_output.weight", (1, d)), ("fc_output.bias", (1,))]
        return out

    def train_flops_per_sample(self):
        """3 x forward GEMM FLOPs: per token and layer q-proj + out-proj (2 d^2 each) and the MLP (4 d ff); the context side
        (k/v projections of 2 tokens, 2-key attention) and the d -> 1 head are below 1 % and left out."""
        return 3.0 * self.tokens * self.layers * (2.0 * 2 * self.d_model ** 2 + 4.0 * self.d_model * self.ff_dim)


def flat_layout(cfg):
    """[(name, shape, offset, numel)], total: offsets in elements, each a multiple of ALIGN_ELEMS."""
    table, off = [], 0
    for name, shape in cfg.param_shapes():
        n = 1
        for s in shape:
            n *= s
        table.append((name, shape, off, n))
        off += (n + ALIGN_ELEMS - 1) // ALIGN_ELEMS * ALIGN_ELEMS
    return table, off


# Named workloads.  batch is per GPU.
WORKLOADS = {
    "r0": dict(cfg=SheetConfig(), batch=1024),
    "c1": dict(cfg=GlyphConfig(hidden=(256,), out_h=16, out_w=16), batch=95),
    "c2": dict(cfg=GlyphConfig(hidden=(256,), out_h=16, out_w=16), batch=4096),
    "c3": dict(cfg=GlyphConfig(hidden=(1024, 1024), out_h=32, out_w=32, n_fonts=2), batch=8192),
}
# BASELINE configs[4] and the miniature the parity fixtures pin (tests/golden/pixel_twin.npz): DESIGN.md 8
C5 = PixelConfig()
C5_MINI = PixelConfig(out_h=8, out_w=8)
# one micro-batch of configs[4] per step: 32 glyphs = 131072 pixel tokens, 12.6 GB of saved activations in bf16 mode (the configured
# 2048 glyphs per GPU are 64 such micro-batches; every product already has M = 131072 rows)
WORKLOADS["c5"] = dict(cfg=C5, batch=32)
