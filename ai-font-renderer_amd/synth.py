"""Portable synthetic data: the same integers on every box, no torch RNG involved.

* text generator  -- integer-exact restatement of the reference's seeded LCG text source
                     (reference generate_font.ts:164-199, sample i uses seed i+42, :204)
* weights/targets -- counter-based splitmix64 hash of (tensor_id << 40) + flat_index
                     (SURVEY.md App. D); weights are never stored in fixtures
* dropout masks   -- 32-bit counter hash shared bit-for-bit with the HIP kernels
                     (csrc/afr_common.h: afr_hash32 / afr_keep); replaces the reference's
                     torch `bernoulli_` stream (model.py:137,144,149), which no GPU can replay
"""
import numpy as np

SEED = 42
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _counter(tensor_id, n, seed, lane=0):
    base = (np.uint64(tensor_id) << np.uint64(40)) ^ (np.uint64(seed) << np.uint64(24)) ^ (np.uint64(lane) << np.uint64(60))
    with np.errstate(over="ignore"):
        return splitmix64(base + np.arange(n, dtype=np.uint64))


def hash_u01(tensor_id, n, seed=SEED, lane=0):
    """n doubles in [0,1): top 53 bits of the hash."""
    return (_counter(tensor_id, n, seed, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_uniform(tensor_id, shape, bound, seed=SEED):
    n = int(np.prod(shape))
    return ((2.0 * hash_u01(tensor_id, n, seed) - 1.0) * bound).astype(np.float32).reshape(shape)


def hash_normal(tensor_id, shape, std, seed=SEED):
    n = int(np.prod(shape))
    u1 = hash_u01(tensor_id, n, seed, lane=0)
    u2 = hash_u01(tensor_id, n, seed, lane=1)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def hash_u8(tensor_id, shape, seed=SEED):
    n = int(np.prod(shape))
    return (_counter(tensor_id, n, seed) >> np.uint64(56)).astype(np.uint8).reshape(shape)


# ---------------------------------------------------------------- text (generate_font.ts:164-199)
def lcg_text(seed, min_length=10, max_length=100):
    state = int(seed)

    def rnd():
        nonlocal state
        state = (state * 1664525 + 1013904223) % 4294967296
        return state / 4294967296

    length = int(rnd() * (max_length - min_length + 1)) + min_length
    out = []
    remaining = length
    while remaining > 0:
        wl = min(int(rnd() * 10) + 1, remaining)
        out.append("".join(chr(65 + int(rnd() * 26)) for _ in range(wl)))
        remaining -= wl
        if remaining > 0:
            out.append(" ")
            remaining -= 1
    return "".join(out)


def dataset_strings(n, first_seed=42):
    """Sample i (0-based) of the reference dataset uses seed i+42 (generate_font.ts:204)."""
    return [lcg_text(first_seed + i) for i in range(n)]


def encode_strings(strings, max_length):
    """ord() codes, truncated to max_length and zero padded (helpers.py:52-59,164-173)."""
    x = np.zeros((len(strings), max_length), dtype=np.int64)
    for i, s in enumerate(strings):
        codes = [ord(c) for c in s[:max_length]]
        x[i, :len(codes)] = codes
    return x


def synth_sheet_targets(n, height, width, tensor_id=900, seed=SEED, dark_fraction=0.12):
    """uint8 [n,h,w]: white (255) background, ~12 % hashed anti-aliased dark pixels."""
    cnt = n * height * width
    u = hash_u01(tensor_id, cnt, seed, lane=0)
    shade = hash_u8(tensor_id, (cnt,), seed ^ 0x5A5A)
    t = np.where(u < dark_fraction, shade, np.uint8(255)).astype(np.uint8)
    return t.reshape(n, height, width)


# ---------------------------------------------------------------- dropout counter hash (device twin)
STREAM_EMBED, STREAM_ATTN, STREAM_FC = 1, 2, 3


def dropout_key(seed, step, stream, rank=0):
    """32-bit key for one (seed, step, tensor stream, rank); host side of afr_keep()."""
    v = (int(seed) & 0xFFFFFFFFFFFFFFFF) ^ ((int(step) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    v ^= (int(stream) * 0xC2B2AE3D27D4EB4F) & 0xFFFFFFFFFFFFFFFF
    v ^= (int(rank) * 0x165667B19E3779F9) & 0xFFFFFFFFFFFFFFFF
    return int(splitmix64(np.array([v], dtype=np.uint64))[0]) & 0xFFFFFFFF


def keep_threshold(keep_prob):
    return int(float(keep_prob) * 16777216.0)


def dropout_keep_mask(key, n, keep_prob, offset=0):
    """Boolean keep mask for flat element indices offset..offset+n-1 (bit-exact twin of afr_keep)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        k = np.uint32(key) + hi * np.uint32(0x85EBCA6B)
        h = lo * np.uint32(0x9E3779B1) + k
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x21F0AAAD)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x735A2D97)
        h ^= h >> np.uint32(15)
    return (h >> np.uint32(8)) < np.uint32(keep_threshold(keep_prob))


# ---------------------------------------------------------------- formula-generated parameters
def make_params(cfg, seed=SEED):
    """dict name -> float32 array for every tensor of cfg.param_shapes(), from the counter hash.
    Ranges follow the reference's default initialisers but biases / LayerNorm
    affine are made non-trivial so that parity tests exercise them."""
    out = {}
    for tid, (name, shape) in enumerate(cfg.param_shapes()):
        if name == "positional_encoding":
            a = hash_normal(tid, shape, 0.02, seed)                 # model.py:141
        elif name in ("embedding.weight", "font_embedding.weight"):
            a = hash_normal(tid, shape, 1.0, seed)                  # nn.Embedding default N(0,1)
        elif name.endswith("in_proj_weight"):
            a = hash_uniform(tid, shape, float(np.sqrt(6.0 / (shape[0] + shape[1]))), seed)
        elif name == "layer_norm.weight" or (len(shape) == 1 and name.split(".")[-2][:2] == "ln" and name.endswith(".weight")):
            a = 1.0 + hash_uniform(tid, shape, 0.1, seed)
        elif name == "layer_norm.bias" or (len(shape) == 1 and name.split(".")[-2][:2] == "ln" and name.endswith(".bias")):
            a = hash_uniform(tid, shape, 0.1, seed)
        elif name.endswith(".weight"):
            a = hash_uniform(tid, shape, float(1.0 / np.sqrt(shape[1])), seed)   # kaiming_uniform(a=sqrt5)
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            a = hash_uniform(tid, shape, 0.05, seed)
        else:  # Linear biases: U(+-1/sqrt(fan_in)); fan_in is the matching weight's 2nd dim
            wshape = dict(cfg.param_shapes())[name[:-4] + "weight"]
            a = hash_uniform(tid, shape, float(1.0 / np.sqrt(wshape[1])), seed)
        out[name] = a.astype(np.float32)
    return out


def sheet_dropout_masks(cfg, B, L, seed, step, rank=0):
    """The three keep masks (uint8 0/1) exactly as the HIP kernels derive them."""
    E, H, F = cfg.embed_dim, cfg.heads, cfg.fc_dim
    me = dropout_keep_mask(dropout_key(seed, step, STREAM_EMBED, rank), B * L * E, 1.0 - cfg.p_embed)
    ma = dropout_keep_mask(dropout_key(seed, step, STREAM_ATTN, rank), B * H * L * L, 1.0 - cfg.p_attn)
    mf = dropout_keep_mask(dropout_key(seed, step, STREAM_FC, rank), B * L * F, 1.0 - cfg.p_fc)
    return dict(embed=me.reshape(B, L, E).astype(np.uint8), attn=ma.reshape(B, H, L, L).astype(np.uint8),
                fc=mf.reshape(B, L, F).astype(np.uint8))


# ---------------------------------------------------------------- rasterised glyph targets (BASELINE C1-C4)
_GLYPH_FIXTURE = None


def glyph_bitmap_targets(size, x, font=None):
    """uint8 [B, size, size] targets for codes x (32..126) and font ids: the FreeType rasterisations of FiraCode-Retina
    (16x16; font 0 of 32x32) and Montserrat-Regular (font 1 of 32x32) that tests/golden/make_golden.py drew with
    datagen.render_glyphs from the reference's two TTFs.  None when the fixture file is not there or holds no such size."""
    global _GLYPH_FIXTURE
    import os
    if _GLYPH_FIXTURE is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "glyph_bitmaps.npz")
        _GLYPH_FIXTURE = dict(np.load(path)) if os.path.exists(path) else {}
    x = np.asarray(x, dtype=np.int64)
    if size == 16 and "fira16" in _GLYPH_FIXTURE:
        return _GLYPH_FIXTURE["fira16"][x - 32]
    if size == 32 and "fira_mont32" in _GLYPH_FIXTURE:
        f = np.zeros_like(x) if font is None else np.asarray(font, dtype=np.int64)
        return _GLYPH_FIXTURE["fira_mont32"][f, x - 32]
    return None
