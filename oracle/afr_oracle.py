"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (ai-font-renderer_amd/) never does and fails loudly when libafr.so is missing.

What it is: a first-principles restatement, in explicit tensor algebra (no torch.nn modules, no
autograd), of the arithmetic the reference's training step performs.  The reference's arithmetic
lives in its third-party dependency `torch` (requirements.txt:2, unpinned `torch>=1.7.0`; this
image has 2.10.0+rocm7.0); the call sites restated here are cited per function.  PARITY PINNING:
the reference has no tests or golden vectors of its own, so this restatement is pinned against
outputs of the reference itself, imported in the build container by tests/golden/make_golden.py
(fixtures committed under tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Every function works in the dtype of the tensors it is given (float32 for parity runs, float64
for tight self-checks).  Dropout masks are INPUTS (0/1 tensors); eval mode = masks None.
"""
import math
import torch


# --------------------------------------------------------------------------- sheet model (R0)
def bf16_round(t, site=None):
    """Round-to-nearest-even to bfloat16 and back (a rounding hook that rounds at EVERY site)."""
    return t.to(torch.bfloat16).to(t.dtype)


def _r(rnd, t, site):
    return t if rnd is None else rnd(t, site)


# The three products of a Linear layer (nn.Linear forward, model.py:148,152,183,196, and what autograd derives from it,
# model.py:309), each with a rounding hook on every operand and on the result.  rnd=None -- the only mode the golden
# fixtures pin -- is the reference's plain f32 arithmetic.  A checker of a mixed-precision implementation passes
# rnd(tensor, site) -> tensor with site = "<layer>.<fwd|dw|dx>.<x|w|dy|y>"; WHICH sites an implementation rounds at is
# that checker's knowledge (tests/util.py), never the oracle's.
def linear_fwd(rnd, name, x, W, b):
    return _r(rnd, _r(rnd, x, name + ".fwd.x") @ _r(rnd, W, name + ".fwd.w").t() + b, name + ".fwd.y")


def linear_dw(rnd, name, dy, x):
    return _r(rnd, dy, name + ".dw.dy").t() @ _r(rnd, x, name + ".dw.x")


def linear_dx(rnd, name, dy, W):
    return _r(rnd, _r(rnd, dy, name + ".dx.dy") @ _r(rnd, W, name + ".dx.w"), name + ".dx.y")


def sheet_forward(P, x, cfg, masks=None, rnd=None, relu_mask=None):
    """AttentionFontRenderer.forward, reference model.py:158-204.

    P: dict of the 12 state_dict tensors (model.py:136-152).  x: int64 [B, Lin].
    masks: None (eval) or dict(embed=[B,L,E], attn=[B,H,L,L], fc=[B,L,F]) of {0,1}.
    relu_mask: optional bool [B,L,F] used INSTEAD of (pre > 0): gradients are discontinuous where a
    pre-activation is within rounding of 0, so a checker that wants tight tolerances takes the masks of the
    implementation under test and separately checks that they differ from its own only at |pre| ~ 0.
    Returns (y [B,h,w], cache) -- cache holds what sheet_backward needs.
    """
    B, Lin = x.shape
    L = min(Lin, cfg.max_length)                                  # model.py:163-164
    x = x[:, :L]
    E, H, F = cfg.embed_dim, cfg.heads, cfg.fc_dim
    D = E // H
    dt = P["embedding.weight"].dtype
    e0 = P["embedding.weight"][x]                                  # gather, model.py:167
    if masks is not None:                                          # dropout BEFORE pos-enc, :168
        se = 1.0 / (1.0 - cfg.p_embed)
        e1 = e0 * (masks["embed"].to(dt) * se)
    else:
        e1 = e0
    e = e1 + P["positional_encoding"][:L].unsqueeze(0)             # model.py:171-172
    # nn.MultiheadAttention math path (torch/nn/functional.py multi_head_attention_forward,
    # need_weights=True, batch_first=False): packed in-proj, q scaled by sqrt(1/D), softmax,
    # dropout on probabilities, out-proj.  model.py:144,175-177
    qkv = e @ P["attention.in_proj_weight"].t() + P["attention.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)

    def heads(t):
        return t.reshape(B, L, H, D).permute(0, 2, 1, 3)           # [B,H,L,D]

    qh, kh, vh = heads(q), heads(k), heads(v)
    scale = math.sqrt(1.0 / D)
    S = (qh * scale) @ kh.transpose(-1, -2)                        # [B,H,L,L]
    A = torch.softmax(S, dim=-1)
    if masks is not None:
        sa = 1.0 / (1.0 - cfg.p_attn)
        Ad = A * (masks["attn"].to(dt) * sa)
    else:
        Ad = A
    o = (Ad @ vh).permute(0, 2, 1, 3).reshape(B, L, E)
    a = o @ P["attention.out_proj.weight"].t() + P["attention.out_proj.bias"]
    r = e + a                                                      # residual, model.py:180
    mu = r.mean(-1, keepdim=True)
    var = ((r - mu) ** 2).mean(-1, keepdim=True)                   # biased variance
    rstd = torch.rsqrt(var + cfg.ln_eps)
    xhat = (r - mu) * rstd
    n = xhat * P["layer_norm.weight"] + P["layer_norm.bias"]
    pre = n @ P["fc1.weight"].t() + P["fc1.bias"]                  # model.py:183
    rmask = (pre > 0) if relu_mask is None else relu_mask
    f = pre * rmask.to(dt)
    if masks is not None:                                          # model.py:184
        sf = 1.0 / (1.0 - cfg.p_fc)
        fd = f * (masks["fc"].to(dt) * sf)
    else:
        fd = f
    z = fd.reshape(B, L * F)                                       # model.py:187
    if L < cfg.max_length:                                         # zero-pad branch, :190-193
        z = torch.cat([z, torch.zeros(B, (cfg.max_length - L) * F, dtype=dt)], dim=1)
    u = linear_fwd(rnd, "fc_output", z, P["fc_output.weight"], P["fc_output.bias"])   # model.py:196
    y = u.clamp(0.0, 1.0).reshape(B, cfg.sheet_h, cfg.sheet_w)     # model.py:199-202
    cache = dict(x=x, L=L, e=e, qh=qh, kh=kh, vh=vh, A=A, Ad=Ad, o=o, rstd=rstd, xhat=xhat,
                 n=n, pre=pre, rmask=rmask, z=z, u=u, masks=masks)
    return y, cache


def sheet_backward(P, cache, du, cfg, rnd=None):
    """Reverse of sheet_forward (what loss.backward() does, model.py:309); SURVEY.md App. A.

    du: gradient w.r.t. the pre-clamp output u [B, pixels] (clamp mask already applied).
    Returns dict of the 12 gradients.
    """
    c = cache
    B = du.shape[0]
    L, E, H, F = c["L"], cfg.embed_dim, cfg.heads, cfg.fc_dim
    D = E // H
    masks = c["masks"]
    G = {}
    G["fc_output.weight"] = linear_dw(rnd, "fc_output", du, c["z"])
    G["fc_output.bias"] = du.sum(0)
    dz = linear_dx(rnd, "fc_output", du, P["fc_output.weight"])
    dfd = dz[:, :L * F].reshape(B, L, F)
    df = dfd
    if masks is not None:
        df = df * (masks["fc"].to(du.dtype) * (1.0 / (1.0 - cfg.p_fc)))
    df = df * c["rmask"].to(du.dtype)                             # ReLU mask (threshold_backward)
    G["fc1.weight"] = df.reshape(-1, F).t() @ c["n"].reshape(-1, E)
    G["fc1.bias"] = df.reshape(-1, F).sum(0)
    dn = df @ P["fc1.weight"]
    G["layer_norm.weight"] = (dn * c["xhat"]).reshape(-1, E).sum(0)
    G["layer_norm.bias"] = dn.reshape(-1, E).sum(0)
    g = dn * P["layer_norm.weight"]
    dr = (g - g.mean(-1, keepdim=True) - c["xhat"] * (g * c["xhat"]).mean(-1, keepdim=True)) * c["rstd"]
    de = dr.clone()
    da = dr
    G["attention.out_proj.weight"] = da.reshape(-1, E).t() @ c["o"].reshape(-1, E)
    G["attention.out_proj.bias"] = da.reshape(-1, E).sum(0)
    do = (da @ P["attention.out_proj.weight"]).reshape(B, L, H, D).permute(0, 2, 1, 3)
    dAd = do @ c["vh"].transpose(-1, -2)
    dv = c["Ad"].transpose(-1, -2) @ do
    dA = dAd
    if masks is not None:
        dA = dA * (masks["attn"].to(du.dtype) * (1.0 / (1.0 - cfg.p_attn)))
    A = c["A"]
    dS = A * (dA - (dA * A).sum(-1, keepdim=True))                 # softmax backward
    scale = math.sqrt(1.0 / D)
    dq = (dS @ c["kh"]) * scale
    dk = dS.transpose(-1, -2) @ (c["qh"] * scale)

    def merge(t):
        return t.permute(0, 2, 1, 3).reshape(B, L, E)

    dqkv = torch.cat([merge(dq), merge(dk), merge(dv)], dim=-1)
    G["attention.in_proj_weight"] = dqkv.reshape(-1, 3 * E).t() @ c["e"].reshape(-1, E)
    G["attention.in_proj_bias"] = dqkv.reshape(-1, 3 * E).sum(0)
    de = de + dqkv @ P["attention.in_proj_weight"]
    dP = torch.zeros_like(P["positional_encoding"])
    dP[:L] = de.sum(0)
    G["positional_encoding"] = dP
    de0 = de
    if masks is not None:
        de0 = de0 * (masks["embed"].to(du.dtype) * (1.0 / (1.0 - cfg.p_embed)))
    dEmb = torch.zeros_like(P["embedding.weight"])
    dEmb.index_add_(0, c["x"].reshape(-1), de0.reshape(-1, E))     # embedding_dense_backward
    G["embedding.weight"] = dEmb
    return G


# --------------------------------------------------------------------------- glyph MLP (C1-C4)
def glyph_forward(P, x, font, cfg, rnd=None, relu_masks=None):
    """Per-glyph MLP built from the reference's layer idioms: Embedding gather (model.py:136,167)
    [+ font embedding], Linear+ReLU hidden layers (model.py:148,183), Linear + clamp output
    (model.py:152-156,196-202).  Ancestor: learnings.md:3.  x,font: int64 [B].
    Pinned by tests/golden/glyph_*.npz: the reference itself at max_length=1 (whose tail IS this network) and a
    torch.nn-composed twin for the shapes the reference class cannot express (font table, deeper stacks)."""
    h = P["embedding.weight"][x]
    if cfg.n_fonts > 0:
        h = h + P["font_embedding.weight"][font]
    acts = [h]
    nh = len(cfg.hidden)
    pres, rmasks = [], []
    for i in range(nh):
        pre = linear_fwd(rnd, f"fc{i + 1}", h, P[f"fc{i + 1}.weight"], P[f"fc{i + 1}.bias"])
        pres.append(pre)
        rmasks.append((pre > 0) if relu_masks is None else relu_masks[i])
        h = pre * rmasks[i].to(pre.dtype)
        acts.append(h)
    u = linear_fwd(rnd, "fc_output", h, P["fc_output.weight"], P["fc_output.bias"])
    y = u.clamp(0.0, 1.0).reshape(-1, cfg.out_h, cfg.out_w)
    return y, dict(x=x, font=font, acts=acts, pres=pres, rmasks=rmasks, u=u)


def glyph_backward(P, cache, du, cfg, rnd=None):
    G = {}
    acts = cache["acts"]
    nh = len(cfg.hidden)
    G["fc_output.weight"] = linear_dw(rnd, "fc_output", du, acts[nh])
    G["fc_output.bias"] = du.sum(0)
    d = linear_dx(rnd, "fc_output", du, P["fc_output.weight"])
    for i in reversed(range(nh)):
        d = d * cache["rmasks"][i].to(d.dtype)                     # ReLU mask (threshold_backward)
        G[f"fc{i + 1}.weight"] = linear_dw(rnd, f"fc{i + 1}", d, acts[i])
        G[f"fc{i + 1}.bias"] = d.sum(0)
        d = linear_dx(rnd, f"fc{i + 1}", d, P[f"fc{i + 1}.weight"])
    dEmb = torch.zeros_like(P["embedding.weight"])
    dEmb.index_add_(0, cache["x"], d)                              # embedding_dense_backward
    G["embedding.weight"] = dEmb
    if cfg.n_fonts > 0:
        dF = torch.zeros_like(P["font_embedding.weight"])
        dF.index_add_(0, cache["font"], d)
        G["font_embedding.weight"] = dF
    return G


# --------------------------------------------------------------------------- pixel-token transformer (C5)
def _layernorm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)                   # biased variance, as nn.LayerNorm
    rstd = torch.rsqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * g + b, xhat, rstd


def _layernorm_bwd(dy, xhat, rstd, g):
    """-> (dx, dgamma, dbeta): the LayerNorm backward of sheet_backward, for any leading shape."""
    E = dy.shape[-1]
    dg = (dy * xhat).reshape(-1, E).sum(0)
    db = dy.reshape(-1, E).sum(0)
    gg = dy * g
    dx = (gg - gg.mean(-1, keepdim=True) - xhat * (gg * xhat).mean(-1, keepdim=True)) * rstd
    return dx, dg, db


def pixel_forward(P, x, font, cfg):
    """BASELINE configs[4]'s model as DESIGN.md 8 defines it (config.PixelConfig): one token per output pixel; 4 pre-LayerNorm
    blocks of cross-attention (pixel queries -> the glyph's context tokens [Emb[char], FontEmb[font]]) and a
    Linear-ReLU-Linear MLP, both residual; LayerNorm + Linear d -> 1 per token; clamp.  The reference has no such class:
    every piece is one of its layer idioms -- Embedding gather model.py:136,167; learned positional table :140-141,171-172;
    nn.MultiheadAttention packed in-projection, q * sqrt(1/D), softmax, out-proj :144,175-177; LayerNorm :145,180; Linear +
    ReLU :148,183; Linear + clamp :152-156,196-202.  PARITY: unpinned by the reference; pinned to a torch.nn composition
    of exactly these modules (tests/golden/pixel_twin.npz, made by make_golden.py pixel_twin)."""
    B = x.shape[0]
    d, H, T = cfg.d_model, cfg.heads, cfg.tokens
    D = d // H
    ctx = P["embedding.weight"][x].unsqueeze(1)                                   # [B, 1, d]
    if cfg.n_fonts > 0:
        ctx = torch.cat([ctx, P["font_embedding.weight"][font].unsqueeze(1)], 1)  # [B, C, d]
    C = ctx.shape[1]
    h = P["positional_encoding"].unsqueeze(0).expand(B, T, d)
    scale = math.sqrt(1.0 / D)
    saved = []
    for l in range(cfg.layers):
        q_ = f"layers.{l}."
        Win, bin_ = P[q_ + "attn.in_proj_weight"], P[q_ + "attn.in_proj_bias"]
        n1, xh1, rs1 = _layernorm(h, P[q_ + "ln1.weight"], P[q_ + "ln1.bias"], cfg.ln_eps)
        q = n1 @ Win[:d].t() + bin_[:d]                                           # [B, T, d]
        k = ctx @ Win[d:2 * d].t() + bin_[d:2 * d]                                # [B, C, d]
        v = ctx @ Win[2 * d:].t() + bin_[2 * d:]
        qh = q.reshape(B, T, H, D).permute(0, 2, 1, 3)                            # [B, H, T, D]
        kh = k.reshape(B, C, H, D).permute(0, 2, 1, 3)
        vh = v.reshape(B, C, H, D).permute(0, 2, 1, 3)
        A = torch.softmax((qh * scale) @ kh.transpose(-1, -2), dim=-1)            # [B, H, T, C]
        o = (A @ vh).permute(0, 2, 1, 3).reshape(B, T, d)
        h1 = h + o @ P[q_ + "attn.out_proj.weight"].t() + P[q_ + "attn.out_proj.bias"]
        n2, xh2, rs2 = _layernorm(h1, P[q_ + "ln2.weight"], P[q_ + "ln2.bias"], cfg.ln_eps)
        pre = n2 @ P[q_ + "fc1.weight"].t() + P[q_ + "fc1.bias"]
        f = torch.relu(pre)
        h2 = h1 + f @ P[q_ + "fc2.weight"].t() + P[q_ + "fc2.bias"]
        saved.append(dict(n1=n1, xh1=xh1, rs1=rs1, qh=qh, kh=kh, vh=vh, A=A, o=o, n2=n2, xh2=xh2, rs2=rs2, pre=pre, f=f))
        h = h2
    nf, xhf, rsf = _layernorm(h, P["ln_f.weight"], P["ln_f.bias"], cfg.ln_eps)
    u = (nf @ P["fc_output.weight"].t() + P["fc_output.bias"]).squeeze(-1)        # [B, T]
    y = u.clamp(0.0, 1.0).reshape(B, cfg.out_h, cfg.out_w)
    return y, dict(x=x, font=font, ctx=ctx, saved=saved, nf=nf, xhf=xhf, rsf=rsf, u=u)


def pixel_backward(P, cache, du, cfg):
    """Reverse of pixel_forward (what loss.backward() does on the torch.nn twin); du [B, T]: gradient w.r.t. the pre-clamp
    output (clamp mask applied).  Returns the gradients of every parameter."""
    B = du.shape[0]
    d, H, T = cfg.d_model, cfg.heads, cfg.tokens
    D = d // H
    ctx = cache["ctx"]
    C = ctx.shape[1]
    scale = math.sqrt(1.0 / D)
    G = {}
    G["fc_output.weight"] = (du.reshape(-1, 1) * cache["nf"].reshape(-1, d)).sum(0, keepdim=True)
    G["fc_output.bias"] = du.sum().reshape(1)
    dnf = du.unsqueeze(-1) * P["fc_output.weight"].reshape(1, 1, d)
    dh, G["ln_f.weight"], G["ln_f.bias"] = _layernorm_bwd(dnf, cache["xhf"], cache["rsf"], P["ln_f.weight"])
    dctx = torch.zeros_like(ctx)
    for l in reversed(range(cfg.layers)):
        q_ = f"layers.{l}."
        c = cache["saved"][l]
        Win = P[q_ + "attn.in_proj_weight"]
        # MLP:  h2 = h1 + fc2(relu(fc1(LN2(h1))))
        G[q_ + "fc2.weight"] = dh.reshape(-1, d).t() @ c["f"].reshape(-1, cfg.ff_dim)
        G[q_ + "fc2.bias"] = dh.reshape(-1, d).sum(0)
        dpre = (dh @ P[q_ + "fc2.weight"]) * (c["pre"] > 0).to(dh.dtype)
        G[q_ + "fc1.weight"] = dpre.reshape(-1, cfg.ff_dim).t() @ c["n2"].reshape(-1, d)
        G[q_ + "fc1.bias"] = dpre.reshape(-1, cfg.ff_dim).sum(0)
        dx2, G[q_ + "ln2.weight"], G[q_ + "ln2.bias"] = _layernorm_bwd(dpre @ P[q_ + "fc1.weight"], c["xh2"], c["rs2"], P[q_ + "ln2.weight"])
        dh1 = dh + dx2
        # attention:  h1 = h + out_proj(softmax(q k^T) v)
        G[q_ + "attn.out_proj.weight"] = dh1.reshape(-1, d).t() @ c["o"].reshape(-1, d)
        G[q_ + "attn.out_proj.bias"] = dh1.reshape(-1, d).sum(0)
        do = (dh1 @ P[q_ + "attn.out_proj.weight"]).reshape(B, T, H, D).permute(0, 2, 1, 3)      # [B, H, T, D]
        dA = do @ c["vh"].transpose(-1, -2)                                                      # [B, H, T, C]
        dv = c["A"].transpose(-1, -2) @ do                                                       # [B, H, C, D]
        dS = c["A"] * (dA - (dA * c["A"]).sum(-1, keepdim=True))
        dq = (dS @ c["kh"]) * scale
        dk = dS.transpose(-1, -2) @ (c["qh"] * scale)
        dq_ = dq.permute(0, 2, 1, 3).reshape(B, T, d)
        dk_ = dk.permute(0, 2, 1, 3).reshape(B, C, d)
        dv_ = dv.permute(0, 2, 1, 3).reshape(B, C, d)
        gw = torch.cat([dq_.reshape(-1, d).t() @ c["n1"].reshape(-1, d), dk_.reshape(-1, d).t() @ ctx.reshape(-1, d),
                        dv_.reshape(-1, d).t() @ ctx.reshape(-1, d)], 0)
        G[q_ + "attn.in_proj_weight"] = gw
        G[q_ + "attn.in_proj_bias"] = torch.cat([dq_.reshape(-1, d).sum(0), dk_.reshape(-1, d).sum(0), dv_.reshape(-1, d).sum(0)])
        dctx = dctx + dk_ @ Win[d:2 * d] + dv_ @ Win[2 * d:]
        dx1, G[q_ + "ln1.weight"], G[q_ + "ln1.bias"] = _layernorm_bwd(dq_ @ Win[:d], c["xh1"], c["rs1"], P[q_ + "ln1.weight"])
        dh = dh1 + dx1
    G["positional_encoding"] = dh.sum(0)
    dEmb = torch.zeros_like(P["embedding.weight"])
    dEmb.index_add_(0, cache["x"], dctx[:, 0])
    G["embedding.weight"] = dEmb
    if cfg.n_fonts > 0:
        dF = torch.zeros_like(P["font_embedding.weight"])
        dF.index_add_(0, cache["font"], dctx[:, 1])
        G["font_embedding.weight"] = dF
    return G


# --------------------------------------------------------------------------- loss / optimiser
def mse_loss_grad(u, target, total_elems=None, clamp_mask=None):
    """F.mse_loss(clamp(u,0,1), target) and its gradient w.r.t. u (model.py:156,268-270).
    clamp passes gradient where 0 <= u <= 1 inclusive (torch clamp_backward).  total_elems
    overrides the mean's denominator (global batch * pixels under data parallelism)."""
    u2 = u.reshape(u.shape[0], -1)
    t2 = target.reshape(u2.shape).to(u2.dtype)
    n = float(total_elems if total_elems is not None else u2.numel())
    y = u2.clamp(0.0, 1.0)
    diff = y - t2
    loss = (diff * diff).sum() / n
    cm = ((u2 >= 0) & (u2 <= 1)) if clamp_mask is None else clamp_mask.reshape(u2.shape)
    du = (2.0 / n) * diff * cm.to(u2.dtype)
    return loss, du


def adamw_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.99, eps=1e-8, wd=5e-4):
    """One torch.optim.AdamW update (model.py:273; torch/optim/adamw.py single-tensor path):
    decoupled decay, bias-corrected moments, denom = sqrt(v)/sqrt(1-b2^t) + eps.  t = 1,2,...
    Returns new (p, m, v)."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def adamw_step_(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.99, eps=1e-8, wd=5e-4):
    """The same update IN PLACE, in the operation order of torch/optim/adamw.py's single-tensor path (mul_, lerp_ /
    mul_.addcmul_, addcdiv_): no temporaries the size of the parameters.  For the CPU-baseline timing (bench.py), where the
    functional form above spent more time allocating than computing; the parity tests use the functional form."""
    p.mul_(1.0 - lr * wd)
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
    return p, m, v


def train_step(P, M, V, t, x, target, cfg, font=None, masks=None, lr=1e-3, beta1=0.9, beta2=0.99,
               eps=1e-8, wd=5e-4, inplace=False):
    """zero_grad -> forward -> MSE -> backward -> AdamW (model.py:292-310).  Returns
    (loss, grads, newP, newM, newV); inplace=True updates P, M, V themselves (adamw_step_)."""
    if cfg.kind == "sheet":
        _, cache = sheet_forward(P, x, cfg, masks)
        loss, du = mse_loss_grad(cache["u"], target)
        G = sheet_backward(P, cache, du, cfg)
    elif cfg.kind == "pixel":
        _, cache = pixel_forward(P, x, font, cfg)
        loss, du = mse_loss_grad(cache["u"], target)
        G = pixel_backward(P, cache, du, cfg)
    else:
        _, cache = glyph_forward(P, x, font, cfg)
        loss, du = mse_loss_grad(cache["u"], target)
        G = glyph_backward(P, cache, du, cfg)
    if inplace:
        for k in P:
            adamw_step_(P[k], G[k], M[k], V[k], t, lr, beta1, beta2, eps, wd)
        return loss, G, P, M, V
    nP, nM, nV = {}, {}, {}
    for k in P:
        nP[k], nM[k], nV[k] = adamw_step(P[k], G[k], M[k], V[k], t, lr, beta1, beta2, eps, wd)
    return loss, G, nP, nM, nV


# --------------------------------------------------------------------------- helpers.py twins
def sheet_to_u8(arr):
    """binary_array_to_image's quantisation: (a*255).astype(uint8) truncates (helpers.py:33)."""
    import numpy as np
    return (np.asarray(arr, dtype=np.float32) * 255).astype(np.uint8)
