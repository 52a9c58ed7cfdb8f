"""Host-side driver of libafr.so: owns the device buffers (as PyTorch-ROCm tensors -- torch is only
the allocator and the stream provider here) and mirrors one training iteration of the reference
loop body (model.py:292-310) as forward -> loss_grad -> backward -> [all-reduce] -> adamw.

Needs a GPU and the built extension; raises otherwise (no CPU path).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import GlyphConfig, PixelConfig, SheetConfig

_DT = {"f32": _lib.AFR_F32, "fp32": _lib.AFR_F32, "float32": _lib.AFR_F32, "bf16": _lib.AFR_BF16, "bfloat16": _lib.AFR_BF16}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device=None):
    """The caller's current stream ON `device` (torch.cuda.current_stream() alone is the stream of the process's current
    device, which is GPU 0 unless somebody called set_device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_afr_config(cfg, dtype, max_batch, seed=42, rank=0, flags=0):
    c = _lib.AfrConfig()
    c.reserved = int(flags)          # include/afr.h: bit 0 un-fused optimizer, bit 1 no grouped GEMM launches, bit 2 no fused small-net step
    c.dtype = _DT[dtype]
    c.max_batch = int(max_batch)
    c.vocab = cfg.vocab
    c.embed_dim = getattr(cfg, "embed_dim", 0)      # (PixelConfig: d_model, set below)
    c.seed = int(seed)
    c.rank = int(rank)
    if isinstance(cfg, SheetConfig):
        c.kind = _lib.AFR_KIND_SHEET
        c.out_h, c.out_w = cfg.sheet_h, cfg.sheet_w
        c.max_length, c.heads, c.fc_dim = cfg.max_length, cfg.heads, cfg.fc_dim
        c.p_embed, c.p_attn, c.p_fc, c.ln_eps = cfg.p_embed, cfg.p_attn, cfg.p_fc, cfg.ln_eps
    elif isinstance(cfg, GlyphConfig):
        c.kind = _lib.AFR_KIND_GLYPH
        c.out_h, c.out_w = cfg.out_h, cfg.out_w
        c.n_hidden = len(cfg.hidden)
        for i, h in enumerate(cfg.hidden):
            c.hidden[i] = h
        c.n_fonts = cfg.n_fonts
    elif isinstance(cfg, PixelConfig):
        c.kind = _lib.AFR_KIND_PIXEL                 # BASELINE configs[4] (include/afr.h)
        c.embed_dim = cfg.d_model
        c.out_h, c.out_w = cfg.out_h, cfg.out_w
        c.heads, c.fc_dim, c.n_hidden, c.n_fonts, c.ln_eps = cfg.heads, cfg.ff_dim, cfg.layers, cfg.n_fonts, cfg.ln_eps
    else:
        raise TypeError(f"unknown config {type(cfg)}")
    return c


class Engine:
    """One plan + its device buffers.  `params[name]` are views into the flat float32 buffer in
    state_dict order, so checkpoints interchange with the reference (helpers.py:76-105)."""

    def __init__(self, cfg, dtype="f32", max_batch=1024, device=None, seed=42, rank=0, with_optimizer=True, flags=0, micro_batch=None):
        """micro_batch: train_step / forward_loss + backward of a batch larger than this many samples run as micro-steps of at
        most that many, their gradients summed (gradient accumulation: the saved activations of BASELINE configs[4]'s 2048
        glyphs per GPU would be 800 GB; 32 at a time they are 12.6 GB).  The plan is then sized for micro_batch, not max_batch."""
        if not torch.cuda.is_available():
            raise RuntimeError("ai_font_renderer_amd.Engine needs an MI355X: the hot path has no CPU fallback")
        self.lib = _lib.lib()
        self.micro_batch = int(micro_batch) if micro_batch else None
        if self.micro_batch:
            max_batch = min(int(max_batch), self.micro_batch)
        self._grad_acc = None
        self.cfg, self.dtype, self.max_batch = cfg, dtype, int(max_batch)
        self.seed, self.rank, self.flags = int(seed), int(rank), int(flags)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._plan = None
        self._make_plan(self.max_batch)
        n = self.lib.afr_param_elems(self._plan)
        self.n_flat = int(n)
        with torch.cuda.device(self.device):
            self.flat_params = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.flat_grads = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.device) if with_optimizer else None
            self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.device) if with_optimizer else None
            self.loss_accum = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._bind()
        self.layout = []
        name = C.create_string_buffer(128)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int32()
        shape = (C.c_int64 * 4)()
        for i in range(self.lib.afr_param_count(self._plan)):
            _lib.check(self.lib.afr_param_info(self._plan, i, name, 128, C.byref(off), C.byref(numel), C.byref(ndim), shape))
            self.layout.append((name.value.decode(), tuple(shape[k] for k in range(ndim.value)), off.value, numel.value))
        self.params = {nm: self.flat_params[o:o + k].view(shp) for nm, shp, o, k in self.layout}
        self.grads = {nm: self.flat_grads[o:o + k].view(shp) for nm, shp, o, k in self.layout}
        self.pixels = cfg.pixels
        self.t = 0            # AdamW step counter (model.py:310)
        self._keep = None     # keeps the last inputs alive until backward has consumed them

    def _call(self, fn, *args):
        """One libafr call that enqueues work: on THIS engine's device and on the caller's current stream of that device
        (the engine may live on cuda:k while the process's current device is another GPU)."""
        with torch.cuda.device(self.device):
            _lib.check(fn(*args, _stream(self.device)))

    def _make_plan(self, max_batch):
        if self._plan:
            self.lib.afr_plan_destroy(self._plan)
        self.max_batch = int(max_batch)
        self._c = make_afr_config(self.cfg, self.dtype, self.max_batch, self.seed, self.rank, self.flags)
        self._plan = C.c_void_p()
        _lib.check(self.lib.afr_plan_create(C.byref(self._c), C.byref(self._plan)))

    def _bind(self):
        with torch.cuda.device(self.device):
            self.ws_bytes = int(self.lib.afr_workspace_bytes(self._plan))
            self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.afr_bind(self._plan, _ptr(self.flat_params), _ptr(self.flat_grads), _ptr(self.exp_avg),
                                     _ptr(self.exp_avg_sq), _ptr(self.workspace), self.ws_bytes))

    def ensure_batch(self, B):
        """Grow the plan's workspace for a larger batch; parameters, gradients and moments stay where they are."""
        if B > self.max_batch:
            torch.cuda.synchronize(self.device)
            self._make_plan(B)
            self._bind()
            self.sync_params()

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self.lib.afr_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ---------------------------------------------------------------- parameters
    def load_params(self, tensors):
        """tensors: dict name -> numpy array / torch tensor (any device), state_dict keys."""
        for nm, shp, _, _ in self.layout:
            src = tensors[nm]
            src = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src.detach()
            if tuple(src.shape) != tuple(shp):
                raise ValueError(f"{nm}: shape {tuple(src.shape)} != {tuple(shp)}")
            self.params[nm].copy_(src.to(torch.float32))
        self.sync_params()

    def sync_params(self):
        self._call(self.lib.afr_sync_params, self._plan)

    def state_dict(self):
        return {nm: self.params[nm].detach().clone() for nm, _, _, _ in self.layout}

    def reset_optimizer(self):
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.t = 0

    # ---------------------------------------------------------------- the hot path
    def _prep_x(self, x, font):
        x = x.to(self.device, dtype=torch.int64, non_blocking=True).contiguous()
        if font is not None:
            font = font.to(self.device, dtype=torch.int64, non_blocking=True).contiguous()
        return x, font

    def forward(self, x, font=None, training=False, step=0, want_output=True):
        x, font = self._prep_x(x, font)
        if self.micro_batch and x.shape[0] > self.micro_batch:        # an inference forward of a large batch, micro_batch rows at a time
            outs = [self.forward(x[lo:lo + self.micro_batch], None if font is None else font[lo:lo + self.micro_batch], training, step, want_output)
                    for lo in range(0, x.shape[0], self.micro_batch)]
            return torch.cat(outs) if want_output else None
        self.ensure_batch(x.shape[0])
        if isinstance(self.cfg, SheetConfig):
            if x.dim() != 2:
                raise ValueError("sheet model takes int64 [B, L] codes")
            B, L = x.shape
        else:
            x = x.reshape(-1)
            B, L = x.shape[0], 1
        y = torch.empty(B, self.pixels, dtype=torch.float32, device=self.device) if want_output else None
        self._call(self.lib.afr_forward, self._plan, _ptr(x), _ptr(font), B, L, _ptr(y), int(bool(training)), int(step))
        self._keep = (x, font)
        if y is None:
            return None
        h, w = (self.cfg.sheet_h, self.cfg.sheet_w) if isinstance(self.cfg, SheetConfig) else (self.cfg.out_h, self.cfg.out_w)     # glyph / pixel
        return y.view(B, h, w)

    def _target(self, target):
        if target.dtype == torch.uint8:
            return target.to(self.device, non_blocking=True).contiguous(), _lib.AFR_TARGET_U8
        return target.to(self.device, dtype=torch.float32, non_blocking=True).contiguous(), _lib.AFR_TARGET_F32

    def loss_grad(self, target, mean_elems=None):
        t, td = self._target(target)
        B = t.shape[0]
        me = int(mean_elems) if mean_elems is not None else B * self.pixels
        self._call(self.lib.afr_loss_grad, self._plan, _ptr(t), td, B, me, _ptr(self.loss_accum))
        self._keep_t = t

    def set_output_grad(self, dy):
        """dy = d(loss)/d(clamped output) from a caller-side loss (autograd); float32 [B, pixels]."""
        dy = dy.to(self.device, dtype=torch.float32).contiguous()
        self._call(self.lib.afr_set_output_grad, self._plan, _ptr(dy), dy.shape[0])
        self._keep_t = dy

    def backward(self):
        self._call(self.lib.afr_backward, self._plan)

    @property
    def backward_stages(self):
        return int(self.lib.afr_backward_stages(self._plan))

    def backward_stage(self, stage):
        """Run one backward stage; returns the view of flat_grads that is final after it."""
        off, n = C.c_int64(), C.c_int64()
        self._call(self.lib.afr_backward_stage, self._plan, int(stage), C.byref(off), C.byref(n))
        return self.flat_grads[off.value:off.value + n.value]

    def forward_loss(self, x, target, font=None, step=None, mean_elems=None):
        """Training forward with the loss/grad fused into the last layer's epilogue (no optimizer step)."""
        x, font = self._prep_x(x, font)
        if self.micro_batch and x.shape[0] > self.micro_batch:
            raise ValueError(f"forward_loss / backward work on one micro-batch (<= {self.micro_batch} samples); a batch of {x.shape[0]} "
                             "accumulates through train_step")
        self.ensure_batch(x.shape[0])
        t, td = self._target(target)
        if isinstance(self.cfg, SheetConfig):
            B, L = x.shape
        else:
            x = x.reshape(-1)
            B, L = x.shape[0], 1
        me = int(mean_elems) if mean_elems is not None else B * self.pixels
        st = int(step if step is not None else self.t + 1)
        self._call(self.lib.afr_forward_loss, self._plan, _ptr(x), _ptr(font), _ptr(t), td, B, L, me, _ptr(self.loss_accum), st)
        self._keep, self._keep_t = (x, font), t

    def adamw_step(self, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-4, grad_scale=1.0):
        self.t += 1
        self._call(self.lib.afr_adamw_step, self._plan, lr, betas[0], betas[1], eps, weight_decay, self.t, grad_scale)

    def adamw_range(self, offset, n, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-4, grad_scale=1.0):
        """One AdamW step on the flat-buffer slice [offset, offset + n) only (sharded optimizer under data parallelism:
        parallel.py).  Advances the step counter; the bf16 shadow is NOT refreshed (the caller syncs after its all-gather)."""
        self.t += 1
        o, e = int(offset), int(offset) + int(n)
        self._call(self.lib.afr_op_adamw, _ptr(self.flat_params[o:e]), _ptr(self.flat_grads[o:e]), _ptr(self.exp_avg[o:e]),
                   _ptr(self.exp_avg_sq[o:e]), C.c_void_p(0), int(n), lr, betas[0], betas[1], eps, weight_decay, self.t, grad_scale)

    def train_step(self, x, target, font=None, step=None, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-4,
                   mean_elems=None, do_step=True):
        """zero_grad -> forward -> loss -> backward -> AdamW, one C call (model.py:292-310)."""
        x, font = self._prep_x(x, font)
        if self.micro_batch and x.shape[0] > self.micro_batch:
            return self._train_step_accumulated(x, target, font, step, lr, betas, eps, weight_decay, mean_elems, do_step)
        self.ensure_batch(x.shape[0])
        t, td = self._target(target)
        if isinstance(self.cfg, SheetConfig):
            B, L = x.shape
        else:
            x = x.reshape(-1)
            B, L = x.shape[0], 1
        me = int(mean_elems) if mean_elems is not None else B * self.pixels
        if do_step:
            self.t += 1
        st = int(step if step is not None else self.t)
        self._call(self.lib.afr_train_step, self._plan, _ptr(x), _ptr(font), _ptr(t), td, B, L, me, _ptr(self.loss_accum), st,
                                           int(bool(do_step)), lr, betas[0], betas[1], eps, weight_decay, max(self.t, 1))
        self._keep = (x, font)
        self._keep_t = t

    def _train_step_accumulated(self, x, target, font, step, lr, betas, eps, weight_decay, mean_elems, do_step):
        """Gradient accumulation: the batch in micro-steps of self.micro_batch samples (forward + loss + backward each, the loss
        and its gradient scaled for the WHOLE batch through mean_elems), gradients summed in micro-step order, one AdamW step."""
        B = x.shape[0]
        me = int(mean_elems) if mean_elems is not None else B * self.pixels
        t, _ = self._target(target)
        if self._grad_acc is None:
            self._grad_acc = torch.empty_like(self.flat_grads)
        st = int(step if step is not None else self.t + (1 if do_step else 0))
        for i, lo in enumerate(range(0, B, self.micro_batch)):
            hi = min(B, lo + self.micro_batch)
            # (models with dropout: micro-step i draws its masks from stream st * 65536 + i)
            self.train_step(x[lo:hi], t[lo:hi], font=None if font is None else font[lo:hi], step=st * 65536 + i if step is None else st,
                            mean_elems=me, do_step=False)
            # acc (+)= this micro-step's gradient, by the library's own slab-sum kernel (fixed order: micro-step by micro-step)
            self._call(self.lib.afr_op_reduce, _ptr(self._grad_acc), _ptr(self.flat_grads), 1, self.n_flat, self.n_flat, 1.0, int(i > 0))
        self._call(self.lib.afr_op_reduce, _ptr(self.flat_grads), _ptr(self._grad_acc), 1, self.n_flat, self.n_flat, 1.0, 0)
        if do_step:
            self.adamw_step(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def read_loss(self, reset=True):
        v = float(self.loss_accum.item())
        if reset:
            self.loss_accum.zero_()
        return v

    def error_flags(self):
        out = C.c_uint32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.afr_error_flags(self._plan, _stream(self.device), C.byref(out)))
        return out.value

    def debug_read(self, which, index=0):
        """Copy an internal activation buffer of the last call (u/du, z, dz, glyph activation i) as float32."""
        code = {"u": _lib.BUF_U, "z": _lib.BUF_Z, "dz": _lib.BUF_DZ, "act": _lib.BUF_ACT + int(index), "w1t": _lib.BUF_W1T, "w2t": _lib.BUF_W2T}[which]
        dt = torch.bfloat16 if (self.dtype in ("bf16", "bfloat16") or which in ("w1t", "w2t")) else torch.float32
        es = 2 if dt == torch.bfloat16 else 4
        cap = max(self.max_batch * max(self.pixels, getattr(self.cfg, "flat_dim", 0), *(getattr(self.cfg, "hidden", (0,))), self.cfg.embed_dim) * es, 1 << 20)
        buf = torch.empty(cap, dtype=torch.uint8, device=self.device)
        n = C.c_size_t()
        self._call(self.lib.afr_debug_copy, self._plan, code, _ptr(buf), cap, C.byref(n))
        torch.cuda.synchronize(self.device)
        return buf[:n.value].view(dt).float()

    def debug_sheet_gather(self, x):
        """The rows the sheet front end's in-kernel embedding gather fetched for x [B, L]: float32 [B, min(L, max_length), E]."""
        x, _ = self._prep_x(x, None)
        self.ensure_batch(x.shape[0])
        B, L = x.shape
        e0 = torch.full((B, min(L, self.cfg.max_length), self.cfg.embed_dim), float("nan"), dtype=torch.float32, device=self.device)
        self._call(self.lib.afr_debug_sheet_gather, self._plan, _ptr(x), B, L, _ptr(e0))
        torch.cuda.synchronize(self.device)
        return e0

    # ---------------------------------------------------------------- measurement
    def profile(self, mode=1):
        """0 off, 1 time every launch, 2 time only the kernel that dominated the mode-1 recording."""
        _lib.check(self.lib.afr_profile_dominant(self._plan, int(mode)))

    def profile_table(self):
        buf = C.create_string_buffer(8192)
        _lib.check(self.lib.afr_profile_dump(self._plan, buf, 8192))
        rows = []
        for line in buf.value.decode().strip().split("\n"):
            if line:
                k, n, tot, avg, fl, by = line.split("\t")
                rows.append(dict(kernel=k, launches=int(n), total_ms=float(tot), avg_ms=float(avg), algo_flops=float(fl), algo_bytes=float(by)))
        return sorted(rows, key=lambda r: -r["total_ms"])

    def profile_read(self):
        name = C.create_string_buffer(128)
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        _lib.check(self.lib.afr_profile_read(self._plan, name, 128, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        return dict(kernel=name.value.decode(), avg_ms=ms.value, launches=n.value, algo_flops=fl.value, algo_bytes=by.value)
