"""CPU, gloo, world_size 2: the data-parallel step logic (ai_font_renderer_amd.parallel) driven with an
oracle-backed stand-in engine.  Checks that sharded steps reproduce the single-process full-batch step:
same summed gradients (global-mean weighting, uneven shards) and identical parameters on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .util import ROOT, GlyphConfig, glyph_inputs, oracle, synth, tparams


class OracleEngine:
    """Engine look-alike on CPU: flat_grads / loss_accum / train_step / adamw_step over the oracle."""

    def __init__(self, cfg):
        from ai_font_renderer_amd.config import flat_layout
        self.cfg = cfg
        self.table, n = flat_layout(cfg)
        # flat buffers in the engine's layout; the per-tensor dicts are views into them
        self.flat_params, self.flat_m, self.flat_v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
        init = tparams(cfg)
        self.P, self.M, self.V = {}, {}, {}
        for name, shape, off, k in self.table:
            self.P[name] = self.flat_params[off:off + k].view(shape)
            self.P[name].copy_(init[name])
            self.M[name] = self.flat_m[off:off + k].view(shape)
            self.V[name] = self.flat_v[off:off + k].view(shape)
        self.flat_grads = torch.zeros(n)
        self.loss_accum = torch.zeros(1)
        self.t = 0

    def train_step(self, x, target, font=None, mean_elems=None, do_step=True, **hyper):
        pixel = getattr(self.cfg, "kind", "") == "pixel"
        _, cache = (oracle.pixel_forward if pixel else oracle.glyph_forward)(self.P, x, font, self.cfg)
        loss, du = oracle.mse_loss_grad(cache["u"], target.float() / 255.0, total_elems=mean_elems)
        G = (oracle.pixel_backward if pixel else oracle.glyph_backward)(self.P, cache, du, self.cfg)
        self.flat_grads.zero_()
        for name, shape, off, n in self.table:
            self.flat_grads[off:off + n] = G[name].reshape(-1)
        self.loss_accum += loss
        if do_step:
            self.adamw_step(**hyper)

    def adamw_step(self, **hyper):
        self.adamw_range(0, self.flat_params.numel(), **hyper)

    def adamw_range(self, offset, n, **hyper):
        """AdamW is element-wise: a slice of the flat buffers is a valid unit of work (the sharded optimizer's)."""
        self.t += 1
        sl = slice(offset, offset + n)
        p, m, v = oracle.adamw_step(self.flat_params[sl], self.flat_grads[sl], self.flat_m[sl], self.flat_v[sl], self.t)
        self.flat_params[sl], self.flat_m[sl], self.flat_v[sl] = p, m, v


CFG = GlyphConfig(hidden=(24, 16), out_h=4, out_w=4, n_fonts=2)
ROWS = 37          # uneven over 2 ranks: 19 + 18


def _case(kind):
    """(config, global rows) of a model kind: the small glyph net, or a two-block miniature of BASELINE configs[4]'s transformer."""
    if kind == "pixel":
        from ai_font_renderer_amd.config import PixelConfig
        return PixelConfig(out_h=2, out_w=4, d_model=64, heads=1, layers=2, ff_dim=32, n_fonts=2), 7
    return CFG, ROWS


def _inputs(cfg, rows):
    if getattr(cfg, "kind", "") == "pixel":
        i = np.arange(rows)
        return (32 + (i * 11) % 95).astype(np.int64), (i % 2).astype(np.int64), synth.hash_u8(933, (rows, cfg.out_h, cfg.out_w))
    return glyph_inputs(cfg, rows)


def _worker(rank, world, port, q, shard=False, kind="glyph"):
    CFG, ROWS = _case(kind)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_font_renderer_amd import parallel
    from ai_font_renderer_amd.parallel import DataParallelStepper, shard_rows
    if shard:
        parallel.SHARD_MIN_BYTES = 0          # take the sharded-optimizer schedule for this small net too
        os.environ["AFR_DP_SCHEDULE"] = "shard"   # (opt-in until a multi-GPU node has run it)
    else:
        parallel.SHARD_MIN_BYTES = 1 << 60
    torch.set_num_threads(1)
    x, font, t = _inputs(CFG, ROWS)
    sl = shard_rows(ROWS, rank, world)
    eng = OracleEngine(CFG)
    st = DataParallelStepper(eng, dist, world)
    assert st.sharded() == shard
    for _ in range(3):
        st.step(torch.from_numpy(x[sl]), torch.from_numpy(t[sl]), torch.from_numpy(font[sl]), mean_elems=ROWS * CFG.pixels)
    loss = st.global_loss()
    q.put((rank, {k: v.numpy() for k, v in eng.P.items()}, eng.flat_grads.numpy().copy(), loss))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("shard,kind", [(False, "glyph"), (True, "glyph"), (False, "pixel")])
def test_two_rank_data_parallel_equals_full_batch(shard, kind):
    """shard=False: sum all-reduce of the flat gradients + replicated AdamW.  shard=True: reduce-scatter -> AdamW on each
    rank's half of the flat buffers -> all-gather of the parameters (the schedule of the 492 MB sheet model)."""
    from ai_font_renderer_amd.parallel import DataParallelStepper, shard_rows
    assert [shard_rows(37, r, 2) for r in range(2)] == [slice(0, 19), slice(19, 37)]
    assert sum(s.stop - s.start for s in (shard_rows(8192 * 8 + 5, r, 8) for r in range(8))) == 8192 * 8 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    port += (1 if shard else 0) + (2 if kind == "pixel" else 0)
    CFG, ROWS = _case(kind)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shard, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process full batch
    x, font, t = _inputs(CFG, ROWS)
    eng = OracleEngine(CFG)
    st = DataParallelStepper(eng, None, 1)
    for _ in range(3):
        st.step(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(font), mean_elems=ROWS * CFG.pixels)
    full_loss = st.global_loss()
    for rank, P, g, loss in res:
        assert abs(loss - full_loss) < 1e-6 * full_loss
        if not shard:                                       # (sharded: only a rank's own half of the buffer holds the sum)
            assert np.abs(g - eng.flat_grads.numpy()).max() < 1e-6 * np.abs(g).max()
        for k in P:
            # (pixel transformer: the key projection's bias has an analytically zero gradient -- softmax is shift-invariant -- so Adam
            # moves it by +-lr per step on rounding noise, in either direction in either run: three steps of lr = 1e-3)
            tol = 3.2e-3 if (kind == "pixel" and k.endswith("attn.in_proj_bias")) else 2e-6 if kind == "glyph" else 2e-5
            assert np.abs(P[k] - eng.P[k].numpy()).max() < tol, (rank, k, np.abs(P[k] - eng.P[k].numpy()).max())
    for k in res[0][1]:                                     # replicas stay bit-identical to each other
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
