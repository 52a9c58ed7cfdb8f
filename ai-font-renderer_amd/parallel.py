"""Data parallelism for the training step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device (model.py:95-106); sharding is new.  Samples are independent, so
the batch is cut into contiguous row slices, one per rank, with NO data-path collective; the only exchange per
step is the sum all-reduce of the flat gradient buffer (all parameters of state_dict() live in one contiguous
float32 buffer).  Large models issue it as two RCCL calls: the last layer's range asynchronously as soon as backward has
produced it (overlapping the rest of the backward pass), the remainder when backward is done; small ones (< 64 MB of
gradients) as one call after backward.  A sharded-optimizer schedule (reduce-scatter -> AdamW on 1/N -> all-gather) exists
as an opt-in (AFR_DP_SCHEDULE=shard) until a multi-GPU node has run it.  Every shard divides its loss by the GLOBAL element count
(`mean_elems`), so the summed gradients equal the full-batch gradients exactly, also for uneven last batches
(192 / 304 rows in the reference's loaders).  Parameters, AdamW moments and the step counter are replicated;
each rank draws its own dropout stream (rank is part of the counter-hash key).

`engine` is anything with train_step(x, target, font=, mean_elems=, do_step=), flat_grads, adamw_step(**hyper)
and loss_accum: the HIP Engine in production; tests drive the same logic on CPU (gloo) with a stand-in.
"""
import os

import torch


def shard_rows(n_rows, rank, world):
    """Contiguous slice of the global batch owned by `rank`: sizes differ by at most one row."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


# Below this gradient size the step uses ONE all-reduce after a monolithic backward: the overlapped two-collective
# schedule costs ~33 us of staging and cross-stream hand-offs per step (measured with a world of one on the 8.5 MB C3
# model: 274 us against 240 us), more than the transfer time it could hide.  The sheet model (492 MB) overlaps.
OVERLAP_MIN_BYTES = int(os.environ.get("AFR_DP_OVERLAP_MIN_BYTES", 64 << 20))
# From this gradient size on (the sheet model: 492 MB, 99.98 % of it one tensor) the optimizer state is SHARDED over the
# ranks instead of replicated: reduce-scatter of the flat gradient buffer -> AdamW on this rank's 1/N slice only (1/N of
# the 28 B/parameter of optimizer traffic, 3.4 GB per step for the sheet model) -> all-gather of the updated parameters.
# Same bytes on the links as the all-reduce it replaces (which is a reduce-scatter + all-gather inside RCCL), parameters
# stay bit-identical on every rank; exp_avg / exp_avg_sq are only meaningful inside a rank's own slice.
SHARD_MIN_BYTES = int(os.environ.get("AFR_DP_SHARD_MIN_BYTES", 64 << 20))
# The sharded schedule is OPT-IN (AFR_DP_SCHEDULE=shard) until it has run on a multi-GPU node: the default for large models
# is the overlapped all-reduce.  "shard-force" takes the sharded path at any size and at a world of one (the real aliasing
# RCCL calls, Engine.adamw_range on an offset slice and the shadow re-sync on one GPU: tests/test_gpu_parallel.py).
def _schedule():
    return os.environ.get("AFR_DP_SCHEDULE", "overlap")
# AFR_DP_GRAD_BF16=1 (opt-in, throughput mode only): exchange the gradients as bf16 -- half the bytes on the xGMI links
# at the price of rounding each rank's gradient to 8 significant bits before the sum (the exact-f32 exchange is the default
# and what the data-parallel tests pin).  Meant for link-bound small models; unmeasured on a multi-GPU node so far.
GRAD_BF16 = os.environ.get("AFR_DP_GRAD_BF16") == "1"


def _has_tensor_collectives(dist):
    """RCCL ("nccl") has reduce_scatter_tensor / all_gather_into_tensor; gloo (the CPU tests) does not.  The path is chosen
    from the BACKEND, once, identically on every rank: never from whether a call raised -- a rank-local failure must not
    make one rank issue a different collective from its peers (a hang or silent corruption instead of an error)."""
    return dist.get_backend() != "gloo"


def _reduce_scatter_inplace(dist, flat, rank, world):
    """Sum `flat` over the ranks; afterwards this rank's slice [rank*n/world, (rank+1)*n/world) holds the sum (the rest is
    unspecified).  RCCL: a true in-place reduce-scatter (errors propagate); gloo: an all-reduce."""
    n = flat.numel() // world
    mine = flat[rank * n:(rank + 1) * n]
    if _has_tensor_collectives(dist):
        dist.reduce_scatter_tensor(mine, flat)
    else:
        dist.all_reduce(flat)
    return mine


def _all_gather_inplace(dist, flat, rank, world):
    """Every rank contributes its slice of `flat`; afterwards all of `flat` is identical everywhere."""
    n = flat.numel() // world
    if _has_tensor_collectives(dist):
        dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n])
    else:
        parts = [torch.empty(n, dtype=flat.dtype, device=flat.device) for _ in range(world)]
        dist.all_gather(parts, flat[rank * n:(rank + 1) * n].clone())
        for r, part in enumerate(parts):
            flat[r * n:(r + 1) * n].copy_(part)


class DataParallelStepper:
    def __init__(self, engine, dist=None, world=1, rank=None):
        self.engine, self.dist, self.world = engine, dist, int(world)
        self.rank = int(rank) if rank is not None else (dist.get_rank() if (dist is not None and self.world > 1 and dist.is_initialized()) else 0)

    def sharded(self):
        eng = self.engine
        sched = _schedule()
        if sched not in ("shard", "shard-force") or self.dist is None or not hasattr(eng, "adamw_range"):
            return False
        ws = self.dist.get_world_size()
        if sched == "shard-force":
            return eng.flat_grads.numel() % ws == 0
        return (self.world > 1 and eng.flat_grads.numel() * 4 >= SHARD_MIN_BYTES and eng.flat_grads.numel() % self.world == 0
                and ws == self.world)

    def step(self, x, target, font=None, mean_elems=None, **hyper):
        """One optimiser step on this rank's shard.  mean_elems = global_rows * pixels."""
        eng = self.engine
        if (self.world == 1 and not (_schedule() == "shard-force" and self.dist is not None)) or self.dist is None:
            eng.train_step(x, target, font=font, mean_elems=mean_elems, do_step=True, **hyper)
            return
        opt = {k: v for k, v in hyper.items() if k in ("lr", "betas", "eps", "weight_decay")}
        stages = getattr(eng, "backward_stages", 0)
        mb = getattr(eng, "micro_batch", None)
        accumulating = bool(mb) and x.shape[0] > mb          # gradient accumulation: the sum exists only after the last micro-step
        if stages and eng.flat_grads.numel() * 4 >= OVERLAP_MIN_BYTES and not self.sharded() and not accumulating:
            # Backward runs last layer first.  Two collectives per step: the last layer's gradient range (half of the
            # bytes in the glyph nets, 99.98 % in the sheet model) is all-reduced ASYNCHRONOUSLY as soon as stage 0 has
            # produced it and overlaps the rest of the backward pass; everything else is one contiguous range
            # [0, start of that range) reduced when the last stage is done.  (One collective per stage overlapped more
            # bytes but paid ~20 us of cross-stream event hand-offs per collective: measured with a world of one.)
            eng.forward_loss(x, target, font=font, step=hyper.get("step"), mean_elems=mean_elems)
            first = eng.backward_stage(0)
            work = self.dist.all_reduce(first, async_op=True)
            for s in range(1, stages):
                eng.backward_stage(s)
            rest = eng.flat_grads[:first.storage_offset()]
            if rest.numel():
                self.dist.all_reduce(rest)
            work.wait()
        elif self.sharded():
            # sharded optimizer: backward -> reduce-scatter -> AdamW on this rank's slice -> all-gather of the parameters
            eng.train_step(x, target, font=font, mean_elems=mean_elems, do_step=False, **hyper)
            ws = self.dist.get_world_size()               # == self.world except under shard-force (a world of one)
            rk = self.rank if ws == self.world else self.dist.get_rank()
            n = eng.flat_grads.numel() // ws
            _reduce_scatter_inplace(self.dist, eng.flat_grads, rk, ws)
            eng.adamw_range(rk * n, n, **opt)
            _all_gather_inplace(self.dist, eng.flat_params, rk, ws)
            if hasattr(eng, "sync_params"):
                eng.sync_params()               # bf16 mode: the shadow of the slices other ranks updated
            return
        else:
            eng.train_step(x, target, font=font, mean_elems=mean_elems, do_step=False, **hyper)
            if GRAD_BF16 and getattr(eng, "dtype", "f32") == "bf16":
                g16 = eng.flat_grads.to(torch.bfloat16)
                self.dist.all_reduce(g16)
                eng.flat_grads.copy_(g16)
            else:
                self.dist.all_reduce(eng.flat_grads)          # sum over ranks; RCCL ring/tree over xGMI
        eng.adamw_step(**opt)

    def global_loss(self, reset=True):
        """Sum of the shards' loss shares == the global mean loss (each share is already / mean_elems)."""
        eng = self.engine
        acc = eng.loss_accum.clone()
        if self.world > 1 and self.dist is not None:
            self.dist.all_reduce(acc)
        if reset:
            eng.loss_accum.zero_()
        return float(acc.item())
