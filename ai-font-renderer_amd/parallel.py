"""Data parallelism for the training step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device (model.py:95-106); sharding is new.  Samples are independent, so
the batch is cut into contiguous row slices, one per rank, with NO data-path collective; the only exchange per
step is the sum all-reduce of the flat gradient buffer (all parameters of state_dict() live in one contiguous
float32 buffer).  Large models issue it as two RCCL calls: the last layer's range asynchronously as soon as backward has
produced it (overlapping the rest of the backward pass), the remainder when backward is done; small ones (< 64 MB of
gradients) as one call after backward.  Every shard divides its loss by the GLOBAL element count
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
OVERLAP_MIN_BYTES = 64 << 20
# AFR_DP_GRAD_BF16=1 (opt-in, throughput mode only): exchange the gradients as bf16 -- half the bytes on the xGMI links
# at the price of rounding each rank's gradient to 8 significant bits before the sum (the exact-f32 exchange is the default
# and what the data-parallel tests pin).  Meant for link-bound small models; unmeasured on a multi-GPU node so far.
GRAD_BF16 = os.environ.get("AFR_DP_GRAD_BF16") == "1"


class DataParallelStepper:
    def __init__(self, engine, dist=None, world=1):
        self.engine, self.dist, self.world = engine, dist, int(world)

    def step(self, x, target, font=None, mean_elems=None, **hyper):
        """One optimiser step on this rank's shard.  mean_elems = global_rows * pixels."""
        eng = self.engine
        if self.world == 1 or self.dist is None:
            eng.train_step(x, target, font=font, mean_elems=mean_elems, do_step=True, **hyper)
            return
        opt = {k: v for k, v in hyper.items() if k in ("lr", "betas", "eps", "weight_decay")}
        stages = getattr(eng, "backward_stages", 0)
        if stages and eng.flat_grads.numel() * 4 >= OVERLAP_MIN_BYTES:
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
