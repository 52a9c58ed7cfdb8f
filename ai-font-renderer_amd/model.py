"""Drop-in for the reference's model module on MI355X: same constants, class, training entry points, artefacts
and CLI (reference model.py:64-127, 129-204, 209-384, 389-454), with the hot path -- forward, MSE, backward,
AdamW -- running in libafr.so (hand-written HIP, see csrc/) instead of torch ops.

    python model.py --train        train, save font_renderer.pth, render the test strings
    python model.py                load (or train if missing) and render
    anything else                  two usage lines, exit status 1

What is deliberately different from the reference, and why:
  * device selection: the reference pins CUDA_VISIBLE_DEVICES="3" (model.py:95); here each process uses the GPU
    given by LOCAL_RANK (one process per GPU), and there is no CPU/MPS fallback -- the product IS the HIP path;
  * the dataset lives in HBM as uint8 sheets and batches are gathered on the device by index; the 2x32 DataLoader
    worker processes and the per-step 78 MB host->device copy (model.py:249-266,295-296) are gone.  Split and
    shuffle order reproduce random_split / DataLoader(shuffle=True, generator=g) draw for draw (_EpochOrder);
  * loss.item() per step (model.py:311) becomes one device->host read per epoch (the loss accumulates on device);
  * dropout uses a counter-hash stream instead of torch's bernoulli_ stream (same rates, same placement);
  * AFR_DTYPE=bf16 selects the throughput mode (bf16 MFMA operands); the default f32 mode is the parity mode.
"""
import datetime
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn

from . import helpers
from .config import SheetConfig
from .helpers import MODEL_FILENAME, load_model, load_string_dataset, render_strings, save_model  # noqa: F401

# ---------------------------------------------------------------- constants (reference model.py:64-87)
SHEET_HEIGHT = 80
SHEET_WIDTH = 240
MAX_CHARS_PER_SHEET = 100
NUM_SAMPLES = 150000
OUTPUT_DIR = "train_output_" + datetime.datetime.now().strftime("%m_%d_%H_%M_%S")
NUM_EPOCHS = 10000
LEARNING_RATE = 0.001
EARLY_STOPPING_PATIENCE = 70
VALIDATION_SPLIT = 0.2
WEIGHT_DECAY = 0.0005
EMBEDDING_DIM = 32
DROPOUT_RATE = 0.2
NUM_ATTENTION_HEADS = 4
SCHEDULER_PATIENCE = 20
SCHEDULER_FACTOR = 0.7
MIN_LEARNING_RATE = 1e-6
SEED = 42
ADAM_BETAS = (0.9, 0.99)                      # model.py:273
COMPUTE_DTYPE = os.environ.get("AFR_DTYPE", "f32")

random.seed(SEED)
np.random.seed(SEED)
torch.manual_seed(SEED)

_LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
if torch.cuda.is_available():
    device = torch.device("cuda", _LOCAL_RANK)
    torch.cuda.set_device(device)              # one process per GPU: torch ops AND libafr launches of this process go to it
else:                                          # importable for inspection; constructing a model will raise
    device = torch.device("cpu")


def _init_distributed():
    """torchrun --nproc-per-node N model.py --train: one rank per GPU, gradients all-reduced over RCCL (parallel.py).
    Without WORLD_SIZE > 1 in the environment this is a no-op (single GPU, as the reference)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and dist.is_available() and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=device)      # "nccl" is RCCL on ROCm

# the reference's 15 evaluation inputs (model.py:111-127): rendered every 5 epochs and at the end
test_strings = [
    "HELLO LEANN I LOVE YOU SO MUCH I HOPE YOU HAVE A GREAT DAY",
    "TWO WORLDS ONE FAMILY TRUST YOUR HEART LET FATE DECIDE TO GUIDE THESE LIVES WE SEE",
    "A PARADISE UNTOUCHED BY MAN WITHIN THIS WORLD BLESSED WITH LOVE A SIMPLE LIFE THEY LIVE IN PEACE",
    "SOFTLY TREAD THE SAND BELOW YOUR FEET NOW TWO WORLDS ONE FAMILY TRUST YOUR HEART LET FATE",
    "BENEATH THE SHELTER OF THE TREES ONLY LOVE CAN ENTER HERE A SIMPLE LIFE THEY LIVE IN PEACE",
    "THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG",
    "ABCDEFGHIJKLMNOPQRSTUVWXYZ",
    "W" * 20,
    "I" * 20,
    "ALTERNATING CASE TEST   SPACES",
    "CLAUDE IS RENDERING FONTS",
    "ZYXWVUTSRQPONMLKJIHGFEDCBA",
    "AEIOU BCDFGHJKLMNPQRSTVWXYZ",
    "EXACTLY TWENTY CHARS",
    " " * 20,
]


# ---------------------------------------------------------------- model
class _Bag(nn.Module):
    """Names a group of parameters so that state_dict() keys match the reference's module tree."""

    def __init__(self, **tensors):
        super().__init__()
        for k, v in tensors.items():
            if isinstance(v, nn.Module):
                self.add_module(k, v)
            else:
                self.register_parameter(k, v)


def _reference_style_init(cfg):
    """Default initial values exactly as torch constructs the reference's layers, in the reference's creation order
    (model.py:136-152), so the same torch seed gives the same starting point."""
    E, F = cfg.embed_dim, cfg.fc_dim
    emb = nn.Embedding(cfg.vocab, E)
    pos = torch.zeros(cfg.max_length, E)
    nn.init.normal_(pos, mean=0, std=0.02)
    attn = nn.MultiheadAttention(embed_dim=E, num_heads=cfg.heads, dropout=cfg.p_attn)
    ln = nn.LayerNorm(E)
    fc1 = nn.Linear(E, F)
    fco = nn.Linear(F * cfg.max_length, cfg.sheet_h * cfg.sheet_w)
    return {
        "positional_encoding": pos, "embedding.weight": emb.weight.detach(),
        "attention.in_proj_weight": attn.in_proj_weight.detach(), "attention.in_proj_bias": attn.in_proj_bias.detach(),
        "attention.out_proj.weight": attn.out_proj.weight.detach(), "attention.out_proj.bias": attn.out_proj.bias.detach(),
        "layer_norm.weight": ln.weight.detach(), "layer_norm.bias": ln.bias.detach(),
        "fc1.weight": fc1.weight.detach(), "fc1.bias": fc1.bias.detach(),
        "fc_output.weight": fco.weight.detach(), "fc_output.bias": fco.bias.detach(),
    }


class _EngineForward(torch.autograd.Function):
    """forward(x) through libafr; backward routes d(loss)/d(sheet) into afr_backward and exposes the gradients as
    .grad views of the engine's flat gradient buffer (overwrite semantics == the reference's zero_grad + backward)."""

    @staticmethod
    def forward(ctx, anchor, module, x):
        step = module._next_step() if module.training else 0
        ctx.module = module
        return module.engine.forward(x, training=module.training, step=step)

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.module
        eng = m.engine
        eng.set_output_grad(grad_out.reshape(grad_out.shape[0], -1))
        eng.backward()
        for name, p in m.named_parameters():
            p.grad = eng.grads[name]
        return None, None, None


class AttentionFontRenderer(nn.Module):
    """Reference model.py:129-204.  forward(x: int64 [B,L]) -> float32 [B, SHEET_HEIGHT, SHEET_WIDTH] in [0,1];
    L > max_length is truncated, L < max_length zero-pads the flattened features; an index >= 128 raises
    IndexError (checked when `strict_indices`, default, at the cost of a device sync in eval mode only)."""

    def __init__(self, max_length=MAX_CHARS_PER_SHEET, dtype=None, max_batch=1024, seed=SEED, rank=None, init=True):
        super().__init__()
        from .engine import Engine
        self.max_length = max_length
        self.embedding_dim = EMBEDDING_DIM
        self.config = SheetConfig(max_length=max_length, embed_dim=EMBEDDING_DIM, heads=NUM_ATTENTION_HEADS, fc_dim=64,
                                  sheet_h=SHEET_HEIGHT, sheet_w=SHEET_WIDTH, p_embed=DROPOUT_RATE, p_attn=DROPOUT_RATE,
                                  p_fc=DROPOUT_RATE + 0.05)
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.engine = Engine(self.config, dtype=dtype or COMPUTE_DTYPE, max_batch=max_batch, device=device, seed=seed, rank=rank)
        P = {k: nn.Parameter(v) for k, v in self.engine.params.items()}
        self.positional_encoding = P["positional_encoding"]
        self.embedding = _Bag(weight=P["embedding.weight"])
        self.attention = _Bag(in_proj_weight=P["attention.in_proj_weight"], in_proj_bias=P["attention.in_proj_bias"],
                              out_proj=_Bag(weight=P["attention.out_proj.weight"], bias=P["attention.out_proj.bias"]))
        self.layer_norm = _Bag(weight=P["layer_norm.weight"], bias=P["layer_norm.bias"])
        self.fc1 = _Bag(weight=P["fc1.weight"], bias=P["fc1.bias"])
        self.fc_output = _Bag(weight=P["fc_output.weight"], bias=P["fc_output.bias"])
        self.strict_indices = True
        self._steps = 0
        self._seen_versions = None
        if init:
            self.engine.load_params(_reference_style_init(self.config))

    def _param_versions(self):
        return tuple(p._version for p in self.parameters())

    def _refresh_shadow_if_params_changed(self):
        """bf16 mode: the GEMMs read a bf16 shadow of the f32 masters that only libafr's own AdamW keeps current.  A stock
        torch optimizer (or any in-place edit of a Parameter) bumps the tensor's version counter; when the counters moved
        since the last forward the shadow is re-derived first, so the next forward sees the new weights."""
        if self.engine.dtype not in ("bf16", "bfloat16"):
            return
        v = self._param_versions()
        if v != self._seen_versions:
            self.engine.sync_params()
            self._seen_versions = v

    def _next_step(self):
        self._steps += 1
        return self._steps

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError(f"expected [batch, seq_len] codes, got shape {tuple(x.shape)}")
        self._refresh_shadow_if_params_changed()
        if torch.is_grad_enabled() and self.training:
            y = _EngineForward.apply(self.positional_encoding, self, x)
        else:
            y = self.engine.forward(x, training=self.training, step=self._next_step() if self.training else 0)
        if self.strict_indices and not self.training and self.engine.error_flags():
            raise IndexError("index out of range in self")          # what nn.Embedding raises in the reference
        return y

    # the parameters live in HBM inside the engine: moving the module is a no-op
    def to(self, *args, **kwargs):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self.engine.sync_params()                                  # refresh bf16 shadows
        return out


# ---------------------------------------------------------------- data order (random_split + DataLoader, on device)
class _EpochOrder:
    """Reproduces which samples the reference trains/validates on and in which order (model.py:232-266):
    random_split(generator=manual_seed(42)) is one randperm; per epoch, from the generator the two loaders share,
    each DataLoader iterator draws one base-seed integer and RandomSampler draws two randperms (the second is its
    empty remainder).  Checked against the real loaders in tests/test_host_cpu.py (torch 2.10 sampler internals)."""

    def __init__(self, n, val_fraction=VALIDATION_SPLIT, seed=SEED):
        self.val_size = int(val_fraction * n)
        self.train_size = n - self.val_size
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        self.train_idx, self.val_idx = perm[:self.train_size], perm[self.train_size:]
        self.g = torch.Generator()
        self.g.manual_seed(seed)

    def _iterator_seed(self):
        torch.empty((), dtype=torch.int64).random_(generator=self.g)

    def train_epoch(self):
        self._iterator_seed()
        perm = torch.randperm(self.train_size, generator=self.g)
        torch.randperm(self.train_size, generator=self.g)      # RandomSampler's remainder draw (sliced to length 0)
        return self.train_idx[perm]

    def val_epoch(self):
        self._iterator_seed()
        return self.val_idx


def _num_batches(n, bs):
    return (n + bs - 1) // bs


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_world_size(), dist.get_rank()
    return None, 1, 0


def train_attention_model(model, dataset, batch_size):
    """Reference model.py:209-384: config.txt, 80/20 split, AdamW + ReduceLROnPlateau + early stopping, progress
    prints and test-string dumps every 5 epochs, training_results.txt.  Returns the model."""
    from .parallel import DataParallelStepper, shard_rows
    dist, world, rank = _dist()
    eng = model.engine
    os.makedirs(OUTPUT_DIR, exist_ok=True)
    if rank == 0:
        with open(f"{OUTPUT_DIR}/config.txt", "w") as f:
            f.write("# Training configuration\n")
            for k, v in (("num_epochs", NUM_EPOCHS), ("learning_rate", LEARNING_RATE), ("batch_size", batch_size),
                         ("early_stopping_patience", EARLY_STOPPING_PATIENCE), ("validation_split", VALIDATION_SPLIT),
                         ("weight_decay", WEIGHT_DECAY), ("embedding_dim", EMBEDDING_DIM), ("dropout_rate", DROPOUT_RATE),
                         ("num_attention_heads", NUM_ATTENTION_HEADS), ("max_length", model.max_length),
                         ("max_chars_per_sheet", MAX_CHARS_PER_SHEET), ("num_samples", NUM_SAMPLES), ("data_size", len(dataset)),
                         ("random_seed", SEED), ("sheet_height", SHEET_HEIGHT), ("sheet_width", SHEET_WIDTH)):
                f.write(f"{k} = {v}\n")

    order = _EpochOrder(len(dataset))
    print(f"Dataset split: {order.train_size} training samples, {order.val_size} validation samples")

    # the whole dataset becomes HBM resident: codes int64 [N, L], sheets uint8 [N, H, W] when they are 8-bit exact
    inputs, targets = dataset.tensors
    inputs = inputs.to(device)
    t8 = helpers.targets_as_uint8(targets)
    targets = (t8 if t8 is not None else targets.to(torch.float32)).to(device)
    pixels = SHEET_HEIGHT * SHEET_WIDTH

    # ReduceLROnPlateau is host logic on one float; torch's own class drives it through a one-parameter stand-in
    lr_holder = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=LEARNING_RATE)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(lr_holder, mode="min", factor=SCHEDULER_FACTOR,
                                                           patience=SCHEDULER_PATIENCE, min_lr=MIN_LEARNING_RATE)
    stepper = DataParallelStepper(eng, dist, world)
    best_val_loss = float("inf")
    patience_counter = 0
    best_model_state = None
    epoch = -1
    for epoch in range(NUM_EPOCHS):
        model.train()
        lr = lr_holder.param_groups[0]["lr"]
        idx = order.train_epoch().to(device)
        nb = _num_batches(order.train_size, batch_size)
        for b in range(nb):
            rows = idx[b * batch_size:(b + 1) * batch_size]
            mine = rows[shard_rows(rows.numel(), rank, world)]
            stepper.step(inputs.index_select(0, mine), targets.index_select(0, mine), None, rows.numel() * pixels,
                         step=model._next_step(), lr=lr, betas=ADAM_BETAS, weight_decay=WEIGHT_DECAY)
        avg_train_loss = stepper.global_loss() / nb                 # mean of per-batch means (model.py:311,333)

        model.eval()
        vidx = order.val_epoch().to(device)
        nvb = _num_batches(order.val_size, batch_size)
        for b in range(nvb):
            rows = vidx[b * batch_size:(b + 1) * batch_size]
            mine = rows[shard_rows(rows.numel(), rank, world)]
            eng.forward(inputs.index_select(0, mine), training=False, want_output=False)
            eng.loss_grad(targets.index_select(0, mine), mean_elems=rows.numel() * pixels)
        avg_val_loss = stepper.global_loss() / max(nvb, 1)

        scheduler.step(avg_val_loss)
        is_best = avg_val_loss < best_val_loss
        if is_best:
            best_val_loss = avg_val_loss
            patience_counter = 0
            # The reference keeps `model.state_dict().copy()` (model.py:344): a shallow copy whose tensors alias the
            # live parameters, so "restoring the best state" later is a no-op.  Reproduced: keep aliases, not clones.
            best_model_state = dict(model.state_dict())
        else:
            patience_counter += 1

        if rank == 0:
            if epoch % 5 == 0:
                status = (f"Epoch {epoch}, Train Loss: {avg_train_loss:.6f}, Val Loss: {avg_val_loss:.6f}, "
                          f"LR: {lr_holder.param_groups[0]['lr']:.6f}")
                if is_best:
                    status += " (New Best)"
                print(status)
                render_strings(model, test_strings, output_dir=f"{OUTPUT_DIR}/epoch_{epoch}", sheet_height=SHEET_HEIGHT,
                               sheet_width=SHEET_WIDTH, device=device)
            elif is_best:
                print(f"Epoch {epoch}, New best validation loss: {avg_val_loss:.6f}")
        if patience_counter >= EARLY_STOPPING_PATIENCE:
            if rank == 0:
                print(f"Early stopping at epoch {epoch}, Best Val Loss: {best_val_loss:.6f}")
            model.load_state_dict(best_model_state)
            break

    if best_model_state is not None and patience_counter < EARLY_STOPPING_PATIENCE:
        model.load_state_dict(best_model_state)
        if rank == 0:
            print(f"Training completed, Best Val Loss: {best_val_loss:.6f}")

    if rank == 0:
        final_epoch = epoch + 1 if patience_counter < EARLY_STOPPING_PATIENCE else epoch
        with open(f"{OUTPUT_DIR}/training_results.txt", "w") as f:
            f.write("# Training Results\n")
            f.write(f"final_epoch = {final_epoch}\n")
            f.write(f"best_validation_loss = {best_val_loss:.6f}\n")
            f.write(f"final_learning_rate = {lr_holder.param_groups[0]['lr']:.6f}\n")
            f.write(f"early_stopped = {patience_counter >= EARLY_STOPPING_PATIENCE}\n")
            f.write(f"training_duration_epochs = {final_epoch}\n")
            f.write(f"training_completed = {datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
    return model


def train_string_renderer():
    """Reference model.py:389-421."""
    print("Creating sheet dataset...")
    dataset = load_string_dataset(data_dir="train_input", num_samples=NUM_SAMPLES, sheet_height=SHEET_HEIGHT,
                                  sheet_width=SHEET_WIDTH)
    print("Training attention-based sheet renderer with reduced embedding dimensions (32) and learned positional encoding...")
    batch_size = 1024                                              # the reference's GPU batch size (model.py:408-409)
    model = AttentionFontRenderer(max_length=MAX_CHARS_PER_SHEET, max_batch=batch_size)
    model = model.to(device)
    print(f"Using batch size {batch_size}")
    return train_attention_model(model, dataset, batch_size)


def main(argv=None):
    """The reference's __main__ block (model.py:425-454)."""
    argv = sys.argv if argv is None else argv
    _init_distributed()
    print(f"Using HIP device: {torch.cuda.get_device_name(device) if torch.cuda.is_available() else 'none'}")
    print(f"Device: {device}")
    os.makedirs(OUTPUT_DIR, exist_ok=True)
    if len(argv) > 1:
        if argv[1] == "--train":
            model = train_string_renderer()
            save_model(model)
            render_strings(model, test_strings, output_dir=OUTPUT_DIR, sheet_height=SHEET_HEIGHT, sheet_width=SHEET_WIDTH, device=device)
        else:
            print(f"Unknown option: {argv[1]}")
            print("Available options: --train")
            sys.exit(1)
    else:
        if os.path.exists(MODEL_FILENAME):
            model = load_model(AttentionFontRenderer, MAX_CHARS_PER_SHEET, device=device)
        else:
            print("No saved model found. Training a new model...")
            model = train_string_renderer()
            save_model(model)
        render_strings(model, test_strings, output_dir=OUTPUT_DIR, sheet_height=SHEET_HEIGHT, sheet_width=SHEET_WIDTH, device=device)
