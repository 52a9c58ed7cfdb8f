#!/usr/bin/env python3
"""Training-step throughput of BASELINE configs[4]'s model (config.C5: 64x64 pixel tokens, d_model 512, 4 blocks)
through the C ABI (afr_train_step: forward + MSE + backward + AdamW), with the per-kernel table of the plan's own profiler.
  python tools/c5_step_bench.py [batch=32] [dtype=bf16] [steps=10]
The per-GPU batch of BASELINE's config (16384 / 8 = 2048 glyphs) does not fit one launch sequence's saved activations
(12.6 GB per 32 glyphs in bf16 mode); the step is measured at the batch given and scales linearly in it (every product is M = B*4096 rows)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_font_renderer_amd import synth  # noqa: E402
from ai_font_renderer_amd.config import C5  # noqa: E402
from ai_font_renderer_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = Engine(C5, dtype=dtype, max_batch=B)
eng.load_params(synth.make_params(C5))
x = (32 + torch.arange(B) % 95).cuda()
font = (torch.arange(B) % 2).cuda()
g = torch.Generator().manual_seed(3)
tgt = torch.randint(0, 256, (B, C5.out_h, C5.out_w), dtype=torch.uint8, generator=g).cuda()
for _ in range(3):
    eng.train_step(x, tgt, font=font)
torch.cuda.synchronize()
print(f"workspace {eng.workspace_bytes / 2**30:.2f} GiB" if hasattr(eng, "workspace_bytes") else "", flush=True)
eng.profile(1)
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step(x, tgt, font=font)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = C5.train_flops_per_sample() * B
print(f"C5 train step, batch {B}, {dtype}: {dt * 1e3:.2f} ms = {B / dt:.0f} glyphs/s, {fl / dt / 1e15:.3f} PF of Linear FLOPs "
      f"({fl / dt / 1e15 / 2.5:.2f} of the bf16 MFMA peak); loss {eng.read_loss() / n:.5f}; error flags {eng.error_flags()}")
tab = eng.profile_table()
tot = sum(r["total_ms"] for r in tab)
for r in tab[:24]:
    print(f"  {r['kernel']:44s} n={r['launches']:4d} avg={r['avg_ms'] * 1e3:9.1f} us total={r['total_ms']:8.2f} ms ({100 * r['total_ms'] / tot:4.1f} %)")
