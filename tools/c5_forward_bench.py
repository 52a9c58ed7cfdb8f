#!/usr/bin/env python3
"""Forward throughput of BASELINE configs[4]'s model (config.C5: 64x64 pixel tokens, d_model 512, 4 blocks) through
the C ABI -- forward only (the training step: tools/c5_step_bench.py).
  python tools/c5_forward_bench.py [batch=32] [dtype=bf16]"""
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
eng = Engine(C5, dtype=dtype, max_batch=B, with_optimizer=False)
eng.load_params(synth.make_params(C5))
x = (32 + torch.arange(B) % 95).cuda()
font = (torch.arange(B) % 2).cuda()
for _ in range(3):
    eng.forward(x, font, want_output=False)
torch.cuda.synchronize()
eng.profile(1)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    eng.forward(x, font, want_output=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = C5.train_flops_per_sample() / 3.0 * B
print(f"C5 forward, batch {B}, {dtype}: {dt * 1e3:.2f} ms = {B / dt:.0f} glyphs/s forward, {fl / dt / 1e15:.3f} PF of Linear FLOPs "
      f"({fl / dt / 1e15 / 2.5:.2f} of the bf16 MFMA peak)")
for r in eng.profile_table()[:8]:
    print(f"  {r['kernel']:40s} n={r['launches']:4d} avg={r['avg_ms'] * 1e3:9.1f} us total={r['total_ms']:8.2f} ms")
