#!/usr/bin/env python3
"""Host issue time vs wall time per training step (is the launch stream host-bound?): python tools/hostbound.py"""
import sys, time, torch
sys.path.insert(0, '.')
from ai_font_renderer_amd import synth
from ai_font_renderer_amd.config import WORKLOADS
from ai_font_renderer_amd.engine import Engine
import bench
for name in ('c3', 'c2', 'c1'):
    cfg, B = WORKLOADS[name]['cfg'], WORKLOADS[name]['batch']
    eng = Engine(cfg, dtype=bench.DEFAULT_DTYPE[name], max_batch=B)
    eng.load_params(synth.make_params(cfg))
    x, font, t = bench.make_inputs(name, cfg, B, 0)
    x, t = x.cuda(), t.cuda(); font = font.cuda() if font is not None else None
    for _ in range(20): eng.train_step(x, t, font=font)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): eng.train_step(x, t, font=font)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: host issue {1e6*(t1-t0)/200:.1f} us/step, wall {1e6*(t2-t0)/200:.1f} us/step')
