#!/usr/bin/env python3
"""Where a C3 (or R0) training step spends its time, from INSIDE its GEMM launches: per-workgroup wall-clock stamps of a
-DAFR_GEMM_TIMING build (s_memrealtime, 10 ns ticks; one 1024-workgroup region of the stamp buffer per launch).

  AFR_LIB_PATH=build_exp/timing/libafr.so python tools/step_timeline.py [workload] [flags]
Prints per GEMM launch of ONE steady-state step, relative to the step's first workgroup entry: first entry, last entry,
median prologue (entry -> first tile landed), median / max K loop, median tail (K loop end -> stores drained; for
cooperative split-K workgroups also park / wait / finish), last exit; and the gaps between launches (what the kernels without
stamps -- table, gather, first-layer backward, grouped reduce -- and the kernel boundaries take)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_font_renderer_amd import _lib, synth  # noqa: E402
from ai_font_renderer_amd.config import WORKLOADS  # noqa: E402
from ai_font_renderer_amd.engine import Engine  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg, B = WORKLOADS[name]["cfg"], WORKLOADS[name]["batch"]
lib = _lib.lib()
lib.afr_dbg_gemm_stamps.restype = C.c_int
lib.afr_dbg_gemm_stamps.argtypes = [C.c_void_p]
lib.afr_dbg_gemm_slot_reset.restype = None
eng = Engine(cfg, dtype="bf16", max_batch=B, flags=flags)
eng.load_params(synth.make_params(cfg))
x, font, tgt = bench.make_inputs(name, cfg, B, 0)
x, tgt = x.cuda(), tgt.cuda()
font = font.cuda() if font is not None else None
stamps = torch.zeros(64 * 1024 * 8, dtype=torch.int64, device="cuda")
assert lib.afr_dbg_gemm_stamps(C.c_void_p(stamps.data_ptr())) == 0
for _ in range(50):
    eng.train_step(x, tgt, font=font)
torch.cuda.synchronize()
for rep in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    for _ in range(4):                      # the LAST of four back-to-back steps is the one read (slots wrap at 64)
        lib.afr_dbg_gemm_slot_reset()
        eng.train_step(x, tgt, font=font)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(64, 1024, 8)
    us = lambda a: a * 0.01
    t0 = None
    prev_exit = None
    print(f"--- rep {rep}")
    for slot in range(64):
        r = s[slot]
        r = r[r[:, 0] > 0]
        if len(r) == 0:
            continue
        r = r[r[:, 3] >= r[:, 0]]           # complete records of this (the last) step only
        if t0 is None:
            t0 = r[:, 0].min()
        ent, land, kend, ex = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
        gap = us(ent.min() - prev_exit) if prev_exit is not None else 0.0
        line = (f"launch {slot}: {len(r):4d} wgs  gap {gap:6.2f} | entry {us(ent.min() - t0):7.2f}..{us(ent.max() - t0):7.2f}  prologue med {us(np.median(land - ent)):5.2f}"
                f"  K loop med {us(np.median(kend - land)):6.2f} max {us((kend - land).max()):6.2f}  tail med {us(np.median(ex - kend)):5.2f} max {us((ex - kend).max()):5.2f}"
                f"  exit {us(np.median(ex) - t0):7.2f}..{us(ex.max() - t0):7.2f}  span {us(ex.max() - ent.min()):6.2f}")
        print(line)
        co = r[r[:, 5] == 1]
        if len(co) and (co[:, 6] > 0).all():
            print(f"      cooperative wgs ({len(co)}): park med {us(np.median(co[:, 6] - co[:, 2])):5.2f} max {us((co[:, 6] - co[:, 2]).max()):5.2f}"
                  f"  wait med {us(np.median(co[:, 7] - co[:, 6])):5.2f} max {us((co[:, 7] - co[:, 6]).max()):5.2f}"
                  f"  finish med {us(np.median(co[:, 3] - co[:, 7])):5.2f} max {us((co[:, 3] - co[:, 7]).max()):5.2f}   exit {us(co[:, 3].max() - t0):7.2f}")
            ot = r[r[:, 5] != 1]
            if len(ot):
                print(f"      other wgs ({len(ot)}): K loop med {us(np.median(ot[:, 2] - ot[:, 1])):6.2f}  tail med {us(np.median(ot[:, 3] - ot[:, 2])):5.2f} max {us((ot[:, 3] - ot[:, 2]).max()):5.2f}   exit {us(ot[:, 3].max() - t0):7.2f}")
        prev_exit = ex.max()
