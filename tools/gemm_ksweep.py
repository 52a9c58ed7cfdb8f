#!/usr/bin/env python3
"""Per-K-tile slope and fixed cost of afr_op_gemm at a given M x N: times K = 256 .. 4096, hot (back to back) and cold
(a 256 MiB buffer rewritten and A re-written between launches, single launches bracketed by events).
usage: python tools/gemm_ksweep.py M N [lay] [variant]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_font_renderer_amd import _lib  # noqa: E402

M, N = int(sys.argv[1]), int(sys.argv[2])
lay = sys.argv[3] if len(sys.argv) > 3 else "-"
variant = int(sys.argv[4]) if len(sys.argv) > 4 else None
lib = C.CDLL(_lib.LIB_PATH)
lib.afr_op_gemm.restype = C.c_int
lib.afr_op_gemm.argtypes = _lib.SIGNATURES["afr_op_gemm"][1]
if variant is not None:
    lib.afr_dbg_set_gemm_variant(variant)
ak, bk = "a" in lay, "b" in lay
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
flags = (_lib.GEMM_A_KSTRIDED if ak else 0) | (_lib.GEMM_B_KSTRIDED if bk else 0) | _lib.GEMM_OUT_BF16
rows = []
for K in (256, 512, 1024, 2048, 4096):
    A = (torch.rand((K, M) if ak else (M, K), device="cuda") - 0.5).to(torch.bfloat16)
    B = (torch.rand((K, N) if bk else (N, K), device="cuda") - 0.5).to(torch.bfloat16)
    Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def launch():
        rc = lib.afr_op_gemm(_lib.AFR_BF16, flags, p(A), p(B), p(Cb), None, None, M, N, K, M if ak else K, N if bk else K, N, N, 1, st)
        assert rc == 0, rc
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    hot = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        hot.append(e0.elapsed_time(e1) / 20 * 1e3)
    cold = []
    for r in range(12):
        junk.add_(1.0)
        A.add_(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    rows.append((K, float(np.median(hot)), float(np.median(cold))))
    print(f"K={K:5d}  hot {rows[-1][1]:7.1f} us   cold(single, +~2.6 us event cost) {rows[-1][2]:7.1f} us")
ks = np.array([r[0] for r in rows], dtype=float) / 64
for nm, col in (("hot", 1), ("cold", 2)):
    y = np.array([r[col] for r in rows])
    a, b = np.polyfit(ks, y, 1)
    print(f"{nm}: {a:.3f} us per K-tile + {b:.2f} us fixed   ({M}x{N} lay={lay})")
