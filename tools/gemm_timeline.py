#!/usr/bin/env python3
"""Where one bf16 GEMM launch spends its time: per-block wall-clock stamps from a -DAFR_GEMM_TIMING build.

  AFR_LIB_PATH=build_exp/timing/libafr.so python tools/gemm_timeline.py M N K [lay: - a b ab] [splitk] [reps]
Prints, over all blocks of the launch (10 ns ticks -> us): dispatch skew (entry - first entry), prologue (entry ->
first tile landed), K loop, epilogue (loop end -> stores drained), and the launch's span first entry -> last exit.
The operands are re-written by an elementwise kernel before every timed launch when COLD=1 (so they come from
HBM / the infinity cache as in the training step, not from L2)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_font_renderer_amd import _lib  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
lay = sys.argv[4] if len(sys.argv) > 4 else "-"
splitk = int(sys.argv[5]) if len(sys.argv) > 5 else 1
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
cold = os.environ.get("COLD") == "1"
lib = C.CDLL(_lib.LIB_PATH)
lib.afr_op_gemm.restype = C.c_int
lib.afr_op_gemm.argtypes = _lib.SIGNATURES["afr_op_gemm"][1]
lib.afr_dbg_gemm_stamps.restype = C.c_int
lib.afr_dbg_gemm_stamps.argtypes = [C.c_void_p]
ak, bk = "a" in lay, "b" in lay
A = (torch.rand((K, M) if ak else (M, K), device="cuda") - 0.5).to(torch.bfloat16)
B = (torch.rand((K, N) if bk else (N, K), device="cuda") - 0.5).to(torch.bfloat16)
Cb = torch.empty(splitk, M, N, device="cuda", dtype=torch.float32 if splitk > 1 else torch.bfloat16)
flags = (_lib.GEMM_A_KSTRIDED if ak else 0) | (_lib.GEMM_B_KSTRIDED if bk else 0) | (_lib.GEMM_OUT_BF16 if splitk == 1 else 0)
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
assert lib.afr_dbg_gemm_stamps(C.c_void_p(stamps.data_ptr())) == 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")      # 256 MiB: evicts L2 and most of the infinity cache


def launch():
    rc = lib.afr_op_gemm(_lib.AFR_BF16, flags, p(A), p(B), p(Cb), None, None, M, N, K, M if ak else K, N if bk else K, N, N, splitk, st)
    assert rc == 0, rc


for _ in range(3):
    launch()
torch.cuda.synchronize()
rows = []
for r in range(reps):
    if cold:
        junk.add_(1.0)
        A.add_(0)      # re-written just before the launch, as the previous layer's kernel would have
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    us = lambda a: a * 0.01
    q = lambda a: f"min {us(a.min()):6.2f} med {us(np.median(a)):6.2f} max {us(a.max()):6.2f}"
    print(f"rep {r}: blocks {len(s)}  event {e0.elapsed_time(e1) * 1e3:7.1f} us  span {us(s[:, 3].max() - t0):6.2f} us")
    print(f"   entry skew   {q(s[:, 0] - t0)}")
    print(f"   prologue     {q(s[:, 1] - s[:, 0])}")
    print(f"   K loop       {q(s[:, 2] - s[:, 1])}")
    print(f"   epilogue     {q(s[:, 3] - s[:, 2])}")
    print(f"   exit - t0    {q(s[:, 3] - t0)}")
    if r == reps - 1:
        xcc = s[:, 4]
        print("   blocks per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
        cu = (s[:, 5] >> 8) & 0xF
        se = (s[:, 5] >> 13) & 0x7
        ids = xcc * 1000 + se * 16 + cu
        print("   distinct (xcc,se,cu):", len(np.unique(ids)))
nt = (K // splitk + 63) // 64
print(f"{M}x{N}x{K} lay={lay} sk={splitk}: {nt} K-tiles per block; 2MNK = {2.0 * M * N * K / 1e9:.2f} GFLOP (6.9 us at 2.5 PF for 17.2)")
