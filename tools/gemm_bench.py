#!/usr/bin/env python3
"""Micro-benchmark of afr_op_gemm (one process, interleaved rounds).  AFR_LIBS=lib1.so,lib2.so compares builds.
usage: python tools/gemm_bench.py M N K [dtype] [flags: a b for k-strided A/B] [splitk]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_font_renderer_amd import _lib  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
lay = sys.argv[5] if len(sys.argv) > 5 else "-"
splitk = int(sys.argv[6]) if len(sys.argv) > 6 else 1
libs = os.environ.get("AFR_LIBS", _lib.LIB_PATH).split(",")
variants = [int(v) for v in os.environ.get("VARIANTS", "").split(",") if v]       # -DAFR_GEMM_LAB builds: afr_dbg_set_gemm_variant
tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
ak, bk = "a" in lay, "b" in lay
A = (torch.rand((K, M) if ak else (M, K), device="cuda") - 0.5).to(tdt)
B = (torch.rand((K, N) if bk else (N, K), device="cuda") - 0.5).to(tdt)
Cb = torch.empty(splitk, M, N, device="cuda", dtype=torch.float32 if splitk > 1 else tdt)
flags = (_lib.GEMM_A_KSTRIDED if ak else 0) | (_lib.GEMM_B_KSTRIDED if bk else 0) | (_lib.GEMM_OUT_BF16 if (dtype == "bf16" and splitk == 1) else 0)
handles = []
for path in libs:
    lib = C.CDLL(path)
    lib.afr_op_gemm.restype = C.c_int
    lib.afr_op_gemm.argtypes = _lib.SIGNATURES["afr_op_gemm"][1]
    nm = os.path.basename(os.path.dirname(path)) + "/" + os.path.basename(path)
    if variants:
        for v in variants:
            handles.append((f"{nm}#v{v}", lib, v))
    else:
        handles.append((nm, lib, None))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def run(lib, n, v=None):
    if v is not None:
        lib.afr_dbg_set_gemm_variant(v)
    for _ in range(n):
        rc = lib.afr_op_gemm(_lib.AFR_BF16 if dtype == "bf16" else _lib.AFR_F32, flags, p(A), p(B), p(Cb), None, None, M, N, K,
                             M if ak else K, N if bk else K, N, N, splitk, st)
        assert rc == 0, rc


res = {n: [] for n, _, _ in handles}
for n, lib, v in handles:
    run(lib, 5, v)
torch.cuda.synchronize()
for rnd in range(5):
    for n, lib, v in handles:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(lib, 20, v)
        e1.record()
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
for n in res:
    us = sorted(res[n])[len(res[n]) // 2]
    print(f"{n:40s} {M}x{N}x{K} {dtype} lay={lay} sk={splitk}: median {us:8.1f} us  min {min(res[n]):8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s")
