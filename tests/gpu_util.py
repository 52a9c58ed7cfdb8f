"""Helpers for the -m gpu parity tests: everything goes through the C ABI (ctypes), torch only holds memory."""
import ctypes as C

import numpy as np
import torch

from ai_font_renderer_amd import _lib


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(dtype, A_log, B_log, a_kstrided=False, b_kstrided=False, bias=None, relu=False, aux=None, out_bf16=False, splitk=1):
    """C[m][n] = sum_k A_log[m][k] * B_log[n][k] through afr_op_gemm, with the operands stored in the
    requested orientation.  dtype 'f32' | 'bf16'.  Returns a float32 CPU tensor."""
    lib = _lib.lib()
    M, K = A_log.shape
    N = B_log.shape[0]
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    A = dev(A_log.t().contiguous() if a_kstrided else A_log, tdt)
    B = dev(B_log.t().contiguous() if b_kstrided else B_log, tdt)
    flags = 0
    if a_kstrided:
        flags |= _lib.GEMM_A_KSTRIDED
    if b_kstrided:
        flags |= _lib.GEMM_B_KSTRIDED
    bias_d = None
    if bias is not None:
        flags |= _lib.GEMM_BIAS
        bias_d = dev(bias, torch.float32)
    if relu:
        flags |= _lib.GEMM_RELU
    aux_d = None
    if aux is not None:
        flags |= _lib.GEMM_RELU_MASK
        aux_d = dev(aux, tdt)
    if out_bf16:
        flags |= _lib.GEMM_OUT_BF16
    Cd = torch.full((splitk, M, N), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    _lib.check(lib.afr_op_gemm(_lib.AFR_F32 if dtype == "f32" else _lib.AFR_BF16, flags, ptr(A), ptr(B), ptr(Cd), ptr(bias_d),
                               ptr(aux_d), M, N, K, M if a_kstrided else K, N if b_kstrided else K, N, N, splitk, stream()))
    if splitk > 1:
        out = torch.empty(M, N, dtype=torch.float32, device="cuda")
        _lib.check(lib.afr_op_reduce(ptr(out), ptr(Cd), splitk, M * N, M * N, 1.0, 0, stream()))
        torch.cuda.synchronize()
        return out.cpu()
    torch.cuda.synchronize()
    return Cd[0].float().cpu()


def gemm_fix(A_log, B_log, head_tiles, splitk, a_kstrided=False, b_kstrided=False, bias=None, relu=False, aux=None, out_bf16=False,
             repeats=1):
    """The same product through afr_op_gemm_fix (bf16, 256x256 tiles, split-K reduced inside the launch).  Returns the
    float32 CPU result(s) of `repeats` launches on one workspace and the workspace's counter words after the last one."""
    lib = _lib.lib()
    M, K = A_log.shape
    N = B_log.shape[0]
    A = dev(A_log.t().contiguous() if a_kstrided else A_log, torch.bfloat16)
    B = dev(B_log.t().contiguous() if b_kstrided else B_log, torch.bfloat16)
    flags = (_lib.GEMM_A_KSTRIDED if a_kstrided else 0) | (_lib.GEMM_B_KSTRIDED if b_kstrided else 0)
    bias_d = aux_d = None
    if bias is not None:
        flags |= _lib.GEMM_BIAS
        bias_d = dev(bias, torch.float32)
    if relu:
        flags |= _lib.GEMM_RELU
    if aux is not None:
        flags |= _lib.GEMM_RELU_MASK
        aux_d = dev(aux, torch.bfloat16)
    if out_bf16:
        flags |= _lib.GEMM_OUT_BF16
    need = int(lib.afr_op_gemm_fix_workspace_bytes(M, N, head_tiles, splitk))
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(repeats):
        Cd = torch.full((M, N), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
        _lib.check(lib.afr_op_gemm_fix(flags, ptr(A), ptr(B), ptr(Cd), ptr(bias_d), ptr(aux_d), M, N, K, M if a_kstrided else K,
                                       N if b_kstrided else K, N, N, head_tiles, splitk, ptr(ws), need, stream()))
        torch.cuda.synchronize()
        outs.append(Cd.float().cpu())
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    tail = tiles - min(head_tiles, tiles)
    counters = ws[tail * splitk * 256 * 256 * 4:].view(torch.int32).cpu()
    return outs, counters


def gemm_pair(dy, x, W, aux=None, repeats=1):
    """A layer's gradient pair through afr_op_gemm_pair (ONE grouped 256x256 launch, cooperative split-K).  dy [B][N],
    x [B][K], W [N][K], aux [B][K] float tensors (rounded to bf16 here).  Returns the list of (dW f32 [N][K], db f32 [N],
    dX f32 [B][K]) CPU results of `repeats` launches on one workspace, and the split."""
    lib = _lib.lib()
    B, N = dy.shape
    K = x.shape[1]
    sk, need = C.c_int(), C.c_size_t()
    _lib.check(lib.afr_op_gemm_pair_plan(B, N, K, C.byref(sk), C.byref(need)))
    dyd, xd, Wd = dev(dy, torch.bfloat16), dev(x, torch.bfloat16), dev(W, torch.bfloat16)
    auxd = dev(aux, torch.bfloat16) if aux is not None else None
    ws = torch.full((need.value,), 0x5A, dtype=torch.uint8, device="cuda")     # dirty on purpose: the op owns its state
    outs = []
    for _ in range(repeats):
        dW = torch.full((N, K), float("nan"), dtype=torch.float32, device="cuda")
        dbp = torch.full((sk.value, N), float("nan"), dtype=torch.float32, device="cuda")
        dX = torch.full((B, K), float("nan"), dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.afr_op_gemm_pair(ptr(dyd), ptr(xd), ptr(Wd), ptr(auxd), ptr(dW), ptr(dbp), ptr(dX), B, N, K, ptr(ws), need.value, stream()))
        torch.cuda.synchronize()
        outs.append((dW.cpu(), dbp.sum(0).cpu(), dX.float().cpu()))
    return outs, sk.value
