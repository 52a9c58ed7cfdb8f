"""GPU: single-kernel parity through the C ABI (afr_op_*) against fp64 CPU references."""
import ctypes as C
import itertools

import numpy as np
import pytest
import torch

from .util import oracle, synth

pytestmark = pytest.mark.gpu


def _rand(tid, shape, bound=1.0):
    return torch.from_numpy(synth.hash_uniform(tid, shape, bound))


def _b16(t):
    return t.to(torch.bfloat16).to(torch.float32)


SHAPES = [(128, 128, 64), (200, 136, 96), (95, 256, 32), (300, 24, 640), (8, 1024, 1024 + 64)]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("ak,bk", list(itertools.product([False, True], repeat=2)))
def test_gemm_all_orientations_and_ragged_shapes(dtype, ak, bk):
    from .gpu_util import gemm
    for si, (M, N, K) in enumerate(SHAPES):
        if ak and M % 8:
            M = (M + 7) // 8 * 8           # a k-strided operand's contiguous extent must be a multiple of 8
        A, B = _rand(10 + si, (M, K)), _rand(20 + si, (N, K))
        if dtype == "bf16":
            A, B = _b16(A), _b16(B)
        ref = A.double() @ B.double().t()
        got = gemm(dtype, A, B, ak, bk)
        tol = 2e-6 * K ** 0.5 + 1e-6
        assert float((got.double() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), (dtype, ak, bk, M, N, K)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_identity_with_asymmetric_operand(dtype):
    """A = I against an asymmetric B exposes a transposed C write (MFMA C/D layout check)."""
    from .gpu_util import gemm
    n = 128
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125      # exact in bf16
    for ak, bk in itertools.product([False, True], repeat=2):
        got = gemm(dtype, torch.eye(n), B, ak, bk)
        assert torch.equal(got, B.t().contiguous()), (dtype, ak, bk)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_epilogues_and_splitk(dtype):
    from .gpu_util import gemm
    M, N, K = 264, 136, 320
    A, B = _rand(31, (M, K)), _rand(32, (N, K), 0.2)
    bias, aux = _rand(33, (N,)), _rand(34, (M, N))
    if dtype == "bf16":
        A, B, aux = _b16(A), _b16(B), _b16(aux)
    ref = A.double() @ B.double().t()
    scale = float(ref.abs().max())
    got = gemm(dtype, A, B, bias=bias, relu=True)
    assert float((got.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-4 * scale
    got = gemm(dtype, A, B, b_kstrided=True, aux=aux)
    assert float((got.double() - ref * (aux > 0)).abs().max()) < 1e-4 * scale
    got = gemm(dtype, A, B, bias=bias, relu=True, out_bf16=True)
    assert float((got.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-2 * scale
    for sk in (2, 3, 5):
        got = gemm(dtype, A, B, a_kstrided=True, b_kstrided=True, splitk=sk)
        assert float((got.double() - ref).abs().max()) < 1e-4 * scale, sk


def test_bf16_gemm_with_a_thousand_256x256_tiles_takes_the_256x256_body_and_matches_fp64():
    """Plain bf16 products whose 256x256 tiles make four or more rounds of the chip run on the 256x256 body (gemm.hip
    bf16_use_body256; the pixel transformer's 131072-row products): every layout the training path uses there, ragged in M,
    with the bias / ReLU / bf16-output / ReLU-gate epilogues, against fp64."""
    from .gpu_util import gemm
    M, N, K = 65536 + 40, 1024, 256                     # 257 x 4 tiles
    A, B = _b16(_rand(61, (M, K))), _b16(_rand(62, (N, K), 0.2))
    bias, aux = _rand(63, (N,)), _b16(_rand(64, (M, N)))
    ref = A.double() @ B.double().t()
    scale = float(ref.abs().max())
    got = gemm("bf16", A, B)
    assert float((got.double() - ref).abs().max()) < 1e-5 * scale
    got = gemm("bf16", A, B, bias=bias, relu=True, out_bf16=True)
    assert float((got.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-2 * scale
    got = gemm("bf16", A, B, b_kstrided=True, aux=aux, out_bf16=True)
    assert float((got.double() - ref * (aux > 0)).abs().max()) < 1e-2 * scale
    del ref, got, aux
    # the forward layout with a long K and few columns (fc2: 2048 -> 512)
    M, N, K = 131072, 512, 512
    A, B = _b16(_rand(65, (M, K))), _b16(_rand(66, (N, K), 0.2))
    ref = A.double() @ B.double().t()
    got = gemm("bf16", A, B, bias=_rand(67, (N,)), out_bf16=True)
    assert float((got.double() - (ref + _rand(67, (N,)).double())).abs().max()) < 1e-2 * float(ref.abs().max())


def test_f32_gemm_is_a_k_ordered_fma_chain_close_to_fp64():
    """Parity mode: exact-f32 MFMA, error ~1e-7 * sum|a b| (the 1e-4 bitmap bar needs K=6400 products)."""
    from .gpu_util import gemm
    A, B = _rand(41, (64, 6400)), _rand(42, (192, 6400), 0.0125)
    ref = A.double() @ B.double().t()
    got = gemm("f32", A, B)
    assert float((got.double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("act", ["f32", "bf16"])
@pytest.mark.parametrize("tgt", ["u8", "f32"])
def test_mse_grad_matches_oracle(act, tgt):
    from ai_font_renderer_amd import _lib
    from .gpu_util import dev, ptr, stream
    rows, cols = 37, 19200
    u = _rand(51, (rows, cols), 1.5) + 0.3
    u.view(-1)[:4] = torch.tensor([0.0, 1.0, -0.0, 1.0000001])          # inclusive clamp boundaries
    tu8 = synth.hash_u8(52, (rows, cols))
    tf = torch.from_numpy(tu8.astype(np.float32) / 255.0)
    adt = torch.float32 if act == "f32" else torch.bfloat16
    ud = dev(u, adt)
    uref = ud.float().cpu()
    loss_ref, du_ref = oracle.mse_loss_grad(uref.double(), tf.double(), total_elems=rows * cols * 3)
    td = dev(torch.from_numpy(tu8)) if tgt == "u8" else dev(tf)
    loss = torch.zeros(1, device="cuda")
    scratch = torch.zeros(1040, device="cuda")
    du = torch.empty_like(ud)
    _lib.check(_lib.lib().afr_op_mse_grad(_lib.AFR_F32 if act == "f32" else _lib.AFR_BF16, ptr(ud), ptr(td),
                                          _lib.AFR_TARGET_U8 if tgt == "u8" else _lib.AFR_TARGET_F32, ptr(du), rows, cols,
                                          rows * cols * 3, ptr(loss), ptr(scratch), stream()))
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(loss_ref)) < 1e-5 * float(loss_ref) + 1e-9
    tol = 1e-6 if act == "f32" else 1e-2
    assert float((du.float().cpu().double() - du_ref).abs().max()) <= tol * float(du_ref.abs().max())
    # zero gradient exactly where the clamp is active
    assert torch.equal(du.float().cpu() == 0, (du_ref == 0) | (du.float().cpu() == 0))
    assert float(du.float().cpu()[(uref < 0) | (uref > 1)].abs().max()) == 0.0


def test_adamw_three_steps_match_torch_optim():
    """afr_op_adamw vs torch.optim.AdamW itself (the reference's optimizer, model.py:273) on CPU."""
    from ai_font_renderer_amd import _lib
    from .gpu_util import dev, ptr, stream
    n = 64 * 1000
    p0 = _rand(61, (n,), 0.5)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=5e-4, betas=(0.9, 0.99))
    pd, md, vd = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for t in (1, 2, 3):
        g = _rand(70 + t, (n,), 0.01)
        p.grad = g.clone()
        opt.step()
        _lib.check(_lib.lib().afr_op_adamw(ptr(pd), ptr(dev(g)), ptr(md), ptr(vd), ptr(shadow), n, 1e-3, 0.9, 0.99, 1e-8, 5e-4, t,
                                           1.0, stream()))
    torch.cuda.synchronize()
    assert float((pd.cpu() - p.detach()).abs().max()) < 2e-6
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))


def test_reduce_and_convert():
    from ai_font_renderer_amd import _lib
    from .gpu_util import dev, ptr, stream
    s = _rand(81, (7, 1003))
    out = dev(torch.ones(1003))
    _lib.check(_lib.lib().afr_op_reduce(ptr(out), ptr(dev(s)), 7, 1003, 1003, 0.5, 1, stream()))
    torch.cuda.synchronize()
    ref = torch.ones(1003)
    acc = torch.zeros(1003)
    for i in range(7):
        acc = acc + s[i]
    assert torch.allclose(out.cpu(), ref + acc * 0.5, atol=1e-6)
    b = torch.empty(7 * 1003, dtype=torch.bfloat16, device="cuda")
    _lib.check(_lib.lib().afr_op_f32_to_bf16(ptr(dev(s)), ptr(b), 7 * 1003, stream()))
    torch.cuda.synchronize()
    assert torch.equal(b.cpu(), s.reshape(-1).to(torch.bfloat16))
