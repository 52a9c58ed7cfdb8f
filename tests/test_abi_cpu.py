"""CPU: the C-ABI library loads and exports every symbol include/afr.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from .util import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "afr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(afr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ai_font_renderer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = _header_functions()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/afr.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib().afr_version() == 1


def test_plan_layout_matches_python_layout_without_gpu():
    """Plan creation is host-only: the flat layout must equal config.flat_layout (checkpoint contract)."""
    from ai_font_renderer_amd import _lib, config
    from ai_font_renderer_amd.engine import make_afr_config
    lib = _lib.lib()
    for cfg in (config.SheetConfig(), config.SheetConfig(max_length=10, sheet_h=8, sheet_w=24),
                config.WORKLOADS["c1"]["cfg"], config.WORKLOADS["c3"]["cfg"]):
        c = make_afr_config(cfg, "f32", 64)
        plan = C.c_void_p()
        _lib.check(lib.afr_plan_create(C.byref(c), C.byref(plan)))
        table, total = config.flat_layout(cfg)
        assert lib.afr_param_elems(plan) == total
        assert lib.afr_param_count(plan) == len(table)
        name = C.create_string_buffer(128)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int32()
        shape = (C.c_int64 * 4)()
        for i, (nm, shp, o, n) in enumerate(table):
            _lib.check(lib.afr_param_info(plan, i, name, 128, C.byref(off), C.byref(numel), C.byref(ndim), shape))
            assert (name.value.decode(), off.value, numel.value) == (nm, o, n)
            assert tuple(shape[k] for k in range(ndim.value)) == tuple(shp)
        assert lib.afr_workspace_bytes(plan) > 0
        lib.afr_plan_destroy(plan)


def test_bad_configs_are_rejected_with_a_message():
    from ai_font_renderer_amd import _lib, config
    from ai_font_renderer_amd.engine import make_afr_config
    lib = _lib.lib()
    plan = C.c_void_p()
    c = make_afr_config(config.GlyphConfig(hidden=(30,)), "f32", 8)      # width not a multiple of 8
    assert lib.afr_plan_create(C.byref(c), C.byref(plan)) < 0
    assert b"multiple" in lib.afr_last_error()
    c = make_afr_config(config.SheetConfig(max_length=500), "f32", 8)
    assert lib.afr_plan_create(C.byref(c), C.byref(plan)) < 0
    with pytest.raises(_lib.AfrError):
        _lib.check(lib.afr_plan_create(C.byref(c), C.byref(plan)))


def test_operands_of_2gib_or_more_are_rejected_not_silently_zero_filled():
    """The bf16 LDS-DMA path addresses an operand with 32-bit byte offsets (gemm.hip make_rsrc / stage_inst): an operand of
    2 GiB or more would read zeros past the limit.  Argument validation only -- nothing is launched."""
    from ai_font_renderer_amd import _lib, config
    from ai_font_renderer_amd.engine import make_afr_config
    lib = _lib.lib()
    fake = C.c_void_p(0x1000)
    M, K = 32768, 32768                                           # 32768^2 bf16 = 2 GiB
    rc = lib.afr_op_gemm(_lib.AFR_BF16, 0, fake, fake, fake, None, None, M, 1024, K, K, K, 1024, 0, 1, None)
    assert rc == -4 and b"2 GiB" in lib.afr_last_error()          # AFR_EUNSUPPORTED
    plan = C.c_void_p()
    c = make_afr_config(config.SheetConfig(), "bf16", 60000)      # 60000 x 19200 bf16 activations = 2.3 GB
    assert lib.afr_plan_create(C.byref(c), C.byref(plan)) == -4
    assert b"2 GiB" in lib.afr_last_error()
    c = make_afr_config(config.SheetConfig(), "f32", 60000)       # the f32 kernels index with 64-bit arithmetic
    assert lib.afr_plan_create(C.byref(c), C.byref(plan)) == 0
    lib.afr_plan_destroy(plan)


def test_grouped_reduce_refuses_more_segments_than_it_can_hold():
    """afr_rtable_add used to drop the segment past its capacity silently (a gradient would have gone missing); now it is an error."""
    from ai_font_renderer_amd import _lib
    lib = _lib.lib()
    n = 33
    ptrs = (C.c_void_p * n)(*[0x1000] * n)
    ns = (C.c_int * n)(*[2] * n)
    st = (C.c_int64 * n)(*[64] * n)
    ln = (C.c_int64 * n)(*[64] * n)
    assert lib.afr_op_reduce_group(n, ptrs, ptrs, ns, st, ln, None) == -1      # AFR_EINVAL
    assert b"at most 32" in lib.afr_last_error()
