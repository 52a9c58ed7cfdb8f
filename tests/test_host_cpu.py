"""CPU: host-side logic around the hot path -- data order, file formats, quantisation, CLI behaviour."""
import os
import subprocess
import sys

import numpy as np
import torch
import torch.utils.data as data

from .util import ROOT, load, synth


def test_epoch_order_reproduces_random_split_and_dataloader_shuffle():
    """_EpochOrder must yield the samples the reference's loaders would (model.py:232-266): same split, same
    per-epoch permutation, including the generator draws each DataLoader iterator makes."""
    from ai_font_renderer_amd.model import _EpochOrder
    n, bs = 103, 16
    ds = data.TensorDataset(torch.arange(n))
    val = int(0.2 * n)
    tr, va = data.random_split(ds, [n - val, val], generator=torch.Generator().manual_seed(42))
    g = torch.Generator()
    g.manual_seed(42)
    for workers in (0, 2):
        g.manual_seed(42)
        tl = data.DataLoader(tr, batch_size=bs, shuffle=True, generator=g, num_workers=workers)
        vl = data.DataLoader(va, batch_size=bs, shuffle=False, generator=g, num_workers=workers)
        order = _EpochOrder(n)
        for _ in range(3):
            got_t = torch.cat([b[0] for b in tl])
            got_v = torch.cat([b[0] for b in vl])
            assert torch.equal(got_t, order.train_epoch()), workers
            assert torch.equal(got_v, order.val_epoch()), workers


def test_helpers_match_reference_fixtures(tmp_path):
    """binary_array_to_image truncation (helpers.py:33) and image_to_binary_array (helpers.py:107-123) against
    outputs of the reference's own helpers (tests/golden/helpers.npz)."""
    from ai_font_renderer_amd import datagen, helpers
    fx = load("helpers.npz")
    img = helpers.binary_array_to_image(fx["arr"], output_path=str(tmp_path / "o" / "a.bmp"))
    assert np.array_equal(np.array(img), fx["arr_u8"])
    assert np.array_equal(helpers.image_to_u8_array(str(tmp_path / "o" / "a.bmp")), fx["arr_u8"])
    g = fx["gray"]
    p1, p2 = tmp_path / "1.bmp", tmp_path / "2.bmp"
    p1.write_bytes(datagen.bmp24_topdown(np.stack([g, g, g], -1)))
    p2.write_bytes(datagen.bmp24_topdown(fx["rgb"]))
    assert os.path.getsize(p1) == 57654
    assert np.array_equal(helpers.image_to_binary_array(str(p1)), fx["gray_f32"])
    assert np.array_equal(helpers.image_to_binary_array(str(p2)), fx["rgb_f32"])
    t8 = helpers.targets_as_uint8(torch.from_numpy(fx["gray_f32"]))
    assert t8 is not None and np.array_equal(t8.numpy(), g)
    assert helpers.targets_as_uint8(torch.rand(4, 4)) is None


def test_datagen_writes_the_reference_layout_and_loader_reads_it(tmp_path):
    from ai_font_renderer_amd import datagen, helpers
    texts = datagen.generate(str(tmp_path / "train_input"), 6)
    assert texts[0] == "P JAL WZ MQWPCDYYX EOGYVE MBANVV"
    assert (tmp_path / "train_input" / "data.txt").read_text().split("\n") == texts
    ds = helpers.load_string_dataset(str(tmp_path / "train_input"), num_samples=6)
    x, t = ds.tensors
    assert x.dtype == torch.int64 and t.dtype == torch.float32 and t.shape == (6, 80, 240)
    assert x.shape[1] == max(len(s) for s in texts) and int(x[0, 0]) == ord("P")
    assert 0.02 < float((t < 0.5).float().mean()) < 0.3                  # some ink on a white sheet
    ds8 = helpers.load_string_dataset_u8(str(tmp_path / "train_input"), num_samples=6)
    assert torch.equal(helpers.targets_as_uint8(t), ds8.tensors[1])
    try:
        helpers.load_string_dataset(str(tmp_path / "train_input"), num_samples=7)
        assert False
    except ValueError as e:                                              # reference helpers.py:149-150
        assert "Not enough strings" in str(e)
    os.remove(tmp_path / "train_input" / "3.bmp")
    try:
        helpers.load_string_dataset(str(tmp_path / "train_input"), num_samples=6)
        assert False
    except FileNotFoundError as e:                                       # reference helpers.py:156-157
        assert "Image file not found" in str(e)


def test_wrap_text_is_greedy_by_measured_width():
    from ai_font_renderer_amd.datagen import wrap_text
    assert wrap_text(len, "AA BBB C DDDD", 6) == ["AA BBB", "C DDDD"]
    assert wrap_text(len, "TOOLONGWORD X", 4) == ["TOOLONGWORD", "X"]
    assert wrap_text(len, "A  B", 10) == ["A  B"]


def test_cli_unknown_option_prints_usage_and_exits_1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "model.py"), "--bogus"], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 1
    assert r.stdout.strip().splitlines()[-2:] == ["Unknown option: --bogus", "Available options: --train"]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ai-font-renderer_amd/ (nor the root shims) may mention it."""
    pkg = os.path.join(ROOT, "ai-font-renderer_amd")
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(pkg) for f in fs if f.endswith((".py", ".hip", ".h", ".cpp"))]
    files += [os.path.join(ROOT, "model.py"), os.path.join(ROOT, "helpers.py")]
    for f in files:
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src and "afr_oracle" not in src, f


def test_engine_calls_run_on_the_engines_device_and_its_stream(monkeypatch):
    """An Engine on cuda:k must enqueue on cuda:k's current stream with cuda:k current, whatever the process's current
    device is (LOCAL_RANK != 0 without a set_device used to launch every kernel on GPU 0's stream).  No GPU needed: the
    guard and the stream lookup are observed through stand-ins."""
    import contextlib
    import ctypes as C
    from ai_font_renderer_amd import engine as E
    seen = []

    class FakeStream:
        def __init__(self, dev):
            self.cuda_stream = 0x5000 + (dev.index if dev is not None else 99)

    @contextlib.contextmanager
    def fake_device(dev):
        seen.append(("enter", dev))
        yield
        seen.append(("exit", dev))

    monkeypatch.setattr(torch.cuda, "device", fake_device)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: FakeStream(device))
    eng = E.Engine.__new__(E.Engine)
    eng.device = torch.device("cuda", 3)

    def fn(a, b, stream):
        seen.append(("call", a, b, stream.value))
        return 0

    eng._call(fn, 1, 2)
    assert seen == [("enter", eng.device), ("call", 1, 2, 0x5003), ("exit", eng.device)]
    eng.__dict__.clear()                                       # nothing for __del__ to tear down
