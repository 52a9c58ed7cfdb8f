"""Probe how hipExtStreamCreateWithCUMask maps mask bits to CUs on this part: time the R0 step's kernels on masked streams."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import torch
from ai_font_renderer_amd import synth
from ai_font_renderer_amd.config import WORKLOADS
from ai_font_renderer_amd.engine import Engine
import bench

hip = C.CDLL("libamdhip64.so")
def masked_stream(words):
    arr = (C.c_uint32 * len(words))(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

name = sys.argv[1] if len(sys.argv) > 1 else 'r0'
cfg, B = WORKLOADS[name]['cfg'], WORKLOADS[name]['batch']
eng = Engine(cfg, dtype='bf16', max_batch=B)
eng.load_params(synth.make_params(cfg))
x, font, t = bench.make_inputs(name, cfg, B, 0)
x, t = x.cuda(), t.cuda()
font = font.cuda() if font is not None else None
def run(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3): eng.train_step(x, t, font=font)
        torch.cuda.synchronize()
        eng.profile(1)
        for _ in range(3): eng.train_step(x, t, font=font)
        torch.cuda.synchronize()
        tab = eng.profile_table()
        eng.profile(0)
    print(label, ' | '.join(f"{r['kernel']} {1000 * r['avg_ms']:.0f}" for r in tab))
run(torch.cuda.current_stream(), 'default      ')
F = 0xFFFFFFFF
for label, words in [('all 256 bits ', [F] * 8), ('low 128 bits ', [F] * 4 + [0] * 4), ('high 128 bits', [0] * 4 + [F] * 4),
                     ('even bits    ', [0x55555555] * 8), ('low16 of 32  ', [0x0000FFFF] * 8), ('64 bits      ', [F, F, 0, 0, 0, 0, 0, 0]),
                     ('32 bits      ', [F, 0, 0, 0, 0, 0, 0, 0])]:
    run(masked_stream(words), label)
