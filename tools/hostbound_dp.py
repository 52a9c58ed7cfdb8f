import os, sys, time, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from ai_font_renderer_amd import synth
from ai_font_renderer_amd.config import WORKLOADS
from ai_font_renderer_amd.engine import Engine
from ai_font_renderer_amd.parallel import DataParallelStepper
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
name = 'c3'
cfg, B = WORKLOADS[name]['cfg'], WORKLOADS[name]['batch']
eng = Engine(cfg, dtype='bf16', max_batch=B)
eng.load_params(synth.make_params(cfg))
x, font, t = bench.make_inputs(name, cfg, B, 0)
x, t, font = x.cuda(), t.cuda(), font.cuda()
st = DataParallelStepper(eng, dist, 2)
me = B * cfg.pixels
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6*(t1-t0)/n, 1e6*(t2-t0)/n
print('dp path  host/wall us', timeit(lambda: st.step(x, t, font, me)))
def nocomm():
    eng.forward_loss(x, t, font=font, mean_elems=me)
    for s in range(eng.backward_stages): eng.backward_stage(s)
    eng.adamw_step()
print('staged-no-collectives host/wall us', timeit(nocomm))
def onear():
    eng.forward_loss(x, t, font=font, mean_elems=me)
    for s in range(eng.backward_stages): eng.backward_stage(s)
    dist.all_reduce(eng.flat_grads)
    eng.adamw_step()
print('staged-one-blocking-allreduce host/wall us', timeit(onear))
def two_blocking():
    eng.forward_loss(x, t, font=font, mean_elems=me)
    first = eng.backward_stage(0)
    dist.all_reduce(first)
    for s in range(1, eng.backward_stages): eng.backward_stage(s)
    dist.all_reduce(eng.flat_grads[:first.storage_offset()])
    eng.adamw_step()
print('staged-two-blocking-allreduces host/wall us', timeit(two_blocking))
def two_async_late_wait():
    eng.forward_loss(x, t, font=font, mean_elems=me)
    first = eng.backward_stage(0)
    w1 = dist.all_reduce(first, async_op=True)
    for s in range(1, eng.backward_stages): eng.backward_stage(s)
    w2 = dist.all_reduce(eng.flat_grads[:first.storage_offset()], async_op=True)
    w1.wait(); w2.wait()
    eng.adamw_step()
print('staged-two-async host/wall us', timeit(two_async_late_wait))
comm = torch.cuda.Stream()
def two_side_stream():
    cur = torch.cuda.current_stream()
    eng.forward_loss(x, t, font=font, mean_elems=me)
    first = eng.backward_stage(0)
    ev = torch.cuda.Event(); ev.record(cur)
    with torch.cuda.stream(comm):
        comm.wait_event(ev)
        dist.all_reduce(first)                     # "blocking" only for the side stream
    for s in range(1, eng.backward_stages): eng.backward_stage(s)
    dist.all_reduce(eng.flat_grads[:first.storage_offset()])
    cur.wait_stream(comm)
    eng.adamw_step()
print('staged-two-collectives-side-stream host/wall us', timeit(two_side_stream))
def mono():
    eng.train_step(x, t, font=font, mean_elems=me, do_step=False)
    dist.all_reduce(eng.flat_grads)
    eng.adamw_step()
print('monolithic-backward-one-allreduce host/wall us', timeit(mono))
print('fused-single host/wall us', timeit(lambda: eng.train_step(x, t, font=font)))
dist.destroy_process_group()
