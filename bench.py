#!/usr/bin/env python3
"""bench.py -- glyphs/sec of one full training step (forward + MSE + backward + AdamW) of the hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|r0|c5] [--dtype bf16|f32]
         (run directly with N > 1 it starts its own N rank processes before touching a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL)

Workloads (BASELINE.json configs) -- per-GPU batch is fixed, so scaling is WEAK:
  c3 (default)  32x32 glyphs, Emb+FontEmb -> 1024 -> 1024 -> 1024, bf16, 8192 glyphs per GPU.  This is
                configs[2], the largest single-GPU configuration, and the per-GPU shard of configs[3]
                (65536 glyphs over 8 GPUs), the configuration the metric's 1/2/4/8 scaling is quoted on.
  c2            16x16 glyphs, hidden 256, bf16, 4096 glyphs (configs[1]; launch-latency bound)
  c1            same net, fp32, 95 glyphs (configs[0], the CPU-runnable case)
  r0            the reference's own AttentionFontRenderer (80x240 sheets of <=100 chars), 1024 sheets per GPU
  c5            configs[4]'s per-pixel-token transformer as DESIGN.md 8 defines it (64x64 tokens, d_model 512, 4 blocks), bf16
                operands, ONE micro-batch of 32 glyphs (131072 token rows) per step -- not the fp8 / 2048-per-GPU form of the config
Before the W warm-up steps the GPU is pre-heated with inference forwards of the same model (--preheat-ms, default 30 ms of
host time; reported as "preheat_ms"): the part needs ~20 ms of sustained work to reach its operating point.
Inputs are synthetic and resident in HBM before the timed region: the 95 printable ASCII codes (x font ids) repeated over
the batch, with the FreeType rasterisations of FiraCode-Retina / Montserrat-Regular as uint8 targets (tests/golden/
glyph_bitmaps.npz); R0: strings from the seeded text generator, hashed sheets.  A step = the loop body of reference model.py:292-310.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ai_font_renderer_amd import synth  # noqa: E402
from ai_font_renderer_amd.config import WORKLOADS, PixelConfig, SheetConfig  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}        # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0
DESCR = {
    "c3": "C3: 32x32 glyph MLP 32->1024->1024->1024, char+font embedding, batch 8192/GPU (BASELINE configs[2]; per-GPU shard of configs[3])",
    "c2": "C2: 16x16 glyph MLP 32->256->256, batch 4096 (BASELINE configs[1])",
    "c1": "C1: 16x16 glyph MLP 32->256->256, fp32, batch 95 (BASELINE configs[0])",
    "r0": "R0: reference AttentionFontRenderer, 100 chars -> 80x240 sheet, batch 1024/GPU",
    "c5": "C5 micro-batch: 64x64 per-pixel-token transformer (d_model 512, 8 heads, 4 blocks, ff 2048; DESIGN.md 8), bf16 operands, 32 glyphs = 131072 tokens per micro-step (BASELINE configs[4] asks fp8 and 2048 glyphs/GPU: --batch 2048 runs that shard as 64 accumulated micro-steps)",
}
DEFAULT_DTYPE = {"c3": "bf16", "c2": "bf16", "c1": "f32", "r0": "bf16", "c5": "bf16"}
DEFAULT_STEPS = {"c3": (200, 20), "c2": (500, 50), "c1": (500, 50), "r0": (30, 5), "c5": (20, 3)}


def make_inputs(name, cfg, B, rank):
    """Synthetic batch for this rank: rows [rank*B, (rank+1)*B) of the global batch."""
    if isinstance(cfg, SheetConfig):
        strings = synth.dataset_strings(B, first_seed=42 + rank * B)
        x = synth.encode_strings(strings, cfg.max_length)
        t = synth.synth_sheet_targets(B, cfg.sheet_h, cfg.sheet_w, tensor_id=940 + rank)
        return torch.from_numpy(x), None, torch.from_numpy(t)
    i = np.arange(rank * B, (rank + 1) * B)
    x = (32 + (i % 95)).astype(np.int64)                              # the 95 printable ASCII codes, repeated
    font = ((i // 95) % max(cfg.n_fonts, 1)).astype(np.int64)
    t = synth.glyph_bitmap_targets(cfg.out_h, x, font) if cfg.out_h == cfg.out_w else None     # FiraCode (+ Montserrat) glyphs
    if t is None and cfg.out_h == cfg.out_w == 64:
        t32 = synth.glyph_bitmap_targets(32, x, font)                 # 64x64: the 32x32 rasterisations, every pixel doubled
        t = np.repeat(np.repeat(t32, 2, axis=1), 2, axis=2) if t32 is not None else None
    if t is None:
        t = synth.hash_u8(950 + rank, (B, cfg.out_h, cfg.out_w))
    return torch.from_numpy(x), (torch.from_numpy(font) if cfg.n_fonts > 0 else None), torch.from_numpy(t)


def host_cores():
    """CPU cores this process may really use: cgroup quota, else affinity mask (a GPU box gives each GPU a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("AFR_CPU_THREADS", min(n, 16)))


def reference_bitmap_diff(cfg, sample):
    """The metric's second half: max-abs difference between the bitmaps the engine drew (sample["y"], its benchmarked dtype)
    and the unrounded f32 CPU reference path on the SAME f32 master weights and inputs (the oracle: checker only)."""
    from oracle import afr_oracle as oracle            # checker / baseline only -- never on the product path
    P = sample["params"]
    if isinstance(cfg, SheetConfig):
        yref, _ = oracle.sheet_forward(P, sample["x"], cfg)
    elif isinstance(cfg, PixelConfig):
        yref, _ = oracle.pixel_forward(P, sample["x"], sample["font"], cfg)
    else:
        yref, _ = oracle.glyph_forward(P, sample["x"], sample["font"], cfg)
    return float((sample["y"].reshape(yref.shape) - yref).abs().max())


def cpu_baseline(name, cfg, B, budget_s=12.0):
    """The oracle (a CPU port of the reference's step in plain torch ops) timed on this box's host cores."""
    from oracle import afr_oracle as oracle            # checker / baseline only -- never on the product path
    cores = host_cores()
    torch.set_num_threads(cores)
    if isinstance(cfg, PixelConfig):
        B = min(B, 2)                                  # 258 GFLOP per glyph and step: a bounded sample of the same step
    x, font, t = make_inputs(name, cfg, B, 0)
    tgt = t.to(torch.float32) / 255.0
    P = {k: torch.from_numpy(v) for k, v in synth.make_params(cfg).items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    masks = None
    if isinstance(cfg, SheetConfig):
        masks = {k: torch.from_numpy(v) for k, v in synth.sheet_dropout_masks(cfg, B, cfg.max_length, 42, 1).items()}
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        _, _, P, M, V = oracle.train_step(P, M, V, step + 1, x, tgt, cfg, font=font, masks=masks, inplace=True)
        dt = time.perf_counter() - t0
        step += 1
        if step > 1:
            times.append(dt)                                          # first step is warm-up
        if (time.perf_counter() - t_start > budget_s and len(times) >= 2) or len(times) >= 400:
            break                                                     # ~12 s of CPU work (at least 2 timed steps)
    med = float(np.median(times))
    return {"value": B / med, "unit": "glyphs/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} steps of batch {B} after 1 warm-up, median step {med * 1e3:.1f} ms, fp32, torch {torch.__version__} CPU ops",
            "provenance": "oracle.train_step (in-place AdamW); in the build container its step takes 0.87-0.96x the imported reference's "
                          "own training step on R0 shapes (the reference also draws its dropout masks, ~27 % of its step; 1.2x "
                          "against the reference with dropout off) and 1.02x a torch.nn/autograd twin on C3: "
                          "tests/golden/cpu_step_times.json, made by tests/golden/make_golden.py cpu_step_times"}


PROFILE_ROUND = "r03"      # the round whose committed PMC passes describe THIS code (profiles/<round>/pmc_traffic.json)


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from THIS round's committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
    separately on this same command; FETCH doubled per the gfx950 rule; tools/pmc_summary.py).  No file for this round:
    (None, reason) -- never an older round's numbers.  File present but the kernel missing from it: an error (the engine
    reports "gemm_bf16<1,1,2>", rocprofv3 the full template argument list "gemm_bf16<1,1,2,0,0>": matched on that prefix)."""
    rel = f"profiles/{PROFILE_ROUND}/pmc_traffic.json"
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f"no PMC passes committed for {PROFILE_ROUND} yet"
    table = json.load(open(path)).get(workload)
    if table is None:
        return None, f"{rel} holds no PMC passes of workload {workload}"          # (collected for c3 and r0 only)
    stem = kernel[:-1] if kernel.endswith(">") else kernel
    hits = [k for k in table if k == kernel or (kernel.endswith(">") and k.startswith(stem) and k[len(stem):len(stem) + 1] in (",", ">"))]
    if len(hits) != 1:
        raise SystemExit(f"{rel}: kernel {kernel!r} of workload {workload!r} matches {hits or 'nothing'} (keys: {sorted(table)})")
    d = table[hits[0]]
    return d["read_bytes"] + d["write_bytes"], rel


PREHEAT_MS = 30.0


def step_flops(cfg, B):
    """Algorithmic FLOPs of one training step, SURVEY.md 8(d): 3 x the forward GEMM FLOPs (forward, dX, dW)."""
    if isinstance(cfg, SheetConfig):
        L, E, F, H = cfg.max_length, cfg.embed_dim, cfg.fc_dim, cfg.heads
        fwd = 2.0 * L * (E * 3 * E + E * E + E * F) + 2.0 * 2 * L * L * E + 2.0 * (L * F) * cfg.pixels      # 248.27 MFLOP at R0
        return 3.0 * fwd * B
    if isinstance(cfg, PixelConfig):
        return cfg.train_flops_per_sample() * B
    return 3.0 * B * sum(2.0 * n * k for n, k in cfg.layer_dims())


def widest_linear(cfg):
    """(N, K) of the widest Linear: fc_output for the sheet model, the hidden x hidden layers of the glyph nets."""
    if isinstance(cfg, SheetConfig):
        return cfg.pixels, cfg.max_length * cfg.fc_dim
    if isinstance(cfg, PixelConfig):
        return cfg.ff_dim, cfg.d_model
    return max(cfg.layer_dims(), key=lambda nk: nk[0] * nk[1])


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` run directly: start N rank processes (one per GPU, RCCL) BEFORE this process touches a
    GPU, let rank 0's JSON line through on stdout, and exit with the launcher's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def measure(name, dtype, B, K, W, rank, world, dist, with_roofline=True, want_sample=False):
    """W warm-up steps, then exactly K timed steps between fences.  Returns (engine, elapsed_s, dominant, table, loss)."""
    from ai_font_renderer_amd.engine import Engine
    from ai_font_renderer_amd.parallel import DataParallelStepper
    cfg = WORKLOADS[name]["cfg"]
    # (the pixel transformer above its workload's micro-batch: gradient accumulation, micro-batch rows at a time)
    micro = WORKLOADS[name]["batch"] if (isinstance(cfg, PixelConfig) and B > WORKLOADS[name]["batch"]) else None
    eng = Engine(cfg, dtype=dtype, max_batch=B, rank=rank, flags=int(os.environ.get("AFR_ENGINE_FLAGS", "0")), micro_batch=micro)   # flags: kernel A/B runs
    eng.load_params(synth.make_params(cfg))                           # same formula-generated weights on every rank
    x, font, tgt = make_inputs(name, cfg, B, rank)
    x, tgt = x.cuda(), tgt.cuda()
    font = font.cuda() if font is not None else None
    force_dp = os.environ.get("AFR_BENCH_FORCE_DP") == "1"
    stepper = DataParallelStepper(eng, dist, 2 if (force_dp and world == 1) else world)
    mean_elems = world * B * cfg.pixels

    def run(n):
        for _ in range(n):
            stepper.step(x, tgt, font, mean_elems)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # GPU pre-heat, NOT training steps: inference forwards of the same model for ~PREHEAT_MS of host time.  An MI355X takes
    # ~20 ms of sustained work to reach its operating point (measured on C3 with lr = 0, i.e. identical work every step:
    # 215 us/step over the first 20 steps, 202 us from step ~100 on; with this pre-heat 202 from the first step).  A training
    # run is thousands of steps, so the operating point is what the metric describes; --preheat-ms 0 measures from cold.
    if PREHEAT_MS > 0:
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < PREHEAT_MS:
            eng.forward(x, font, want_output=False)
        torch.cuda.synchronize()
    table, dom = [], None
    if with_roofline:
        # W warm-up steps.  The first of them (up to 3) run with every launch bracketed by events, to find the dominant kernel
        # and the per-shape table; reading that table back leaves the GPU idle for milliseconds, so it comes BEFORE the rest
        # of the warm-up: the timed region then starts on a GPU that has just been running the step, not on an idle one.
        n0 = 1 if W >= 2 else 0                                       # the very first step pays one-time costs (code-object load)
        run(n0)
        n1 = min(max(W - n0, 1), 3)
        eng.profile(1)
        run(n1)
        torch.cuda.synchronize()
        table = eng.profile_table()
        eng.profile(3)                                                # from here on: events around every 4th launch of the dominant kernel
        run(max(W - n0 - n1, 0))
    else:
        run(max(W, 1))
    fence()
    t0 = time.perf_counter()
    run(K)
    fence()
    elapsed = time.perf_counter() - t0
    if with_roofline:
        dom = eng.profile_read()
        eng.profile(0)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = eng.read_loss() / max(K + W, 1)
    if eng.error_flags():
        raise SystemExit("device error flag set (embedding index out of range / cooperative split-K timeout)")
    # a sample of the bitmaps the trained engine draws, with its f32 master weights, for the max-abs-diff leg (rank 0, CPU side)
    if rank == 0 and want_sample:
        ns = min(B, 16 if isinstance(cfg, SheetConfig) else 2 if isinstance(cfg, PixelConfig) else 190)
        if isinstance(cfg, PixelConfig):
            # after the run's steps at the default learning rate the transformer draws saturated (all-zero) bitmaps, on which every
            # precision agrees: its bitmaps are compared on the initial weights instead
            eng.load_params(synth.make_params(cfg))
            x, font = torch.tensor([40, 77]).cuda(), torch.tensor([0, 1]).cuda()     # ('(', font 0), ('M', font 1): half their pixels land inside (0, 1)
        xs, fs = x[:ns], (font[:ns] if font is not None else None)
        eng.sample = {"x": xs.cpu(), "font": fs.cpu() if fs is not None else None, "y": eng.forward(xs, fs).cpu(),
                      "params": {k: v.cpu() for k, v in eng.state_dict().items()}}
    return eng, elapsed, dom, table, loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (changes the workload: for experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity-mode (f32) and R0 side records")
    ap.add_argument("--table", action="store_true", help="also print the per-kernel time table to stderr")
    ap.add_argument("--preheat-ms", type=float, default=30.0, help="inference forwards issued for this long before the warm-up steps (0: none)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))                # nothing above this line touches a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    name = args.workload
    cfg = WORKLOADS[name]["cfg"]
    B = args.batch or WORKLOADS[name]["batch"]
    dtype = args.dtype or DEFAULT_DTYPE[name]
    K = args.steps if args.steps is not None else DEFAULT_STEPS[name][0]
    W = args.warmup if args.warmup is not None else DEFAULT_STEPS[name][1]

    global PREHEAT_MS
    PREHEAT_MS = args.preheat_ms
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = os.environ.get("AFR_BENCH_FORCE_DP") == "1"      # rehearse the multi-rank code path with a world of one
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    # from cold first (no pre-heat forwards; the process has done next to no GPU work so far), then the steady-state run
    cold_ms = None
    if args.preheat_ms > 0 and not args.no_extras:
        PREHEAT_MS = 0.0
        e0, el0, _, _, _ = measure(name, dtype, B, K, W, rank, world, dist, with_roofline=False)
        cold_ms = el0 / K * 1e3
        del e0
        torch.cuda.empty_cache()
        PREHEAT_MS = args.preheat_ms
    want_diff = world == 1 and not args.no_cpu_baseline
    eng, elapsed, dom, table, loss = measure(name, dtype, B, K, W, rank, world, dist, want_sample=want_diff)
    samples = {dtype: getattr(eng, "sample", None)}

    if rank == 0:
        value = world * B * K / elapsed
        ms_step = elapsed / K * 1e3
        # The dominant kernel against the roof SURVEY.md 8(d) names for it: the MFMA peak of its operand type for the dense
        # products (C3, R0 in f32), HBM for the streaming kernels and for R0's bf16 weight-gradient GEMM with the fused
        # optimizer (26 B of p/m/v traffic per output element; 8(d): "bf16 mode: HBM until B >~ 2700/GPU").
        secs = dom["avg_ms"] * 1e-3
        kern = dom["kernel"]
        f32_kernel = kern.startswith(("sheet_", "gemm_f32")) or "<f32>" in kern
        peak_fl = PEAK["f32"] if f32_kernel else PEAK[dtype]           # f32 VALU peak == f32 MFMA peak (157.3 TF)
        fl = dom["algo_flops"] / secs / 1e12
        by = dom["algo_bytes"] / secs / 1e9
        # which roof binds the dominant kernel: the one it is closer to (R0's bf16 weight-gradient GEMM with the fused optimizer
        # moves 26 B of p/m/v per output element and sits at 0.55 of HBM against 0.14 of MFMA; C3's products the other way)
        hbm_bound = dom["algo_flops"] <= 0 or by / HBM_PEAK_GBS > fl / peak_fl
        if hbm_bound:
            roof = {"bound": "hbm", "kernel": kern, "achieved": by, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / HBM_PEAK_GBS}
        else:
            # every dense product of the path runs on the matrix cores (the f32 forms at the f32 VALU's rate, 157.3 TF)
            roof = {"bound": "mfma", "kernel": kern, "achieved": fl, "peak": peak_fl, "unit": "TFLOP/s", "frac": fl / peak_fl}
        tr, src = pmc_traffic(name, kern) if (dtype == DEFAULT_DTYPE[name] and not args.batch) else (None, None)
        roof.update({"traffic": tr, "traffic_unit": f"bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, {src})" if tr is not None else src,
                     "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"], "algo_flops_per_launch": dom["algo_flops"],
                     "algo_bytes_per_launch": dom["algo_bytes"], "mfma_frac": fl / peak_fl if dom["algo_flops"] > 0 else None,
                     "hbm_frac": by / HBM_PEAK_GBS})
        # whole step against the MFMA peak, and the three products of the widest Linear (per-shape rows of the warm-up table)
        sf = step_flops(cfg, B)
        roof["step"] = {"algo_flops": sf, "achieved_tflops": sf / (ms_step * 1e-3) / 1e12, "peak": PEAK[dtype],
                        "frac": sf / (ms_step * 1e-3) / 1e12 / PEAK[dtype]}
        wn, wk = widest_linear(cfg)
        Bm = B * cfg.tokens if isinstance(cfg, PixelConfig) else B          # rows of a Linear's products (every pixel token is a row)
        # (operand orientation, MxNxK) of the three products: forward x.W^T, input gradient dy.W, weight gradient dy^T.x
        shapes = {"fwd": ("<0,0", f"[{Bm}x{wn}x{wk}]"), "dX": ("<0,1", f"[{Bm}x{wk}x{wn}]"), "dW": ("<1,1", f"[{wn}x{wk}x{Bm}]")}
        wl = {}
        for r in table:
            for role, (ori, sh) in shapes.items():
                plain = r["kernel"].startswith("gemm") and ori in r["kernel"] and r["kernel"].endswith(sh)
                body = r["kernel"].startswith("gemm_bf16_group256" + sh + ori)          # a plain product on the 256x256 body: name[shape]<a,b>
                if (plain or body) and role not in wl and r["avg_ms"] > 0:
                    tf = r["algo_flops"] / (r["avg_ms"] * 1e-3) / 1e12
                    wl[role] = {"kernel": r["kernel"], "avg_us": r["avg_ms"] * 1e3, "tflops": tf, "frac": tf / (PEAK["f32"] if "f32" in r["kernel"] else PEAK[dtype])}
        # a layer's weight and input gradients leave as ONE grouped launch when both qualify (afr_gemm_pair_plan)
        for r in table:
            if r["kernel"].startswith("gemm_bf16_group") and shapes["dX"][1][1:-1] in r["kernel"] and shapes["dW"][1][1:-1] in r["kernel"] \
                    and "dW+dX" not in wl and r["avg_ms"] > 0:
                tf = r["algo_flops"] / (r["avg_ms"] * 1e-3) / 1e12
                wl["dW+dX"] = {"kernel": r["kernel"], "avg_us": r["avg_ms"] * 1e3, "tflops": tf, "frac": tf / PEAK[dtype],
                               "note": "weight gradient (split-K slabs) and input gradient of the layer in one grouped launch"}
        roof["widest_linear"] = {"layer": f"{wk}->{wn}", "source": "hipEvent brackets, warm-up steps (every launch timed)", **wl}
        out = {
            "metric": "glyphs/sec training (batch fwd+bwd+step)", "value": value, "unit": "glyphs/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic", "preheat_ms": PREHEAT_MS, "cold_ms_per_step": cold_ms,
            "config": {"workload": DESCR[name], "per_gpu_batch": B, "global_batch": world * B,
                       "parallelism": f"dp{world}" if world > 1 else "single", "params": int(sum(n for _, _, _, n in eng.layout)),
                       "mean_loss": loss},
            "roofline": roof,
        }
        if args.table:
            for r in table:
                print(f"{r['kernel']:44s} n={r['launches']:4d} avg={r['avg_ms'] * 1e3:9.1f} us total={r['total_ms']:9.3f} ms", file=sys.stderr)
    del eng
    torch.cuda.empty_cache()
    if world == 1 and rank == 0:
        if not args.no_extras and not args.batch:
            # parity mode of the SAME config (exact-f32 MFMA everywhere: the mode that meets the 1e-4 bitmap bar)
            if dtype != "f32":
                ep, el, _, _, _ = measure(name, "f32", B, max(5, K // 4), 3, 0, 1, None, with_roofline=False, want_sample=want_diff)
                samples["f32"] = getattr(ep, "sample", None)
                del ep
                kk = max(5, K // 4)
                out["parity_mode"] = {"dtype": "f32", "ms_per_step": el / kk * 1e3, "value": B * kk / el, "unit": "glyphs/s", "steps": kk,
                                      "note": "same workload with f32 operands (v_mfma_f32_32x32x2_f32); bitmaps within 2e-5 of the reference"}
                torch.cuda.empty_cache()
            if name == "c3":
                # the reference's own model (R0) beside the headline config: sheets/s, bf16 throughput mode
                rb = WORKLOADS["r0"]["batch"]
                e2, el, d2, _, _ = measure("r0", "bf16", rb, 10, 3, 0, 1, None)
                out["r0"] = {"workload": DESCR["r0"], "dtype": "bf16", "ms_per_step": el / 10 * 1e3, "value": rb * 10 / el, "unit": "sheets/s",
                             "steps": 10, "dominant_kernel": d2["kernel"], "dominant_avg_ms": d2["avg_ms"],
                             "dominant_hbm_GBs": d2["algo_bytes"] / (d2["avg_ms"] * 1e-3) / 1e9,
                             "step_mfma_frac": step_flops(WORKLOADS["r0"]["cfg"], rb) / (el / 10) / 1e12 / PEAK["bf16"]}
                del e2
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, cfg, B)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            # max-abs bitmap diff vs the CPU reference path, for the benchmarked dtype and for the parity mode
            note = ("engine eval forward after the run's training steps vs the f32 oracle on the engine's own f32 master weights; "
                    "%d samples; tests assert <= 2.5e-2 (bf16) / 2e-5 (f32): tests/test_gpu_models.py")
            if isinstance(cfg, PixelConfig):
                note = ("engine eval forward on the INITIAL weights (the trained net saturates) vs the f32 oracle; %d samples; "
                        "tests assert <= 2.5e-2 (bf16) / 2e-5 (f32) at C5-mini size: tests/test_gpu_pixel.py")
            if samples.get(dtype):
                out["max_abs_bitmap_diff"] = {"value": reference_bitmap_diff(cfg, samples[dtype]), "dtype": dtype,
                                              "note": note % samples[dtype]["y"].shape[0]}
            if samples.get("f32") and "parity_mode" in out:
                out["parity_mode"]["max_abs_bitmap_diff"] = reference_bitmap_diff(cfg, samples["f32"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
