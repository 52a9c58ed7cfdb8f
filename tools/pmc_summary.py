#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as the microarch guide prescribes)
into per-kernel HBM traffic per launch.  FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-B/lane streams).

  python tools/pmc_summary.py <dir prefix> <workload> [...] [--json out.json]
expects <prefix>_<workload>_FETCH_SIZE/ and <prefix>_<workload>_WRITE_SIZE/ (rocprofv3 -d outputs, csv)."""
import collections
import csv
import glob
import json
import re
import sys


def rows(d):
    """(kernel name, counter name, value) per dispatch of a rocprofv3 -d output: the csv of --output-format csv, or the
    rocpd database ROCm 7.2 writes by default (its counters_collection view)."""
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
        return
    import sqlite3
    db = glob.glob(d + "/**/*_results.db", recursive=True)[0]
    yield from sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id")


def load(d):
    acc = collections.defaultdict(list)
    for k, _, v in rows(d):
        acc[k].append(v)
    return acc


def kernel_stats(d):
    """[(kernel, calls, total us, average us, percent)] from a --kernel-trace --stats run's rocpd database."""
    import sqlite3
    db = glob.glob(d + "/**/*_results.db", recursive=True)[0]
    return list(sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels"))


def short(name):
    m = re.search(r"(gemm_bf16_group256|gemm_bf16_group|gemm_(?:bf16|f32)<[^>]*>)", name)
    if m:
        return m.group(1).replace(" ", "")
    for k in ("sheet_bwd", "sheet_fwd", "adamw", "reduce_group", "reduce_slabs", "mse_grad", "glyph1_step", "glyph_l1_fwd", "glyph_l1_bwd_fused", "glyph_l1_bwd",
              "pixel_ln_bwd", "pixel_add_ln", "pixel_attn_bwd", "pixel_attn", "pixel_head_bwd", "pixel_head", "pixel_ctx_bwd", "pixel_ctx", "pixel_accum", "pixel_cast",
              "glyph_table", "glyph_combo", "gemm_fp8", "f32_to_fp8", "glyph_embed_bwd", "glyph_embed", "transpose_bf16", "f32_to_bf16", "clamp_out", "clamp_bwd"):
        if k in name:
            return k
    return name[:40]


def main():
    if sys.argv[1] == "--stats":                      # python tools/pmc_summary.py --stats <rocprofv3 -d dir> > kernel_stats.csv
        w = csv.writer(sys.stdout)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in kernel_stats(sys.argv[2]):
            w.writerow(r)
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    prefix, workloads = args[0], [a for a in args[1:] if a != out_json]
    result = {}
    for w in workloads:
        fe, wr = load(f"{prefix}_{w}_FETCH_SIZE"), load(f"{prefix}_{w}_WRITE_SIZE")
        print(f"== {w}: HBM traffic per launch (KB counters -> MB; FETCH x2 corrected)")
        result[w] = {}
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            f = 2.0 * sum(fe[k]) / len(fe[k]) * 1024.0
            ww = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024.0
            print(f"{short(k):22s} n={len(fe[k]):4d} read={f / 1e6:10.2f} MB  write={ww / 1e6:10.2f} MB  total={(f + ww) / 1e6:10.2f} MB")
            result[w][short(k)] = {"read_bytes": f, "write_bytes": ww, "launches": len(fe[k])}
    if out_json:
        json.dump(result, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
