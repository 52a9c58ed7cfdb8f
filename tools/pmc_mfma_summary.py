#!/usr/bin/env python3
"""Summarise the MFMA-busy PMC pass of bench.py (rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE, its own run, no trace flags) per kernel.

  python tools/pmc_mfma_summary.py <dir prefix> <workload> [...]      expects <prefix>_<workload>_MFMA/

Columns (averages per launch):
  mfma_busy   SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe was busy, summed over the chip's 1024 SIMDs
  gui_active  GRBM_GUI_ACTIVE: cycles the dispatch kept the GPU active, summed over the 8 XCDs
  util        mfma_busy / (1024 SIMDs x gui_active / 8): the fraction of the dispatch's SIMD-cycles with the matrix pipe busy
              (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived metrics).  GRBM_GUI_ACTIVE includes launch and
              drain, so on dispatches of a few tens of microseconds util under-reads the steady-state loop.
  mops_bf16   SQ_INSTS_VALU_MFMA_MOPS_BF16: bf16 matrix operations issued, in units of 512 FLOP."""
import collections
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary  # noqa: E402


def main():
    prefix, workloads = sys.argv[1], sys.argv[2:]
    for w in workloads:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for k, c, v in pmc_summary.rows(f"{prefix}_{w}_MFMA"):
            acc[pmc_summary.short(k)][c].append(v)
        print(f"== {w}: per launch averages")
        print(f"{'kernel':24s} {'n':>5s} {'mfma_busy':>12s} {'gui_active':>12s} {'sq_busy':>12s} {'util':>6s} {'mops_bf16':>12s} {'TFLOP(mops*512)':>16s}")
        out = []
        for k, c in acc.items():
            avg = {n: sum(v) / len(v) for n, v in c.items()}
            n = max(len(v) for v in c.values())
            busy, gui = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), avg.get("GRBM_GUI_ACTIVE", 0.0)
            util = busy / (1024.0 * gui / 8.0) if gui > 0 else float("nan")
            out.append((busy * n, k, n, busy, gui, avg.get("SQ_BUSY_CYCLES", 0.0), util, avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)))
        for _, k, n, busy, gui, sqb, util, mops in sorted(out, reverse=True):
            print(f"{k:24s} {n:5d} {busy:12.0f} {gui:12.0f} {sqb:12.0f} {util:6.3f} {mops:12.0f} {mops * 512 / 1e12:16.4f}")


if __name__ == "__main__":
    main()
