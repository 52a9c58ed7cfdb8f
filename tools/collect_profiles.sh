#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <round, e.g. r03>
# 1. bench logs (c1, c2, c3, r0, c5)                        -> gpurun_out/<round>/bench_*.log
# 2. rocprofv3 --kernel-trace --stats of bench.py        -> gpurun_out/<round>/stats_{c3,r0}/
# 3. PMC passes, each in its own run with no trace flags -> gpurun_out/<round>/pmc_{c3,r0}_{FETCH_SIZE,WRITE_SIZE,MFMA}/
# The program follows `--` directly (python3 bench.py ...): no env / shell wrapper between rocprofv3 and it.
set -e
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
B="--steps 100 --warmup 10 --no-extras --no-cpu-baseline --preheat-ms 0"     # no pre-heat forwards in the kernel statistics
for w in c1 c2 c3 r0; do
  python3 bench.py --workload $w --steps 200 --warmup 20 --table --no-extras > $O/bench_$w.log 2>&1
  echo "bench $w done"
done
python3 bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1
echo "bench default done"
python3 bench.py --workload c5 --table > $O/bench_c5.log 2>&1          # one micro-batch of BASELINE configs[4]
echo "bench c5 done"
python3 tools/c5_step_bench.py 32 bf16 > $O/c5_step.log 2>&1
echo "c5 step table done"
for w in c1 c2; do       # the latency-bound small nets: kernel statistics only
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o $w -- python3 bench.py --workload $w $B > $O/stats_$w.log 2>&1
  echo "stats $w done"
done
for w in c3 r0; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o $w -- python3 bench.py --workload $w $B > $O/stats_$w.log 2>&1
  echo "stats $w done"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_${w}_$c -o $w -- python3 bench.py --workload $w $B > $O/pmc_${w}_$c.log 2>&1
    echo "pmc $w $c done"
  done
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $O/pmc_${w}_MFMA -o $w -- python3 bench.py --workload $w $B > $O/pmc_${w}_MFMA.log 2>&1
  echo "pmc $w MFMA done"
done
# the pixel transformer's micro-batch step: HBM traffic only (its launches are 10-100x longer: 10 steps)
B5="--workload c5 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --preheat-ms 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_c5_$c -o c5 -- python3 bench.py $B5 > $O/pmc_c5_$c.log 2>&1
  echo "pmc c5 $c done"
done
python3 tools/pmc_summary.py $O/pmc c3 r0 c5 --json $O/pmc_traffic.json > $O/pmc_hbm_traffic.txt
python3 tools/pmc_mfma_summary.py $O/pmc c3 r0 > $O/pmc_mfma_busy.txt
for w in c1 c2 c3 r0; do python3 tools/pmc_summary.py --stats $O/stats_$w > $O/${w}_kernel_stats.csv; done
echo "profiles collected under $O"
