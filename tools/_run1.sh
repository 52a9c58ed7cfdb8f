set -x
export AFR_LIB_PATH=$PWD/build_exp/timing/libafr.so
for cold in 0 1; do
COLD=$cold python tools/gemm_timeline.py 8192 1024 1024 - 1 3
COLD=$cold python tools/gemm_timeline.py 8192 1024 1024 b 1 3
COLD=$cold python tools/gemm_timeline.py 1024 1024 8192 ab 8 3
done
python tools/gemm_timeline.py 8192 8192 8192 - 1 2
unset AFR_LIB_PATH
python bench.py --table --steps 100 --warmup 10
