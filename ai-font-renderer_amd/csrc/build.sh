#!/bin/bash
# Build libafr.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
mkdir -p build
pids=()
for f in gemm elementwise sheet glyph_fused pixel afr_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ afr_common.h -nt build/$f.o ] || [ ../../include/afr.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# The GEMM kernels sit at the register limit (256 VGPRs around the K loops): an edit that makes one of them spill puts scratch
# traffic and vmcnt(0) waits INSIDE its K loop.  Refuse such a build (the remarks cost one extra device-only compile).
if [ ! -f build/gemm.remarks ] || [ gemm.hip -nt build/gemm.remarks ] || [ afr_common.h -nt build/gemm.remarks ]; then
  hipcc $FLAGS --cuda-device-only -c gemm.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> build/gemm.remarks.tmp || true
  mv build/gemm.remarks.tmp build/gemm.remarks
fi
if grep -E "ScratchSize \[bytes/lane\]: [1-9]" build/gemm.remarks > /dev/null; then
  echo "build.sh: a kernel of gemm.hip uses scratch (register spills):" >&2
  grep -B4 -E "ScratchSize \[bytes/lane\]: [1-9]" build/gemm.remarks | grep -E "Function Name|ScratchSize" >&2
  exit 1
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o libafr.so build/gemm.o build/elementwise.o build/sheet.o build/glyph_fused.o build/pixel.o build/afr_api.o
echo "built $(pwd)/libafr.so"
