#!/bin/bash
# Build libafr.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
mkdir -p build
pids=()
for f in gemm elementwise sheet glyph_fused afr_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ afr_common.h -nt build/$f.o ] || [ ../../include/afr.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libafr.so build/gemm.o build/elementwise.o build/sheet.o build/glyph_fused.o build/afr_api.o
echo "built $(pwd)/libafr.so"
