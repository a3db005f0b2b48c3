#!/bin/bash
# Build libx2vlm_hip.so (gfx950 only) in-tree: x2-vlm_amd/libx2vlm_hip.so.  Incremental (sources newer than their objects); --clean: from scratch.
set -e
cd "$(dirname "$0")"
OUT=../libx2vlm_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
if [ "$1" = "--clean" ]; then rm -rf ../_build $OUT; fi     # every object from source (a driver proving a from-source build: X2_CLEAN_BUILD=1)
mkdir -p ../_build
pids=()
for f in runtime gemm attention rowwise heads masking optim comm; do
  if [ ! -f ../_build/$f.o ] || [ $f.hip -nt ../_build/$f.o ] || [ x2_common.h -nt ../_build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o ../_build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC ../_build/*.o -ldl -o $OUT
echo "built $(realpath $OUT)"
