#!/bin/bash
# Build libx2vlm_hip.so (gfx950 only) in-tree: x2-vlm_amd/libx2vlm_hip.so
set -e
cd "$(dirname "$0")"
OUT=../libx2vlm_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
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
