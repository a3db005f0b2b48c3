#!/bin/bash
# Probe build of the kernels: -DX2_PROBE compiles the measurement-only switches in (NT GEMM ablation bits: x2_tune(2, 4) = no
# epilogue, (2, 16) = sc1 stores; per-phase s_memtime stamps of gemm_nt256_kernel).  The shipped library
# (x2-vlm_amd/csrc/build.sh) has none of them: a benchmark cannot be told to skip work.  Use:
#   bash probes/build_probe.sh && X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so python probes/bench_nt256.py
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../x2-vlm_amd/csrc"
OUT="$HERE/_probe"
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DX2_PROBE"
pids=()
for f in runtime gemm attention rowwise heads masking optim comm; do
  if [ ! -f "$OUT/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ "$SRC/x2_common.h" -nt "$OUT/$f.o" ]; then
    hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/*.o -ldl -o "$OUT/libx2vlm_hip_probe.so"
echo "built $OUT/libx2vlm_hip_probe.so"
