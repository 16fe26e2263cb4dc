#!/bin/bash
# A/B on the GPU box: bash tools/ab.sh "<quick_bench cases>" lib1.so lib2.so ...   (libs relative to vector-quantization-by-ml_amd/lib)
mkdir -p gpurun_out/r3
cases=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  echo "=== $lib (rep $rep)"
  VQ_MI355X_LIB=$PWD/vector-quantization-by-ml_amd/lib/$lib python tools/quick_bench.py $cases 2>&1 | grep -v amdgpu.ids
done
done
