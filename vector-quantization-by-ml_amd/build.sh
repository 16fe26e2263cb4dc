#!/bin/bash
# Build the MI355X (gfx950) shared library in-tree.  hipcc cross-compiles without a GPU.
# csrc/vq_kernels.hip is ONE source; it is compiled once per build part (-DVQ_PART=n, see "Build parts" in that file) in
# parallel and the objects are linked.  VQ_BUILD_SINGLE=1 compiles it as a single translation unit instead (same library,
# ~2 minutes on one core).
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p lib build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${VQ_LIB_OUT:-lib/libvq_mi355x.so}
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-function ${VQ_EXTRA_FLAGS:-}"
if [ "${VQ_BUILD_SINGLE:-0}" = "1" ]; then
    $HIPCC $FLAGS -shared csrc/vq_kernels.hip -o "$OUT"
    exit 0
fi
tag=$(echo "$OUT $FLAGS" | md5sum | cut -c1-8)
pids=()
for part in 0 1 2 3 4 5 6; do
    $HIPCC $FLAGS -DVQ_PART=$part -c csrc/vq_kernels.hip -o build/vq_part${part}_$tag.o &
    pids+=($!)
done
rc=0
for pid in "${pids[@]}"; do wait "$pid" || rc=1; done
[ $rc -eq 0 ] || { echo "build.sh: a part failed to compile" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC build/vq_part[0-6]_$tag.o -o "$OUT"
rm -f build/vq_part[0-6]_$tag.o
