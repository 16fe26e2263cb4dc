#!/bin/bash
# Build the MI355X (gfx950) shared library in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p lib
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -Wno-unused-value \
    ${VQ_EXTRA_FLAGS:-} csrc/vq_kernels.hip -o lib/libvq_mi355x.so
