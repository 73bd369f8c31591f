#!/bin/bash
# Build-time guard for kernels_pipe.hip: instances that take their halo rows by inline-asm LDS-DMA (no ReLU in
# front: template argument RELU = false, mangled "ILb0E") count on the compiler's vmcnt bookkeeping seeing every
# other vector-memory operation of the loop; a scratch (spill) access would break that silently.
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${ARCH:-gfx950}
cd "$(dirname "$0")"
mkdir -p build
$HIPCC --offload-arch=$ARCH -O3 -std=c++17 -S --cuda-device-only kernels_pipe.hip -o build/kernels_pipe.s 2>/dev/null
bad=$(awk '/\.name:/ {name=$2} /\.vgpr_spill_count:/ {if (name ~ /sepconv_pipe_kernelILb0E/ && $2 != 0) print name, $2}' build/kernels_pipe.s)
if [ -n "$bad" ]; then
  echo "check_spills: LDS-DMA instances of sepconv_pipe_kernel spill registers:" >&2
  echo "$bad" >&2
  exit 1
fi
echo "check_spills: ok (no spills in the LDS-DMA instances)"
