#!/bin/bash
# Build-time guard for kernels_pipe.hip: instances that take their halo rows by inline-asm LDS-DMA (no ReLU in
# front: template argument RELU = false, mangled "ILb0E") count on the compiler's vmcnt bookkeeping seeing every
# other vector-memory operation of the loop; ANY scratch access -- a register spill, or a dynamically indexed
# array demoted to private memory -- would break that count silently.
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${ARCH:-gfx950}
cd "$(dirname "$0")"
mkdir -p build
$HIPCC --offload-arch=$ARCH -O3 -std=c++17 -S --cuda-device-only kernels_pipe.hip -o build/kernels_pipe.s 2>/dev/null
report=$(awk '/\.name:/ {name=$2}
              /\.private_segment_fixed_size:/ {priv[name]=$2}
              /\.vgpr_spill_count:/ {spill[name]=$2}
              END {for (n in spill) if (n ~ /sepconv_pipe_kernelI[^L]*Lb0E/) print n, spill[n], priv[n]}' build/kernels_pipe.s)
if [ -z "$report" ]; then
  echo "check_spills: no sepconv_pipe_kernel<RELU=false> instance found in the assembly (mangling changed?)" >&2
  exit 1
fi
bad=$(echo "$report" | awk '$2 != 0 || $3 != 0')
if [ -n "$bad" ]; then
  echo "check_spills: LDS-DMA instances of sepconv_pipe_kernel use scratch (name, vgpr spills, private bytes):" >&2
  echo "$bad" >&2
  exit 1
fi
echo "check_spills: ok ($(echo "$report" | wc -l) LDS-DMA instances, no spills, no private segment)"
