#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel, as the compiler reports them (no GPU needed).
#   tools/kernel_resources.sh [name-filter]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c -x hip "$ROOT/libhuffman_amd/csrc/hufgpu_api.hip" \
    -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk -v f="${1:-.}" '/Function Name:/ {name=$3; show = (name ~ f)} show && /Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:/'
