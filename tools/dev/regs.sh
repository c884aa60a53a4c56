#!/bin/bash
# usage: tools/dev/regs.sh [extra hipcc flags]   -> resource usage of the one kernel of tools/dev/one_kernel.hip
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -Rpass-analysis=kernel-resource-usage "$@" \
  tools/dev/one_kernel.hip -o /tmp/one.o 2>&1 | grep -E "error|Function Name|SGPRs:|VGPRs:|AGPRs|Scratch|Occupancy|Spill|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//'
