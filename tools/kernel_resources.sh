#!/bin/bash
# Prints registers / scratch / occupancy of every kernel of rt_kernels.hip as the compiler reports them.
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -Rpass-analysis=kernel-resource-usage -c rt_kernels.hip -o /tmp/rt_kernels_res.o 2>&1 |
  grep -E "Function Name|[^l]VGPRs:|ScratchSize|Occupancy|LDS Size" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - -
