#!/bin/bash
# sweeps the tuning knobs of the extend kernel one at a time (cu_cover2, 1000 spp); output: gpurun_out/sweep.log
set -u
L=gpurun_out/sweep.log
: > $L
bash tools/sweep.sh cu_cover2 1000 RT_AMD_LEAF_THRESHOLD 24 32 40 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 1000 RT_AMD_BLOCKED_THRESHOLD 8 12 16 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 1000 RT_AMD_FETCH_THRESHOLD 20 28 36 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 1000 RT_AMD_STEP_DROPOUT 8 12 16 >> $L 2>&1 || exit 1
