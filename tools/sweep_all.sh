#!/bin/bash
# sweeps the tuning knobs of the extend kernel one at a time (cu_cover2, 400 spp); output: gpurun_out/sweep.log
set -u
L=gpurun_out/sweep.log
: > $L
bash tools/sweep.sh cu_cover2 400 RT_AMD_LEAF_THRESHOLD 24 32 40 48 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 400 RT_AMD_BLOCKED_THRESHOLD 6 12 20 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 400 RT_AMD_FETCH_THRESHOLD 8 16 24 32 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 400 RT_AMD_STEP_DROPOUT 4 8 12 16 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 400 RT_AMD_SLOTS 2097152 4194304 6291456 8388608 >> $L 2>&1 || exit 1
bash tools/sweep.sh cu_cover2 400 RT_AMD_EXTEND_BLOCKS_PER_CU 6 12 >> $L 2>&1 || exit 1
