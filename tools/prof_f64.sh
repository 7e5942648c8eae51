#!/bin/bash
# Profile of one Float64 frame (on the GPU box): tools/prof_f64.sh <tag> [scene] [columns]
#   kernel trace + stats, then one rocprofv3 --pmc run per counter group, each over tools/f64_one.py
set -u
TAG=$1; SCENE=${2:-cu_cover2}; COLS=${3:-64}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/f64_one.py $SCENE $COLS > $OUT/trace.log 2>&1 || exit 1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH" \
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/f64_one.py $SCENE 16 > $OUT/pmc_$i.log 2>&1 || exit 1
  echo "pass $i done"
done
