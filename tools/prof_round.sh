#!/bin/bash
# The profiling run whose summaries go to profiles/ (on the GPU box): tools/prof_round.sh <tag> [scene] [spp]
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py --no-other` (the default bench command without the frames of
#      the other configurations, which launch the same kernels on other scenes and would blur the per-kernel averages)
#   2. one rocprofv3 --pmc run per counter group over one frame (tools/prof_run.py)
# Afterwards, here: python tools/profile_report.py gpurun_out/<tag> <tag> <scene>
set -u
TAG=$1; SCENE=${2:-cu_cover2}; SPP=${3:-1000}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o t -- python3 bench.py --no-other > $OUT/bench_under_rocprof.json 2> $OUT/bench.err || exit 1
[ "${ONLY_TRACE:-0}" = "1" ] && exit 0
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH" \
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/prof_run.py $SCENE $SPP > $OUT/pmc_$i.log 2>&1 || exit 1
  echo "pass $i done"
done
