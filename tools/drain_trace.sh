#!/bin/bash
# Kernel trace of short frames (125 pattern columns of cu_cover2): tools/drain_trace.sh <tag>; afterwards tools/drain_report.py
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/frame_cost.py --columns 125 --repeat 1 > $OUT/out.txt 2> $OUT/err.txt || exit 1
