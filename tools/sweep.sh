#!/bin/bash
# usage: tools/sweep.sh <scene> <spp> VAR v1 v2 ...   (one env knob of librt_amd.so, one frame per value)
SCENE=$1; SPP=$2; VAR=$3; shift 3
for v in "$@"; do
  printf "%s=%s  " $VAR $v
  env $VAR=$v timeout -k 10 120 python -u tools/prof_run.py $SCENE $SPP || exit 1
done
