#!/bin/bash
# usage: tools/knob_sweep.sh <lib> "<VAR=v VAR2=v2>" ["<VAR=v ...>" ...]: bench.py's main line (4 frames) with another build of
# the library and one set of environment knobs per argument
lib=$1; shift
for knobs in "$@"; do
  printf "%-28s %-50s " "$lib" "$knobs"
  env $knobs timeout -k 10 120 python tools/ab_bench.py "$lib" --steps 4 --warmup 1 --no-cpu --no-other 2>/dev/null | tail -1 |
    python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s" % d["value"])' || exit 1
done
