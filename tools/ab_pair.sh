#!/bin/bash
# A/B on one box: bench.py's main line with another build of the library (argument 1) and with the tree's, alternating.
other=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  timeout -k 10 120 python tools/ab_bench.py "$other" --steps 6 --warmup 2 --no-cpu --no-other 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("other", d["value"], d["ms_per_step"])' || exit 1
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-other 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("tree ", d["value"], d["ms_per_step"])' || exit 1
done
