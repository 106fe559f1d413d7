#!/bin/bash
# Alternating runs of bench.py under several environment settings on ONE box.
# usage: bash tools/gpu_multi.sh <reps> <bench args or ""> "<env 1>" "<env 2>" ...
REPS=$1; ARGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in $(seq 1 $REPS); do
  for E in "$@"; do
    env $E timeout -k 10 150 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs $ARGS 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep $rep [$E] ms', d['ms_per_step'])" || exit 1
  done
done
