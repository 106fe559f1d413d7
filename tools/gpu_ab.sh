#!/bin/bash
# A/B on ONE box, alternating: bench.py with two environment settings.
# usage: bash tools/gpu_ab.sh "<env A>" "<env B>" [reps] [extra bench args]
A=$1; B=$2; REPS=${3:-3}; shift $(( $# < 3 ? $# : 3 ))
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in $(seq 1 $REPS); do
  for which in A B; do
    if [ $which = A ]; then E=$A; else E=$B; fi
    env $E timeout -k 10 150 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs "$@" 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which rep $rep [$E] ms', d['ms_per_step'])" || exit 1
  done
done
