#!/bin/bash
# A/B of an environment switch on ONE box, alternating: bench lines at a given env count.
# usage: bash tools/gpu_env_ab.sh <tag> <nenvs> <reps> "<VAR=value ...>" ["<VAR=value ...>" ...]   ("-" = no switch)
TAG=$1; NENVS=$2; REPS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
for rep in $(seq 1 $REPS); do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then pre=""; else pre="$setting"; fi
    out=$(env $pre timeout -k 10 200 python3 bench.py --nenvs $NENVS --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1)
    echo "$out" >> gpurun_out/${TAG}.log
    echo "rep $rep [$setting] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"])')"
  done
done
