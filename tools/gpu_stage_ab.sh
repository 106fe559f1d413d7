#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: stage timings at a batch (tools/stage_bench.py).
# usage: bash tools/gpu_stage_ab.sh <tag> <batch> <libA.so|default> <libB.so|default> [reps] [stage ...]
TAG=$1; BATCH=$2; A=$3; B=$4; REPS=${5:-2}; shift $(( $# < 5 ? $# : 5 ))
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
for rep in $(seq 1 $REPS); do
  for which in "$A" "$B"; do
    if [ "$which" = default ]; then unset DERL_AMD_LIBRARY; else export DERL_AMD_LIBRARY=$which; fi
    echo "== rep $rep library $which" | tee -a gpurun_out/${TAG}.log
    timeout -k 10 200 python3 tools/stage_bench.py $BATCH 10 "$@" 2>&1 | tee -a gpurun_out/${TAG}.log | python3 -c "
import sys, json
for line in sys.stdin:
  try: d = json.loads(line)
  except Exception: continue
  print(f\"  {d['name']:16s} {d['us']:8.1f} us\")"
  done
done
