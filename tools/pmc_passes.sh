#!/bin/bash
# rocprofv3 --pmc passes over tools/stage_bench.py (one counter group per run: --pmc never shares a
# run with a trace).  usage: bash tools/pmc_passes.sh <tag> <batch> "<group 1>" "<group 2>" ...
set -e
TAG=$1; BATCH=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
  rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/${TAG}_$i -o pmc -- python3 $R/tools/stage_bench.py $BATCH 2 > $R/gpurun_out/${TAG}_$i.log 2>&1
  i=$((i+1))
done
