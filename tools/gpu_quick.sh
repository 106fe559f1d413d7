#!/bin/bash
# one pytest selection on the GPU box: usage bash tools/gpu_quick.sh <tag> <pytest args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
timeout -k 10 900 python -m pytest "$@" -m gpu -q -x --durations=15 > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -40 gpurun_out/${TAG}_tests.log
exit $rc
