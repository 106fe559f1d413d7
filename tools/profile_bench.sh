#!/bin/bash
# Runs on the GPU box: default bench line, the same command under rocprofv3 --kernel-trace --stats,
# and config 3 / an 8-GPU-shard-sized run for DESIGN.md's table.  Outputs under gpurun_out/<tag>_*;
# copy the ones to keep into profiles/.   usage: bash tools/profile_bench.sh <tag>
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json
python3 $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_shard32.json
python3 $R/tools/bench_configs.py c3 5 > $R/gpurun_out/${TAG}_c3.json 2> /dev/null
python3 $R/tools/bench_configs.py c5 20 > $R/gpurun_out/${TAG}_c5.json 2> /dev/null
