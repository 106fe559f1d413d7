#!/bin/bash
# Runs on the GPU box: config 5's shard (A2C, 512 envs x 5 steps) under rocprofv3 --kernel-trace --stats.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5prof -o c5 -- python3 $R/tools/bench_configs.py c5 20 > $R/gpurun_out/c5_under_rocprof.json 2> /dev/null || exit 1
cat $R/gpurun_out/c5_under_rocprof.json | head -c 600
rm -f $R/gpurun_out/c5prof/*kernel_trace.csv
