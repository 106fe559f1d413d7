#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_c3prof -o c3 -- python3 $R/tools/bench_configs.py c3 5 > $R/gpurun_out/r03_c3prof.json 2> $R/gpurun_out/r03_c3prof.err
echo rc $?
cat $R/gpurun_out/r03_c3prof.json
head -25 $R/gpurun_out/r03_c3prof/*/c3_kernel_stats.csv 2>/dev/null || find $R/gpurun_out/r03_c3prof -name "*kernel_stats.csv" | head
