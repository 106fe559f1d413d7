#!/bin/bash
# kernel trace of a short bench run + tools/gpu_iter_timeline.py.  usage: bash tools/gpu_timeline.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_trace -o k -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs "$@" > $R/gpurun_out/${TAG}.json 2> /dev/null || exit 1
python3 $R/tools/gpu_iter_timeline.py $R/gpurun_out/${TAG}_trace/k_kernel_trace.csv 20
rm -f $R/gpurun_out/${TAG}_trace/*kernel_trace.csv
