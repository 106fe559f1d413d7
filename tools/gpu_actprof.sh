#!/bin/bash
# kernel durations of the rollout's act step: usage bash tools/gpu_actprof.sh <tag> [batch ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/tools/act_bench.py "$@" > $R/gpurun_out/${TAG}.log 2>&1 || exit 1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${TAG}_prof/k_kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
  print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f}")
PY
cp $R/gpurun_out/${TAG}_prof/k_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_prof
