#!/bin/bash
# all kernel stats of a short bench run (top 30 by time): usage bash tools/gpu_kstats_all.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs "$@" > $R/gpurun_out/${TAG}.json 2> /dev/null || exit 1
rm -f $R/gpurun_out/${TAG}_prof/*kernel_trace.csv
python3 - <<PY
import csv, json
print(json.load(open("$R/gpurun_out/${TAG}.json"))["ms_per_step"], "ms per iteration under rocprofv3")
rows=list(csv.DictReader(open("$R/gpurun_out/${TAG}_prof/k_kernel_stats.csv")))
for r in rows[:45]:
  print(f"{r['Name'][:88]:88s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
