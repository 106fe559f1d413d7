#!/bin/bash
# kernel stats of a short bench run: usage bash tools/gpu_kstats.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs "$@" > $R/gpurun_out/${TAG}.json 2> /dev/null || exit 1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${TAG}_prof/k_kernel_stats.csv")))
for r in rows:
  n=r['Name']
  if any(k in n for k in ("heads_loss","finalize","adam","pack_fused","transpose","split_planes","colsum","sumsq","adv_norm","permute_reduce","categorical_loss","loss_reduce","igemm_tn_kernel<5","igemm_nt_kernel<6","igemm_nt_lat_kernel<4")):
    print(f"{n[:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -f $R/gpurun_out/${TAG}_prof/*kernel_trace.csv
