#!/bin/bash
# rocprofv3 --pmc passes over tools/act_bench.py (the rollout's act step): usage bash tools/gpu_actpmc.sh <tag> <batch> "<group 1>" ...
TAG=$1; BATCH=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/${TAG}_$i -o pmc -- python3 $R/tools/act_bench.py $BATCH > $R/gpurun_out/${TAG}_$i.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/${TAG}_$i/**/pmc_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
  name = r["Kernel_Name"]
  if "convstack_roll_kernel" in name or "tail_act" in name:
    acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
  print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "dispatches", len(next(iter(v.values()))))
PY
  i=$((i+1))
done
