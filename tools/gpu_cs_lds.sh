#!/bin/bash
# Where do the conv-stack forward's LDS bank conflicts come from?  SQ LDS counters of convstack_train_kernel in the diag
# flavour's timing variants (WRONG results, same instruction mix minus one part): 0 = everything, 2 = no LDS re-reads in
# the conv1 / conv2 loops, 4 = no conv0 MFMAs (and their frame / weight-plane reads), 7 = both and no weight reloads.
# usage: bash tools/gpu_cs_lds.sh <tag>
TAG=${1:-cslds}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export DERL_AMD_LIBRARY=libderl_amd_diag.so
export DX_CS_DIAG=0
for V in 0 2 4 7; do
  export DX_CS_VARIANT=$V
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/${TAG}_$V -o pmc -- python3 $R/tools/stage_bench.py 8192 2 0 > $R/gpurun_out/${TAG}_$V.log 2>&1 || { tail -3 $R/gpurun_out/${TAG}_$V.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/${TAG}_$V/**/pmc_counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
  if "convstack_train" in r["Kernel_Name"]:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant $V", {c: round(sum(x) / len(x)) for c, x in acc.items()})
PY
done
