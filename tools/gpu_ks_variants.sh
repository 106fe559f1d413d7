#!/bin/bash
# Runs on the GPU box (diag flavour): conv0_wgrad_ks.hip's timing variants -- which of {loads, copy, multiply} sets its time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
export DERL_AMD_LIBRARY=diag
for v in ${VARIANTS:-0 1 2 4 3 5 6 7}; do
  echo "variant $v (1 = no multiply, 2 = no loads, 4 = no copy)"
  DX_KS_VARIANT=$v timeout -k 10 200 python3 tools/stage_bench.py 8192 10 7 2>&1 | grep '"stage"' || exit 1
done
