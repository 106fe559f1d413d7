#!/bin/bash
# Runs on the GPU box: the backward stages' traversal orders (DX_BWD_ORDER, igemm.hpp: bwd_descending) -- stage times at
# minibatch 8192 for a few masks, then parity of the default.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
for m in ${MASKS:-0 21 16 20 5 29 31}; do
  echo "DX_BWD_ORDER=$m"
  DX_BWD_ORDER=$m timeout -k 10 200 python3 tools/stage_bench.py ${BATCH:-8192} 10 2>&1 | grep '"stage"' || exit 1
done
