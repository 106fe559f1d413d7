#!/bin/bash
# phase stamps of the training forward's conv-stack kernel under the diag flavour's timing variants (DX_CS_VARIANT)
# usage: bash tools/gpu_cs_variants.sh <tag> "<variants>" "<waves>"
TAG=${1:-csv}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
for v in ${2:-0 1 2 3 4 7}; do
  for w in ${3:-0 4}; do
    echo "== variant $v wave $w" >> gpurun_out/${TAG}.log
    DX_CS_VARIANT=$v DX_CS_DIAG=$w DX_CS_STEP=5 timeout -k 10 120 python3 tools/cs_stamps.py 8192 0 2>&1 | tail -10 >> gpurun_out/${TAG}.log || exit 1
  done
done
cat gpurun_out/${TAG}.log
