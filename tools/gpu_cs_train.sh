#!/bin/bash
# A/B of the training forward's conv-stack kernels on one box: parity tests of the forward, phase stamps, trunk timing.
# usage: bash tools/gpu_cs_train.sh <tag> [pytest -k expression] [other library.so]
TAG=${1:-cst}
SEL=${2:-"training_forward or test_forward or conv_stack or loss_and_gradients"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
timeout -k 10 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x -k "$SEL" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -15 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then echo "tests failed ($rc): no further GPU step"; exit $rc; fi
for w in 0 4; do
  DX_CS_DIAG=$w DX_CS_STEP=5 timeout -k 10 120 python3 tools/cs_stamps.py 8192 0 2> gpurun_out/${TAG}_stamps_w$w.log || exit 1
  tail -12 gpurun_out/${TAG}_stamps_w$w.log
done
# (A/B against another build of the library kept beside the default one: DERL_AMD_LIBRARY=<name>.so, see tools/gpu_stage_ab.sh)
for lib in default ${3:-default}; do
  if [ "$lib" = default ]; then unset DERL_AMD_LIBRARY; else export DERL_AMD_LIBRARY=$lib; fi
  timeout -k 10 120 python3 tools/trunk_bench.py 8192 1024 >> gpurun_out/${TAG}_trunk_$lib.log 2>&1 || exit 1
  echo "library $lib"; tail -2 gpurun_out/${TAG}_trunk_$lib.log
done
