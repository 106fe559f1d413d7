#!/bin/bash
# the rollout's conv-stack kernel on one box: parity tests, phase stamps, rollout timing (optionally against another build
# of the library kept beside the default one: usage bash tools/gpu_roll.sh <tag> [other library.so])
TAG=${1:-roll}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
timeout -k 10 600 python -m pytest tests/test_cnn_gpu.py tests/test_ppo_e2e_gpu.py -m gpu -q -x -k "rollout or conv_stack or fused_rollout_act or wide_action or extreme or known" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -15 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for w in 0 4 7; do
  DX_CS_DIAG=$w DX_CS_STEP=5 timeout -k 10 120 python3 tools/cs_stamps.py 256 16 2> gpurun_out/${TAG}_stamps_w$w.log || exit 1
  tail -11 gpurun_out/${TAG}_stamps_w$w.log
done
for lib in default ${2:-default}; do
  if [ "$lib" = default ]; then unset DERL_AMD_LIBRARY; else export DERL_AMD_LIBRARY=$lib; fi
  for n in 256 128 64 32; do
    timeout -k 10 120 python3 tools/rollout_bench.py $n 128 2>&1 | tail -1 | sed "s/^/library $lib /"
  done
done
