#!/bin/bash
# Runs on the GPU box: parity of the first-layer kernels, then A/B timings (forward with four /
# eight waves per tile, weight gradient with / without the group-of-four offsets) at update and
# rollout sizes, then the in-kernel phase stamps of both kernels.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x > gpurun_out/conv0_tests.log 2>&1
rc=$?; tail -5 gpurun_out/conv0_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for b in 8192 1024 128; do
  echo "default batch=$b"
  timeout -k 10 200 python3 tools/stage_bench.py $b 5 0 13 2>&1 | grep '"stage"' || exit 1
  echo "waves=4 group4=0 batch=$b"
  DX_C0_WAVES=4 DX_C0_GROUP4=0 timeout -k 10 200 python3 tools/stage_bench.py $b 5 0 13 2>&1 | grep '"stage"' || exit 1
done
export DERL_AMD_LIBRARY=diag DX_C0_DIAG=1
for g in 1 0; do for b in 8192 1024; do
  DX_C0_GROUP4=$g timeout -k 10 200 python3 tools/stage_bench.py $b 1 0 13 2>&1 | grep "conv0_.*_b16" | tail -2 || exit 1
done; done
