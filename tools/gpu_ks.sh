#!/bin/bash
# Runs on the GPU box: parity of the first layer's weight gradient (conv0_wgrad_ks.hip), its stage time beside the tile
# kernel's at update and shard sizes, then its in-kernel phase stamps (diag flavour).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
T=${1:-ks}
timeout -k 10 400 python -m pytest tests/test_cnn_gpu.py -m gpu -x -q -k "first_layer_weight_gradient or test_loss_and_gradients_match_reference_golden or (test_backward_ragged_batches_with_gather and (37 or 1024))" > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${T}_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for b in 8192 1024; do
  echo "ks batch=$b"
  timeout -k 10 200 python3 tools/stage_bench.py $b 10 2>&1 | grep '"stage"' || exit 1
  echo "tile kernel batch=$b"
  DX_CONV0_KS=0 timeout -k 10 200 python3 tools/stage_bench.py $b 10 7 2>&1 | grep '"stage"' || exit 1
done
export DERL_AMD_LIBRARY=diag DX_C0_DIAG=1
for b in 8192 1024; do
  timeout -k 10 200 python3 tools/stage_bench.py $b 1 7 2>&1 | grep "conv0_wgrad_ks" | tail -2 || exit 1
done
