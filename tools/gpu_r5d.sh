#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
timeout -k 10 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x -k "diagnostic_switches or extreme_magnitudes or training_forward or conv_stack or golden" --durations=5 > gpurun_out/r5d_tests2.log 2>&1 || { tail -40 gpurun_out/r5d_tests2.log; exit 1; }
tail -8 gpurun_out/r5d_tests2.log
bash tools/gpu_ab.sh "DX_CONVSTACK_TRAIN_ROLES=1" "DX_CONVSTACK_TRAIN_ROLES=0" 2 > gpurun_out/r5d_ab_roles.log 2>&1; cat gpurun_out/r5d_ab_roles.log
bash tools/gpu_stage_ab.sh r5d_dgrad 8192 default libderl_amd_base.so 2 2>&1 | grep -E "==|dgrad|conv_stack"
timeout -k 10 120 python3 tools/bench_configs.py c3 20 > gpurun_out/r5d_c3.json 2>&1; cat gpurun_out/r5d_c3.json | cut -c1-300
