#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
timeout -k 10 900 python -m pytest tests/test_native_epoch_gpu.py tests/test_ppo_e2e_gpu.py tests/test_reference_known_answers_gpu.py tests/test_update_ops_gpu.py tests/test_distributed.py tests/test_mlp_gpu.py -m gpu -q -x --durations=12 > gpurun_out/r5c_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r5c_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x -k "diagnostic_switches or extreme_magnitudes" --durations=5 > gpurun_out/r5c_tests2.log 2>&1 || { tail -40 gpurun_out/r5c_tests2.log; exit 1; }
tail -8 gpurun_out/r5c_tests2.log
bash tools/gpu_ab.sh "DX_CONVSTACK_TRAIN_ROLES=1" "DX_CONVSTACK_TRAIN_ROLES=0" 2 > gpurun_out/r5c_ab_roles.log 2>&1; cat gpurun_out/r5c_ab_roles.log
bash tools/gpu_stage_ab.sh r5c_dgrad 8192 default libderl_amd_base.so 2 2>&1 | grep -E "==|dgrad|conv_stack"
timeout -k 10 120 python3 tools/bench_configs.py c3 20 > gpurun_out/r5c_c3.json 2>&1; cat gpurun_out/r5c_c3.json | cut -c1-300
