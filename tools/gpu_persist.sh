#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 300 python -m pytest tests/test_native_epoch_gpu.py -m gpu -q -x > gpurun_out/r03d_tests.log 2>&1
rc=$?
tail -30 gpurun_out/r03d_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -m pytest tests/test_ppo_e2e_gpu.py tests/test_mlp_gpu.py -m gpu -q -x -k "config3 or mlp" > gpurun_out/r03d_tests2.log 2>&1
rc=$?
tail -15 gpurun_out/r03d_tests2.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python3 tools/bench_configs.py c3 10 || exit 1
