#!/bin/bash
# config 3 (persistent MLP epoch): its tests, the in-kernel phase stamps and the bench line.  usage: bash tools/gpu_c3.sh <tag>
TAG=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_native_epoch_gpu.py tests/test_mlp_gpu.py -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
DX_MLP_PERSIST_STAMPS=1 timeout -k 10 200 python3 tools/bench_configs.py c3 2 > gpurun_out/${TAG}_stamps.log 2>&1 || exit 1
grep mlp_persist gpurun_out/${TAG}_stamps.log | tail -2
timeout -k 10 200 python3 tools/bench_configs.py c3 20 > gpurun_out/${TAG}_bench.log 2>&1 || exit 1
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-200
