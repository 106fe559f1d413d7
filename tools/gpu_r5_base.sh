#!/bin/bash
# round-5 baseline: GPU suite with per-test durations, phase stamps of the one-launch training forward, trunk timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=70 > gpurun_out/r5_base_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5_base_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed ($rc)"; exit $rc; fi
for w in 0 4 7; do
  DX_CS_DIAG=$w DX_CS_STEP=5 timeout -k 10 120 python3 tools/cs_stamps.py 8192 0 2> gpurun_out/r5_base_stamps_w$w.log || exit 1
done
timeout -k 10 120 python3 tools/trunk_bench.py 8192 1024 > gpurun_out/r5_base_trunk.log 2>&1 || exit 1
cat gpurun_out/r5_base_trunk.log
exit $rc
