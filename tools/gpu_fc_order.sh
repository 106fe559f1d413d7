#!/bin/bash
# fc layer tile walk (DX_NTP_ROWS_INNER 0 / 1): parity, stage times, FETCH / WRITE counters.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -x > gpurun_out/r03c_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03c_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 0 1; do
  echo "ROWS_INNER=$v"
  DX_NTP_ROWS_INNER=$v timeout -k 10 200 python3 tools/stage_bench.py 8192 10 3 8 || exit 1
done
bash tools/pmc_passes.sh r03pmc 8192 "FETCH_SIZE" "WRITE_SIZE" && python3 tools/pmc_traffic.py gpurun_out/r03pmc_0 gpurun_out/r03pmc_1 gpurun_out/r03_pmc_traffic.json && cat gpurun_out/r03_pmc_traffic.json | head -c 3000
