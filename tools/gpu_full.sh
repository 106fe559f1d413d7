#!/bin/bash
# Full -m gpu suite, then an A/B of the rollout lanes at 256 envs (same box, alternating).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03e_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r03e_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for rep in 1 2 3; do
  for l in 1 2; do
    DX_ROLLOUT_LANES=$l timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $l rep $rep ms', d['ms_per_step'], 'host_unblocked', d['config']['host_enqueue_ms_unblocked'])" || exit 1
  done
done
exit $rc
