#!/bin/bash
# Rollout lanes at multi-GPU shard sizes: iteration time of bench.py --nenvs N for DX_ROLLOUT_LANES = 1, 2, 4.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for n in 32 64 128 256; do
  for l in 1 2 4; do
    DX_ROLLOUT_LANES=$l timeout -k 10 120 python3 bench.py --nenvs $n --steps 15 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nenvs $n lanes $l ms', d['ms_per_step'], 'host_unblocked', d['config']['host_enqueue_ms_unblocked'])" || exit 1
  done
done
