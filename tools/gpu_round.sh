#!/bin/bash
# Runs on the GPU box: a pytest selection of the -m gpu suite (bounded), then -- unless it was
# killed -- the default bench line and the 8-GPU-shard-sized bench.
# usage: bash tools/gpu_round.sh <tag> [pytest args ...]
TAG=${1:-t}
shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
SEL=${@:-tests}
timeout -k 10 900 python -m pytest $SEL -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -25 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed ($rc): no further GPU step"; exit $rc; fi
timeout -k 10 200 python3 bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_shard32.json 2> gpurun_out/${TAG}_shard32.err || exit 1
cat gpurun_out/${TAG}_shard32.json
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_quick.json 2> gpurun_out/${TAG}_bench_quick.err || exit 1
cat gpurun_out/${TAG}_bench_quick.json
exit $rc
