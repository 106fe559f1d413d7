#!/bin/bash
# Runs on the GPU box: the -m gpu suite (bounded), then -- unless the suite was killed -- a
# rocprofv3 kernel trace of the 8-GPU-shard-sized bench.  usage: bash tools/gpu_round.sh <tag>
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -15 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed ($rc): no further GPU step"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s32prof -o s32 -- python3 $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_shard32_rocprof.json 2> $R/gpurun_out/${TAG}_shard32_rocprof.err
echo "rocprof rc $?"
exit $rc
