#!/bin/bash
# Runs on the GPU box: default bench line, the same command under rocprofv3 --kernel-trace --stats,
# the 8-GPU-shard-sized run (plain and under rocprofv3) and configs 3 / 5.  Outputs under
# gpurun_out/<tag>_*; copy the ones to keep into profiles/.   usage: bash tools/profile_bench.sh <tag>
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o bench -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err || exit 1
# the same command with the backward's side stream off: per-kernel durations of stages running alone
# (in the default loop a data-gradient stage shares the chip with a weight-gradient stage)
DX_BWD_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_serial -o bench -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/${TAG}_bench_under_rocprof_serial.json 2> $R/gpurun_out/${TAG}_rocprof_serial.err || exit 1
timeout -k 10 200 python3 $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $R/gpurun_out/${TAG}_shard32.json 2> /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s32prof -o s32 -- python3 $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $R/gpurun_out/${TAG}_shard32_under_rocprof.json 2> /dev/null || exit 1
# the 4- and 2-GPU shards of the same run (64 / 128 envs): the strong-scaling points of DESIGN.md section 5
timeout -k 10 200 python3 $R/bench.py --nenvs 64 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $R/gpurun_out/${TAG}_shard64.json 2> /dev/null || exit 1
timeout -k 10 200 python3 $R/bench.py --nenvs 128 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $R/gpurun_out/${TAG}_shard128.json 2> /dev/null || exit 1
timeout -k 10 200 python3 $R/tools/bench_configs.py c3 10 > $R/gpurun_out/${TAG}_c3.json 2> /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c3prof -o c3 -- python3 $R/tools/bench_configs.py c3 5 > $R/gpurun_out/${TAG}_c3_under_rocprof.json 2> /dev/null || exit 1
timeout -k 10 200 python3 $R/tools/bench_configs.py c5 100 > $R/gpurun_out/${TAG}_c5.json 2> /dev/null || exit 1
cat $R/gpurun_out/${TAG}_bench.json | head -c 6000
echo
cat $R/gpurun_out/${TAG}_shard32.json | head -c 1500
