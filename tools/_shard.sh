set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/shard32.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s32 -o s32 -- python3 $R/bench.py --nenvs 32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/shard32_prof.json
python $R/tools/host_profile.py 32 10 > $R/gpurun_out/shard32_host.txt
