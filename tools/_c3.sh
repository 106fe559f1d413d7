set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_configs.py c3 5 > $R/gpurun_out/c3.txt 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3b -o c3 -- python3 $R/tools/bench_configs.py c3 3 > $R/gpurun_out/c3_prof.txt 2>&1
python $R/tools/host_profile_c3.py > $R/gpurun_out/c3_host.txt 2>&1
