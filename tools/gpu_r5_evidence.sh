#!/bin/bash
# Round-5 evidence, part 2 (counters and power; part 1 = tools/profile_bench.sh r05): PMC traffic and SQ counters of
# every stage and of the rollout kernel (separate --pmc passes, counters only), the power probe over every stage + the
# rollout launch, the bare matrix-instruction loops, the in-kernel clocks under sustained load.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"
SQ3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
bash tools/pmc_passes.sh r05pmc 8192 "FETCH_SIZE" "WRITE_SIZE" || exit 1
echo "pmc traffic passes done"
bash tools/gpu_actpmc.sh r05act 256 "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/r05act.log 2>&1 || exit 1
echo "act traffic passes done"
bash tools/pmc_passes.sh r05sq 8192 "$SQ1" "$SQ2" "$SQ3" || exit 1
echo "sq passes done"
bash tools/gpu_actpmc.sh r05actsq 256 "$SQ1" "$SQ2" "$SQ3" > gpurun_out/r05actsq.log 2>&1 || exit 1
echo "act sq passes done"
timeout -k 10 300 python3 tools/power_probe.py 8192 3 gpurun_out/r05_power_rows.json > gpurun_out/r05_power.log 2>&1 || { tail -5 gpurun_out/r05_power.log; exit 1; }
echo "power probe done"
timeout -k 10 120 python3 tools/sustained_clock.py 3 > gpurun_out/r05_clock.log 2>&1; cat gpurun_out/r05_clock.log
timeout -k 10 120 tools/ubench/mfma_power > gpurun_out/r05_mfma_power.txt 2>&1; tail -6 gpurun_out/r05_mfma_power.txt
timeout -k 10 60 tools/ubench/mfma_issue > gpurun_out/r05_mfma_issue.txt 2>&1; cat gpurun_out/r05_mfma_issue.txt
