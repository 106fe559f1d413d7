#!/bin/bash
# A round's evidence, part 2 (counters and power; part 1 = tools/profile_bench.sh <tag>); usage: bash tools/gpu_evidence.sh <tag> [parts]
# parts: any of "traffic sq power" (default all) -- a call must stay inside gpurun's 20 minutes: PMC traffic and SQ counters of
# every stage and of the rollout kernel (separate --pmc passes, counters only), the power probe over every stage + the
# rollout launch, the bare matrix-instruction loops, the in-kernel clocks under sustained load.
TAG=${1:-r06}; PARTS=${2:-"traffic sq power"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
SQ1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"
SQ3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
[[ "$PARTS" == *traffic* ]] && { bash tools/pmc_passes.sh ${TAG}pmc 8192 "FETCH_SIZE" "WRITE_SIZE" || exit 1; }
[[ "$PARTS" == *traffic* ]] && echo "pmc traffic passes done"
[[ "$PARTS" == *traffic* ]] && { bash tools/gpu_actpmc.sh ${TAG}act 256 "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/${TAG}act.log 2>&1 || exit 1; }
[[ "$PARTS" == *traffic* ]] && echo "act traffic passes done"
[[ "$PARTS" == *sq* ]] && { bash tools/pmc_passes.sh ${TAG}sq 8192 "$SQ1" "$SQ2" "$SQ3" || exit 1; }
[[ "$PARTS" == *sq* ]] && echo "sq passes done"
[[ "$PARTS" == *sq* ]] && { bash tools/gpu_actpmc.sh ${TAG}actsq 256 "$SQ1" "$SQ2" "$SQ3" > gpurun_out/${TAG}actsq.log 2>&1 || exit 1; }
[[ "$PARTS" == *sq* ]] && echo "act sq passes done"
if [[ "$PARTS" == *power* ]]; then
timeout -k 10 300 python3 tools/power_probe.py 8192 10 gpurun_out/${TAG}_power_rows.json > gpurun_out/${TAG}_power.log 2>&1 || { tail -5 gpurun_out/${TAG}_power.log; exit 1; }
echo "power probe done"
timeout -k 10 120 python3 tools/sustained_clock.py 3 > gpurun_out/${TAG}_clock.log 2>&1; cat gpurun_out/${TAG}_clock.log
timeout -k 10 120 tools/ubench/mfma_power > gpurun_out/${TAG}_mfma_power.txt 2>&1; tail -6 gpurun_out/${TAG}_mfma_power.txt
timeout -k 10 60 tools/ubench/mfma_issue > gpurun_out/${TAG}_mfma_issue.txt 2>&1; cat gpurun_out/${TAG}_mfma_issue.txt
fi
