"""Copies the judged summaries of a round's evidence runs (tools/profile_bench.sh <tag>, tools/gpu_evidence.sh <tag>) from
gpurun_out/ (scratch) into profiles/ (tracked), and derives the two documents bench.py reads: <tag>_pmc_traffic.json and
<tag>_power.json.     usage: python3 tools/collect_round.py <tag> [commit]        (round 5's copy: tools/r5_collect.py)"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
TAG = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True,
                                                              text=True).stdout.strip()


def copy(src, dst):
  found = glob.glob(os.path.join(G, src))
  if not found:
    print("missing", src)
    return False
  shutil.copyfile(found[0], os.path.join(P, dst))
  print("copied", src, "->", dst)
  return True


for name in ("bench", "bench_under_rocprof", "bench_under_rocprof_serial", "shard32", "shard32_under_rocprof", "shard64", "shard128", "c3",
             "c3_under_rocprof", "c5"):
  copy(f"{TAG}_{name}.json", f"{TAG}_{name}.json")
copy(TAG + "_prof/**/bench_kernel_stats.csv", TAG + "_bench_kernel_stats.csv") or copy(TAG + "_prof/bench_kernel_stats.csv", TAG + "_bench_kernel_stats.csv")
copy(TAG + "_prof_serial/bench_kernel_stats.csv", TAG + "_bench_kernel_stats_serial.csv")
copy(TAG + "_prof/bench_domain_stats.csv", TAG + "_bench_domain_stats.csv")
copy(TAG + "_s32prof/s32_kernel_stats.csv", TAG + "_shard32_kernel_stats.csv")
copy(TAG + "_c3prof/c3_kernel_stats.csv", TAG + "_c3_kernel_stats.csv")
copy(TAG + "_mfma_power.txt", TAG + "_mfma_power.txt")
copy(TAG + "_mfma_issue.txt", TAG + "_mfma_issue.txt")

# HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE)
if os.path.isdir(os.path.join(G, TAG + "pmc_0")):
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import pmc_traffic
  pmc_traffic.main(os.path.join(G, TAG + "pmc_0"), os.path.join(G, TAG + "pmc_1"), os.path.join(P, TAG + "_pmc_traffic.json"), commit,
                   os.path.join(G, TAG + "act_0"), os.path.join(G, TAG + "act_1"))


# SQ counters per kernel: matrix pipe busy, LDS bank conflicts, executed bf16 matrix flops
def counters(prefix, passes):
  acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
  for i in range(passes):
    for path in glob.glob(os.path.join(G, f"{prefix}_{i}", "**", "*counter_collection.csv"), recursive=True):
      with open(path) as f:
        for row in csv.DictReader(f):
          slot = acc[row["Kernel_Name"]][row["Counter_Name"]]
          slot[0] += 1
          slot[1] += float(row["Counter_Value"])
  return {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}


KEEP = ("convstack", "conv_wgrad_b6", "conv2_wgrad_stream", "conv_dgrad_b6", "conv0_wgrad_b16", "conv0_wgrad_ks", "tail_loss", "tail_bwd", "tail_grads", "tail_greduce")
lines = ["# rocprofv3 --pmc passes (counters only, three groups, separate runs) at commit " + commit + ":",
         "#   bash tools/gpu_evidence.sh " + TAG + "   (tools/pmc_passes.sh <tag>sq 8192 ... over tools/stage_bench.py: the update's stages at minibatch 8192;",
         "#   tools/gpu_actpmc.sh <tag>actsq 256 ... over tools/act_bench.py: the rollout kernel, one act step of 256 envs)",
         "# mean per dispatch.  pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 shader engines;",
         "# lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; executed bf16 matrix flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512"]
for prefix in (TAG + "sq", TAG + "actsq"):
  for kernel, cs in sorted(counters(prefix, 3).items()):
    if not any(k in kernel for k in KEEP):
      continue
    busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(cs.get("SQ_BUSY_CYCLES", 0) / 32, 1)
    conflict = cs.get("SQ_LDS_BANK_CONFLICT", 0) / max(cs.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    lines.append(f"{kernel[:110]}")
    lines.append(f"    pipe_busy {busy:.3f}  lds_conflict {conflict:.3f}  executed_bf16_GFLOP {cs.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512 / 1e9:.1f}")
    lines.append("    " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(cs.items())))
with open(os.path.join(P, TAG + "_pmc_sq_counters.txt"), "w") as f:
  f.write("\n".join(lines) + "\n")
print("\n".join(lines[5:25]))

# power: the probe's rows + the in-kernel clocks under sustained load
rows_path = os.path.join(G, TAG + "_power_rows.json")
if os.path.exists(rows_path):
  with open(rows_path) as f:
    doc = json.load(f)
  doc["commit"] = commit
  clocks = []
  clock_log = os.path.join(G, TAG + "_clock.log")
  if os.path.exists(clock_log):
    for line in open(clock_log):
      try:
        clocks.append(json.loads(line))
      except ValueError:
        pass
  doc["in_kernel_clock"] = {"what": "s_memtime / s_memrealtime around ONE stamped launch right after seconds of back-to-back launches of the same "
                                    "kernel (tools/sustained_clock.py, diag flavour)", "rows": clocks}
  with open(os.path.join(P, TAG + "_power.json"), "w") as f:
    json.dump(doc, f, indent=1)
  print("wrote profiles/" + TAG + "_power.json")
