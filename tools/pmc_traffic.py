"""Turns the two rocprofv3 --pmc passes of tools/stage_bench.py (FETCH_SIZE and WRITE_SIZE, separate
runs as MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc_traffic.json: HBM-side bytes per
launch of every network stage.  Counter unit: KiB; FETCH_SIZE is doubled (gfx950 tallies the 128-B
requests of wide streaming reads at 64 B).

  bash tools/pmc_passes.sh <tag>pmc 8192 "FETCH_SIZE" "WRITE_SIZE"
  python3 tools/pmc_traffic.py gpurun_out/<tag>pmc_0 gpurun_out/<tag>pmc_1 profiles/<tag>_pmc_traffic.json \
      [commit [gpurun_out/<tag>act_0 gpurun_out/<tag>act_1]]     (the last two: bash tools/gpu_actpmc.sh <tag>act 256 FETCH_SIZE WRITE_SIZE)"""
import collections
import csv
import glob
import json
import os
import sys

STAGE_OF = [  # substring of the kernel name -> stage
    ("conv0_fwd_b16", "conv0_fwd"), ("igemm_nt_kernel<1,", "conv1_fwd"), ("igemm_nt_kernel<2,", "conv2_fwd"),
    ("ntp_kernel<1,", "conv1_fwd"), ("ntp_kernel<2,", "conv2_fwd"), ("ntp_kernel<3,", "fc_fwd"), ("ntp_kernel<8,", "fc_dgrad"),
    ("ntp_kernel<10,", "conv2_dgrad"), ("ntp_kernel<12,", "conv1_dgrad"), ("nt_dma_kernel<1,", "fc_fwd"),
    ("nt_dma_kernel<3,", "fc_dgrad"), ("igemm_nt_small_kernel<3,", "fc_fwd"), ("fc_wgrad_kernel", "fc_wgrad"), ("colsum_kernel", "fc_wgrad_bias"),
    ("igemm_tn_kernel<7,", "fc_wgrad"), ("igemm_nt_small_kernel<8,", "fc_dgrad"),
    ("conv_wgrad_direct_kernel<9,", "conv2_wgrad"), ("igemm_tn_kernel<9,", "conv2_wgrad"),
    ("igemm_nt_pix_kernel<10,", "conv2_dgrad"), ("conv_wgrad_direct_kernel<11,", "conv1_wgrad"),
    ("igemm_tn_kernel<11,", "conv1_wgrad"), ("igemm_nt_pix_kernel<12,", "conv1_dgrad"),
    ("conv0_wgrad_b16", "conv0_wgrad"), ("conv0_wgrad_ks", "conv0_wgrad"),  # round 6: the K-split kernel (conv0_wgrad_ks.hip)
    ("finalize_fused_kernel", "finalize"), ("permute_reduce_kernel", "finalize"),
    ("permute_reduce_greduce_kernel", "finalize_with_tail_greduce"),  # round 5: the training loop's merged slab reduction
    ("fc_row_unpermute_reduce", "finalize_fc"),
    # the factored tail (csrc/tail.hip, heads.hip) and the rollout's one-kernel step (csrc/convstack.hip)
    ("tail_loss_bwd_kernel", "tail_loss_bwd"),  # round 5: loss + the backward pass over y2 in one launch
    ("tail_loss_kernel", "tail_loss"), ("tail_bwd_kernel", "tail_bwd"), ("tail_greduce_kernel", "tail_greduce"),
    ("tail_grads_kernel", "tail_grads_products"),
    # round 4: the update's conv stages on the bf16 matrix cores (convstack.hip `train`, wgrad_b6.hip, dgrad_b6.hip)
    ("convstack_image_kernel<true>", "conv_stack_fwd"), ("convstack_image_kernel<(bool)1>", "conv_stack_fwd"),
    ("convstack_train_kernel", "conv_stack_fwd"),  # round 5: the role-specialised training forward (convstack_train.hip)
    ("conv_wgrad_b6_kernel<1>", "conv1_wgrad"), ("conv_wgrad_b6_kernel<2>", "conv2_wgrad"),
    ("conv2_wgrad_stream_kernel", "conv2_wgrad"),  # round 6: the pixel contraction streamed across images
    ("conv_dgrad_b6_kernel<1>", "conv1_dgrad"), ("conv_dgrad_b6_kernel<2>", "conv2_dgrad"),
]
ROLLOUT_STEP = "rollout_step (convstack_roll_kernel, 256 images, one step)"


def is_rollout_kernel(name):
  """The ROLLOUT kernel of the conv stack only: convstack_roll_kernel (round 5; the training forward is its own symbol,
  convstack_train_kernel, now).  Round 4's tool matched any `convstack_image` symbol and the training flavour --
  launched by the same tool's warm-up -- won."""
  return "convstack_roll_kernel" in name or "convstack_image_kernel<false>" in name or "convstack_image_kernel<(bool)0>" in name


def mean_per_kernel(root, counter):
  acc = collections.defaultdict(lambda: [0, 0.0])
  for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        if row["Counter_Name"] != counter:
          continue
        acc[row["Kernel_Name"]][0] += 1
        acc[row["Kernel_Name"]][1] += float(row["Counter_Value"])
  return {k: v[1] / v[0] for k, v in acc.items()}


def main(fetch_dir, write_dir, out_path, commit=None, act_fetch_dir=None, act_write_dir=None):
  fetch = mean_per_kernel(fetch_dir, "FETCH_SIZE")
  write = mean_per_kernel(write_dir, "WRITE_SIZE")
  stages = {}
  if act_fetch_dir and act_write_dir:  # the passes over tools/act_bench.py 256 (the rollout's act step: the same kernel symbol)
    af = {k: v for k, v in mean_per_kernel(act_fetch_dir, "FETCH_SIZE").items() if is_rollout_kernel(k)}
    aw = {k: v for k, v in mean_per_kernel(act_write_dir, "WRITE_SIZE").items() if is_rollout_kernel(k)}
    for kernel in af:
      raw, wr = af[kernel] * 1024, aw.get(kernel, 0.0) * 1024
      stages[ROLLOUT_STEP] = {"kernel": kernel, "FETCH_SIZE_bytes_raw": int(raw), "FETCH_SIZE_bytes_x2_gfx950": int(2 * raw),
                              "WRITE_SIZE_bytes": int(wr), "hbm_bytes": int(2 * raw + wr)}
  for kernel in sorted(set(fetch) | set(write)):
    stage = next((s for key, s in STAGE_OF if key in kernel), None)
    if stage is None:
      continue
    raw = fetch.get(kernel, 0.0) * 1024
    wr = write.get(kernel, 0.0) * 1024
    stages[stage] = {"kernel": kernel, "FETCH_SIZE_bytes_raw": int(raw), "FETCH_SIZE_bytes_x2_gfx950": int(2 * raw),
                     "WRITE_SIZE_bytes": int(wr), "hbm_bytes": int(2 * raw + wr)}
  doc = {"what": "HBM-side traffic per launch at minibatch 8192 (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                 "separate passes over tools/stage_bench.py 8192; counter unit KiB; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)",
         "commit": commit,
         "stages": stages}
  with open(out_path, "w") as f:
    json.dump(doc, f, indent=1)
  for name, s in stages.items():
    print(f"{name:16s} fetch {s['FETCH_SIZE_bytes_x2_gfx950'] / 1e6:8.1f} MB  write {s['WRITE_SIZE_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
  main(*sys.argv[1:7])
