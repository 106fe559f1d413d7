"""world_size-2 coverage of the data-parallel path (gloo): the sharding rules on the CPU with
the oracle as compute, and -- on the GPU box -- two ranks sharing the one GPU through the HIP
kernels against a single-process step."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(mode, port):
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.join(ROOT, "tests", "dist_worker.py"), mode]
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert out.stdout.count(f"{mode} OK") == 2, out.stdout[-2000:]


def test_sharding_rules_world_size_2_gloo_cpu():
  launch("cpu_math", 29511)


def test_native_comm_bootstrap_ranks_agree_on_failure_world_size_2_gloo_cpu():
  """A library failure on ONE rank during the communicator bootstrap (unique id on rank 0, or
  dx_comm_init on either rank) makes BOTH ranks fall back, and no rank is left in a collective:
  dist_worker.bootstrap_agreement."""
  launch("bootstrap_agreement", 29518)


def test_native_comm_bootstrap_with_one_rank_really_unable_to_load_rccl_world_size_2_gloo_cpu():
  """No scripted stubs: rank 1's library is pointed at an RCCL that does not exist, so its readiness check
  (dx_comm_available) really fails in dlopen; neither rank may enter ncclCommInitRank
  (dist_worker.bootstrap_real_failure)."""
  launch("bootstrap_real_failure", 29520)


@pytest.mark.gpu
def test_two_ranks_one_gpu_step_matches_single_process():
  launch("gpu_step", 29512)


@pytest.mark.gpu
def test_minibatch_advantage_statistics_with_one_all_reduce_per_rollout():
  launch("gpu_minibatch_stats", 29513)


@pytest.mark.gpu
def test_two_rank_data_parallel_ppo_learns_and_replicas_stay_identical():
  launch("gpu_learns", 29514)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["-1", "1"])
def test_rccl_communicator_behind_the_c_abi_runs_the_data_path_at_world_size_1(overlap):
  """dx_comm_init / dx_allreduce_grads / dx_allreduce_sum_f64 / dx_comm_broadcast_f32 on a real
  RCCL communicator (one rank: all this box has), bootstrapped over torch.distributed's nccl
  backend, with the sharded code path forced: see dist_worker.rccl_one_rank.  Second case: the
  backward's side stream forced on at the test's small minibatches (at 2-4 GPUs a shard's
  minibatch is >= 2,048 samples and the all-reduces sit between two overlapped halves)."""
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29516", RANK="0", WORLD_SIZE="1",
             LOCAL_RANK="0", DERL_AMD_FORCE_COLLECTIVES="1", OMP_NUM_THREADS="2", DX_BWD_OVERLAP=overlap)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "rccl_one_rank"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert "rccl_one_rank OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["-1", "1"])
def test_gradient_all_reduces_are_ordered_against_the_backward_and_the_optimizer_step(overlap):
  """What world size 1 cannot show with an identity all-reduce: the diag flavour's
  DX_COMM_TEST_HOOK delays and doubles every reduced piece on the communicator's stream, so the
  update is right only if the reductions run after the backward wrote their piece and before the
  norm / Adam read it (dist_worker.rccl_ordering)."""
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1",
             LOCAL_RANK="0", DERL_AMD_FORCE_COLLECTIVES="1", OMP_NUM_THREADS="2", DX_BWD_OVERLAP=overlap,
             DERL_AMD_LIBRARY="diag", DX_COMM_TEST_HOOK="3000:2")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "rccl_ordering"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert "rccl_ordering OK" in out.stdout


@pytest.mark.gpu
def test_bench_two_ranks_reports_ranks_seen_and_refuses_non_rccl_backend():
  """bench.py under torch.distributed.run: the JSON line carries what lets a reader check the
  multi-GPU run (ranks_seen from an all-reduce of ones, the backend, the all-reduce bytes per
  update).  Two ranks share this box's one GPU, so the backend is gloo: refused without
  --allow-gloo (a scaling number must come from RCCL), accepted as a rehearsal with it."""
  import json
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", DERL_AMD_DIST_BACKEND="gloo")
  base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
          "--master-addr", "127.0.0.1", "--master-port", "29515", os.path.join(ROOT, "bench.py"),
          "--gpus", "2", "--steps", "1", "--warmup", "1", "--nenvs", "16", "--nsteps", "8",
          "--no-roofline", "--no-cpu-baseline"]
  refused = subprocess.run(base, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert refused.returncode != 0 and "RCCL" in refused.stderr + refused.stdout
  out = subprocess.run(base + ["--allow-gloo"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  cfg = d["config"]
  assert d["n_gpus"] == 2 and cfg["ranks_seen"] == 2 and cfg["backend"] == "gloo"
  assert cfg["nenvs_per_gpu"] == 8 and cfg["parallelism"] == "dp2" and d["scaling"] == "strong"
  reduce = cfg["allreduce_bytes_per_update"]
  assert reduce["issued"] and reduce["gradient_total"] == 4 * 1_686_693 == sum(reduce["pieces"])
  mismatch = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                            env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT, capture_output=True, text=True,
                            timeout=600)
  assert mismatch.returncode != 0 and "WORLD_SIZE" in mismatch.stderr


def _plain_env():
  """The driver's environment: nothing of torch.distributed.run's contract set."""
  env = dict(os.environ, OMP_NUM_THREADS="2")
  for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DERL_AMD_DIST_BACKEND",
              "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
    env.pop(key, None)
  return env


@pytest.mark.gpu
def test_bench_started_plainly_with_gpus_2_spawns_its_own_ranks():
  """What the driver types: `python bench.py --gpus 2 ...` with no WORLD_SIZE.  bench.py starts its two ranks itself
  (torch.distributed.run as a fresh child, before anything touches the GPU) and forwards rank 0's one line; on this
  one-GPU box the ranks share the GPU and reduce over gloo (--allow-gloo), which the line says."""
  import json
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-gloo", "--steps", "1",
                        "--warmup", "1", "--nenvs", "16", "--nsteps", "8", "--no-roofline", "--no-cpu-baseline"],
                       env=_plain_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  lines = out.stdout.splitlines()
  assert len(lines) == 1, out.stdout[-2000:]  # ONE JSON line on stdout, everything else on stderr
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["backend"] == "gloo"
  assert "rehearsal" in d["config"]["collectives"]


def test_bench_started_plainly_forwards_its_ranks_failure():
  """The same plain command where the ranks cannot run (no GPU in the CPU suite's container; on a GPU box an
  impossible shard: 3 envs over 2 ranks): the launcher exits non-zero with the ranks' own error on stderr and
  prints no result line."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-gloo", "--steps", "1",
                        "--warmup", "0", "--nenvs", "3", "--nsteps", "8", "--no-roofline", "--no-cpu-baseline"],
                       env=_plain_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode != 0
  assert out.stdout.strip() == "", out.stdout[-2000:]
  assert "torch.distributed.run" in out.stderr and "failed with exit code" in out.stderr
