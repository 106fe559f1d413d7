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


@pytest.mark.gpu
def test_two_ranks_one_gpu_step_matches_single_process():
  launch("gpu_step", 29512)


@pytest.mark.gpu
def test_minibatch_advantage_statistics_with_one_all_reduce_per_rollout():
  launch("gpu_minibatch_stats", 29513)


@pytest.mark.gpu
def test_two_rank_data_parallel_ppo_learns_and_replicas_stay_identical():
  launch("gpu_learns", 29514)
