"""Stress of tests/test_native_epoch_gpu.py::test_native_epoch_normalises_only_when_the_transform_opted_in:
repeats the native / per-update pair and reports the FIRST quantity that differs (minibatch
advantages as seen by the caller, parameters after each step).  usage: python3 tools/flake_native_norm.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_native_epoch_gpu import make_alg  # noqa: E402
import derl_amd as derl  # noqa: E402
from derl_amd.runners.onpolicy import IterateWithMinibatches, TransformInteractions  # noqa: E402
from derl_amd.runners.trajectory_transforms import NormalizeAdvantages  # noqa: E402


def rewire(alg, eps):
  iterate = alg.runner.runner
  assert isinstance(iterate, IterateWithMinibatches)
  if eps is None:
    alg.runner = iterate
  else:
    normalize = NormalizeAdvantages(epsilon=eps)
    iterate.prepare = normalize.prepare
    alg.runner = TransformInteractions(iterate, [normalize])
  return alg


def one(native, eps):
  alg, calls = make_alg("gaussian", native, 32, 16, 2, 4)
  rewire(alg, eps)
  it = alg.runner.run()
  trace = []
  for _ in range(8):
    data = next(it)
    derl.summary.stop_recording()
    adv = data["advantages"].clone()
    act = data["actions"].clone()
    alg.step(data)
    trace.append((adv, act, alg.model.engine.params.clone()))
  return trace, list(calls)


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for rep in range(reps):
  for eps in (None, 0.25):
    a, calls_a = one(True, eps)
    b, calls_b = one(False, eps)
    data_diff = [i for i, (x, y) in enumerate(zip(a, b)) if not (torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]))]
    final_diff = not torch.equal(a[-1][2], b[-1][2])
    if data_diff or final_diff:
      bad += 1
      # same rollout, other permutation?  the multiset of an epoch's raw actions decides
      ea = torch.cat([x[1].reshape(-1) for x in a[:4]]).sort().values
      eb = torch.cat([y[1].reshape(-1) for y in b[:4]]).sort().values
      print(f"rep {rep} eps {eps}: minibatches {data_diff} differ, final params differ: {final_diff}; "
            f"same rollout (epoch multiset of actions equal): {torch.equal(ea, eb)}; native calls {calls_a}", flush=True)
print(f"{reps} repetitions, {bad} mismatching pairs")
