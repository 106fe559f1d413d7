"""Golden vectors of derl's MaxBetweenFrames and QueueFrames (derl/env/atari_wrappers.py:121-163):
run HERE (where /root/reference exists) with `python -m tests.golden.generate_atari`; writes
tests/golden/atari_frames.npz.  The two wrappers are plain NumPy classes, so they run unmodified on
scripted frame streams; the env beneath them is a stand-in that replays the stream.  Per env, the
stack is QueueFrames(MaxBetweenFrames(env)) driven the way gym's ObservationWrapper drives it
(step -> observation(obs); a finished episode -> reset()), as EnvBatch does (env_batch.py:66-70).
ImagePreprocessing (cv2) sits between the two in derl's stack and is NOT pinned (cv2 is absent)."""
import os

import numpy as np

from . import _ref_import
from .atari_inputs import CASES, stream

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "atari_frames.npz")


class ReplayEnv:
  """Replays one env's column of a scripted stream."""
  spec = None

  def __init__(self, data, index, space):
    self.data, self.index, self.t = data, index, -1
    self.observation_space = space
    self.action_space = None
    self.unwrapped = self

  def reset(self, **kwargs):
    del kwargs
    return self.data["reset"][self.index] if self.t < 0 else self.data["after_done"][self.t, self.index]

  def step(self, action):
    del action
    self.t += 1
    return self.data["frames"][self.t, self.index], 0.0, bool(self.data["dones"][self.t, self.index]), {}


def main():
  derl = _ref_import.import_reference()
  from derl.env.atari_wrappers import MaxBetweenFrames, QueueFrames  # pylint: disable=import-error
  import gym.spaces as spaces  # the stand-in Box
  del derl
  result = {}
  for name, (nenvs, h, w, c, steps, _) in CASES.items():
    data = stream(name)
    shape = (h, w) if c is None else (h, w, c)
    space = spaces.Box(np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8), shape, np.uint8)
    for concat in ((False,) if c is None else (False, True)):
      envs = [ReplayEnv(data, i, space) for i in range(nenvs)]
      maxed = [MaxBetweenFrames(env) for env in envs]
      queued = [QueueFrames(m, 4, concat=concat) for m in maxed]
      # reset: QueueFrames.reset -> MaxBetweenFrames.reset -> env.reset
      first = np.stack([q.reset() for q in queued])
      outs, maxes = [], []
      for t in range(steps):
        row, mrow = [], []
        for q, m, env in zip(queued, maxed, envs):
          raw, _, done, _ = env.step(None)
          mx = m.observation(raw)          # gym.ObservationWrapper.step of MaxBetweenFrames
          ob = q.observation(mx)           # ... of QueueFrames
          if done:                         # EnvBatch.step: ob = env.reset()
            ob = q.reset()
          row.append(ob)
          mrow.append(mx)
        outs.append(np.stack(row))
        maxes.append(np.stack(mrow))
      tag = f"{name}.{'concat' if concat else 'stack'}"
      result[f"{tag}.reset"] = first
      result[f"{tag}.obs"] = np.stack(outs)
      result[f"{tag}.max"] = np.stack(maxes)
  np.savez_compressed(OUT, **result)
  print("wrote", OUT, {k: v.shape for k, v in result.items()})


if __name__ == "__main__":
  main()
