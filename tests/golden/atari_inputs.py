"""Scripted raw-frame streams for the Atari-preprocessing golden vectors: one
stream of uint8 frames per env with episode boundaries, regenerated from a seed wherever needed."""
import numpy as np

CASES = {
    # name: (nenvs, height, width, channels or None for grayscale frames, steps, seed)
    "gray": (3, 12, 10, None, 17, 5),
    "rgb": (2, 8, 7, 3, 11, 6),
}


def stream(name):
  """Returns dict(reset (N,H,W[,C]) first frames, frames (T,N,H,W[,C]), dones (T,N) bool,
  after_done (T,N,H,W[,C]) the first frame of the next episode where dones is set)."""
  nenvs, h, w, c, steps, seed = CASES[name]
  rs = np.random.RandomState(seed)
  shape = (h, w) if c is None else (h, w, c)
  draw = lambda *lead: rs.randint(0, 256, size=lead + shape).astype(np.uint8)
  return dict(reset=draw(nenvs), frames=draw(steps, nenvs), dones=rs.uniform(size=(steps, nenvs)) < 0.2,
              after_done=draw(steps, nenvs))
