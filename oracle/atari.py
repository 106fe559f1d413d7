"""Atari frame pipeline of a batched env (CPU oracle, NumPy).  TEST INFRASTRUCTURE ONLY.

Follows derl/env/atari_wrappers.py:121-137 (MaxBetweenFrames), :140-163 (QueueFrames) with the
auto-reset of derl/env/env_batch.py:66-70 applied per env; pinned to vectors recorded from the
reference's classes (tests/golden/atari_frames.npz).  ``gray_resize`` restates :95-118
(ImagePreprocessing) as the reference's cv2 calls resolve in a default build -- BT.601 luma in
14-bit fixed point, bilinear resize because cv2.INTER_AREA is passed in the ``dst`` position --
but cv2 is absent from this image: PARITY UNPINNED for that step."""
import numpy as np


class FramePipeline:
  """Batch-first state of MaxBetweenFrames + QueueFrames for N envs."""
  def __init__(self, nframes=4, concat=False):
    self.nframes, self.concat = nframes, concat
    self.last = None    # (N, *frame): MaxBetweenFrames.last_obs per env
    self.queue = None   # list of the nframes most recent frames, oldest first

  def _join(self):
    return np.concatenate(self.queue, -1) if self.concat else np.stack(self.queue, -1)

  def reset(self, frames):
    """MaxBetweenFrames.reset (:135-137) then QueueFrames.reset (:158-163): K copies."""
    self.last = np.array(frames)
    self.queue = [np.array(frames) for _ in range(self.nframes)]
    return self._join()

  def step(self, frames, dones=None, reset_frames=None):
    """One env step: returns (observations, maxed frames).  Envs with ``dones`` set restart from
    ``reset_frames`` (what EnvBatch.step's env.reset() returns through the wrapper stack)."""
    maxed = np.maximum(frames, self.last)       # :130-133
    self.last = np.array(frames)
    self.queue = self.queue[1:] + [maxed.copy()]  # deque(maxlen=K).append (:152-153)
    if dones is not None and np.any(dones):
      for n in np.flatnonzero(dones):
        self.last[n] = reset_frames[n]
        for frame in self.queue:
          frame[n] = reset_frames[n]
    return self._join(), maxed


def gray_resize(frames, out_h=84, out_w=84, gray=True):
  """(N, H, W, 3 or 1) uint8 -> (N, out_h, out_w[, C]) uint8; see the module docstring."""
  x = np.asarray(frames)
  if gray and x.shape[-1] == 3:
    x = ((x[..., 0].astype(np.int64) * 4899 + x[..., 1].astype(np.int64) * 9617 +
          x[..., 2].astype(np.int64) * 1868 + 8192) >> 14)[..., None]
  elif gray:
    x = x[..., :1]
  x = x.astype(np.float32)
  H, W = x.shape[1:3]
  f32 = np.float32
  def fma(a, b, c):  # float32 fused multiply-add: the product of two float32 is exact in float64
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)
  fy = np.clip(fma(np.arange(out_h, dtype=f32) + f32(0.5), f32(H) / f32(out_h), f32(-0.5)), 0, H - 1).astype(f32)
  fx = np.clip(fma(np.arange(out_w, dtype=f32) + f32(0.5), f32(W) / f32(out_w), f32(-0.5)), 0, W - 1).astype(f32)
  y0, x0 = fy.astype(np.int64), fx.astype(np.int64)
  y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
  wy, wx = (fy - y0).astype(f32)[None, :, None, None], (fx - x0).astype(f32)[None, None, :, None]
  s00, s01 = x[:, y0][:, :, x0], x[:, y0][:, :, x1]
  s10, s11 = x[:, y1][:, :, x0], x[:, y1][:, :, x1]
  top = fma(np.broadcast_to(wx, s00.shape), s01 - s00, s00)
  bot = fma(np.broadcast_to(wx, s00.shape), s11 - s10, s10)
  v = fma(np.broadcast_to(wy, s00.shape), bot - top, top)
  out = np.clip(np.floor(v + f32(0.5)), 0, 255).astype(np.uint8)
  return out[..., 0] if gray else out
