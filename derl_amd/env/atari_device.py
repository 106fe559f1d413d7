"""Atari frame pipeline of a batched env on the device: the observation
wrappers of derl's Nature-DQN stack (derl/env/make_env.py:121-137) applied to a whole batch of
raw emulator frames with three launches instead of per-env NumPy / cv2 calls:

  MaxBetweenFrames  (derl/env/atari_wrappers.py:121-137)  dx_frame_max_u8
  ImagePreprocessing (:95-118, optional)                   dx_gray_resize_u8   [parity unpinned]
  QueueFrames        (:140-163)                            dx_frame_queue_u8

Frames are uint8, batch-first ``(N, H, W[, C])``; host arrays are uploaded, device tensors used
in place.  The newest observation can be written straight into a rollout-buffer slot (``out=``),
the previous slot being the queue's state -- nothing is copied besides the shift itself."""
import numpy as np
import torch

from .. import _lib


def _u8(frames, device):
  if isinstance(frames, np.ndarray):
    frames = torch.from_numpy(np.ascontiguousarray(frames))
  if frames.dtype != torch.uint8:
    raise ValueError(f"frames must be uint8, got {frames.dtype}")
  return frames.to(device).contiguous()


class DeviceAtariFrames:
  """``preprocess=(height, width, grayscale)`` inserts derl's ImagePreprocessing between the
  maximum and the queue (None: frames are queued as they are)."""
  def __init__(self, nframes=4, concat=False, preprocess=None, device="cuda"):
    self.nframes, self.concat, self.preprocess = int(nframes), bool(concat), preprocess
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise _lib.NativeError("DeviceAtariFrames needs a HIP device; derl_amd has no CPU path")
    self.last = None
    self.observations = None

  def _check(self, frames):
    per_env = frames[0].numel()
    if per_env % 4:
      raise ValueError(f"a frame must be a multiple of 4 bytes, got shape {tuple(frames.shape[1:])}")
    if frames.ndim not in (3, 4):
      raise ValueError(f"frames must be (N, H, W) or (N, H, W, C), got {tuple(frames.shape)}")

  def _process(self, frames):
    if self.preprocess is None:
      return frames
    height, width, gray = self.preprocess
    n, h, w = frames.shape[:3]
    c = frames.shape[3] if frames.ndim == 4 else 1
    out = torch.empty((n, height, width) if gray or frames.ndim == 3 else (n, height, width, c),
                      dtype=torch.uint8, device=self.device)
    _lib.call("dx_gray_resize_u8", _lib.ptr(frames), _lib.ptr(out), n, h, w, c, int(height), int(width),
              int(bool(gray)), _lib.stream_ptr(self.device))
    return out

  def _obs_shape(self, frame):
    if self.concat:
      if frame.ndim != 4:
        raise ValueError("concat=True needs frames with a channel axis")
      return tuple(frame.shape[:3]) + (frame.shape[3] * self.nframes,)
    return tuple(frame.shape) + (self.nframes,)

  def reset(self, frames, out=None):
    """All envs start an episode: the queue holds K copies of the (preprocessed) first frame."""
    frames = _u8(frames, self.device)
    self._check(frames)
    self.last = frames.clone()
    frame = self._process(frames)
    shape = self._obs_shape(frame)
    if self.concat:
      obs = frame.repeat(1, 1, 1, self.nframes)
    else:
      obs = frame.unsqueeze(-1).expand(shape).contiguous()
    if out is not None:
      out.copy_(obs)
      obs = out
    self.observations = obs
    return obs

  def step(self, frames, dones=None, reset_frames=None, out=None):
    """One env step for the whole batch.  ``dones`` (N,) bool with ``reset_frames`` (the raw first
    frames of the episodes that start) applies the env batch's auto-reset.  Returns the new
    observations (``out`` if given: e.g. the next slot of a rollout buffer)."""
    if self.last is None:
      raise RuntimeError("call reset() first")
    frames = _u8(frames, self.device)
    if frames.shape != self.last.shape:
      raise ValueError(f"frames {tuple(frames.shape)} do not match the reset frames {tuple(self.last.shape)}")
    if (dones is None) != (reset_frames is None):
      raise ValueError("dones and reset_frames come together")
    n = frames.shape[0]
    stream = _lib.stream_ptr(self.device)
    done_u8 = reset_raw = None
    if dones is not None:
      done_u8 = torch.as_tensor(np.asarray(dones) if not isinstance(dones, torch.Tensor) else dones)
      done_u8 = done_u8.to(self.device).to(torch.uint8).contiguous()
      reset_raw = _u8(reset_frames, self.device)
    maxed = torch.empty_like(frames)
    _lib.call("dx_frame_max_u8", _lib.ptr(frames), _lib.ptr(self.last), _lib.ptr(done_u8), _lib.ptr(reset_raw),
              _lib.ptr(maxed), n, frames[0].numel(), stream)
    self.maxed = maxed
    frame = self._process(maxed)
    reset_frame = self._process(reset_raw) if reset_raw is not None else None
    prev = self.observations
    if out is None:
      out = torch.empty_like(prev)
    elif out.data_ptr() == prev.data_ptr():
      raise ValueError("out must not alias the previous observations")
    channels = frame.shape[3] if frame.ndim == 4 else 1
    _lib.call("dx_frame_queue_u8", _lib.ptr(prev), _lib.ptr(frame), _lib.ptr(done_u8), _lib.ptr(reset_frame),
              _lib.ptr(out), n, frame[0].numel(), channels, self.nframes, int(self.concat), stream)
    self.observations = out
    return out
