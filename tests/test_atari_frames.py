"""Atari frame pipeline: MaxBetweenFrames + QueueFrames against vectors recorded
from the reference's own classes (tests/golden/generate_atari.py), for the CPU oracle and -- on the
GPU -- the device kernels through the C-ABI, bit for bit; the gray / resize step (cv2 in the
reference, absent here) is checked device-vs-oracle only and is marked parity-unpinned."""
import os

import numpy as np
import numpy.testing as nt
import pytest

from oracle.atari import FramePipeline, gray_resize
from tests.golden.atari_inputs import CASES, stream

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "atari_frames.npz"))
VARIANTS = [("gray", False), ("rgb", False), ("rgb", True)]


@pytest.mark.parametrize("name,concat", VARIANTS)
def test_oracle_matches_reference(name, concat):
  data = stream(name)
  tag = f"{name}.{'concat' if concat else 'stack'}"
  pipe = FramePipeline(4, concat)
  nt.assert_array_equal(pipe.reset(data["reset"]), GOLD[f"{tag}.reset"])
  for t in range(CASES[name][4]):
    obs, maxed = pipe.step(data["frames"][t], data["dones"][t], data["after_done"][t])
    nt.assert_array_equal(maxed, GOLD[f"{tag}.max"][t])
    nt.assert_array_equal(obs, GOLD[f"{tag}.obs"][t])


@pytest.mark.gpu
@pytest.mark.parametrize("name,concat", VARIANTS)
def test_device_pipeline_matches_reference(name, concat):
  from derl_amd.env import DeviceAtariFrames
  data = stream(name)
  tag = f"{name}.{'concat' if concat else 'stack'}"
  if data["reset"][0].size % 4:
    pytest.skip("frame bytes not a multiple of 4")
  pipe = DeviceAtariFrames(4, concat)
  nt.assert_array_equal(pipe.reset(data["reset"]).cpu().numpy(), GOLD[f"{tag}.reset"])
  for t in range(CASES[name][4]):
    obs = pipe.step(data["frames"][t], data["dones"][t], data["after_done"][t])
    nt.assert_array_equal(pipe.maxed.cpu().numpy(), GOLD[f"{tag}.max"][t])
    nt.assert_array_equal(obs.cpu().numpy(), GOLD[f"{tag}.obs"][t])


@pytest.mark.gpu
def test_device_pipeline_writes_rollout_slots_at_atari_size():
  """The Nature-DQN shape: 210x160x3 frames -> gray 84x84 -> 4-frame stack written into
  consecutive slots of a (T+1, N, 84, 84, 4) rollout buffer; every stage against the oracle."""
  import torch
  from derl_amd.env import DeviceAtariFrames
  rs = np.random.RandomState(3)
  N, T = 5, 6
  raw = rs.randint(0, 256, size=(T + 1, N, 210, 160, 3)).astype(np.uint8)
  resets = rs.randint(0, 256, size=(T, N, 210, 160, 3)).astype(np.uint8)
  dones = rs.uniform(size=(T, N)) < 0.3
  buf = torch.empty((T + 1, N, 84, 84, 4), dtype=torch.uint8, device="cuda:0")
  pipe = DeviceAtariFrames(4, preprocess=(84, 84, True))
  ref = FramePipeline(4)
  nt.assert_array_equal(pipe.reset(raw[0], out=buf[0]).cpu().numpy(), ref.reset(gray_resize(raw[0])))
  last = raw[0].copy()
  for t in range(T):
    obs = pipe.step(raw[t + 1], dones[t], resets[t], out=buf[t + 1])
    assert obs.data_ptr() == buf[t + 1].data_ptr()
    maxed = np.maximum(raw[t + 1], last)
    last = np.where(dones[t][:, None, None, None], resets[t], raw[t + 1])
    # the oracle's queue works on preprocessed frames: feed it the preprocessed max / reset frames
    ref.last = np.zeros_like(gray_resize(maxed))
    expect, _ = ref.step(gray_resize(maxed), dones[t], gray_resize(resets[t]))
    nt.assert_array_equal(pipe.maxed.cpu().numpy(), maxed)
    nt.assert_array_equal(obs.cpu().numpy(), expect)
  with pytest.raises(ValueError):
    pipe.step(raw[0], out=pipe.observations)


def test_gray_resize_oracle_basics():
  """Identity at equal size, exact luma of a constant image, range preserved (the oracle of the
  unpinned step is at least self-consistent)."""
  rs = np.random.RandomState(0)
  img = rs.randint(0, 256, size=(2, 84, 84, 1)).astype(np.uint8)
  nt.assert_array_equal(gray_resize(img), img[..., 0])
  flat = np.full((1, 210, 160, 3), (200, 100, 50), np.uint8)
  expect = (200 * 4899 + 100 * 9617 + 50 * 1868 + 8192) >> 14
  assert np.all(gray_resize(flat) == expect)
