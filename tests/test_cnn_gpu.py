"""GPU parity of the Nature-DQN forward / backward (dx_cnn_*, through the C-ABI) and of the
fused categorical loss against the reference's golden vectors and the CPU oracle.

Tolerances (fp32 MFMA is a k-ordered fma chain, the CPU reference sums in another order):
forward outputs rtol 1e-4 / atol 2e-5 (golden fixtures themselves carry 1e-5..1e-6 from
the reference's tests, alg/ppo_test.py:22-28), gradients rtol 1e-4 / atol 1e-5."""
import hashlib
import os

import numpy as np
import numpy.testing as nt
import pytest
import torch

import inputs as gi
import oracle
from oracle import models as om

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def make_engine(num_actions, weights, max_batch=64):
  from derl_amd.cnn_engine import CnnEngine
  eng = CnnEngine(num_actions, max_batch=max_batch, device=DEV)
  eng.load_state_dict(weights)
  return eng


def layer_by_layer_conv_stack(eng, obs, batch, idx=None):
  """conv0 / conv1 / conv2 forward as three separate stage launches (dx_cnn_stage: the fp32-MFMA / conv0_b16
  kernels, whatever route dx_cnn_forward takes): copies of y0, y1, y2."""
  import ctypes
  from derl_amd import _lib
  eng.pack()
  for stage in range(3):
    _lib.call("dx_cnn_stage", ctypes.byref(eng.ctx), stage, _lib.ptr(obs), 1, _lib.ptr(idx), batch,
              _lib.stream_ptr(eng.device))
  torch.cuda.synchronize()
  return (eng.y0[:batch * 12800].clone(), eng.y1[:batch * 5184].clone(), eng.y2[:batch * 3136].clone())


@pytest.mark.parametrize("num_actions,seed", [(4, 21), (6, 22)])
def test_forward_matches_reference_golden(num_actions, seed):
  weights = gi.nature_cnn_weights(num_actions, seed)
  obs = gi.frames(32, seed + 100)
  eng = make_engine(num_actions, weights)
  head = eng.forward(torch.from_numpy(obs).to(DEV))
  torch.cuda.synchronize()
  head = head.cpu().numpy()
  tag = f"cnn_a{num_actions}"
  with np.load(os.path.join(G, "act.npz")) as g:
    nt.assert_allclose(head[:, :num_actions], g[f"{tag}.raw_logits"], rtol=1e-4, atol=2e-5)
    nt.assert_allclose(head[:, num_actions:num_actions + 1], g[f"{tag}.values"], rtol=1e-4, atol=2e-5)
    nt.assert_array_equal(head[:, num_actions + 1:], 0)
    nt.assert_allclose(eng.hid[:32 * 512].view(32, 512).cpu().numpy(), g[f"{tag}.hidden"],
                       rtol=1e-4, atol=2e-5)


def oracle_activations(weights, obs):
  import torch.nn.functional as F
  x = torch.from_numpy(obs).permute(0, 3, 1, 2)
  x = (x.float() / 255 if x.dtype == torch.uint8 else x).contiguous()
  acts = []
  for i, s in enumerate((4, 2, 1)):
    x = F.relu(F.conv2d(x, torch.from_numpy(weights[f"base.conv-{i}.weight"]),
                        torch.from_numpy(weights[f"base.conv-{i}.bias"]), stride=s))
    acts.append(x.permute(0, 2, 3, 1).contiguous().numpy())  # NHWC like the engine
  return acts


@pytest.mark.parametrize("batch", [1, 3, 33, 129, 256, 320, 512])  # 256..512: the 64x64 ring shape
def test_forward_layers_and_ragged_batches(batch):
  weights = gi.nature_cnn_weights(4, 7)
  obs = gi.frames(batch, 5 + batch)
  eng = make_engine(4, weights, max_batch=max(batch, 256))
  head = eng.forward(torch.from_numpy(obs).to(DEV))
  torch.cuda.synchronize()
  ref = oracle_activations(weights, obs)
  for name, r in zip(("y0", "y1", "y2"), ref):
    got = getattr(eng, name)[:r.size].cpu().numpy().reshape(r.shape)
    nt.assert_allclose(got, r, rtol=1e-4, atol=2e-5, err_msg=name)
  logits, values = oracle.nature_cnn_forward(weights, obs)
  nt.assert_allclose(head[:, :4].cpu().numpy(), logits.numpy(), rtol=1e-4, atol=2e-5)
  nt.assert_allclose(head[:, 4:5].cpu().numpy(), values.numpy(), rtol=1e-4, atol=2e-5)


def test_forward_float_observations_upstream_fixture():
  """models_test.py:41-45: NatureCNNBase on torch.rand(32,84,84,4) float input."""
  torch.manual_seed(0)
  convs = [torch.nn.Conv2d(4, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)]
  linear = torch.nn.Linear(3136, 512)
  weights = {}
  for i, c in enumerate(convs):
    weights[f"base.conv-{i}.weight"] = c.weight.detach().numpy()
    weights[f"base.conv-{i}.bias"] = c.bias.detach().numpy()
  weights["base.linear.weight"] = linear.weight.detach().numpy()
  weights["base.linear.bias"] = linear.bias.detach().numpy()
  weights["output_layers.0.weight"] = np.zeros((4, 512), np.float32)
  weights["output_layers.0.bias"] = np.zeros(4, np.float32)
  weights["output_layers.1.weight"] = np.zeros((1, 512), np.float32)
  weights["output_layers.1.bias"] = np.zeros(1, np.float32)
  inputs = torch.rand(32, 84, 84, 4)
  eng = make_engine(4, weights)
  eng.forward(inputs.to(DEV))
  torch.cuda.synchronize()
  hidden = eng.hid[:32 * 512].view(32, 512).cpu().numpy()
  nt.assert_allclose(hidden, om.nature_cnn_hidden(weights, inputs).numpy(), rtol=1e-4, atol=1e-5)
  expected = np.load(os.path.join(G, "upstream", "dqn-base-outputs.npy"))
  if np.allclose(om.nature_cnn_hidden(weights, inputs).numpy(), expected, atol=1e-5):
    nt.assert_allclose(hidden, expected, atol=1e-5)


def test_forward_with_sample_index_gather():
  weights = gi.nature_cnn_weights(4, 9)
  obs = gi.frames(40, 77)
  idx = np.random.RandomState(0).permutation(40)[:24].astype(np.int32)
  eng = make_engine(4, weights)
  head = eng.forward(torch.from_numpy(obs).to(DEV), torch.from_numpy(idx).to(DEV)).clone()
  direct = eng.forward(torch.from_numpy(obs[idx]).to(DEV))
  torch.cuda.synchronize()
  nt.assert_array_equal(head.cpu().numpy(), direct.cpu().numpy())


def run_loss_and_backward(eng, data, mode, cliprange, vcoef, ecoef, num_actions, sample_idx=None, route="layers"):
  """``route`` "layers": dx_cnn_forward + dx_categorical_loss_f32 + dx_cnn_backward (every layer its own GEMM
  stages); "update": what a training update launches -- dx_cnn_forward_trunk + dx_cnn_heads_loss_f32 +
  dx_cnn_backward_part(3), i.e. with <= 7 actions the FACTORED tail (csrc/tail.hip: linear layer + heads as
  one affine map of y2)."""
  from derl_amd import ops
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
  obs = t(data["observations"])
  if route == "update":
    B = sample_idx.numel() if sample_idx is not None else obs.shape[0]
    eng.reserve(B)
    eng._ensure_backward()
    eng.forward_trunk(obs, sample_idx)
    partials = torch.empty(8 * ((B + 7) // 8), dtype=torch.float64, device=DEV)
    loss = torch.empty(8, dtype=torch.float32, device=DEV)
    eng.heads_loss(B, t(data["actions"]), t(data["log_prob"]) if mode == 0 else None, t(data["advantages"]),
                   t(data["values"].reshape(-1)) if mode == 0 else None, t(data["value_targets"].reshape(-1)), mode,
                   cliprange, vcoef, ecoef, B, partials, loss)
    eng.backward(obs, sample_idx, part=3)
    torch.cuda.synchronize()
    return loss.cpu().numpy(), eng.named_views(eng.grads)
  head = eng.forward(obs, sample_idx)
  eng._ensure_backward()
  B = head.shape[0]
  dhead = eng.dhead[:B * 32].view(B, 32)
  loss = ops.categorical_loss(
      head, t(data["actions"]), t(data["log_prob"]) if mode == 0 else None,
      t(data["advantages"]), t(data["values"].reshape(-1)) if mode == 0 else None,
      t(data["value_targets"].reshape(-1)), num_actions, mode, cliprange, vcoef, ecoef, dhead)
  eng.backward(obs, sample_idx)
  torch.cuda.synchronize()
  return loss.cpu().numpy(), eng.named_views(eng.grads)


@pytest.mark.parametrize("route", ["layers", "update"])
@pytest.mark.parametrize("name", ["ppo_step_cnn", "a2c_step_cnn", "a2c_step_cnn_late"])
def test_loss_and_gradients_match_reference_golden(name, route):
  from tests.test_oracle_golden import oracle_step_case, _check_summary
  cfg, g, params, names, data = oracle_step_case(name)
  eng = make_engine(cfg["num_actions"], params)
  mode = 0 if cfg["alg"] == "ppo" else 1
  loss, grads = run_loss_and_backward(eng, data, mode, cfg.get("cliprange"), cfg["value_loss_coef"],
                                      cfg["entropy_coef"], cfg["num_actions"], route=route)
  nt.assert_allclose(loss[0], g["loss0"], rtol=1e-5, atol=1e-5)  # alg/ppo_test.py:28 tolerance
  for k in names:
    _check_summary(grads[k].cpu().numpy(), g, f"grad0.{k}", rtol=1e-4, atol=1e-5)
  # every gradient tensor in full against the oracle's autograd
  if mode == 0:
    terms, ograds = oracle.ppo_loss_and_grads(params, data, "cnn", cfg["cliprange"],
                                              cfg["value_loss_coef"], cfg["entropy_coef"])
  else:
    terms, ograds = oracle.a2c_loss_and_grads(params, data, "cnn", cfg["value_loss_coef"],
                                              cfg["entropy_coef"])
  nt.assert_allclose(loss[1], terms["policy_loss"], rtol=1e-4, atol=1e-5)
  nt.assert_allclose(loss[2], terms["entropy"], rtol=1e-5, atol=1e-6)
  nt.assert_allclose(loss[3], terms["value_loss"], rtol=1e-4, atol=1e-5)
  for k in names:
    scale = np.abs(ograds[k]).max()
    nt.assert_allclose(grads[k].cpu().numpy(), ograds[k], rtol=1e-4, atol=1e-5 + 1e-5 * scale,
                       err_msg=k)


# 1024 / 2100: the training-size code paths (128x64 forward tiles from 65536 rows, wgrad with the
# XCD-aware block numbering from 64 reduction slices, pixel-group dgrads with a ragged last group)
# 2048: the persistent ring kernels with few tiles per workgroup + nt_dma for the linear layer
# 1152 = 9 groups of 128 images: the ring kernels with one XCD holding two image groups, the others one
@pytest.mark.parametrize("route", ["layers", "update"])
@pytest.mark.parametrize("batch", [1, 5, 37, 130, 1024, 1152, 2048, 2100, 2560, 8192])  # 8192 = BASELINE minibatch; 2560 = config 5's shard (K-split linear forward, uneven row groups per XCD)
def test_backward_ragged_batches_with_gather(batch, route):
  check_backward_against_float64(batch, route, 6)


# EVERY output width 2 .. 19 the factored tail instantiates a kernel for (tail_bwd_kernel<2..8>, tail_bwd_wide_kernel<9..19>,
# tail_loss_bwd_kernel<2..8>, tail_grads_kernel<8|16|24>: each its own register allocation and row padding) at one small
# and one >= 1,024 ragged batch: real Atari ids land on 3 (Freeway, Skiing), 10, 14 and 16 actions too
@pytest.mark.parametrize("A,batch", [(A, 21) for A in range(1, 19)] + [(A, 1030) for A in range(1, 19)] + [(18, 2048)])
def test_wide_action_sets_stay_on_the_factored_tail(A, batch):
  """derl builds heads of any width (derl/models.py:186-203) for any Atari id (derl/env/make_env.py:94-106: the full
  action set has 18 actions).  Up to 18 actions the linear layer + heads stay ONE affine map of y2 (csrc/tail.hip: the
  A + 1 outputs padded to 16 or 24 rows; tail_bwd's half-row workgroups from 9 outputs on; Wc rows beyond 12 read from L2
  in the loss pass) and heads + loss + the heads' backward stay one launch: loss and every gradient against the float64
  oracle on the engine's ReLU branch, and the routes the library reports."""
  import ctypes
  from derl_amd import _lib
  eng = check_backward_against_float64(batch, "update", A)
  lib = _lib.load()
  assert lib.dx_cnn_tail_factored(ctypes.byref(eng.ctx)) == 1 and lib.dx_cnn_fused_heads(ctypes.byref(eng.ctx)) == 1
  assert eng.fused_heads()
  for stage in (3, 4, 5, 6, 7, 8):  # linear layer and heads: forward, weight and data gradients
    assert lib.dx_cnn_last_route(stage).decode() == "tail_factored", (stage, lib.dx_cnn_last_route(stage))
  obs = torch.from_numpy(gi.frames(64, 5)).to(DEV)
  actions = torch.empty(64, dtype=torch.int64, device=DEV)
  log_prob, values = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
  eng.act(obs, actions, log_prob, values)
  assert lib.dx_cnn_last_route(3).decode() == "convstack (tail_factored)"  # the rollout's tail inside the conv-stack launch
  wide = make_engine(19, gi.nature_cnn_weights(19, 3), max_batch=64)  # beyond 18 actions: layer by layer again
  assert lib.dx_cnn_tail_factored(ctypes.byref(wide.ctx)) == 0 and not wide.fused_heads()


_FLOAT64_ORACLE = {}  # (batch, actions, digest of the engine's ReLU masks) -> the float64 oracle's verdicts and gradients


def test_tail_scratch_is_sized_for_every_batch_up_to_max_batch():
  """The factored tail's workgroup count is not monotone in the batch (2,100 rows -> 234 workgroups of 9 rows, 2,048 rows
  -> 256 of 8), so an engine reserved for 2,100 samples must hold the LARGER plan of a 2,048-sample update: with 18 actions
  (24 padded rows) the old max_batch-only sizing ran 150 K floats past the linear layer's slab region."""
  check_backward_against_float64(2048, "update", 18, max_batch=2100)


def check_backward_against_float64(batch, route, A, max_batch=None):
  rs = np.random.RandomState(batch)
  weights = gi.nature_cnn_weights(A, 31)
  pool = gi.frames(batch + 7, 1000 + batch)
  idx = rs.permutation(batch + 7)[:batch].astype(np.int32)
  data = dict(observations=pool, actions=rs.randint(0, A, batch).astype(np.int64),
              log_prob=(rs.standard_normal(batch) * 0.1 - 1.7).astype(np.float32),
              advantages=rs.standard_normal(batch).astype(np.float32),
              values=rs.standard_normal((batch, 1)).astype(np.float32) * 0.2,
              value_targets=rs.standard_normal((batch, 1)).astype(np.float32))
  eng = make_engine(A, weights, max_batch=max_batch or max(256, batch))
  loss, grads = run_loss_and_backward(eng, data, 0, 0.1, 0.25, 0.01, A,
                                      torch.from_numpy(idx).to(DEV), route=route)
  odata = dict(data, observations=pool[idx])
  # float64 evaluation of the oracle ON THE ReLU BRANCH THE ENGINE TOOK.  A pre-activation that
  # lies within float32 rounding of zero may fall on either side of the ReLU depending on the
  # summation order (at batch 130 torch-CPU's own NCHW and channels-last float32 paths disagree
  # on one conv-1 unit, which moves conv-0/1 gradients by ~1e-3 of their scale).  So the masks
  # are read back from the engine's kept activations (y > 0), checked against the oracle's own
  # float64 masks -- they may differ ONLY on units whose float64 pre-activation is within 3e-6
  # of zero relative to the layer's scale -- and the gradient comparison is then tight at every
  # batch size, the BASELINE minibatch 8192 included.
  masks = engine_relu_masks(eng, batch)
  # (the float64 passes below are the expensive part of this file -- 30 s at batch 8192 -- and depend on the kernels only
  # through the masks: routes and switch settings that took the same ReLU branch share them)
  key = (batch, A, hashlib.sha1(b"".join(np.packbits(m).tobytes() for m in masks)).hexdigest())
  if key not in _FLOAT64_ORACLE:
    flipped, worst = mask_disagreement(weights, pool[idx], masks)
    ambiguous = count_ambiguous_relu_units(weights, pool[idx])
    terms, ograds = oracle.ppo_loss_and_grads(weights, odata, "cnn", 0.1, 0.25, 0.01,
                                              dtype=torch.float64, relu_masks=masks)
    _FLOAT64_ORACLE[key] = (flipped, worst, ambiguous, terms, ograds)
  flipped, worst, ambiguous, terms, ograds = _FLOAT64_ORACLE[key]
  assert worst < 3e-6, (flipped, worst)
  assert flipped <= ambiguous
  nt.assert_allclose(loss[0], terms["loss"], rtol=1e-4, atol=1e-5)
  for k, og in ograds.items():
    scale = np.abs(og).max()
    nt.assert_allclose(grads[k].cpu().numpy(), og, rtol=1e-4, atol=1e-5 + 1e-5 * scale, err_msg=k)
  return eng


@pytest.mark.parametrize("route", ["layers", "update"])
@pytest.mark.parametrize("batch", [37, 300])
def test_backward_with_float_observations(batch, route):
  """float32 observations (derl/models.py:117-124 divides only uint8 by 255) take the layer-by-layer fp32 forward
  -- the one-launch conv stack reads uint8 frames -- and then the SAME image-resident bf16 gradient kernels as a
  uint8 minibatch: loss and every gradient against the float64 oracle on the engine's ReLU branch."""
  from derl_amd import _lib
  rs = np.random.RandomState(7 + batch)
  A = 4
  weights = gi.nature_cnn_weights(A, 33)
  pool = (gi.frames(batch + 3, 2000 + batch).astype(np.float32) / 255).astype(np.float32)
  idx = rs.permutation(batch + 3)[:batch].astype(np.int32)
  data = dict(observations=pool, actions=rs.randint(0, A, batch).astype(np.int64),
              log_prob=(rs.standard_normal(batch) * 0.1 - 1.4).astype(np.float32),
              advantages=rs.standard_normal(batch).astype(np.float32),
              values=rs.standard_normal((batch, 1)).astype(np.float32) * 0.2,
              value_targets=rs.standard_normal((batch, 1)).astype(np.float32))
  eng = make_engine(A, weights, max_batch=max(256, batch))
  loss, grads = run_loss_and_backward(eng, data, 0, 0.1, 0.25, 0.01, A, torch.from_numpy(idx).to(DEV), route=route)
  lib = _lib.load()
  assert lib.dx_cnn_last_route(1).decode() != "convstack_train"        # conv1 forward: an fp32 stage
  assert lib.dx_cnn_last_route(10).decode() == "dgrad_b6"
  masks = engine_relu_masks(eng, batch)
  flipped, worst = mask_disagreement(weights, pool[idx], masks)
  assert worst < 3e-6, (flipped, worst)
  terms, ograds = oracle.ppo_loss_and_grads(weights, dict(data, observations=pool[idx]), "cnn", 0.1, 0.25, 0.01,
                                            dtype=torch.float64, relu_masks=masks)
  nt.assert_allclose(loss[0], terms["loss"], rtol=1e-4, atol=1e-5)
  for k, og in ograds.items():
    scale = np.abs(og).max()
    nt.assert_allclose(grads[k].cpu().numpy(), og, rtol=1e-4, atol=1e-5 + 1e-5 * scale, err_msg=k)


@pytest.mark.parametrize("shape,batch", [((80, 96), 40), ((80, 96), 640), ((64, 64), 40), ((64, 64), 640)])
def test_other_frame_geometries_match_the_oracle(shape, batch):
  """The reference's NatureCNN takes any frame size (models.py:94-124 derives the linear layer's width
  from the conv stack).  80x96 frames -> 19x23, 8x10, 6x8 feature maps (an output width that is not
  a multiple of 4: the first layer's per-pixel offset table; dedicated weight-gradient kernels that
  do not cover the shape), 64x64 -> 15x15, 6x6, 4x4 (fewer than 256 first-layer pixels per image: no
  direct first-layer kernel at all).  Batch 40 takes the small-batch kernels, 640 = five groups of
  128 images the persistent ring kernels.  Forward layer by layer against torch-CPU float32,
  gradients against float64 autograd on the ReLU branch the engine took."""
  H, W = shape
  rs = np.random.RandomState(H * 1000 + batch)
  A = 5
  dims = [((H - 8) // 4 + 1, (W - 8) // 4 + 1)]
  dims.append(((dims[0][0] - 4) // 2 + 1, (dims[0][1] - 4) // 2 + 1))
  dims.append((dims[1][0] - 2, dims[1][1] - 2))
  flat = dims[2][0] * dims[2][1] * 64
  weights = {}
  for name, wshape in [("base.conv-0", (32, 4, 8, 8)), ("base.conv-1", (64, 32, 4, 4)), ("base.conv-2", (64, 64, 3, 3)),
                       ("base.linear", (512, flat)), ("output_layers.0", (A, 512)), ("output_layers.1", (1, 512))]:
    fan_in = int(np.prod(wshape[1:]))
    weights[f"{name}.weight"] = (rs.standard_normal(wshape) * 1.4 / np.sqrt(fan_in)).astype(np.float32)
    weights[f"{name}.bias"] = (rs.standard_normal(wshape[0]) * 0.05).astype(np.float32)
  base = rs.randint(0, 256, size=(batch + 3, H, W, 4)).astype(np.int32)
  pool = np.where(rs.uniform(size=(batch + 3, H, W, 1)) < 0.35, base, base // 8).astype(np.uint8)
  idx = rs.permutation(batch + 3)[:batch].astype(np.int32)
  from derl_amd.cnn_engine import CnnEngine
  eng = CnnEngine(A, input_shape=(H, W, 4), max_batch=max(256, batch), device=DEV)
  eng.load_state_dict(weights)
  data = dict(observations=pool, actions=rs.randint(0, A, batch).astype(np.int64),
              log_prob=(rs.standard_normal(batch) * 0.1 - 1.7).astype(np.float32),
              advantages=rs.standard_normal(batch).astype(np.float32),
              values=rs.standard_normal((batch, 1)).astype(np.float32) * 0.2,
              value_targets=rs.standard_normal((batch, 1)).astype(np.float32))
  loss, grads = run_loss_and_backward(eng, data, 0, 0.1, 0.25, 0.01, A, torch.from_numpy(idx).to(DEV))
  obs = pool[idx]
  masks = []
  for name, (h, w), c, r in zip(("y0", "y1", "y2"), dims, (32, 64, 64), oracle_activations(weights, obs)):
    y = getattr(eng, name)[:batch * h * w * c].view(batch, h, w, c)
    nt.assert_allclose(y.cpu().numpy(), r, rtol=1e-4, atol=2e-5, err_msg=name)
    masks.append((y > 0).permute(0, 3, 1, 2).contiguous().cpu().numpy())
  terms, ograds = oracle.ppo_loss_and_grads(weights, dict(data, observations=obs), "cnn", 0.1, 0.25, 0.01,
                                            dtype=torch.float64, relu_masks=masks)
  nt.assert_allclose(loss[0], terms["loss"], rtol=1e-4, atol=1e-5)
  for k, og in ograds.items():
    scale = np.abs(og).max()
    nt.assert_allclose(grads[k].cpu().numpy(), og, rtol=1e-4, atol=1e-5 + 1e-5 * scale, err_msg=k)


def engine_relu_masks(eng, batch):
  """The ReLU masks of the engine's last forward: kept post-activation outputs > 0 (NHWC in
  HBM), returned NCHW like the oracle's activations."""
  masks = []
  for name, (h, c) in (("y0", (20, 32)), ("y1", (9, 64)), ("y2", (7, 64))):
    y = getattr(eng, name)[:batch * h * h * c].view(batch, h, h, c)
    masks.append((y > 0).permute(0, 3, 1, 2).contiguous().cpu().numpy())
  return masks


def _float64_preactivations(weights, obs, masks=None):
  import torch.nn.functional as F
  x = torch.from_numpy(obs).permute(0, 3, 1, 2)
  x = (x.float() / 255 if x.dtype == torch.uint8 else x).double().contiguous()  # (float observations are taken as they are)
  for i, s in enumerate((4, 2, 1)):
    x = F.conv2d(x, torch.from_numpy(weights[f"base.conv-{i}.weight"]).double(),
                 torch.from_numpy(weights[f"base.conv-{i}.bias"]).double(), stride=s)
    yield x
    x = F.relu(x) if masks is None else x * torch.from_numpy(masks[i]).double()


def mask_disagreement(weights, obs, masks):
  """(number of units where the given masks differ from sign(float64 pre-activation), the
  largest |pre-activation| / layer scale among them), pre-activations evaluated on the GIVEN
  branch so that later layers see what the engine saw."""
  flipped, worst = 0, 0.0
  for x, m in zip(_float64_preactivations(weights, obs, masks), masks):
    differ = (x > 0) != torch.from_numpy(m)
    if differ.any():
      flipped += int(differ.sum())
      worst = max(worst, float(x[differ].abs().max() / x.abs().max()))
  return flipped, worst


def count_ambiguous_relu_units(weights, obs, rel=3e-6):
  """Number of conv pre-activations within `rel` of zero relative to their layer's scale
  (float64 evaluation): units whose ReLU mask depends on float32 summation order."""
  count = 0
  for x in _float64_preactivations(weights, obs):
    count += int((x.abs() < rel * x.abs().max()).sum())
  return count


@pytest.mark.parametrize("batch,A", [(1, 4), (7, 6), (128, 4), (256, 4), (300, 18), (1500, 6)]
                         + [pytest.param(40 + A, A, id=f"width{A}") for A in range(1, 20) if A not in (4, 6, 18)])  # every sample_step / output-group width; 19: the fallback
def test_fused_rollout_act_matches_unfused_path(batch, A):
  """dx_cnn_act (split-K linear layer + fused heads/sampling launch) against the plain
  forward + dx_categorical_act_f32 path and the oracle, with supplied uniforms."""
  from derl_amd import ops
  weights = gi.nature_cnn_weights(A, 40 + A)
  obs_np = gi.frames(batch, 9 + batch)
  obs = torch.from_numpy(obs_np).to(DEV)
  eng = make_engine(A, weights, max_batch=max(batch, 64))
  u = torch.rand(batch, device=DEV)
  head = eng.forward(obs)
  a_ref, lp_ref, v_ref = ops.categorical_act(head, A, u)
  actions = torch.empty(batch, dtype=torch.int64, device=DEV)
  log_prob = torch.empty(batch, device=DEV)
  values = torch.empty(batch, device=DEV)
  eng.act(obs, actions, log_prob, values, uniforms=u)
  torch.cuda.synchronize()
  # different summation order in the linear layer (split K): tiny logit differences may flip a
  # sample that sits on a CDF boundary
  assert (actions != a_ref).float().mean().item() <= 2e-3
  same = (actions == a_ref).cpu().numpy()
  nt.assert_allclose(values.cpu().numpy(), v_ref.cpu().numpy(), rtol=1e-4, atol=2e-5)
  nt.assert_allclose(log_prob.cpu().numpy()[same], lp_ref.cpu().numpy()[same], rtol=1e-4, atol=2e-5)
  logits, vals = oracle.nature_cnn_forward(weights, obs_np)
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, actions.cpu().numpy())
  nt.assert_allclose(log_prob.cpu().numpy(), lp.numpy(), rtol=1e-4, atol=2e-5)
  nt.assert_allclose(values.cpu().numpy(), vals.numpy()[:, 0], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("batch", [1, 5, 128, 300])
def test_image_resident_conv_stack_matches_layer_by_layer_kernels(batch):
  """The rollout's one-launch conv stack (csrc/convstack.hip: one workgroup per frame, y0 / y1 kept
  in LDS, conv0 on exact bf16 planes, conv1 / conv2 as 16x16x4 fp32 MFMA tiles + one pixel on the
  vector ALUs) against the layer-by-layer forward's y2 and the float64 oracle -- frames with
  extreme bytes (0 / 255 rows) included, and the route recorded."""
  from derl_amd import _lib
  weights = gi.nature_cnn_weights(4, 77)
  obs_np = gi.frames(batch, 31 + batch)
  obs_np[0, :3] = 255
  obs_np[-1, -5:] = 0
  obs = torch.from_numpy(obs_np).to(DEV)
  eng = make_engine(4, weights, max_batch=max(batch, 64))
  want = layer_by_layer_conv_stack(eng, obs, batch)[2]
  eng.y2.fill_(float("nan"))
  actions = torch.empty(batch, dtype=torch.int64, device=DEV)
  log_prob, values = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
  eng.act(obs, actions, log_prob, values, uniforms=torch.rand(batch, device=DEV))
  torch.cuda.synchronize()
  lib = _lib.load()
  assert [lib.dx_cnn_last_route(i).decode() for i in range(3)] == ["convstack"] * 3
  got = eng.y2[:batch * 3136]
  scale = float(want.abs().max())
  nt.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * max(scale, 1.0))
  # ReLU sides agree except where the pre-activation is within rounding of zero
  flipped = ((got > 0) != (want > 0)) & (torch.maximum(got, want) > 1e-4 * scale)
  assert not bool(flipped.any())
  _, vals = oracle.nature_cnn_forward(weights, obs_np)
  nt.assert_allclose(values.cpu().numpy(), vals.numpy()[:, 0], rtol=1e-4, atol=2e-5)
  # fp32 ACCURACY of the bf16-split layers, measured: the conv stack in float64 (the dequantisation
  # x / 255 in float32 as the reference does it) against the fp32-MFMA kernels' error and this
  # kernel's.  conv1 / conv2 multiply six of the nine exact bf16 x bf16 products of every fp32 x fp32:
  # its error must stay at the level of the fp32 chain's own rounding.
  import torch.nn.functional as F
  n = min(batch, 32)
  x = (torch.from_numpy(obs_np[:n]).permute(0, 3, 1, 2).float() / 255).double()
  for i, stride in enumerate((4, 2, 1)):
    x = F.relu(F.conv2d(x, torch.from_numpy(weights[f"base.conv-{i}.weight"]).double(),
                        torch.from_numpy(weights[f"base.conv-{i}.bias"]).double(), stride=stride))
  exact = x.permute(0, 2, 3, 1).reshape(n, 3136).numpy()  # NHWC like the device buffers
  err_stack = np.abs(got[:n * 3136].cpu().numpy().reshape(n, 3136) - exact).max()
  err_fp32 = np.abs(want[:n * 3136].cpu().numpy().reshape(n, 3136) - exact).max()
  assert err_stack <= max(2.0 * err_fp32, 2e-6 * scale), (err_stack, err_fp32, scale)


@pytest.mark.parametrize("batch", [3, 300, 1024])
def test_training_forward_in_one_launch_matches_the_layer_by_layer_stages(batch):
  """dx_cnn_forward_trunk on uint8 frames runs the conv stack of the whole minibatch as ONE launch of the
  image-resident kernel (convstack.hip, `train`: every workgroup walks its share of the images, the next
  frame arrives by LDS-DMA): y0, y1, y2 -- what the backward reads -- against the three layer-by-layer
  stages, with the minibatch's gather, more images than workgroups and a ragged share."""
  from derl_amd import _lib
  weights = gi.nature_cnn_weights(4, 78)
  pool = batch + 7
  obs_np = gi.frames(pool, 41 + batch)
  obs_np[1, :3] = 255
  obs_np[-2, -5:] = 0
  obs = torch.from_numpy(obs_np).to(DEV)
  idx = torch.randperm(pool, device=DEV)[:batch].to(torch.int32)
  eng = make_engine(4, weights, max_batch=batch)
  want = layer_by_layer_conv_stack(eng, obs, batch, idx)
  for buf in (eng.y0, eng.y1, eng.y2):
    buf.fill_(float("nan"))
  eng.forward_trunk(obs, idx)
  torch.cuda.synchronize()
  lib = _lib.load()
  assert [lib.dx_cnn_last_route(i).decode() for i in range(3)] == ["convstack_train"] * 3
  for got, ref, n in zip((eng.y0, eng.y1, eng.y2), want, (12800, 5184, 3136)):
    got = got[:batch * n]
    scale = float(ref.abs().max())
    nt.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    flipped = ((got > 0) != (ref > 0)) & (torch.maximum(got, ref) > 1e-4 * scale)
    assert not bool(flipped.any())


_WGRAD_PROBE = """
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests/golden")
import inputs as gi
from derl_amd.cnn_engine import CnnEngine
dev = torch.device("cuda")
torch.manual_seed(11)
B = 300
eng = CnnEngine(4, max_batch=B, device=dev)
eng.load_state_dict(gi.nature_cnn_weights(4, 3))
obs = torch.from_numpy(gi.frames(B, 9)).to(dev)
eng.forward(obs)
eng._ensure_backward()
eng.dhead[:B * 32].normal_()
grads = eng.backward(obs).double()
torch.cuda.synchronize()
# the same two weight gradients in float64 from the buffers the kernels read (NHWC)
y0 = eng.y0[:B * 12800].view(B, 20, 20, 32).permute(0, 3, 1, 2).double()
y1 = eng.y1[:B * 5184].view(B, 9, 9, 64).permute(0, 3, 1, 2).double()
dy1 = eng.dy1[:B * 5184].view(B, 9, 9, 64).permute(0, 3, 1, 2).double()
dy2 = eng.dy2[:B * 3136].view(B, 7, 7, 64).permute(0, 3, 1, 2).double()
gw1 = torch.nn.grad.conv2d_weight(y0, (64, 32, 4, 4), dy1, stride=2)
gw2 = torch.nn.grad.conv2d_weight(y1, (64, 64, 3, 3), dy2, stride=1)
ctx = eng.ctx
for name, ref, off in (("conv1", gw1, ctx.off_w[1]), ("conv2", gw2, ctx.off_w[2])):
  got = grads[off:off + ref.numel()].view_as(ref)
  print("ERR", name, float((got - ref).abs().max()), float(ref.abs().max()))
for name, ref, off in (("bias1", dy1.sum((0, 2, 3)), ctx.off_b[1]), ("bias2", dy2.sum((0, 2, 3)), ctx.off_b[2])):
  got = grads[off:off + 64]
  print("ERR", name, float((got - ref).abs().max()), float(ref.abs().max()))
# the two data gradients: float64 transposed convolutions of the gradient buffers, masked by the layer inputs
sd = gi.nature_cnn_weights(4, 3)
w1, w2 = (torch.from_numpy(sd["base.conv-%d.weight" % i]).double().to(dev) for i in (1, 2))
ref1 = torch.nn.grad.conv2d_input((B, 64, 9, 9), w2, dy2, stride=1) * (y1 > 0)
ref0 = torch.nn.grad.conv2d_input((B, 32, 20, 20), w1, dy1, stride=2) * (y0 > 0)
got0 = eng.dy0[:B * 12800].view(B, 20, 20, 32).permute(0, 3, 1, 2).double()
print("ERR", "dgrad2", float((dy1 - ref1).abs().max()), float(ref1.abs().max()))
print("ERR", "dgrad1", float((got0 - ref0).abs().max()), float(ref0.abs().max()))
"""


def test_bf16_split_gradients_are_as_accurate_as_the_fp32_kernels():
  """conv1 / conv2 weight and data gradients on the bf16 matrix cores (wgrad_b6.hip, dgrad_b6.hip: both
  operands split exactly into three bf16 terms, six of the nine products) against float64 sums over the very
  buffers the kernels read: the error must stay at the level of the fp32-MFMA kernels' own (DX_WGRAD_B6=0
  DX_DGRAD_B6=0; the switches are read once per process: one child process per setting)."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  errs = {}
  for setting in ("0", "1"):
    env = dict(os.environ)
    env["DX_WGRAD_B6"] = env["DX_DGRAD_B6"] = setting
    out = subprocess.run([sys.executable, "-c", _WGRAD_PROBE.format(root=root)], env=env, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for line in out.stdout.splitlines():
      if line.startswith("ERR"):
        _, name, err, scale = line.split()
        errs[(setting, name)] = (float(err), float(scale))
  for name in ("conv1", "conv2", "bias1", "bias2", "dgrad1", "dgrad2"):
    err_b6, scale = errs[("1", name)]
    err_fp32 = errs[("0", name)][0]
    assert err_b6 <= max(2.0 * err_fp32, 2e-6 * scale), (name, err_b6, err_fp32, scale)


@pytest.mark.parametrize("batch,gather", [(1, False), (5, True), (37, True), (300, False), (1030, True)])
def test_first_layer_weight_gradient_against_float64_sums_over_its_own_operands(batch, gather):
  """conv0_wgrad_ks.hip (the pixel contraction split over the waves; the frame converted to bf16 once per image, dY0 split
  exactly into three bf16 planes, transposed LDS reads) against float64 sums over the very buffers the kernel reads --
  the uint8 frames (through the minibatch's gather table) and the dY0 the backward left: exact products, so only the
  fp32 accumulation's own roundings remain (<= 2e-6 of the gradient's scale).  1, 5, 37, 300 frames: one workgroup per
  frame; 1,030: 256 workgroups of 4 - 5 frames, accumulators kept across them."""
  import ctypes
  from derl_amd import _lib
  eng = make_engine(4, gi.nature_cnn_weights(4, 3), max_batch=max(batch, 64))
  pool = torch.from_numpy(gi.frames(batch + 3, 40 + batch)).to(DEV)
  idx = torch.from_numpy(np.random.RandomState(batch).permutation(batch + 3)[:batch].astype(np.int32)).to(DEV) if gather else None
  obs = pool if gather else pool[:batch].contiguous()
  eng.forward(obs, idx)
  eng._ensure_backward()
  eng.dhead[:batch * 32].normal_(generator=torch.Generator(DEV).manual_seed(batch))
  grads = eng.backward(obs, idx).double()
  torch.cuda.synchronize()
  assert _lib.load().dx_cnn_last_route(13).decode() == "conv0_ks"
  frames = (pool[idx.long()] if gather else obs).permute(0, 3, 1, 2).double() / 255
  dy0 = eng.dy0[:batch * 12800].view(batch, 20, 20, 32).permute(0, 3, 1, 2).double()
  ref = torch.nn.grad.conv2d_weight(frames, (32, 4, 8, 8), dy0, stride=4)
  ctx = eng.ctx
  got = grads[ctx.off_w[0]:ctx.off_w[0] + ref.numel()].view_as(ref)
  scale = float(ref.abs().max())
  assert float((got - ref).abs().max()) <= 2e-6 * scale, (float((got - ref).abs().max()), scale)
  bias = dy0.sum((0, 2, 3))
  assert float((grads[ctx.off_b[0]:ctx.off_b[0] + 32] - bias).abs().max()) <= 2e-6 * float(bias.abs().max())


def test_exact_split_keeps_extreme_magnitudes_and_never_hides_a_non_finite_value():
  """The conv layers split every fp32 operand EXACTLY into three bf16 terms (csrc/bf16_split.hpp).  bf16 has fp32's
  exponent range, so (i) activations of extreme but finite magnitude (1e-30 .. 1e30 here, by scaling the first layer's
  weights) still come out at fp32 accuracy on both conv-stack kernels; (ii) the split of an infinite or NaN activation is
  NOT its fp32 value -- Inf gives (Inf, Inf - Inf = NaN, NaN), where an fp32 fma chain would keep Inf -- but it is never
  a FINITE number: an overflowed activation makes every output that depends on it non-finite, on the rollout kernel and
  on the training forward alike.  That is the contract: non-finite in, non-finite out (derl's own fp32 path turns
  Inf x 0 and Inf - Inf into NaN just the same)."""
  A, batch = 4, 48
  base = gi.nature_cnn_weights(A, 77)
  obs_np = gi.frames(batch, 78)
  obs = torch.from_numpy(obs_np).to(DEV)
  for scale in (1e30, 1e-30):
    weights = {k: v.copy() for k, v in base.items()}
    weights["base.conv-0.weight"] = (weights["base.conv-0.weight"] * np.float32(scale)).astype(np.float32)
    weights["base.conv-0.bias"] = (weights["base.conv-0.bias"] * np.float32(scale)).astype(np.float32)
    w64 = {k: v.astype(np.float64) for k, v in weights.items()}
    eng = make_engine(A, weights, max_batch=64)
    eng.forward_trunk(obs)   # the training forward (convstack_train.hip)
    y2_train = eng.y2[:batch * 3136].clone()
    actions = torch.empty(batch, dtype=torch.int64, device=DEV)
    log_prob, values = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
    eng.act(obs, actions, log_prob, values)   # the rollout kernel (convstack.hip)
    torch.cuda.synchronize()
    y2_roll = eng.y2[:batch * 3136].clone()
    x = torch.from_numpy(obs_np).permute(0, 3, 1, 2).float().div(255).double()
    for i, stride in enumerate((4, 2, 1)):
      x = torch.relu(torch.nn.functional.conv2d(x, torch.from_numpy(w64[f"base.conv-{i}.weight"]),
                                                 torch.from_numpy(w64[f"base.conv-{i}.bias"]), stride=stride))
    expect = x.permute(0, 2, 3, 1).reshape(-1).numpy()  # NHWC like ctx->y2
    top = np.abs(expect).max()
    assert np.isfinite(top) and top > 0
    for got in (y2_train, y2_roll):
      nt.assert_allclose(got.cpu().numpy().astype(np.float64) / top, expect / top, rtol=1e-4, atol=2e-6)
  weights = {k: v.copy() for k, v in base.items()}
  weights["base.conv-0.bias"][3] = np.inf  # one channel of y0 overflows at every pixel
  eng = make_engine(A, weights, max_batch=64)
  eng.forward_trunk(obs)
  torch.cuda.synchronize()
  assert not torch.isfinite(eng.y2[:batch * 3136]).any()  # every y2 value reads that channel through some tap
  actions = torch.empty(batch, dtype=torch.int64, device=DEV)
  log_prob, values = torch.empty(batch, device=DEV), torch.empty(batch, device=DEV)
  eng.act(obs, actions, log_prob, values)
  torch.cuda.synchronize()
  assert not torch.isfinite(eng.y2[:batch * 3136]).any() and not torch.isfinite(values).any()


def test_backward_in_two_parts_equals_whole_backward():
  """dx_cnn_backward_part(0) then (1) (the split a data-parallel caller overlaps its all-reduce
  with) gives bit-identical gradients, and part 0 alone already finalises the tail."""
  weights = gi.nature_cnn_weights(4, 3)
  obs = torch.from_numpy(gi.frames(96, 5)).to(DEV)
  eng = make_engine(4, weights, max_batch=96)
  eng.forward(obs)
  eng._ensure_backward()
  eng.dhead[:96 * 32].normal_()
  dhead = eng.dhead.clone()
  whole = eng.backward(obs).clone()
  eng.forward(obs)
  eng.dhead.copy_(dhead)
  eng.grads.fill_(float("nan"))
  eng.backward(obs, part=0)
  off = eng.tail_offset
  assert torch.equal(eng.grads[off:], whole[off:])
  assert torch.isnan(eng.grads[:off]).all()
  eng.backward(obs, part=1)
  assert torch.equal(eng.grads, whole)


_STREAMS_PROBE = """
import hashlib, sys, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests/golden")
import inputs as gi
from derl_amd.cnn_engine import CnnEngine
dev = torch.device("cuda")
torch.manual_seed(5)
eng = CnnEngine(4, max_batch=4608, device=dev)
eng.load_state_dict(gi.nature_cnn_weights(4, 3))
obs = torch.randint(0, 256, (4608, 84, 84, 4), dtype=torch.uint8, device=dev)
idx = torch.randperm(4608, device=dev).to(torch.int32)
digest = hashlib.sha256()
for rep in range(3):  # repeated: a race between the two streams would not hit the same way every time
  eng.forward(obs, idx)
  eng._ensure_backward()
  torch.manual_seed(7)
  eng.dhead[:4608 * 32].normal_()
  grads = eng.backward(obs, idx)
  torch.cuda.synchronize()
  digest.update(grads.cpu().numpy().tobytes())
  digest.update(eng.head[:4608 * 32].cpu().numpy().tobytes())
print("DIGEST", digest.hexdigest())
"""


def test_side_stream_routes_are_bit_identical_to_the_serial_ones():
  """The backward with its weight-gradient stages (and most of the slab reduction) on the side stream
  (DX_BWD_OVERLAP) and the forward as two half-batch chains (DX_FWD_LANES) launch the SAME kernels
  as the serial order on other streams: gradients and outputs must be bit-identical -- a difference
  would be a missing dependency between the streams (a race), not rounding.  The switches are read
  once per process: one child process per setting, 4,608 gathered samples (the whole batch and its
  halves of 2,304 rows take the same kernel routes), three repetitions each."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  digests = {}
  for setting in ("DX_BWD_OVERLAP=0 DX_FWD_LANES=0", "DX_BWD_OVERLAP=1 DX_FWD_LANES=0", "DX_BWD_OVERLAP=1 DX_FWD_LANES=1",
                  "DX_BWD_OVERLAP=1 DX_BWD_SIDE_FINALIZE=0"):
    env = dict(os.environ)
    env.update(item.split("=") for item in setting.split())
    out = subprocess.run([sys.executable, "-c", _STREAMS_PROBE.format(root=root)], env=env, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    digests[setting] = [line for line in out.stdout.splitlines() if line.startswith("DIGEST")][-1]
  assert len(set(digests.values())) == 1, digests


SWITCH_SETTINGS = ["DX_CONV0_F32=1", "DX_DGRAD_PIX=0", "DX_TN_SWIZZLE=0", "DX_LAT_MAX_TILES=0", "DX_WGRAD_DIRECT=0",
                   "DX_WGRAD_DIRECT_MIN_B=1", "DX_NT_DMA=0", "DX_NTP=0",
                   "DX_FC_FACTORED=0",                           # linear layer + heads layer by layer in updates and rollouts
                   "DX_TAIL_FUSED=0",                            # the factored tail's loss and its backward pass as two launches
                   "DX_FINALIZE_MERGED=0",                       # the tail's G reduction as its own launch, not beside the conv slabs'
                   "DX_CONVSTACK=0", "DX_CONVSTACK=0 DX_FC_FACTORED=0 DX_FC_ROLLOUT=0",  # the rollout's layer-by-layer kernels
                   "DX_CONVSTACK_TRAIN=0 DX_WGRAD_B6=0 DX_DGRAD_B6=0",  # the update's fp32-MFMA conv stages
                   "DX_BWD_OVERLAP=1 DX_FWD_LANES=1",            # side-stream routes at every batch size
                   "DX_BWD_OVERLAP=0 DX_C0_WAVES=4 DX_C0_GROUP4=0 DX_CONV0_KS=0",  # and their serial / round-2 twins
                   "DX_CONV0_KS=0",                              # the first layer's weight gradient on the 256-pixel-tile kernel
                   "DX_BWD_ORDER=0", "DX_BWD_ORDER=10",          # every backward stage ascending / the data gradients descending (default 21: the weight gradients)
                   "DX_WGRAD_B6_STREAM=0"]                       # conv2's weight gradient image by image (two K steps each) instead of one run of pixels


def test_diagnostic_switches_keep_parity():
  """Every alternative kernel route behind an environment switch (diagnostic switches) passes the same
  golden / oracle comparisons.  The library caches a switch at first use and dx_reload_env() drops the cache, so ONE
  child process walks all settings (tests/switch_walk.py: environment edited, cache dropped, a subset of this file run
  again); until round 5 each setting was its own process and the 14 of them took 230 s of the suite."""
  import subprocess
  import sys
  here = os.path.dirname(os.path.abspath(__file__))
  out = subprocess.run([sys.executable, os.path.join(here, "switch_walk.py")] + SWITCH_SETTINGS, capture_output=True, text=True,
                       timeout=1500, cwd=os.path.dirname(here))
  verdicts = [line for line in out.stdout.splitlines() if line.startswith("SWITCH ")]
  assert out.returncode == 0 and len(verdicts) == len(SWITCH_SETTINGS) and all(v.endswith("-> 0") for v in verdicts), \
      "\n".join(verdicts) + "\n" + out.stdout[-3000:] + out.stderr[-1500:]


def test_switch_cache_is_dropped_by_dx_reload_env():
  """The mechanism the walk rests on: a switch changed after its first use only takes effect behind dx_reload_env, and
  the route the library reports follows it (dx_cnn_last_route)."""
  from derl_amd import _lib
  lib = _lib.load()
  eng = make_engine(4, gi.nature_cnn_weights(4, 3), max_batch=64)
  obs = torch.randint(0, 256, (64, 84, 84, 4), dtype=torch.uint8, device=DEV)
  saved = os.environ.get("DX_CONVSTACK_TRAIN")
  try:
    os.environ.pop("DX_CONVSTACK_TRAIN", None)
    assert lib.dx_reload_env() == 0
    eng.forward_trunk(obs)
    assert lib.dx_cnn_last_route(1).decode() == "convstack_train"
    os.environ["DX_CONVSTACK_TRAIN"] = "0"
    eng.forward_trunk(obs)
    assert lib.dx_cnn_last_route(1).decode() == "convstack_train"  # cached: not read again
    assert lib.dx_reload_env() == 0
    eng.mark_dirty()
    eng.forward_trunk(obs)
    assert lib.dx_cnn_last_route(1).decode() != "convstack_train"
  finally:
    if saved is None:
      os.environ.pop("DX_CONVSTACK_TRAIN", None)
    else:
      os.environ["DX_CONVSTACK_TRAIN"] = saved
    lib.dx_reload_env()
    eng.mark_dirty()


def test_baseline_minibatches_take_the_fast_kernel_families():
  """Perf guard for the kernel routes (they depend on tile counts and on the batch being whole
  128-image groups): BASELINE's minibatches -- 8192 on one GPU, 1024 as the 8-GPU shard, 2048 in
  between -- must run their GEMM stages on the persistent ring (`ntp`) / the image-resident and
  linear-layer weight-gradient kernels, and a batch that is NOT a multiple of 128 images must be
  seen to leave them (the cliff is a property to know about, not a silent one):
  dx_cnn_last_route reports the family every stage took."""
  import ctypes
  from derl_amd import _lib
  stages = ["conv0_fwd", "conv1_fwd", "conv2_fwd", "fc_fwd", "heads_fwd", "heads_wgrad", "heads_dgrad",
            "fc_wgrad", "fc_dgrad", "conv2_wgrad", "conv2_dgrad", "conv1_wgrad", "conv1_dgrad", "conv0_wgrad",
            "finalize"]
  lib = _lib.load()

  def routes(batch):
    eng = make_engine(4, gi.nature_cnn_weights(4, 3), max_batch=batch)
    obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=DEV)
    eng._ensure_backward()
    eng.pack()
    eng.dhead[:batch * 32].zero_()
    for stage in range(len(stages)):
      _lib.call("dx_cnn_stage", ctypes.byref(eng.ctx), stage, _lib.ptr(obs), 1, None, batch,
                _lib.stream_ptr(eng.device))
    torch.cuda.synchronize()
    return {name: lib.dx_cnn_last_route(i).decode() for i, name in enumerate(stages)}

  ring = ("conv1_fwd", "conv2_fwd", "fc_dgrad")
  for batch in (8192, 2048, 1024):
    got = routes(batch)
    for name in ring:
      assert got[name] == "ntp", (batch, name, got)
    assert got["conv2_wgrad"] == got["conv1_wgrad"] == "wgrad_b6", (batch, got)
    assert got["conv2_dgrad"] == got["conv1_dgrad"] == "dgrad_b6", (batch, got)
    assert got["fc_wgrad"] == "wgrad_fc", (batch, got)
    assert got["conv0_fwd"] == "conv0_b16" and got["conv0_wgrad"] == "conv0_ks", (batch, got)
  assert routes(8192)["fc_fwd"] == "ntp"
  ragged = routes(8192 - 64)  # 63.5 groups of 128 images: the ring's dgrad tiles (one pixel x 128 images) do not exist
  assert ragged["fc_dgrad"] != "ntp", ragged
  assert ragged["conv2_dgrad"] == ragged["conv1_dgrad"] == "dgrad_b6", ragged  # (image-resident: any batch)
  # (the forward stages tile output PIXELS and still find whole 64-row tiles: they stay on the ring)


@pytest.mark.parametrize("batch,A,mode", [(5, 4, 0), (64, 6, 0), (1000, 7, 1), (2048, 4, 0)])
def test_fused_heads_and_loss_launch_matches_the_separate_launches(batch, A, mode):
  """dx_cnn_forward_trunk + dx_cnn_heads_loss_f32 + dx_cnn_backward_part(3) (heads forward, loss,
  loss reduction and the heads' dgrad / wgrad in ONE launch) against dx_cnn_forward +
  dx_categorical_loss_f32 + dx_cnn_backward on the same minibatch: head outputs, the eight loss
  scalars, dL/dhead, dL/dhid and EVERY gradient of the network (the heads' own gradients come from
  the fused launch's slabs, the rest flows through its dhid)."""
  from derl_amd import ops
  weights = gi.nature_cnn_weights(A, 60 + A)
  eng = make_engine(A, weights, max_batch=max(batch, 64))
  rs = np.random.RandomState(batch + A)
  obs = torch.from_numpy(gi.frames(batch, 17 + batch)).to(DEV)
  actions = torch.from_numpy(rs.randint(0, A, batch)).to(DEV)
  old_lp = torch.from_numpy((-np.log(A) + 0.3 * rs.standard_normal(batch)).astype(np.float32)).to(DEV)
  adv = torch.from_numpy(rs.standard_normal(batch).astype(np.float32)).to(DEV)
  old_v = torch.from_numpy((0.2 * rs.standard_normal(batch)).astype(np.float32)).to(DEV)
  vt = torch.from_numpy(rs.standard_normal(batch).astype(np.float32)).to(DEV)
  partials = torch.empty(8 * ((batch + 7) // 8), dtype=torch.float64, device=DEV)
  clip = 0.1 if mode == 0 else None
  # separate launches
  head_a = eng.forward(obs).clone()
  eng._ensure_backward()
  dhead = eng.dhead[:batch * 32].view(batch, 32)
  terms_a = ops.categorical_loss(head_a, actions, old_lp if mode == 0 else None, adv, old_v if mode == 0 else None,
                                 vt, A, mode, clip, 0.25, 0.01, dhead, batch, partials).clone()
  dhead_a = dhead.clone()
  eng.backward(obs)
  grads_a, dhid_a = eng.grads.clone(), eng.dhid[:batch * 512].clone()
  # one launch
  eng.grads.zero_()
  eng.forward_trunk(obs)
  terms_b = torch.empty(8, dtype=torch.float32, device=DEV)
  head_b = eng.heads_loss(batch, actions, old_lp if mode == 0 else None, adv, old_v if mode == 0 else None, vt,
                          mode, clip, 0.25, 0.01, batch, partials, terms_b).clone()
  dhead_b, dhid_b = eng.dhead[:batch * 32].view(batch, 32).clone(), eng.dhid[:batch * 512].clone()
  eng.backward(obs, part=3)
  grads_b = eng.grads.clone()
  nt.assert_allclose(head_b[:, :A + 1].cpu().numpy(), head_a[:, :A + 1].cpu().numpy(), rtol=1e-5, atol=2e-6)
  assert float(head_b[:, A + 1:].abs().max()) == 0.0
  nt.assert_allclose(terms_b.cpu().numpy(), terms_a.cpu().numpy(), rtol=2e-5, atol=1e-6)
  scale = float(dhead_a.abs().max())
  nt.assert_allclose(dhead_b.cpu().numpy(), dhead_a.cpu().numpy(), rtol=1e-4, atol=1e-6 * scale)
  nt.assert_allclose(dhid_b.cpu().numpy(), dhid_a.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(dhid_a.abs().max()))
  va, vb = eng.named_views(grads_a), eng.named_views(grads_b)
  for key in va:
    ref = va[key].cpu().numpy()
    nt.assert_allclose(vb[key].cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max() + 1e-9, err_msg=key)
  assert int(eng._loss_counter()[0]) == 0  # the last workgroup left the ticket word ready for the next launch
