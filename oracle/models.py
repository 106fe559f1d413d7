"""Nature-DQN conv stack and MuJoCo MLP forward passes (CPU oracle, torch-CPU fp32).

Follows derl/models.py:94-124 (NatureCNNBase), :166-214 (NatureCNNModel),
:224-237 (MLP), :240-271 (MuJoCoModel).  Parameters are plain dicts keyed by the
reference's ``state_dict()`` names so that checkpoints interchange.
"""
import numpy as np
import torch
import torch.nn.functional as F

NATURE_CNN_KEYS = (
    "base.conv-0.weight", "base.conv-0.bias",
    "base.conv-1.weight", "base.conv-1.bias",
    "base.conv-2.weight", "base.conv-2.bias",
    "base.linear.weight", "base.linear.bias",
)


def _t(x):
  return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


def nature_cnn_forward(params, observations, relu_masks=None):
  """observations (B,84,84,4) uint8 or float -> list of head outputs.

  models.py:117-124: NHWC -> NCHW permute, uint8 -> float()/255, then
  conv(k8,s4)+ReLU, conv(k4,s2)+ReLU, conv(k3,s1)+ReLU, flatten (NCHW order),
  linear 3136->512 with NO ReLU after it (:112-115), then one linear per head
  (:198-202).  Returns [head_0, head_1, ...] with shapes (B, units).

  ``relu_masks`` (three bool arrays, NCHW, one per conv layer) replaces each ReLU by a
  multiplication with the given 0/1 mask: the same piecewise-linear function evaluated on a
  PRESCRIBED branch.  Large-batch parity tests pass the masks the device kernels used, so that
  a pre-activation within float32 rounding of zero (whose side depends on summation order)
  does not decide the comparison; the tests separately bound how many units may differ.
  """
  x = _t(observations)
  x = x.permute(0, 3, 1, 2)
  if x.dtype == torch.uint8:
    x = x.float() / 255
  # float64 parameters select the high-precision evaluation used as the reference for
  # large batches (the float32 dequantisation above is kept: it is part of the semantics)
  x = x.to(_t(params["base.conv-0.weight"]).dtype).contiguous()
  for i, stride in enumerate((4, 2, 1)):
    x = F.conv2d(x, _t(params[f"base.conv-{i}.weight"]),
                 _t(params[f"base.conv-{i}.bias"]), stride=stride)
    x = F.relu(x) if relu_masks is None else x * _t(relu_masks[i]).to(x.dtype)
  x = torch.flatten(x, 1)
  hidden = F.linear(x, _t(params["base.linear.weight"]), _t(params["base.linear.bias"]))
  outputs = []
  i = 0
  while f"output_layers.{i}.weight" in params:
    outputs.append(F.linear(hidden, _t(params[f"output_layers.{i}.weight"]),
                            _t(params[f"output_layers.{i}.bias"])))
    i += 1
  return outputs


def nature_cnn_hidden(params, observations):
  """The 512-d base output (models.py:94-124) -- what dqn-base-outputs.npy pins."""
  x = _t(observations).permute(0, 3, 1, 2)
  if x.dtype == torch.uint8:
    x = x.float() / 255
  x = x.contiguous()
  for i, stride in enumerate((4, 2, 1)):
    x = F.relu(F.conv2d(x, _t(params[f"base.conv-{i}.weight"]),
                        _t(params[f"base.conv-{i}.bias"]), stride=stride))
  return F.linear(torch.flatten(x, 1), _t(params["base.linear.weight"]),
                  _t(params["base.linear.bias"]))


def mlp_forward(params, prefix, x, nlayers=3):
  """Linear/Tanh stack with no activation after the last layer (models.py:224-237);
  ``prefix`` is e.g. ``module_list.0`` and layer i lives at index 2*i."""
  for i in range(nlayers):
    x = F.linear(x, _t(params[f"{prefix}.{2 * i}.weight"]),
                 _t(params[f"{prefix}.{2 * i}.bias"]))
    if i + 1 < nlayers:
      x = torch.tanh(x)
  return x


def mujoco_forward(params, observations):
  """(mean, std, values...) of MuJoCoModel (models.py:261-271): independent MLPs per
  output, std = exp(logstd) repeated over the batch; inputs cast to the model dtype."""
  x = _t(observations).to(_t(params["logstd"]).dtype)  # collocate_inputs(): model dtype
  outs = []
  i = 0
  while f"module_list.{i}.0.weight" in params:
    outs.append(mlp_forward(params, f"module_list.{i}", x))
    i += 1
  std = torch.exp(_t(params["logstd"]))[None].repeat_interleave(x.shape[0], 0)
  return (outs[0], std, *outs[1:])


def mujoco_keys(nmlps=2):
  keys = ["logstd"]
  for m in range(nmlps):
    for layer in (0, 2, 4):
      keys += [f"module_list.{m}.{layer}.weight", f"module_list.{m}.{layer}.bias"]
  return tuple(keys)


def _orthogonal_(tensor):
  torch.nn.init.orthogonal_(tensor)  # models.py:135-138 (gain 1), biases zero
  return tensor


def init_nature_cnn(output_units=(4, 1), input_channels=4, seed=None):
  """Orthogonal weights / zero biases as a state_dict-shaped dict of float32 tensors.

  RNG consumption follows the reference so that seeded models coincide
  (models.py:102-115,190-195): every layer is first built with torch's default
  init (which draws from the global generator), and only then does
  ``self.apply(orthogonal_init)`` redraw the weights in module order."""
  if seed is not None:
    torch.manual_seed(seed)
  layers = [("base.conv-0", torch.nn.Conv2d(input_channels, 32, 8, 4)),
            ("base.conv-1", torch.nn.Conv2d(32, 64, 4, 2)),
            ("base.conv-2", torch.nn.Conv2d(64, 64, 3, 1)),
            ("base.linear", torch.nn.Linear(3136, 512))]
  layers += [(f"output_layers.{i}", torch.nn.Linear(512, n))
             for i, n in enumerate(output_units)]
  params = {}
  for name, layer in layers:
    params[f"{name}.weight"] = _orthogonal_(layer.weight.detach())
    params[f"{name}.bias"] = torch.zeros_like(layer.bias.detach())
  return params


def init_mujoco(observation_dim, output_units=(6, 1), seed=None):
  """models.py:247-258: one 64-64 tanh MLP per output, all built (default init) first,
  then orthogonal init in module order, logstd = 0."""
  if seed is not None:
    torch.manual_seed(seed)
  layers = []
  for m, nout in enumerate(output_units):
    dims = (observation_dim, 64, 64, nout)
    for layer, (nin, no) in enumerate(zip(dims[:-1], dims[1:])):
      layers.append((f"module_list.{m}.{2 * layer}", torch.nn.Linear(nin, no)))
  params = {"logstd": torch.zeros(output_units[0])}
  for name, layer in layers:
    params[f"{name}.weight"] = _orthogonal_(layer.weight.detach())
    params[f"{name}.bias"] = torch.zeros_like(layer.bias.detach())
  return params
