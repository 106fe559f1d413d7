"""Known answers of the reference's own unit tests, replayed on the HIP path:
derl/policies_test.py:9-29 (TorchTestCase seeds torch with 0, the model is created first, then the
observation is drawn from the same stream) and the structural checks of derl/models_test.py:47-140.
Sampled actions are not a parity target (the device sampler is counter-based); the constants that
do not depend on the sampler's stream are."""
import numpy as np
import numpy.testing as nt
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def orthogonal_rows_or_cols(arr):
  arr = arr.reshape(arr.shape[0], -1)
  gram = arr.T @ arr if arr.shape[0] > arr.shape[1] else arr @ arr.T
  nt.assert_allclose(gram, np.eye(gram.shape[0]), atol=1e-5)


def test_policies_test_normal_constants():
  import derl_amd as derl
  torch.manual_seed(0)
  model = derl.MuJoCoModel(3, (2, 1))
  obs = torch.randn(3)  # policies_test.py:24
  policy = derl.ActorCriticPolicy(model)
  out = policy.act(dict(observations=obs[None].numpy()), training=True)
  nt.assert_allclose(out["values"].cpu().numpy().reshape(-1), [-0.18482158], rtol=2e-6)
  actions = torch.tensor([[-1.7938228, 1.0464325]], device=DEV)  # what torch's sampler drew upstream
  nt.assert_allclose(out["distribution"].log_prob(actions).cpu().numpy(), [-3.7467263], rtol=2e-6)
  act = policy.act(obs.numpy())  # unbatched input, NumPy out
  assert list(act.keys()) == ["actions", "log_prob", "values"]
  assert act["actions"].shape == (2,) and act["values"].shape == (1,)
  nt.assert_allclose(act["values"], [-0.18482158], rtol=2e-6)


def test_policies_test_categorical_constants():
  import derl_amd as derl
  torch.manual_seed(0)
  model = derl.NatureCNNModel((6, 1))
  obs = torch.rand(84, 84, 4)  # float observations in [0, 1), policies_test.py:14
  policy = derl.ActorCriticPolicy(model)
  out = policy.act(dict(observations=obs[None].numpy()), training=True)
  nt.assert_allclose(out["values"].cpu().numpy().reshape(-1), [0.257305294], rtol=2e-6)
  lp = out["distribution"].log_prob(torch.tensor([3], device=DEV))
  nt.assert_allclose(lp.cpu().numpy(), [-1.80754196], rtol=2e-6)
  act = policy.act(obs.numpy())
  assert list(act.keys()) == ["actions", "log_prob", "values"] and act["actions"].shape == ()
  nt.assert_allclose(act["values"], [0.257305294], rtol=2e-6)


def test_models_test_structure_nature_cnn():
  import derl_amd as derl
  torch.manual_seed(0)
  model = derl.NatureCNNModel(output_units=(4, 1))
  weights = [p for n, p in model.named_parameters() if n.endswith("weight")]
  biases = [p for n, p in model.named_parameters() if n.endswith("bias")]
  assert len(weights) == 6 and len(biases) == 6  # models_test.py:63-74
  for w in weights:
    orthogonal_rows_or_cols(w.detach().cpu().numpy().astype(np.float64))
  for b in biases:
    assert float(b.detach().abs().max()) == 0.0
  outs = model(torch.rand(84, 84, 4))  # broadcast, models_test.py:76-81
  assert len(outs) == 2 and tuple(outs[0].shape) == (4,) and tuple(outs[1].shape) == (1,)


def test_models_test_structure_mujoco():
  import derl_amd as derl
  torch.manual_seed(0)
  model = derl.MuJoCoModel(4, (5, 1))
  names = [n for n, _ in model.named_parameters()]
  assert len(names) == 12 + 1 and "logstd" in names  # two 3-layer nets + logstd
  for n, p in model.named_parameters():
    if n.endswith("weight"):
      orthogonal_rows_or_cols(p.detach().cpu().numpy().astype(np.float64))
    elif n.endswith("bias"):
      assert float(p.detach().abs().max()) == 0.0
  mean, std, values = model(torch.rand(2, 4))  # models_test.py:114-124
  assert tuple(mean.shape) == (2, 5) and tuple(std.shape) == (2, 5) and tuple(values.shape) == (2, 1)
  nt.assert_array_equal(std.detach().cpu().numpy(), 1.0)
  outs = model(torch.rand(4).double())  # broadcast + dtype, models_test.py:126-140
  assert tuple(outs[0].shape) == (5,) and tuple(outs[1].shape) == (5,) and tuple(outs[2].shape) == (1,)
