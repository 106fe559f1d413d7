"""GPU parity of dx_gae_f32 (through the C-ABI) against the golden vectors of the
reference, the NumPy oracle and, at full sizes, the plain-C oracle + linearity."""
import ctypes
import os

import numpy as np
import numpy.testing as nt
import pytest
import torch

import inputs as gi
import oracle

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
# fp32 scan vs the reference's float64-intermediate recursion (|adv| <~ 10): stated tolerance
RTOL, ATOL = 1e-5, 1e-5


def run_gae(rewards, resets, values, last_values, gamma, lam):
  from derl_amd import ops
  dev = torch.device("cuda:0")
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  adv, vt = ops.gae(t(rewards.astype(np.float32)), t(resets), t(values), t(last_values),
                    gamma, lam)
  torch.cuda.synchronize()
  return adv.cpu().numpy(), vt.cpu().numpy()


@pytest.mark.parametrize("case", list(gi.GAE_CASES))
def test_gae_matches_reference_golden(case):
  d = gi.gae_inputs(case)
  unbatched = d["rewards"].ndim == 1
  r, z, v = d["rewards"], d["resets"], d["values"][..., 0]
  lv = d["last_values"].reshape(-1)
  if unbatched:
    r, z, v = r[:, None], z[:, None], v[:, None]
  adv, vt = run_gae(r, z, v, lv, d["gamma"], d["lambda_"])
  with np.load(os.path.join(G, "gae.npz")) as g:
    nt.assert_allclose(adv.reshape(-1), g[f"{case}.advantages"], rtol=RTOL, atol=ATOL)
    nt.assert_allclose(vt.reshape(-1), g[f"{case}.value_targets"][:, 0], rtol=RTOL, atol=ATOL)


def c_oracle():
  path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "liboracle_gae.so")
  if not os.path.exists(path):
    import __graft_entry__
    __graft_entry__.build_oracle()
  lib = ctypes.CDLL(path)
  lib.oracle_gae_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_double, ctypes.c_double,
                                                         ctypes.c_void_p, ctypes.c_void_p]
  return lib


@pytest.mark.parametrize("T,N", [(128, 1 << 16), (128, 1 << 18), (64, 2048 * 8), (5, 4096 * 64),
                                 (2048, 1), (3, 5), (129, 4099), (130, 130), (17, 131072 + 4)])
def test_gae_full_sizes_against_c_oracle(T, N):
  rs = np.random.RandomState(T * 7 + N)
  rewards = (np.sign(rs.standard_normal((T, N))) * (rs.uniform(size=(T, N)) < 0.3)).astype(np.float32)
  resets = rs.uniform(size=(T, N)) < 0.02
  values = rs.standard_normal((T, N)).astype(np.float32)
  last = rs.standard_normal(N).astype(np.float32)
  adv, vt = run_gae(rewards, resets, values, last, 0.99, 0.95)
  ref_adv, ref_vt = np.empty_like(values), np.empty_like(values)
  ru8 = resets.astype(np.uint8)
  c_oracle().oracle_gae_f32(rewards.ctypes.data, ru8.ctypes.data, values.ctypes.data,
                            last.ctypes.data, T, N, 0.99, 0.95, ref_adv.ctypes.data,
                            ref_vt.ctypes.data)
  nt.assert_allclose(adv, ref_adv, rtol=RTOL, atol=ATOL)
  nt.assert_allclose(vt, ref_vt, rtol=RTOL, atol=ATOL)


def test_gae_linearity_and_reset_isolation():
  """Size-independent properties: GAE is linear in (rewards, values, last_values) for
  fixed resets, and an env's outputs before a reset do not depend on what follows it."""
  T, N = 128, 1 << 15
  rs = np.random.RandomState(3)
  resets = rs.uniform(size=(T, N)) < 0.05
  mk = lambda: (rs.standard_normal((T, N)).astype(np.float32),
                rs.standard_normal((T, N)).astype(np.float32),
                rs.standard_normal(N).astype(np.float32))
  (r1, v1, l1), (r2, v2, l2) = mk(), mk()
  a1, _ = run_gae(r1, resets, v1, l1, 0.99, 0.95)
  a2, _ = run_gae(r2, resets, v2, l2, 0.99, 0.95)
  a12, _ = run_gae(r1 + r2, resets, v1 + v2, l1 + l2, 0.99, 0.95)
  nt.assert_allclose(a12, a1 + a2, rtol=1e-4, atol=1e-4)
  # perturb everything strictly after the first reset of each env: earlier outputs fixed
  first = np.where(resets.any(0), resets.argmax(0), T)
  later = np.arange(T)[:, None] > first[None, :]
  r3 = np.where(later, r1 + 5, r1).astype(np.float32)
  v3 = np.where(later, v1 - 3, v1).astype(np.float32)
  a3, _ = run_gae(r3, resets, v3, l1 + 1, 0.99, 0.95)
  keep = (np.arange(T)[:, None] <= first[None, :]) & resets.any(0)[None, :]
  nt.assert_array_equal(a3[keep], a1[keep])


def test_gae_argument_errors():
  from derl_amd import ops
  dev = torch.device("cuda:0")
  z = torch.zeros(4, 8, device=dev)
  with pytest.raises(ValueError):
    ops.gae(z, torch.zeros(4, 7, device=dev, dtype=torch.bool), z, torch.zeros(8, device=dev), .99, .95)
  with pytest.raises(ValueError):
    ops.gae(z.cpu(), torch.zeros(4, 8, dtype=torch.bool), z.cpu(), torch.zeros(8), .99, .95)
  with pytest.raises(ValueError):
    ops.gae(z, torch.zeros(4, 8, device=dev, dtype=torch.bool), z, torch.zeros(9, device=dev), .99, .95)
  # empty input is a no-op, not an error
  e = torch.zeros(0, 8, device=dev)
  adv, _ = ops.gae(e, torch.zeros(0, 8, device=dev, dtype=torch.bool), e, torch.zeros(8, device=dev), .99, .95)
  assert adv.shape == (0, 8)
