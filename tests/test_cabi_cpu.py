"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol that
include/derl_amd.h declares, the ctypes table matches the header, and host-side argument
validation fails with an error string (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
  text = open(os.path.join(ROOT, "include", "derl_amd.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(dx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
  import __graft_entry__
  from derl_amd import build
  if not os.path.exists(build.LIB):
    __graft_entry__.build()
  from derl_amd import _lib
  return _lib


def test_library_exports_every_declared_symbol(lib):
  handle = lib.load()
  names = header_functions()
  assert len(names) >= 4
  for name in names:
    assert hasattr(handle, name), f"{name} declared in include/derl_amd.h but not exported"
  assert sorted(lib.SIGNATURES) == names, "ctypes table and header disagree"
  assert handle.dx_abi_version() == lib.ABI_VERSION


def test_host_side_validation_reports_errors(lib):
  handle = lib.load()
  # negative shape: rejected on the host before any HIP call
  status = handle.dx_gae_f32(None, None, None, None, -1, 4, 0.99, 0.95, None, None, None)
  assert status == -1
  assert "negative shape" in lib.last_error()
  # null pointers with a non-empty shape
  status = handle.dx_gae_f32(None, None, None, None, 2, 4, 0.99, 0.95, None, None, None)
  assert status == -1 and "null pointer" in lib.last_error()
  with pytest.raises(lib.NativeError):
    lib.call("dx_gae_f32", None, None, None, None, 2, 4, 0.99, 0.95, None, None, None)
  # empty problem is a no-op
  assert handle.dx_gae_f32(None, None, None, None, 0, 4, 0.99, 0.95, None, None, None) == 0


def test_ops_refuse_cpu_tensors(lib):
  import torch
  from derl_amd import ops
  z = torch.zeros(2, 3)
  with pytest.raises(ValueError, match="no CPU path"):
    ops.gae(z, torch.zeros(2, 3, dtype=torch.bool), z, torch.zeros(3), 0.99, 0.95)


def test_missing_library_fails_loudly(lib, monkeypatch, tmp_path):
  """No fallback: without libderl_amd.so every native call raises NativeError naming the build."""
  monkeypatch.setattr(lib, "_lib", None)
  monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "libderl_amd.so"))
  with pytest.raises(lib.NativeError, match="no CPU fallback"):
    lib.load()
  with pytest.raises(lib.NativeError):
    lib.call("dx_abi_version")


def test_diagnostics_live_in_their_own_header_and_library(lib):
  """dx_diag_* microbenchmarks are not part of the product boundary: declared in
  include/derl_amd_diag.h, exported by libderl_amd_diag.so only."""
  import ctypes
  from derl_amd import build
  assert not [name for name in header_functions() if name.startswith("dx_diag")]
  text = open(os.path.join(ROOT, "include", "derl_amd_diag.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  names = sorted(set(re.findall(r"\b(dx_diag_[a-z0-9_]+)\s*\(", text)))
  assert len(names) >= 4
  product = lib.load()
  for name in names:
    assert not hasattr(product, name), f"{name} leaked into libderl_amd.so"
  path = build.build_diag_library()
  diag = ctypes.CDLL(path)
  for name in names:
    assert hasattr(diag, name), f"{name} declared in derl_amd_diag.h but not exported"


WRONG_RESULT_SWITCHES = ("DX_NTP_DIAG", "DX_NT_DIAG", "DX_WD_DIAG", "DX_FC_DIAG", "DX_NTP_NWG", "DX_WD_NWG")


def test_bisecting_switches_are_compiled_out_of_the_product_library(lib):
  """The in-kernel stamps and bisecting switches of the ring / weight-gradient kernels (some
  compute wrong results on purpose) exist only in the -DDX_DIAG=1 flavour, libderl_amd_diag.so:
  the product library never reads those variables -- their names are not even in its image."""
  from derl_amd import build
  if not os.path.exists(build.HIPCC):
    pytest.skip("no hipcc on this box: the diag flavour cannot be built")
  image = open(lib.LIB_PATH, "rb").read()
  for name in WRONG_RESULT_SWITCHES:
    assert name.encode() not in image, f"{name} is read by the product library"
  diag = open(build.build_diag_library(), "rb").read()
  for name in WRONG_RESULT_SWITCHES:
    assert name.encode() in diag, f"{name} is gone from the diagnostic flavour too"


def test_host_layer_under_address_sanitizer():
  """The sanitizer build of the shim: every product source compiled
  host-only with -fsanitize=address, driven through the validation / planning layer of every
  entry point by tools/asan_host_check.py with clang's ASan runtime preloaded."""
  import subprocess
  import sys
  from derl_amd import build
  runtime = build.asan_runtime()
  if runtime is None or not os.path.exists(build.HIPCC):
    pytest.skip("no hipcc / ASan runtime on this box")
  build.build_host_asan()
  env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_host_check.py")], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0 and "asan host check OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
  assert "AddressSanitizer" not in out.stderr
