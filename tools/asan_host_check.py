"""Runs the HOST side of the C-ABI (argument validation, planning, context set-up) under
AddressSanitizer: loads derl_amd/libderl_amd_hostasan.so (derl_amd.build.build_host_asan: every
product source compiled --cuda-host-only with -fsanitize=address) and drives every entry point
into its validation layer -- null pointers, negative / zero / misaligned / oversized shapes, wrong
struct sizes -- plus the pure-host planners (dx_cnn_init / dx_mlp_init for a sweep of shapes).
No kernel is launched (there is no GPU in the build container).  Exits non-zero on an ASan
report or an unexpected status.

  LD_PRELOAD=$(python -c 'from derl_amd import build; print(build.asan_runtime())') \\
  ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_check.py
(tests/test_cabi_cpu.py::test_host_layer_under_address_sanitizer does exactly that)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd import _lib, build  # noqa: E402


def main():
  import torch  # noqa: F401  (libamdhip64 before the library)
  lib = ctypes.CDLL(build.ASAN_LIB, mode=ctypes.RTLD_LOCAL)
  for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = _lib._RESTYPES.get(name, ctypes.c_int)
  assert lib.dx_abi_version() == _lib.ABI_VERSION
  calls = 0

  def expect_error(name, *args):
    nonlocal calls
    status = getattr(lib, name)(*args)
    calls += 1
    assert status < 0, f"{name}{args} accepted bad arguments (status {status})"
    assert lib.dx_last_error(), name

  # every entry point with all-null / zero arguments of the right arity
  skip = {"dx_abi_version", "dx_last_error", "dx_device_info", "dx_launch_count", "dx_cnn_last_route"}
  for name, argtypes in _lib.SIGNATURES.items():
    if name in skip:
      continue
    zeros = []
    for t in argtypes:
      zeros.append(None if t in (_lib.P, ctypes.c_char_p) or isinstance(t, type(ctypes.POINTER(ctypes.c_int)))
                   else t(0).value)
    status = getattr(lib, name)(*zeros)
    calls += 1
    assert status <= 0, (name, status)  # an empty problem may be a no-op; anything else is refused
  # shapes that must be refused before any launch
  expect_error("dx_gae_f32", None, None, None, None, -1, 4, 0.99, 0.95, None, None, None)
  expect_error("dx_gae_f32", None, None, None, None, 2, 4, 0.99, 0.95, None, None, None)
  fake = ctypes.c_void_p(0x1000)  # never dereferenced on the host
  expect_error("dx_frame_max_u8", fake, fake, None, None, fake, 3, 7, None)          # bytes per env % 4
  expect_error("dx_frame_max_u8", fake, fake, fake, None, fake, 3, 8, None)          # dones without resets
  expect_error("dx_frame_queue_u8", fake, fake, None, None, fake, 2, 10, 3, 4, 0, None)  # elems % C
  expect_error("dx_frame_queue_u8", fake, fake, None, None, fake, 2, 12, 3, 4, 0, None)  # prev == out
  expect_error("dx_gray_resize_u8", fake, fake, 1, 210, 160, 2, 84, 84, 1, None)     # channels
  expect_error("dx_gather_rows", fake, fake, fake, -1, 16, None)
  # planners: pure host arithmetic writing into the caller's struct
  for shape in ((84, 84, 4), (36, 36, 4), (100, 120, 4), (210, 160, 4)):
    for actions in (1, 4, 18, 31):
      for batch in (1, 7, 256, 8192, 20000):
        ctx = _lib.CnnCtx()
        ctx.struct_bytes = ctypes.sizeof(_lib.CnnCtx)
        ctx.in_h, ctx.in_w, ctx.in_c = shape
        ctx.num_actions, ctx.max_batch = actions, batch
        status = lib.dx_cnn_init(ctypes.byref(ctx))
        calls += 1
        assert status == 0 and ctx.param_count > 0 and ctx.slab_count > 0, (shape, actions, batch, status)
        # forward / backward / act on a context without buffers: refused, nothing launched
        for entry, args in (("dx_cnn_forward", (fake, 1, None, batch, None)),
                            ("dx_cnn_backward", (fake, 1, None, batch, None)),
                            ("dx_cnn_stage", (9, fake, 1, None, batch, None))):
          expect_error(entry, ctypes.byref(ctx), *args)
  bad = _lib.CnnCtx()
  bad.struct_bytes = 8
  expect_error("dx_cnn_init", ctypes.byref(bad))
  for obs_dim in (1, 4, 17, 64, 111, 376):
    for out in (1, 6, 17):
      ctx = _lib.MlpCtx()
      ctx.struct_bytes = ctypes.sizeof(_lib.MlpCtx)
      ctx.obs_dim, ctx.policy_out, ctx.has_logstd, ctx.max_batch = obs_dim, out, 1, 4096
      status = lib.dx_mlp_init(ctypes.byref(ctx))
      calls += 1
      assert status == 0 and ctx.param_count > 0, (obs_dim, out, status)
      expect_error("dx_mlp_forward", ctypes.byref(ctx), fake, 16, None)
  print(f"asan host check OK: {calls} calls into the validation / planning layer")


if __name__ == "__main__":
  main()
