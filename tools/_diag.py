"""ctypes binding of libderl_amd_diag.so (include/derl_amd_diag.h): the microbenchmark entry
points, kept out of the product library.  tools/ only."""
import ctypes
import os

from derl_amd import _lib, build

P, c_int = ctypes.c_void_p, ctypes.c_int
SIGNATURES = {
    "dx_diag_mfma_f32": [c_int, c_int, P, P],
    "dx_diag_mfma_f32_chain": [c_int, c_int, P, P],
    "dx_diag_lds_mfma_f32": [c_int, c_int, c_int, P, P],
    "dx_diag_gemm_loop_f32": [P, P, c_int, c_int, c_int, P, P],
}
_handle = None


def load():
  global _handle
  if _handle is None:
    if not os.path.exists(build.DIAG_LIB):
      build.build_diag_library()
    _handle = ctypes.CDLL(build.DIAG_LIB)
    for name, argtypes in SIGNATURES.items():
      fn = getattr(_handle, name)
      fn.argtypes, fn.restype = argtypes, c_int
  return _handle


def call(name, *args):
  status = getattr(load(), name)(*args)
  if status != 0:
    raise _lib.NativeError(f"{name} failed ({status}): {_lib.last_error()}")
