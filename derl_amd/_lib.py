"""ctypes binding of the C-ABI in include/derl_amd.h (libderl_amd.so).

Fails loudly: there is no CPU or PyTorch fallback for any hot-path op.  If the shared
library is missing or a call returns non-zero, a ``NativeError`` carrying
``dx_last_error()`` is raised.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libderl_amd.so")
ABI_VERSION = 1

c_int, c_float, c_void_p, c_char_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p
c_size_t, c_int64, c_uint64, c_double = ctypes.c_size_t, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
P = c_void_p

# name -> argtypes; every function returns int (0 = OK) unless listed in _RESTYPES
SIGNATURES = {
    "dx_abi_version": [],
    "dx_last_error": [],
    "dx_device_info": [c_int, c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "dx_gae_f32": [P, P, P, P, c_int, c_int, c_float, c_float, P, P, P],
}
_RESTYPES = {"dx_last_error": c_char_p}

_lib = None


class NativeError(RuntimeError):
  """A C-ABI call failed (carries dx_last_error())."""


def load():
  """Loads libderl_amd.so once.  ``import torch`` must have happened first so that the
  HIP runtime torch ships (soname libamdhip64.so.7) is the one the library binds to."""
  global _lib
  if _lib is not None:
    return _lib
  import torch  # noqa: F401  (loads libamdhip64 before our DT_NEEDED is resolved)
  if not os.path.exists(LIB_PATH):
    raise NativeError(
        f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  derl_amd has no CPU fallback.")
  lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
  for name, argtypes in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the .so is stale
    fn.argtypes = argtypes
    fn.restype = _RESTYPES.get(name, c_int)
  if lib.dx_abi_version() != ABI_VERSION:
    raise NativeError(f"libderl_amd.so ABI {lib.dx_abi_version()} != binding {ABI_VERSION}")
  _lib = lib
  return lib


def last_error():
  msg = load().dx_last_error()
  return msg.decode() if msg else ""


def check(status, what):
  if status != 0:
    raise NativeError(f"{what} failed with status {status}: {last_error()}")


def call(name, *args):
  """Calls a C-ABI function and raises NativeError on a non-zero status."""
  check(getattr(load(), name)(*args), name)


def ptr(tensor):
  """Device (or host) address of a torch tensor as a void*; None -> NULL."""
  if tensor is None:
    return None
  return c_void_p(tensor.data_ptr())


def stream_ptr(device=None):
  """The current torch HIP stream as void* (torch is the stream/memory plumbing)."""
  import torch
  return c_void_p(torch.cuda.current_stream(device).cuda_stream)
