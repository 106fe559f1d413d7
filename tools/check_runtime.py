"""Prints which HIP runtime copies are mapped after loading torch + libderl_amd.so."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from derl_amd import _lib
_lib.load()
torch.zeros(1, device="cuda")
libs = sorted({line.split()[-1] for line in open("/proc/self/maps")
               if "amdhip64" in line or "libderl_amd" in line or "hsa-runtime" in line})
print("\n".join(libs))
import ctypes
name = ctypes.create_string_buffer(256)
cu, lds = ctypes.c_int(), ctypes.c_int()
_lib.call("dx_device_info", 0, name, ctypes.byref(cu), ctypes.byref(lds))
print(name.value.decode(), "CUs", cu.value, "LDS/block", lds.value)
