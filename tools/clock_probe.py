"""Samples the GPU clock / power (rocm-smi) while (a) the pure MFMA loop and (b) the full GEMM K loop
run for ~2 s each.  usage: python tools/clock_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd import _lib  # noqa: E402
from tools import _diag  # noqa: E402

dev = torch.device("cuda:0")
out = torch.zeros(4, device=dev)
stream = _lib.stream_ptr(dev)
samples, stop = [], [False]


def sampler():
  while not stop[0]:
    try:
      txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True,
                           timeout=10).stdout
      keep = [l.strip() for l in txt.splitlines() if "sclk" in l or "mclk" in l or "Power" in l or "fclk" in l]
      samples.append((time.perf_counter(), keep))
    except Exception as exc:  # pylint: disable=broad-except
      samples.append((time.perf_counter(), [repr(exc)]))
    time.sleep(0.05)


def phase(name, fn, seconds=2.5):
  samples.clear()
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    for _ in range(20):
      fn()
    torch.cuda.synchronize()
  mid = samples[len(samples) // 2:] if samples else []
  print("==", name, "samples", len(samples), flush=True)
  for _, keep in mid[:3]:
    print("   ", " | ".join(keep), flush=True)


thread = threading.Thread(target=sampler, daemon=True)
thread.start()
time.sleep(0.5)
print("== idle", samples[-1][1] if samples else None, flush=True)
phase("pure MFMA loop", lambda: _diag.call("dx_diag_mfma_f32", 1024, 20000, _lib.ptr(out), stream))
tiles, ktiles = 5184, 16
A = torch.randn(tiles * 128, 32 * ktiles, device=dev)
B = torch.randn(64, 32 * ktiles, device=dev)
phase("GEMM loop without global loads", lambda: _diag.call("dx_diag_gemm_loop_f32", _lib.ptr(A), _lib.ptr(B), tiles, ktiles, 1, _lib.ptr(out), stream))
phase("GEMM loop with global loads (HBM stream)", lambda: _diag.call("dx_diag_gemm_loop_f32", _lib.ptr(A), _lib.ptr(B), tiles, ktiles, 2, _lib.ptr(out), stream))
phase("GEMM loop, A cache-resident", lambda: _diag.call("dx_diag_gemm_loop_f32", _lib.ptr(A), _lib.ptr(B), tiles, ktiles, 4, _lib.ptr(out), stream))
stop[0] = True
