"""Builds the in-tree native libraries with hipcc for gfx950 (cross-compiles without a GPU).

``build_library()`` compiles ``derl_amd/csrc/*.hip`` into ``derl_amd/libderl_amd.so`` --
the C-ABI declared in ``include/derl_amd.h``.  The .so is git-ignored but travels to the
GPU box with the working tree.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "_obj")
LIB = os.path.join(PKG, "libderl_amd.so")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  mtime = os.path.getmtime(target)
  return any(os.path.getmtime(d) > mtime for d in deps)


def _compile(src, headers, verbose):
  obj = os.path.join(OBJ, os.path.basename(src) + ".o")
  if _stale(obj, [src] + headers):
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
  return obj


def build_library(force=False, verbose=False, jobs=4):
  """Compiles (if stale) and returns the path of libderl_amd.so."""
  sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
  headers = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [
      os.path.join(os.path.dirname(PKG), "include", "derl_amd.h")]
  if not sources:
    raise RuntimeError(f"no HIP sources under {CSRC}")
  os.makedirs(OBJ, exist_ok=True)
  if force:
    for f in glob.glob(os.path.join(OBJ, "*.o")):
      os.remove(f)
  if not os.path.exists(HIPCC):
    if os.path.exists(LIB):
      return LIB  # no toolchain on this box: use the shipped build
    raise RuntimeError("hipcc not found and no prebuilt libderl_amd.so")
  with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
    objs = list(pool.map(lambda s: _compile(s, headers, verbose), sources))
  if force or _stale(LIB, objs):
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
  return LIB


if __name__ == "__main__":
  print(build_library(verbose=True))
