"""Builds the in-tree native libraries with hipcc for gfx950 (cross-compiles without a GPU).

``build_library()``       derl_amd/csrc/*.hip -> derl_amd/libderl_amd.so: the C-ABI declared in
                          include/derl_amd.h (the product).
``build_diag_library()``  the product sources with -DDX_DIAG=1 (in-kernel stamps and the bisecting
                          switches DX_NTP_DIAG / DX_NT_DIAG / DX_WD_DIAG / DX_FC_DIAG / DX_NTP_NWG /
                          DX_WD_NWG, some of which compute wrong results on purpose: compiled OUT of
                          libderl_amd.so) + derl_amd/csrc/experiments/diag.hip (the microbenchmark
                          entry points of include/derl_amd_diag.h) -> derl_amd/libderl_amd_diag.so.
                          Self-contained; tools/ load it with DERL_AMD_LIBRARY=diag.
``build_host_asan()``     the HOST side of every product source (argument validation, planning,
                          launch set-up) compiled with -fsanitize=address into
                          derl_amd/libderl_amd_hostasan.so; device code is not built
                          (--cuda-host-only).  GPU AddressSanitizer is not available on this pool;
                          this is the sanitizer build of the shim.
``DERL_AMD_EXPERIMENTS=1`` adds csrc/experiments/igemm_b3.hip (the split-bf16 GEMM, a recorded
                          negative result) to the product build behind -DDX_EXPERIMENT_B3.

The .so files are git-ignored but travel to the GPU box with the working tree.
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
DIAG_LIB = os.path.join(PKG, "libderl_amd_diag.so")
ASAN_LIB = os.path.join(PKG, "libderl_amd_hostasan.so")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
INCLUDE = os.path.join(os.path.dirname(PKG), "include")


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  mtime = os.path.getmtime(target)
  return any(os.path.getmtime(d) > mtime for d in deps)


def _headers():
  return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h")))


def _compile(src, headers, verbose, extra=(), tag=""):
  obj = os.path.join(OBJ, os.path.basename(src) + tag + ".o")
  if _stale(obj, [src] + headers):
    cmd = [HIPCC, *FLAGS, *extra, "-c", src, "-o", obj]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
  return obj


def _link(objs, lib, verbose, extra=()):
  cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, *extra, "-o", lib]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.run(cmd, check=True)


def build_library(force=False, verbose=False, jobs=4):
  """Compiles (if stale) and returns the path of libderl_amd.so."""
  experiments = os.environ.get("DERL_AMD_EXPERIMENTS", "0") not in ("", "0")
  sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
  extra, tag = (), ""
  if experiments:
    sources.append(os.path.join(CSRC, "experiments", "igemm_b3.hip"))
    extra, tag = ("-DDX_EXPERIMENT_B3",), ".exp"
  headers = _headers()
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
    objs = list(pool.map(lambda s: _compile(s, headers, verbose, extra, tag), sources))
  marker = os.path.join(OBJ, ".flavour")
  flavour = "exp" if experiments else "default"
  previous = open(marker).read() if os.path.exists(marker) else ""
  if force or previous != flavour or _stale(LIB, objs):
    _link(objs, LIB, verbose, ["-ldl"])
    with open(marker, "w") as f:
      f.write(flavour)
  return LIB


def _uses_diag_macro(src):
  with open(src) as f:
    return "DX_DIAG" in f.read()


def build_diag_library(verbose=False, jobs=4):
  """The diagnostic flavour: every product source that looks at DX_DIAG recompiled with
  -DDX_DIAG=1 (the others' objects are shared with the product build) + the microbenchmarks."""
  if not os.path.exists(HIPCC):
    return DIAG_LIB if os.path.exists(DIAG_LIB) else None
  build_library(verbose=verbose)
  os.makedirs(OBJ, exist_ok=True)
  headers = _headers()
  sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))

  def one(src):
    if _uses_diag_macro(src):
      return _compile(src, headers, verbose, ("-DDX_DIAG=1",), ".diag")
    return _compile(src, headers, verbose)

  with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
    objs = list(pool.map(one, sources))
  objs.append(_compile(os.path.join(CSRC, "experiments", "diag.hip"), headers, verbose))
  if _stale(DIAG_LIB, objs):
    _link(objs, DIAG_LIB, verbose, ["-ldl"])
  return DIAG_LIB


def build_host_asan(verbose=False, jobs=4):
  """Host-only AddressSanitizer build of the product sources (no device code)."""
  if not os.path.exists(HIPCC):
    raise RuntimeError("hipcc not found")
  os.makedirs(OBJ, exist_ok=True)
  sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
  headers = _headers()
  extra = ("--cuda-host-only", "-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-O1")

  def one(src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".hostasan.o")
    if _stale(obj, [src] + headers):
      cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-function", *extra, "-c", src, "-o", obj]
      if verbose:
        print(" ".join(cmd), flush=True)
      subprocess.run(cmd, check=True)
    return obj

  with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
    objs = list(pool.map(one, sources))
  if _stale(ASAN_LIB, objs):
    # a host-only object still refers to the device image of its translation unit
    # (__hip_fatbin_<hash>, normally embedded by the offload bundler): define each as an empty
    # blob -- the HIP runtime registers it lazily and no kernel is ever launched from this build
    undefined = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True, check=True).stdout
    blobs = sorted({tok for tok in undefined.split() if tok.startswith("__hip_fatbin_")})
    stub = os.path.join(OBJ, "hostasan_fatbin_stub.c")
    with open(stub, "w") as f:
      f.write("/* generated by derl_amd/build.py: empty device images for the host-only ASan build */\n")
      for name in blobs:
        f.write(f"const char {name}[16] __attribute__((aligned(16))) = {{0}};\n")
    stub_obj = stub[:-2] + ".o"
    subprocess.run(["gcc", "-fPIC", "-c", stub, "-o", stub_obj], check=True)
    cmd = [HIPCC, "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", *objs, stub_obj, "-ldl", "-o", ASAN_LIB]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
  return ASAN_LIB


def asan_runtime():
  """Path of clang's shared ASan runtime (to LD_PRELOAD under a non-instrumented python)."""
  out = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
  path = out.stdout.strip()
  if os.path.isabs(path) and os.path.exists(path):
    return path
  found = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
  return found[0] if found else None


if __name__ == "__main__":
  print(build_library(verbose=True))
  print(build_diag_library(verbose=True))
