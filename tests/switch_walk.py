"""Child process of test_cnn_gpu.py::test_diagnostic_switches_keep_parity: walks the kernel-route switch settings in ONE
process.  The library caches every DX_* switch at first use; dx_reload_env() (include/derl_amd.h) drops the cache, so a
setting is: environment edited, cache dropped, the golden / oracle subset of test_cnn_gpu.py run again (fresh engines
per test).  Prints `SWITCH <setting> -> <pytest exit code>` per setting; exit code 1 if any setting failed.
usage: python tests/switch_walk.py "<setting>" ["<setting>" ...]      (a setting: space-separated NAME=VALUE items)"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
SUBSET = ("(test_forward_matches_reference_golden or test_loss_and_gradients_match_reference_golden or "
          "(test_backward_ragged_batches_with_gather and (130 or 1024)) or (test_fused_rollout_act_matches_unfused_path and not width))")


def main(settings):
  from derl_amd import _lib
  lib = _lib.load()
  touched, failed = set(), []
  for setting in settings:
    for name in touched:  # back to the defaults first
      os.environ.pop(name, None)
    touched.clear()
    for item in setting.split():
      name, value = item.split("=")
      os.environ[name] = value
      touched.add(name)
    assert lib.dx_reload_env() == 0
    code = pytest.main([os.path.join(HERE, "test_cnn_gpu.py"), "-x", "-q", "-m", "gpu", "-k", SUBSET, "-p", "no:cacheprovider"])
    print(f"SWITCH {setting} -> {int(code)}", flush=True)
    if int(code) != 0:
      failed.append(setting)
  return 1 if failed else 0


if __name__ == "__main__":
  sys.exit(main(sys.argv[1:]))
