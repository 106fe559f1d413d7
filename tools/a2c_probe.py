"""How sensitive is the A2C image-bandit learning check to float32 summation order?  Runs it under
the kernel-route switches given on the command line (one child process each).
usage: python3 tools/a2c_probe.py "DX_NTP_SMALL=0" "DX_LAT_MAX_TILES=0 A2C_LR=5e-5" ...  (A2C_LR, A2C_NENVS, A2C_ITERS
set the run's hyper-parameters)"""
import os
import subprocess
import sys

CODE = """
import sys; sys.path.insert(0, '.')
import numpy as np
from tools.quadrant_learns import run
import os
curve, _ = run(iterations=int(os.environ.get('A2C_ITERS', 300)), nenvs=int(os.environ.get('A2C_NENVS', 64)), horizon=5,
               seed=int(sys.argv[1]), lr=float(os.environ.get('A2C_LR', 1e-4)), algorithm='a2c')
print(round(float(np.mean(curve[-20:])), 3))
"""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for setting in sys.argv[1:] or [""]:
  env = dict(os.environ)
  for item in setting.split():
    key, val = item.split("=")
    env[key] = val
  finals = []
  for seed in (0, 1, 2, 3, 4):
    out = subprocess.run([sys.executable, "-c", CODE, str(seed)], env=env, cwd=root, capture_output=True, text=True)
    finals.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:])
  print(f"{setting or '(default)':28s} final mean reward, seeds 0-4: {finals}", flush=True)
