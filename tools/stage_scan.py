"""Stage table over several batch sizes: TFLOP/s of every stage relative to its minibatch-8192 figure
(finds batch sizes where a stage falls onto a slow kernel route or an unbalanced walk).
usage: python3 tools/stage_scan.py [batch ...]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batches = [int(v) for v in sys.argv[1:]] or [1152, 1536, 2048, 2560, 3072, 4096, 6144]
table = {}
for b in batches + [8192]:
  out = subprocess.run([sys.executable, os.path.join(root, "tools", "stage_bench.py"), str(b), "5"],
                       capture_output=True, text=True, timeout=300)
  rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{"stage"')]
  table[b] = {r["name"]: (r["us"], r["TFLOPs"]) for r in rows}
ref = table[8192]
names = [n for n in ref if ref[n][1] and ref[n][1] > 10]
print("%-12s" % "stage" + "".join("%9d" % b for b in batches))
for n in names:
  print("%-12s" % n + "".join("%9.2f" % (table[b][n][1] / ref[n][1]) for b in batches))
print("(TFLOP/s relative to minibatch 8192; us:)")
for n in names:
  print("%-12s" % n + "".join("%9.1f" % table[b][n][0] for b in batches))
