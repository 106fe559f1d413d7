"""Sums rocprofv3 --pmc counter CSVs per kernel: usage python3 tools/pmc_summary.py <dir> [name filter]
Prints per (kernel, counter): dispatches, mean value per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
  with open(path) as f:
    for row in csv.DictReader(f):
      name = row.get("Kernel_Name", "")
      if flt and flt not in name:
        continue
      key = (name[:90], row["Counter_Name"])
      acc[key][0] += 1
      acc[key][1] += float(row["Counter_Value"])
for (name, counter), (n, total) in sorted(acc.items()):
  print(f"{name:90s} {counter:32s} n={n:4d} mean={total / n:.4g}")
