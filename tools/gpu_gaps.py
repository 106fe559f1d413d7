"""GPU idle time inside the timed iterations of a bench run, from a rocprofv3 kernel trace:
merges the kernels of all streams into busy intervals and reports the gaps (where, how long).
usage: python3 tools/gpu_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = []
with open(path) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the training loop = the last 60 % of the trace by time (after warm-up, before the tail)
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
lo, hi = t0 + 0.35 * (t1 - t0), t0 + 0.9 * (t1 - t0)
sel = [r for r in rows if lo <= r[0] <= hi]
busy_end, prev_name = sel[0][1], sel[0][2]
idle, gaps = 0, []
for s, e, name in sel[1:]:
  if s > busy_end:
    gap = (s - busy_end) / 1e3
    idle += s - busy_end
    if gap >= min_gap:
      gaps.append((gap, prev_name[:60], name[:60]))
  if e > busy_end:
    busy_end, prev_name = e, name
span = sel[-1][1] - sel[0][0]
print(f"span {span / 1e6:.2f} ms, idle {idle / 1e6:.3f} ms = {100.0 * idle / span:.2f} %, {len(sel)} launches, "
      f"mean gap {idle / 1e3 / max(1, len(sel) - 1):.2f} us")
import collections
by = collections.Counter()
tot = collections.Counter()
for gap, a, b in gaps:
  by[(a, b)] += 1
  tot[(a, b)] += gap
print(f"gaps >= {min_gap} us: {len(gaps)}, total {sum(g for g, _, _ in gaps) / 1e3:.3f} ms")
for key, t in tot.most_common(14):
  print(f"  {t / 1e3:7.3f} ms in {by[key]:5d} gaps (mean {t / by[key]:6.1f} us)  after {key[0]}  before {key[1]}")
