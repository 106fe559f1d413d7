"""Timeline of ONE iteration of a bench run from a rocprofv3 kernel trace: the last complete iteration (from one rollout
launch to the next), every idle gap >= min_gap_us and the busy stretches between them.
usage: python3 tools/gpu_iter_timeline.py <kernel_trace.csv> [min_gap_us] [rollout kernel substring]"""
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
roll = sys.argv[3] if len(sys.argv) > 3 else "convstack_roll_kernel"
rows = []
with open(path) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [s for s, _, n in rows if roll in n]
# iterations begin where a rollout launch follows a non-rollout kernel by more than the horizon's own launches
firsts = [s for i, s in enumerate(starts) if i == 0 or s - starts[i - 1] > 2.5e6]
if len(firsts) < 3:
  firsts = starts[::2]
lo, hi = firsts[-3], firsts[-2]
sel = [r for r in rows if lo <= r[0] < hi]
print(f"iteration of {(hi - lo) / 1e6:.3f} ms, {len(sel)} launches")
busy_start, busy_end, names = sel[0][0], sel[0][1], [sel[0][2]]
kernel_time = 0
for s, e, n in sel:
  kernel_time += e - s
for s, e, n in sel[1:] + [(hi, hi, "next iteration")]:
  if s > busy_end + min_gap * 1e3:
    print(f"  {(busy_start - lo) / 1e3:9.1f} us  busy {(busy_end - busy_start) / 1e3:8.1f} us  ({len(names)} launches: {names[0][:50]} .. {names[-1][:50]})")
    print(f"  {(busy_end - lo) / 1e3:9.1f} us  IDLE {(s - busy_end) / 1e3:8.1f} us")
    busy_start, busy_end, names = s, e, [n]
  else:
    busy_end = max(busy_end, e)
    names.append(n)
print(f"sum of kernel durations {kernel_time / 1e6:.3f} ms")
