"""Sum of idle gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (diagnostic).

usage: python tools/gap_probe.py <kernel_trace.csv>
Prints, for the last adam-to-adam window (one train step): wall time, busy time, idle time, launches, and the idle
time grouped by the kernel that FOLLOWS the gap."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 2:
    sys.exit("need two optimizer steps in the trace")
a, b = adam[-2], adam[-1]
win = rows[a + 1:b + 1]
wall = int(win[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
gaps = defaultdict(lambda: [0, 0])
prev_end = int(rows[a]["End_Timestamp"])
idle = 0
for r in win:
    g = int(r["Start_Timestamp"]) - prev_end
    if g > 0:
        idle += g
        k = r["Kernel_Name"].split("(")[0][-60:]
        gaps[k][0] += g
        gaps[k][1] += 1
    prev_end = max(prev_end, int(r["End_Timestamp"]))
print(f"step wall {wall/1e6:.3f} ms  busy(sum) {busy/1e6:.3f} ms  idle {idle/1e6:.3f} ms  launches {len(win)}")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g/1e3:9.1f} us {n:5d} gaps  avg {g/n/1e3:6.2f} us  before {k}")
