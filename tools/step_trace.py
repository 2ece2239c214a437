"""Per-launch listing of one train step from a rocprofv3 --kernel-trace CSV (diagnostic).

usage: python tools/step_trace.py <kernel_trace.csv> [min_us]
Prints every launch of the last adam-to-adam window that took at least min_us (default 30): start offset, duration,
grid and a shortened kernel name, then the totals per kernel name."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]
win = rows[a + 1:b + 1]
t0 = int(win[0]["Start_Timestamp"])
tot = defaultdict(lambda: [0.0, 0])


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:64]


for r in win:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = short(r["Kernel_Name"])
    tot[k][0] += d
    tot[k][1] += 1
    if d >= min_us:
        grid = "x".join(str(int(r[c]) // max(1, int(r[w]))) for c, w in
                        (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"),
                         ("Grid_Size_Z", "Workgroup_Size_Z")) if c in r)
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f} us  {grid:14s} {k}")
print("---- totals")
for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{d:9.1f} us {n:4d}  {k}")
