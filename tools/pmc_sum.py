"""Sum a rocprofv3 --pmc counter per kernel name (diagnostic).

usage: python tools/pmc_sum.py <counter_collection.csv> [substring]
Prints calls, total and mean counter value for every kernel whose name contains `substring` (default: all)."""
import csv
import re
import sys
from collections import defaultdict

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0, 0.0])
ctr = None
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sub not in n:
        continue
    ctr = r["Counter_Name"]
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    acc[n][0] += 1
    acc[n][1] += float(r["Counter_Value"])
print("counter", ctr)
for n, (c, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{v:16.0f} total {v / c:14.0f} mean {c:5d} calls  {n}")
