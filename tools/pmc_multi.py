"""Per-kernel means of every counter of one rocprofv3 --pmc pass (several counters per pass).

usage: python tools/pmc_multi.py <dir with *_counter_collection.csv> [kernel substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sub not in n:
            continue
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
        a = acc[n][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for n, cs in acc.items():
    print(n)
    for c, (k, v) in sorted(cs.items()):
        print(f"   {c:28s} {v / k:16.0f} mean over {k} dispatches")
