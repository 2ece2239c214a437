"""Average duration per kernel name from a rocprofv3 kernel_stats.csv (diagnostic): python tools/kstat.py <csv> [substr...]"""
import csv
import sys
subs = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if not subs or any(s in n for s in subs):
        print(f"{float(r['AverageNs']) / 1e3:9.1f} us avg {int(r['Calls']):6d} calls {float(r['TotalDurationNs']) / 1e3:11.1f} us total  {n[:80]}")
