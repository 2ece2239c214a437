#!/usr/bin/env python3
"""Per-step stream timeline from a rocprofv3 kernel trace (…_kernel_trace.csv): for the last training step, how long each
HIP queue is busy, the time only ONE queue is busy (exposed critical path), and the branch queue's launches in order.
usage: tools/branch_timeline.py <kernel_trace.csv> [--list]"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step ends with the adam kernel; take the last complete step
    ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    lo, hi = ends[-2] + 1, ends[-1]
    step = rows[lo:hi + 1]
    t0 = int(step[0]["Start_Timestamp"])
    t1 = int(step[-1]["End_Timestamp"])
    print(f"step: {(t1 - t0) / 1e3:.1f} us, {len(step)} launches")
    ev = []
    for r in step:
        ev.append((int(r["Start_Timestamp"]), 1, r["Queue_Id"]))
        ev.append((int(r["End_Timestamp"]), -1, r["Queue_Id"]))
    ev.sort()
    busy = {}
    only = {}
    none = 0
    active = {}
    prev = t0
    for t, d, q in ev:
        live = [k for k, v in active.items() if v > 0]
        dt = t - prev
        for k in live:
            busy[k] = busy.get(k, 0) + dt
        if len(live) == 1:
            only[live[0]] = only.get(live[0], 0) + dt
        if not live:
            none += dt
        active[q] = active.get(q, 0) + d
        prev = t
    for q in sorted(busy):
        print(f"queue {q}: busy {busy[q] / 1e3:8.1f} us, alone {only.get(q, 0) / 1e3:8.1f} us")
    print(f"no queue busy: {none / 1e3:.1f} us")
    if "--list" in sys.argv:
        # which kernels run while their queue is the only one busy
        agg = {}
        for r in step:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            others = [(int(o["Start_Timestamp"]), int(o["End_Timestamp"])) for o in step
                      if o["Queue_Id"] != r["Queue_Id"] and int(o["End_Timestamp"]) > s and int(o["Start_Timestamp"]) < e]
            cov = 0
            cur = s
            for os_, oe in sorted(others):
                if oe <= cur:
                    continue
                cov += min(oe, e) - max(os_, cur)
                cur = max(cur, min(oe, e))
            m = re.search(r"(\w+_kernel\w*(<[^(]*>)?)", r["Kernel_Name"])
            name = (m.group(1) if m else r["Kernel_Name"])[:60]
            a = agg.setdefault((r["Queue_Id"], name), [0, 0, 0])
            a[0] += 1
            a[1] += e - s
            a[2] += (e - s) - cov
        for (q, name), (n, tot, alone) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:25]:
            print(f"q{q} {name:60s} n {n:4d} total {tot / 1e3:8.1f} alone {alone / 1e3:8.1f}")


if __name__ == "__main__":
    main()
