#!/usr/bin/env python3
"""How much of a training step has a chip-filling kernel in flight?  From a rocprofv3 --kernel-trace CSV (last complete
adam-to-adam window): time covered by the persistent / matrix kernels (conv*, wgrad*, tf_chain*: grids of one workgroup per
CU), time where only light kernels run (normalisation statistics, pooling, heads, loss, ...) listed by the kernels that
run there, and idle time.  usage: tools/heavy_cover.py <kernel_trace.csv> [--list N]"""
import collections
import csv
import re
import sys

HEAVY = re.compile(r"conv_|convt_|wgrad|tf_chain_")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    step = rows[ends[-2] + 1:ends[-1] + 1]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    ev = []
    for i, r in enumerate(step):
        h = bool(HEAVY.search(r["Kernel_Name"])) and "reduce" not in r["Kernel_Name"]
        ev.append((int(r["Start_Timestamp"]), 1, h, i))
        ev.append((int(r["End_Timestamp"]), -1, h, i))
    ev.sort()
    nh = nl = 0
    live = set()
    last = t0
    heavy_t = light_t = idle_t = 0
    light_by = collections.Counter()
    segs = []
    for t, d, h, i in ev:
        dt = t - last
        if dt > 0:
            if nh:
                heavy_t += dt
            elif nl:
                light_t += dt
                names = sorted({short(step[j]["Kernel_Name"]) for j in live})
                for n in names:
                    light_by[n] += dt / len(names)
                segs.append((last - t0, dt, names))
            else:
                idle_t += dt
                segs.append((last - t0, dt, ["<idle>"]))
        last = t
        if h:
            nh += d
        else:
            nl += d
            (live.add if d > 0 else live.discard)(i)
    print(f"step {(t1 - t0) / 1e3:.1f} us: heavy kernel in flight {heavy_t / 1e3:.1f}, only light kernels {light_t / 1e3:.1f}, idle {idle_t / 1e3:.1f}")
    for n, t in light_by.most_common(25):
        print(f"  {t / 1e3:8.1f} us  {n}")
    if "--list" in sys.argv:
        k = int(sys.argv[sys.argv.index("--list") + 1])
        for s, dt, names in sorted(segs, key=lambda x: -x[1])[:k]:
            print(f"  at {s / 1e3:9.1f} for {dt / 1e3:7.1f} us: {', '.join(names)}")


main()
