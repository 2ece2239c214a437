"""Timeline of the last training step of a rocprofv3 --kernel-trace CSV: every launch with start / end offset (us),
queue (= stream) and name, so that overlap between the caller's stream, the weight-gradient side stream and the
branch stream can be read off.  usage: python tools/timeline.py <kernel_trace.csv> [out.txt]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]
win = rows[a + 1:b + 1]
t0 = int(win[0]["Start_Timestamp"])
queues = {}


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:70]


out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
busy_until = 0
for r in win:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{q}  {short(r['Kernel_Name'])}", file=out)
print("queues:", queues, file=out)
# per queue busy time and pairwise overlap
iv = {}
for r in win:
    q = queues[r.get("Queue_Id", "?")]
    iv.setdefault(q, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, l in iv.items():
    print(f"q{q}: {len(l)} launches, {sum(e - s for s, e in l) / 1e3:.1f} us of kernel time", file=out)
print(f"window {(int(win[-1]['End_Timestamp']) - t0) / 1e3:.1f} us", file=out)
