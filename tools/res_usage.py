"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file) to one line per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "")
    if pat and not re.search(pat, dem):
        continue

    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"

    print("%-92s V%s A%s scr%s lds%s occ%s" % (dem[:92], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"),
                                             g("LDS Size [bytes/block]"), g("Occupancy [waves/SIMD]")))
