"""Largest idle gaps between consecutive kernels of one training step (rocprofv3 --kernel-trace CSV of bench.py).
usage: python tools/trace_gaps.py <kernel_trace.csv> [n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if "stem_stats_kernel" in r["Kernel_Name"]]
k = len(starts) - 3
a, b = starts[k], starts[k + 1]
step = [r for r in rows if a <= r["s"] < b]
nm = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("frhip::", "")[:48]
gaps = []
end = step[0]["e"]
for prev, cur in zip(step, step[1:]):
    end = max(end, prev["e"])
    gaps.append((cur["s"] - end, (cur["s"] - a) / 1e6, nm(prev), nm(cur)))
gaps.sort(reverse=True)
print("step %.3f ms, idle %.3f ms in %d gaps > 20 us" % ((b - a) / 1e6, sum(g[0] for g in gaps if g[0] > 0) / 1e6, sum(1 for g in gaps if g[0] > 20000)))
for g in gaps[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print("  %8.1f us at t=%6.2f ms   after %-48s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
