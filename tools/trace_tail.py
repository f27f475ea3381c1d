"""Timeline of the last part of one training step from a rocprofv3 --kernel-trace CSV: which queue runs what near the join of the
weight-gradient side stream.  usage: python tools/trace_tail.py <kernel_trace.csv> [ms_from_end] """
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
span = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
nm = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("frhip::", "")[:44]
qs = sorted(set(r["Queue_Id"] for r in step))
print("step %.3f ms; queues %s" % ((b - a) / 1e6, qs))
# busy time of each queue per 1-ms slice of the step
nsl = int((b - a) / 1e6) + 1
for q in qs:
    busy = [0.0] * nsl
    for r in step:
        if r["Queue_Id"] != q:
            continue
        s, e = r["s"] - a, r["e"] - a
        for i in range(int(s / 1e6), min(int(e / 1e6), nsl - 1) + 1):
            busy[i] += max(0, min(e, (i + 1) * 1e6) - max(s, i * 1e6)) / 1e6
    print("queue %s busy per ms: %s" % (q, " ".join("%.2f" % v for v in busy)))
for r in step:
    if (b - r["s"]) / 1e6 <= span and (r["e"] - r["s"]) > 15000:
        print("  t=%7.3f  q%s  %7.1f us  %s" % ((r["s"] - a) / 1e6, r["Queue_Id"], (r["e"] - r["s"]) / 1e3, nm(r)))
