"""Launch-by-launch timeline of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py: start offset, duration, queue,
workgroups and kernel, in start order -- shows which stage of the network a launch belongs to and what runs beside it.
usage: python tools/trace_timeline.py <kernel_trace.csv> [step_index] [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if "stem_im2col" in r["Kernel_Name"] or "stem_stats_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else len(starts) - 3
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
a, b = starts[k], starts[k + 1]
step = [r for r in rows if a <= r["s"] < b]
queues = sorted({r["Queue_Id"] for r in step}, key=lambda q: -sum(1 for r in step if r["Queue_Id"] == q))
print("step %d: %.3f ms, %d kernels, queues %s" % (k, (b - a) / 1e6, len(step), queues))
prev_end = {}
for r in step:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("frhip::", "").replace("_ZN5frhip", "")
    n = re.sub(r"^\d+", "", n)[:60]
    q = queues.index(r["Queue_Id"])
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
    gap = (r["s"] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r["e"]
    d = (r["e"] - r["s"]) / 1e3
    if d >= min_us:
        print("%9.1f us  q%d  %8.1f us  gap %6.1f  wgs %6d  %s" % ((r["s"] - a) / 1e3, q, d, gap, wg, n))
