"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals of ONE training step and the per-queue
busy time.  usage: python tools/trace_summary.py <kernel_trace.csv> [step_index]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if "stem_im2col" in r["Kernel_Name"] or "stem_stats_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
step = [r for r in rows if a <= r["s"] < b]
print("step %d: %.3f ms, %d kernels" % (k, (b - a) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for qid, rs in byq.items():
    print("  queue %s: %d kernels, busy %.2f ms" % (qid, len(rs), sum(r["e"] - r["s"] for r in rs) / 1e6))
agg, cnt = collections.Counter(), collections.Counter()
for r in step:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("frhip::", "").replace("_ZN5frhip", "")
    n = re.sub(r"^\d+", "", n)[:56]
    agg[n] += (r["e"] - r["s"]) / 1e6
    cnt[n] += 1
for n, t in agg.most_common(26):
    print("  %-58s %4d %8.3f ms  avg %7.1f us" % (n, cnt[n], t, t / cnt[n] * 1e3))
print("  sum of kernel durations %.2f ms" % sum(agg.values()))
