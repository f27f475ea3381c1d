"""Per-layer averages of the halo / nine-tap launches of one step (rocprofv3 --kernel-trace CSV, FRHIP_OVERLAP_WGRAD=0).
usage: python tools/halo_layers.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if "stem_stats_kernel" in r["Kernel_Name"]]
k = len(starts) - 3
a, b = starts[k], starts[k + 1]
head = [r["s"] for r in rows if a <= r["s"] < b and "head_kernel" in r["Kernel_Name"]]
mid = head[0] if head else (a + b) // 2
agg = collections.defaultdict(list)
for r in rows:
    if a <= r["s"] < b and ("halo_kernel" in r["Kernel_Name"] or "taps9" in r["Kernel_Name"]):
        wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        agg[("halo" if "halo" in r["Kernel_Name"] else "taps9", wg, "fwd" if r["s"] < mid else "bwd")].append((r["e"] - r["s"]) / 1e3)
for key, v in sorted(agg.items()):
    print("%-6s wgs=%5d %s  n=%2d  avg %6.1f us  sum %.2f ms" % (key[0], key[1], key[2], len(v), sum(v) / len(v), sum(v) / 1e3))
