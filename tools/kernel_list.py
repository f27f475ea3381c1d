import csv, sys, collections
# full per-step kernel list (all names) of the last step in a kernel trace
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# step boundaries: sgd_multi_kernel is the last kernel(s) of a step
ends = [i for i, n in enumerate(names) if n.startswith("frhip::sgd_multi") or "sgd_multi_kernel" in n]
# group consecutive
last = ends[-1]; prev = max(e for e in ends if e < last - 5)
seg = rows[prev + 1:last + 1]
c = collections.Counter(); t = collections.Counter()
for r in seg:
    n = r["Kernel_Name"][:90]; c[n] += 1; t[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(len(seg), "kernels")
for n, k in sorted(c.items(), key=lambda kv: -t[kv[0]]): print(f"{k:5d} {t[n]:9.1f} us  {n}")
