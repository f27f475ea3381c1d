"""print per-kernel counter values (last dispatch of each kernel) from tools/pmc_run.sh output"""
import csv, glob, sys, collections
out = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
vals = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/g*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"][:60]
        if filt and filt not in name:
            continue
        vals[name][r["Counter_Name"]] = (float(r["Counter_Value"]), r.get("Grid_Size"), r.get("Dispatch_Id"))
for name, d in vals.items():
    print(name)
    for k, (v, g, i) in d.items():
        print("   %-32s %16.0f" % (k, v))
