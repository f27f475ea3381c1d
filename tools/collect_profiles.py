"""Turn gpurun_out/final (tools/final_profiles.sh) + gpurun_out/pmc into the committed profiles/<tag>_* artefacts.
usage: python tools/collect_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
rows = list(csv.DictReader(open(os.path.join(F, "stats", "r50_kernel_stats.csv"))))
tot = sum(int(r["TotalDurationNs"]) for r in rows) / 1e6
with open(os.path.join(P, TAG + "_final_bench_kernel_stats.txt"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline   (%s final, MI355X, B=512 ResNet50 bf16, eager + side-stream wgrad)\n" % TAG)
    out.write("# total kernel time %.1f ms; the run contains 1 recorded eager step + 3 warm-up + 10 timed steps + the roofline probe (4 x 106 conv calls)\n" % tot)
    out.write("# per-kernel averages are inflated where the side stream overlaps (weight gradients vs the BN passes); the serial anatomy of one step is the *_final_step_anatomy.txt next to this file\n")
    out.write("%-100s %7s %12s %10s %6s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:40]:
        out.write("%-100s %7d %12.3f %10.1f %6.1f\n" % (r["Name"][:100], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
shutil.copy(os.path.join(F, "stats", "r50_kernel_stats.csv"), os.path.join(P, TAG + "_final_bench_kernel_stats.csv"))
hdr = ("# one training step, weight gradients on the main stream (FRHIP_OVERLAP_WGRAD=0) so kernel durations add up to the step:\n"
       "# rocprofv3 --kernel-trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline ; tools/trace_summary.py\n")
open(os.path.join(P, TAG + "_final_step_anatomy.txt"), "w").write(hdr + open(os.path.join(F, "step_anatomy.txt")).read())
hdr2 = ("# Swin34 (BASELINE cfg 4), one training step, weight gradients on the main stream (FRHIP_OVERLAP_WGRAD=0):\n"
        "# rocprofv3 --kernel-trace -- python3 bench.py --network Swin34 --steps 6 --warmup 3 --no-cpu-baseline ; tools/trace_summary.py\n")
open(os.path.join(P, TAG + "_final_swin34_step_anatomy.txt"), "w").write(hdr2 + open(os.path.join(F, "swin_step_anatomy.txt")).read())
d = json.loads(open(os.path.join(F, "bench.json")).read().strip().splitlines()[-1])
open(os.path.join(P, TAG + "_final_bench.json"), "w").write(json.dumps(d, indent=1) + "\n")
# keep only the newest PMC pass of each counter
for sub in ("fetch", "write"):
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[:-1]:
        for g in glob.glob(f.replace("_counter_collection.csv", "_*")):
            os.remove(g)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_report.py"), os.path.join(ROOT, "gpurun_out", "pmc"), TAG],
                      stdout=subprocess.DEVNULL)
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["cpu_baseline"]["value"])
