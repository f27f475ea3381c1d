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
hdr_o = ("# the same step as it really runs: weight gradients on the side stream (queue 2), co-resident with the main stream's kernels;\n"
         "# per-kernel durations are inflated by the sharing, the queues' busy times overlap.  rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline\n")
open(os.path.join(P, TAG + "_final_step_anatomy_overlapped.txt"), "w").write(hdr_o + open(os.path.join(F, "step_anatomy_overlapped.txt")).read())
hdr_a = ("# AlterNet50 @192 with the fp8 forward path (BASELINE cfg 5), one training step, weight gradients on the main stream:\n"
         "# rocprofv3 --kernel-trace -- python3 bench.py --network AlterNet50 --fp8 --steps 6 --warmup 3 --no-cpu-baseline ; tools/trace_summary.py\n")
open(os.path.join(P, TAG + "_final_alternet50_fp8_step_anatomy.txt"), "w").write(hdr_a + open(os.path.join(F, "alt_step_anatomy.txt")).read())
# MFMA utilisation counters of the dominant kernels (tools/pmc_run.sh: one rocprofv3 --pmc pass per counter group, kernel-trace only)
have_pmc = any(os.path.exists(os.path.join(F, "pmc_%s.txt" % w)) for w in ("fwd", "dgrad", "wgrad", "wgrad8"))
with open(os.path.join(P, TAG + "_pmc_mfma_util.txt") if have_pmc else os.devnull, "w") as out:       # QUICK=1 runs keep the committed counters
    out.write("# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/pmc_one.py fwd|dgrad|wgrad 14 256 256   (B = 512, 256-channel 14x14 layer, bf16)\n"
              "# groups: tools/pmc_run.sh.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4 (four SIMDs per CU).\n"
              "# wgrad = the 4-wave co-resident kernel the step uses (14 x 14: tn_rows14_kernel, 32x32x16 MFMAs of which 14 of 16 K slots carry pixels);\n"
              "# wgrad8 = the 8-wave pixel-stream tile (FRHIP_T9_NARROW=0; the stand-alone-fastest choice until round 4).\n")
    for what in ("fwd", "dgrad", "wgrad", "wgrad8"):
        f = os.path.join(F, "pmc_%s.txt" % what)
        if not os.path.exists(f):
            continue
        cur, vals = None, {}
        for line in open(f):
            if not line.startswith("   "):
                cur = line.strip()
                vals[cur] = {}
            else:
                k, v = line.split()
                vals[cur][k] = float(v)
        for k, v in vals.items():
            if ("halo_kernel" in k or "halo_wide_kernel" in k or "taps9" in k or "rows14" in k) and v.get("SQ_BUSY_CU_CYCLES"):
                out.write("\n[%s] %s\n" % (what, k[:110]))
                out.write("  MFMA utilisation            %.3f\n" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CU_CYCLES"] / 4))
                wc = v["SQ_WAVE_CYCLES"]
                out.write("  wave cycles: waiting (s_waitcnt / barrier) %.2f, issue-stalled %.2f, issuing %.2f\n" % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc))
                out.write("  LDS: bank-conflict cycles %.2f of LDS-active cycles; LDS issue stall %.3f of wave cycles\n" % (v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_WAIT_INST_LDS"] / wc))
                out.write("  instructions: MFMA %.3g, VALU %.3g (%.2f per MFMA), SALU %.3g, LDS %.3g; MFMA || VALU co-execution %.2f of MFMA-busy cycles\n" % (
                    v["SQ_INSTS_MFMA"], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"], v["SQ_VALU_MFMA_COEXEC_CYCLES"] / v["SQ_VALU_MFMA_BUSY_CYCLES"]))
                for name in sorted(v):
                    out.write("    %-32s %16.0f\n" % (name, v[name]))
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
if glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "fetch", "*", "*_counter_collection.csv")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_report.py"), os.path.join(ROOT, "gpurun_out", "pmc"), TAG],
                          stdout=subprocess.DEVNULL)
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["cpu_baseline"]["value"])

# in-kernel clock, ablation and inference throughput (tools/clock_probe.py, tools/ablate.py, tools/bench_eval.py)
for src, dst, hdr in (("clock.txt", "_inkernel_clock.txt", "# tools/clock_probe.py run: s_memtime / s_memrealtime stamps around the main loop of every workgroup (diagnostic build)\n"),
                      ("ablate.txt", "_halo_ablation.txt", "# tools/ablate.py run: timing-only variants of the 4-wave halo kernel (results of the ablated variants are wrong by design), forward, no statistics\n"),
                      ("eval.txt", "_eval_throughput.txt", "# tools/bench_eval.py ResNet50 512: encoder in eval mode, BatchNorm folded into the conv epilogues (1) against separate passes (0)\n")):
    f = os.path.join(F, src)
    if os.path.exists(f):
        body = "".join(l for l in open(f) if "amdgpu.ids" not in l)
        open(os.path.join(P, TAG + dst), "w").write(hdr + body)
