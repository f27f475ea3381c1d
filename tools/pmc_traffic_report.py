"""Turn the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into profiles/<tag>_pmc_hbm_traffic.txt (per kernel) and
profiles/<tag>_pmc_traffic.json (the dominant conv kernels, read by bench.py for roofline.traffic).
usage: python tools/pmc_traffic_report.py gpurun_out/pmc r01"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(sub, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals


fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
rows = []
for k in fetch:
    if k not in write or "frhip" not in k:
        continue
    f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
    rows.append((len(fetch[k]) * (2 * f + w), k, len(fetch[k]), f, w, (2 * f + w) * 1024 / 1e6))
rows.sort(reverse=True)
out = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % tag)
with open(out, "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline\n")
    fh.write("# MI355X, B=512 ResNet50 bf16.  Counter unit = KB.  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024:\n")
    fh.write("# MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE is exact.\n")
    fh.write("%-96s %6s %12s %12s %14s\n" % ("kernel", "calls", "fetch_KB_avg", "write_KB_avg", "hbm_MB/launch"))
    for _, k, n, f, w, mb in rows:
        fh.write("%-96s %6d %12.1f %12.1f %14.1f\n" % (k[:96], n, f, w, mb))
# dominant kernel = the bf16 conv implicit GEMMs with the store epilogue (halo + generic NT)
sel = [r for r in rows if "halo_wide_kernel" in r[1] or
       (("halo_kernel" in r[1] or ("nt_kernel" in r[1] and "Li0EEEv" in r[1])) and "DF16b" in r[1] and "halo8" not in r[1])]
n = sum(r[2] for r in sel)
avg = sum(r[2] * r[5] for r in sel) / n * 1e6
import hashlib


def csrc_digest():
    """same as bench.csrc_digest(): ties this measurement to the conv kernel sources it was taken on"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "face-recognition-pytorch_amd", "frhip", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(("igemm_halo", "igemm_nt", "common")):       # the forward / data-gradient conv kernels and what they include
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


js = {"kernel": "frhip conv implicit GEMM (halo_kernel + halo_wide_kernel + nt_kernel, bf16): forward + data-gradient launches",
      "csrc_digest": csrc_digest(),
      "hbm_bytes_per_launch": int(avg), "launches_sampled": n,
      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction)",
      "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"}
json.dump(js, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(out, js)
