"""Timing-only ablation of the halo conv kernel (GPU box only; results of the ablated variants are wrong by design).

build:  python tools/ablate.py build      (here, cross-compiles variants into frhip/build/abl/)
run:    python tools/ablate.py run        (on the GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "face-recognition-pytorch_amd")
sys.path[:0] = [ROOT, PKG]
from frhip import build as fb

ABL = os.path.join(fb.HERE, "build", "abl")
VARIANTS = {"full": 0, "nobar": 1, "nomfma": 2, "nolds": 4, "nodma": 8, "noepi": 16, "mfma_only": 1 | 4 | 8 | 16,
            "nolds_noepi": 4 | 16, "lds_only": 1 | 2 | 8 | 16}
UNIT = os.environ.get("ABL_UNIT", "igemm_halo")
# ABL_DEFS="name1:-DX=1,-DY=2;name2:-DX=3": extra named variants built with the given defines (no ablation bits)
EXTRA = {}
for item in filter(None, os.environ.get("ABL_DEFS", "").split(";")):
    nm, flags = item.split(":")
    EXTRA[nm] = flags.split(",")
if EXTRA:
    VARIANTS = {nm: 0 for nm in EXTRA}
if os.environ.get("ABL_ONLY"):          # ABL_ONLY=full,nodma,noepi: subset of the variants
    VARIANTS = {nm: VARIANTS[nm] for nm in os.environ["ABL_ONLY"].split(",")}


def build():
    os.makedirs(ABL, exist_ok=True)
    fb.build()
    objs = [os.path.join(fb.HERE, "build", s.replace(".hip", ".o")) for s in fb.SOURCES if s != UNIT + ".hip"]
    procs = []
    for name, bits in VARIANTS.items():
        o = os.path.join(ABL, "%s_%s.o" % (UNIT, name))
        cmd = [fb.HIPCC] + fb.FLAGS + ["-DFRHIP_ABL=%d" % bits] + EXTRA.get(name, []) + ["-c", os.path.join(fb.CSRC, UNIT + ".hip"), "-o", o]
        procs.append((name, o, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, o, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(out.decode())
        lib = os.path.join(ABL, "libfrhip_%s_%s.so" % (UNIT, name))
        subprocess.check_call([fb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + objs)
        print("built", lib)


def run():
    import torch
    B = int(os.environ.get("B", "512"))
    shapes = [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]
    libs = {}
    for name in VARIANTS:
        libs[name] = ctypes.CDLL(os.path.join(ABL, "libfrhip_%s_%s.so" % (UNIT, name)))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(160 << 20, dtype=torch.uint8, device="cuda")
    for (h, c, k) in shapes:
        x = torch.randn(B, h, h, c, device="cuda").bfloat16()
        w = (torch.randn(k, 3, 3, c, device="cuda") * 0.05).bfloat16()
        y = torch.empty(B, h, h, k, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(k, 3, 3, c, device="cuda")
        line = "h=%2d c=%3d |" % (h, c)
        for name, L in libs.items():
            def call():
                if UNIT == "igemm_tn":
                    rc = L.frhip_conv_wgrad(0, ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dw.data_ptr()),
                                            B, h, h, c, k, 3, 3, 1, 1, 0, ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream)
                else:
                    rc = L.frhip_conv_fwd(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                          None, B, h, h, c, k, 3, 3, 1, 1, stream)
                assert rc == 0
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            line += " %s %6.1fus |" % (name, us)
        print(line, flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
