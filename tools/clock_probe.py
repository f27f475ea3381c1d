"""In-kernel clock of the halo conv kernels (MI355X_MICROARCH "DVFS give-back" item 6): a diagnostic build of igemm_halo.hip
(-DFRHIP_CLOCK_STAMP=1) stamps s_memtime / s_memrealtime around the main loop of every workgroup; clock = ticks / real ticks x 100 MHz,
median over the workgroups of the last launch after >= 2 s of back-to-back launches.  Random bf16 operands and all-zero operands.

build:  python tools/clock_probe.py build     (here; cross-compiles build/abl/libfrhip_igemm_halo_clock.so)
run:    python tools/clock_probe.py run       (GPU box)
"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "face-recognition-pytorch_amd")
sys.path[:0] = [ROOT, PKG]
from frhip import build as fb

ABL = os.path.join(fb.HERE, "build", "abl")
LIB = os.path.join(ABL, "libfrhip_igemm_halo_clock.so")


def build():
    os.makedirs(ABL, exist_ok=True)
    fb.build()
    objs = [os.path.join(fb.HERE, "build", s.replace(".hip", ".o")) for s in fb.SOURCES if s != "igemm_halo.hip"]
    o = os.path.join(ABL, "igemm_halo_clock.o")
    subprocess.check_call([fb.HIPCC] + fb.FLAGS + ["-DFRHIP_CLOCK_STAMP=1", "-c", os.path.join(fb.CSRC, "igemm_halo.hip"), "-o", o])
    subprocess.check_call([fb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, o] + objs)
    print("built", LIB)


def run():
    import numpy as np
    import torch
    L = ctypes.CDLL(LIB)
    B = 512
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    secs = float(os.environ.get("CLOCK_SECS", "2.0"))
    print("# in-kernel clock of the halo conv main loop, B = %d, %.1f s of back-to-back launches per line" % (B, secs))
    print("# dir    h   c   data     us/launch  TFLOP/s   clock GHz (median / p10 / p90 over workgroups)   MFMA-peak at that clock")
    for (h, c) in [(56, 64), (28, 128), (14, 256), (7, 512)]:
        for direction in ("fwd", "dgrad"):
            for data in ("random", "zeros"):
                if data == "random":
                    x = torch.randn(B, h, h, c, device="cuda").bfloat16()
                    w = (torch.randn(c, 3, 3, c, device="cuda") * 0.05).bfloat16()
                else:
                    x = torch.zeros(B, h, h, c, device="cuda", dtype=torch.bfloat16)
                    w = torch.zeros(c, 3, 3, c, device="cuda", dtype=torch.bfloat16)
                y = torch.empty(B, h, h, c, device="cuda", dtype=torch.bfloat16)

                def call():
                    if direction == "fwd":
                        rc = L.frhip_conv_fwd(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                              None, B, h, h, c, c, 3, 3, 1, 1, stream)
                    else:
                        rc = L.frhip_conv_dgrad(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                                None, B, h, h, c, c, 3, 3, 1, 1, stream)
                    assert rc == 0, rc
                call()
                torch.cuda.synchronize()
                t_end = time.time() + secs
                n = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                while time.time() < t_end:
                    for _ in range(50):
                        call()
                    torch.cuda.synchronize()
                e0.record()
                for _ in range(50):
                    call()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / 50
                wgs = min(8192, (B * h * h + 255) // 256 * max(1, c // (128 if (direction == "dgrad" and c >= 128) else 64)))
                buf = (ctypes.c_ulonglong * (2 * wgs))()
                assert L.frhip_dbg_clock_read(buf, wgs) == 0
                a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
                a = a[a[:, 1] > 0]
                ghz = a[:, 0] / a[:, 1] * 0.1
                tf = 2.0 * B * h * h * c * c * 9 / us / 1e6
                med = float(np.median(ghz))
                print("%-6s %3d %4d  %-7s %9.1f %9.1f   %.3f / %.3f / %.3f   %7.1f TFLOP/s  (achieved = %.3f of it)" % (
                    direction, h, c, data, us, tf, med, np.percentile(ghz, 10), np.percentile(ghz, 90), 2516.6 * med / 2.4,
                    tf / (2516.6 * med / 2.4)), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
