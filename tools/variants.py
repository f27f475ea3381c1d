"""Build named variants of ONE translation unit with extra -D flags (linked against the other objects of libfrhip.so):
    python tools/variants.py bn "u4:-DEW_UNROLL=4" "r16:-DEW_ROWS=16"
-> face-recognition-pytorch_amd/frhip/build/var/libfrhip_<unit>_<name>.so ; run anything with FRHIP_LIB_PATH=<that file>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
from frhip import build as fb

unit = sys.argv[1]
out = os.path.join(fb.HERE, "build", "var")
os.makedirs(out, exist_ok=True)
fb.build()
objs = [os.path.join(fb.HERE, "build", s.replace(".hip", ".o")) for s in fb.SOURCES if s != unit + ".hip"]
procs = []
for item in sys.argv[2:]:
    name, flags = item.split(":")
    o = os.path.join(out, "%s_%s.o" % (unit, name))
    cmd = [fb.HIPCC] + fb.FLAGS + [f for f in flags.split(",") if f] + ["-c", os.path.join(fb.CSRC, unit + ".hip"), "-o", o]
    procs.append((name, o, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for name, o, p in procs:
    txt, _ = p.communicate()
    if p.returncode:
        raise RuntimeError(txt.decode())
    lib = os.path.join(out, "libfrhip_%s_%s.so" % (unit, name))
    subprocess.check_call([fb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + objs)
    print(lib)
