"""Build libfrhip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libfrhip.so")
SOURCES = ["misc.hip", "bn.hip", "stem.hip", "stem_fused.hip", "stem_algebra.hip", "igemm_nt.hip", "igemm_halo.hip", "igemm_tn.hip", "igemm_fp8.hip", "head.hip", "pfc_sample.hip", "margin.hip", "winattn.hip", "winattn_mfma.hip", "cpb.hip", "optim.hip", "augment.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
# experiment builds (tools/ab_libs.sh): FRHIP_VARIANT=<name> FRHIP_CXXFLAGS="-DX=1 ..." python -m frhip.build
#   -> build/var/libfrhip_<name>.so from its own object directory; the in-tree libfrhip.so is untouched
VARIANT = os.environ.get("FRHIP_VARIANT", "")
if VARIANT:
    FLAGS = FLAGS + os.environ.get("FRHIP_CXXFLAGS", "").split()
    LIB = os.path.join(HERE, "build", "var", "libfrhip_%s.so" % VARIANT)


def _digest(paths):
    """content hash of the inputs of one object (source, every header, the flags): a snapshot copy may scramble mtimes"""
    h = hashlib.sha1(" ".join(FLAGS).replace(ROOT, "").encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps):
    stamp = target + ".sha1"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libfrhip.so.  Returns the library path."""
    objdir = os.path.join(HERE, "build", "var", VARIANT) if VARIANT else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "frhip.h"))
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, o, [s] + headers, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, o, deps, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        with open(o + ".sha1", "w") as f:
            f.write(_digest(deps))
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
