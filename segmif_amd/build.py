"""Builds libsegmif_hip.so (gfx950) in-tree with hipcc.  No torch cpp_extension, no hipify:
the sources are native HIP and the library has a plain C ABI (include/segmif_hip.h)."""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsegmif_hip.so")
SOURCES = ["igemm.hip", "conv3x3.hip", "conv3x3_split.hip", "conv3x3_planes.hip", "gemm_split.hip", "gemm_pairs.hip", "wgrad.hip", "backward.hip", "attention.hip", "attention_split.hip", "attention_bwd.hip", "rowops.hip", "linattn.hip", "crosspath.hip", "metrics.hip", "losses.hip", "common.hip", "comm.hip", "mixffn.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).replace(ROOT, ".").encode())  # (not the checkout's absolute path: the same tree on the GPU box is the same build)
    for f in SOURCES + ["igemm_common.h", "device_once.h", "planes16.h", os.path.join(ROOT, "include", "segmif_hip.h")]:
        p = f if os.path.isabs(f) else os.path.join(CSRC, f)
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"[segmif_amd] built {LIB} for {ARCH}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
