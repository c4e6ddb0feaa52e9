"""Build libnebulae_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnebulae_hip.so")
SOURCES = ["api.hip", "svgf.hip", "gi.hip", "gi_build.hip", "gi_sun_table.hip", "raysort.hip", "strips.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-fno-slp-vectorize"]  # keeps LLVM from forming v_pk_*_f32 pairs: they issue at half rate on gfx950 (tools/ubench_valu2.hip)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG_DIR, "..", "include", "nebulae_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("NEB_EXTRA_HIPCC_FLAGS", "").split()  # tuning builds only (e.g. -DNEB_SORT_BITS=16)
    cmd = [_hipcc()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
