"""Builds libxsw.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "xsw.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("xsw.hip", "xsw_device.hpp", "xsw_exhaustive.hpp", "xsw_gmf.hpp", "xsw_nesz.hpp", "xsw_lutbuild.hpp")] + [
    os.path.join(REPO, "include", "xsw.h")]
LIB = os.environ.get("XSW_LIB") or os.path.join(HERE, "libxsw.so")  # XSW_LIB: experiment builds only
ARCH = "gfx950"


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libxsw.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile csrc/xsw.hip -> libxsw.so for gfx950.  -ffp-contract=off: the kernels decide exact
    float64 orderings; FMAs appear only where written explicitly."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(HERE, "csrc"),
           "-o", LIB, SRC, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
