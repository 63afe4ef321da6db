"""Builds libxsw.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "xsw.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("xsw.hip", "xsw_device.hpp", "xsw_exhaustive.hpp", "xsw_gmf.hpp", "xsw_nesz.hpp", "xsw_lutbuild.hpp", "xsw_band.hpp")] + [
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
           "-o", LIB, SRC, "-Wl,-rpath,/opt/rocm/lib"] + os.environ.get("XSW_EXTRA_FLAGS", "").split()  # experiment builds only
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def kernel_resources(lib=None):
    """Per-kernel register / LDS / scratch figures read from the gfx950 code object inside libxsw.so (the
    `amdhsa.kernels` metadata note: .vgpr_count, .agpr_count, .sgpr_count, .group_segment_fixed_size,
    .private_segment_fixed_size), plus the waves per SIMD the VGPR count allows (512 VGPRs per SIMD lane, allocation
    granule 8, at most 8 waves).  Returns a list of dicts sorted by kernel name."""
    import re
    import tempfile
    lib = lib or LIB
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "unused.so")])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o",
                               f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}", f"--output={co}"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
        try:
            names = subprocess.check_output([f"{llvm}/llvm-cxxfilt"] + re.findall(r"\.name:\s+(\S+)", notes), text=True).splitlines()
        except (OSError, subprocess.CalledProcessError):
            names = re.findall(r"\.name:\s+(\S+)", notes)
    out = []
    for blk, name in zip(re.split(r"\n\s+- (?=\.agpr_count)", notes)[1:], names):
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
        vg, ag = g("vgpr_count"), g("agpr_count")
        alloc = -(-max(vg + ag, 1) // 8) * 8
        out.append(dict(kernel=name, vgpr=vg, agpr=ag, sgpr=g("sgpr_count"), lds_bytes=g("group_segment_fixed_size"),
                        scratch_bytes=g("private_segment_fixed_size"), waves_per_simd_by_vgpr=min(8, 512 // alloc)))
    return sorted(out, key=lambda d: d["kernel"])


def write_kernel_resources(path):
    rows = kernel_resources()
    with open(path, "w") as f:
        f.write("# per-kernel resources of libxsw.so's gfx950 code object (xsarsea_amd/_build.py: kernel_resources)\n")
        f.write("vgpr\tagpr\tsgpr\tlds_B\tscratch_B\twaves/SIMD(vgpr)\tkernel\n")
        for r in rows:
            f.write(f"{r['vgpr']}\t{r['agpr']}\t{r['sgpr']}\t{r['lds_bytes']}\t{r['scratch_bytes']}\t{r['waves_per_simd_by_vgpr']}\t{r['kernel']}\n")
    return rows


def code_object_sha256(lib=None):
    """SHA-256 of the .hip_fatbin section (the device code) of libxsw.so: what a counter profile was measured on."""
    import hashlib
    import tempfile
    lib = lib or LIB
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib,
                               os.path.join(td, "unused.so")])
        with open(fat, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--resources":
        build()
        write_kernel_resources(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--code-sha":
        print(code_object_sha256())
    else:
        print(build(force=True, verbose=True))
