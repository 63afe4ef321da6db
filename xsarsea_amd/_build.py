"""Builds libxsw.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "xsw.hip")          # context, LUT install, C ABI, the HBM-bound kernels
SRC_TU = os.path.join(CSRC, "xsw_invert_tu.hip")  # the search kernels of one (input dtype, output dtype) pair: -DXSW_PAIR=0..3
DEPS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [os.path.join(REPO, "include", "xsw.h")]
LIB = os.environ.get("XSW_LIB") or os.path.join(HERE, "libxsw.so")  # XSW_LIB: experiment builds only
OBJDIR = os.path.join(REPO, "build", "obj")
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
    """Compile csrc/*.hip -> libxsw.so for gfx950: five translation units side by side (xsw.hip + the search kernels of each
    dtype pair), then one link.  -ffp-contract=off: the kernels decide exact float64 orderings; FMAs appear only where
    written explicitly."""
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    extra = os.environ.get("XSW_EXTRA_FLAGS", "").split()  # experiment builds only
    tag = os.path.basename(LIB).replace(".", "_")
    os.makedirs(OBJDIR, exist_ok=True)
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
              "-I" + os.path.join(REPO, "include"), "-I" + CSRC] + extra
    jobs = [(SRC, os.path.join(OBJDIR, f"{tag}_main.o"), [])]
    jobs += [(SRC_TU, os.path.join(OBJDIR, f"{tag}_pair{k}.o"), [f"-DXSW_PAIR={k}"]) for k in range(4)]
    procs = []
    for src, obj, defs in jobs:
        cmd = common + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + [obj for _, obj, _ in jobs] + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


def _code_objects(lib, td):
    """The gfx950 code objects inside libxsw.so: its .hip_fatbin section holds one offload bundle per translation unit."""
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "unused.so")])
    with open(fat, "rb") as f:
        blob = f.read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m for m in range(len(blob)) if blob.startswith(magic, m)] if blob.count(magic) < 64 else []
    out = []
    for k, st in enumerate(starts):
        piece, co = os.path.join(td, f"bundle{k}.bin"), os.path.join(td, f"dev{k}.co")
        with open(piece, "wb") as f:
            f.write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o",
                               f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={piece}", f"--output={co}"])
        out.append(co)
    return out


def kernel_resources(lib=None):
    """Per-kernel register / LDS / scratch figures read from the gfx950 code objects inside libxsw.so (the
    `amdhsa.kernels` metadata note: .vgpr_count, .agpr_count, .sgpr_count, .group_segment_fixed_size,
    .private_segment_fixed_size), plus the waves per SIMD the VGPR count allows (512 VGPRs per SIMD lane, allocation
    granule 8, at most 8 waves).  Returns a list of dicts sorted by kernel name."""
    import re
    import tempfile
    lib = lib or LIB
    llvm = "/opt/rocm/lib/llvm/bin"
    out = []
    with tempfile.TemporaryDirectory() as td:
        for co in _code_objects(lib, td):
            notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
            raw = re.findall(r"\.name:\s+(\S+)", notes)
            if not raw:
                continue
            try:
                names = subprocess.check_output([f"{llvm}/llvm-cxxfilt"] + raw, text=True).splitlines()
            except (OSError, subprocess.CalledProcessError):
                names = raw
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
