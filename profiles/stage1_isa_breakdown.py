#!/usr/bin/env python3
"""Static instruction breakdown of k_invert_band (the production instantiation: float32 rasters -> complex64, mono, ROLE 1) by PHASE
of the source, from the device assembly of the shipped sources (round-5 review item: 1 400 instructions per lane of stage 1 are too
many to tune blind).

    python3 profiles/stage1_isa_breakdown.py > profiles/r05_stage1_isa_breakdown.txt      (build container: hipcc only, no GPU)

How: xsw_invert_tu.hip is compiled for gfx950 with -gline-tables-only (line tables do not change the generated code) to assembly;
every instruction of the kernel's body is attributed to the innermost source line its .loc names, i.e. to the inlined function that
line belongs to, and functions / line ranges are grouped into the phases of the kernel.  Static counts: an instruction inside a loop
is counted once; the loops are listed separately with their per-trip bodies, and the dynamic estimate multiplies them by measured
trip counts (profiles/r03_stage1_counters.json: 21.8 VALU + 7.0 SALU per pixel for stage 1 + store; r04_pmc_counters_summary.json:
44.3 VALU + 17.9 SALU per pixel for the whole kernel; 9.4 passes per wave on the benchmark scene, LABBOOK section 7c)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "xsarsea_amd", "csrc")
KERNEL = "_ZN3xsw13k_invert_bandIffLb0ELb0ELi1EEEvNS_9DevTablesENS_5KArgsE"


def function_starts(path):
    """[(line, name)] of the function definitions of a source file (good enough for these headers: a line that starts a
    `__device__` / `__global__` / template function and names it before a '(')."""
    out = []
    pending = None
    for n, line in enumerate(open(path), 1):
        m = re.match(r"\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", line)
        if m and not line.strip().startswith("//"):
            out.append((n, m.group(1)))
            continue
        if re.match(r"\s*template\s*<", line):
            pending = n
            continue
        if pending:
            m = re.match(r"\s*(?:__device__|__global__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", line)
            if m:
                out.append((n, m.group(1)))
            pending = None
    return out


def marker_lines(path, markers):
    """line numbers of the first occurrence of each marker string"""
    txt = open(path).read().split("\n")
    res = {}
    for key, needle in markers.items():
        for n, line in enumerate(txt, 1):
            if needle in line:
                res[key] = n
                break
    return res


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "tu.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(REPO, "include"),
                               "-I" + CSRC, "-DXSW_PAIR=0", "-gline-tables-only", "--cuda-device-only", "-S", os.path.join(CSRC, "xsw_invert_tu.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    files, fdirs = {}, {}
    for line in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3))
            fdirs[os.path.join(m.group(2), m.group(3)) if not os.path.isabs(m.group(3)) else m.group(3)] = os.path.basename(m.group(3))
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end") or "s_endpgm" in text[i] and ".Lfunc_end" in text[i + 1])
    fstarts = {f: function_starts(os.path.join(CSRC, f)) for f in ("xsw_device.hpp", "xsw_band.hpp")}
    # the math / intrinsics headers: one-line wrappers of ocml (bitcode without line tables: its instructions carry the wrapper's line)
    libfuncs = {}
    for d, f in fdirs.items():
        if f in fstarts or not os.path.exists(d):
            continue
        lst = []
        for n, line in enumerate(open(d, errors="ignore"), 1):
            m = re.match(r"\s*(?:static\s+)?(?:inline\s+)?(?:__device__\s+|__DEVICE__\s+|__host__\s+)*(?:[A-Za-z_][A-Za-z0-9_:<>]*[\s\*&]+)+([A-Za-z_][A-Za-z0-9_]*)\s*\([^;]*\)\s*(?:\{|$)", line)
            if m and m.group(1) not in ("if", "for", "while", "return", "switch"):
                lst.append((n, m.group(1)))
        libfuncs[f] = lst
    band = os.path.join(CSRC, "xsw_band.hpp")
    dev = os.path.join(CSRC, "xsw_device.hpp")
    bm = marker_lines(band, {"tail": "if (L.tail_min) {", "bins": "const double thr_lo = P.s_co - W.band_d", "run": "if (ROLE == 1 || (ROLE == 2 && strip_walk)) {",
                             "sort": "int base = 0;", "slot": "const bool to_rec = ROLE == 1", "passes": "// ---- stage 2: band passes", "pickup": "pos = res_[64 + lane];"})
    dm = marker_lines(dev, {"gallop": "if (SEEDED && q == 0) {", "side": "// q > 0: any score seen is a valid upper bound", "jub": "const double jub = (rbest + m2)"})

    def phase_of(fname, line):
        if fname not in fstarts:
            lib = libfuncs.get(fname)
            if lib is None:
                return f"library: {fname}"
            fn = "?"
            for n, name in lib:
                if n <= line:
                    fn = name
            return f"library: {fn} ({fname})"
        fn = "?"
        for n, name in fstarts[fname]:
            if n <= line:
                fn = name
        if fname == "xsw_device.hpp":
            if fn in ("to_db", "log10_fast"):
                return "stage 1: sigma0 -> dB (to_db, log10_fast)"
            if fn == "nearest_index":
                return "stage 1: incidence bin (nearest_index)"
            if fn in ("load_pixel", "ld"):
                return "stage 1: load + classify pixel, |m|, direction (load_pixel)"
            if fn == "ray_probe":
                return "stage 1: ray probes (ray_probe: 16-byte load + two scores)"
            if fn == "co_window_lanes":
                if dm["gallop"] <= line < dm["side"]:
                    return "stage 1: first ray, seeded gallop + bisection control (co_window_lanes)"
                if dm["side"] <= line < dm["jub"]:
                    return "stage 1: side rays control (co_window_lanes)"
                return "stage 1: rays set-up, bound, band radius (co_window_lanes)"
            if fn in ("box_from_jub", "chunk_geom", "seg_lanes"):
                return "stage 1: window geometry (box_from_jub)"
            if fn in ("store_pixel", "angle_of_quotient", "hypot_glibc"):
                return "store: winds from the winning index (store_pixel)"
            if fn in ("vmin", "vmax", "dpp_d", "seg_min_d", "wave_min_d", "rd_lane_d", "rd_lane_i", "mul24_sv", "wave_max_i"):
                return "helpers (v_min/v_max, DPP minima, 24-bit multiplies) -- called from the passes mostly"
            return f"xsw_device.hpp: {fn}"
        if fn == "co_band_pass" or fn == "ld_co":
            return "stage 2: band passes (co_band_pass, all 11 class instantiations)"
        if fn in ("wave_tail", "list_append", "mask_mark"):
            return "tail: hand-over lists, strip masks (wave_tail)"
        if fn == "band_wave":
            if line < bm["tail"]:
                return "stage 1: band_wave prologue"
            if line < bm["bins"]:
                return "stage 1: tail cut / tail sweep decision"
            if line < bm["run"]:
                return "stage 1: threshold bins of the inverse table"
            if line < bm["sort"]:
                return "stage 1: run length along the a-priori direction, hand-over decision"
            if line < bm["slot"]:
                return "stage 1: class sort (ballots, ranks, promotions)"
            if line < bm["passes"]:
                return "stage 1: slot / record write"
            if line < bm["pickup"]:
                return "stage 2: pass dispatch loops"
            return "tail: result pick-up"
        if fn == "k_invert_band":
            return "kernel prologue (tile walk)"
        return f"xsw_band.hpp: {fn}"

    counts = collections.defaultdict(lambda: collections.Counter())
    cur = ("?", 0)
    for l in text[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        ins = l.strip()
        if not ins or ins.startswith((".", ";", "//")) or ins.endswith(":"):
            continue
        op = ins.split()[0]
        if not re.match(r"^(v_|s_|global_|flat_|buffer_|ds_|scratch_)", op):
            continue
        ph = phase_of(*cur)
        kind = ("VMEM" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "LDS" if op.startswith("ds_") else
                "SALU" if op.startswith("s_") else "VALU")
        counts[ph][kind] += 1
        if kind == "VALU":
            sub = ("f64" if re.search(r"_f64|_fma_f64", op) else "f32" if "_f32" in op else "cvt" if op.startswith("v_cvt") else
                   "int/mov/cmp/select")
            counts[ph]["VALU " + sub] += 1
    print(__doc__)
    print(f"kernel {KERNEL}\n")
    hdr = f"{'phase':<86s} {'VALU':>6s} {'f64':>5s} {'f32':>5s} {'cvt':>4s} {'int..':>6s} {'SALU':>6s} {'VMEM':>5s} {'LDS':>4s}"
    print(hdr)
    print("-" * len(hdr))
    tot = collections.Counter()
    for ph in sorted(counts, key=lambda p: (not p.startswith("kernel"), not p.startswith("stage 1"), not p.startswith("stage 2"), p)):
        c = counts[ph]
        tot.update(c)
        print(f"{ph:<86s} {c['VALU']:6d} {c['VALU f64']:5d} {c['VALU f32']:5d} {c['VALU cvt']:4d} {c['VALU int/mov/cmp/select']:6d} {c['SALU']:6d} {c['VMEM']:5d} {c['LDS']:4d}")
    print("-" * len(hdr))
    print(f"{'static total':<86s} {tot['VALU']:6d} {tot['VALU f64']:5d} {tot['VALU f32']:5d} {tot['VALU cvt']:4d} {tot['VALU int/mov/cmp/select']:6d} {tot['SALU']:6d} {tot['VMEM']:5d} {tot['LDS']:4d}")
    s1 = collections.Counter()
    for ph, c in counts.items():
        if ph.startswith("stage 1") or (ph.startswith("library") and not any(k in ph for k in ("atomic", "__shfl", "amd_warp"))) or ph.startswith("kernel prologue"):
            s1.update(c)
    probe = counts["stage 1: ray probes (ray_probe: 16-byte load + two scores)"]
    gal = counts["stage 1: first ray, seeded gallop + bisection control (co_window_lanes)"]
    side = counts["stage 1: side rays control (co_window_lanes)"]
    print("\nDynamic estimate of stage 1 per WAVE (64 pixels), VALU instructions:")
    print(f"  straight-line part of stage 1 (everything but the ray loops, each instruction once): {s1['VALU'] - probe['VALU'] - gal['VALU'] - side['VALU']}")
    print(f"  one ray probe (ray_probe is inlined at {max(1, round(probe['VMEM']))} sites; per site): ~{probe['VALU'] / max(probe['VMEM'], 1):.0f} VALU + 1 load;"
          f" gallop / bisection control per trip of the first ray: ~{gal['VALU']} (static body incl. set-up); side rays control: ~{side['VALU']}")
    print("  measured stage 1 + store alone (-DXSW_TIMING_STAGE1_ONLY, profiles/r03_stage1_counters.json): 21.8 VALU x 64 = 1395 per wave")
    print("  => with 4 side-ray probes (2 rays x XSW_RAY_SIDE_STEPS = 2) the first ray runs (1395 - straight - 4 probes - side control) / (probe + control) trips:"
          " the wave's slowest lane sets the count (see the figure above); the rest of the kernel's 44.3 x 64 = 2835 VALU per wave is the passes")


if __name__ == "__main__":
    sys.exit(main())
