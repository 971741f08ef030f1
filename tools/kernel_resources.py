#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of the gfx950 code objects in csrc/build/*.o.

    python tools/kernel_resources.py [-o profiles/rNN_kernel_resources.txt]

Every .o that hipcc wrote carries a clang offload bundle in its .hip_fatbin section; the gfx950 code object is
taken out of it (llvm-objcopy, clang-offload-bundler) and its AMDGPU metadata note is read with llvm-readelf.
The numbers are the compiler's own: what DESIGN.md and the reviews quote comes from this table.

waves/SIMD: 512 VGPRs per lane and SIMD (arch + acc registers, allocated in blocks of 8), at most 8 waves;
a kernel with a launch bound of one wave per workgroup is additionally limited by its LDS (160 KB per CU).
No GPU is needed.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = p.stdout.split("\n") if p.returncode == 0 else names
    except OSError:  # no demangler: the mangled names are still unambiguous
        out = names
    return [short(o) for o in out[:len(names)]]


def short(name):
    """k_rs_pass<16>(unsigned int const*, ...) -> k_rs_pass<16>"""
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, c in enumerate(name):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            return name[:i]
    return name


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "k.co")
    p = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True)
    if p.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []
    p = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                        "--input=" + fat, "--output=" + co], capture_output=True, text=True)
    if p.returncode != 0:
        return []
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    cur = None
    for line in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S.*)?$", line)
        if not m:
            continue
        key, val = m.group(1), (m.group(2) or "").strip()
        # a kernel's block starts with .agpr_count (keys are sorted inside a block) or .args
        if key in ("agpr_count", "args") and (cur is None or "name" in cur and key in cur):
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["name"] = val
        elif key in FIELDS:
            cur[key] = int(val)
        elif key == "args":
            cur["args"] = 1
    os.remove(fat)
    os.remove(co)
    return [k for k in out if "name" in k and "vgpr_count" in k]


def waves_per_simd(k):
    regs = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output")
    ap.add_argument("--build-dir", default=os.path.join(ROOT, "uniformgrid-raytracing_amd", "csrc", "build"))
    a = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(a.build_dir, "*.o"))):
            ks = kernels_of(obj, tmp)
            names = demangle([k["name"] for k in ks])
            for k, n in zip(ks, names):
                if n.startswith("rocprim::") or n.startswith("void rocprim::"):
                    continue  # the library's kernels behind option sort_library
                rows.append((os.path.basename(obj)[:-2], n, k))
    if not rows:
        sys.exit("no gfx950 code objects under %s (run make -C uniformgrid-raytracing_amd/csrc first)" % a.build_dir)
    hdr = "%-14s %-58s %5s %5s %6s %5s %6s %8s %7s %6s %5s" % ("file", "kernel", "vgpr", "agpr", "vspill", "sgpr", "sspill",
                                                                 "scratchB", "ldsB", "wgsize", "w/SIMD")
    lines = ["# tools/kernel_resources.py: AMDGPU metadata of the gfx950 code objects in csrc/build (llvm-readelf --notes)",
             "# vspill/sspill = spilled vector/scalar registers, scratchB = bytes of scratch per lane, w/SIMD = waves per SIMD by registers",
             hdr]
    for f, n, k in rows:
        lines.append("%-14s %-58s %5d %5d %6d %5d %6d %8d %7d %6d %5d" % (
            f, n[:58], k.get("vgpr_count", 0), k.get("agpr_count", 0), k.get("vgpr_spill_count", 0), k.get("sgpr_count", 0),
            k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0),
            k.get("max_flat_workgroup_size", 0), waves_per_simd(k)))
    text = "\n".join(lines) + "\n"
    if a.output:
        with open(a.output, "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
