#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code objects in libpscv.so (VGPRs + AGPRs, SGPRs, static LDS, scratch) from the code-object
metadata notes, and what they allow per CU: waves per SIMD by registers (MI355X_MICROARCH.md: 512 per SIMD lane, granule 8).
Usage: python scripts/dev/kernel_resources.py [lib] [substring ...]"""
import os, re, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lint_isa

def main():
    lib_path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "wild_deep_mvs_amd", "libpscv.so")
    pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
    od = lint_isa.objdump()
    readelf = os.path.join(os.path.dirname(od), "llvm-readelf")
    filt = shutil.which("c++filt") or "c++filt"
    rows = []
    with tempfile.TemporaryDirectory() as td:
        lib = os.path.join(td, "lib.so")
        shutil.copy(lib_path, lib)
        subprocess.run([od, "--offloading", lib], cwd=td, capture_output=True, check=True)
        for f in sorted(x for x in os.listdir(td) if "amdgcn" in x):
            txt = subprocess.run([readelf, "--notes", os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
    names = subprocess.run([filt], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds_static':>10s} {'scratch':>8s} {'wg':>5s} {'waves/SIMD':>10s}  kernel")
    for r, n in zip(rows, names):
        n = n.replace("void pscv::", "").replace("pscv::", "")
        n = n[:n.find("(")] if "(" in n else n
        if pats and not any(p in n for p in pats):
            continue
        try:
            alloc = -(-int(r[1]) // 8) * 8
            wps = min(8, 512 // alloc)
        except Exception:
            wps = "?"
        print(f"{r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>10s} {r[5]:>8s} {r[6]:>5s} {str(wps):>10s}  {n[:110]}")

main()
