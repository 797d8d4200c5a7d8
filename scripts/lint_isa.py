#!/usr/bin/env python3
"""ISA lint of libpscv.so (round 4): no packed fp32 instruction whose LOW result lane reads the HIGH half of its SECOND source pair.

Finding (self-checking reproducer scripts/ubench/lds_pk_overlap.hip `opsel_victim` + scripts/dev/opsel_probe.py):
on this MI355X pool `v_pk_mul_f32` / `v_pk_add_f32` / `v_pk_fma_f32` with the `op_sel` bit of SRC1 set (`op_sel:[0,1]`, `op_sel:[0,1,0]`:
the low 32-bit result lane takes the upper register of the second source pair) return a wrong LOW result in lanes 48-63 of a wave while
MFMA waves of another kernel (the engine's conv0) run on a second stream -- millions of wrong results per second; 0 when launched
alone.  Measured exact under the same overlap: the `op_sel` bit of src0 or src2, every `op_sel_hi` form (the usual low-half broadcast
`op_sel_hi:[1,0,1]`), `v_pk_mov_b32 op_sel:[1,0]`.  The SLP vectorizer emits the failing form freely (two scalars packed into one
register pair, the second one selected with op_sel), so the build is checked: every gfx950 code object embedded in the library is
disassembled and any such instruction fails the lint.  (Positive control: scripts/ubench/liblpo.so, the reproducer, contains the form.)

Usage: python scripts/lint_isa.py [path/to/libpscv.so]
Exit code 1 = the form was found; 2 = the lint could not run (no llvm-objdump, extraction failed) -- `make` removes the library only
on 1; tests/test_isa_lint_cpu.py enforces the check either way."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

BAD = re.compile(r"\b(v_pk_(?:fma|mul|add|min|max)_f32)\b.*\bop_sel:\[([01]),([01])(?:,([01]))?\]")


def objdump():
    """llvm-objdump next to $HIPCC / hipcc on PATH, in the ROCm tree, or on PATH; None if there is none."""
    cands = []
    for hipcc in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if hipcc:
            root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
            cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    cands += ["/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


def findings(lib_path: str):
    """[(kernel symbol, instruction text)] of every offending instruction, and the number of packed instructions seen."""
    out, seen = [], 0
    od = objdump()
    if od is None:
        raise RuntimeError("llvm-objdump not found (looked next to hipcc, in /opt/rocm and on PATH)")
    with tempfile.TemporaryDirectory() as td:
        lib = os.path.join(td, os.path.basename(lib_path))
        shutil.copy(lib_path, lib)
        subprocess.run([od, "--offloading", lib], cwd=td, capture_output=True, check=True)
        objs = [f for f in os.listdir(td) if "amdgcn" in f]
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {lib_path}")
        for f in sorted(objs):
            dis = subprocess.run([od, "-d", "--no-show-raw-insn", os.path.join(td, f)],
                                 capture_output=True, text=True, check=True).stdout
            sym = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    sym = m.group(1)
                    continue
                if "v_pk_" not in line:
                    continue
                seen += 1
                b = BAD.search(line)
                if b and b.group(3) == "1":      # op_sel bit of src1
                    out.append((sym, line.strip().split("//")[0].strip()))
    return out, seen


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "wild_deep_mvs_amd", "libpscv.so")
    try:
        bad, seen = findings(lib)
    except Exception as e:      # could not run is not "found the pattern"
        print(f"[lint_isa] could not run on {lib}: {type(e).__name__}: {e}")
        return 2
    if bad:
        per = {}
        for sym, ins in bad:
            per.setdefault(sym, []).append(ins)
        print(f"[lint_isa] {len(bad)} packed fp32 instruction(s) whose low lane reads the HIGH half of src1 (of {seen} packed instructions) in {lib}:")
        for sym, lst in sorted(per.items()):
            print(f"  {sym}: {len(lst)}  e.g. {lst[0]}")
        print("  -> compile that file with -fno-slp-vectorize (or without packed fp32: $(NOPK) in csrc/Makefile), see DESIGN.md section 7")
        return 1
    print(f"[lint_isa] ok: {seen} packed instructions, no packed fp32 arithmetic reads the high half of src1 for its low result lane ({os.path.basename(lib)})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
