#!/usr/bin/env python3
"""Per-category instruction count of the sweep loop of `warp_cost_lds_kernel<f16, f16, PROJ, VARIANCE>` (the headline kernel),
derived from the compiler's assembly (round-3 review, item 2 i).  Traces ONE trip of the loop along the common path -- all four
source views staged, boxes strictly inside the image (mode FAST) -- by following the basic blocks in layout order and taking every
conditional branch the way that path takes it: the blocks of the clipped (GEN: `v_med3_i32` clamps) and direct-tap (`global_load`)
flavours are skipped.  Usage: python scripts/dev/isa_breakdown.py > profiles/r04_warp_isa_breakdown.txt
(The packed diagnostic build `warp_cost_lds_pk_kernel` / `--pk` left the source in round 5.)"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "wild_deep_mvs_amd", "csrc")


def asm(pk: bool = False):
    if pk:
        raise SystemExit("--pk: the packed diagnostic build of the warp kernel was removed in round 5")
    flags = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    out = "/tmp/wl_isa.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-w"] + flags +
                   ["warp_cost_tiled.hip", "-o", out], cwd=CSRC, check=True)
    name = "_ZN4pscv20warp_cost_lds_kernelINS_5f16_tES1_Li0ELi0EEEvNS_8WarpArgsE"
    lines, on = [], False
    for ln in open(out):
        if ln.startswith(name + ":"):
            on = True
        if on:
            lines.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    return lines


def blocks(lines):
    """basic blocks of the inner loop: label -> list of instructions, in layout order"""
    start = max(i for i, l in enumerate(lines) if "This Inner Loop Header" in l)      # the sweep loop is the last innermost loop
    while not re.match(r"\.LBB\d+_\d+:", lines[start]):                              # (the header comment may sit on a continuation line)
        start -= 1
    # the loop's back-edge block sits right in front of the header in layout order
    back = max(i for i in range(start) if re.match(r"\.LBB\d+_\d+:", lines[i]))
    out, cur = collections.OrderedDict(), None
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"\.LBB\d+_\d+:", lines[i]) and "in Loop" not in lines[i]), len(lines))
    for l in lines[back:end]:
        m = re.match(r"(\.LBB\d+_\d+):", l) or re.match(r"; (%bb\.\d+):", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        t = l.strip()
        if cur is None or not t or t.startswith(";") or t.startswith("."):
            continue
        out[cur].append(t.split(";")[0].strip())
    return out


def classify(ins, phase):
    op = ins.split()[0]
    if op.startswith("s_"):
        return "scalar (branches, masks, loop)"
    if op.startswith("ds_read"):
        return "LDS tap reads (ds_read_b128)"
    if op.startswith("global_store"):
        return "store"
    if "dpp" in ins:
        return "DPP quad broadcasts (weights + tap address)"
    if op.startswith("v_cvt_pk_f16") or op.startswith("v_cvt_f16") or op.startswith("v_pack"):
        return "fp32 -> fp16 pack"
    if phase == "view":
        if op in ("v_add_u32_e32", "v_add_u32_e64", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32"):
            return "tap address arithmetic"
        if re.match(r"v_(pk_)?(mul|fma|fmac|add)_f32", op):
            return "blend + the two sums (reference arithmetic)"
    if phase == "final" and re.match(r"v_(pk_)?(mul|fma|fmac|add|sub)_f32", op):
        return "variance expression (reference arithmetic)"
    if phase == "head":
        if op.startswith("v_mov_b64") or op.startswith("v_mov_b32"):
            return "accumulator init (sums := reference feature)"
        return "sample coordinates (ray terms -> weights, texel offset)"
    if op == "v_lshl_add_u64":
        return "output pointer"
    return "other vector ALU"


def main():
    pk = "--pk" in sys.argv
    bl = blocks(asm(pk))
    labels = list(bl)
    counts, seq = collections.Counter(), []
    views = 0
    for lab in labels:
        ins = bl[lab]
        text = " ".join(ins)
        if "global_load" in text or "v_med3_i32" in text:
            continue                         # direct-tap / clipped flavours: not on the traced path
        has_reads = any(i.startswith("ds_read") for i in ins)
        n_fp = sum(bool(re.match(r"v_(pk_)?(mul|fma|fmac|add)_f32", i.split()[0])) for i in ins)
        if any(i.startswith("global_store") for i in ins) or (views == 4 and n_fp >= 8):
            phase = "final"
        elif has_reads or any("dpp" in i for i in ins) or (n_fp >= 16 and views < 4 and "v_rcp_f32" not in text):
            phase = "view"
            if n_fp >= 16:
                views += 1
        elif "v_rcp_f32" in text or "v_floor_f32" in text or "v_mov_b64" in text:
            phase = "head"
        else:
            phase = "glue"
        if has_reads and "offset:64" not in text:
            continue                         # the GEN flavour's read block (taps addressed through broadcast DX / DY steps)
        for i in ins:
            counts[classify(i, phase)] += 1
        seq.append((lab, phase, len(ins)))
    total_v = sum(v for k, v in counts.items() if not k.startswith("scalar") and not k.startswith("LDS") and k != "store")
    print(f"warp_cost_lds_kernel<f16, f16, PROJ, VARIANCE>, {'PACKED (diagnostic)' if pk else 'scalar fp32 (product)'} build: one trip of the sweep loop")
    print("= one depth plane of a wave = 16 voxels x 4 source views (quad lane l: channels 8l..8l+7), FAST path\n")
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        share = f"{100.0 * v / total_v:5.1f} % of the vector-ALU instructions" if not (k.startswith("scalar") or k.startswith("LDS") or k == "store") else ""
        print(f"{v:5d}  {k:58s} {share}")
    ref_ops = counts["blend + the two sums (reference arithmetic)"] + counts["variance expression (reference arithmetic)"]
    print(f"\nvector-ALU instructions per trip: {total_v}  ({total_v / 16:.1f} per voxel of a wave-trip; reference arithmetic = "
          f"{ref_ops})")
    print("blocks on the traced path:", ", ".join(f"{l}[{p}:{n}]" for l, p, n in seq))


if __name__ == "__main__":
    main()
