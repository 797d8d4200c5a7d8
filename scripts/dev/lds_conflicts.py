#!/usr/bin/env python3
"""Dev: LDS cycles of the B-fragment `ds_read_b128`s of the brick kernels under the gfx950 lane-group model
(/opt/skills/guides/MI355X_MICROARCH.md, LDS table: a wave's ds_read_b128 = four groups of 16 lanes, 64 banks of 4 B,
N distinct addresses on a bank within a group = N cycles), summed over a layer's k-steps, for candidate voxel strides."""
import sys

G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]


def cycles(addr):
    tot = 0
    for grp in GROUPS:
        banks = {}
        for l in grp:
            a = addr(l) // 4
            for d in range(4):
                banks.setdefault((a + d) % 64, set()).add(a + d)
        tot += max(len(v) for v in banks.values())
    return tot


def layer(kind, cin, vs, bh, bw, swz=None):
    """kind: 0 stride 1, 1 stride 2 (dense, 27 taps): average cycles per B read over the k-steps."""
    sxy = 2 if kind == 1 else 1
    nsteps = (27 * cin + 31) // 32
    tot = 0
    for s in range(nsteps):
        def addr(l):
            n, g = l & 15, l >> 4
            kk0 = s * 32 + g * 8
            tap = min(kk0 // cin, 26)
            kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
            vox = (kd * bh + kh) * bw + kw + n * sxy
            ch = (kk0 % cin) // 8
            if swz:
                return swz(vox, ch)
            return vox * vs + ch * 16
        tot += cycles(addr)
    return tot / nsteps


if __name__ == "__main__":
    for kind, bh, bw, name in ((0, 6, 18, "stride 1, brick 6 x 18"), (1, 5, 33, "stride 2, brick 5 x 33"), (0, 3, 18, "stride 1, brick 3 x 18")):
        for cin in (8, 16, 32, 64):
            row = []
            for vs in range(cin * 2, cin * 2 + 80, 16):
                row.append(f"{vs}:{layer(kind, cin, vs, bh, bw):.2f}")
            print(f"{name:26s} C_in {cin:2d}  " + "  ".join(row))
