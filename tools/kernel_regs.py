#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of a gfx950 assembly file (hipcc -save-temps=obj) from its .amdhsa metadata.
usage: tools/kernel_regs.py x-hip-amdgcn-amd-amdhsa-gfx950.s [name substring]    (waves per SIMD = 512 / (vgpr + agpr rounded up to 8), capped at 8)"""
import re
import subprocess
import sys


def kernels(path):
    txt = open(path).read()
    md = txt[txt.index("amdhsa.kernels:"):]
    out = []
    for blk in re.split(r"\n  - ", md)[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        name = g("name")
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        v, a = int(g("vgpr_count")), int(g("agpr_count"))
        tot = ((v + 7) // 8) * 8 + ((a + 7) // 8) * 8
        out.append(dict(name=name.replace("(anonymous namespace)::", ""), vgpr=v, agpr=a, sgpr=int(g("sgpr_count")), spill=int(g("vgpr_spill_count")), scratch=int(g("private_segment_fixed_size")),
                        lds=int(g("group_segment_fixed_size")), waves=min(8, 512 // max(tot, 1))))
    return out


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-100s %5s %5s %5s %6s %8s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "lds", "waves"))
    for k in kernels(sys.argv[1]):
        if sub in k["name"]:
            print("%-100s %5d %5d %5d %6d %8d %7d %6d" % (k["name"][:100], k["vgpr"], k["agpr"], k["sgpr"], k["spill"], k["scratch"], k["lds"], k["waves"]))
