#!/usr/bin/env python3
"""Static instruction statistics of one gfx950 kernel from hipcc's -save-temps assembly: registers, scratch, and for every loop (backward branch) the instruction
mix of its body (VALU / MFMA / LDS / global / scalar).  A 64-lane VALU instruction occupies its SIMD for 4 cycles (packed and dot forms too), so
`valu * 4` bounds the cycles per iteration from below — how the fused TU kernel and the distortion lists were analysed.
usage: hipcc --offload-arch=gfx950 -O3 ... -save-temps=obj -c x.hip -o /tmp/isa/x.o ; tools/isa_stats.py /tmp/isa/x-hip-amdgcn-amd-amdhsa-gfx950.s <kernel substring>"""
import re
import sys
from collections import Counter


def category(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    return "salu"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    text = open(path).read().split("\n")
    start = None
    for i, l in enumerate(text):
        if l.endswith(":") is False and re.match(r"^(\S+):\s*(;.*)?$", l) and pat in l and not l.startswith("."):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    name = text[start].split(":")[0]
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    labels, ins = {}, []
    for l in text[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith("."):
            continue
        ins.append(s.split(";")[0].strip())
    print(name, "instructions:", len(ins))
    for key in ("num_vgpr", "num_agpr", "numbered_sgpr", "private_seg_size"):
        for l in text[end:end + 60]:
            if name + "." + key in l:
                print("  ", key, l.split(",")[-1].strip())
    print("   total mix:", dict(Counter(category(i.split()[0]) for i in ins)))
    loops = []
    for idx, i in enumerate(ins):
        p = i.split()
        if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= idx:
            loops.append((labels[p[-1]], idx, p[-1]))
    for a, b, lab in sorted(loops):
        mix = Counter(category(i.split()[0]) for i in ins[a:b + 1])
        print("   loop %-10s %6d instr  %s  >= %d VALU cycles" % (lab, b - a + 1, dict(mix), 4 * mix["valu"] + 0))


if __name__ == "__main__":
    main()
