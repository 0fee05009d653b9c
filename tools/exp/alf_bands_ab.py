#!/usr/bin/env python3
"""A/B of the ALF statistics in bands on the real encoder: frames/s of 1080p / 4K x 65 at T threads with
  cpu        mask 0
  mctf       mask 144                                   (MCTF search + filter only)
  whole      mask 144 + 8192, $VVHIP_ALF_BANDS=0        (the picture in one call inside deriveFilter), gate off ($VVHIP_ALF_MIN_CTUS_PER_THREAD=0)
  bands      mask 144 + 8192, bands                      gate off
md5 must be equal everywhere.  Usage (GPU box): python tools/exp/alf_bands_ab.py [--width 1920 --height 1080 --frames 65 --threads 8 --reps 2]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    legs = [("cpu", 0, {}), ("mctf", 144, {}), ("whole", 144 + 8192, {"VVHIP_ALF_BANDS": "0", "VVHIP_ALF_MIN_CTUS_PER_THREAD": "0"}),
            ("bands", 144 + 8192, {"VVHIP_ALF_MIN_CTUS_PER_THREAD": "0"})]
    out = {}
    for rep in range(a.reps):
        for name, mask, env in legs:
            e = dict(os.environ); e.update(env)
            r = e2e_fps.run(dict(w=a.width, h=a.height, frames=a.frames, threads=a.threads, mask=mask), env=e)
            o = out.setdefault(name, {"fps": [], "md5": set(), "calls": None})
            o["fps"].append(round(r["fps"], 2)); o["md5"].add(r["md5"])
            if r["calls"]:
                o["calls"] = {"alf_pictures": r["calls"][16], "bands": r["calls"][39], "band_pictures": r["calls"][40],
                              "row_ms_per_picture": round(r["calls"][41] / 1e6 / max(1, r["calls"][16]), 3), "serial_ms_per_picture": round(r["calls"][42] / 1e6 / max(1, r["calls"][16]), 3), "pcie": r.get("pcie_MB_per_picture")}
            print(name, rep, round(r["fps"], 2), r["md5"], flush=True)
    md5 = set().union(*[o["md5"] for o in out.values()])
    for o in out.values():
        o["md5"] = sorted(o["md5"])
    print(json.dumps({"clip": "%dx%d x %d, %d threads" % (a.width, a.height, a.frames, a.threads), "identical": len(md5) == 1, "legs": out}))


if __name__ == "__main__":
    main()
