#!/usr/bin/env python3
"""The two CU-level call sites outside the motion search, measured in the running encoder: hook bit 262144 = InterSearch::xEstimateInterResidualQT (a CU's
component TUs in one device round trip), 524288 = EncCu::addRegularCandsToPruningList (a CU's regular merge candidates in one device call).  Prints one JSON line with
calls/frame, wall microseconds per call (inside the hook, all encoder threads), frames/s next to the CPU encoder's and whether the bitstreams are identical.

  python tools/sites_measure.py [--width 1920 --height 1080 --frames 17 --threads 8]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    base = dict(w=a.width, h=a.height, frames=a.frames, threads=a.threads)
    cpu = e2e_fps.run(dict(base, mask=0))
    out = {"clip": "%dx%d x%d faster, %d threads" % (a.width, a.height, a.frames, a.threads), "cpu_fps": round(cpu["fps"], 2), "sites": {}}
    for name, mask, calls_i, units_i, ns_i in (("residual_loop", 262144, 31, 32, 36), ("merge_pruning", 524288, 34, 35, 37)):
        r = e2e_fps.run(dict(base, mask=mask))
        c = r["calls"]
        out["sites"][name] = {"mask": mask, "fps": round(r["fps"], 2), "fps_ratio": round(r["fps"] / cpu["fps"], 3), "bitstream_identical": r["md5"] == cpu["md5"],
                              "device_calls_per_frame": round(c[calls_i] / a.frames, 1), "units_per_call": round(c[units_i] / max(1, c[calls_i]), 2),
                              "us_per_call": round(c[ns_i] / 1e3 / max(1, c[calls_i]), 1), "hook_seconds_all_threads": round(c[ns_i] / 1e9, 3),
                              "pcie_MB_per_picture": r.get("pcie_MB_per_picture")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
