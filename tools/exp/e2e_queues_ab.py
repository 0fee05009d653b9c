#!/usr/bin/env python3
"""A/B on the real encoder (1080p x 65, T threads): CPU kernels vs --SIMD=HIP's production mask under different $GPU_MAX_HW_QUEUES (the encoder's worker contexts create one HIP
stream each) and with the ALF statistics in bands on / off at this size.  python tools/exp/e2e_queues_ab.py [--threads 8 --reps 5]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    legs = [("cpu", 0, {}), ("hip q4", 8336, {"GPU_MAX_HW_QUEUES": "4"}), ("hip q8", 8336, {"GPU_MAX_HW_QUEUES": "8"}), ("hip q10", 8336, {"GPU_MAX_HW_QUEUES": "10"}),
            ("hip q8 alf gate 100", 8336, {"GPU_MAX_HW_QUEUES": "8", "VVHIP_ALF_MIN_CTUS_PER_THREAD": "100"}), ("mctf only q8", 144, {"GPU_MAX_HW_QUEUES": "8"})]
    out = {n: [] for n, _, _ in legs}
    md5 = set()
    for rep in range(a.reps):
        for name, mask, env in legs:
            e = dict(os.environ, VVHIP_E2E_KEEP_QUEUES="1")
            e.pop("GPU_MAX_HW_QUEUES", None)
            e.update(env)
            r = e2e_fps.run(dict(w=1920, h=1080, frames=65, threads=a.threads, mask=mask), env=e)
            out[name].append(round(r["fps"], 2)); md5.add(r["md5"])
        print(rep, {n: v[-1] for n, v in out.items()}, os.getloadavg(), flush=True)
    print(json.dumps({"threads": a.threads, "identical": len(md5) == 1, "median": {n: statistics.median(v) for n, v in out.items()}, "runs": out}))


if __name__ == "__main__":
    main()
