#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU against known byte counts in this repository's access patterns (tools/calib/fetch_calib.hip).

  python tools/calib_fetch.py [--json OUT]        (on the GPU box; two rocprofv3 passes of ~3 s each, one counter per pass)

Per pattern: known bytes per dispatch, the counter's average per dispatch (FETCH_SIZE / WRITE_SIZE are reported in KiB of 64-byte-tallied requests) and
factor = known bytes / (counter x 1024).  bench.py applies the factor of the pattern that matches a kernel's dominant accesses instead of a literal x2
(rows16 for the per-lane row gathers of the distortion / stage / window kernels, store8 for Distortion results)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_round as P  # noqa: E402

BIN = os.path.join(ROOT, "tools", "calib", "fetch_calib")
PATTERN_NOTE = {"stream16": "16 B per lane, consecutive, aligned", "stream16_odd": "16 B per lane, consecutive, 2-byte aligned", "rows16": "eight-lane teams: 16 B of a different plane row per lane (2-byte aligned)",
                "rec8": "8 B per lane, consecutive", "store8": "8 B stores per lane, consecutive", "store16": "16 B stores per lane, consecutive"}


def run_pass(counter, timeout=120):
    out = "/tmp/vvhip_calib_%d_%s" % (os.getpid(), counter)
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc", counter, "-d", out, "--", BIN], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 --pmc %s: rc %d: %s" % (counter, r.returncode, r.stdout[-300:]))
    known = {l.split()[1]: int(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("KNOWN ")}
    rows = P.counter_table(P.find_db(out))
    shutil.rmtree(out, ignore_errors=True)
    per = {}
    for k, c, n, s, av in rows:
        name = k.split("(")[0].replace("void ", "").strip()
        if c == counter and name in known:
            per[name] = av
    return known, per


def calibrate():
    if not os.path.exists(BIN):
        raise RuntimeError("tools/calib/fetch_calib is not built (__graft_entry__.build())")
    res = {"unit": "factor = known bytes / (counter x 1024 B); 2.0 = the counter tallies 128-byte requests as 64 B", "patterns": {}}
    known, fetch = run_pass("FETCH_SIZE")
    _, write = run_pass("WRITE_SIZE")
    for name in known:
        cnt = (write if name.startswith("store") else fetch).get(name)
        res["patterns"][name] = {"counter": "WRITE_SIZE" if name.startswith("store") else "FETCH_SIZE", "known_bytes": known[name], "counter_avg_per_dispatch": cnt,
                                 "factor": (known[name] / (cnt * 1024.0)) if cnt else None, "pattern": PATTERN_NOTE.get(name, "")}
    return res


if __name__ == "__main__":
    r = calibrate()
    print(json.dumps(r, indent=1))
    if "--json" in sys.argv:
        json.dump(r, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
