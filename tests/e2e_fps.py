#!/usr/bin/env python3
"""End-to-end frames/sec of the REAL reference encoder (oracle/_ref, compiled from /root/reference) on a BASELINE-config-2 style clip,
CPU kernels vs the same encoder with the MCTF stage on the MI355X (hook mask 16 = whole-picture motion estimation, + 128 = bilateral filter), same host, same
threads, bitstream md5 compared (SURVEY §8d "Metric").  Test infrastructure: prints one JSON line; run it through gpurun.

  python tests/e2e_fps.py [--width 1920 --height 1080 --frames 17 --threads 8 --masks 0,16]
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def synth_clip(width, height, frames, seed=1080):
    """SURVEY §8d config-2 generator: textured base, global pan (3,1) px/frame, 128x128 object moving (7,2) px/frame, per-frame noise;
    10-bit, chroma = affine of subsampled luma.  The clip is cached under /tmp (several runs of a comparison encode the same clip)."""
    cache = os.path.join("/tmp", "vvhip_clip_%dx%d_%d_%d.npz" % (width, height, frames, seed))
    if os.path.exists(cache):
        try:
            d = np.load(cache)
            return d["y"], d["u"], d["v"]
        except Exception:
            pass
    y, u, v = _synth_clip(width, height, frames, seed)
    try:
        tmp = cache + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, y=y, u=u, v=v)
        os.replace(tmp, cache)
    except Exception:
        pass
    return y, u, v


def synth_clip_chunk(width, height, first, frames, seed=1080):
    """frames first .. first + frames - 1 of ONE long sequence of the same generator (pan and object keep moving; the per-frame noise is seeded per frame, so any chunk can be
    produced on its own): the GOP chunks the N encoder instances of bench.py --gpus N encode"""
    cache = os.path.join("/tmp", "vvhip_chunk_%dx%d_%d_%d_%d.npz" % (width, height, first, frames, seed))
    if os.path.exists(cache):
        try:
            d = np.load(cache)
            return d["y"], d["u"], d["v"]
        except Exception:
            pass
    yy, xx = np.mgrid[0:height + 64 + 8 * frames, 0:width + 64 + 8 * frames]
    yy, xx = yy + first, xx + 3 * first                           # the window of the endless texture this chunk pans over
    base = 512 + 180 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 120 * np.sin((xx + yy) / 11.0) + 60 * np.sin(xx / 3.1) * np.sin(yy / 4.3)
    obj = 700 + 150 * np.sin(np.mgrid[0:128, 0:128][1] / 5.0)
    ys, us, vs = [], [], []
    for k in range(frames):
        t = first + k
        fr = np.random.default_rng([seed, t])
        f = base[k:k + height, 3 * k:3 * k + width] + fr.normal(0, 12, (height, width))
        oy, ox = (100 + 2 * t) % (height - 128), (200 + 7 * t) % (width - 128)
        f[oy:oy + 128, ox:ox + 128] = obj
        f = np.clip(f + fr.normal(0, 3, f.shape), 0, 1023)
        ys.append(f.astype(np.int16))            # (converted per frame — the same values as converting the stack: an 8K x 65 clip would hold 26 GB of float64 frames otherwise)
        sub = f[::2, ::2]
        us.append(np.clip(512 + 0.2 * (sub - 512), 0, 1023).astype(np.int16))
        vs.append(np.clip(512 - 0.15 * (sub - 512), 0, 1023).astype(np.int16))
    g = lambda a: np.ascontiguousarray(np.stack(a))
    y, u, v = g(ys), g(us), g(vs)
    try:
        tmp = cache + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, y=y, u=u, v=v)
        os.replace(tmp, cache)
    except Exception:
        pass
    return y, u, v


def _synth_clip(width, height, frames, seed):
    rng = np.random.default_rng(seed)
    pad = 64 + 8 * frames
    yy, xx = np.mgrid[0:height + pad, 0:width + pad]
    base = 512 + 180 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 120 * np.sin((xx + yy) / 11.0) + 60 * np.sin(xx / 3.1) * np.sin(yy / 4.3)
    base = base + rng.normal(0, 12, base.shape)
    obj = 700 + 150 * np.sin(np.mgrid[0:128, 0:128][1] / 5.0)
    ys, us, vs = [], [], []
    for t in range(frames):
        f = base[t:t + height, 3 * t:3 * t + width].copy()
        oy, ox = (100 + 2 * t) % (height - 128), (200 + 7 * t) % (width - 128)
        f[oy:oy + 128, ox:ox + 128] = obj
        f = np.clip(f + rng.normal(0, 3, f.shape), 0, 1023)
        ys.append(f.astype(np.int16))            # (converted per frame — the same values as converting the stack: an 8K x 65 clip would hold 26 GB of float64 frames otherwise)
        sub = f[::2, ::2]
        us.append(np.clip(512 + 0.2 * (sub - 512), 0, 1023).astype(np.int16))
        vs.append(np.clip(512 - 0.15 * (sub - 512), 0, 1023).astype(np.int16))
    g = lambda a: np.ascontiguousarray(np.stack(a))
    return g(ys), g(us), g(vs)


WORKER = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import e2e_util as E, e2e_fps as F
cfg = json.loads(sys.argv[1])
L = E.load(cfg["mask"] != 0, cfg.get("lib"))
if cfg["mask"]:
    import torch  # one HIP runtime in the process
    assert L.vvref_install_hip_hooks(cfg["mask"]) == 0
yuv = F.synth_clip_chunk(cfg["w"], cfg["h"], cfg["first"], cfg["frames"]) if "first" in cfg else F.synth_clip(cfg["w"], cfg["h"], cfg["frames"])
md5, n, secs = E.encode(L, yuv, cfg["w"], cfg["h"], 10, 10, threads=cfg["threads"], preset=E.PRESETS[cfg.get("preset", "faster")], simd=cfg.get("simd"), options=cfg.get("options"))
calls = None
if cfg["mask"]:
    c = np.zeros(43, np.uint64); L.vvref_hip_hook_calls_ex(c.ctypes.data, 43); calls = [int(x) for x in c]
out = {"mask": cfg["mask"], "md5": md5, "bytes": n, "secs": secs, "fps": cfg["frames"] / secs, "calls": calls}
if calls:
    out["pcie_MB_per_picture"] = {"up": round(calls[25] / 1e6 / cfg["frames"], 3), "down": round(calls[26] / 1e6 / cfg["frames"], 3)}
print(json.dumps(out))
''' % os.path.join(ROOT, "tests")


def run(cfg, timeout=3000, env=None):
    # the encoder process runs with the runtime's own number of hardware queues: bench.py's setting is for ITS five-plus-two replay streams, and the encoder measured slower with
    # it (1080p, 8 threads, five alternating runs: 4 queues 48.7 fps, 8: 47.3, 10: 47.4; tools/exp/e2e_queues_ab.py).  $VVHIP_E2E_KEEP_QUEUES=1: pass the variable through.
    env = dict(os.environ if env is None else env)
    if env.get("VVHIP_E2E_KEEP_QUEUES") != "1":
        env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, "-c", WORKER, json.dumps(cfg)], capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])          # (a profiling build prints its stage table when the library unloads: after this line)
    if cfg.get("keep_stdout"):
        out["stdout"] = r.stdout
    return out


def parse_time_profile(text):
    """the stage table an ENABLE_TIME_PROFILING build prints when the encoder closes (CommonLib/TimeProfiler.h operator<<: one line per stage, name then milliseconds, ...)
    -> {stage name: ms}"""
    import re
    stages = {}
    for line in text.splitlines():
        m = re.match(r"^\s*(P_[A-Z0-9_]+)\s+([0-9.eE+-]+)", line)
        if m:
            try:
                stages[m.group(1)] = float(m.group(2))
            except ValueError:
                pass
    for k in ("P_STAGES", "P_IGNORE"):
        stages.pop(k, None)
    return stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--masks", default="0,16,144")
    ap.add_argument("--preset", default="faster")
    a = ap.parse_args()
    res = [run(dict(w=a.width, h=a.height, frames=a.frames, threads=a.threads, mask=int(m), preset=a.preset)) for m in a.masks.split(",")]
    out = {"clip": "%dx%d 10-bit synthetic (config-2 generator), %d frames, preset %s, QP 32" % (a.width, a.height, a.frames, a.preset),
           "threads": a.threads, "host_cpus": os.cpu_count(), "runs": res, "bitstreams_identical": len({r["md5"] for r in res}) == 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
