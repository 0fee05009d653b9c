"""Work lists RECORDED from the reference encoder (bindings/vvenc/vvenc_hip_recorder.*, hook bit 131072): loader and host-side list building.

A recorded picture (poc<N>.json + poc<N>.bin) holds what the encoder's own search pushed through the kernel tables of the hot path for that picture:
  me / cand   every InterSearch::xMotionEstimation call (EncoderLib/InterSearch.cpp:1976-2130) with its integer candidates in call order
  stage       every xPatternRefinement stage (:760-880): base vector + the 9 offsets, costs of the positions the encoder evaluated
  dist        every other call through RdCost::m_afpDistortFunc[0][*] (merge / AMVP / intra / residual SSE ...), operands as plane positions or pool blocks
  tu          every forward transform TrQuant::xT (CommonLib/TrQuant.cpp:481-564) with its residual block
  dmvr        every DMVR sub-block search (CommonLib/InterPrediction.cpp:1312-1392) with its result
  planes      the original picture (as the CTU copies read it) and the luma reconstruction of every reference picture, pool = compact sample blocks
Record layouts mirror bindings/vvenc/vvenc_hip_recorder.h (packed, little-endian).  `record()` drives the encoder built with the binding to produce them.
"""
import json
import os

import numpy as np

OPERAND = [("plane", "<i2"), ("pad", "<i2"), ("x", "<i4"), ("y", "<i4")]
DT = {
    "planes": np.dtype([("kind", "<i4"), ("poc", "<i4"), ("comp", "<i4"), ("width", "<i4"), ("height", "<i4"), ("stride", "<i4"), ("margin", "<i4"), ("offset", "<i8")]),
    "me": np.dtype([("cuX", "<i4"), ("cuY", "<i4"), ("w", "<i2"), ("h", "<i2"), ("refPlane", "<i2"), ("bi", "u1"), ("list", "u1"), ("patternPool", "<i4"),
                    ("firstCand", "<i4"), ("nCand", "<i4"), ("firstStage", "<i4"), ("nStage", "<i4")]),
    "cand": np.dtype([("me", "<i4"), ("x", "<i4"), ("y", "<i4"), ("df", "u1"), ("subShift", "u1"), ("pad", "<u2"), ("cost", "<u8")]),
    "stage": np.dtype([("me", "<i4"), ("baseX", "<i4"), ("baseY", "<i4"), ("baseHor", "<i2"), ("baseVer", "<i2"), ("iFrac", "u1"), ("hadMode", "u1"), ("reduceTap", "u1"),
                       ("altHpel", "u1"), ("pad", "<i4"), ("cost", "<u8", (9,))]),
    "dist": np.dtype([("df", "u1"), ("subShift", "u1"), ("bitDepth", "u1"), ("ctx", "u1"), ("w", "<i2"), ("h", "<i2"),
                      ("org_plane", "<i2"), ("org_pad", "<i2"), ("org_x", "<i4"), ("org_y", "<i4"), ("cur_plane", "<i2"), ("cur_pad", "<i2"), ("cur_x", "<i4"), ("cur_y", "<i4"), ("cost", "<u8"), ("maskPool", "<i4")]),
    "tu": np.dtype([("comp", "u1"), ("trHor", "u1"), ("trVer", "u1"), ("flags", "u1"), ("w", "<i2"), ("h", "<i2"), ("qp", "<i2"), ("bitDepth", "<i2"), ("x", "<i4"), ("y", "<i4"), ("pool", "<i4")]),
    "dmvr": np.dtype([("ref0Plane", "<i2"), ("ref1Plane", "<i2"), ("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("frac0x", "<i2"), ("frac0y", "<i2"), ("frac1x", "<i2"), ("frac1y", "<i2"),
                      ("dx", "<i2"), ("dy", "<i2"), ("mvdX", "<i2"), ("mvdY", "<i2"), ("pad", "<i4"), ("minCost", "<u8")]),
    "plane_data": np.dtype("<i2"), "pool": np.dtype("<i2"),
}
SKIPPED = np.uint64(0xFFFFFFFFFFFFFFFF)

# DFunc numbering of the reference (CommonLib/TypeDef.h:339-382): table index -> function family of the C ABI
DF_SSE, DF_SAD, DF_HAD, DF_HAD_2SAD, DF_SAD_MASK, DF_HAD_FAST = 0, 8, 16, 24, 25, 26


def df_family(df):
    df = int(df)
    if df < 8:
        return "SSE"
    if df < 16:
        return "SAD"
    if df < 24:
        return "HAD"
    if df == 24:
        return "HAD_2SAD"
    if df == 25:
        return "SAD_MASK"
    return "HAD_fast"


# the nine refinement offsets of a stage in the order the encoder walks them (s_acMvRefineH / s_acMvRefineQ, EncoderLib/InterSearch.cpp:67-91)
REFINE_H = np.array([(0, 0), (0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, -1), (-1, 1), (1, 1)], np.int32)
REFINE_Q = np.array([(0, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (1, 1)], np.int32)


class RecordedPicture:
    def __init__(self, base):
        """base: path without extension (…/poc23)"""
        self.meta = json.load(open(base + ".json"))
        raw = np.memmap(base + ".bin", dtype=np.uint8, mode="r")
        self.sec = {}
        for s in self.meta["sections"]:
            dt = DT[s["name"]]
            assert s["bytes"] == s["count"] * dt.itemsize, (s, dt.itemsize)
            self.sec[s["name"]] = np.frombuffer(raw, dtype=dt, count=s["count"], offset=s["offset"]) if s["count"] else np.zeros(0, dt)
        for k, v in self.sec.items():
            setattr(self, k, v)
        self.poc, self.width, self.height = self.meta["poc"], self.meta["width"], self.meta["height"]
        self.tlayer, self.slice_type, self.slice_qp = self.meta["tlayer"], self.meta["slice_type"], self.meta["slice_qp"]

    def plane_array(self, i):
        """the samples of plane i as dumped: (height + 2 margin) x (width + 2 margin) int16, margin included"""
        p = self.planes[i]
        m = int(p["margin"])
        hh, ww = int(p["height"]) + 2 * m, int(p["width"]) + 2 * m
        return self.plane_data[int(p["offset"]):int(p["offset"]) + hh * ww].reshape(hh, ww), m

    @property
    def sample_pairs(self):
        return int(self.meta["sample_pairs_luma"]) + int(self.meta["sample_pairs_chroma"])

    def stage_positions(self, st):
        """quarter-sample positions (relative to the block at (baseX, baseY)) of the nine candidates of stage records `st`: (refine[i] + base) * iFrac, like the binding's patternCosts"""
        ref = np.where((st["iFrac"] == 2)[:, None, None], REFINE_H[None], REFINE_Q[None])
        base = np.stack([st["baseHor"], st["baseVer"]], 1).astype(np.int32)[:, None, :]
        return (ref + base) * st["iFrac"].astype(np.int32)[:, None, None]

    def summary(self):
        out = {"poc": self.poc, "tlayer": self.tlayer, "slice_type": self.slice_type, "slice_qp": self.slice_qp, "sample_pairs": self.sample_pairs,
               "sample_pairs_per_1p5WH": round(self.sample_pairs / (1.5 * self.width * self.height), 2)}
        me, cand, st, d, tu = self.me, self.cand, self.stage, self.dist, self.tu
        mw = me["w"][cand["me"]].astype(np.int64) if cand.size else np.zeros(0, np.int64)
        mh = me["h"][cand["me"]].astype(np.int64) if cand.size else np.zeros(0, np.int64)
        out["me_calls"] = int(me.size)
        out["me_bi"] = int(me["bi"].sum())
        out["int_candidates"] = int(cand.size)
        out["int_candidate_pairs"] = int((mw * mh).sum())
        ev = (st["cost"] != SKIPPED).sum(1) if st.size else np.zeros(0, np.int64)
        sw = me["w"][st["me"]].astype(np.int64) if st.size else np.zeros(0, np.int64)
        sh = me["h"][st["me"]].astype(np.int64) if st.size else np.zeros(0, np.int64)
        out["subpel_stages"] = int(st.size)
        out["subpel_positions_evaluated"] = int(ev.sum())
        out["subpel_pairs"] = int((ev * sw * sh).sum())
        out["other_calls"] = int(d.size)
        out["other_pairs"] = int((d["w"].astype(np.int64) * d["h"]).sum())
        out["tus"] = int(tu.size)
        out["tu_coefficients"] = int((tu["w"].astype(np.int64) * tu["h"]).sum())
        out["dmvr_subblocks"] = int(self.dmvr.size)
        return out


def load_dir(path):
    pics = {}
    for f in sorted(os.listdir(path)):
        if f.startswith("poc") and f.endswith(".json"):
            p = RecordedPicture(os.path.join(path, f[:-5]))
            pics[p.poc] = p
    return pics


RECORD_WORKER = r'''
import sys, json, os
sys.path.insert(0, %r)
import e2e_util as E, e2e_fps as F
cfg = json.loads(sys.argv[1])
L = E.load(True)
yuv = F.synth_clip(cfg["w"], cfg["h"], cfg["frames"])
md5, n, secs = E.encode(L, yuv, cfg["w"], cfg["h"], 10, 10, threads=cfg["threads"], preset=E.PRESETS[cfg.get("preset", "faster")], simd="HIP:131072")
print(json.dumps({"md5": md5, "bytes": n, "secs": secs}))
'''


def record(out_dir, width=1920, height=1080, frames=65, pocs=(15, 23, 5, 2), threads=8, preset="faster", light=False, timeout=1800):
    """runs the reference encoder built with the binding (bindings/vvenc/_build/libvvenc_hip_enc.so) on the BASELINE config-2 style clip with the recorder hook:
    the encoder keeps its CPU kernels (bitstream = the CPU encoder's), the work lists of the pictures `pocs` land in out_dir.  Needs no GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, VVHIP_RECORD_DIR=out_dir, VVHIP_RECORD_POCS=",".join(str(p) for p in pocs))
    if light:
        env["VVHIP_RECORD_LIGHT"] = "1"
    cfg = dict(w=width, h=height, frames=frames, threads=threads, preset=preset)
    r = subprocess.run([sys.executable, "-c", RECORD_WORKER % os.path.join(root, "tests"), json.dumps(cfg)], capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError("recording failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    import sys
    for poc, p in load_dir(sys.argv[1]).items():
        print(json.dumps(p.summary()))
