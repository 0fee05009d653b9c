"""bench.py's constants and step order (host logic only: importable without torch or a device)."""
import os
import time

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_GHZ, N_CU, N_SIMD = 2.4, 256, 1024
LAYER_POCS = {0: 31, 1: 15, 2: 23, 3: 3, 4: 5, 5: 2}          # one recorded picture per temporal layer of the 65-frame encode (its GOP anchors at POC 31 / 63)
KERNEL_NAMES = {"ME_stage": "meStageKernel", "ME_int": "meIntKernel", "ME_item": "meItemKernel", "TU": "tuMxMultiKernel", "DMVR": "dmvrRefineKernel",
                # north-star leg C (MCTF, tools/bench_mctf.py): candidate scoring of estimateLumaLn, the above / left candidates scored in parallel, the resolution of the above / left recurrence, the bilateral filter
                "MCTF_search": "meSearch", "MCTF_nb": "meNeighbourKernel", "MCTF_fix": "meFixKernel", "MCTF_apply": "mctfApplyKernel"}
KERNEL_LABELS = {"MCTF_search": "meSearchKernel (+ meSearchCoopKernel on the coarse levels)"}          # what a row is called where the prefix above is not a kernel's whole name


def kernel_label(cls):
    return KERNEL_LABELS.get(cls, KERNEL_NAMES.get(cls, cls))


STEP_CLASSES = ("ME_stage", "ME_int", "ME_item", "TU", "DMVR")      # the classes of `value`'s step (legs A + B); the MCTF classes join the table when leg C runs at the GOP's cadence


# The replay order of the 32 pictures of a GOP cycle (1 x TL0, 1 x TL1, 2 x TL2, 4 x TL3, 8 x TL4, 16 x TL5): a low-discrepancy interleaving, NOT the coding order — the recorded
# pictures are independent work for the device, and a run of K steps should hold the layers close to their GOP share whatever K is.  The cycle starts at its key (intra) picture;
# the first 20 positions hold 1 x TL0, 0 x TL1, 1 x TL2, 3 x TL3, 5 x TL4, 10 x TL5 (time-weighted within ~1 % of the whole cycle's mean on the recorded 1080p lists), 32 = the exact mix.
STEP_LAYERS = (0, 5, 4, 5, 3, 5, 4, 5, 2, 5, 4, 5, 3, 5, 4, 5, 3, 5, 4, 5, 2, 5, 4, 5, 1, 5, 4, 5, 3, 5, 4, 5)


def layer_of_step(s):
    """temporal layer of the picture step s replays"""
    return STEP_LAYERS[s % 32]


def step_of_rank(k, rank, world):
    """N ranks: rank r's k-th step is position k + r * (32 / N) of the same cycle — at any time the ranks work on N different pictures of one sequence, and every rank's window
    of K steps holds (nearly) the same layer mix (taking every N-th position instead would hand one rank all the heavy layers: the odd positions are all TL5)"""
    return k + (rank * 32) // max(1, world)


GOP_WEIGHT = {l: sum(1 for s in range(32) if layer_of_step(s) == l) for l in range(6)}


# ---------------------------------------------------------------------------------------------------------------------- recording
def prepare_recordings(width, height, frames, pocs, tag="faster", threads=8):
    """the recorded lists of the pictures `pocs` (cached under /tmp: the rocprofv3 passes and later runs reuse them)"""
    from vvenc_amd import recorded as R
    d = os.path.join("/tmp", "vvhip_rec_%dx%d_%d_%s" % (width, height, frames, tag))
    info = {"dir": d, "recorded_now": False}
    need = [p for p in pocs if not os.path.exists(os.path.join(d, "poc%d.json" % p))]
    if need:
        t0 = time.perf_counter()
        res = R.record(d, width, height, frames, pocs=need, threads=threads, preset=tag)
        info.update(recorded_now=True, record_s=round(time.perf_counter() - t0, 2), encoder_md5=res["md5"], encoder_s=round(res["secs"], 2))
    return {p: R.RecordedPicture(os.path.join(d, "poc%d" % p)) for p in pocs}, info
