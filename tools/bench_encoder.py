"""bench.py's encoder legs: the real reference encoder end to end (CPU kernels vs --SIMD=HIP), N encoder instances for N > 1, the MCTF stage workload and the
preset-medium 4K picture (BASELINE configs[3]'s lists)."""
import os
import sys
import time

import torch

from vvenc_amd import sharding
from bench_common import prepare_recordings
from bench_reference import cpu_baseline, host_cpu_info, parity_check, usable_cores

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---------------------------------------------------------------------------------------------------------------------- device side
class Mctf1080:
    """what tools/bench_synthetic.mctf_stage needs from a workload: one picture pair of the config-2 generator"""

    def __init__(self, width, height, bit_depth=10):
        from vvenc_amd.workload import synth_frame_pair
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self.cur_np, self.ref_np = synth_frame_pair(width, height, 1080 if width == 1920 else 2160, bit_depth)


def e2e_encoder(width, height, frames, threads, pairs, other_threads=(), scalar=True, stage_split=True):
    """the real reference encoder end to end (SURVEY 8d metric): CPU kernels vs --SIMD=HIP, same clip, same threads, `pairs` alternating pairs after one discarded
    run (median AND best per side: single runs spread by up to 20 % on a shared host); the same pair once for every T of `other_threads` (T = 1, T = usable cores); one
    --SIMD=SCALAR run in the md5 set (the reference's own invariant, cmake/modules/vvencTests.cmake:52-53); the stage split of a -DENABLE_TIME_PROFILING=1 build
    (CommonLib/TimeProfiler.h:72-110), single-threaded, that bounds what the device stages can gain (Amdahl).  Subprocesses: the SIMD level is process-wide."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    import e2e_util
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        return {"skipped": "the compiled reference encoder (oracle/_ref) and the encoder with the binding (bindings/vvenc/_build) are not both built"}
    prod = e2e_production_mask()
    base = dict(w=width, h=height, frames=frames)
    med = lambda v: sorted(v)[len(v) // 2]
    same = []          # per thread count: every run of the row (CPU kernels and --SIMD=HIP) gave ONE bitstream.  (Rows are not compared with each other: above 8 threads the
                       # encoder switches its inter-frame line synchronisation on, which restricts the motion search and changes the stream — with the CPU encoder alone.)

    def pair_rows(t, n, discard):
        if discard:
            e2e_fps.run(dict(base, threads=t, mask=0), timeout=1200)          # discarded run: clip cache, page cache, clocks
        runs = [e2e_fps.run(dict(base, threads=t, mask=m), timeout=1800) for m in (0, prod) * n]
        same.append(len({r["md5"] for r in runs}) == 1)
        c, h = [r["fps"] for r in runs if r["mask"] == 0], [r["fps"] for r in runs if r["mask"] == prod]
        return runs, {"threads": t, "pairs": n, "cpu_fps": round(med(c), 2), "hip_fps": round(med(h), 2), "speedup": round(med(h) / med(c), 3),
                      "cpu_fps_best": round(max(c), 2), "hip_fps_best": round(max(h), 2), "speedup_best": round(max(h) / max(c), 3), "runs_fps": [round(r["fps"], 2) for r in runs]}
    runs, head = pair_rows(threads, pairs, True)
    out = {"clip": "%dx%d 10-bit synthetic (config-2 generator), %d frames, preset faster, QP 32" % (width, height, frames)}
    out.update(head)
    out.update({"runs_order": "cpu, hip alternating, %d pairs after one discarded run; cpu_fps / hip_fps are medians, *_best the fastest run of each side" % pairs,
                "hook_mask": prod, "md5": runs[0]["md5"], "pcie_MB_per_picture": runs[1].get("pcie_MB_per_picture"),
                "device_stages": "MCTF motion estimation (all references of a picture per call) + bilateral filter, ALF statistics of whole pictures (--SIMD=HIP production mask)"})
    rows = []
    for t in other_threads:
        try:
            rows.append(pair_rows(t, 1, False)[1])
        except Exception as e:
            rows.append({"threads": t, "error": str(e)[-200:]})
    if rows:
        out["other_threads"] = rows
    if scalar:
        # the reference's own invariant (cmake/modules/vvencTests.cmake:52-53: --SIMD=SCALAR stream == default stream) holds on its 80x44 test clip; on some clips (416x240 'pan')
        # the reference's SCALAR stream differs from its SSE41 / AVX2 stream with MCTF on — with the CPU encoder alone.  Round 6 located what it is NOT: the encoder with its MCTF
        # search and filter replaced by scalar-row arithmetic (the test double of tests/sim) still emits the AVX2 stream, and the reference's scalar and x86 rows of the filter agree
        # sample for sample (tests/test_oracle_vs_reference.py, tolerance 0: the x86 row's single-precision "+ 0.5f" is exact wherever it could matter).  The three-way comparison
        # is therefore made WITH MCTF on: hip_equals_default is the claim (the comparison target of north_star is the x86 row), scalar_equals_default is reported.
        try:
            r = e2e_fps.run(dict(base, threads=threads, mask=0, simd="SCALAR"), timeout=1800)
            hip_md5 = [x["md5"] for x in runs if x["mask"] == prod][0]
            out["scalar"] = {"fps": round(r["fps"], 2), "md5_equal_default": r["md5"] == runs[0]["md5"], "hip_equals_default": hip_md5 == runs[0]["md5"], "hip_equals_scalar": hip_md5 == r["md5"],
                             "mctf": "on (the encoder's defaults)", "runs": "--SIMD=SCALAR, default SIMD (AVX2), --SIMD=HIP on the same clip and thread count"}
            off = "MCTF=0;BIM=0"
            three = [e2e_fps.run(dict(base, threads=threads, mask=m, simd=sd, options=off), timeout=1800)["md5"] for m, sd in ((0, "SCALAR"), (0, None), (prod, None))]
            out["scalar"].update({"md5_equal": len(set(three)) == 1, "md5_equal_options": off})
            out["scalar"]["note"] = ("MCTF ON: hip_equals_default (--SIMD=HIP == AVX2: asserted in bitstreams_identical too), md5_equal_default (the reference's own SCALAR vs AVX2); "
                                     "md5_equal: SCALAR == AVX2 == HIP with MCTF off")
        except Exception as e:
            out["scalar"] = {"error": str(e)[-200:]}
    out["bitstreams_identical"] = all(same)
    out["md5_set"] = "per thread count: default SIMD (AVX2) == --SIMD=HIP over %d + %d pairs" % (pairs, len(rows))
    if stage_split:
        try:
            out["stage_split"] = e2e_stage_split(width, height, min(frames, 33))
        except Exception as e:
            out["stage_split"] = {"error": str(e)[-300:]}
    return out


HOT_PATH_STAGES = {"north_star_A_distortion_in_motion_search": ("P_INTER_MVD_SEARCH", "P_INTER_MVD_SEARCH_B", "P_FRAC_PEL", "P_QPEL", "P_QPEL_INTERP", "P_HPEL_INTERP", "P_INTER_MRG_EST_RD_CAND", "P_INTER_MRG_DMVR"),
                   "north_star_B_transform_quant": ("P_TRAFO", "P_QUANT", "P_DEQUANT"),
                   "north_star_C_mctf": ("P_MCTF", "P_MCTF_SEARCH", "P_MCTF_SEARCH_SUBPEL", "P_MCTF_APPLY"),
                   "alf": ("P_ALF", "P_ALF_CLASS", "P_ALF_STATS", "P_ALF_ENC", "P_ALF_MERGE", "P_ALF_DERIVE_COEF", "P_ALF_ENC_CTB", "P_ALF_REC")}
DEVICE_STAGES_IN_PRODUCTION = ("P_MCTF", "P_MCTF_SEARCH", "P_MCTF_SEARCH_SUBPEL", "P_MCTF_APPLY", "P_ALF_STATS")


def e2e_stage_split(width, height, frames):
    """the reference's own stage timers (ENABLE_TIME_PROFILING build of oracle/_ref, `make -C oracle/ref prof`; single-threaded like SURVEY A.4) on the first `frames` frames of the
    clip: the share of the encoder's time inside the north-star stages and inside the stages --SIMD=HIP runs on the device -> the Amdahl bound of the e2e speedup"""
    import e2e_fps
    import e2e_util
    so = os.path.join(os.path.dirname(e2e_util.REF_SO), "prof", "libvvenc_ref_prof.so")
    if not os.path.exists(so):
        return {"skipped": "oracle/_ref/prof/libvvenc_ref_prof.so is not built (make -C oracle/ref prof)"}
    r = e2e_fps.run(dict(w=width, h=height, frames=frames, threads=0, mask=0, lib=so, keep_stdout=True), timeout=1800)
    stages = e2e_fps.parse_time_profile(r.get("stdout", ""))
    tot = sum(stages.values())
    if not tot:
        return {"error": "no stage table in the encoder's output", "fps": round(r["fps"], 2)}
    share = lambda names: round(sum(stages.get(n, 0.0) for n in names) / tot, 4)
    dev = share(DEVICE_STAGES_IN_PRODUCTION)
    dev_hi = share(tuple(set(DEVICE_STAGES_IN_PRODUCTION + HOT_PATH_STAGES["alf"])))          # (the profiler books the ALF statistics under P_ALF when its sub-stages are not split out)
    return {"frames": frames, "threads": 0, "fps_profiled_build": round(r["fps"], 2), "total_ms": round(tot, 1),
            "share": {k: share(v) for k, v in HOT_PATH_STAGES.items()}, "share_top": {k: round(v / tot, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])[:8]},
            "device_stage_share": dev, "amdahl_bound_speedup": round(1.0 / (1.0 - dev), 3) if dev < 1 else None,
            "device_stage_share_with_all_of_alf": dev_hi, "amdahl_bound_speedup_with_all_of_alf": round(1.0 / (1.0 - dev_hi), 3) if dev_hi < 1 else None,
            "note": "shares of the single-threaded encoder's stage time (TimeProfiler exclusive times); device_stage_share = the stages --SIMD=HIP's production mask moves to the GPU "
                    "(MCTF search + apply, ALF statistics): 1 / (1 - share) bounds e2e `speedup` if those stages cost nothing; the ALF statistics are booked under P_ALF as a whole when the "
                    "build does not split them out: _with_all_of_alf is the upper end"}


def e2e_production_mask():
    return 16 + 128 + 8192


# ---------------------------------------------------------------------------------------------------------------------- N encoder instances (N > 1)
def e2e_instances(rank, local_rank, world, width=1920, height=1080, frames=65):
    """The BASELINE metric at N GPUs: N encoder instances, one per rank / GPU, each with its share of the host cores, over GOP chunks of ONE sequence (chunk r = frames
    r * 33 .. r * 33 + 32 of the config-2 generator's endless clip; every chunk starts with its own intra picture like a closed-GOP segment — how a sequence is split for
    chunk-parallel encoding; inside one encoder the reference's own GOP parallelism is EncGOP.cpp:1647-1651 / vvencCfg.cpp:2188-2199).  All instances run at the same
    time, first with CPU kernels, then with --SIMD=HIP on their GPU: aggregate fps = N * frames / the slowest instance's ENCODE time (the encoder's own clock around its
    encode loop: process start, `import torch` and context creation of an instance are not part of a sequence's frame rate; the wall-clock figure is reported next to it);
    per-chunk md5 CPU == HIP."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    import e2e_util
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        return {"skipped": "the compiled reference encoder (oracle/_ref) and the encoder with the binding (bindings/vvenc/_build) are not both built"} if rank == 0 else None
    threads = max(1, usable_cores(host_cpu_info()) // world)
    ndev = max(1, torch.cuda.device_count())
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = str(local_rank % ndev)                     # the instance sees ONE device: its rank's GPU
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VVHIP_SHARE_DEVICE", "VVHIP_DIST_BACKEND"):
        env.pop(k, None)
    prod = e2e_production_mask()
    cfg = dict(w=width, h=height, frames=frames, first=rank * frames, threads=threads)
    e2e_fps.synth_clip_chunk(width, height, rank * frames, frames)          # (the chunk's clip is made before the clock starts; the instances load it from the cache)
    res = {}
    for name, mask in (("warm", 0), ("cpu", 0), ("hip", prod)):
        sharding.barrier()
        t0 = time.perf_counter()
        try:
            r = e2e_fps.run(dict(cfg, mask=mask), timeout=1200, env=env)
        except Exception as e:
            r = {"md5": "error: " + str(e)[-200:], "fps": 0.0}
        dt = sharding.max_over_ranks(time.perf_counter() - t0, device="cuda")          # (device tensors: RCCL has no host reductions)
        enc = sharding.max_over_ranks(float(r.get("secs") or 1e9), device="cuda")        # the slowest instance's encode time
        res[name] = (r, dt, enc)
    same = 1.0 if res["cpu"][0]["md5"] == res["hip"][0]["md5"] and not res["cpu"][0]["md5"].startswith("error") else 0.0
    all_same = -sharding.max_over_ranks(-same, device="cuda")                # min over ranks
    gathered = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(gathered, {"rank": rank, "chunk_first_frame": rank * frames, "cpu_fps": round(res["cpu"][0]["fps"], 2), "hip_fps": round(res["hip"][0]["fps"], 2), "md5": res["cpu"][0]["md5"][:12],
                                          "md5_hip": res["hip"][0]["md5"][:12]})
    if rank != 0:
        return None
    cpu_fps, hip_fps = world * frames / res["cpu"][2], world * frames / res["hip"][2]
    cpu_wall, hip_wall = world * frames / res["cpu"][1], world * frames / res["hip"][1]
    return {"instances": world, "frames_per_chunk": frames, "threads_per_instance": threads, "clip": "%dx%d 10-bit, chunk r = frames %d r .. %d r + %d of one endless config-2 sequence, preset faster" % (width, height, frames, frames, frames - 1),
            "cpu_fps_aggregate": round(cpu_fps, 2), "hip_fps_aggregate": round(hip_fps, 2), "speedup": round(hip_fps / cpu_fps, 3) if cpu_fps else None,
            "cpu_fps_aggregate_wall": round(cpu_wall, 2), "hip_fps_aggregate_wall": round(hip_wall, 2),
            "chunk_bitstreams_identical": bool(all_same == 1.0), "hook_mask": prod, "per_instance": gathered,
            "timing": "aggregate = N x frames / the slowest instance's encode time (all instances start at one barrier and run concurrently); _wall: from the barrier to the slowest instance's exit, i.e. "
                      "with process start, `import torch` and HIP context creation of the instance (≈1.5 s, a one-off per sequence, not per chunk of a long one); one discarded CPU run first",
            "note": "one encoder process is host-bound (DESIGN 7): N-GPU frames/s in the sense of the metric is N instances; it scales with the host cores each instance gets, the GPUs are never the limit"}


# ---------------------------------------------------------------------------------------------------------------------- BASELINE configs[3]'s lists (preset medium)
def replay_medium_4k(hp, streams=5):
    """one 3840x2160 picture of a preset-MEDIUM encode (BASELINE configs[3]'s geometry and preset: CTU 128, multi-type tree -> rectangular blocks 4..128, GEO masked SADs, two
    references per list) through the same batched path: nothing of the recording left out, every output against the encoder's own values, time per picture"""
    from vvenc_amd.replay import RecordedWorkload
    pics, info = prepare_recordings(3840, 2160, 9, [4], tag="medium", threads=16)
    wl = RecordedWorkload(hp, pics[4])
    lanes = [hp.fork() for _ in range(streams)]          # (lanes on their contexts' own streams: one HIP stream per lane, HotPath.fork)
    wl.bind_lanes(lanes)
    for _ in range(3):
        wl.run_lanes()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        wl.run_lanes()
    torch.cuda.synchronize()
    ms = 1000.0 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        wl.run()
    torch.cuda.synchronize()
    ms1 = 1000.0 * (time.perf_counter() - t0) / n
    par = parity_check({5: wl})
    me = wl.pic.me
    shapes = sorted({(int(w), int(h)) for w, h in zip(me["w"].tolist(), me["h"].tolist())})
    out = {"clip": "3840x2160 10-bit synthetic config-2 clip, 9 frames, preset medium, picture POC 4 (TL5)", "ms_per_picture": round(ms, 4), "pictures_per_s": round(1000.0 / ms, 1),
           "ms_per_picture_single_stream": round(ms1, 4), "sample_pairs": int(wl.pic.sample_pairs), "sample_pairs_per_1p5WH": round(wl.pic.sample_pairs / (1.5 * 3840 * 2160), 1),
           "recorded_calls_outside_the_lists": wl.dropped, "nothing_dropped": bool(wl.nothing_dropped), "parity": par,
           "work": {"me_calls": int(me.size), "me_block_shapes": ["%dx%d" % s_ for s_ in shapes], "integer_candidates": int(wl.plan_cands.size), "subpel_stages": int(wl.stage_jobs.size),
                    "table_calls": int(wl.items.size), "masked_sad_calls": int(wl.mask_items.size), "tus": int(sum(g["n"] for g in wl.tu_groups)),
                    "tu_shapes": sorted({"%dx%d" % (g["w"], g["h"]) for g in wl.tu_groups}), "dmvr_subblocks": int(sum(g["n"] for g in wl.dmvr_groups)), "plan": wl.me_info},
           "recording": info}
    try:
        cb = cpu_baseline({5: wl}, passes=3)
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "passes", "threads_pinned", "seconds_per_picture_by_layer") if k in cb}
    except Exception as e:
        out["cpu_baseline"] = {"error": str(e)[:200]}
    for c in lanes:
        c.close()
    return out

