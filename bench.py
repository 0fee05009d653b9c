#!/usr/bin/env python3
"""bench.py — pictures/sec of the MI355X hot path on work lists RECORDED from the reference encoder, with the roofline of the dominant kernel, in-run parity, the CPU baseline
and the end-to-end encoder.  The LAST stdout line is one compact JSON object (<= 8 KB, tools/bench_line.py); the full result goes to bench_detail.json and to stdout before it.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

`value` (BASELINE configs[1]): a step = ONE picture's hot-path work exactly as the reference encoder produced it.  Before the clock starts the encoder built with the binding
(bindings/vvenc) encodes the 1920x1080 10-bit config-2 clip (65 frames, preset faster) on its CPU kernels with the work-list recorder on (hook bit 131072): every call through
RdCost's table, every InterSearch::xMotionEstimation with its integer candidates and xPatternRefinement stages, every TU of TrQuant::xT with its residual, every DMVR sub-block —
for one picture of each temporal layer.  The lists + the pictures' planes are uploaded once; step s replays the picture of layer STEP_LAYERS[s mod 32] (tools/bench_common.py):
    motion-search plan   integer candidates (LDS windows) + sub-pel refinement stages (interpolation fused with the Hadamard) + merge / AMVP / intra / SSE table calls
    TU lists             fused xT -> needRdoq -> quant -> dequant -> xIT -> SSE, luma + chroma, DCT-2 / DST-7, 4..64
    DMVR lists           bilinear prediction + 25-point search + error surface per sub-block
on five HIP streams.  N GPUs: rank r replays position k + 32 r / N of the same cycle at its step k; the reconstructed picture a sharded encoder would hand to the other ranks is
broadcast (RCCL) every --exchange-every steps inside the timed region, overlapped; value = N * K pictures / max-over-ranks time, "weak".  The default cadence is ONE broadcast
per GOP cycle of 32 steps and rank: north_star's sharding by GOPs — every rank works on its own GOP chunk, the chunk's key picture is the one reference that crosses ranks
(DESIGN 7).  Reported next to it for N > 1: no_exchange (kernel scaling alone) and exchange_every_reference (a broadcast every 2 steps: the case in which the pictures of ONE
GOP are spread over the ranks and all 16 reference pictures of the cycle cross — the worst case, `value`'s cadence up to round 5).

Objects of the result (rank 0; each can be switched off; a failure is reported in place and never costs the headline).  In the compact line: a summary of each.
  roofline      dominant kernel: achieved = algorithmic bytes (SURVEY 8d per batch unit) / launch time from this run's rocprofv3 kernel trace, frac = achieved / 8 TB/s; traffic =
                fabric bytes per launch from this run's --pmc passes (calibrated on this GPU); frac_physical, frac_unique; the L1 / VALU fractions and which resource binds;
                cross-checks (no class above peak, the step's sum below peak) decide whether `basis` may call it an HBM fraction (tools/bench_profile.py)
  parity        every value the timed launches produced against the costs the REAL encoder computed while it was recorded and the reference's x86-SIMD entries on the TU lists
  cpu_baseline  the reference's own AVX2 entries on the host cores over the same lists: median of 5 pinned passes, the one-thread figure, the load average (tools/bench_reference.py)
  value_4k ...  the same on lists recorded from the 3840x2160 x 65 encode (BASELINE configs[2]'s geometry)
  config3_medium_4k, mctf, e2e / e2e_4k (the real encoder: CPU kernels vs --SIMD=HIP, T = 8 / 1 / all cores, --SIMD=SCALAR md5, stage split), e2e_instances (N > 1) (tools/bench_encoder.py)
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# a picture's five launch groups go to five HIP streams: the runtime deals streams onto 4 hardware queues by default (two groups would share one and serialize);
# measured on the recorded 1080p lists: 2 / 4 / 8 queues -> 8 519 / 12 036 / 12 326 pictures/s.  Round 6: streams are dealt IN CREATION ORDER, and every lane used to create two
# (a torch stream + its context's own): with 8 queues the DMVR lane shared the stage lane's queue.  Lanes now run on their contexts' own streams (HotPath.fork): one process =
# default stream + base context + 5 lanes + the MCTF lane = 8 streams on 8 queues, none shared: 15.1 -> 16.2 k pictures/s, GOP cycle with leg C 3.53 -> 3.30 ms
# (profiles/r06_hw_queues.log).  Leg C's jobs — independent pictures — are dealt over TWO lanes: a job is a chain of ~23 dependent launches whose boundaries cost ~11 us each
# under load, one lane was the combined cycle's critical path (3.30 -> 3.0-3.1 ms per GOP cycle with two; four change nothing more): 9 streams, 10 queues.  N > 1 adds the
# collective library's stream and the two exchange streams: 12 queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "10")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # multi-process GPU work on this pool: dmabuf IPC only (RCCL's buffer sharing fails with the legacy mode)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vvenc_amd import sharding  # noqa: E402
from bench_common import GOP_WEIGHT, KERNEL_NAMES, LAYER_POCS, STEP_LAYERS, kernel_label, layer_of_step, prepare_recordings, step_of_rank  # noqa: E402,F401
from bench_reference import ReferenceJobs, RecJob, cpu_baseline, parity_check  # noqa: E402,F401  (tests import these through bench)
import bench_line  # noqa: E402
import bench_mctf as BM  # noqa: E402


# ---------------------------------------------------------------------------------------------------------------------- one replay pass (a resolution)
def replay_pass(args, hp, rank, world, width, height, steps, warmup):
    """records (or loads) the layer pictures of the width x height encode, puts them on the device, times `steps` steps and collects the per-kernel times.
    -> (core fields of the JSON line, workloads, kern)"""
    from vvenc_amd.replay import RecordedWorkload
    rec_info = {}
    if rank == 0:
        pics, rec_info = prepare_recordings(width, height, 65, sorted(LAYER_POCS.values()))
    sharding.barrier()
    if rank != 0:
        pics, _ = prepare_recordings(width, height, 65, sorted(LAYER_POCS.values()))
    workloads = {layer: RecordedWorkload(hp, pics[poc]) for layer, poc in LAYER_POCS.items()}
    for l, wl in workloads.items():
        if not wl.nothing_dropped:
            raise RuntimeError("layer %d: recorded calls outside the lists: %s" % (l, wl.dropped))

    # ---- lanes: the refinement stages, integer windows and table calls of the plan, the TU lists and the DMVR lists of a picture are independent work: five HIP streams
    lanes, streams = None, []
    if args.streams > 1:
        # (lanes on their contexts' own streams unless $VVHIP_BENCH_TORCH_STREAMS=1: one HIP stream per lane, see HotPath.fork)
        own = os.environ.get("VVHIP_BENCH_TORCH_STREAMS") != "1"
        lanes = [hp.fork(None if own else torch.cuda.Stream()) for _ in range(5 if args.streams >= 5 else 3)]
        streams = [l.stream for l in lanes]
        for wl in workloads.values():
            wl.bind_lanes(lanes)

    # ---- north-star leg C at the GOP's cadence (tools/bench_mctf.py): the cycle's four filtered pictures resident, issued on two more streams (the cycle's four jobs in turn) by the steps that replay them
    mc = None
    if not args.no_mctf and (world == 1 or (width, height) == (args.width, args.height)):
        n_ml = max(1, int(os.environ.get("VVHIP_BENCH_MCTF_LANES", "2"))) if (args.streams > 1 and world == 1) else 1
        mk = lambda: hp.fork(None if os.environ.get("VVHIP_BENCH_TORCH_STREAMS") != "1" else torch.cuda.Stream())
        mc = BM.MctfCadence(hp, width, height, lane=mk() if args.streams > 1 else hp, more_lanes=[mk() for _ in range(n_ml - 1)])
    # N > 1: a filtered picture's reference originals ARRIVE through the picture exchange — the rank next to the one that filters the picture owns them (it ingested them), fills
    # the slot and every rank joins the broadcast; the filtering rank waits for it and runs the search against the RECEIVED planes: a broadcast with a consumer
    mex, mex_count, mex_last = None, [0], {}
    if mc is not None and world > 1:
        mex = sharding.PictureExchange(mc.exchange_shapes(), slots=2, device=hp.device)

    # ---- the reference-picture exchange of the sharded sequence (N > 1): ring of two reconstructed pictures (luma + 2 chroma planes with margins)
    ex = None
    if world > 1:
        m = 80
        shp = (height + 2 * m, ((width + 2 * m + 7) // 8) * 8)
        ex = sharding.PictureExchange([shp, (shp[0] // 2, shp[1] // 2), (shp[0] // 2, shp[1] // 2)], slots=2, device=hp.device)
        ex.publish(0, 0)
    step_no = [0]                                        # this rank's step count k; it replays position step_of_rank( k ) of the cycle
    ex_count = [0]

    with_mctf = [False]

    def step():
        k = step_no[0]
        s = step_of_rank(k, rank, world)
        if with_mctf[0] and mex is None:
            mc.issue_step(s)                             # (first: the filtered picture's search + filter run beside this and the following pictures' lists)
        elif with_mctf[0]:
            for r2 in range(world):                      # every rank walks the same list of this step's exchanges (collectives in one order)
                job = BM.job_of_step(step_of_rank(k, r2, world))
                if job is None:
                    continue
                e, owner = mex_count[0], (r2 + 1) % world
                mex_count[0] += 1
                slot = mex.slot(e)
                lane_stream = [mc.lane.stream] if hasattr(mc.lane, "stream") else []
                if rank == owner:
                    torch.cuda.current_stream().wait_stream(mex.stream)      # (the slot's previous broadcast has left it)
                    for st in lane_stream:
                        torch.cuda.current_stream().wait_stream(st)          # (and its previous consumer has read it)
                    mc.fill_slot(job, slot)
                mex.publish(e, owner, readers=lane_stream)
                if rank == r2:
                    mex.wait(e, lane_stream or None)
                    mc.issue_from_slot(job, slot)
                    mex_last[job[1]] = slot
                else:
                    mex.pending.pop(e, None)
        if ex is not None:
            if k % args.exchange_every == 0 and args.exchange_every < (1 << 29):
                e = k // args.exchange_every
                ex.publish(e + 1, (e + 1) % world, readers=streams if lanes else ())      # the next reference picture is in flight while this picture's launches run
                ex_count[0] += 1
                ex.wait(e, streams if lanes else None)
        wl = workloads[layer_of_step(s)]
        if lanes:
            wl.run_lanes()
        else:
            wl.run()
        step_no[0] = k + 1

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # untimed settling (a freshly acquired box stalls once, a few milliseconds into its first multi-queue phase): whole GOP cycles until two agree
    if world == 1:
        prev = None
        for _ in range(40):
            tb = time.perf_counter()
            for _ in range(32):
                step()
            torch.cuda.synchronize()
            cur = time.perf_counter() - tb
            if prev is not None and abs(cur - prev) < 0.15 * min(cur, prev):
                break
            prev = cur
    else:
        for _ in range(64):
            step()
        torch.cuda.synchronize()

    # ---- THE timed region: exactly `steps` steps between barrier + synchronize on both sides, max over ranks
    step_no[0] = 0
    ex_before = ex_count[0]
    sharding.barrier()
    torch.cuda.synchronize()
    enq = [0.0] * (steps + 1)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        enq[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    enq[0] = t0
    enq_us = sorted(1e6 * (b - a) for a, b in zip(enq[:-1], enq[1:]))
    dt = sharding.max_over_ranks(dt_local, device="cuda")
    ex_count_timed = ex_count[0] - ex_before

    # ---- the same K steps WITH leg C: the steps that replay the POC-32 / 16 / 8 / 24 pictures also queue that picture's MCTF (4 / 4 / 2 / 2 motion estimations + the filter
    #      of Y, U, V) on the MCTF lanes — `value_with_mctf`; and whole GOP cycles (the cadence's stable figure: 12 estimations + 4 filters per 32 steps)
    mctf_region = None
    if mc is not None and world > 1:
        # N > 1: K steps with leg C, references through the exchange (barrier + max over ranks like the headline region); then the parity of what was computed from RECEIVED planes
        with_mctf[0] = True
        step_no[0] = 0
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        sharding.barrier()
        step_no[0] = 0
        n_before = mex_count[0]
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        sharding.barrier()
        dtm = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        with_mctf[0] = False
        par = mc.exchange_parity(mex_last)
        bad = sharding.sum_over_ranks(par["mismatches"] + (1 if par["corruption_detected"] is False else 0), device="cuda")
        checked = sharding.sum_over_ranks(par["jobs_checked"], device="cuda")
        detected = sharding.sum_over_ranks(1 if par["corruption_detected"] else 0, device="cuda")
        mctf_region = {"value": steps * world / dtm, "ms_per_step": 1000.0 * dtm / steps, "steps": steps, "exchanges_in_the_timed_steps": mex_count[0] - n_before,
                       "bytes_per_exchange": int(sum(p.numel() * 2 for p in mex.slots[0])),
                       "parity_exchange": {"status": "bit-exact" if bad == 0 and checked > 0 else ("not checked" if checked == 0 else "MISMATCH"), "jobs_checked_over_ranks": int(checked),
                                           "ranks_that_detected_a_corrupted_slot": int(detected),
                                           "what": "motion fields computed from the RECEIVED reference planes == the fields from the rank's own copies; self-test: part of a received "
                                                   "plane inverted -> the fields change (the consumer reads the slot)"},
                       "note": "N > 1: the K steps with north-star leg C at the GOP's cadence; a filtered picture's reference originals are broadcast by the neighbouring rank (their "
                               "owner) and the filtering rank searches against the received planes — every broadcast has one consumer; `value` is the same steps without leg C"}
    if mc is not None and world == 1:
        with_mctf[0] = True
        step_no[0] = 0
        for _ in range(32):
            step()
        torch.cuda.synchronize()
        step_no[0] = 0
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t1
        step_no[0] = 0
        ncyc = 3 if width <= 1920 else 2
        t1 = time.perf_counter()
        for _ in range(32 * ncyc):
            step()
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t1) / ncyc
        with_mctf[0] = False
        # the same whole cycles WITHOUT leg C (same loop, same order): what the cadence adds to a GOP cycle
        step_no[0] = 0
        t1 = time.perf_counter()
        for _ in range(32 * ncyc):
            step()
        torch.cuda.synchronize()
        dtc0 = (time.perf_counter() - t1) / ncyc
        mctf_region = {"value": steps / dtm, "ms_per_step": 1000.0 * dtm / steps, "steps": steps,
                       "mctf_jobs_in_the_timed_steps": [BM.job_of_step(i)[1] for i in range(steps) if BM.job_of_step(i)],
                       "gop_cycle": {"value": 32.0 / dtc, "ms_per_step": 1000.0 * dtc / 32.0, "ms_per_cycle": 1000.0 * dtc, "ms_per_cycle_without_mctf": 1000.0 * dtc0,
                                     "value_without_mctf": 32.0 / dtc0, "cycles": ncyc,
                                     "note": "whole GOP cycles of 32 steps in step order: 12 motion estimations (4 / 4 / 2 / 2 references) + 4 bilateral filters of Y, U, V per cycle, the four jobs dealt over the two MCTF lanes"},
                       "note": "the timed region's K steps again with north-star leg C issued at the GOP's cadence (tools/bench_mctf.py: the steps replaying POC 32 / 16 / 8 / 24 queue that picture's "
                               "MCTF search + filter on the two MCTF lanes (independent pictures' jobs in turn); originals of the same clip, resident); `value` is the same steps without it"}

    # extras (not `value`): the same K steps serialized on one stream; per layer, the multi-stream time of one picture (-> the GOP-weighted rate); N > 1: without the picture exchange
    serial = None
    if lanes:
        t1 = time.perf_counter()
        for i in range(steps):
            workloads[layer_of_step(step_of_rank(i, rank, world))].run()
        torch.cuda.synchronize()
        dts = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        serial = {"value": steps * world / dts, "unit": "frames/s", "ms_per_step": 1000.0 * dts / steps, "note": "same pictures, every launch on ONE stream (no picture exchange); not the headline value"}
    layer_ms = {}
    if world == 1:
        for l, wl in workloads.items():
            n = 12 if l else 4
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                wl.run_lanes() if lanes else wl.run()
            torch.cuda.synchronize()
            layer_ms[l] = 1000.0 * (time.perf_counter() - t1) / n
    # N > 1: the same steps (a) without the picture exchange = kernel scaling alone, (b) with ONE broadcast per GOP cycle of 32 steps and rank = the cadence when every rank
    # works on its own GOP chunk and only the chunk's key picture crosses ranks (= the timed region above at the default --exchange-every 32), (c) with a broadcast every 2 steps =
    # the worst case: the pictures of one GOP spread over the ranks, every reference picture of the cycle crosses
    no_exchange = per_cycle = every_ref = None
    if ex is not None:
        def timed(every):
            sharding.barrier()
            torch.cuda.synchronize()
            step_no[0], keep = 0, args.exchange_every
            args.exchange_every = every
            t1 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            sharding.barrier()
            args.exchange_every = keep
            return sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        dtn = timed(1 << 30)
        no_exchange = {"value": steps * world / dtn, "unit": "frames/s", "ms_per_step": 1000.0 * dtn / steps, "note": "same steps without the reference-picture broadcast: kernel scaling alone"}
        dtc = timed(32)
        per_cycle = {"value": steps * world / dtc, "unit": "frames/s", "ms_per_step": 1000.0 * dtc / steps, "every_steps": 32,
                     "note": "same steps with one reference-picture broadcast per GOP cycle (32 steps) and rank: the cadence of N ranks working on their own GOP chunks"}
        dtr = timed(2)
        every_ref = {"value": steps * world / dtr, "unit": "frames/s", "ms_per_step": 1000.0 * dtr / steps, "every_steps": 2,
                     "note": "same steps with a reference-picture broadcast every 2 steps: the pictures of ONE GOP spread over the ranks, all 16 reference pictures of the cycle cross (worst case)"}
    # (the per-kernel pass runs AFTER the timed region: its serialized launches and host synchronisations leave idle gaps in which the GPU's clocks fall back, and a timed region of
    #  K = 20 steps is 1.4 ms — too short to ramp them up again: a run that timed right behind this pass read 80.5 us per step where the per-layer times said 69)
    # ---- per-kernel durations: every layer's launches serialized on one stream, HIP events around each kernel (inside the library for the plan's kernels); outside the timed region
    per_layer = {}
    for layer, wl in workloads.items():
        hp.me_plan_set_timing(wl.plan, True)
        acc = {"ME_stage": 0.0, "ME_int": 0.0, "ME_item": 0.0, "TU": 0.0, "DMVR": 0.0}
        reps = 8
        for it in range(reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            wl.run_me()
            e[0].record()
            wl.run_tu()
            e[1].record()
            wl.run_dmvr()
            e[2].record()
            torch.cuda.synchronize()
            t = hp.me_plan_last_times(wl.plan)
            if it == 0:
                continue
            acc["ME_stage"] += t[0]
            acc["ME_int"] += t[1] + t[2]
            acc["ME_item"] += t[3]
            acc["TU"] += e[0].elapsed_time(e[1])
            acc["DMVR"] += e[1].elapsed_time(e[2])
        hp.me_plan_set_timing(wl.plan, False)
        per_layer[layer] = {k: v / reps for k, v in acc.items()}

    # ---- leg C per class: HIP events inside the library around every launch of a motion estimation (vvhip_mctf_set_timing) + around the three filter launches, per job, serialized;
    #      the scored candidates and their algorithmic bytes from the library's counters (vvhip_mctf_set_stats)
    mctf_cls = None
    if mc is not None and world == 1:
        lane = mc.lane
        cyc = {"MCTF_search": 0.0, "MCTF_nb": 0.0, "MCTF_fix": 0.0, "MCTF_apply": 0.0, "MCTF_rest": 0.0}
        reps = 4
        lane.mctf_set_timing(True)
        for job in BM.JOBS:
            for it in range(reps + 1):
                mc.issue(job, apply=False)
                t = lane.mctf_last_times()
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                st = getattr(lane, "stream", None) or torch.cuda.current_stream()
                e[0].record(st)
                mc.issue(job, me=False)
                e[1].record(st)
                torch.cuda.synchronize()
                if it == 0:
                    continue
                cyc["MCTF_search"] += t[0] / reps
                cyc["MCTF_nb"] += t[1] / reps
                cyc["MCTF_fix"] += t[2] / reps
                cyc["MCTF_rest"] += (t[3] + t[4]) / reps
                cyc["MCTF_apply"] += e[0].elapsed_time(e[1]) / reps
        lane.mctf_set_timing(False)
        mc.count()
        mctf_cls = cyc

    if rank != 0:
        return None, workloads, None

    for wl in workloads.values():
        wl.update_tu_alg_bytes()                         # (sparse-output contract: 2 w h + 24 for a TU whose levels are all zero — from the statistics of the runs above)
    frames = steps * world
    pairs_by_layer = {l: workloads[l].pic.sample_pairs for l in workloads}
    wsum = float(sum(GOP_WEIGHT.values()))
    lay_seq = [layer_of_step(step_of_rank(i, rank, world)) for i in range(steps)]
    out = {
        "value": frames / dt, "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": 1000.0 * dt / steps,
        "host_enqueue_us_per_step": {"p50": round(enq_us[len(enq_us) // 2], 1), "p90": round(enq_us[int(len(enq_us) * 0.9)], 1), "max": round(enq_us[-1], 1),
                                     "drain_ms_after_last_enqueue": round(1000.0 * (t0 + dt_local - enq[-1]), 3)},
        "config": {"workload": "%shot-path work lists RECORDED from the reference encoder (%dx%d 10-bit synthetic clip, 65 frames, preset faster; one picture per temporal layer), "
                               "replayed resident in HBM: a step = one picture's integer-ME SAD windows + sub-pel SATD stages + table calls + fused TU transform/quant lists + DMVR lists on 5 HIP streams; "
                               "steps follow the GOP's 1:1:2:4:8:16 layer mix from the key picture on" % ("BASELINE configs[1]: " if ( width, height ) == ( 1920, 1080 ) else ( "BASELINE configs[2]'s geometry: " if ( width, height ) == ( 3840, 2160 ) else "(test size) " ), width, height),
                   "layer_pocs": {str(l): p for l, p in LAYER_POCS.items()},
                   "pictures_per_32_steps_by_layer": {str(l): GOP_WEIGHT[l] for l in GOP_WEIGHT},
                   "pictures_in_the_timed_steps_by_layer": {str(l): lay_seq.count(l) for l in range(6)},
                   "sample_pairs_per_frame": int(sum(GOP_WEIGHT[l] * pairs_by_layer[l] for l in pairs_by_layer) / wsum),
                   "sample_pairs_per_frame_by_layer": {str(l): int(v) for l, v in pairs_by_layer.items()},
                   "sample_pairs_note": "counted like the reference's own counter (CommonLib/SearchSpaceCounter.cpp:106-165 via RdCost.cpp:150-156: w x h per table call, DMVR excluded): "
                                        "%.1f x 1.5 W H GOP-weighted" % (sum(GOP_WEIGHT[l] * pairs_by_layer[l] for l in pairs_by_layer) / wsum / (1.5 * width * height)),
                   "coefficients_per_frame": int(sum(GOP_WEIGHT[l] * workloads[l].tu_coefficients for l in workloads) / wsum),
                   "work_per_layer": {str(l): {"me_calls": int(workloads[l].pic.me.size), "integer_candidates": int(workloads[l].plan_cands.size), "integer_positions_distinct": workloads[l].distinct_positions,
                                               "subpel_stages": int(workloads[l].stage_jobs.size), "subpel_positions": int(workloads[l].stage_evaluated.sum()), "table_calls": int(workloads[l].items.size), "tus": int(sum(g["n"] for g in workloads[l].tu_groups)),
                                               "dmvr_subblocks": int(sum(g["n"] for g in workloads[l].dmvr_groups)), "plan": workloads[l].me_info} for l in workloads},
                   "recorded_calls_outside_the_lists": 0,
                   "tu_outputs": ("sparse (vvhip_tu_set_sparse_outputs): a TU whose levels are all zero gets its statistics only, as the reference's caller reads neither its levels nor its "
                                  "reconstruction (InterSearch.cpp:3696-3714); all-zero TUs by layer: %s" % {str(l): workloads[l].tu_zero_share for l in workloads})
                                 if all(getattr(workloads[l], "tu_sparse", False) for l in workloads) else "dense: every output of every TU is written",
                   "subpel_candidates_per_block": round(float(np.mean([workloads[l].stage_evaluated.sum() / max(1, workloads[l].pic.me.size) for l in workloads if workloads[l].pic.me.size])), 2),
                   "launches_per_frame": "motion-search plan (refinement stages per tap set + integer windows + table calls) + 1 TU launch (+ one per rectangular TU shape) + 0-1 DMVR launch per reference pair",
                   "schedule": "workgroups of every launch in XCD-band order: XCD x (workgroup index mod 8) takes the x-th contiguous eighth of each class in picture order ($VVHIP_ME_XCD_BAND=0: heaviest first)",
                   "hip_streams": len(lanes) if lanes else 1, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "recording": rec_info,
                   "sharding": "one picture per rank and step: rank r replays position k + 32 r / N of the cycle at its step k (N different pictures of one sequence at any time, the same layer mix per rank), no data-path collective"
                               + (", reconstructed picture (luma + chroma, %.1f MB) RCCL-broadcast from its owner every %d step(s) inside the timed region, overlapped with the launches"
                                  % (sum(p.numel() * 2 for p in ex.slots[0]) / 1e6, args.exchange_every) if ex is not None else "")},
    }
    if layer_ms:
        ms = sum(GOP_WEIGHT[l] * layer_ms[l] for l in layer_ms) / wsum
        out["gop_weighted"] = {"value": 1000.0 / ms, "unit": "frames/s", "ms_per_picture_by_layer": {str(l): round(v, 4) for l, v in layer_ms.items()},
                               "note": "per layer: the picture's launches on the five streams, back to back, 12 times (4 for the intra picture); weighted 1 : 1 : 2 : 4 : 8 : 16 — what `value` converges to over whole GOP cycles"}
    if serial:
        out["single_stream"] = serial
    if ex is not None:
        import torch.distributed as dist
        backend = dist.get_backend() if dist.is_initialized() else "none"
        out["exchange"] = {"pictures": ex_count_timed, "bytes_per_picture": int(sum(p.numel() * 2 for p in ex.slots[0])), "collective": "broadcast", "backend": backend, "overlapped": True, "every_steps": args.exchange_every}
        if backend == "nccl":          # (a fabric number only on RCCL: over gloo the broadcast goes through host memory and says nothing about xGMI)
            out["exchange"]["exchange_ms_per_picture"] = round(1000.0 * (dt - (steps * world / no_exchange["value"])) / max(1, ex_count_timed), 4) if no_exchange else None
        out["no_exchange"], out["exchange_per_gop_cycle"], out["exchange_every_reference"] = no_exchange, per_cycle, every_ref
        out["n_gpu_forms"] = "value: --exchange-every %d (default 32 = one GOP chunk per rank, the chunk's key picture crosses ranks: north_star's sharding by GOPs, DESIGN 7); " \
                             "exchange_per_gop_cycle = the same cadence timed again; exchange_every_reference = a broadcast every 2 steps (one GOP spread over the ranks: worst case, " \
                             "`value`'s cadence up to round 5); no_exchange = kernel scaling alone" % args.exchange_every

    # ---- kernels
    alg = {l: {"ME_stage": None, "ME_int": None, "ME_item": None, "TU": workloads[l].alg_bytes_tu, "DMVR": workloads[l].alg_bytes_dmvr} for l in workloads}
    for l, wl in workloads.items():
        alg[l].update(wl.alg_bytes_by_kernel)
    kern = {}
    for k in ("ME_stage", "ME_int", "ME_item", "TU", "DMVR"):
        ms = sum(GOP_WEIGHT[l] * per_layer[l][k] for l in per_layer) / wsum
        ab = sum(GOP_WEIGHT[l] * (alg[l][k] or 0) for l in per_layer) / wsum
        ub = sum(GOP_WEIGHT[l] * ((workloads[l].unique_bytes_by_kernel or {}).get(k) or 0) for l in per_layer) / wsum
        kern[k] = {"kernel": KERNEL_NAMES[k], "avg_ms_per_picture": ms,
                   **({"per_position_bytes_per_picture": int(sum(GOP_WEIGHT[l] * workloads[l].alg_bytes_int_per_position for l in per_layer) / wsum),
                       "all_candidates_bytes_per_picture": int(sum(GOP_WEIGHT[l] * workloads[l].alg_bytes_int_all_candidates for l in per_layer) / wsum)} if k == "ME_int" else {}), "ms_by_layer": {str(l): round(per_layer[l][k], 4) for l in per_layer}, "alg_bytes_per_picture": int(ab),
                   "unique_bytes_per_picture": int(ub), "nominal_alg_GBps": (ab / (ms * 1e-3) / 1e9) if ms > 0 else None}
    if mctf_cls is not None:
        for k in ("MCTF_search", "MCTF_nb", "MCTF_fix", "MCTF_apply"):
            ms = mctf_cls[k] / 32.0
            ab = mc.alg_bytes_per_cycle[k] / 32.0
            kern[k] = {"kernel": kernel_label(k), "avg_ms_per_picture": ms, "ms_per_gop_cycle": round(mctf_cls[k], 4), "alg_bytes_per_picture": int(ab),
                       "unique_bytes_per_picture": int(mc.unique_bytes_per_cycle[k] / 32.0), "nominal_alg_GBps": (ab / (ms * 1e-3) / 1e9) if ms > 0 else None}
        out["with_mctf"] = mctf_region
        out["with_mctf"]["ms_per_gop_cycle_by_class_serialized"] = {k: round(v, 4) for k, v in mctf_cls.items()}
        out["with_mctf"]["scored_candidates_per_gop_cycle"] = mc.candidates_per_cycle
        out["with_mctf"]["alg_bytes_per_gop_cycle"] = mc.alg_bytes_per_cycle
        out["with_mctf"]["per_candidate_bytes_per_gop_cycle"] = mc.per_candidate_bytes_per_cycle
        out["with_mctf"]["alg_bytes_note"] = ("SURVEY 8d per SCORED candidate, counted by the kernels (vvhip_mctf_set_stats): integer vector 4 w h; fractional vector (w + 3)(h + 3) 2 + 2 w h "
                                              "(4-tap search filter); the dense integer grids of a block and the refinement rings in 8d's WINDOW form ((w + 2R)^2 2 + 2 w h per block + 8 per position; a ring: "
                                              "(w + 4)(h + 4) 2 + 2 w h + 8 per position: the window is staged once in LDS) — per_candidate_bytes counts every such position at its own "
                                              "4 w h / (w + 3)(h + 3) 2 + 2 w h instead: a work rate (LDS-level reuse), not memory traffic; "
                                              "filter: per block 4 w h + per reference (w + 5)(h + 5) 2 + 24")
        out["_mctf"] = mc
    elif mctf_region is not None:
        out["with_mctf"] = mctf_region
    out["kernels"] = kern
    out["kernels_measured"] = "HIP events on the launch stream around every kernel (inside vvhip_me_plan_run for the plan's kernels), 8 passes per layer with the launches serialized; GOP-weighted"
    return out, workloads, kern


# ---------------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--streams", type=int, default=5, help="HIP streams a picture's independent launch groups are issued on: 5 = refinement stages / integer windows / table calls / TU / DMVR, "
                                                           "3 = motion-search plan / TU / DMVR, 1 = serialized")
    ap.add_argument("--exchange-every", type=int, default=32, help="N > 1: one reference-picture broadcast every this many steps (a step is one picture).  Default 32: every rank works "
                    "on its own GOP chunk and the chunk's key picture is the one reference that crosses ranks; 2 = the pictures of one GOP spread over the ranks (16 of the 32 pictures "
                    "of a random-access GOP cycle, temporal layers 0-4, are references of other pictures) — reported as exchange_every_reference either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-mctf", action="store_true")
    ap.add_argument("--no-4k", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-medium", action="store_true", help="skip the replay of the preset-medium 4K picture (BASELINE configs[3]'s lists)")
    ap.add_argument("--quick", action="store_true", help="kernel iterations: the headline pass only (= --no-e2e --no-profile --no-4k --no-mctf --no-cpu-baseline --no-medium)")
    ap.add_argument("--profile-md", default=None, help="also write the rocprofv3 summary of this run (kernel table + counters per kernel class) as markdown to this path (the 4K pass: PATH with _4k before the extension)")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 passes of the inner run (kernel trace + one pass per counter)")
    ap.add_argument("--e2e-threads", type=int, default=8)
    ap.add_argument("--e2e-pairs", type=int, default=5)
    ap.add_argument("--e2e-full", action="store_true", help="the T = 1 row of the 3840x2160 encoder run too (about 2 minutes more)")
    ap.add_argument("--detail", default="bench_detail.json", help="file name (under the repository root, and under gpurun_out/ when that exists) of the full result object")
    ap.add_argument("--inner-mctf", action="store_true", help="(internal) the short run rocprofv3 wraps for the MCTF stage: six motion estimations of one 1080p picture against 4 references")
    ap.add_argument("--inner", action="store_true", help="(internal) the short run rocprofv3 wraps: the recorded pictures' launches serialized on one stream, no extras, no output line")
    args = ap.parse_args()
    if args.quick:
        args.no_e2e = args.no_profile = args.no_4k = args.no_mctf = args.no_cpu_baseline = args.no_medium = True

    rank, local_rank, world = sharding.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: vvenc_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.replay import RecordedWorkload
    import bench_encoder as BE
    import bench_profile as BP
    hp = HotPath("cuda:%d" % local_rank)
    if args.inner_mctf:
        wl = BE.Mctf1080(args.width, args.height)
        cur = hp.plane(wl.cur_np, 128)
        refs = [hp.plane(np.roll(wl.ref_np, (k, -2 * k), (0, 1)), 128) for k in range(4)]
        outs, _ = hp.mctf_motion_estimation(cur, refs, wl.bit_depth, 16, 4, args.width >= 1920)
        for _ in range(6):
            hp.mctf_motion_estimation(cur, refs, wl.bit_depth, 16, 4, args.width >= 1920, out=outs)
        torch.cuda.synchronize()
        return
    if args.inner:
        pics, _ = prepare_recordings(args.width, args.height, 65, sorted(LAYER_POCS.values()))
        workloads = {layer: RecordedWorkload(hp, pics[poc], unique_bytes=False) for layer, poc in LAYER_POCS.items()}
        mc = None if args.no_mctf else BM.MctfCadence(hp, args.width, args.height)
        for s in range(args.warmup + args.steps):
            if mc is not None:
                mc.issue_step(s)                         # (serialized on the same stream: the rocprofv3 passes see leg C's kernels at the GOP's cadence)
            workloads[layer_of_step(s)].run()
        torch.cuda.synchronize()
        return

    core, workloads, kern = replay_pass(args, hp, rank, world, args.width, args.height, args.steps, args.warmup)
    inst = None
    if world > 1 and not args.no_e2e:
        try:
            inst = BE.e2e_instances(rank, local_rank, world)
        except Exception as e:
            inst = {"error": str(e)[:300]} if rank == 0 else None
    if rank != 0:
        return
    out = {"metric": "frames/sec + bit-exact vs CPU, 1080p/4K 10-bit preset=faster at 1/2/4/8 GPU", "value": core["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": core["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16", "data": "synthetic",
           "value_note": "pictures/s of the north-star hot path (recorded lists resident in HBM) — the path's throughput, NOT encoder fps; the encoder's frames/s with the device stages is e2e.hip_fps"}
    mc = core.pop("_mctf", None)
    out.update({k: v for k, v in core.items() if k not in out})
    if core.get("with_mctf"):
        out["value_with_mctf"], out["ms_per_step_with_mctf"] = core["with_mctf"]["value"], core["with_mctf"]["ms_per_step"]
    if inst is not None:
        out["e2e_instances"] = inst
    can_profile = not args.no_profile and world == 1 and bool(shutil.which("rocprofv3"))

    def profiled(width, height, kern_, workloads_, ms_per_step, md, mc_=None):
        """roofline objects of one resolution from this run's own rocprofv3 passes"""
        live = None
        res = {}
        if can_profile:
            try:
                live = BP.live_profile(width, height)
                res["kernel_trace"] = live["kernel_trace"]
                if live.get("pmc_errors"):
                    res["pmc_errors"] = live["pmc_errors"]
            except Exception as e:
                res["kernel_trace"] = {"error": str(e)[:300]}
        wsum = float(sum(GOP_WEIGHT.values()))
        uniq = {k: sum(GOP_WEIGHT[l] * ((workloads_[l].unique_bytes_by_kernel or {}).get(k) or 0) for l in workloads_) / wsum for k in KERNEL_NAMES}
        if mc_ is not None:
            uniq.update({k: v / 32.0 for k, v in mc_.unique_bytes_per_cycle.items()})
        roof, allk, checks = BP.roofline_objects(kern_, live, calib, uniq, ms_per_step, md)
        res["roofline"], res["roofline_checks"] = roof, checks
        if allk:
            res["roofline_all_kernels"] = allk
        return res

    calib = BP.counter_calibration() if can_profile else {"measured": False, "factors": {"rows16": 2.0, "stream16": 2.0, "store8": 1.0}, "how": "not run", "pattern_of_class": BP.FETCH_PATTERN}
    out["counter_calibration"] = calib
    # (with leg C's classes in the table the step time of the cross-check is the with-MCTF GOP cycle's)
    step_ms = core["with_mctf"]["gop_cycle"]["ms_per_step"] if (core.get("with_mctf") or {}).get("gop_cycle") else core["ms_per_step"]
    out.update(profiled(args.width, args.height, kern, workloads, step_ms, args.profile_md, mc))

    if not args.no_parity:
        try:
            out["parity"] = parity_check(workloads)
        except Exception as e:
            out["parity"] = {"status": "not checked", "error": str(e)[:300]}
    if mc is not None and not args.no_parity:
        try:
            out["parity_mctf"] = mc.parity()
        except Exception as e:
            out["parity_mctf"] = {"status": "not checked", "error": str(e)[:300]}
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(workloads)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
        if mc is not None:
            try:
                out["cpu_baseline"]["mctf"] = BM.cpu_baseline_mctf(args.width, args.height, out["cpu_baseline"].get("cores") or os.cpu_count())
            except Exception as e:
                out["cpu_baseline"]["mctf"] = {"error": str(e)[:200]}
    mc = None

    # ---- the 3840x2160 replay (BASELINE: "1080p/4K"): the same pass on lists recorded from the 4K x 65 encode
    if world == 1 and not args.no_4k and (args.width, args.height) == (1920, 1080):
        del workloads
        torch.cuda.empty_cache()
        try:
            c4, w4, k4 = replay_pass(args, hp, rank, world, 3840, 2160, args.steps, max(4, args.warmup // 2))
            out["value_4k"], out["ms_per_step_4k"], out["steps_4k"] = c4["value"], c4["ms_per_step"], c4["steps"]
            out["config_4k"] = c4["config"]
            mc4 = c4.pop("_mctf", None)
            if c4.get("with_mctf"):
                out["value_4k_with_mctf"], out["ms_per_step_4k_with_mctf"] = c4["with_mctf"]["value"], c4["with_mctf"]["ms_per_step"]
            for k in ("gop_weighted", "single_stream", "kernels", "with_mctf"):
                if k in c4:
                    out[k + "_4k"] = c4[k]
            md4 = (os.path.splitext(args.profile_md)[0] + "_4k" + os.path.splitext(args.profile_md)[1]) if args.profile_md else None
            step_ms4 = c4["with_mctf"]["gop_cycle"]["ms_per_step"] if (c4.get("with_mctf") or {}).get("gop_cycle") else c4["ms_per_step"]
            for k, v in profiled(3840, 2160, k4, w4, step_ms4, md4, mc4).items():
                out[k + "_4k"] = v
            if mc4 is not None and not args.no_parity:
                try:
                    out["parity_mctf_4k"] = mc4.parity(pocs=(8,))          # (one picture: two 3840x2160 motion estimations + its filter on the host take ~1 s)
                except Exception as e:
                    out["parity_mctf_4k"] = {"status": "not checked", "error": str(e)[:300]}
            if not args.no_parity:
                try:
                    out["parity_4k"] = parity_check(w4)
                except Exception as e:
                    out["parity_4k"] = {"status": "not checked", "error": str(e)[:300]}
            if not args.no_cpu_baseline:
                try:
                    out["cpu_baseline_4k"] = cpu_baseline(w4, passes=3)
                except Exception as e:
                    out["cpu_baseline_4k"] = {"value": None, "sample": "failed: %r" % (e,)}
                if mc4 is not None:
                    try:
                        out["cpu_baseline_4k"]["mctf"] = BM.cpu_baseline_mctf(3840, 2160, out["cpu_baseline_4k"].get("cores") or os.cpu_count())
                    except Exception as e:
                        out["cpu_baseline_4k"]["mctf"] = {"error": str(e)[:200]}
            mc4 = None
            del w4
            torch.cuda.empty_cache()
        except Exception as e:
            out["value_4k"] = None
            out["error_4k"] = str(e)[:400]

    if world == 1 and not args.no_4k and not args.no_medium and (args.width, args.height) == (1920, 1080):
        try:
            out["config3_medium_4k"] = BE.replay_medium_4k(hp)
        except Exception as e:
            out["config3_medium_4k"] = {"error": str(e)[:300]}
    if not args.no_mctf and world == 1:
        try:
            import bench_synthetic as BS
            m, _ = BS.mctf_stage(hp, BE.Mctf1080(args.width, args.height), 4)
            out["mctf"] = {k: v for k, v in m.items() if k != "me_frac_hbm_unique"}
            if can_profile:
                try:
                    out["mctf"]["profile"] = BP.mctf_profile(args)
                except Exception as e:
                    out["mctf"]["profile"] = {"error": str(e)[:200]}
            if not args.no_4k:
                m4, _ = BS.mctf_stage(hp, BE.Mctf1080(3840, 2160), 4, reps=3)
                out["mctf_4k"] = {k: v for k, v in m4.items() if k != "me_frac_hbm_unique"}
        except Exception as e:
            out["mctf"] = {"error": str(e)[:300]}
    if not args.no_e2e and world == 1:
        from bench_reference import host_cpu_info, usable_cores
        cores = usable_cores(host_cpu_info())
        others = [t for t in (1, cores) if t != args.e2e_threads]
        try:
            out["e2e"] = BE.e2e_encoder(1920, 1080, 65, args.e2e_threads, args.e2e_pairs, other_threads=others)
        except Exception as e:
            out["e2e"] = {"error": str(e)[:300]}
        if not args.no_4k:
            try:
                out["e2e_4k"] = BE.e2e_encoder(3840, 2160, 65, args.e2e_threads, 3, other_threads=others if args.e2e_full else [t for t in others if t != 1])
            except Exception as e:
                out["e2e_4k"] = {"error": str(e)[:300]}
    bench_line.emit(out, ROOT, args.detail)


if __name__ == "__main__":
    main()
