#!/usr/bin/env python3
"""experiment: how leg C (MCTF at the GOP's cadence) overlaps with the five-stream A + B step.  Whole GOP cycles of 32 steps:
   ab        the recorded pictures' lists on five streams (= bench.py's gop cycle without MCTF)
   mctf      only the cycle's four MCTF jobs, on one lane / two lanes
   both      ab + mctf on one lane / two lanes / graph-launched"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from bench_common import LAYER_POCS, layer_of_step, prepare_recordings  # noqa: E402
import bench_mctf as BM  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.replay import RecordedWorkload
    hp = HotPath("cuda:0")
    pics, _ = prepare_recordings(w, h, 65, sorted(LAYER_POCS.values()))
    wls = {l: RecordedWorkload(hp, pics[p], unique_bytes=False) for l, p in LAYER_POCS.items()}
    lanes = [hp.fork(torch.cuda.Stream()) for _ in range(5)]
    for wl in wls.values():
        wl.bind_lanes(lanes)
    prio = int(os.environ.get("MCTF_PRIO", "0"))
    ml = [hp.fork(torch.cuda.Stream(priority=prio)) for _ in range(2)]
    print("MCTF lane priority", prio, "range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
    mc = BM.MctfCadence(hp, w, h, lane=ml[0])
    for job in BM.JOBS:                       # scratch of both lanes
        mc.issue(job, ctx=ml[0]); mc.issue(job, ctx=ml[1])
    torch.cuda.synchronize()
    graphs = {}
    for k, job in enumerate(BM.JOBS):
        lane = ml[k % 2]
        graphs[job[1]] = (lane, lane.graph_capture(lambda job=job, lane=lane: mc.issue(job, ctx=lane)))

    finish = [""]

    def cycles(ab, mode, n=3):
        def one():
            for s in range(32):
                job = BM.job_of_step(s)
                if job is not None and mode:
                    k = BM.JOBS.index(job)
                    if mode == "one":
                        mc.issue(job, ctx=ml[0])
                    elif mode == "two":
                        mc.issue(job, ctx=ml[k % 2])
                    elif mode == "graph":
                        lane, g = graphs[job[1]]
                        lane.graph_launch(g)
                    elif mode == "me_only":
                        mc.issue(job, ctx=ml[k % 2], apply=False)
                    elif mode == "apply_only":
                        mc.issue(job, ctx=ml[k % 2], me=False)
                if ab:
                    wls[layer_of_step(s)].run_lanes()
        one()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(lanes[0].stream)
        t = time.perf_counter()
        for _ in range(n):
            one()
        th = time.perf_counter() - t
        # when does each side finish?  (an event at the end of every lane's queue against the event recorded before the first launch)
        ends = []
        for l in list(lanes) + list(ml):
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.ExternalStream(int(l.L.vvhip_get_stream(l.ctx) or 0)))      # (the stream the context launches on NOW: graph capture moved the MCTF lanes to their own)
            ends.append(e)
        torch.cuda.synchronize()
        el = [ev0.elapsed_time(e) for e in ends]
        finish[0] = "A+B lanes done at %.3f ms, MCTF lanes at %.3f ms (of %d cycles)" % (max(el[:len(lanes)]), max(el[len(lanes):]), n)
        return 1000.0 * (time.perf_counter() - t) / n, 1000.0 * th / n

    for name, ab, mode in (("ab", True, None), ("mctf one lane", False, "one"), ("mctf two lanes", False, "two"), ("mctf graph (two lanes)", False, "graph"),
                           ("both one lane", True, "one"), ("both two lanes", True, "two"), ("both graph", True, "graph"), ("ab + me only", True, "me_only"), ("ab + apply only", True, "apply_only"),
                           ("ab", True, None)):
        ms, host = cycles(ab, mode)
        print("%-28s %8.3f ms per GOP cycle   (host enqueue %7.3f ms)   %8.0f pictures/s   %s" % (name, ms, host, 32000.0 / ms, finish[0]), flush=True)


if __name__ == "__main__":
    main()
