#!/usr/bin/env python3
"""experiment: P pictures in flight — the recorded pictures' five launch groups on P independent sets of streams (step s goes to set s mod P) against one set (bench.py's step).
   python tools/pp_overlap.py W H QUEUES            all configurations in one process (round-6 form: lanes = torch streams + the contexts' own: queue sharing confounds it)
   python tools/pp_overlap.py W H QUEUES P PER      ONE configuration per process, lanes on their contexts' own streams (one HIP stream per lane; QUEUES >= P * PER + 2)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[3] if len(sys.argv) > 3 else "8")
import torch  # noqa: E402

from bench_common import LAYER_POCS, layer_of_step, prepare_recordings  # noqa: E402


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.replay import RecordedWorkload
    hp = HotPath("cuda:0")
    pics, _ = prepare_recordings(w, h, 65, sorted(LAYER_POCS.values()))
    wls = {l: RecordedWorkload(hp, pics[p], unique_bytes=False) for l, p in LAYER_POCS.items()}
    one = len(sys.argv) > 5
    for P, per in (((int(sys.argv[4]), int(sys.argv[5])),) if one else ((1, 5), (2, 5), (2, 3), (4, 3), (4, 5), (1, 5))):
        sets = [[hp.fork(None if one else torch.cuda.Stream()) for _ in range(per)] for _ in range(P)]
        calls = [{l: list(wl.bind_lanes(sets[k])) for l, wl in wls.items()} for k in range(P)]

        def cycle():
            for s in range(32):
                for c in calls[s % P][layer_of_step(s)]:
                    c()
        cycle()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            cycle()
        th = time.perf_counter() - t
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t) / 4
        print("P = %d x %d streams (GPU_MAX_HW_QUEUES %s): %7.3f ms per GOP cycle (host %6.3f)  %8.0f pictures/s" % (P, per, os.environ["GPU_MAX_HW_QUEUES"], ms, 1000.0 * th / 4, 32000.0 / ms), flush=True)
        for st in sets:
            for c in st:
                c.close()


if __name__ == "__main__":
    main()
