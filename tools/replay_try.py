import sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd import recorded as R, replay
hp = HotPath("cuda:0")
pics = R.load_dir(sys.argv[1])
for poc, pic in pics.items():
    t0 = time.time()
    wl = replay.RecordedWorkload(hp, pic)
    t1 = time.time()
    wl.run(); torch.cuda.synchronize()
    chk = wl.check_against_recording()
    # timing
    for _ in range(3): wl.run()
    torch.cuda.synchronize()
    hp.me_plan_set_timing(wl.plan, True)
    parts = []
    for _ in range(5):
        wl.run_me(); torch.cuda.synchronize(); parts.append(hp.me_plan_last_times(wl.plan))
    hp.me_plan_set_timing(wl.plan, False)
    parts = (np.median(np.array(parts), 0) * 1000).round(1).tolist()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    n = 20
    ev[0].record()
    for _ in range(n): wl.run_me()
    ev[1].record()
    for _ in range(n): wl.run_tu()
    ev[2].record()
    for _ in range(n): wl.run_dmvr()
    ev[3].record()
    torch.cuda.synchronize()
    line = json.dumps({"poc": poc, "tl": pic.tlayer, "build_s": round(t1 - t0, 2), "check": chk, "info": wl.me_info, "dropped": wl.items_dropped,
                      "me_parts_us_stage_int_0_item": parts, "us": {"me": 1000 * ev[0].elapsed_time(ev[1]) / n, "tu": 1000 * ev[1].elapsed_time(ev[2]) / n, "dmvr": 1000 * ev[2].elapsed_time(ev[3]) / n},
                      "n": {"int_jobs": int(wl.int_jobs.size), "cands": int(wl.plan_cands.size), "stages": int(wl.stage_jobs.size), "items": int(wl.items.size), "tus": sum(g["n"] for g in wl.tu_groups)}})
    print(line)
    open("/root/repo/gpurun_out/replay_try.jsonl", "a").write(line + "\n")
