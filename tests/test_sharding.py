"""CPU tier, world_size 2, gloo: the multi-process path bench.py uses for N>1 GPUs (picture sharding, picture broadcast,
barrier + max-over-ranks timing).  The compute itself needs a GPU and is covered by -m gpu tests; this checks the plumbing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % k))) for k in range(2)]
    assert sorted(res[0]["mine"] + res[1]["mine"]) == list(range(13))
    assert res[0]["dt_max"] == res[1]["dt_max"] and res[0]["total_frames"] == 13
    # the two ranks together produced exactly the motion fields one process produces (keys owned and published round-robin, dependent pictures dealt round-robin)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    single = dist_worker.sharded_motion_fields(0, 1)
    merged = dict(res[0]["fields"]); merged.update(res[1]["fields"])
    assert set(res[0]["fields"]).isdisjoint(res[1]["fields"]) and len(res[0]["fields"]) > 0 and len(res[1]["fields"]) > 0
    assert merged == single and len(single) == 6, (merged, single)


def test_single_process_defaults():
    from vvenc_amd import sharding
    assert sharding.frames_of_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert sharding.max_over_ranks(1.5) == 1.5 and sharding.sum_over_ranks(3) == 3.0
