"""worker for tests/test_sharding.py (run under torch.distributed.run, gloo, CPU): exercises the N>1 path of bench.py —
picture ownership, the reference-picture broadcast, barrier + max-over-ranks timing and the rank-0 aggregation."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vvenc_amd import sharding  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = sharding.init("gloo")
    n_frames = 13
    mine = sharding.frames_of_rank(n_frames, rank, world)
    # every frame has exactly one owner
    owned = torch.zeros(n_frames, dtype=torch.int64)
    owned[mine] = 1
    torch.distributed.all_reduce(owned)
    assert bool((owned == 1).all()), owned
    assert all(sharding.owner_of(f, world) == rank for f in mine)
    # picture broadcast: the owner of picture 5 fills it, everybody else receives it bit-exactly (int16 plane incl. margin)
    src = sharding.owner_of(5, world)
    g = torch.Generator().manual_seed(5)
    ref = torch.randint(0, 1024, (72 + 32, 128 + 32), generator=g, dtype=torch.int16)
    pic = ref.clone() if rank == src else torch.zeros_like(ref)
    sharding.broadcast_picture(pic, src)
    assert torch.equal(pic, ref)
    # timing protocol of bench.py: barrier, timed region, barrier, max over ranks; value = total units / max time
    sharding.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))           # rank 1 is the slow one
    sharding.barrier()
    dt = time.perf_counter() - t0
    dt_max = sharding.max_over_ranks(dt)
    total_frames = sharding.sum_over_ranks(len(mine))
    assert dt_max >= 0.05 * world - 1e-3 and total_frames == n_frames
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "mine": mine, "dt_max": dt_max, "total_frames": total_frames}, f)
    sharding.barrier()


if __name__ == "__main__":
    main()
