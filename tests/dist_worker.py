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
    # the sharded path itself on a small sequence: key pictures are produced by their owners and published through the PictureExchange ring, the pictures between two
    # keys are dealt round-robin and matched against both keys with the CPU oracle's MCTF motion estimation (the same call sequence bench.py's N>1 step runs on GPUs)
    fields = sharded_motion_fields(rank, world)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "mine": mine, "dt_max": dt_max, "total_frames": total_frames, "fields": fields}, f)
    sharding.barrier()


SEQ = dict(n_frames=9, gop=4, w=96, h=80, pad=16)


def sequence_picture(p):
    """picture p of the test sequence (deterministic; any rank can make any picture, only owners do)"""
    import numpy as np
    yy, xx = np.mgrid[0:SEQ["h"], 0:SEQ["w"]]
    rng = np.random.default_rng(100 + p)
    y = 512 + 200 * np.sin((xx + 2 * p) / 9.0) * np.cos((yy - p) / 7.0) + rng.normal(0, 6, xx.shape)
    return np.clip(y, 0, 1023).astype(np.int16)


def sharded_motion_fields(rank, world):
    import hashlib
    import numpy as np
    from oracle.oracle import Oracle
    orc = Oracle()
    pad = SEQ["pad"]
    shape = (SEQ["h"] + 2 * pad, SEQ["w"] + 2 * pad)
    ex = sharding.PictureExchange([shape], slots=3, device="cpu")

    def produce_key(p, slot):
        slot[0].zero_()
        slot[0][pad:pad + SEQ["h"], pad:pad + SEQ["w"]] = torch.from_numpy(sequence_picture(p))

    def process(q, prev_slot, next_slot):
        cur = sequence_picture(q)
        out = []
        for s in (prev_slot, next_slot):
            ref = s[0][pad:pad + SEQ["h"], pad:pad + SEQ["w"]].numpy().copy()
            f = orc.mctf_me(cur, ref, 10, 8, 4, False)[4]
            out.append(hashlib.md5(b"".join(np.ascontiguousarray(f[k]).tobytes() for k in ("x", "y", "error", "rmsme"))).hexdigest())
        return out

    res = sharding.run_sharded_gops(SEQ["n_frames"], SEQ["gop"], rank, world, ex, produce_key, process)
    return {str(k): v for k, v in res.items()}


if __name__ == "__main__":
    main()
