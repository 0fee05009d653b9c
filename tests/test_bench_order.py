"""CPU tier: the replay order of bench.py (host logic only: no device, no recording).

  * STEP_LAYERS holds exactly the GOP's layer mix (1 x TL0, 1 x TL1, 2 x TL2, 4 x TL3, 8 x TL4, 16 x TL5 per 32 pictures) and starts at the key picture;
  * every prefix is close to the mix (low discrepancy): what makes a driver run of K = 20 steps representative of the GOP;
  * with N ranks every rank replays a window of the same cycle at its own offset: N different positions at any time, (nearly) the same layer mix per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_step_layers_hold_the_gop_mix_and_start_at_the_key_picture():
    import bench
    assert len(bench.STEP_LAYERS) == 32 and bench.STEP_LAYERS[0] == 0
    assert {l: bench.STEP_LAYERS.count(l) for l in range(6)} == {0: 1, 1: 1, 2: 2, 3: 4, 4: 8, 5: 16} == bench.GOP_WEIGHT
    assert [bench.layer_of_step(s) for s in range(32, 64)] == list(bench.STEP_LAYERS)


def test_every_prefix_is_close_to_the_mix():
    import bench
    share = {l: bench.GOP_WEIGHT[l] / 32.0 for l in range(6)}
    for n in range(8, 65):
        seq = [bench.layer_of_step(s) for s in range(n)]
        for l in range(6):
            assert abs(seq.count(l) - n * share[l]) <= 1.0, (n, l, seq.count(l), n * share[l])
    # the driver's K = 20: the intra picture is in, the mix is 1 / 0 / 1 / 3 / 5 / 10
    seq = [bench.layer_of_step(s) for s in range(20)]
    assert [seq.count(l) for l in range(6)] == [1, 0, 1, 3, 5, 10]
    # time-weighted with the per-layer costs of the recorded 1080p lists (us per picture on five streams, profiles/bench_r04_c.json): within 2 % of the cycle's mean
    cost = {0: 183.3, 1: 102.9, 2: 80.3, 3: 68.7, 4: 76.6, 5: 76.5}
    mean = sum(bench.GOP_WEIGHT[l] * cost[l] for l in cost) / 32.0
    assert abs(sum(cost[l] for l in seq) / 20.0 / mean - 1.0) < 0.02


def test_rank_windows_are_different_positions_with_the_same_mix():
    import bench
    for world in (2, 4, 8):
        for k in range(40):
            pos = {bench.step_of_rank(k, r, world) % 32 for r in range(world)}
            assert len(pos) == world                                   # N different pictures of the cycle at any time
        for r in range(world):
            seq = [bench.layer_of_step(bench.step_of_rank(k, r, world)) for k in range(32)]
            assert {l: seq.count(l) for l in range(6)} == bench.GOP_WEIGHT      # a whole cycle per rank: the exact mix
            top = [bench.layer_of_step(bench.step_of_rank(k, r, world)) for k in range(20)]
            assert top.count(5) == 10 and top.count(0) <= 1
