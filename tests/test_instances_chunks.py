"""CPU tier: the N-instance form of the end-to-end metric (bench.py --gpus N: tools/bench_encoder.e2e_instances) — N encoder instances on GOP chunks of ONE sequence.

  * a chunk encoded next to the other instances (concurrently, with its share of the host's threads, its own device selection in the environment) gives the SAME bitstream as
    that chunk encoded by a single instance on its own: chunk boundaries are closed (every chunk starts with its own intra picture) and nothing leaks between instances;
  * chunk r of the endless clip is frames r * F .. r * F + F - 1 of one sequence: the generator's chunk equals the slice of a longer chunk (pan, object and per-frame noise
    continue across the boundary);
  * the same with the binding installed and the device library's test double injected (tests/sim): --SIMD=HIP production mask per instance == the CPU bitstream per chunk."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VVHIP_ALF_MIN_CTUS_PER_THREAD", "0")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fps  # noqa: E402
import e2e_util  # noqa: E402

W, H, F = 416, 240, 17          # (the generator moves a 128x128 object: the clip is taller than that)
SIM = os.path.join(ROOT, "tests", "sim", "libvvhip_sim.so")


def _spawn(cfg, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.Popen([sys.executable, "-c", e2e_fps.WORKER, json.dumps(cfg)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def _result(p):
    out, err = p.communicate(timeout=900)
    assert p.returncode == 0, err[-2000:]
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_chunks_are_slices_of_one_sequence():
    long = e2e_fps.synth_clip_chunk(W, H, 0, 2 * F)
    for r in range(2):
        part = e2e_fps.synth_clip_chunk(W, H, r * F, F)
        for a, b in zip(long, part):
            assert np.array_equal(a[r * F:(r + 1) * F], b), r


@pytest.mark.ref
def test_concurrent_instances_reproduce_the_single_instance_bitstream_of_each_chunk():
    if not os.path.exists(e2e_util.REF_SO):
        pytest.skip("oracle/_ref/libvvenc_ref.so (the compiled reference) not built")
    for r in range(2):
        e2e_fps.synth_clip_chunk(W, H, r * F, F)                                   # (cached before the instances start, as e2e_instances does)
    # N = 2 instances at the same time, two threads each, each with its own device selection in the environment (what e2e_instances sets per rank)
    procs = [_spawn(dict(w=W, h=H, frames=F, first=r * F, threads=2, mask=0), env={"HIP_VISIBLE_DEVICES": str(r)}) for r in range(2)]
    together = [_result(p) for p in procs]
    # every chunk alone, one instance with all four threads
    alone = [_result(_spawn(dict(w=W, h=H, frames=F, first=r * F, threads=4, mask=0))) for r in range(2)]
    for r in range(2):
        assert together[r]["md5"] == alone[r]["md5"] and together[r]["bytes"] == alone[r]["bytes"], (r, together[r], alone[r])
    assert together[0]["md5"] != together[1]["md5"]                               # (different pictures: the chunks are not copies of each other)


@pytest.mark.sim
def test_concurrent_instances_with_the_binding_reproduce_the_cpu_bitstream_of_each_chunk():
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO) and os.path.exists(SIM)):
        pytest.skip("needs oracle/_ref, bindings/vvenc/_build and tests/sim (python __graft_entry__.py)")
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the test double refuses to stand in for it")
    prod = 16 + 128 + 8192                                                         # --SIMD=HIP's production mask (bench_encoder.e2e_production_mask)
    cpu = [_result(_spawn(dict(w=W, h=H, frames=F, first=r * F, threads=2, mask=0))) for r in range(2)]
    procs = [_spawn(dict(w=W, h=H, frames=F, first=r * F, threads=2, mask=prod), env={"LD_PRELOAD": SIM, "HIP_VISIBLE_DEVICES": str(r)}) for r in range(2)]
    hip = [_result(p) for p in procs]
    for r in range(2):
        assert hip[r]["calls"][9] >= 1 and hip[r]["calls"][21] >= 1, hip[r]["calls"]          # MCTF filter pictures, device motion-estimation calls of THIS instance
        assert hip[r]["md5"] == cpu[r]["md5"] and hip[r]["bytes"] == cpu[r]["bytes"], (r, cpu[r], hip[r])
