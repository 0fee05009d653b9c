"""CPU tier: bench.py's LAST stdout line (tools/bench_line.py) — the driver keeps only the tail of stdout, and round 4's single 34 KB line was cut there (BENCH_r04.json
parsed: null).  A full-size result object (round 4's committed profiles/bench_r04_f.json, widened with this round's objects) must format into ONE line of at most 8 KB that
still carries the contract's fields; an oversized object must still come out under the hard cap."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _full_size_result():
    out = json.load(open(os.path.join(ROOT, "profiles", "bench_r04_f.json")))
    assert len(json.dumps(out)) > 30000            # the object that was lost
    roof = out["roofline"]
    roof.update({"frac_unique": 0.0641, "unique_bytes_per_launch": 16.2e6, "basis_short": "algorithmic bytes (SURVEY 8d per batch unit) / launch time / 8 TB/s; cross-checks passed; "
                 "LDS reuse makes alg = 8.9x the fabric traffic: the launch is bound by binding_resource, see frac_physical / frac_unique"})
    out["roofline_checks"] = out["roofline_checks_4k"] = {"ok": True, "max_class_alg_frac": 0.61, "sum_alg_over_step_time_GBps": 5123.4, "sum_alg_MB_per_step": 360.0, "rule": "x" * 200}
    for k in ("roofline_all_kernels", "roofline_all_kernels_4k"):
        for r in out[k].values():
            r.update({"alg_frac": 0.5123, "unique_frac": 0.0712, "fabric_frac": 0.0611, "binding_resource": "valu", "binding_frac": 0.4444})
    out["cpu_baseline"].update({"value_1thread": 21.3, "loadavg_before_after": [3.1, 14.2], "scaling_over_1thread": 11.2})
    for k, (w, h) in (("e2e", (1920, 1080)), ("e2e_4k", (3840, 2160))):
        out[k].update({"pairs": 5, "cpu_fps_best": 48.58, "hip_fps_best": 49.9, "speedup_best": 1.027, "md5_set": "default SIMD (AVX2) x 8, --SIMD=HIP x 7, --SIMD=SCALAR x 1",
                       "other_threads": [{"threads": 1, "pairs": 1, "cpu_fps": 6.1, "hip_fps": 6.6, "speedup": 1.08, "runs_fps": [6.1, 6.6]}, {"threads": 16, "pairs": 1, "cpu_fps": 61.0, "hip_fps": 60.2, "speedup": 0.99, "runs_fps": [61.0, 60.2]}],
                       "scalar": {"fps": 21.2, "md5_equal": True, "md5_equal_default": False, "note": "z" * 200},
                       "stage_split": {"frames": 33, "threads": 0, "share": {"north_star_A_distortion_in_motion_search": 0.2101, "north_star_B_transform_quant": 0.0911, "north_star_C_mctf": 0.0791, "alf": 0.1950},
                                       "share_top": {"P_X%d" % i: 0.1 for i in range(8)}, "device_stage_share": 0.0991, "amdahl_bound_speedup": 1.11, "note": "y" * 300}})
    out["value_note"] = "pictures/s of the north-star hot path (recorded lists resident in HBM) — the path's throughput, NOT encoder fps; the encoder's frames/s with the device stages is e2e.hip_fps"
    return out


def test_full_size_result_formats_into_one_line_under_8k():
    import bench_line
    out = _full_size_result()
    s = bench_line.compact(out, "bench_detail.json")
    assert "\n" not in s and len(s.encode()) <= bench_line.TARGET_BYTES, len(s)
    line = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "detail"):
        assert k in line, k
    assert "dropped_from_line" not in line
    assert line["config"]["workload"]
    for k in ("bound", "kernel", "avg_launch_ms", "alg_bytes_per_launch", "achieved", "peak", "unit", "frac", "traffic", "frac_physical", "binding_resource", "binding_frac", "basis", "checks"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "value_1thread", "loadavg_before_after"):
        assert k in line["cpu_baseline"], k
    assert line["parity"]["status"] == "bit-exact" and line["parity_4k"]["status"] == "bit-exact"
    assert line["value_4k"] and line["ms_per_step_4k"] and "frac" in line["roofline_4k"]
    for k in ("e2e", "e2e_4k"):
        assert {"cpu_fps", "hip_fps", "speedup", "bitstreams_identical", "cpu_fps_best", "other_threads", "scalar_md5_equal", "stage_split"} <= set(line[k]), line[k].keys()
    assert abs(line["value"] - out["value"]) / out["value"] < 1e-3 and abs(line["ms_per_step"] - out["ms_per_step"]) / out["ms_per_step"] < 1e-3


def test_oversized_result_still_ends_under_the_hard_cap():
    import bench_line
    out = _full_size_result()
    big = copy.deepcopy(out)
    big["parity"]["checked"] = {"k%d" % i: i for i in range(400)}                      # grows past the target: optional groups go first
    s = bench_line.compact(big, "bench_detail.json")
    assert len(s) <= bench_line.HARD_CAP_BYTES
    assert json.loads(s).get("dropped_from_line")
    big["parity"]["checked"] = {"k%d" % i: i for i in range(4000)}                     # past the hard cap: the contract's fields only
    s = bench_line.compact(big, "bench_detail.json")
    line = json.loads(s)
    assert len(s) <= bench_line.HARD_CAP_BYTES and "roofline" in line
    assert line["value"] and line["roofline"]["frac"] and line["cpu_baseline"]["value"]


def test_line_never_exceeds_the_tail_the_driver_keeps():
    """between the target and the old 16 KB cap a line went out unchanged (several 300-character error strings, other_threads rows, the N > 1 fields): every
    oversized shape must end at or below 8 192 bytes — the figure tests/test_gpu_parity.py asserts on the real run — and keep the contract's fields"""
    import bench_line
    assert bench_line.HARD_CAP_BYTES <= 8192 and bench_line.TARGET_BYTES <= bench_line.HARD_CAP_BYTES
    base = _full_size_result()
    shapes = []
    a = copy.deepcopy(base)                                   # long error strings in every optional leg + N > 1 fields
    for k in ("e2e", "e2e_4k", "config3_medium_4k", "mctf", "mctf_4k"):
        a[k] = {"error": "E" * 300}
    a["error_4k"] = "F" * 400
    a["e2e_instances"] = {"error": "G" * 300}
    a["exchange"] = {"pictures": 10, "bytes_per_rank": 7700000, "collective": "broadcast", "every_steps": 2, "backend": "nccl", "exchange_ms_per_picture": 0.0123}
    a["no_exchange"] = a["exchange_per_gop_cycle"] = {"value": 1.0, "ms_per_step": 2.0, "every_steps": 32}
    a["cpu_baseline"]["sample"] = "S" * 2000
    a["value_note"] = "N" * 3000
    shapes.append(a)
    b = copy.deepcopy(base)                                   # many side rows
    for k in ("e2e", "e2e_4k"):
        b[k]["other_threads"] = [{"threads": t, "cpu_fps": 1.5, "hip_fps": 1.6, "speedup": 1.07} for t in range(1, 60)]
        b[k]["md5_set"] = "m" * 900
    shapes.append(b)
    c = copy.deepcopy(base)                                   # a class table that grew
    c["roofline_all_kernels"] = {"class%d" % i: dict(next(iter(base["roofline_all_kernels"].values()))) for i in range(60)}
    c["config"]["workload"] = "W" * 5000
    shapes.append(c)
    for o in shapes:
        s = bench_line.compact(o, "bench_detail.json")
        assert len(s.encode()) <= bench_line.TARGET_BYTES, len(s)
        line = json.loads(s)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
            assert k in line, k
        assert line["roofline"]["frac"] and line["cpu_baseline"]["value"]


def test_emit_prints_the_compact_line_last(tmp_path, capsys):
    import bench_line
    out = _full_size_result()
    s = bench_line.emit(out, str(tmp_path), "detail.json")
    printed = capsys.readouterr().out.rstrip("\n").split("\n")
    assert printed[-1] == s and len(printed[-1]) <= bench_line.TARGET_BYTES
    assert json.load(open(tmp_path / "detail.json"))["kernels"] == out["kernels"]            # nothing of the run is lost: the full object is next to it
