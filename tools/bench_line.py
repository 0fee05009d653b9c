"""bench.py's LAST stdout line: one compact JSON object (target <= 7.5 KB, hard cap 8 KB: the tail the driver keeps) that carries the measurement on its own; everything else of the run goes to
bench_detail.json and to stdout BEFORE that line.  Host logic only (no torch, no device): tests/test_bench_line.py formats a full-size dummy run through it.

The driver keeps the tail of stdout: round 4's single 34 KB line was cut and its record was lost (BENCH_r04.json parsed: null)."""
import json
import os

TARGET_BYTES, HARD_CAP_BYTES = 7680, 8192           # (the target leaves room for what a driver appends behind stdout inside an 8 KB tail)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(v, n=4):
    if isinstance(v, float):
        return float("%.*g" % (n + 1, v)) if abs(v) >= 1 else round(v, n)
    if isinstance(v, dict):
        return {k: _r(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, n) for x in v]
    return v


ROOF_KEYS = ("bound", "kernel", "avg_launch_ms", "launches_per_picture", "alg_bytes_per_launch", "achieved", "peak", "unit", "frac", "traffic", "frac_physical", "unique_bytes_per_launch", "frac_unique",
             "traffic_over_unique", "l2_hit_rate", "l1_access_frac", "valu_issue_frac", "binding_resource", "binding_frac")
ROOF_4K_KEYS = ("kernel", "avg_launch_ms", "alg_bytes_per_launch", "achieved", "frac", "traffic", "frac_physical", "frac_unique", "l2_hit_rate", "binding_resource", "binding_frac")
CLASS_KEYS = ("avg_launch_us", "alg_frac", "fabric_frac", "l2_hit_rate", "l1_access_frac", "valu_issue_frac")          # (unique / traffic-over-unique: detail file, profiles/)
CLASS_KEYS_MCTF = ("avg_launch_us", "alg_frac", "fabric_frac", "l1_access_frac", "valu_issue_frac")          # (the MCTF rows: the full set is in the detail file and in profiles/)
CLASS_KEYS_MCTF_4K = ("avg_launch_us", "alg_frac", "valu_issue_frac")
CLASS_KEYS_4K = ("avg_launch_us", "alg_frac", "fabric_frac", "l2_hit_rate", "l1_access_frac", "valu_issue_frac")
E2E_KEYS = ("threads", "pairs", "cpu_fps", "hip_fps", "speedup", "cpu_fps_best", "hip_fps_best", "speedup_best", "bitstreams_identical")


def _roof(r, keys, checks=None):
    o = _pick(r, keys)
    if "basis_short" in r or "basis" in r:
        o["basis"] = r.get("basis_short") or r["basis"][:240]
    if checks:
        o["checks"] = _pick(checks, ("ok", "max_class_alg_frac", "sum_alg_over_step_time_GBps"))
    return o


def _e2e(e):
    if not isinstance(e, dict):
        return None
    if "error" in e or "skipped" in e:
        return _pick(e, ("error", "skipped"))
    o = _pick(e, E2E_KEYS)
    if e.get("other_threads"):
        o["other_threads"] = [_pick(r, ("threads", "cpu_fps", "hip_fps", "speedup", "error")) for r in e["other_threads"]]
    if isinstance(e.get("scalar"), dict):
        o["scalar_md5_equal"] = e["scalar"].get("md5_equal")
        o["scalar_md5_equal_default"] = e["scalar"].get("md5_equal_default")
        o["hip_equals_default_mctf_on"] = e["scalar"].get("hip_equals_default")
    ss = e.get("stage_split")
    if isinstance(ss, dict) and "share" in ss:
        o["stage_split"] = {"share": ss["share"], "device_stage_share": ss.get("device_stage_share"), "amdahl_bound_speedup": ss.get("amdahl_bound_speedup"),
                            "amdahl_bound_speedup_with_all_of_alf": ss.get("amdahl_bound_speedup_with_all_of_alf")}
    return o


def compact(out, detail_path):
    """the driver-facing line from the full result object `out` (see bench.py's docstring for the objects)"""
    line = _pick(out, ("metric", "value", "value_with_mctf", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_with_mctf", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, ("dtype", "data")))
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("workload", "hip_streams", "sample_pairs_per_frame", "coefficients_per_frame", "recorded_calls_outside_the_lists"))
    if "value_note" in out:
        line["value_note"] = out["value_note"]
    if "roofline" in out:
        line["roofline"] = _roof(out["roofline"], ROOF_KEYS, out.get("roofline_checks"))
    if out.get("roofline_all_kernels"):
        line["classes"] = {k: _pick(v, CLASS_KEYS_MCTF if k.startswith("MCTF") else CLASS_KEYS) for k, v in out["roofline_all_kernels"].items()}
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "passes", "value_fastest_passes", "value_slowest_passes", "value_1thread", "loadavg_before_after"))
        line["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:240]
        if isinstance(cb.get("mctf"), dict):
            line["cpu_baseline"]["mctf"] = _pick(cb["mctf"], ("value", "value_1thread", "unit", "cores", "kind", "error", "skipped"))
    if "parity" in out:
        p = out["parity"]
        line["parity"] = _pick(p, ("status", "mismatches", "error"))
        if isinstance(p.get("checked"), dict):
            line["parity"]["checked"] = p["checked"]
    if isinstance(out.get("parity_mctf"), dict):
        line["parity_mctf"] = _pick(out["parity_mctf"], ("status", "fields", "field_mismatches", "planes", "plane_mismatches", "error"))
    for k in ("gop_weighted", "single_stream"):
        if k in out:
            line[k] = _pick(out[k], ("value", "ms_per_step"))
    if isinstance(out.get("with_mctf"), dict) and isinstance(out["with_mctf"].get("parity_exchange"), dict):
        line["parity_exchange"] = _pick(out["with_mctf"]["parity_exchange"], ("status", "jobs_checked_over_ranks", "ranks_that_detected_a_corrupted_slot"))
        line["mctf_exchanges"] = _pick(out["with_mctf"], ("exchanges_in_the_timed_steps", "bytes_per_exchange"))
    if isinstance(out.get("with_mctf"), dict) and isinstance(out["with_mctf"].get("gop_cycle"), dict):
        line["gop_cycle_with_mctf"] = _pick(out["with_mctf"]["gop_cycle"], ("value", "value_without_mctf", "ms_per_cycle", "ms_per_cycle_without_mctf"))
    # ---- 3840x2160
    line.update(_pick(out, ("value_4k", "value_4k_with_mctf", "ms_per_step_4k", "ms_per_step_4k_with_mctf", "error_4k")))
    if isinstance(out.get("with_mctf_4k"), dict) and isinstance(out["with_mctf_4k"].get("gop_cycle"), dict):
        line["gop_cycle_with_mctf_4k"] = _pick(out["with_mctf_4k"]["gop_cycle"], ("value", "value_without_mctf"))
    if isinstance(out.get("parity_mctf_4k"), dict):
        line["parity_mctf_4k"] = _pick(out["parity_mctf_4k"], ("status", "error"))
    if "roofline_4k" in out:
        line["roofline_4k"] = _roof(out["roofline_4k"], ROOF_4K_KEYS, out.get("roofline_checks_4k"))
        line["roofline_4k"].pop("basis", None)
    if out.get("roofline_all_kernels_4k"):
        line["classes_4k"] = {k: _pick(v, CLASS_KEYS_MCTF_4K if k.startswith("MCTF") else CLASS_KEYS_4K) for k, v in out["roofline_all_kernels_4k"].items()}
    if "parity_4k" in out:
        line["parity_4k"] = _pick(out["parity_4k"], ("status", "mismatches", "error"))
    if "cpu_baseline_4k" in out:
        line["cpu_baseline_4k"] = _pick(out["cpu_baseline_4k"], ("value", "cores", "value_1thread"))
        if isinstance(out["cpu_baseline_4k"].get("mctf"), dict):
            line["cpu_baseline_4k"]["mctf"] = _pick(out["cpu_baseline_4k"]["mctf"], ("value", "value_1thread"))
    # ---- the other legs
    m3 = out.get("config3_medium_4k")
    if isinstance(m3, dict):
        line["config3_medium_4k"] = _pick(m3, ("ms_per_picture", "pictures_per_s", "nothing_dropped", "error"))
        if isinstance(m3.get("parity"), dict):
            line["config3_medium_4k"]["parity"] = m3["parity"].get("status")
    for k in ("mctf", "mctf_4k"):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("me_ms_per_picture", "me_ms_per_picture_4_in_flight", "filter_ms_per_picture", "error"))
    for k in ("e2e", "e2e_4k"):
        if k in out:
            line[k] = _e2e(out[k])
    # ---- N > 1
    if "exchange" in out:
        line["exchange"] = _pick(out["exchange"], ("pictures", "bytes_per_rank", "collective", "every_steps", "backend", "exchange_ms_per_picture"))
    for k in ("no_exchange", "exchange_per_gop_cycle", "exchange_every_reference"):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("value", "ms_per_step", "every_steps"))
    if isinstance(out.get("e2e_instances"), dict):
        line["e2e_instances"] = _pick(out["e2e_instances"], ("instances", "threads_per_instance", "cpu_fps_aggregate", "hip_fps_aggregate", "speedup", "chunk_bitstreams_identical", "error", "skipped"))
    line["detail"] = detail_path
    line = _r(line)
    s = json.dumps(line, separators=(",", ":"))
    # shrink in a fixed order until the line fits the target (the caps are asserted by tests/test_bench_line.py on a full-size dummy AND on oversized objects): first the optional
    # groups, then the long strings, then the encoder legs' side rows, then the contract's fields alone — a line is never lost to its own size
    dumps = lambda: json.dumps(line, separators=(",", ":"))

    def note(what):
        d = line.get("dropped_from_line")
        line["dropped_from_line"] = (d if isinstance(d, list) else []) + [what]

    for drop in ("classes_4k", "classes", "single_stream", "gop_weighted", "mctf_4k", "mctf"):
        if len(s) <= TARGET_BYTES:
            break
        if line.pop(drop, None) is not None:
            note(drop)
        s = dumps()
    if len(s) > TARGET_BYTES:                     # long strings
        for path, n in ((("value_note",), 120), (("config", "workload"), 200), (("cpu_baseline", "sample"), 120), (("roofline", "basis"), 120)):
            d = line
            for k in path[:-1]:
                d = d.get(k) if isinstance(d, dict) else None
            if isinstance(d, dict) and isinstance(d.get(path[-1]), str) and len(d[path[-1]]) > n:
                d[path[-1]] = d[path[-1]][:n]
        for v in line.values():                   # error strings of any leg
            if isinstance(v, dict):
                for k in ("error", "skipped"):
                    if isinstance(v.get(k), str):
                        v[k] = v[k][:80]
        note("long strings cut")
        s = dumps()
    if len(s) > TARGET_BYTES:                     # the encoder legs' side rows
        for k in ("e2e", "e2e_4k"):
            if isinstance(line.get(k), dict):
                for q in ("other_threads", "stage_split", "md5_set"):
                    line[k].pop(q, None)
        note("e2e side rows")
        s = dumps()
    for drop in ("e2e_instances", "exchange", "exchange_per_gop_cycle", "no_exchange", "exchange_every_reference", "config3_medium_4k", "cpu_baseline_4k", "parity_4k", "roofline_4k", "e2e_4k", "e2e", "value_note"):
        if len(s) <= TARGET_BYTES:
            break
        if line.pop(drop, None) is not None:
            note(drop)
        s = dumps()
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "parity", "detail")
    if len(s) > TARGET_BYTES:
        line = {k: line[k] for k in keep if k in line}
        if isinstance(line.get("parity"), dict):
            line["parity"] = _pick(line["parity"], ("status", "mismatches"))
        if isinstance(line.get("roofline"), dict):
            line["roofline"] = _pick(line["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"))
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"] = _pick(line["cpu_baseline"], ("value", "unit", "cores", "kind"))
            line["cpu_baseline"]["sample"] = str((out.get("cpu_baseline") or {}).get("sample", ""))[:120]
        line["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
        line["dropped_from_line"] = "everything but the contract's fields (line over %d bytes)" % TARGET_BYTES
        s = dumps()
    if len(s) > HARD_CAP_BYTES:          # (cannot happen with the fields above: every string left is cut to a fixed length)
        line = {k: line[k] for k in keep[:12] if k in line}
        s = dumps()
    return s


def emit(out, root, name="bench_detail.json"):
    """writes the full object to <root>/<name> (and to <root>/gpurun_out/ when that exists: it travels back from a gpurun box), prints it to stdout one top-level key per line,
    then prints the compact line LAST and returns it"""
    paths = [os.path.join(root, name)]
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        paths.append(os.path.join(root, "gpurun_out", name))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
    print("---- detail (also in %s) ----" % name)
    for k, v in out.items():
        print(json.dumps({k: v}))
    print("---- result line ----", flush=True)
    s = compact(out, name)
    print(s, flush=True)
    return s
