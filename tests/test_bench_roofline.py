"""CPU tier: bench.py's roofline objects (tools/bench_profile.roofline_objects) on synthetic counters — the cross-checks that decide whether an algorithmic figure may be quoted
as a fraction of the HBM peak: no class above the peak, the step's sum below the peak; otherwise `basis` says "reuse (LDS/L1) — not an HBM fraction".
(The window kernel's byte accounting itself is checked on a real recording: tests/test_recorder.py::test_window_kernel_byte_accounting.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _kern(alg_mb):
    return {k: {"kernel": k, "avg_ms_per_picture": ms, "alg_bytes_per_picture": int(a * 1e6), "nominal_alg_GBps": a * 1e6 / (ms * 1e-3) / 1e9}
            for k, (ms, a) in alg_mb.items()}


def _live(kern, us, fetch_mb, valu=0.4, l1=0.3):
    import bench_profile as BP
    per = {}
    for k in kern:
        t = us[k] * 1e-6
        per[k] = {"launches": 32, "total_ns": 32 * us[k] * 1e3, "fetch_kib": 32 * fetch_mb[k] * 1e6 / 1024 / 2.0, "n_fetch_kib": 32, "write_kib": 0.0, "n_write_kib": 32,
                  "valu_insts": 32 * valu * BP.N_SIMD * BP.CLOCK_GHZ * 1e9 * t / 4.0, "n_valu_insts": 32, "l1_accesses": 32 * l1 * BP.N_CU * BP.CLOCK_GHZ * 1e9 * t, "n_l1_accesses": 32,
                  "l2_hits": 32 * 900.0, "n_l2_hits": 32, "l2_misses": 32 * 100.0, "n_l2_misses": 32}
    return {"per_class": per, "kernel_trace": {"command": "synthetic", "kernels": []}}


CALIB = {"factors": {"rows16": 2.0, "stream16": 2.0, "store8": 1.0}, "how": "synthetic"}


def test_checks_pass_and_fractions_are_what_the_bytes_say():
    import bench_profile as BP
    kern = _kern({"ME_stage": (0.030, 120.0), "ME_int": (0.016, 24.0), "ME_item": (0.024, 130.0), "TU": (0.018, 40.0), "DMVR": (0.011, 17.0)})
    us = {"ME_stage": 30.0, "ME_int": 16.0, "ME_item": 24.0, "TU": 18.0, "DMVR": 11.0}
    live = _live(kern, us, {"ME_stage": 15.0, "ME_int": 13.0, "ME_item": 80.0, "TU": 43.0, "DMVR": 7.5})
    uniq = {"ME_stage": 16e6, "ME_int": 17e6, "ME_item": 74e6, "TU": 40e6, "DMVR": 8.6e6}
    roof, allk, checks = BP.roofline_objects(kern, live, CALIB, uniq, 0.066)
    assert checks["ok"] is True and checks["max_class_alg_frac"] <= 1.0 and checks["sum_alg_over_step_time_GBps"] <= BP.HBM_PEAK_GBS
    assert roof["kernel"] == "meStageKernel" and roof["bound"] == "hbm" and roof["peak"] == BP.HBM_PEAK_GBS
    assert abs(roof["frac"] - 120e6 / 30e-6 / 1e9 / 8000.0) < 1e-6 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["frac_physical"] - 15e6 / 30e-6 / 1e9 / 8000.0) < 2e-4 and abs(roof["frac_unique"] - 16e6 / 30e-6 / 1e9 / 8000.0) < 2e-4
    assert roof["binding_resource"] == "valu" and abs(roof["binding_frac"] - 0.4) < 1e-3                  # the line carries what binds the launch next to `frac`
    assert "not an HBM fraction" not in roof["basis_short"] and "cross-checks passed" in roof["basis_short"]
    assert allk["ME_item"]["binding_resource"] == "hbm"                                                   # 80 MB in 24 us = 0.42 of the peak > VALU 0.4
    assert all(r["alg_frac"] <= 1.0 for r in allk.values())


def test_a_class_above_the_peak_switches_the_basis():
    import bench_profile as BP
    # round 4's accounting: every listed candidate of the window kernel counted with its whole block -> 3.6 x the HBM peak
    kern = _kern({"ME_stage": (0.076, 520.0), "ME_int": (0.0317, 908.7), "ME_item": (0.087, 518.0), "TU": (0.035, 82.0), "DMVR": (0.027, 60.0)})
    us = {"ME_stage": 76.0, "ME_int": 31.7, "ME_item": 87.0, "TU": 35.0, "DMVR": 27.0}
    live = _live(kern, us, {"ME_stage": 60.0, "ME_int": 44.0, "ME_item": 355.0, "TU": 93.0, "DMVR": 28.0})
    roof, allk, checks = BP.roofline_objects(kern, live, CALIB, {k: 1e6 for k in kern}, 0.2445)
    assert checks["ok"] is False and checks["max_class_alg_frac"] > 3.0 and checks["sum_alg_over_step_time_GBps"] > BP.HBM_PEAK_GBS
    assert roof["basis"].startswith(BP.BASIS_REUSE) and roof["basis_short"].startswith(BP.BASIS_REUSE)


def test_no_live_pass_reports_no_traffic_and_no_verdict():
    import bench_profile as BP
    kern = _kern({"ME_stage": (0.030, 120.0), "TU": (0.018, 40.0)})
    roof, allk, checks = BP.roofline_objects(kern, None, CALIB, {}, 0.066)
    assert roof["traffic"] is None and allk == {} and checks["ok"] is None and roof["frac"] > 0
