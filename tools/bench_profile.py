"""bench.py's measurement side: the rocprofv3 passes over `bench.py --inner` (kernel trace + one --pmc pass per counter), the FETCH_SIZE / WRITE_SIZE calibration and the
roofline objects built from them."""
import glob
import json
import os
import shutil
import subprocess
import sys

from bench_common import CLOCK_GHZ, HBM_PEAK_GBS, KERNEL_NAMES, N_CU, N_SIMD, kernel_label

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")

# ---------------------------------------------------------------------------------------------------------------------- profiling passes
def run_inner_profile(width, height, prof, steps, tag, mode="--inner"):
    outdir = os.path.join("/tmp", "vvhip_prof_%d_%s" % (os.getpid(), tag))
    shutil.rmtree(outdir, ignore_errors=True)
    cmd = ["rocprofv3"] + prof + ["-d", outdir, "--", sys.executable, BENCH_PY, mode, "--steps", str(steps), "--warmup", "0",
                                  "--width", str(width), "--height", str(height)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 %s pass: rc %d: %s" % (tag, r.returncode, r.stdout[-400:]))
    dbs = sorted(glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        raise RuntimeError("rocprofv3 %s pass left no database" % tag)
    return dbs[-1], outdir


def class_of_kernel(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for cls, sub in KERNEL_NAMES.items():
        if n.startswith(sub):
            return cls
    return None


ALL_COUNTERS = (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_accesses"), ("SQ_INSTS_VALU", "valu_insts"), ("TCC_HIT_sum", "l2_hits"), ("TCC_MISS_sum", "l2_misses"))


def live_profile(width, height, counters=ALL_COUNTERS):
    """kernel trace + one --pmc pass per counter over a short inner run (32 steps = one GOP cycle, launches serialized): per kernel class the average duration and the RAW
    counters per launch (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; the calibrated byte factors are applied by the caller)"""
    import profile_round as P
    out, dirs = {}, []
    db, d = run_inner_profile(width, height, ["--kernel-trace", "--stats"], 32, "trace")
    dirs.append(d)
    rows = P.kernel_table(db)
    tot = sum(r[2] for r in rows) or 1
    out["kernel_trace"] = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --inner --steps 32 --width %d --height %d (one GOP cycle of recorded pictures, launches serialized on one stream)" % (width, height),
                           "kernels": [{"name": k.replace("(anonymous namespace)::", "")[:90], "calls": n, "avg_us": round(av / 1e3, 2), "total_us": round(s / 1e3, 1), "pct": round(100.0 * s / tot, 1)}
                                       for k, n, s, av, mn, mx in rows[:12]]}
    cls = {}
    for k, n, s, av, mn, mx in rows:
        c = class_of_kernel(k)
        if c:
            e = cls.setdefault(c, {"launches": 0, "total_ns": 0.0})
            e["launches"] += n
            e["total_ns"] += s
    for counter, key in counters:
        try:
            db, d = run_inner_profile(width, height, ["--pmc", counter], 32, counter)
            dirs.append(d)
            for k, c, n, s, av in P.counter_table(db):
                kc = class_of_kernel(k)
                if kc:
                    e = cls.setdefault(kc, {})
                    e[key] = e.get(key, 0.0) + s
                    e["n_" + key] = e.get("n_" + key, 0) + n
        except Exception as ex:
            out.setdefault("pmc_errors", []).append("%s: %s" % (counter, str(ex)[:160]))
    out["per_class"] = cls
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return out


# which calibration pattern (tools/calib/fetch_calib.hip) a kernel class's reads look like: per-lane 16-byte row gathers out of picture planes, or streams of compact blocks
FETCH_PATTERN = {"ME_stage": "rows16", "ME_int": "rows16", "ME_item": "rows16", "DMVR": "rows16", "TU": "stream16",
                 "MCTF_search": "rows16", "MCTF_nb": "rows16", "MCTF_fix": "rows16", "MCTF_apply": "rows16"}


def counter_calibration():
    """FETCH_SIZE / WRITE_SIZE factors measured on THIS GPU against known byte counts (tools/calib_fetch.py; two short rocprofv3 passes).  Falls back to the guide's figure for
    wide coalesced reads (x2) and x1 for writes, and says so."""
    try:
        import calib_fetch
        c = calib_fetch.calibrate()
        f = {k: v["factor"] for k, v in c["patterns"].items() if v.get("factor")}
        if not {"rows16", "stream16", "store8"} <= set(f):
            raise RuntimeError("patterns missing: %s" % sorted(f))
        return {"measured": True, "factors": {k: round(v, 4) for k, v in f.items()}, "how": "tools/calib_fetch.py: every byte of a 512 MiB buffer read / written once per pattern, "
                "factor = known bytes / (counter x 1024)", "pattern_of_class": FETCH_PATTERN}
    except Exception as e:
        return {"measured": False, "factors": {"rows16": 2.0, "stream16": 2.0, "store8": 1.0}, "how": "calibration failed (%s): MI355X_MICROARCH.md's x2 for wide coalesced reads, x1 for writes" % str(e)[:120],
                "pattern_of_class": FETCH_PATTERN}


BASIS_ALG = ("achieved = ALGORITHMIC bytes of one launch (SURVEY 8d per batch unit: integer windows = a job's window read once + its block + 8 B per distinct position; sub-pel stages = "
             "(w+taps)(h+taps) 2 + 2 w h + 8 per scored position; table calls 4 w h + 8; TUs 6 w h + 24; MCTF per SCORED candidate (counted by the kernels): integer 4 w h, fractional (w+3)(h+3) 2 + 2 w h) / the launch's average duration in this run's rocprofv3 kernel trace; frac = achieved / 8 TB/s")
BASIS_REUSE = "reuse (LDS/L1) — not an HBM fraction"


def roofline_objects(kern, live, calib, unique_by_class, ms_per_step, profile_md=None):
    """-> (roofline of the dominant kernel class, the same positions for every class, cross-checks).  Three byte counts per launch and class, all over the same launch duration:
    algorithmic, unique (the union of what the launch touches), fabric traffic (this run's counters x the factor calibrated on this GPU); next to them the L1 access
    and VALU issue fractions that bind these kernels.  The cross-checks say whether the algorithmic figures can be read as memory traffic at all: every class <= peak and the
    step's sum of algorithmic bytes / the step time <= peak; if not, `basis` says so instead of quoting a fraction of HBM."""
    dom = max(kern, key=lambda k: kern[k]["avg_ms_per_picture"])
    sum_alg = sum(kern[k]["alg_bytes_per_picture"] for k in kern)
    checks = {"sum_alg_MB_per_step": round(sum_alg / 1e6, 1), "sum_alg_over_step_time_GBps": round(sum_alg / (ms_per_step * 1e-3) / 1e9, 1) if ms_per_step else None}
    roof = {"bound": "hbm", "kernel": kernel_label(dom), "class": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": kern[dom]["avg_ms_per_picture"],
            "alg_bytes_per_launch": kern[dom]["alg_bytes_per_picture"], "achieved": kern[dom]["nominal_alg_GBps"], "frac": (kern[dom]["nominal_alg_GBps"] or 0.0) / HBM_PEAK_GBS, "traffic": None}
    allk = {}
    if not live or not live.get("per_class"):
        roof["basis"] = BASIS_ALG + "; no live rocprofv3 pass (--no-profile or rocprofv3 absent): HIP-event duration per picture instead of the trace's per-launch average, no counter traffic"
        checks["ok"] = None
        return roof, allk, checks
    fac = calib["factors"]
    for k, c in live["per_class"].items():
        n = max(1, c.get("launches", 1))
        t_k = (c["total_ns"] / n) * 1e-9 if c.get("total_ns") else None
        if not t_k or k not in kern:
            continue
        pk = lambda key: (c.get(key, 0.0) / max(1, c.get("n_" + key, 0))) if c.get("n_" + key) else None
        f_, w_, l_, v_, h_, m_ = pk("fetch_kib"), pk("write_kib"), pk("l1_accesses"), pk("valu_insts"), pk("l2_hits"), pk("l2_misses")
        ff = fac.get(FETCH_PATTERN.get(k, "rows16"), 2.0)
        traffic = (f_ * 1024.0 * ff + (w_ or 0.0) * 1024.0 * fac.get("store8", 1.0)) if f_ is not None else None
        lpp = n / 32.0                                        # launches per picture (the inner run is one GOP cycle of 32 pictures)
        alg = kern[k]["alg_bytes_per_picture"] / lpp
        uniq = (unique_by_class.get(k) or 0) / lpp
        gbps = lambda b: b / t_k / 1e9
        r = {"kernel": kernel_label(k), "launches_per_picture": round(lpp, 2), "avg_launch_us": round(t_k * 1e6, 2),
             "alg_MB_per_launch": round(alg / 1e6, 2), "alg_frac": round(gbps(alg) / HBM_PEAK_GBS, 4),
             "unique_MB_per_launch": round(uniq / 1e6, 2) if uniq else None, "unique_frac": round(gbps(uniq) / HBM_PEAK_GBS, 4) if uniq else None,
             "fabric_MB_per_launch": round(traffic / 1e6, 2) if traffic is not None else None, "fabric_frac": round(gbps(traffic) / HBM_PEAK_GBS, 4) if traffic is not None else None,
             "fetch_factor": round(ff, 3), "traffic_over_unique": round(traffic / uniq, 2) if traffic is not None and uniq else None,
             "l2_hit_rate": round(h_ / (h_ + m_), 3) if h_ is not None and m_ is not None and h_ + m_ > 0 else None,
             "l1_access_frac": round(l_ / (N_CU * CLOCK_GHZ * 1e9 * t_k), 4) if l_ else None,
             "valu_issue_frac": round(v_ * 4.0 / (N_SIMD * CLOCK_GHZ * 1e9 * t_k), 4) if v_ else None}
        if kern[k].get("per_position_bytes_per_picture"):      # the integer windows' per-position figure: a WORK rate (every candidate counts its block again), named as such
            r["per_position_MB_per_launch"] = round(kern[k]["per_position_bytes_per_picture"] / lpp / 1e6, 2)
            r["per_position_rate_over_hbm_peak"] = round(gbps(kern[k]["per_position_bytes_per_picture"] / lpp) / HBM_PEAK_GBS, 4)
            r["per_position_note"] = BASIS_REUSE + ": 4 w h' per distinct scored position, the window's samples counted once per candidate"
        fr = {"hbm": r["fabric_frac"] or 0.0, "l1_access": r["l1_access_frac"] or 0.0, "valu": r["valu_issue_frac"] or 0.0}
        r["binding_resource"] = max(fr, key=lambda q: fr[q])
        r["binding_frac"] = fr[r["binding_resource"]]
        allk[k] = r
    fracs = [r["alg_frac"] for r in allk.values()]
    checks["max_class_alg_frac"] = max(fracs) if fracs else None
    checks["ok"] = bool(fracs and max(fracs) <= 1.0 and (checks["sum_alg_over_step_time_GBps"] or 0) <= HBM_PEAK_GBS)
    checks["rule"] = "every class's algorithmic bytes / launch time <= 8 TB/s AND the sum of a step's algorithmic bytes / the multi-stream step time <= 8 TB/s"
    if dom in allk:
        d = allk[dom]
        t_s = d["avg_launch_us"] * 1e-6
        lpp = d["launches_per_picture"]
        alg = kern[dom]["alg_bytes_per_picture"] / (live["per_class"][dom]["launches"] / 32.0)
        traffic = d["fabric_MB_per_launch"] * 1e6 if d["fabric_MB_per_launch"] is not None else None
        roof.update({"avg_launch_ms": t_s * 1e3, "launches_per_picture": lpp, "alg_bytes_per_launch": alg, "achieved": alg / t_s / 1e9, "frac": alg / t_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "achieved_physical": (traffic / t_s / 1e9) if traffic else None, "frac_physical": d["fabric_frac"],
                     "unique_bytes_per_launch": d["unique_MB_per_launch"] * 1e6 if d["unique_MB_per_launch"] else None, "frac_unique": d["unique_frac"],
                     "traffic_over_unique": d["traffic_over_unique"], "traffic_over_alg_bytes": (traffic / alg) if traffic and alg else None,
                     "fetch_factor": d["fetch_factor"], "l2_hit_rate": d["l2_hit_rate"], "l1_access_frac": d["l1_access_frac"], "valu_issue_frac": d["valu_issue_frac"],
                     "binding_resource": d["binding_resource"], "binding_frac": d["binding_frac"]})
        reuse = (alg / traffic) if traffic else None
        roof["basis_short"] = ("algorithmic bytes (SURVEY 8d per batch unit) / launch time / 8 TB/s; cross-checks passed" if checks["ok"] else BASIS_REUSE) + \
            "; LDS reuse makes alg = %sx the fabric traffic: the launch is bound by binding_resource, see frac_physical / frac_unique" % ("%.1f" % reuse if reuse else "?")
        roof["basis"] = (BASIS_ALG if checks["ok"] else BASIS_REUSE + " (cross-checks failed: " + json.dumps({q: checks[q] for q in ("max_class_alg_frac", "sum_alg_over_step_time_GBps")}) + ")") + \
            ".  traffic = fabric bytes per launch from this run's --pmc passes (FETCH_SIZE x the calibrated factor + WRITE_SIZE; Infinity-Cache hits included -> frac_physical is an upper bound of " \
            "the HBM fraction); frac_unique = the union of the bytes the launch touches / time / peak.  The kernel stages windows in LDS, so the algorithmic bytes are %sx the traffic: what binds the " \
            "launch is binding_resource at binding_frac of its ceiling (VALU: wave instructions x 4 cycles / %d SIMDs x %.1f GHz; L1: one access per 64-byte granule and instruction / %d CUs), not HBM" \
            % ("%.1f" % reuse if reuse else "?", N_SIMD, CLOCK_GHZ, N_CU)
    else:
        roof["basis"] = BASIS_ALG + "; the PMC passes did not see the dominant kernel"
    if profile_md:
        try:
            cols = ("launches_per_picture", "avg_launch_us", "alg_MB_per_launch", "alg_frac", "unique_MB_per_launch", "unique_frac", "fabric_MB_per_launch", "fabric_frac", "traffic_over_unique",
                    "l2_hit_rate", "l1_access_frac", "valu_issue_frac", "binding_resource")
            with open(profile_md, "w") as f:
                f.write("# rocprofv3 summary of `python bench.py` (written by bench.py --profile-md from its own passes)\n\n")
                f.write("Inner run: `%s`\n\n" % live["kernel_trace"]["command"])
                f.write("## rocprofv3 --kernel-trace --stats\n\n| kernel | calls | avg us | total us | % |\n|---|---|---|---|---|\n")
                for r in live["kernel_trace"]["kernels"]:
                    f.write("| `%s` | %d | %.2f | %.1f | %.1f |\n" % (r["name"], r["calls"], r["avg_us"], r["total_us"], r["pct"]))
                f.write("\n## rocprofv3 --pmc, one pass per counter; per launch.  Fabric traffic = FETCH_SIZE x 1024 x the calibrated factor of the class's access pattern + WRITE_SIZE x 1024 "
                        "(calibration: %s; factors %s).  FETCH_SIZE includes Infinity-Cache hits.  *_frac = bytes / launch time / 8 TB/s.\n\n" % (calib["how"], json.dumps(calib["factors"])))
                f.write("| class | kernel | " + " | ".join(cols) + " |\n|---|---|" + "---|" * len(cols) + "\n")
                for k, r in allk.items():
                    f.write("| %s | `%s` | %s |\n" % (k, r["kernel"], " | ".join(str(r[c]) for c in cols)))
                if "ME_int" in allk and "per_position_MB_per_launch" in allk["ME_int"]:
                    f.write("\nInteger windows, per-position figure (%s): %s MB per launch = %s x 8 TB/s.\n" % (BASIS_REUSE, allk["ME_int"]["per_position_MB_per_launch"], allk["ME_int"]["per_position_rate_over_hbm_peak"]))
                f.write("\nCross-checks: %s\n" % json.dumps(checks))
                f.write("\nHIP-event time per picture (GOP-weighted, launches serialized): " + ", ".join("%s %.1f us" % (k, kern[k]["avg_ms_per_picture"] * 1e3) for k in kern) + "\n")
        except Exception as e:
            roof["profile_md_error"] = str(e)[:200]
    return roof, allk, checks


def mctf_profile(args):
    """the MCTF motion estimation under rocprofv3 (7 calls of one 1080p picture against 4 references): per picture the time of the parallel candidate scoring, of the sequential
    sweep (= the critical path of phase B: one workgroup per reference) and the VALU issue fraction of the scoring kernel"""
    import profile_round as P
    out, dirs = {}, []
    db, d = run_inner_profile(args.width, args.height, ["--kernel-trace", "--stats"], 1, "mctf_trace", "--inner-mctf")
    dirs.append(d)
    calls = 7.0
    t = {}
    for k, n, sm, av, mn, mx in P.kernel_table(db):
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith("me") or name.startswith("subsample") or name.startswith("extend") or name.startswith("initMvs"):
            t[name] = t.get(name, 0.0) + sm
    out["us_per_picture_by_kernel"] = {k: round(v / 1e3 / calls, 1) for k, v in sorted(t.items(), key=lambda kv: -kv[1])}
    out["phase_a_us"] = round(t.get("meSearchKernel", 0.0) / 1e3 / calls, 1)
    out["critical_path_us"] = round((t.get("meDiagKernel", 0.0) + t.get("meWavefrontKernel", 0.0)) / 1e3 / calls, 1)
    out["critical_path_note"] = "the anti-diagonal sweep of phase B (MCTF.cpp:1289-1306): one workgroup per reference, cols + rows dependent steps per level; everything else of the call is parallel over blocks"
    try:
        db, d = run_inner_profile(args.width, args.height, ["--pmc", "SQ_INSTS_VALU"], 1, "mctf_valu", "--inner-mctf")
        dirs.append(d)
        for k, c, n, sm, av in P.counter_table(db):
            if "meSearchKernel" in k and t.get("meSearchKernel"):
                out["phase_a_valu_issue_frac"] = round(sm * 4.0 / (N_SIMD * CLOCK_GHZ * 1e9 * t["meSearchKernel"] * 1e-9), 3)
    except Exception as ex:
        out["pmc_error"] = str(ex)[:160]
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return out

