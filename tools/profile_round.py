#!/usr/bin/env python3
"""Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):

  1. rocprofv3 --kernel-trace --stats   -- python bench.py --inner --steps S --warmup W     (the inner run bench.py itself profiles: the picture's three launches
                                                                                              serialized on one stream + two MCTF stages; no extras, no output line)
  2. rocprofv3 --pmc FETCH_SIZE         -- same command, fewer steps   (separate pass, no tracing domains)
  3. rocprofv3 --pmc WRITE_SIZE         -- same

and writes gpurun_out/<tag>.md (kernel table + counter tables) and gpurun_out/pmc_<tag>.json (bytes per launch and kernel class,
FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 counts a 128-byte request as 64 B).  ROCm 7.2 stores results as rocpd SQLite.

usage: python tools/profile_round.py <tag> [--steps 50] [--warmup 5] [-- extra bench args]"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def class_of(name):
    """bench.py's kernel class of a rocprof kernel name"""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "")
    head = n.split("(")[0]
    if head.startswith("sadSseMixedKernel"):
        return "SAD_SSE"
    if head.startswith("sadSseMultiKernel<"):
        arg = head[len("sadSseMultiKernel<"):].rstrip(">")
        return "SSE" if arg.endswith("1") else "SAD"
    if head.startswith("hadTile8MultiKernel") or head.startswith("hadTile8PkMultiKernel"):
        return "HAD_fast"
    if head.startswith("tuRdoRowMultiKernel") or head.startswith("tuMxMultiKernel"):
        return "TU"
    if head.startswith("tuRdoRowKernel<"):
        return "TU" + head[len("tuRdoRowKernel<"):].split(",")[0]
    return None


def find_db(d):
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        raise SystemExit("no rocpd database under " + d)
    return dbs[-1]


def run_pass(outdir, prof_args, bench_args):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3"] + prof_args + ["-d", outdir, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        raise SystemExit("rocprofv3 pass failed: " + " ".join(prof_args))
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    return find_db(outdir), (line[-1] if line else None)


def kernel_table(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    return cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by 1 order by 3 desc" % name).fetchall()


def counter_table(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    return cur.execute("select %s, counter_name, count(*), sum(value), avg(value) from counters_collection group by 1,2 order by 4 desc" % kcol).fetchall()


def main():
    tag = sys.argv[1]
    steps, warm, extra = "50", "5", []
    a = sys.argv[2:]
    while a:
        x = a.pop(0)
        if x == "--steps": steps = a.pop(0)
        elif x == "--warmup": warm = a.pop(0)
        elif x == "--": extra = a; break
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    base = ["--inner"] + extra      # the trace holds the serialized launches (+ the MCTF stage kernels) only
    db1, line = run_pass(os.path.join(out, "prof_%s_trace" % tag), ["--kernel-trace", "--stats"], ["--steps", steps, "--warmup", warm] + base)
    db2, _ = run_pass(os.path.join(out, "prof_%s_fetch" % tag), ["--pmc", "FETCH_SIZE"], ["--steps", "5", "--warmup", "1"] + base)
    db3, _ = run_pass(os.path.join(out, "prof_%s_write" % tag), ["--pmc", "WRITE_SIZE"], ["--steps", "5", "--warmup", "1"] + base)

    md = ["# Profile `%s`" % tag, "",
          "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps %s --warmup %s %s` on 1x MI355X (rocpd database, summarised by tools/profile_round.py)." % (steps, warm, " ".join(base)), "",
          "| kernel | calls | total (us) | avg (us) | min (us) | max (us) | % |", "|---|---|---|---|---|---|---|"]
    rows = kernel_table(db1)
    tot = sum(r[2] for r in rows) or 1
    for k, n, s, av, mn, mx in rows[:16]:
        md.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k[:100], n, s / 1e3, av / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    classes = {}
    for title, db, key in (("FETCH_SIZE (KB per dispatch; gfx950: multiply by 2, MI355X_MICROARCH.md HBM section)", db2, "fetch"), ("WRITE_SIZE (KB per dispatch)", db3, "write")):
        md += ["", "## PMC pass: %s — separate run, --steps 5 --warmup 1" % title, "", "| kernel | counter | dispatches | sum | avg/dispatch |", "|---|---|---|---|---|"]
        for i, (k, c, n, s, av) in enumerate(counter_table(db)):
            if i < 14:
                md.append("| %s | %s | %d | %.6g | %.6g |" % (k[:100], c, n, s, av))
            cls = class_of(k)
            if cls:
                e = classes.setdefault(cls, {"fetch_kb": 0.0, "write_kb": 0.0, "launches_fetch": 0, "launches_write": 0})
                e[key + "_kb"] += s
                e["launches_" + key] += n
    pmc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 5 --warmup 1 %s; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated" % " ".join(base),
           "tag": tag, "classes": {}}
    for cls, e in classes.items():
        f = 2.0 * 1024.0 * e["fetch_kb"] / max(1, e["launches_fetch"])
        w = 1024.0 * e["write_kb"] / max(1, e["launches_write"])
        pmc["classes"][cls] = {"launches": e["launches_fetch"], "fetch_bytes_per_launch_x2_corrected": f, "write_bytes_per_launch": w, "traffic_bytes_per_launch": f + w}
    md += ["", "(the inner run prints no bench line; the bench line of the round is profiles/bench_%s.json)" % tag, ""]
    open(os.path.join(out, "%s.md" % tag), "w").write("\n".join(md))
    json.dump(pmc, open(os.path.join(out, "pmc_%s.json" % tag), "w"), indent=1)
    print("\n".join(md[:30]))
    print(json.dumps(pmc["classes"], indent=1))


if __name__ == "__main__":
    main()
