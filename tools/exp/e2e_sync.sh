# encoder end to end, 16 and 8 threads: CPU vs --SIMD=HIP with blocking (default) and spinning host waits
python - <<PY
import sys, json, os
sys.path.insert(0, "tests")
import e2e_fps
for t in (16, 8):
    res = {"threads": t}
    runs = []
    for rep in range(3):
        for tag, mask, env in (("cpu", 0, None), ("block", 8336, "block"), ("spin", 8336, "spin")):
            if env: os.environ["VVHIP_SYNC"] = env
            r = e2e_fps.run(dict(w=1920, h=1080, frames=65, threads=t, mask=mask))
            runs.append((tag, round(r["fps"], 2), r["md5"]))
    med = lambda v: sorted(v)[len(v) // 2]
    for tag in ("cpu", "block", "spin"):
        res[tag] = med([f for g, f, _ in runs if g == tag])
    res["identical"] = len({m for _, _, m in runs}) == 1
    res["runs"] = [(g, f) for g, f, _ in runs]
    print(json.dumps(res))
PY
