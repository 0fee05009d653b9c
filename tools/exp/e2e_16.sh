python - <<PY
import sys, json
sys.path.insert(0, "tests")
import e2e_fps
for t in (16, 12):
    runs = [e2e_fps.run(dict(w=1920, h=1080, frames=65, threads=t, mask=m)) for m in (0, 8336) * 6]
    med = lambda v: sorted(v)[len(v) // 2]
    cpu = med([r["fps"] for r in runs if r["mask"] == 0]); hip = med([r["fps"] for r in runs if r["mask"] == 8336])
    print(json.dumps({"threads": t, "cpu_fps": round(cpu, 2), "hip_fps": round(hip, 2), "speedup": round(hip / cpu, 3), "identical": len({r["md5"] for r in runs}) == 1, "runs": [round(r["fps"], 2) for r in runs]}))
PY
