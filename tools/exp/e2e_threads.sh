# encoder end to end at several thread counts: CPU vs --SIMD=HIP production mask, 1080p x 65 faster, three alternating pairs each
for t in 2 4 8 16; do
python - $t <<PY
import sys, json
sys.path.insert(0, "tests")
import e2e_fps
t = int(sys.argv[1])
runs = [e2e_fps.run(dict(w=1920, h=1080, frames=65, threads=t, mask=m)) for m in (0, 8336) * 3]
med = lambda v: sorted(v)[len(v) // 2]
cpu = med([r["fps"] for r in runs if r["mask"] == 0]); hip = med([r["fps"] for r in runs if r["mask"] == 8336])
print(json.dumps({"threads": t, "cpu_fps": round(cpu, 2), "hip_fps": round(hip, 2), "speedup": round(hip / cpu, 3), "identical": len({r["md5"] for r in runs}) == 1,
                  "alf_pictures_on_device": runs[1]["calls"][16], "mctf_pictures": runs[1]["calls"][9], "runs": [round(r["fps"], 2) for r in runs]}))
PY
done
