mkdir -p gpurun_out
B="python bench.py --no-e2e --no-mctf --no-4k --no-cpu-baseline --no-profile --no-parity"
run() { tag=$1; shift; env "$@" $B > gpurun_out/exp_$tag.json 2> gpurun_out/exp_$tag.err; python - "$tag" <<PY
import json,sys
t=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/exp_%s.json"%t) if l.startswith("{\"metric\"")][-1])
    print(t, round(d["value"]), {k:round(v["avg_ms_per_picture"]*1000,1) for k,v in d["kernels"].items()})
except Exception as e:
    print(t,"ERR",e, open("gpurun_out/exp_%s.err"%t).read()[-600:])
PY
}
run base X=1
run pad4k VVHIP_ME_LDS_PAD=4096
run pad12k VVHIP_ME_LDS_PAD=12288
run pad24k VVHIP_ME_LDS_PAD=24576
run bw160 VVHIP_ME_BUNDLE_WORK=160
run bw640 VVHIP_ME_BUNDLE_WORK=640
run bw1280 VVHIP_ME_BUNDLE_WORK=1280
for m in 64:64 64:256 64:955 64:3000 32:256 32:2133 32:6000 16:600 8:800 4:600 64:955,32:2133,16:600,8:800,4:600 64:955,32:2133; do python tools/tu_mix.py $m; done
