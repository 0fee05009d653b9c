# round 5: TU kernel A/B in one gpurun call (boxes differ by 5 %: always compare inside one call)
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tu or transform or quant" > gpurun_out/r05_tu_tests.log 2>&1; tail -3 gpurun_out/r05_tu_tests.log
for cfg in "base 5" "new 5" "new 6" "base 5" "new 5" "new 6"; do set -- $cfg
  lib=vvenc_amd/libvvenc_hip.so; [ $1 = base ] && lib=vvenc_amd/libvvenc_hip_base.so
  VVHIP_LIB=$PWD/$lib python bench.py --quick --steps 64 --warmup 32 --streams $2 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$1 streams $2: value %.0f ms_per_step %.4f single_stream %s gop %s' % (d['value'], d['ms_per_step'], d.get('single_stream'), d.get('gop_weighted')))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
