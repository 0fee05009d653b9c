# round 5: the all-zero-level shortcut of the fused TU kernel — parity first, then A/B (base = the previous commit's library) in one call
python -m pytest tests/test_gpu_corners.py tests/test_gpu_parity.py -q -m gpu -k "tu or transform or quant" > gpurun_out/r05_tu_zero_tests.log 2>&1; tail -3 gpurun_out/r05_tu_zero_tests.log
for cfg in base new base new; do
  lib=vvenc_amd/libvvenc_hip.so; [ $cfg = base ] && lib=vvenc_amd/libvvenc_hip_base.so
  VVHIP_LIB=$PWD/$lib python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$cfg: value %.0f ms_per_step %.4f single_stream %s parity %s' % (d['value'], d['ms_per_step'], d.get('single_stream'), d.get('parity')))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()}, 'TU by layer', d['kernels']['TU']['ms_by_layer'])
PY
done
for cfg in base new; do
  lib=vvenc_amd/libvvenc_hip.so; [ $cfg = base ] && lib=vvenc_amd/libvvenc_hip_base.so
  VVHIP_LIB=$PWD/$lib python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$cfg 4K: value %.0f ms_per_step %.4f parity %s' % (d['value'], d['ms_per_step'], d.get('parity')))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
