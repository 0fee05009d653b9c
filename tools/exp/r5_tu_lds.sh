# round 5: the TU workgroup's LDS 44.5 -> 28.2 KB (64-point instance) / 27.6 -> 19.5 KB: inverse-pass operands read from the records on the long path only
python -m pytest tests/test_gpu_corners.py tests/test_gpu_parity.py -q -m gpu -k "tu or all_zero or transform" 2>&1 | tail -2
for lib in libvvenc_hip_base libvvenc_hip libvvenc_hip_base libvvenc_hip; do
  VVHIP_LIB=$PWD/vvenc_amd/$lib.so python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$lib: value %.0f ms_per_step %.4f gop %.0f parity %s' % (d['value'], d['ms_per_step'], d['gop_weighted']['value'], d['parity']['status']))"
  python -c "
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})"
done
for lib in libvvenc_hip_base libvvenc_hip; do
  VVHIP_LIB=$PWD/vvenc_amd/$lib.so python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$lib 4K: value %.0f ms_per_step %.4f parity %s' % (d['value'], d['ms_per_step'], d['parity']['status']))"
  VVHIP_LIB=$PWD/vvenc_amd/$lib.so python tools/tu_mix.py 64:955,32:2133,16:600,8:800,4:600 --reps 100 | tail -1
done
