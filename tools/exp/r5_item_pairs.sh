# round 5: SAD / SSE table calls with two row chunks per lane in flight: 8 waves per SIMD (spills) / 7 waves (no spills) / one chunk (base), 1080p and 4K, in one call
for v in it8_0 it7_1 it8_1 it8_0 it7_1 it8_1; do
  VVHIP_LIB=$PWD/vvenc_amd/libvvenc_hip_$v.so python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$v 1080p: value %.0f ms_per_step %.4f gop %.0f parity %s' % (d['value'], d['ms_per_step'], d['gop_weighted']['value'], d['parity']['status']))"
  python -c "
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()}, 'item by layer', d['kernels']['ME_item']['ms_by_layer'])"
done
for v in it8_0 it7_1 it8_1; do
  VVHIP_LIB=$PWD/vvenc_amd/libvvenc_hip_$v.so python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$v 4K: value %.0f ms_per_step %.4f parity %s' % (d['value'], d['ms_per_step'], d['parity']['status']))"
  python -c "
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})"
done
