# round 5: a wide stage unit's horizontal variants as waves of their own ($VVHIP_ME_SPLIT_VARIANTS=0: one wave, the form up to round 4) — parity, then A/B in one call
python -m pytest tests/test_gpu_me_shapes.py tests/test_gpu_corners.py tests/test_gpu_replay.py -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for sv in 0 1 0 1; do
  VVHIP_ME_SPLIT_VARIANTS=$sv python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('split $sv 1080p: value %.0f ms_per_step %.4f gop %.0f single %.0f parity %s' % (d['value'], d['ms_per_step'], d['gop_weighted']['value'], d['single_stream']['value'], d['parity']['status']))"
  python -c "
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()}, 'stage by layer', d['kernels']['ME_stage']['ms_by_layer'], d['config']['work_per_layer']['5']['plan'])"
done
for sv in 0 1; do
  VVHIP_ME_SPLIT_VARIANTS=$sv python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('split $sv 4K: value %.0f ms_per_step %.4f parity %s' % (d['value'], d['ms_per_step'], d['parity']['status']))"
  python -c "
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})"
done
