rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower 2>/dev/null | grep -v "^$" | head -30
python -m pytest tests/test_gpu_corners.py tests/test_gpu_parity.py -q -m gpu -k "tu or all_zero" 2>&1 | tail -2
for lib in libvvenc_hip_base libvvenc_hip libvvenc_hip_base libvvenc_hip; do
  for mix in 32:8192 64:2048 16:2048; do echo -n "$lib "; VVHIP_LIB=$PWD/vvenc_amd/$lib.so python tools/tu_mix.py $mix --reps 100 --amp 1 2>&1 | tail -1; done
  VVHIP_LIB=$PWD/vvenc_amd/$lib.so python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$lib: value %.0f ms_per_step %.4f parity %s' % (d['value'], d['ms_per_step'], d['parity']['status']))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
