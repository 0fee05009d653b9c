# round 5: the TU launch structure on the recorded 4K lists (two launches above 8 192 tiles by default) and the resident-wave budget, after the kernel's register / zero-path work
for cfg in "8192 0" "100000 0" "100000 6144" "100000 8192" "8192 6144" "8192 0"; do set -- $cfg
  VVHIP_TU_ONE_LAUNCH_TILES=$1 VVHIP_TU_RESIDENT_WAVES=$2 python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('4K one-launch-tiles $1 resident $2: value %.0f ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
for cfg in "8192 0" "8192 3072" "8192 6144"; do set -- $cfg
  VVHIP_TU_ONE_LAUNCH_TILES=$1 VVHIP_TU_RESIDENT_WAVES=$2 python bench.py --quick --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('1080p one-launch-tiles $1 resident $2: value %.0f ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
