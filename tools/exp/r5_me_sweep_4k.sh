for cfg in "80 16" "160 16" "320 16" "80 32" "80 8" "80 16"; do set -- $cfg
  VVHIP_ME_BUNDLE_WORK=$1 VVHIP_ME_CAND_CAP=$2 python bench.py --quick --steps 32 --warmup 8 --width 3840 --height 2160 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('4K bundle $1 candcap $2: value %.0f ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
  python - <<PY
import json; d = json.load(open('bench_detail_ab.json')); print('   kernels us:', {k: round(v['avg_ms_per_picture'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
done
