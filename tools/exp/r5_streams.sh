for cfg in "5 8" "3 8" "5 4" "5 16" "5 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$2 python bench.py --quick --no-parity --steps 64 --warmup 32 --streams $1 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('streams $1 hw queues $2: value %.0f ms_per_step %.4f gop %.0f' % (d['value'], d['ms_per_step'], d['gop_weighted']['value']))"
done
