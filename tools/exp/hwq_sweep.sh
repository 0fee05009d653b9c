#!/bin/bash
# bench.py's headline pass (legs A + B, then with leg C) under different numbers of hardware queues and the two lane forms (HotPath.fork: lanes on their contexts' own streams /
# on extra torch streams): value, gop_weighted, value_with_mctf, GOP cycle.   QS="7 8" MODES="own torch" REPS=2 tools/exp/hwq_sweep.sh
cd "$(dirname "$0")/../.."
for rep in $(seq 1 ${REPS:-2}); do
  for mode in ${MODES:-own torch}; do
    for q in ${QS:-7 8}; do
      if [ $mode = torch ]; then export VVHIP_BENCH_TORCH_STREAMS=1; else unset VVHIP_BENCH_TORCH_STREAMS; fi
      GPU_MAX_HW_QUEUES=$q python bench.py --no-e2e --no-profile --no-4k --no-cpu-baseline --no-medium ${EXTRA} > /dev/null 2>&1
      python -c "
import json; d=json.load(open('bench_detail.json')); w=d.get('with_mctf',{}); g=w.get('gop_cycle',{})
print('lanes $mode, queues $q: value %.0f gop_weighted %.0f | with_mctf %.0f cycle %.3f ms (without %.3f)' % (d['value'], d['gop_weighted']['value'], d.get('value_with_mctf',0), g.get('ms_per_cycle',0), g.get('ms_per_cycle_without_mctf',0)))"
    done
  done
done
