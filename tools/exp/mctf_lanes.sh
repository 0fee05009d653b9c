#!/bin/bash
# leg C's four jobs of a GOP cycle dealt over 1 / 2 / 4 lanes (streams) in bench.py ($VVHIP_BENCH_MCTF_LANES), 12 hardware queues so that no stream shares one
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for n in 1 2 4; do
    VVHIP_BENCH_MCTF_LANES=$n GPU_MAX_HW_QUEUES=12 python bench.py --no-e2e --no-profile --no-4k --no-cpu-baseline --no-medium > /dev/null 2>&1
    python -c "
import json; d=json.load(open('bench_detail.json')); w=d.get('with_mctf',{}); g=w.get('gop_cycle',{})
print('MCTF lanes $n: value %.0f | with_mctf %.0f cycle %.3f ms (without %.3f) parity_mctf %s' % (d['value'], d.get('value_with_mctf',0), g.get('ms_per_cycle',0), g.get('ms_per_cycle_without_mctf',0), d.get('parity_mctf',{}).get('status')))"
  done
done
