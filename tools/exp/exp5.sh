# round 4, state e: fewer TU workgroups per CU (extra LDS reserved per TU workgroup) against the five-stream step — a TU wave's 168 / 232 registers keep the other streams' kernels off its SIMD
B="python bench.py --no-e2e --no-profile --no-4k --no-mctf --no-cpu-baseline --no-medium"
$B > /dev/null 2>&1
for kv in "A=1" "VVHIP_TU_LDS_PAD=40000" "VVHIP_TU_LDS_PAD=60000" "VVHIP_TU_LDS_PAD=12000" "A=2"; do
  env $kv $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$kv', 'value %.0f step %.2f us gop %.0f single %.0f |' % (d['value'], d['ms_per_step']*1000, d['gop_weighted']['value'], d['single_stream']['value']), ' '.join('%s %.1f' % (k, v['avg_ms_per_picture']*1000) for k,v in d['kernels'].items()), '| by layer', ' '.join('%.0f' % (1000*v) for v in d['gop_weighted']['ms_per_picture_by_layer'].values()), d['parity']['status'])"
done
