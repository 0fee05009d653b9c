# round 4, state d: stage bundle size and candidates per window again after the instruction work / the candidate de-duplication (quick bench lines, one box)
B="python bench.py --no-e2e --no-profile --no-4k --no-mctf --no-cpu-baseline --no-medium"
$B > /dev/null 2>&1
for kv in "A=1" "VVHIP_ME_BUNDLE_WORK=60" "VVHIP_ME_BUNDLE_WORK=100" "VVHIP_ME_CAND_CAP=24" "VVHIP_ME_CAND_CAP=32" "VVHIP_ME_CAND_CAP=8" "A=2"; do
  env $kv $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$kv', 'value %.0f step %.2f us gop %.0f single %.0f |' % (d['value'], d['ms_per_step']*1000, d['gop_weighted']['value'], d['single_stream']['value']), ' '.join('%s %.1f' % (k, v['avg_ms_per_picture']*1000) for k,v in d['kernels'].items()))"
done
