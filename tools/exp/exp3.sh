for sf in 0 256; do echo "small-first $sf"; for m in 64:955,32:2133,16:600,8:800,4:600 64:955,32:2133,4:600 64:955,32:2133,16:600,8:25,4:30; do VVHIP_TU_SMALL_FIRST=$sf python tools/tu_mix.py $m 2>/dev/null; done; done
B="python bench.py --no-e2e --no-mctf --no-4k --no-cpu-baseline --no-profile --no-parity"
for bw in 80 160; do VVHIP_ME_BUNDLE_WORK=$bw $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('bw', $bw, round(d['value']), {k:round(v['avg_ms_per_picture']*1000,1) for k,v in d['kernels'].items()})"; done
