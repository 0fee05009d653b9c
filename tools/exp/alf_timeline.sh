#!/bin/bash
# per-call timeline of the ALF statistics hooks on the real encoder (GPU box): whole-picture call vs bands, 1080p x 33 at 8 threads
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for mode in 0 1; do
  rm -f /tmp/alf_tl_$mode.txt
  VVHIP_ALF_TIMELINE=/tmp/alf_tl_$mode.txt VVHIP_ALF_BANDS=$mode VVHIP_ALF_MIN_CTUS_PER_THREAD=0 python tests/e2e_fps.py --width ${W:-1920} --height ${H:-1080} --frames ${F:-33} --threads 8 --masks 8336 | python -c "import json,sys; d=json.load(sys.stdin); print('bands=$mode fps', round(d['runs'][0]['fps'],2))"
  cp /tmp/alf_tl_$mode.txt gpurun_out/alf_tl_$mode.txt
  python - <<PY
import statistics
rows=[l.split() for l in open('/tmp/alf_tl_$mode.txt')]
for what in ('row','picture'):
    r=[(float(x[3]),float(x[4])) for x in rows if x[0]==what]
    if not r: continue
    w=[a for a,b in r]; i=sorted(b for a,b in r)
    print(' %s calls %d: inside us median %.0f p90 %.0f max %.0f sum %.0f | lock wait us median %.0f max %.0f' % (what, len(r), statistics.median(i), i[int(len(i)*0.9)], i[-1], sum(i), statistics.median(w), max(w)))
PY
done
