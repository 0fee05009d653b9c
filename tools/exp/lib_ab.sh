#!/bin/bash
# A/B of two builds of the device library on the recorded lists (1080p and 4K): tools/exp/_ab/libvvenc_hip_old.so (built from an earlier me.hip) against the tree's library.
# Per run: value, gop_weighted, per-class launch time.  Usage (GPU box): tools/exp/lib_ab.sh [reps]
cd "$(dirname "$0")/../.."
for res in "1920 1080" "3840 2160"; do
  set -- $res
  for rep in $(seq 1 ${1:+${REPS:-2}}); do
    for lib in old new; do
      if [ $lib = old ]; then export VVHIP_LIB=$PWD/tools/exp/_ab/libvvenc_hip_old.so; else unset VVHIP_LIB; fi
      python bench.py --quick --width $1 --height $2 > /dev/null 2>&1
      python -c "
import json; d=json.load(open('bench_detail.json')); k=d['kernels']
print('$1x$2 $lib: value', round(d['value'],1), 'gop_weighted', round(d['gop_weighted']['value'],1), ' '.join('%s %.2f' % (c, k[c]['avg_ms_per_picture']*1e3) for c in ('ME_stage','ME_item','ME_int','TU','DMVR') if c in k))"
    done
  done
done
