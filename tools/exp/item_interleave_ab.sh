#!/bin/bash
# A/B of the table calls' band-major class interleave ($VVHIP_ME_ITEM_INTERLEAVE = sub-bands per XCD eighth; 0 = classes in turn) on the recorded 4K and 1080p lists
for res in "3840 2160" "1920 1080"; do
  set -- $res
  for v in 16 0 4 64 16 0; do
    VVHIP_ME_ITEM_INTERLEAVE=$v python bench.py --quick --width $1 --height $2 > /dev/null 2>&1
    python -c "
import json; d=json.load(open('bench_detail.json')); k=d['kernels']['ME_item']
print('$1x$2 interleave $v: value', round(d['value'],1), 'gop_weighted', round(d['gop_weighted']['value'],1), 'ME_item us', round(k['avg_ms_per_picture']*1e3,2), k['ms_by_layer'])"
  done
done
