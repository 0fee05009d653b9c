# round 5: what each launch group costs the five-stream step (recorded 1080p lists): the step without it
for skip in "" stage int item tu dmvr "stage,int" "item,tu,dmvr"; do
  VVHIP_BENCH_SKIP_LANES=$skip python bench.py --quick --no-parity --steps 64 --warmup 32 --detail bench_detail_ab.json 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('without [%s]: ms_per_step %.4f  gop_weighted %s' % ('$skip', d['ms_per_step'], d.get('gop_weighted')))"
done
