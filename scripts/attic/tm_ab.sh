# A/B runs of a bench workload (default c5_50: the tiled pair accumulator) under environment settings, one per argument, e.g.
#   bash scripts/tm_ab.sh SIMKA_X=1 SIMKA_PAIRS_LEGACY=1 SIMKA_TM_SPAN=3072      (WL=c3_10 selects another workload)
run() { echo "== $*"; env "$@" timeout 300 python bench.py --workload ${WL:-c5_50} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('ms/step %.1f' % d['ms_per_step'], d['config'].get('matrix_checksum'), {k: round(v,1) for k,v in r['kernel_ms_per_step'].items() if v > 5})"; }
for v in "$@"; do run $v; done
