# A/B of library builds on the GPU box: LIBS="name=path ..." (default: the in-tree build), one bench line each
WL=${WL:-c3}; SAMPLES=${SAMPLES:-6}
for spec in ${LIBS:-default=}; do
  name=${spec%%=*}; path=${spec#*=}
  [ -n "$path" ] && export SIMKA_LIB_OVERRIDE=$PWD/$path || unset SIMKA_LIB_OVERRIDE
  timeout 600 python bench.py --lanes ${LANES:-1} --no-two-streams --workload $WL --samples $SAMPLES ${READS:+--reads $READS} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$name', '$WL', 'ms/step %.2f' % d['ms_per_step'], {a:round(b,2) for a,b in k.items() if b>0.05}, d['config']['matrix_checksum'])" || echo "$name FAILED"
done
