# A/B of ONE library build under two environments: ENVS="name=VAR=1 ..." (a name with an empty setting = the defaults), one bench line each
WL=${WL:-c3}; SAMPLES=${SAMPLES:-12}
for spec in ${ENVS:-default=}; do
  name=${spec%%=*}; setting=${spec#*=}
  env $setting timeout 600 python bench.py --lanes ${LANES:-1} --no-two-streams --workload $WL --samples $SAMPLES ${READS:+--reads $READS} --steps 2 --warmup 1 --no-cpu-baseline --no-from-host --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$name', '$WL', 'ms/step %.2f' % d['ms_per_step'], {a:round(b,2) for a,b in k.items() if b>0.05}, d['config']['matrix_checksum'])" || echo "$name FAILED"
done
