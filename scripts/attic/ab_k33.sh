for spec in "base=simka_amd/lib/libsimka_base.so:" "tsl8=simka_amd/lib/libsimka_tsl8.so:" "tsl8p96=simka_amd/lib/libsimka_tsl8.so:96" "tsl8p128=simka_amd/lib/libsimka_tsl8.so:128" "base96=simka_amd/lib/libsimka_base.so:96"; do
  name=${spec%%=*}; rest=${spec#*=}; path=${rest%%:*}; pp=${rest#*:}
  export SIMKA_LIB_OVERRIDE=$PWD/$path
  [ -n "$pp" ] && export SIMKA_WIDE_PER_PART=$pp || unset SIMKA_WIDE_PER_PART
  timeout 600 python bench.py --lanes 1 --no-two-streams --workload c2_k33 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-from-host 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$name', 'ms/step %.2f' % d['ms_per_step'], {a:round(b,2) for a,b in k.items() if b>0.05}, d['config']['matrix_checksum'])" || echo "$name FAILED"
done
