cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not c3-20 and not hang and not first_contact" 2>&1 | tail -3
for WL in $WLS; do
  for spec in $SPECS; do
    name=${spec%%=*}; path=${spec#*=}
    [ -n "$path" ] && export SIMKA_LIB_OVERRIDE=$PWD/$path || unset SIMKA_LIB_OVERRIDE
    timeout 900 python bench.py --no-two-streams --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-from-host --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('$name', '$WL', 'ms/step %.2f' % d['ms_per_step'], {a:round(b,2) for a,b in k.items() if b>1}, d['config']['matrix_checksum'])" || echo "$name FAILED"
  done
done
