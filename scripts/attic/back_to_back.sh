# does a bench process that starts right after another one run slower?  (same library, same workload; ms per step, the summed kernel time,
# and -- SIMKA_BENCH_TRACE=1 -- where the host spends a step)
run() { timeout 600 python bench.py --lanes 1 --no-two-streams --workload c3 --samples 12 --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-e2e --no-from-host 2>gpurun_out/b2b_$1.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.2f' % d['ms_per_step'], 'kernels %.2f' % d['timing']['device_kernels_ms'])"; grep "step trace" gpurun_out/b2b_$1.err | tail -1; }
export SIMKA_BENCH_TRACE=1
run first; run second; run third; STEPS=10 run fourth_10_steps; run fifth
