# N=500 (tiled pair accumulator) at small read depth: simple-only vs simple+complex
timeout ${BENCH_TIMEOUT:-200} python bench.py --workload c3 --samples 500 --reads 100000 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('n500 simple ms/step', round(d['ms_per_step'],1), {a:round(b,1) for a,b in k.items() if b>1})"
BENCH_TIMEOUT=${BENCH_TIMEOUT:-200} WLS=c5_50 bash scripts/bench_c3.sh
