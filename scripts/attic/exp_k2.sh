run() { python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if b>0.3})"; }
run base
SIMKA_L1=9 run l1_9
SIMKA_L1=10 run l1_10
SIMKA_L1=11 run l1_11
SIMKA_L1=7 run l1_7
