# insert-strategy micro-benchmark, phase timers of k_skm_count_fast, SQ + HBM counters of the count-side kernels
export TMPDIR=/tmp
O=gpurun_out/r02_probe; mkdir -p $O
timeout 300 scripts/ubench/insert_strategies > $O/insert_strategies.txt 2>&1
cat $O/insert_strategies.txt
SIMKA_LANES=1 SIMKA_LIB_OVERRIDE=$PWD/simka_amd/lib/libsimka_hip_phase.so timeout 400 python bench.py --workload c3 --samples 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/phase.json 2> $O/phase.err
grep phases $O/phase.err | tail -4
SAMPLES=4 bash scripts/pmc_skm.sh 2>&1 | tail -12
SAMPLES=4 bash scripts/pmc_io.sh 2>&1 | tail -12
