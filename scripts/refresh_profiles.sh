# One gpurun call that regenerates the judged artifacts under gpurun_out/profiles_new/ (copy them into profiles/ afterwards):
#   r01_c2_hbm_traffic.json, r01_c2_kernel_stats.csv, r01_c2_bench.json, r01_c2_bench_under_rocprof.json, r01_c3_bench_summary.txt
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profiles_new
mkdir -p $O
bash scripts/pmc_traffic.sh c2 > /dev/null 2>&1
python scripts/traffic_summary.py $R/gpurun_out/traffic c2 $O/r01_c2_hbm_traffic.json
python scripts/rocpd_kernel_stats.py $(ls $R/gpurun_out/traffic/*trace_c2*results.db | head -1) $O/r01_c2_kernel_stats.csv | head -14
cp $R/gpurun_out/traffic/bench_under_rocprof_c2.json $O/r01_c2_bench_under_rocprof.json
# the traffic file must be in profiles/ for bench.py to pick it up
cp $O/r01_c2_hbm_traffic.json $R/profiles/r01_c2_hbm_traffic.json
python bench.py --workload c2 --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/r01_c2_bench.json
cat $O/r01_c2_bench.json | cut -c1-600
{
  echo "# C3 (100 samples x 10M 150 bp reads, k=31, -simple-dist, abundance-min 2) and scaled shapes, 1 MI355X"
  BENCH_TIMEOUT=900 WLS="c3 c3_10 c5_50" bash scripts/bench_c3.sh
} > $O/r01_c3_bench_summary.txt 2>&1
cat $O/r01_c3_bench_summary.txt
