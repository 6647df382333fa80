# One gpurun call that regenerates the judged round-2 artifacts under gpurun_out/profiles_new/ (copy them into profiles/ afterwards):
#   r02_c3_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, FETCH x2 on gfx950)
#   r02_c3_kernel_stats.csv  rocprofv3 --kernel-trace summary of the bench command
#   r02_c3_bench.json        the bench line (reads the traffic file), r02_c3_bench_under_rocprof.json the line of the traced run
export TMPDIR=/tmp
R=$PWD
WL=${WL:-c3}
O=$R/gpurun_out/profiles_new
T=$R/gpurun_out/traffic
mkdir -p $O $T
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T -o fetch_$WL -- python $R/bench.py --workload $WL --steps 1 --warmup 0 --prof-steps 1 --no-cpu-baseline --no-two-streams > /dev/null 2> $T/err_fetch.txt
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T -o write_$WL -- python $R/bench.py --workload $WL --steps 1 --warmup 0 --prof-steps 1 --no-cpu-baseline --no-two-streams > /dev/null 2> $T/err_write.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $T -o trace_$WL -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-two-streams > $O/r02_${WL}_bench_under_rocprof.json 2> $T/err_trace.txt
cd $R
python scripts/traffic_summary.py $T $WL $O/r02_${WL}_hbm_traffic.json
python scripts/rocpd_kernel_stats.py $(find $T -name "*trace_${WL}*results.db" | head -1) $O/r02_${WL}_kernel_stats.csv | head -3
head -12 $O/r02_${WL}_kernel_stats.csv
cp $O/r02_${WL}_hbm_traffic.json $R/profiles/r02_${WL}_hbm_traffic.json
timeout 1200 python bench.py --workload $WL --steps 5 --warmup 1 2> $O/bench.err | tail -1 > $O/r02_${WL}_bench.json
cut -c1-700 $O/r02_${WL}_bench.json
rm -f $T/*.db $T/*/*.db
