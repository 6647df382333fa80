# One gpurun call that regenerates the judged round-6 artifacts under gpurun_out/profiles_new/ (copy them into profiles/ afterwards):
#   r06_<wl>_hbm_traffic.json   FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, FETCH x2 on gfx950)
#   r06_<wl>_kernel_stats.csv   rocprofv3 --kernel-trace summary of the bench command
#   r06_<wl>_sq_counters.txt    SQ counters per kernel (two --pmc passes of 8 SQ slots): what bounds each kernel
#   r06_<wl>_bench.json         the bench line (reads the traffic file), r06_<wl>_bench_under_rocprof.json the line of the traced run
export TMPDIR=/tmp
R=$PWD
WL=${WL:-c3}
O=$R/gpurun_out/profiles_new
T=$R/gpurun_out/traffic
mkdir -p $O $T
ARGS="--workload $WL --no-cpu-baseline --no-two-streams --no-from-host --no-e2e"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T -o fetch_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_fetch.txt
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T -o write_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_write.txt
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $T -o sqa_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_sqa.txt
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $T -o sqb_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_sqb.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $T -o trace_$WL -- python $R/bench.py $ARGS --steps 3 --warmup 1 > $O/r06_${WL}_bench_under_rocprof.json 2> $T/err_trace.txt
cd $R
python scripts/traffic_summary.py $T $WL $O/r06_${WL}_hbm_traffic.json
python scripts/rocpd_kernel_stats.py $(find $T -name "*trace_${WL}*results.db" | head -1) $O/r06_${WL}_kernel_stats.csv | head -3
python scripts/sq_summary.py $T $WL $O/r06_${WL}_sq_counters.json > $O/r06_${WL}_sq_counters.txt; cp $O/r06_${WL}_sq_counters.json $R/profiles/
cat $O/r06_${WL}_sq_counters.txt
head -12 $O/r06_${WL}_kernel_stats.csv
cp $O/r06_${WL}_hbm_traffic.json $R/profiles/r06_${WL}_hbm_traffic.json
timeout 1500 python bench.py --workload $WL --steps 5 --warmup 1 2> $O/bench.err | tail -1 > $O/r06_${WL}_bench.json
cut -c1-700 $O/r06_${WL}_bench.json
rm -rf $T
