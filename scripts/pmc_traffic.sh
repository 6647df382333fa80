# HBM traffic of the hot kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), kernel-trace only.
export TMPDIR=/tmp
R=$PWD
WL=${1:-c2}
mkdir -p $R/gpurun_out/traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic -o fetch_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/traffic/err_fetch.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic -o write_$WL -- python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/traffic/err_write.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/traffic -o trace_$WL -- python bench.py --workload $WL --steps 3 --warmup 1 > $R/gpurun_out/traffic/bench_under_rocprof_$WL.json 2> $R/gpurun_out/traffic/err_trace.txt
python bench.py --workload $WL --steps 5 --warmup 1 > $R/gpurun_out/traffic/bench_$WL.json 2>/dev/null
ls $R/gpurun_out/traffic
