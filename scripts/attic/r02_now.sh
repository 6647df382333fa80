# round-2 state check: GPU tests, the C3 bench line, rocprofv3 kernel stats of C3 (super-k-mer pipeline)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_now
mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 1200 python bench.py --steps 3 --warmup 1 ) > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 3000 $O/bench_c3.json; tail -5 $O/bench_c3.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c3_rocprof.json 2> $O/rocprof.err
cd $R
ls $O/prof | head
python scripts/rocpd_kernel_stats.py $(ls $O/prof/*results.db | head -1) $O/r02_c3_kernel_stats.csv 2>&1 | tail -3
head -20 $O/r02_c3_kernel_stats.csv
rm -rf $O/prof/*.db
