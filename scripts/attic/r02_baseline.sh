# round-2 baseline before the kernel rework: GPU tests, the C3 bench line, rocprofv3 kernel stats of C3
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_base
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python - > $O/comm.log 2>&1 <<'PY'
import torch, simka_amd
from simka_amd.api import Comm
c = Comm(Comm.unique_id(), 1, 0, 0)
t = torch.ones(4, dtype=torch.int64, device="cuda")
c.allreduce_u64(t.data_ptr(), 4, None)
torch.cuda.synchronize()
print("comm world=1 ok", t.tolist())
c.close()
PY
tail -2 $O/comm.log
( time timeout 1200 python bench.py --steps 3 --warmup 1 ) > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 1500 $O/bench_c3.json; tail -5 $O/bench_c3.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c3_rocprof.json 2> $O/rocprof.err
cd $R
ls $O/prof | head
python scripts/rocpd_kernel_stats.py $(ls $O/prof/*results.db | head -1) $O/r02_base_c3_kernel_stats.csv 2>&1 | tail -3
head -12 $O/r02_base_c3_kernel_stats.csv
rm -rf $O/prof/*.db
