export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --lanes 1 --no-two-streams --workload c3 --samples 12 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-from-host"
$B > /dev/null 2>&1
mkdir -p $R/gpurun_out/gap; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/gap -o gap -- $B 2>/dev/null | tail -1 | cut -c1-200
ls -la $R/gpurun_out/gap | head
