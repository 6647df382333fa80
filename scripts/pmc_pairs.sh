export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2 -o a -- python bench.py --workload c3_10 --samples 100 --reads 200000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc2/err_a.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/pmc2 -o b -- python bench.py --workload c3_10 --samples 100 --reads 200000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc2/err_b.txt
ls $R/gpurun_out/pmc2
