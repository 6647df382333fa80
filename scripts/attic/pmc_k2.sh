export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
for mode in fast slow; do
  if [ $mode = slow ]; then export SIMKA_K2_SLOW=1; else unset SIMKA_K2_SLOW; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc -o sq_$mode -- python bench.py --workload c2 --samples 3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc/err_$mode.txt
  rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc -o sq2_$mode -- python bench.py --workload c2 --samples 3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/pmc/err_$mode.txt
done
ls -la $R/gpurun_out/pmc | head -30
