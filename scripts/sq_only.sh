# SQ counters per kernel for one workload (two --pmc passes), printed: WL=c2_k33 bash scripts/sq_only.sh
export TMPDIR=/tmp
R=$PWD
WL=${WL:-c2_k33}
T=/tmp/sq_$WL
mkdir -p $T
ARGS="--workload $WL --no-cpu-baseline --no-two-streams --no-e2e"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $T -o sqa_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_sqa.txt
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $T -o sqb_$WL -- python $R/bench.py $ARGS --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_sqb.txt
cd $R
python scripts/sq_summary.py $T $WL
