# SQ / LDS counters of k_skm_count_fast for several library builds: LIBS="name=path ..."
export TMPDIR=/tmp
R=$PWD
for spec in $LIBS; do
  name=${spec%%=*}; path=${spec#*=}
  O=$R/gpurun_out/pmc_ab/$name; mkdir -p $O
  export SIMKA_LIB_OVERRIDE=$R/$path
  cd /tmp
  SIMKA_LANES=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O -o a -- python $R/bench.py --workload c3 --samples 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/err_a.txt
  SIMKA_LANES=1 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL --output-format csv -d $O -o b -- python $R/bench.py --workload c3 --samples 2 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/err_b.txt
  cd $R
  python - $O $name <<'PY'
import csv, glob, collections, sys, os
O, name = sys.argv[1], sys.argv[2]
for tag in "ab":
    fs = glob.glob(O + "/**/%s_counter_collection.csv" % tag, recursive=True)
    if not fs: print(name, "no csv", tag); continue
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(fs[0])):
        if os.environ.get("KFILTER", "count_fast") not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(name, tag, {a: "%.3g" % b for a, b in sorted(acc.items())})
PY
done
