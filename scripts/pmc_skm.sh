# SQ counters of the super-k-mer kernels (two passes: 8 SQ slots each), c3_10 with a few samples; prints per-kernel sums
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_skm
mkdir -p $O
ARGS="--workload c3_10 --samples ${SAMPLES:-8} --steps 1 --warmup 0 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O -o a -- python $R/bench.py $ARGS > /dev/null 2> $O/err_a.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O -o b -- python $R/bench.py $ARGS > /dev/null 2> $O/err_b.txt
cd $R
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out/pmc_skm")
for tag in "ab":
    fs = glob.glob(O + "/**/%s_counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("no csv for", tag, glob.glob(O + "/**/*.csv", recursive=True)[:5]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "skm" not in k and "k_group" not in k and "k_pairs" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(tag, k, {a: "%.3g" % b for a, b in v.items()})
PY
