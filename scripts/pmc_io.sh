# HBM traffic of the kernels (FETCH_SIZE / WRITE_SIZE in separate passes), a few C3-size samples
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_io
mkdir -p $O
ARGS="--workload ${WL:-c3} --samples ${SAMPLES:-4} --steps 1 --warmup 0 --no-cpu-baseline"
cd /tmp
SIMKA_LANES=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o f -- python $R/bench.py $ARGS > /dev/null 2> $O/err_f.txt
SIMKA_LANES=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o w -- python $R/bench.py $ARGS > /dev/null 2> $O/err_w.txt
cd $R
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out/pmc_io")
res = collections.defaultdict(dict)
for tag, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    fs = glob.glob(O + "/**/%s_counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != cn: continue
        k = r["Kernel_Name"].split("(")[0][:44]
        acc[k] += float(r["Counter_Value"]) * 1024.0; n[k] += 1
    for k in acc: res[k][cn] = acc[k] / n[k]; res[k]["n"] = n[k]
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    if k.startswith("k_") or "k_skm" in k or "k_pairs" in k:
        print("%-46s launches %4d  fetch x2 %8.1f MB  write %8.1f MB per launch" % (k, v["n"], 2 * v.get("FETCH_SIZE", 0) / 1e6, v.get("WRITE_SIZE", 0) / 1e6))
PY
