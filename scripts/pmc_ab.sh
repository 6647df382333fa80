# SQ counters of one kernel for several library builds: LIBS="name=path ..." KERNEL=k_skm_count_fast WL=c3 SAMPLES=4 bash scripts/pmc_ab.sh
export TMPDIR=/tmp
R=$PWD; WL=${WL:-c3}; SAMPLES=${SAMPLES:-4}; KERNEL=${KERNEL:-k_skm_count_fast}
T=$R/gpurun_out/pmc_ab; rm -rf $T; mkdir -p $T
for spec in ${LIBS:-default=}; do
  name=${spec%%=*}; path=${spec#*=}
  [ -n "$path" ] && export SIMKA_LIB_OVERRIDE=$R/$path || unset SIMKA_LIB_OVERRIDE
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES} --output-format csv -d $T -o pmc_$name -- python $R/bench.py --workload $WL --samples $SAMPLES --no-cpu-baseline --no-two-streams --no-from-host --no-e2e --steps 1 --warmup 0 --prof-steps 1 > /dev/null 2> $T/err_$name.txt)
  python - "$T" "$name" "$KERNEL" <<'P'
import csv, glob, sys, collections
T, name, kern = sys.argv[1:4]
acc = collections.Counter(); n = set()
for f in glob.glob(T + "/**/*pmc_%s*counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(kern + "<") or r["Kernel_Name"].split("(")[0] == kern or kern in r["Kernel_Name"].split("(")[0]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print(name, kern, "launches", len(n), {k: "%.3g" % v for k, v in sorted(acc.items())})
P
done
