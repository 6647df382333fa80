# kernel-time summary of bench.py --workload c2_k33 (13 steps under rocprofv3: 2 warm-up + 5 timed + 2 x 3 untimed)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk33 && rocprofv3 --kernel-trace --stats -d /tmp/pk33 -o k33 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-c2_k33} ${KS:+--kmer-size $KS} --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > /tmp/pk33.log 2>&1
grep '^{' /tmp/pk33.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'checksum', d['config']['matrix_checksum'])"
f=$(find /tmp/pk33 -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out && cp $f $GRAFT_REPO_ROOT/gpurun_out/${WL:-c2_k33}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-40s calls %6s  total %9.2f ms  avg %9.1f us  %5.1f %%" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("all kernels: %.1f ms" % (tot / 1e6))
PY
