# FETCH_SIZE / WRITE_SIZE of scripts/ubench/fetch_calib against the bytes its kernels are known to move (run on the GPU box)
export TMPDIR=/tmp
R=$PWD; T=$R/gpurun_out/fetch_calib; rm -rf $T; mkdir -p $T
(cd /tmp && $R/scripts/ubench/fetch_calib > $T/known.txt && for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T -o pmc_$c -- $R/scripts/ubench/fetch_calib > /dev/null 2> $T/err_$c.txt; done)
python3 - "$T" <<'P'
import csv, glob, sys, collections
T = sys.argv[1]
known = {}
for line in open(T + "/known.txt"):
    w = line.split()
    if w and w[0] == "known":
        i = w.index("read")
        known[" ".join(w[1:i])] = {"read": int(w[i + 1]), "write": int(w[i + 3]), "payload": int(w[i + 5]) if len(w) > i + 5 else None}
cnt = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(T + "/**/*pmc_%s*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            cnt[name][r["Counter_Name"]] = cnt[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("# rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB x 1024) against the bytes each kernel is known to move (whole 128-byte lines), MI355X")
print("%-28s %14s %14s %8s %14s %14s %8s" % ("kernel", "known read", "FETCH_SIZE", "ratio", "known write", "WRITE_SIZE", "ratio"))
for k, v in known.items():
    c = cnt.get(k, {})
    f, w = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    print("%-28s %14d %14.0f %8s %14d %14.0f %8s" % (k, v["read"], f, ("%.3f" % (f / v["read"])) if v["read"] else "-", v["write"], w, ("%.3f" % (w / v["write"])) if v["write"] else "-"))
P
