# rocprofv3 kernel + memory-copy trace of the driver on C3 at full depth (files kept by scripts/e2e_full.py with KEEP=1)
export TMPDIR=/tmp
R=$PWD
KEEP=1 python scripts/e2e_full.py 10000000 -ingest-window 4 2>&1 | grep -E "kept|GB/s" | tail -2
D=$(ls -d /tmp/simka_e2e_* | head -1)
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/e2e_trace -o t -- $R/simka_amd/bin/simka -in $D/in.txt -out $D/out2 -out-tmp $D/tmp2 -kmer-size 31 -abundance-min 2 -simple-dist -max-reads -1 -verbose 2 2>&1 | grep -E "main thread|bound" 
cd $R
find /tmp/e2e_trace -name "*stats*" | head
for f in $(find /tmp/e2e_trace -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv" -o -name "*domain_stats.csv"); do echo == $f; head -12 $f | cut -c1-170; done
rm -rf $D /tmp/e2e_trace
