# kernel-level view of the driver's device-side ingest: e2e_big.py's files, `simka` under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/ing_prof
mkdir -p $O
KEEP=1 python $R/scripts/e2e_big.py 20 1000000 10 > $O/e2e.txt 2>&1
D=$(grep "^kept" $O/e2e.txt | cut -d' ' -f2)
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o ing -- $R/simka_amd/bin/simka -in $D/in.txt -out $D/out -out-tmp $D/tmp -kmer-size 31 -abundance-min 2 -simple-dist -max-reads -1 -verbose 0 > $O/run.txt 2>&1
cd $R
python scripts/rocpd_kernel_stats.py $(find $O -name "*results.db" | head -1) $O/ing_kernel_stats.csv > /dev/null
head -14 $O/ing_kernel_stats.csv | cut -c1-160
rm -rf $D $O/*.db $O/*/*.db
