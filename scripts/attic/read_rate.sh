set -e
cd scripts/ubench && hipcc -O2 -o read_rate read_rate.hip -lpthread 2>/dev/null; cd ../..
KEEP=1 python scripts/e2e_full.py 10000000 -ingest-window 8 2>&1 | grep -E "kept|GB/s" | tail -3 > gpurun_out/r04_read_rate.txt
D=$(grep kept gpurun_out/r04_read_rate.txt | awk '{print $2}')
scripts/ubench/read_rate $D 2 >> gpurun_out/r04_read_rate.txt 2>&1
rm -rf $D
cat gpurun_out/r04_read_rate.txt
