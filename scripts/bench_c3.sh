for wl in ${WLS:-c3}; do
timeout ${BENCH_TIMEOUT:-900} python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('$wl ms/step', round(d['ms_per_step'],1), 'distinct/s %.3g' % d['value'], 'occ/s %.3g' % d['config']['kmer_occurrences_per_s'], 'path_frac', round(d['roofline']['path_frac'],4), d['config']['geometry'], {a:round(b,1) for a,b in k.items() if b>1})" || echo "$wl FAILED"
done
