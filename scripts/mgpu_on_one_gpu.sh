# The N>1 bench path end to end on a ONE-GPU box: 2 (and 3) ranks share GPU 0, torch.distributed on gloo instead of RCCL.
# Compares the printed job totals (distinct / solid k-mers) and a distance-matrix checksum with the 1-rank run.
export SIMKA_BENCH_BACKEND=gloo
W=${1:-c2}
for NP in 1 2 3; do
  for MODE in sample partition; do
    if [ $NP = 1 ] && [ $MODE = partition ]; then continue; fi
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port $((29500 + NP)) \
      bench.py --gpus $NP --steps 1 --warmup 1 --workload $W --reads 200000 --no-cpu-baseline --mgpu $MODE 2>&1 | grep '^{' | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('ranks', d['n_gpus'], '$MODE', 'ms/step %.1f' % d['ms_per_step'], 'distinct %d solid %d occ %d' % (c['distinct_kmers'], c['solid_kmers'], c['kmer_occurrences']), 'checksum', c.get('matrix_checksum'))" || echo "ranks $NP $MODE FAILED"
  done
done
