#!/bin/bash
# First contact with a multi-GPU node: the N > 1 path of the C ABI, one stage at a time, each under a timeout; prints the FIRST stage
# that fails and stops.  usage: bash scripts/first_contact.sh [ranks=2]
#   1  all-reduce of u64 words through simka_comm_allreduce_u64 (RCCL communicator of the C ABI, bootstrapped by torch.distributed)
#   2  uneven all-to-all through simka_comm_alltoallv
#   3  bench.py --gpus N --mgpu partition on c2 (partition shards + ONE all-reduce, north_star's split)
#   4  bench.py --gpus N --mgpu sample on c2 (sample shards + spectrum all-to-all + one all-reduce)
#   5  the C++ driver: simka -nb-gpus N -gpu-shards partition on the reference's example (N contexts in ONE process, RCCL communicators created by
#      its worker threads from one unique id, simka_totals_allreduce + simka_stats_allreduce_head); the 20 CSVs must equal tests/golden/truth.
#      SIMKA_BENCH_BACKEND=gloo: -gpu-shared (one device, the host sums)
# The matrix checksums of 3 and 4 must equal the one-rank run's.  SIMKA_BENCH_BACKEND=gloo: dry run on a one-GPU box (the ranks share
# GPU 0, gloo instead of RCCL) -- what the GPU test suite runs.
N=${1:-2}
T=${FIRST_CONTACT_TIMEOUT:-600}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 200)) "$@"; }
fail() { echo "[first-contact] FAILED at stage: $1"; exit 1; }
echo "[first-contact] $N ranks, backend ${SIMKA_BENCH_BACKEND:-nccl}"
timeout $T python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 scripts/first_contact_comm.py allreduce || fail "1 all-reduce (simka_comm_allreduce_u64)"
timeout $T python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29612 scripts/first_contact_comm.py alltoallv || fail "2 all-to-all (simka_comm_alltoallv)"
ref=$(timeout $T python bench.py --gpus 1 --steps 1 --warmup 1 --workload c2 --reads 200000 --no-cpu-baseline --no-e2e --no-two-streams --no-from-host 2>/dev/null | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['config']['matrix_checksum'])") || fail "0 one-rank reference run of bench.py"
[ -n "$ref" ] || fail "0 one-rank reference run of bench.py (no line)"
for MODE in partition sample; do
  out=$(timeout $T python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus $N --steps 1 --warmup 1 --workload c2 --reads 200000 --no-cpu-baseline --mgpu $MODE 2>/dev/null | grep '^{')
  [ -n "$out" ] || fail "bench.py --gpus $N --mgpu $MODE (no line)"
  got=$(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['matrix_checksum'], d['n_gpus'], '%.1f' % d['ms_per_step'])")
  set -- $got
  [ "$1" = "$ref" ] || fail "bench.py --gpus $N --mgpu $MODE (matrix checksum $1, one rank: $ref)"
  echo "[first-contact] bench.py --gpus $2 --mgpu $MODE ok: checksum $1, $3 ms per step"
done
O=$(mktemp -d)
SHARED=""; [ "${SIMKA_BENCH_BACKEND:-nccl}" = gloo ] && SHARED="-gpu-shared"
timeout $T simka_amd/bin/simka -in tests/golden/example/simka_input.txt -out $O/out -out-tmp $O/tmp -kmer-size 31 -abundance-min 2 -simple-dist -complex-dist \
    -nb-gpus $N -gpu-shards partition $SHARED > $O/log.txt 2>&1 || { tail -5 $O/log.txt; fail "5 simka -nb-gpus $N -gpu-shards partition"; }
grep -q "partition shards, one all-reduce" $O/log.txt || fail "5 simka -gpu-shards partition (the driver took another route)"
for f in $O/out/*.csv.gz; do
  b=$(basename $f .gz)
  [ -f tests/golden/truth/results_k31_t2/$b ] || continue
  zcat $f | cmp -s - tests/golden/truth/results_k31_t2/$b || fail "5 simka -gpu-shards partition ($b differs from the golden)"
done
echo "[first-contact] simka -nb-gpus $N -gpu-shards partition ok ($(grep -o 'one all-reduce [a-zA-Z ]*' $O/log.txt | head -1)): goldens byte for byte"
rm -rf $O
echo "[first-contact] all stages passed"
