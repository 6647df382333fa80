for v in "$@"; do
  echo "== $v"
  SIMKA_LIB_OVERRIDE=simka_amd/lib/libsimka_$v.so python bench.py --workload c2_k33 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-two-streams --prof-steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernel_ms_per_step']; print(round(d['ms_per_step'],2), d['config']['matrix_checksum'], {a:round(b,2) for a,b in k.items() if b})"
done
