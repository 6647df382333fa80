# quick A/B of the count pipeline on the GPU box: workloads x env settings -> one line each
for wl in ${WLS:-c2 c3_10}; do
  timeout 400 python bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/skm_$wl.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/skm_$wl.json").read())
    print("$wl", "${TAG:-}", round(d["ms_per_step"],1), "ms/step", "%.3g"%d["value"], {k:round(v,2) for k,v in d["roofline"]["kernel_ms_per_step"].items() if v>0.05}, d["config"]["matrix_checksum"])
except Exception as e:
    print("$wl FAILED", e, open("gpurun_out/skm_$wl.json").read()[-600:])
PY
done
