import sys, json, subprocess
sys.path.insert(0, "/root/repo")
import bench
bench.WORKLOADS["c5_50"] = dict(bench.WORKLOADS["c5_50"], complex=False)
sys.argv = ["bench.py", "--workload", "c5_50", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-two-streams", "--no-from-host"]
bench.main()
