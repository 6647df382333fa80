"""the three end-to-end legs of bench.py (tenth depth, .fastq.gz, full depth) on their own: usage e2e_legs.py [workload]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, simka_amd, bench
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
print(json.dumps(bench.e2e_from_fasta(wl, simka_amd.load_library(), torch, torch.device("cuda:0")), indent=1))
