"""End-to-end wall time of the `simka` driver on synthetic FASTA files (ingest + H2D + GPU + CSV), next to the ingest alone
(-parse-only).  usage: cli_e2e.py [nb_samples] [reads_per_sample]"""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simka_amd import synth, build as b

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
L = 100
d = tempfile.mkdtemp(prefix="simka_e2e_")
g = synth.genome_len_for(R, L)
pool, gw = synth.genome_pool_cpu(g)
t0 = time.time()
lines = []
for s in range(n):
    ids, cdf = synth.sample_profile(s)
    pk = synth.reads_cpu(R, L, pool, gw, g, ids, cdf, synth.sample_seed(s))
    a = synth.unpack_ascii(pk, R * L).reshape(R, L)
    rec = np.empty((R, L + 4), dtype=np.uint8)          # ">r\n" + read + "\n"
    rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = a; rec[:, 3 + L] = ord("\n")
    fn = os.path.join(d, "s%d.fasta" % s)
    rec.tofile(fn)
    lines.append("S%d: %s" % (s, fn))
open(os.path.join(d, "in.txt"), "w").write("\n".join(lines) + "\n")
size = sum(os.path.getsize(os.path.join(d, "s%d.fasta" % s)) for s in range(n))
print("generated %d FASTA files, %.2f GB in %.1f s" % (n, size / 1e9, time.time() - t0))
base = [b.CLI_PATH, "-in", os.path.join(d, "in.txt"), "-out", os.path.join(d, "out"), "-out-tmp", os.path.join(d, "tmp"), "-kmer-size", "21",
        "-abundance-min", "2", "-max-reads", "-1", "-verbose", "0"]
for extra, name in ((["-parse-only"], "host ingest only"), (["-host-parse"], "end to end, host parse"), ([], "end to end, device parse"), ([], "end to end, device parse (warm)")):
    t = time.time()
    r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stdout
    print("%-20s %.2f s  (%.2f GB/s of FASTA, %.3g k-mer occurrences/s)" % (name, dt, size / dt / 1e9, n * R * (L - 20) / dt))
subprocess.run(["rm", "-rf", d])
