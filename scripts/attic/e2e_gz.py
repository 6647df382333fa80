"""the driver on .fastq.gz inputs (100 samples listing 10 distinct files of 1M x 150 bp reads), -verbose 2, and -parse-only: where the host-parse route spends its time"""
import os, subprocess, sys, time, tempfile, shutil, multiprocessing
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, simka_amd, bench
from simka_amd import build as b
n, D, R, L, k = int(os.environ.get("NS", "100")), int(os.environ.get("ND", "10")), 1_000_000, 150, 31
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
d = tempfile.mkdtemp(prefix="simka_gz_")
try:
    _, reads = bench.gen_device_samples(lib, torch, dict(n=D, reads=R, L=L), dev)
    lut = torch.tensor([ord(c) for c in "ACTG"], dtype=torch.uint8, device=dev)
    sh = torch.arange(32, device=dev, dtype=torch.int64) * 2
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(64)
    for s in range(D):
        w = reads[s][: (R * L + 31) // 32]
        codes = ((w[:, None] >> sh[None, :]) & 3).reshape(-1)[: R * L]
        rec = torch.empty((R, 2 * L + 7), dtype=torch.uint8, device=dev)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = lut[codes].reshape(R, L)
        rec[:, 3 + L] = ord("\n"); rec[:, 4 + L] = ord("+"); rec[:, 5 + L] = ord("\n"); rec[:, 6 + L: 6 + 2 * L] = ord("I"); rec[:, 6 + 2 * L] = ord("\n")
        raw = rec.cpu().numpy(); rows = R // 64
        parts = list(pool.map(bench._gz_member, [raw[r0: r0 + rows] for r0 in range(0, R, rows)]))
        open(os.path.join(d, "q%d.fastq.gz" % s), "wb").write(b"".join(parts))
    pool.shutdown(); del reads; torch.cuda.empty_cache()
    open(os.path.join(d, "in.txt"), "w").write("".join("S%d: %s\n" % (s, os.path.join(d, "q%d.fastq.gz" % (s % D))) for s in range(n)))
    base = [b.CLI_PATH, "-in", os.path.join(d, "in.txt"), "-out", os.path.join(d, "out"), "-out-tmp", os.path.join(d, "tmp"), "-kmer-size", str(k),
            "-abundance-min", "2", "-simple-dist", "-max-reads", "-1", "-verbose", "2"]
    for extra in [x.split() for x in os.environ.get("RUNS", ";-nb-cores 32;").split(";")]:
        time.sleep(6)
        t = time.time()
        r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(extra, "%.2f s" % (time.time() - t), "rc", r.returncode, flush=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("main thread") or ln.startswith("loader threads") or "parse" in ln.lower() and "s" in ln:
                print("   ", ln[:300])
finally:
    shutil.rmtree(d, ignore_errors=True)
