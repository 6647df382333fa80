"""End-to-end wall time of the `simka` driver at C3's shape and a tenth of its depth: 100 samples x 1M x 150 bp reads as FASTA text
(15.4 GB listed; 10 distinct files of 154 MB, each listed by 10 samples -- the driver reads, parses and counts every listed file),
device-side parser (default) against -host-parse.  usage: e2e_big.py [nb_samples] [reads_per_sample] [distinct_files]"""
import os, subprocess, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simka_amd, bench
from simka_amd import build as b

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L, k = 150, 31
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
d = tempfile.mkdtemp(prefix="simka_e2e_")
try:
    wl = dict(n=D, reads=R, L=L)
    _, reads = bench.gen_device_samples(lib, torch, wl, dev)
    lut = torch.tensor([ord(c) for c in "ACTG"], dtype=torch.uint8, device=dev)
    t0 = time.time()
    for s in range(D):
        w = reads[s][: (R * L + 31) // 32]
        sh = torch.arange(32, device=dev, dtype=torch.int64) * 2
        codes = ((w[:, None] >> sh[None, :]) & 3).reshape(-1)[: R * L]
        asc = lut[codes].reshape(R, L)
        rec = torch.empty((R, L + 4), dtype=torch.uint8, device=dev)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = asc; rec[:, 3 + L] = ord("\n")
        rec.cpu().numpy().tofile(os.path.join(d, "f%d.fasta" % s))
        del codes, asc, rec
    del reads
    torch.cuda.empty_cache()
    open(os.path.join(d, "in.txt"), "w").write("".join("S%d: %s\n" % (s, os.path.join(d, "f%d.fasta" % (s % D))) for s in range(n)))
    size = sum(os.path.getsize(os.path.join(d, "f%d.fasta" % (s % D))) for s in range(n))
    print("%d samples listing %.2f GB of FASTA (%d distinct files written in %.1f s)" % (n, size / 1e9, D, time.time() - t0), flush=True)
    base = [b.CLI_PATH, "-in", os.path.join(d, "in.txt"), "-out", os.path.join(d, "out"), "-out-tmp", os.path.join(d, "tmp"), "-kmer-size", str(k),
            "-abundance-min", "2", "-simple-dist", "-max-reads", "-1", "-verbose", "2"]
    occ = float(n) * R * (L - k + 1)
    runs = [(["-parse-only"], "host ingest only (-parse-only)"), (["-host-parse"], "end to end, host parse"), ([], "end to end, device parse"),
            ([], "end to end, device parse (again)")]
    if os.environ.get("MGPU"):      # the -nb-gpus routes with every context on this one GPU: spectra device-to-device against the host bounce
        runs = [([], "one context"), (["-nb-gpus", "2", "-gpu-shared"], "-nb-gpus 2, spectra on the device"), (["-nb-gpus", "2", "-gpu-shared", "-host-spectra"], "-nb-gpus 2, spectra through the host")]
    sums = []
    for extra, name in runs:
        t = time.time()
        r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = time.time() - t
        assert r.returncode == 0, r.stdout[-2000:]
        for ln in r.stdout.splitlines():
            if ln.startswith("main thread") or ln.startswith("process:"):
                print("   ", ln)
        print("%-34s %7.2f s  %6.2f GB/s of FASTA  %.3g k-mer occurrences/s" % (name, dt, size / dt / 1e9, occ / dt), flush=True)
        if "-parse-only" not in extra:
            import hashlib, glob
            h = hashlib.sha1()
            for f in sorted(glob.glob(os.path.join(d, "out", "*.csv.gz"))):
                h.update(open(f, "rb").read())
            sums.append(h.hexdigest())
    assert len(set(sums)) <= 1, sums
    print("matrices identical across the runs:", sums[0][:16] if sums else None)
finally:
    if os.environ.get("KEEP"):
        print("kept", d)
    else:
        shutil.rmtree(d, ignore_errors=True)
