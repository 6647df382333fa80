"""the driver on C3 at full depth (10 distinct 1.54-GB FASTA files listed 10 times each), -verbose 2, several reader windows: usage e2e_full.py [reads] [extra driver args ...]"""
import os, subprocess, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, simka_amd, bench
from simka_amd import build as b
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n, D, L, k = 100, 10, 150, 31
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
d = tempfile.mkdtemp(prefix="simka_e2e_")
try:
    _, reads = bench.gen_device_samples(lib, torch, dict(n=D, reads=R, L=L), dev)
    lut = torch.tensor([ord(c) for c in "ACTG"], dtype=torch.uint8, device=dev)
    sh = torch.arange(32, device=dev, dtype=torch.int64) * 2
    for s in range(D):
        rec = torch.empty((R, L + 4), dtype=torch.uint8, device=dev)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3 + L] = ord("\n")
        step = 1 << 20
        w = reads[s]
        for r0 in range(0, R, step):
            r1 = min(R, r0 + step)
            ws = w[(r0 * L) // 32: (r1 * L + 31) // 32 + 1]
            codes = ((ws[:, None] >> sh[None, :]) & 3).reshape(-1)
            o0 = r0 * L - ((r0 * L) // 32) * 32
            rec[r0:r1, 3:3 + L] = lut[codes[o0: o0 + (r1 - r0) * L]].reshape(r1 - r0, L)
        rec.cpu().numpy().tofile(os.path.join(d, "f%d.fasta" % s))
        reads[s] = None
    del reads; torch.cuda.empty_cache()
    open(os.path.join(d, "in.txt"), "w").write("".join("S%d: %s\n" % (s, os.path.join(d, "f%d.fasta" % (s % D))) for s in range(n)))
    size = n * R * (L + 4)
    base = [b.CLI_PATH, "-in", os.path.join(d, "in.txt"), "-out", os.path.join(d, "out"), "-out-tmp", os.path.join(d, "tmp"), "-kmer-size", str(k),
            "-abundance-min", "2", "-simple-dist", "-max-reads", "-1", "-verbose", "2"]
    for extra in ([], [], ["-ingest-window", "8"], ["-ingest-host-upload"], ["-no-numa-bind"], []) if len(sys.argv) <= 2 else [sys.argv[2:]]:
        time.sleep(float(os.environ.get('PAUSE', '10')))      # (the driver of the run before leaves 100+ GB of device memory to be reclaimed: back-to-back runs wait for it)
        def cpu_stat():
            try:
                return {a: int(b_) for a, b_ in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
            except Exception:
                return {}
        c0 = cpu_stat()
        t = time.time()
        r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = time.time() - t
        c1 = cpu_stat()
        print(extra, "%.2f s  %.2f GB/s" % (dt, size / dt / 1e9), "rc", r.returncode,
              "| CPU time %.1f s (%.1f CPUs busy), throttled periods %d, throttled %.1f s" % ((c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e6, (c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e6 / dt,
                                                                                             c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0), (c1.get("throttled_usec", 0) - c0.get("throttled_usec", 0)) / 1e6), flush=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("main thread") or ln.startswith("process:"):
                print("   ", ln)
finally:
    if os.environ.get("KEEP"):
        print("kept", d)
    else:
        shutil.rmtree(d, ignore_errors=True)
