#!/usr/bin/env python3
"""bench.py -- distinct k-mers/s of the whole Simka hot path (count -> merge -> N x N matrices) on MI355X.

A "step" = one full job over synthetic reads that are already resident in HBM (2-bit packed):
simka_reset, simka_count_sample x N, simka_merge, [all-reduce of the accumulators over ranks],
download, host finalisation into the float32 distance matrices.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c3_10 ...]

N>1: `python bench.py --gpus N` spawns its own N ranks (one per GPU); it also runs under
`python -m torch.distributed.run --nproc-per-node N ...` as the driver launches it.  Both decompositions of the fixed job are
timed (partition shards + one all-reduce; sample shards + spectrum all-to-all + one all-reduce): strong scaling, `value` is the
whole-job rate of the faster one.  The collectives are RCCL calls made by the C ABI.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# what limits each kernel (SQ / TCC counters, profiles/r05_c3_sq_counters.txt; DESIGN.md sections 4 and 6) -- the path-level roofline is HBM
# bytes, but only k_skm_chunksort is bound by HBM itself
KERNEL_BOUND = {
    "k_skm_scan": "instruction issue + barriers (canonical m-mer hashes, sliding minimum; 24 waves per CU, 11 barriers per tile)",
    "k_skm_split": "HBM (64-byte runs: partial-line writes amplify the traffic)",
    "k_skm_chunksort": "HBM (every record read once and written once in whole lines: a streaming sort of 8192-record chunks in LDS)",
    "k_skm_count_fast": "serial instruction streams of 4 waves per SIMD (1.0e6 instructions of every class per wave at ~4.7 cycles, 35 % parked at barriers / LDS returns; VALU pipe 89 % behind it)",
    "k_skm_count": "LDS atomics (redo list only)",
    "k_skm_count_wide_fast": "LDS latency + VALU issue (two-word k-mers: claim by 64-bit CAS, compare, count; a table per wave, 16 waves per CU)",
    "k_skm_count_wide": "LDS atomics (redo list only)",
    "k_segment_rows": "HBM (imported spectra only)",
    "k_group": "LDS atomics + gather latency (hash grouping of the N slices of a sub-range)",
    "k_pairs": "pair enumeration on a serial instruction stream (~50 instructions per pair; its three LDS atomics are NOT the bound: -3 % without two of them)",
    "k_pairs_global": "L2 atomics",
}

# resident waves per SIMD of the hot kernels (registers / LDS of the shipped build, profiles/r06_isa_mix.txt)
WAVES_PER_SIMD = {"k_skm_count_fast": 4, "k_skm_scan": 6, "k_group": 5, "k_pairs": 4, "k_skm_chunksort": 4, "k_skm_count_wide_fast": 4}

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(n=10, reads=1_000_000, L=100, k=21, amin=2, simple=False,
               desc="10 synthetic samples x 1M 100 bp reads, k=21, Bray-Curtis + Jaccard"),
    # BASELINE.json configs[2] (the configuration the metric/targets are quoted on)
    "c3": dict(n=100, reads=10_000_000, L=150, k=31, amin=2, simple=True,
               desc="100 samples x 10M 150 bp reads, k=31, -simple-dist, abundance-min 2"),
    # 1/10-scale C3 (same shape, 1M reads per sample)
    "c3_10": dict(n=100, reads=1_000_000, L=150, k=31, amin=2, simple=True,
                  desc="100 samples x 1M 150 bp reads (C3 at 1/10 read depth), k=31, -simple-dist"),
    # C2's reads at k = 33: two-word k-mers -- counted per minimizer partition (k_skm_count_wide_fast), merged by hash buckets + LDS grouping (simka_wide.hip)
    "c2_k33": dict(n=10, reads=1_000_000, L=100, k=33, amin=2, simple=False,
                   desc="10 synthetic samples x 1M 100 bp reads, k=33 (two-word k-mers: partitioned count, bucketed merge), Bray-Curtis + Jaccard"),
    # BASELINE.json configs[4] shape at 1/50 read depth: the tiled (N > LDS tile) pair accumulator + -complex-dist
    "c5_50": dict(n=500, reads=100_000, L=150, k=31, amin=2, simple=True, complex=True,
                  desc="500 samples x 100k 150 bp reads (C5 at 1/50 read depth), k=31, -simple-dist -complex-dist"),
    # BASELINE.json configs[4] shape at 1/5 read depth (500 x 1M reads: 19 GB of packed reads + 47 GB of solid spectra fit one GPU)
    "c5_5": dict(n=500, reads=1_000_000, L=150, k=31, amin=2, simple=True, complex=True,
                 desc="500 samples x 1M 150 bp reads (C5 at 1/5 read depth), k=31, -simple-dist -complex-dist"),
}


def gen_device_samples(lib, torch, wl, dev, which=None):
    """genome pool + per-sample reads, generated on the GPU (k_synth_*), 2-bit packed int64 tensors.
    which: the samples this rank needs (None = all); the others stay None."""
    from simka_amd import synth
    R, L, n = wl["reads"], wl["L"], wl["n"]
    g = synth.genome_len_for(R, L)
    gw = (g + 31) // 32
    pool = torch.empty(synth.NB_GENOMES * gw, dtype=torch.int64, device=dev)
    rc = lib.simka_synth_genomes(None, pool.data_ptr(), synth.NB_GENOMES, gw, synth.POOL_SEED)
    assert rc == 0
    nw = (R * L + 31) // 32
    reads = []
    for s in range(n):
        if which is not None and s not in which:
            reads.append(None)
            continue
        ids, cdf = synth.sample_profile(s)
        d_ids = torch.from_numpy(ids.astype(np.int32)).to(dev)
        d_cdf = torch.from_numpy(cdf.view(np.int32)).to(dev)
        t = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
        rc = lib.simka_synth_reads(None, t.data_ptr(), R, L, pool.data_ptr(), gw, g, d_ids.data_ptr(), d_cdf.data_ptr(),
                                   synth.NB_SEL, synth.sample_seed(s), synth.ERR_THRESHOLD16)
        assert rc == 0
        torch.cuda.synchronize()
        reads.append(t)
    return pool, reads


def cpu_baseline(wl, lib, torch, dev, seconds_budget=15.0):
    """The oracle (CPU restatement of Simka's algorithm, kind="port") timed on a BOUNDED sample of the same
    workload: same generator, same coverage model, fewer samples / reads (generated on the GPU, then downloaded)."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from simka_amd import synth
    cores = os.cpu_count() or 1
    n = min(wl["n"], max(2, min(cores, 32)))
    threads = min(cores, 64)
    L, k = wl["L"], wl["k"]
    # the oracle sustains ~2.5M k-mer occurrences per core-second (sort-count + heap merge)
    R = int(min(wl["reads"], max(2000, seconds_budget * 2.5e6 * min(threads, n) / (n * (L - k + 1)))))
    sub = dict(wl, n=n, reads=R)
    _, reads = gen_device_samples(lib, torch, sub, dev)
    o = oracle_lib.Oracle()
    offs = np.arange(R + 1, dtype=np.uint64) * L
    for s in range(n):
        pk = reads[s].cpu().numpy().view(np.uint64)
        o.add_sample_ascii("S%d" % s, synth.unpack_ascii(pk[: (R * L + 31) // 32], R * L), offs)
    del reads
    t0 = time.perf_counter()
    o.run(k, wl["amin"], simple=wl["simple"], nparts=max(threads * 4, 1), threads=threads)
    for w in range(len(o.matrix_names())):
        if w < 15 or (wl["simple"] and w < 18):
            o.matrix(w)
    dt = time.perf_counter() - t0
    tot = o.totals()
    # a container may be limited to fewer CPUs than it sees (cgroup v2 cpu.max = "quota period"): the threads then share that much CPU time
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    return {"value": float(tot["D_all"].sum()) / dt, "unit": "distinct k-mers/s", "cores": threads, "kind": "port",
            "sample": "%d samples x %d reads x %d bp, k=%d, same generator/coverage (%.1f s CPU wall, %.3g k-mer occurrences/s; %d threads on %d visible CPUs%s)" %
                      (n, R, L, k, dt, float(tot["K_occ"].sum()) / dt, threads, cores, ", cgroup CPU quota %.0f" % quota if quota else ""),
            "cpu_quota": quota, "seconds": dt}


def _gz_member(b):
    import zlib
    if not isinstance(b, (bytes, bytearray)):
        b = memoryview(np.ascontiguousarray(b)).cast("B")
    c = zlib.compressobj(1, zlib.DEFLATED, 31)
    return c.compress(b) + c.flush()


def e2e_from_fasta(wl, lib, torch, dev, max_samples=100, max_reads=1_000_000):
    """t_e2e of SURVEY 8(d): sequence files on disk -> distance-matrix CSVs through the C++ `simka` driver (process start, files read into
    small pinned staging buffers by the loader threads and uploaded by them, text parsed on the GPU -- simka_ingest_* --, count, merge,
    matrices, gz CSVs written), three legs (runs 8 s apart: see driver()):
      * `tenth`: the workload at a tenth of its read depth (c3: 100 samples x 1M x 150 bp = 15.4 GB of FASTA, one file per sample when the
        temp directory has room); run twice (the second run has the files in the page cache) and once with -host-parse;
      * `full_depth`: the workload's own depth -- 10 distinct files, each listed by a tenth of the samples (the driver reads, parses and
        counts every listed file; the box's temp space holds 15 GB, not the 154 GB of c3) -- the largest depth this box allows;
      * `fastq_gz`: the tenth-depth reads as .fastq.gz (what real metagenomes look like): gzip goes through the host parser
        (inflate + parse + 2-bit pack on the driver's worker threads).
    The top-level keys (ms, fasta_GBps ...) are the `tenth` leg's, as in round 3."""
    import shutil
    import subprocess
    import tempfile
    from simka_amd import build as b
    n, L, k = min(wl["n"], max_samples), wl["L"], wl["k"]
    d = tempfile.mkdtemp(prefix="simka_e2e_")
    lut = torch.tensor([ord(c) for c in "ACTG"], dtype=torch.uint8, device=dev)
    sh = torch.arange(32, device=dev, dtype=torch.int64) * 2

    def ascii_reads(t, R):
        w = t[: (R * L + 31) // 32]
        out = torch.empty((R, L), dtype=torch.uint8, device=dev)
        step = 1 << 20                                        # reads per slice: the 8-byte codes of a slice stay below 2 GB
        for r0 in range(0, R, step):
            r1 = min(R, r0 + step)
            ws = w[(r0 * L) // 32: (r1 * L + 31) // 32 + 1]
            codes = ((ws[:, None] >> sh[None, :]) & 3).reshape(-1)
            o0 = r0 * L - ((r0 * L) // 32) * 32
            out[r0:r1] = lut[codes[o0: o0 + (r1 - r0) * L]].reshape(r1 - r0, L)
            del codes
        return out

    def write_files(R, D, kind, tag):
        """D distinct files of R reads each; kind: fasta | fastq_gz.  Returns their paths."""
        sub = dict(wl, n=D, reads=R)
        _, reads = gen_device_samples(lib, torch, sub, dev)
        paths = []
        # (threads, not processes: zlib releases the GIL, and forking / pickling out of a process that holds 40 GB of device state took
        # nine minutes for these 3 GB)
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(min(64, os.cpu_count() or 1)) if kind == "fastq_gz" else None
        for s in range(D):
            asc = ascii_reads(reads[s], R)
            reads[s] = None
            if kind == "fasta":
                rec = torch.empty((R, L + 4), dtype=torch.uint8, device=dev)          # ">r\n" + read + "\n"
                rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = asc; rec[:, 3 + L] = ord("\n")
                path = os.path.join(d, "%s%d.fasta" % (tag, s))
                rec.cpu().numpy().tofile(path)
            else:
                rec = torch.empty((R, 2 * L + 7), dtype=torch.uint8, device=dev)      # "@r\n" + read + "\n+\n" + qualities + "\n"
                rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2] = ord("\n"); rec[:, 3:3 + L] = asc
                rec[:, 3 + L] = ord("\n"); rec[:, 4 + L] = ord("+"); rec[:, 5 + L] = ord("\n"); rec[:, 6 + L: 6 + 2 * L] = ord("I"); rec[:, 6 + 2 * L] = ord("\n")
                raw = rec.cpu().numpy()
                rows = max(1, R // 64)
                parts = list(pool.map(_gz_member, [raw[r0: r0 + rows] for r0 in range(0, R, rows)]))      # concatenated gzip members
                path = os.path.join(d, "%s%d.fastq.gz" % (tag, s))
                with open(path, "wb") as f:
                    for m in parts:
                        f.write(m)
            paths.append(path)
            del asc, rec
        if pool is not None:
            pool.shutdown()
        del reads
        torch.cuda.empty_cache()
        return paths

    def driver(paths, tag, extra=()):
        inp = os.path.join(d, "in_%s.txt" % tag)
        open(inp, "w").write("".join("S%d: %s\n" % (s, paths[s % len(paths)]) for s in range(n)))
        size = sum(os.path.getsize(paths[s % len(paths)]) for s in range(n))
        cmd = [b.CLI_PATH, "-in", inp, "-out", os.path.join(d, "out_" + tag), "-out-tmp", os.path.join(d, "tmp_" + tag),
               "-kmer-size", str(k), "-abundance-min", str(wl["amin"]), "-max-reads", "-1", "-verbose", "0"]
        if wl["simple"]:
            cmd.append("-simple-dist")
        if wl.get("complex"):
            cmd.append("-complex-dist")
        # (untimed pause: the driver process of the run before left its device memory -- 100+ GB at C3's depth -- to the kernel driver, and a
        # process that starts while that is still being reclaimed waits for it in its own allocations: back-to-back runs measured 7.5 s
        # where runs ten seconds apart take 3.6 s)
        time.sleep(float(os.environ.get("SIMKA_BENCH_E2E_PAUSE", "8")))
        t = time.perf_counter()
        r = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = (time.perf_counter() - t) * 1e3
        if r.returncode != 0:
            raise RuntimeError(r.stdout[-400:])
        shutil.rmtree(os.path.join(d, "out_" + tag), ignore_errors=True)
        return dt, size

    try:
        R = min(wl["reads"], max_reads)
        per_file = R * (L + 4)
        free = shutil.disk_usage(d).free
        D = n if free > 2.5 * n * per_file else min(n, 10)
        mark("e2e: writing the tenth-depth FASTA files")
        paths = write_files(R, D, "fasta", "s")
        mark("e2e: files written")
        ts = [driver(paths, "tenth")[0], driver(paths, "tenth")[0]]
        t_host, size = driver(paths, "tenth", ["-host-parse"])
        occ = float(n) * R * (L - k + 1)
        out = {"ms": min(ts), "ms_first_run": ts[0], "ms_host_parse": t_host, "fasta_bytes": size, "fasta_GBps": size / (min(ts) * 1e-3) / 1e9,
               "kmer_occurrences_per_s": occ / (min(ts) * 1e-3),
               "sample": "%d samples x %d reads x %d bp as FASTA files (%d distinct files, %.1f GB listed), k=%d, `simka` driver process start to CSVs "
                         "written; text parsed on the GPU (ms_host_parse: the same run with -host-parse)" % (n, R, L, D, size / 1e9, k)}
        for p_ in paths:
            os.remove(p_)
        try:        # the tenth-depth reads as .fastq.gz, 10 distinct files
            mark("e2e: tenth-depth leg done; writing .fastq.gz")
            gz = write_files(R, min(n, 10), "fastq_gz", "q")
            mark("e2e: gz written")
            tg = [driver(gz, "gz")[0], driver(gz, "gz")[0]]
            gsize = sum(os.path.getsize(gz[s % len(gz)]) for s in range(n))
            out["fastq_gz"] = {"ms": min(tg), "gz_bytes": gsize, "text_bytes": float(n) * R * (2 * L + 7), "text_GBps": float(n) * R * (2 * L + 7) / (min(tg) * 1e-3) / 1e9,
                               "kmer_occurrences_per_s": occ / (min(tg) * 1e-3),
                               "sample": "%d samples x %d reads x %d bp as .fastq.gz (%d distinct files, %.2f GB of gzip listed): inflated, parsed and packed by the driver's worker threads" % (n, R, L, len(gz), gsize / 1e9)}
            for p_ in gz:
                os.remove(p_)
        except Exception as e:
            out["fastq_gz"] = {"ms": None, "sample": "failed: %r" % (e,)}
        try:        # the workload's own depth, 10 distinct files
            Rf, Df = wl["reads"], min(n, 10)
            if Rf > R and shutil.disk_usage(d).free > 1.3 * Df * Rf * (L + 4):
                mark("e2e: gz leg done; writing the full-depth files")
                full = write_files(Rf, Df, "fasta", "f")
                mark("e2e: full-depth files written")
                tf = [driver(full, "full")[0], driver(full, "full")[0]]
                fsize = sum(os.path.getsize(full[s % len(full)]) for s in range(n))
                out["full_depth"] = {"ms": min(tf), "ms_first_run": tf[0], "fasta_bytes": fsize, "fasta_GBps": fsize / (min(tf) * 1e-3) / 1e9,
                                     "kmer_occurrences_per_s": float(n) * Rf * (L - k + 1) / (min(tf) * 1e-3),
                                     "sample": "%d samples x %d reads x %d bp (the workload's own depth): %d distinct FASTA files, each listed by %d samples -- %.0f GB listed, "
                                               "read from the page cache; the largest depth the temp space of this box allows" % (n, Rf, L, Df, n // Df, fsize / 1e9)}
            else:
                out["full_depth"] = {"ms": None, "sample": "skipped: not enough temp space (or the workload is already at this depth)"}
        except Exception as e:
            out["full_depth"] = {"ms": None, "sample": "failed: %r" % (e,)}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


_T0 = time.perf_counter()


def mark(what):
    """coarse wall-clock marks on stderr: where a bench run spends its time"""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("[bench %7.1f s] %s\n" % (time.perf_counter() - _T0, what))
        sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SIMKA_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS),
                    help="default c3 = BASELINE.json configs[2], the configuration the metric and the targets are quoted on (fits one GPU)")
    ap.add_argument("--reads", type=int, default=0, help="override reads per sample")
    ap.add_argument("--kmer-size", type=int, default=0, help="override the workload's k (experiments: the bench line then names a workload of its own)")
    ap.add_argument("--samples", type=int, default=0, help="override number of samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=1, help="streams the samples alternate between in the timed region (library default: 2, "
                    "about 7 %% faster on C3).  1: kernels never overlap, so per-kernel durations add up to the step and agree with "
                    "a rocprofv3 trace of the same command; the two-stream step is reported as timing.step_ms_two_streams")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the untimed two-stream pass (timing.step_ms_two_streams)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the FASTA -> CSV leg (timing.e2e_from_fasta)")
    ap.add_argument("--no-from-host", action="store_true", help="skip the step that starts from pinned HOST memory (timing.step_ms_from_host)")
    ap.add_argument("--prof-steps", type=int, default=3, help="steps of the untimed pass that times every kernel (0 = as many as --steps)")
    ap.add_argument("--log2-partitions", type=int, default=0)
    ap.add_argument("--offsets", action="store_true", help="hand the reads over with an offsets array (variable-length layout, what the "
                    "simka driver uses) instead of fixed_len")
    ap.add_argument("--mgpu", default=os.environ.get("SIMKA_BENCH_MGPU", "both"), choices=["both", "sample", "partition"],
                    help="N > 1: 'partition' = north_star's split: every rank scans everything and keeps its partition shard, one all-reduce; "
                         "'sample' = samples counted on rank s %% N, solid spectra exchanged by partition range (all-to-all), one all-reduce; "
                         "'both' (default) times both and reports the faster one as `value`, the other under `decompositions`")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # launched as `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one process per GPU, RCCL)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import simka_amd
    from simka_amd import dist as sdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run without a launcher and let "
                         "bench.py spawn them)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the simka_amd path has no CPU fallback")
    backend = os.environ.get("SIMKA_BENCH_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; gloo only for single-GPU tests of the N>1 path
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank)" % (world, torch.cuda.device_count()))
    if world > torch.cuda.device_count():
        # several ranks on one device (the gloo tests of the N > 1 path): plain arena allocations -- chunks of a lazily mapped arena
        # being mapped while another context's kernels run on the same device faulted once in ~100 runs of `simka -nb-gpus -gpu-shared`
        os.environ["SIMKA_ARENA_MALLOC"] = "1"
    local = local % torch.cuda.device_count()        # tests run two ranks on one GPU (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl"
            # the data-path collectives run inside the C ABI on RCCL (simka_comm_*, simka_stats_allreduce); torch.distributed
            # bootstraps the communicator (unique id broadcast) and provides the timing barrier
            # (if the C ABI cannot get its communicator -- no librccl.so to dlopen, an init error -- every rank falls back to the same
            # collectives through torch.distributed, which is RCCL as well: the line says which under config.collectives)
            try:
                comm = sdist.create_comm(rank, world, local)
            except Exception as e:
                sys.stderr.write("bench.py rank %d: simka_comm_* unavailable (%r): collectives through torch.distributed\n" % (rank, e))
                comm = None
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close(); comm = None
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    modes = ["single"] if world == 1 else (["partition", "sample"] if args.mgpu == "both" else [args.mgpu])

    wl = dict(WORKLOADS[args.workload])
    if args.reads:
        wl["reads"] = args.reads
    if args.kmer_size:
        wl["k"] = args.kmer_size
        wl["desc"] = wl.get("desc", "") + " [k overridden: %d]" % args.kmer_size
    if args.samples:
        wl["n"] = args.samples
    n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
    lib = simka_amd.load_library()
    need_all = any(m != "sample" for m in modes)
    mark("generating the samples on the device")
    pool, reads = gen_device_samples(lib, torch, wl, dev, which=None if need_all else set(sdist.samples_of(rank, world, n)))
    mark("samples ready")
    nb_bases = R * L
    kocc_per_sample = R * (L - k + 1)
    d_offsets = torch.arange(0, (R + 1) * L, L, dtype=torch.int64, device=dev) if args.offsets else None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_mode(mode):
        """one decomposition of the job: (a) an untimed pass on a ONE-lane context with every kernel timed (per-kernel table:
        no overlap between kernels), (b) warmup + the timed region on the default context (samples alternate between two
        streams), HIP events around the dominant kernel only"""
        by_sample = mode == "sample"

        free_before = [None]

        def close_ctx(c):
            """close a context and wait until its device memory is back (a context sizes its arena from what is free: under a profiler
            the release of the one before can lag, and the next context would come up with half an arena)"""
            c.close()
            if free_before[0] is None:
                return
            t_w = time.perf_counter()
            while torch.cuda.mem_get_info(local)[0] + (2 << 30) < free_before[0] and time.perf_counter() - t_w < 5.0:
                torch.cuda.synchronize()
                time.sleep(0.05)
            if time.perf_counter() - t_w > 0.2:
                mark("%s: waited %.1f s for the device memory of a closed context (%.1f of %.1f GB free)" % (mode, time.perf_counter() - t_w, torch.cuda.mem_get_info(local)[0] / 1e9, free_before[0] / 1e9))

        def make_ctx():
            if free_before[0] is None:
                free_before[0] = torch.cuda.mem_get_info(local)[0]
            return simka_amd.SimkaContext(n, kmer_size=k, abundance_min=wl["amin"], simple_dist=wl["simple"],
                                          complex_dist=wl.get("complex", False), device=local,
                                          shard_index=0 if mode != "partition" else rank, shard_count=1 if mode != "partition" else world,
                                          max_kmers_per_sample=kocc_per_sample, log2_partitions=args.log2_partitions)

        trace = os.environ.get("SIMKA_BENCH_TRACE")

        def make_step(ctx):
            def count(s):
                if args.offsets:
                    ctx.count_sample(s, reads[s].data_ptr(), nb_bases, R, offsets=d_offsets.data_ptr(), on_device=True)
                else:
                    ctx.count_sample(s, reads[s].data_ptr(), nb_bases, R, fixed_len=L, on_device=True)

            def step():
                if by_sample:
                    # rank r counts the samples s % N == r over the whole key space, the solid spectra move to the rank that merges
                    # their partition range (all-to-all), the pair accumulators are all-reduced (simka_amd/dist.py)
                    sdist.count_exchange_merge(ctx, count, n, dev, comm=comm)
                else:
                    tr = [time.perf_counter()] if trace else None      # SIMKA_BENCH_TRACE: where the host spends a step
                    ctx.reset()
                    if tr: tr.append(time.perf_counter())
                    for s in range(n):
                        count(s)
                    if tr: tr.append(time.perf_counter())
                    if wl.get("complex") and world > 1:        # -complex-dist terms need the GLOBAL per-sample totals inside the merge
                        sdist.allreduce_totals_device(ctx, comm=comm)
                    ctx.merge()
                    if tr:
                        tr.append(time.perf_counter())
                        if trace == "threads":
                            import threading, traceback
                            for th_id, fr in sys._current_frames().items():
                                if th_id != threading.get_ident():
                                    sys.stderr.write("other thread %s:\n%s\n" % ([t.name for t in threading.enumerate() if t.ident == th_id], "".join(traceback.format_stack(fr)[-4:])))
                        if trace == "sleep": time.sleep(0.03)
                        if trace == "sync": torch.cuda.synchronize()
                        tr.append(time.perf_counter())
                    # ONE RCCL all-reduce of the flat u64 accumulators (no-op at N=1)
                    sdist.allreduce_stats_device(ctx, totals_already_reduced=bool(wl.get("complex")) and world > 1, comm=comm)
                st = ctx.stats()
                if trace and not by_sample: tr.append(time.perf_counter())
                mats = st.matrices()
                if trace and not by_sample:
                    tr.append(time.perf_counter())
                    sys.stderr.write("step trace (ms): reset %.2f  count calls %.2f  merge %.2f  (%s) %.2f  stats %.2f  matrices %.2f\n" % ((tuple((b_ - a_) * 1e3 for a_, b_ in zip(tr, tr[1:])))[:3] + (trace,) + (tuple((b_ - a_) * 1e3 for a_, b_ in zip(tr, tr[1:])))[3:]))
                return st, mats
            return step

        two_ms = None          # (the library's default, two streams: measured after the line is assembled -- finish())
        # ---- (a) per-kernel times, one lane
        prof_steps = min(args.steps, args.prof_steps) if args.prof_steps else args.steps
        os.environ["SIMKA_LANES"] = "1"
        ctx = make_ctx()
        step = make_step(ctx)
        step()
        ctx.profile_reset()
        ctx.profile_enable(True)
        fence()
        for _ in range(prof_steps):
            step()
        fence()
        ctx.profile_enable(False)
        prof_all = ctx.profile()
        dom = max(prof_all, key=lambda kk: prof_all[kk][1])
        mark("%s: per-kernel pass done" % mode)
        # ---- (b) the timed region: on the SAME context when it runs one lane as well (the default) -- a context sizes its arena from
        # the free device memory, and under rocprofv3 the mapped arena of a closed context only comes back with its address range
        # (scripts/vmm_probe.py), so a second context in the process would come up with half an arena
        if max(1, args.lanes) != 1:
            close_ctx(ctx)
            os.environ["SIMKA_LANES"] = str(max(1, args.lanes))
            ctx = make_ctx()
            step = make_step(ctx)
        for _ in range(args.warmup):
            step()
        ctx.profile_reset()
        ctx.profile_enable(True, only=[dom])
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st, mats = step()
        fence()
        dt = time.perf_counter() - t0
        ctx.profile_enable(False)
        mark("%s: timed region done (%.1f ms per step)" % (mode, dt / args.steps * 1e3))
        if world > 1:
            tdt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            dt = float(tdt.item())
        return dict(mode=mode, ctx=ctx, dt=dt, st=st, mats=mats, two_ms=two_ms, prof_all=prof_all, prof_steps=prof_steps, prof_dom=ctx.profile(), dom=dom,
                    geo=ctx.geometry(), by_sample=by_sample, make_ctx=make_ctx, make_step=make_step, close_ctx=close_ctx)

    def checksum(mats):
        # sha1 of every distance matrix (float32 bytes, by name): equal across --gpus N and decompositions for the same workload
        return __import__("hashlib").sha1(b"".join(np.ascontiguousarray(mats[m]).tobytes() for m in sorted(mats))).hexdigest()[:16]

    def finish(results, failures, hard, note):
        """the bench line from the decompositions that ran (hard: called after a failure or from the watchdog -- host data only, no device
        call that could queue behind a hung collective, no further measurements)"""
        best = results[0]
        ctx, dt, st, mats, prof_all, prof_dom, dom, geo, by_sample = (best[x] for x in ("ctx", "dt", "st", "mats", "prof_all", "prof_dom", "dom", "geo", "by_sample"))
        prof = prof_all
        prof_steps = best["prof_steps"]

        # ---- timing boundaries of SURVEY 8(d), measured outside the timed region (per step, this rank)
        def timed(f, reps=3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3
        t_finalise = t_h2d = t_allreduce = None
        if not hard:
            t_finalise = timed(lambda: ctx.stats().matrices())                       # download + 15..21 matrices on the host
            one = next(r for r in reads if r is not None)
            pinned = torch.empty(one.numel(), dtype=torch.int64).pin_memory()
            scratch = torch.empty_like(one)
            nmine = sum(1 for r in reads if r is not None) if by_sample else n
            t_h2d = timed(lambda: scratch.copy_(pinned, non_blocking=True)) * nmine      # packed reads host -> HBM (pinned), not part of `value`
            if world > 1:
                t_allreduce = timed(lambda: sdist.allreduce_stats_device(ctx, comm=comm))      # (sums the buffer into itself: timing only, after the results were taken)
            del pinned, scratch

        ps = st.per_sample()
        K_dist = float(ps["D_all"].sum())     # distinct canonical k-mers before the filter, summed over samples (whole job)
        K_occ = float(ps["K_occ"].sum())
        K_solid = float(ps["D"].sum())
        ms_per_step = dt / args.steps * 1e3
        value = K_dist / (dt / args.steps)

        # ---- roofline (DESIGN.md section 4 "Algorithmic bytes").  SURVEY 8(d): B_alg = R*L/4 + 16*K_occ + 12*K_dist + 12*K_solid, where
        # 16*K_occ = "write + read each 8-byte k-mer once for partitioning".  The super-k-mer pipeline partitions 16-byte records of
        # ~8.5 k-mers instead (real traffic ~2 B per k-mer and level), so the kernels' MEASURED traffic is far below these
        # design-independent bytes; the attribution follows the roles: the scan writes every k-mer once, the count kernel reads every
        # k-mer once and writes the counted records, the merge reads the solid records once; k_skm_chunksort (the partitioning level) is an extra level (0).
        share = 1.0 / world                    # each rank owns 1/world of the key space (or counts 1/world of the samples)
        scan_reads = n * nb_bases / 4.0 * (share if by_sample else 1.0)       # partition shards: every rank reads every base
        kb = 16.0 if k > 31 else 8.0           # bytes of a k-mer (SURVEY 8d prices 8-byte k-mers; two words from k = 32 on)
        alg_bytes_per_step = {
            "k_skm_scan": scan_reads + kb * K_occ * share,
            "k_skm_count_fast": kb * K_occ * share + (kb + 4.0) * K_dist * share,
            "k_skm_count_wide_fast": kb * K_occ * share + (kb + 4.0) * K_dist * share,
            "k_group": (kb + 4.0) * K_solid * share,
        }
        kern_ms = {kname: ms for kname, (cnt, ms) in prof.items()}
        total_kernel_ms = sum(kern_ms.values())
        dom_launches, dom_ms = prof_dom[dom]
        dom_bytes_per_launch = alg_bytes_per_step.get(dom, 0.0) * args.steps / max(dom_launches, 1)
        dom_avg_ms = dom_ms / max(dom_launches, 1)
        achieved = dom_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        b_alg = scan_reads + (2.0 * kb * K_occ + (kb + 4.0) * K_dist + (kb + 4.0) * K_solid) * share
        # the whole path: B_alg over the step's wall time (kernels of neighbouring samples overlap on two streams, so the sum of the
        # one-lane kernel times is an upper bound of the device time; it is reported as timing.device_kernels_ms)
        path_gbs = b_alg / (dt / args.steps) / 1e9
        # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (scripts/pmc_traffic.sh:
        # FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs, FETCH_SIZE x2 on gfx950); only when it is the same workload
        traffic = None
        traffic_src = None
        tk = {}
        # rocprofv3 reports template instances; the library's profiler reports one name per kernel family
        def family(name):
            if name.startswith("k_skm_scan<"):
                return "k_skm_scan<hist>" if name.rstrip(">").endswith("true") else "k_skm_scan"
            return name.split("<")[0]

        def measured_traffic(kname):
            """launch-weighted mean HBM bytes per launch over the template instances of the family"""
            tot, n = 0.0, 0
            for cand, v in tk.items():
                if family(cand) == kname:
                    tot += v["traffic_bytes_per_launch"] * v["launches"]; n += v["launches"]
            return tot / n if n else None
        try:
            tf = next((f for f in (os.path.join(ROOT, "profiles", "r%02d_%s_hbm_traffic.json" % (rd, args.workload)) for rd in (6, 5, 4, 3, 2)) if os.path.exists(f)), "")
            if world == 1 and not args.reads and not args.samples and not args.kmer_size and tf:
                tk = json.load(open(tf))["kernels"]
                traffic = measured_traffic(dom)
                traffic_src = os.path.relpath(tf, ROOT)
        except Exception:
            traffic = None
        per_kernel = {}
        for kname, (cnt, ms) in prof.items():
            if cnt == 0:
                continue
            ab = alg_bytes_per_step.get(kname, 0.0)
            mt = measured_traffic(kname)
            per_kernel[kname] = {"launches_per_step": cnt / prof_steps, "ms_per_step": ms / prof_steps,
                                 "alg_bytes_per_step": ab, "alg_GBps": (ab * prof_steps / (ms * 1e-3) / 1e9) if ms > 0 else 0.0,
                                 "hbm_traffic_bytes_per_launch": mt,
                                 # what the kernel really moves over HBM (PMC counters) against the 8 TB/s peak -- NOT the attributed bytes above
                                 "hbm_frac_measured": (mt * cnt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (mt and ms > 0) else None,
                                 "bound": KERNEL_BOUND.get(kname)}
        # what bounds the dominant kernel when it is not its bytes: the SQ pass committed under profiles/ (scripts/sq_summary.py ... out.json), per
        # k-mer occurrence of THIS run.  valu_busy next to hbm_frac_measured says at a glance which pipe is full.
        issue = None
        try:
            sf = next((f for f in (os.path.join(ROOT, "profiles", "r%02d_%s_sq_counters.json" % (rd, args.workload)) for rd in (6, 5)) if os.path.exists(f)), "")
            if world == 1 and not args.reads and not args.samples and not args.kmer_size and sf:
                sqk = json.load(open(sf))["kernels"]

                def issue_of(kname, launches_per_step):
                    sq = sqk.get(kname.split("<")[0])
                    if not sq or not sq["launches"] or not launches_per_step:
                        return None
                    kocc_per_launch = K_occ / launches_per_step
                    per_launch = lambda key: sq[key] / sq["launches"]
                    out_ = {"valu_busy": sq["valu_busy"], "active": sq["active"], "wait": sq["wait"], "issue_stall": sq["issue_stall"], "lds_busy": sq["lds_busy"],
                            # lane-instructions per k-mer occurrence of the step's share of a launch (wave instructions x 64 lanes / k-mer occurrences)
                            "inst_per_kmer": {c: per_launch("wave_inst_" + c) * 64.0 / kocc_per_launch for c in ("valu", "salu", "lds", "vmem")},
                            # cycles of a wave's lifetime per instruction it executes (waits included), and the part of them in which the wave had an
                            # instruction in flight: at waves_per_simd resident waves a SIMD issues once per active_cycles_per_inst / waves cycles
                            "wave_cycles_per_inst": sq["cycles_per_inst"], "active_cycles_per_inst": sq["cycles_per_inst"] * sq["active"] if sq["cycles_per_inst"] else None,
                            "waves_per_simd": WAVES_PER_SIMD.get(kname.split("<")[0]), "source": os.path.relpath(sf, ROOT)}
                    out_["inst_per_kmer"]["all"] = sum(out_["inst_per_kmer"].values())
                    return out_
                issue = issue_of(dom, max(dom_launches / max(args.steps, 1), 1))
                for kname, pk_ in per_kernel.items():      # the same reading for every kernel of the step (what bounds each: VALU busy next to its HBM fraction)
                    pk_["issue"] = issue_of(kname, pk_["launches_per_step"])
        except Exception:
            issue = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "issue": issue, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    # `frac` prices the kernel in SURVEY 8(d)'s design-independent bytes (8-byte k-mers); its real HBM traffic is `traffic`:
                    "hbm_frac_measured": (traffic / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and dom_avg_ms > 0) else None,
                    "kernel_bound": KERNEL_BOUND.get(dom),
                    "avg_launch_ms": dom_avg_ms, "launches": dom_launches, "alg_bytes_per_launch": dom_bytes_per_launch,
                    "path_achieved": path_gbs, "path_frac": path_gbs / HBM_PEAK_GBS, "path_alg_bytes_per_step": b_alg,
                    "kernel_ms_per_step": {kk: v / prof_steps for kk, v in kern_ms.items()}, "kernels": per_kernel,
                    "events": "timed region (%d stream%s): HIP events around the %s launches only (avg_launch_ms, achieved); path_*: B_alg over the timed "
                              "step; kernel_ms_per_step, kernels and timing.device_kernels_ms: an untimed pass of %d steps on a one-stream context with "
                              "every kernel timed" % (max(1, args.lanes), "" if args.lanes <= 1 else "s: a launch may share the GPU with the other stream's kernels", dom, prof_steps)}

        pair_updates = float(st.pairs()["a"].sum())            # sum over k-mers of s(s-1)/2 = sum over pairs of the shared distinct k-mers
        if "k_pairs" in per_kernel and per_kernel["k_pairs"]["ms_per_step"] > 0:
            per_kernel["k_pairs"]["pair_updates_per_step"] = pair_updates
            per_kernel["k_pairs"]["pair_updates_per_s"] = pair_updates / world / (per_kernel["k_pairs"]["ms_per_step"] * 1e-3)
        timing = {"step_ms_two_streams": best.get("two_ms"), "device_kernels_ms": total_kernel_ms / prof_steps, "finalise_ms": t_finalise, "h2d_packed_reads_ms": t_h2d,
                  "allreduce_ms": t_allreduce, "step_ms": ms_per_step,
                  "note": "per step on rank 0; h2d = the packed reads of this rank's samples from pinned host memory (inputs are resident in HBM in the timed "
                          "region); e2e_from_fasta: the `simka` driver on FASTA files of a bounded sample of the workload (files on disk -> CSVs)"}
        out = {
            "metric": "distinct k-mers/s end-to-end (count + merge + N x N matrices)", "value": value, "unit": "distinct k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "samples": n, "reads_per_sample": R, "read_len": L,
                       "kmer_size": k, "abundance_min": wl["amin"], "simple_dist": wl["simple"],
                       "parallelism": ("1 GPU" if world == 1 else "samples x%d -> all-to-all of solid spectra by partition range -> 1 all-reduce" % world
                                       if by_sample else "partition shards x%d + 1 all-reduce" % world),
                       "kmer_occurrences": K_occ, "distinct_kmers": K_dist, "solid_kmers": K_solid,
                       "kmer_occurrences_per_s": K_occ / (dt / args.steps), "geometry": geo,
                       # sha1 of every distance matrix (float32 bytes, by name): equal across --gpus N for the same workload
                       "matrix_checksum": checksum(mats)},
            "roofline": roofline,
            "timing": timing,
        }
        if world > 1:
            # every decomposition that ran: the reported one first (`value` is the faster); checksums must agree
            out["decompositions"] = {r["mode"]: {"ms_per_step": r["dt"] / args.steps * 1e3, "value": K_dist / (r["dt"] / args.steps),
                                                 "matrix_checksum": checksum(r["mats"])} for r in results}
            if failures or note:
                out["decomposition_failures"] = dict(failures, **({"note": note} if note else {}))
            out["config"]["collectives"] = ("RCCL through the C ABI (simka_stats_allreduce / simka_exchange_*)" if comm is not None
                                            else "torch.distributed %s" % backend)
            if comm is not None:       # which librccl the C ABI got (one RCCL per process: torch's already loaded copy is reused)
                try:
                    out["config"]["rccl_library"] = simka_amd.api.Comm.library()
                except Exception as e:
                    out["config"]["rccl_library"] = "unknown (%r)" % (e,)
        if hard:
            if rank == 0:
                print(json.dumps(out), flush=True)
            return
        mark("line assembled")
        if world == 1 and args.lanes == 1 and not args.no_two_streams and not by_sample:
            # the library's default (two streams), untimed report: timing.step_ms_two_streams -- after the judged numbers, on a context of its own
            try:
                best["close_ctx"](ctx); best["ctx"] = None; ctx = None
                os.environ["SIMKA_LANES"] = "2"
                c2 = best["make_ctx"]()
                try:
                    step2 = best["make_step"](c2)
                    step2()
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    step2()
                    torch.cuda.synchronize()
                    out["timing"]["step_ms_two_streams"] = (time.perf_counter() - t2) * 1e3
                finally:
                    best["close_ctx"](c2)
            except Exception as e:
                out["timing"]["step_ms_two_streams"] = None
                out["timing"]["two_streams_note"] = "failed: %r" % (e,)
            os.environ["SIMKA_LANES"] = "1"
            mark("two-stream pass done")
        if world == 1 and not args.no_from_host and not args.offsets:
            # the same step with the packed reads in PINNED HOST memory: simka_count_sample(on_device = 0) copies sample i + 1 through the
            # copy stream into its lane's staging buffer while the kernels of sample i run on the other lane (library default: two lanes).
            # Never `value` (inputs resident in HBM there); done = within 5 % of step_ms
            try:
                if ctx is not None:
                    best["close_ctx"](ctx); best["ctx"] = None; ctx = None
                os.environ["SIMKA_LANES"] = "2"
                nw_ = (nb_bases + 31) // 32 + 2
                t_pin = time.perf_counter()
                host = []
                for s_ in range(n):
                    hbuf = torch.empty(nw_, dtype=torch.int64).pin_memory()
                    hbuf.copy_(reads[s_][:nw_])
                    host.append(hbuf)
                torch.cuda.synchronize()
                t_pin = time.perf_counter() - t_pin
                hctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=wl["amin"], simple_dist=wl["simple"], complex_dist=wl.get("complex", False),
                                              device=local, max_kmers_per_sample=kocc_per_sample, log2_partitions=args.log2_partitions)

                def host_step():
                    hctx.reset()
                    for s_ in range(n):
                        hctx.count_sample(s_, host[s_].data_ptr(), nb_bases, R, fixed_len=L, on_device=False, host_pointer=True)
                    hctx.merge()
                    return hctx.stats().matrices()
                host_step()
                torch.cuda.synchronize(); t0_ = time.perf_counter()
                reps_ = 2
                for _ in range(reps_):
                    hm_ = host_step()
                torch.cuda.synchronize()
                t_host = (time.perf_counter() - t0_) / reps_ * 1e3
                out["timing"]["step_ms_from_host"] = t_host
                # like for like: the host-resident step runs on the library's two lanes, so it is compared with the two-lane device-resident step
                # (timing.step_ms_two_streams) when that was measured, with the one-lane timed step otherwise -- the note says which
                two_ = out["timing"].get("step_ms_two_streams")
                out["timing"]["step_from_host_over_step"] = t_host / (two_ if two_ else ms_per_step)
                out["timing"]["step_from_host_over_step_denominator"] = "step_ms_two_streams" if two_ else "ms_per_step (one lane)"
                out["timing"]["step_from_host_note"] = ("packed reads in pinned host memory (%.1f GB, pinned + filled in %.1f s, untimed), H2D on the copy stream under the "
                                                        "other lane's kernels; matrices %s the device-resident step's" % (n * nw_ * 8 / 1e9, t_pin, "equal" if checksum(hm_) == out["config"]["matrix_checksum"] else "DIFFER FROM"))
                hctx.close()
                del host
                ctx = None
            except Exception as e:
                out["timing"]["step_ms_from_host"] = None
                out["timing"]["step_from_host_note"] = "failed: %r" % (e,)
                ctx = None
        mark("from-host step done")
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, lib, torch, dev)
            except Exception as e:       # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "distinct k-mers/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if ctx is not None:
            ctx.close()
        mark("cpu baseline done")
        if rank == 0 and world == 1 and not args.no_e2e and not args.no_cpu_baseline:
            try:
                out["timing"]["e2e_from_fasta"] = e2e_from_fasta(wl, lib, torch, dev)
                mark("e2e legs done")
                out["timing"]["e2e_from_fasta_ms"] = out["timing"]["e2e_from_fasta"]["ms"]
            except Exception as e:
                out["timing"]["e2e_from_fasta"] = {"ms": None, "sample": "failed: %r" % (e,)}
        if rank == 0:
            print(json.dumps(out))

    # Every decomposition runs under its own try / except, and -- once one of them has a result -- under a watchdog: an RCCL error or
    # a hang in the second one (no multi-GPU box has run these collectives yet) must not lose the line of the first.  On such a
    # failure rank 0 prints the line of what did run (`decomposition_failures` says what did not) and every rank leaves with
    # os._exit: the process group may be wedged, so no further collective, no destructor.
    import threading
    results, failures = [], {}
    state = {"done": False}

    def bail(reason):
        if state["done"]:
            return
        state["done"] = True
        try:
            if rank == 0 and results:
                finish(results, failures, hard=True, note=reason)
        finally:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0 if results else 3)

    for mode in modes:
        wd = None
        t_mode = time.perf_counter()
        if world > 1 and results:
            limit = float(os.environ.get("SIMKA_BENCH_MODE_TIMEOUT", "0")) or max(240.0, 8.0 * results[0]["wall"])
            wd = threading.Timer(limit, bail, args=("decomposition '%s' did not finish within %.0f s" % (mode, limit),))
            wd.daemon = True
            wd.start()
        try:
            if os.environ.get("SIMKA_BENCH_FAIL_MODE") == mode:          # tests: this decomposition fails / hangs
                if os.environ.get("SIMKA_BENCH_FAIL_HOW") == "hang":
                    time.sleep(1e6)
                raise RuntimeError("SIMKA_BENCH_FAIL_MODE=%s" % mode)
            r = run_mode(mode)
            r["wall"] = time.perf_counter() - t_mode
        except Exception as e:
            if wd is not None:
                wd.cancel()
            failures[mode] = repr(e)
            sys.stderr.write("bench.py rank %d: decomposition '%s' failed: %r\n" % (rank, mode, e))
            if world > 1 and results:
                bail("decomposition '%s' failed: %r" % (mode, e))
            continue
        if wd is not None:
            wd.cancel()
        if results:                       # keep one context alive: the faster decomposition is the reported one
            keep, drop = (r, results[0]) if r["dt"] < results[0]["dt"] else (results[0], r)
            drop["ctx"].close(); drop["ctx"] = None
            results = [keep, drop]
        else:
            results = [r]
    if not results:
        raise SystemExit("bench.py: no decomposition ran: %r" % (failures,))
    state["done"] = True
    finish(results, failures, hard=False, note=None)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
