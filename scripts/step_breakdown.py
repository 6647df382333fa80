"""Wall-clock breakdown of one bench step (C2 by default): reset | count loop (enqueue) | count drain | merge | stats+matrices."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, simka_amd, bench
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
lib = simka_amd.load_library(); dev = torch.device("cuda:0")
pool, reads = bench.gen_device_samples(lib, torch, wl, dev)
n, R, L, k = wl["n"], wl["reads"], wl["L"], wl["k"]
ctx = simka_amd.SimkaContext(n, kmer_size=k, abundance_min=wl["amin"], simple_dist=wl["simple"], max_kmers_per_sample=R * (L - k + 1))
def T(): return time.perf_counter()
acc = [0.0] * 5
for it in range(6):
    t0 = T(); ctx.reset(); ctx.sync(); t1 = T()
    for s in range(n): ctx.count_sample(s, reads[s].data_ptr(), R * L, R, fixed_len=L, on_device=True)
    t2 = T(); ctx.sync(); t3 = T()
    ctx.merge(); ctx.sync(); t4 = T()
    st = ctx.stats(); m = st.matrices(); t5 = T()
    if it:
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += v * 1e3 / 5
print("reset %.2f | count enqueue %.2f | count drain %.2f | merge %.2f | stats+matrices %.2f  ms (total %.2f)" % (*acc, sum(acc)))
