"""An empty sample next to non-empty ones, all distance families: the CSV files (NaN cells included) of the GPU path and of the
oracle must be byte-identical."""
import sys, os, gzip, glob, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, simka_amd, oracle_lib
reads = [[b"ACGTACGTTGCAAGCTAGCTAGCATCGATCGATCGGGCTAGCTAGCTA" * 2] * 3, [], [b"TTGCAAGCTAGCTAGCATCGATCGATACGTACGTCCCGGGAAATTT" * 2] * 2]
k = 11
ctx = simka_amd.SimkaContext(3, kmer_size=k, abundance_min=1, simple_dist=True, complex_dist=True)
orc = oracle_lib.Oracle()
for s, rs in enumerate(reads):
    packed, off, nb, nfrag = simka_amd.pack_reads(rs)
    ctx.count_sample(s, packed, nb, len(off) - 1, offsets=off, nb_input_reads=len(rs))
    a = np.frombuffer(b"".join(rs), dtype=np.uint8) if rs else np.zeros(0, dtype=np.uint8)
    orc.add_sample_ascii("S%d" % s, a, np.concatenate([[0], np.cumsum([len(r) for r in rs])]).astype(np.uint64))
ctx.merge(); st = ctx.stats()
orc.run(k, 1, simple=True, complex_=True)
d1, d2 = tempfile.mkdtemp(), tempfile.mkdtemp()
st.write_matrices(d1, ["S0", "S1", "S2"], gz=True)
orc.write_matrices(d2, gz=False)
bad = 0
for f in sorted(glob.glob(d1 + "/*.csv.gz")):
    name = os.path.basename(f)[:-3]
    a = gzip.open(f, "rb").read(); b = open(os.path.join(d2, name), "rb").read() if os.path.exists(os.path.join(d2, name)) else None
    if b is not None and a != b:
        bad += 1; print("DIFF", name); print(a.decode()); print(b.decode())
print("compared, differing files:", bad)
