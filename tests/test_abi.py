"""CPU-side checks of the product library: it loads, exports every symbol include/simka_hip.h declares,
fails loudly without a GPU, and its HOST pieces (finalisation, CSV, packer) agree with the oracle/goldens."""
import ctypes as C
import glob
import gzip
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [(21, 0), (21, 2), (31, 0), (31, 2)]


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "simka_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(simka_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(simka_lib):
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(simka_lib, n), "libsimka_hip.so does not export %s" % n
    assert simka_lib.simka_abi_version() == 8


def test_struct_sizes_match_header():
    from simka_amd import api
    assert C.sizeof(api.Config) == 80
    assert C.sizeof(api.Reads) == 48
    assert C.sizeof(api.SampleTotals) == 48


def test_create_fails_loudly_without_gpu(simka_lib):
    import torch
    import simka_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(simka_amd.SimkaError) as e:
        simka_amd.SimkaContext(2, kmer_size=21)
    assert "no CPU fallback" in str(e.value)


def test_create_validates_arguments(simka_lib):
    from simka_amd import api
    h = C.c_void_p()
    cfg = api.Config()
    cfg.struct_size = 4          # wrong ABI size
    assert simka_lib.simka_create(C.byref(cfg), C.byref(h)) == 1
    cfg.struct_size = C.sizeof(api.Config)
    cfg.nb_samples, cfg.kmer_size, cfg.shard_count = 2, 128, 1     # k >= 128 is beyond the reference's largest span (64..127: four-word scan, fingerprints)
    assert simka_lib.simka_create(C.byref(cfg), C.byref(h)) == 1
    cfg.kmer_size, cfg.shard_index = 21, 3                          # shard_index >= shard_count
    assert simka_lib.simka_create(C.byref(cfg), C.byref(h)) == 1
    assert b"shard" in simka_lib.simka_last_error(None)


def test_matrix_catalogue_matches_reference_order(simka_lib, oracle_mod):
    import simka_amd
    o = oracle_mod.Oracle()
    assert [simka_lib.simka_matrix_name(w).decode() for w in range(simka_lib.simka_nb_matrices())] == o.matrix_names()
    assert len(simka_amd.matrix_names(0)) == 15 and len(simka_amd.matrix_names(1)) == 18 and len(simka_amd.matrix_names(3)) == 21


@pytest.mark.parametrize("k,amin", CONFIGS)
def test_host_finalisation_and_csv_vs_goldens(simka_lib, oracle_mod, golden_dir, tmp_path, k, amin):
    """Feed the ORACLE's integer accumulators through the PRODUCT's host finalisation + CSV writer:
    all 20 goldens must come out byte-exact (SimkaDistance.cpp:603-699,920-1226)."""
    import simka_amd
    o = oracle_mod.Oracle()
    o.load_input(os.path.join(golden_dir, "example", "simka_input.txt"))
    o.run(k, amin, simple=True, complex_=True)
    st = simka_amd.Stats(o.n, simka_amd.DIST_SIMPLE | simka_amd.DIST_COMPLEX, o.flat_stats(simple=True, complex_=True))
    out = str(tmp_path / "csv")
    st.write_matrices(out, o.ids(), gz=True)
    truth = os.path.join(golden_dir, "truth", "results_k%d_t%d" % (k, amin))
    n = 0
    for gzf in sorted(glob.glob(os.path.join(out, "*.csv.gz"))):
        ref = os.path.join(truth, os.path.basename(gzf)[:-3])
        if os.path.exists(ref):
            with gzip.open(gzf, "rb") as f, open(ref, "rb") as g:
                assert f.read() == g.read(), os.path.basename(gzf)
            n += 1
    assert n == 20
    iu = np.triu_indices(o.n, 1)
    assert np.array_equal(st.pairs()["canb"], o.acc("canb")[iu])          # D_i + D_j - 2a identity
    np.testing.assert_allclose(st.pairs()["kl"], o.kl()[iu], rtol=1e-12)
    for w, name in enumerate(o.matrix_names()):     # unpinned mat_abundance_jaccard too: equal to the oracle's
        if name in st.matrices():
            assert np.array_equal(st.matrices()[name], o.matrix(w)), name


def test_pack_read_splits_at_non_acgt(simka_lib):
    import simka_amd
    from simka_amd import synth
    packed, offsets, nb, nin = simka_amd.pack_reads([b"ACGTNNacgtRTTGA", b"", b"GGGG"])
    assert nin == 3 and nb == 4 + 4 + 4 + 4
    assert list(offsets) == [0, 4, 8, 12, 16]
    assert synth.unpack_ascii(packed, nb).tobytes() == b"ACGTACGTTTGAGGGG"


def test_stats_layout_roundtrip(simka_lib):
    import simka_amd
    n = 7
    P = n * (n - 1) // 2
    size = simka_lib.simka_stats_nb_u64(n, simka_amd.DIST_SIMPLE)
    assert size == 8 + 6 * P + 5 * n
    assert simka_lib.simka_stats_nb_u64(n, 3) == 8 + 8 * P + 5 * n + 2 * P
    flat = np.arange(size, dtype=np.uint64)
    st = simka_amd.Stats(n, simka_amd.DIST_SIMPLE, flat)
    assert st.layout == {"nacc": 6, "acc0": 8, "tot0": 8 + 6 * P, "derived": 8 + 6 * P + 5 * n, "nb_pairs": P, "head": 8 + 6 * P, "total": size}
    assert list(st.per_sample()["D"]) == list(range(8 + 6 * P, 8 + 6 * P + n))
    assert st.pairs()["S_ij"][0] == 8
    assert st.dense("S")[0, 1] == 8 and st.dense("S")[1, 0] == 8 + P


def test_cli_error_conventions(simka_lib):
    """Reference conventions (src/SimkaPotara.hpp:376-387, src/core/SimkaAlgorithm.cpp:207-210): validation failures exit(1)
    with a message; without a GPU the driver fails loudly (EXCEPTION: ..., EXIT_FAILURE) instead of falling back."""
    import subprocess
    import torch
    from simka_amd import build as b
    b.build()
    inp = os.path.join(ROOT, "tests", "golden", "example", "simka_input.txt")
    r = subprocess.run([b.CLI_PATH, "-in", "/nonexistent/input.txt", "-out-tmp", "/tmp"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "ERROR: Input filename does not exist" in r.stderr
    r = subprocess.run([b.CLI_PATH, "-in", inp, "-out-tmp", "/tmp/simka_cli_t", "-max-memory", "100"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Please run Simka with higher memory usage than 500 MB" in r.stdout
    r = subprocess.run([b.CLI_PATH, "-out-tmp", "/tmp"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "-in" in r.stdout
    r = subprocess.run([b.CLI_PATH, "-in", inp, "-out-tmp", "/tmp", "-bogus"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Unknown parameter" in r.stdout
    if not torch.cuda.is_available():
        r = subprocess.run([b.CLI_PATH, "-in", inp, "-out", "/tmp/simka_cli_o", "-out-tmp", "/tmp/simka_cli_t", "-verbose", "0"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "EXCEPTION" in r.stdout and "no CPU fallback" in r.stdout


def test_create_rejects_unknown_config_flags(simka_lib):
    """simka_config.flags (reserved0 up to ABI 5): a bit the library does not know is an error, not a silent mode switch -- checked before
    any device is touched, so this runs without a GPU (advisor, round 4)."""
    import ctypes as C
    import simka_amd
    from simka_amd import api
    cfg = api.Config()
    cfg.struct_size = C.sizeof(api.Config)
    cfg.nb_samples, cfg.kmer_size, cfg.abundance_min, cfg.abundance_max = 2, 21, 2, 999999999
    cfg.shard_index, cfg.shard_count = 0, 1
    cfg.flags = 0x10
    h = C.c_void_p()
    assert simka_lib.simka_create(C.byref(cfg), C.byref(h)) == 1      # SIMKA_ERR_INVALID
    assert b"flags" in simka_lib.simka_last_error(None)
