"""ctypes binding of include/simka_hip.h plus a small host-side mirror of the reference flow.

The reference drives the path as   simkaCount (per sample)  ->  simkaMerge (per partition)  ->
SimkaStatistics += / outputMatrix   (ref: src/SimkaPotara.hpp:813-1187).  `SimkaContext` keeps that
shape: count_sample() per sample, merge(), stats(), matrices()/write_matrices().

There is no CPU fallback: if libsimka_hip.so is missing, or no HIP device is visible, this raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

SIMKA_OK = 0
ERR_NAMES = {1: "INVALID", 2: "HIP", 3: "NOMEM", 4: "OVERFLOW", 5: "STATE", 6: "IO", 7: "UNSUPPORTED"}
DIST_SIMPLE = 1
DIST_COMPLEX = 2


class SimkaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("simka error %s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nb_samples", C.c_uint32), ("kmer_size", C.c_uint32),
                ("abundance_min", C.c_uint32), ("abundance_max", C.c_uint32), ("dist_flags", C.c_uint32),
                ("device", C.c_int32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("log2_partitions", C.c_uint32), ("log2_subranges", C.c_uint32), ("flags", C.c_uint32),
                ("max_kmers_per_sample", C.c_uint64), ("solid_capacity", C.c_uint64), ("csr_capacity", C.c_uint64),
                ("stream", C.c_void_p)]


class Reads(C.Structure):
    _fields_ = [("packed", C.c_void_p), ("nb_bases", C.c_uint64), ("nb_reads", C.c_uint64), ("offsets", C.c_void_p),
                ("fixed_len", C.c_uint32), ("on_device", C.c_uint32), ("nb_input_reads", C.c_uint64)]


class SampleTotals(C.Structure):
    _fields_ = [("nb_reads", C.c_uint64), ("nb_distinct", C.c_uint64), ("nb_kmers", C.c_uint64), ("sum_sq", C.c_uint64),
                ("kmer_occurrences", C.c_uint64), ("distinct_all", C.c_uint64)]


class SpectrumInfo(C.Structure):
    _fields_ = [("nb_records", C.c_uint64), ("nb_partitions", C.c_uint64), ("key_words", C.c_uint64)]


_U64P = C.POINTER(C.c_uint64)


class StatsView(C.Structure):
    _fields_ = [("nb_samples", C.c_uint32), ("dist_flags", C.c_uint32), ("nb_pairs", C.c_uint64),
                ("nb_distinct", _U64P), ("nb_kmers", _U64P), ("sum_sq", _U64P), ("shared_ij", _U64P), ("shared_ji", _U64P),
                ("distinct_shared", _U64P), ("bray_curtis", _U64P), ("chord", _U64P), ("hellinger", _U64P),
                ("whittaker", _U64P), ("canberra", _U64P), ("kl", C.POINTER(C.c_double)),
                ("nb_distinct_kmers", C.c_uint64), ("nb_shared_kmers", C.c_uint64)]


_lib = None


def load_library(build_if_missing=True):
    """dlopen libsimka_hip.so (building it first if the sources are newer). Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch wheels bundle their own libamdhip64: import it FIRST so that this process ends up with a single
        # HIP runtime (loading /opt/rocm's copy before torch's leaves two runtimes fighting over the device)
        import torch  # noqa: F401
    except Exception:
        pass
    path = _build.LIB_PATH
    ov = os.environ.get("SIMKA_LIB_OVERRIDE")      # A/B runs: another build of the same sources, accepted only from the package's own lib directory
    if ov and os.path.dirname(os.path.realpath(ov)) == os.path.realpath(_build.LIB_DIR):
        path = ov
    if build_if_missing and not os.path.exists(path):
        _build.build()
    if not os.path.exists(path):
        raise RuntimeError("libsimka_hip.so is missing (%s): run `python -m simka_amd.build`; there is no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sigs = {
        "simka_abi_version": (i32, []),
        "simka_create": (i32, [C.POINTER(Config), C.POINTER(vp)]),
        "simka_destroy": (None, [vp]),
        "simka_last_error": (C.c_char_p, [vp]),
        "simka_sync": (i32, [vp]),
        "simka_reset": (i32, [vp]),
        "simka_count_sample": (i32, [vp, u32, C.POINTER(Reads)]),
        "simka_get_sample_totals": (i32, [vp, u32, C.POINTER(SampleTotals)]),
        "simka_ingest_begin": (i32, [vp, u32]),
        "simka_ingest_text": (i32, [vp, u32, vp, u64, i32, C.POINTER(u64), C.POINTER(i32)]),
        "simka_ingest_count": (i32, [vp, u32, C.POINTER(u64), C.POINTER(u64)]),
        "simka_ingest_text_device": (i32, [vp, u32, vp, u64, i32, C.POINTER(u64), C.POINTER(i32)]),
        "simka_device_upload": (i32, [i32, vp, vp, u64]),
        "simka_device_cpulist": (i32, [i32, C.c_char_p, u64]),
        "simka_sample_spectrum_info": (i32, [vp, u32, C.POINTER(SpectrumInfo)]),
        "simka_export_sample": (i32, [vp, u32, vp, vp, vp]),
        "simka_import_sample": (i32, [vp, u32, C.POINTER(SampleTotals), vp, u64, vp, vp, u64]),
        "simka_export_sample_device": (i32, [vp, u32, vp, vp, vp]),
        "simka_import_sample_device": (i32, [vp, u32, C.POINTER(SampleTotals), vp, u64, vp, vp, u64]),
        "simka_samples_spectrum_info": (i32, [vp, vp, u32, vp, vp]),
        "simka_gather_samples_device": (i32, [vp, vp, u32, vp, vp, vp]),
        "simka_import_samples_device": (i32, [vp, vp, u32, vp, u64, u64, vp, vp, u64, vp, vp, u64]),
        "simka_pack_plan": (i32, [vp, vp, u32, u32, vp]),
        "simka_pack_run": (i32, [vp, vp, vp, vp, u32, u32]),
        "simka_import_block_device": (i32, [vp, vp, u32, vp, u64, u64, u32, vp, u64, vp, vp, u64]),
        "simka_device_memory": (i32, [i32, C.POINTER(u64), C.POINTER(u64)]),
        "simka_device_alloc": (i32, [i32, u64, C.POINTER(vp)]),
        "simka_device_free": (i32, [i32, vp]),
        "simka_device_copy": (i32, [i32, vp, i32, vp, u64]),
        "simka_host_alloc": (i32, [u64, C.POINTER(vp)]),
        "simka_host_free": (i32, [vp]),
        "simka_default_log2_partitions": (u32, [u64, u32]),
        "simka_gather_samples_device_wide": (i32, [vp, vp, u32, vp, vp, vp, vp]),
        "simka_import_samples_device_wide": (i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp]),
        "simka_merge": (i32, [vp]),
        "simka_stats_device_buffer": (i32, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "simka_stats_download": (i32, [vp, vp, u64, C.POINTER(StatsView)]),
        "simka_stats_describe": (i32, [u32, u32, vp, u64, C.POINTER(StatsView)]),
        "simka_stats_nb_u64": (u64, [u32, u32]),
        "simka_stats_layout": (i32, [u32, u32, C.POINTER(u64)]),
        "simka_stats_device_ranges": (i32, [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]),
        "simka_totals_download": (i32, [vp, vp]),
        "simka_totals_upload": (i32, [vp, vp]),
        "simka_comm_library": (i32, [C.c_char_p, u64]),
        "simka_comm_unique_id": (i32, [vp]),
        "simka_comm_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "simka_comm_destroy": (None, [vp]),
        "simka_comm_last_error": (C.c_char_p, [vp]),
        "simka_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "simka_stats_allreduce": (i32, [vp, vp]),
        "simka_totals_allreduce": (i32, [vp, vp]),
        "simka_stats_allreduce_head": (i32, [vp, vp]),
        "simka_comm_allreduce_u64": (i32, [vp, vp, u64, vp]),
        "simka_comm_alltoallv": (i32, [vp, vp, vp, vp, vp, vp, vp, u32, vp]),
        "simka_nb_matrices": (i32, []),
        "simka_matrix_name": (C.c_char_p, [i32]),
        "simka_matrix_enabled": (i32, [i32, u32]),
        "simka_compute_matrix": (i32, [C.POINTER(StatsView), i32, vp]),
        "simka_write_matrix_csv": (i32, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, vp, i32]),
        "simka_pack_read": (C.c_int64, [C.c_char_p, u64, vp, C.POINTER(u64), vp]),
        "simka_profile_enable": (i32, [vp, i32]),
        "simka_profile_reset": (i32, [vp]),
        "simka_profile_nb_kernels": (i32, [vp]),
        "simka_profile_get": (i32, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double)]),
        "simka_arena_info": (i32, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "simka_get_geometry": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]),
        "simka_count_paths": (i32, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "simka_synth_genomes": (i32, [vp, vp, u32, u64, u64]),
        "simka_synth_reads": (i32, [vp, vp, u64, u32, vp, u64, u64, vp, vp, u32, u64, u32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def matrix_names(dist_flags=DIST_SIMPLE | DIST_COMPLEX):
    lib = load_library()
    return [lib.simka_matrix_name(w).decode() for w in range(lib.simka_nb_matrices()) if lib.simka_matrix_enabled(w, dist_flags)]


def stats_layout(nb_samples, dist_flags):
    """word offsets of the flat statistics buffer (simka_stats_layout)."""
    out = (C.c_uint64 * 8)()
    rc = load_library().simka_stats_layout(nb_samples, dist_flags, out)
    if rc != SIMKA_OK:
        raise SimkaError(rc, "simka_stats_layout")
    return {"nacc": out[0], "acc0": out[1], "tot0": out[2], "derived": out[3], "nb_pairs": out[4], "head": out[5], "total": out[6]}


class Stats:
    """Host copy of the flat accumulator buffer + typed views (== SimkaStatistics)."""

    def __init__(self, nb_samples, dist_flags, flat):
        self.lib = load_library()
        self.nb_samples = int(nb_samples)
        self.dist_flags = int(dist_flags)
        self.layout = stats_layout(self.nb_samples, self.dist_flags)
        flat = np.ascontiguousarray(flat, dtype=np.uint64)
        if flat.size < self.layout["total"]:       # a buffer without the host-derived tail (e.g. straight from an all-reduce)
            flat = np.concatenate([flat, np.zeros(self.layout["total"] - flat.size, dtype=np.uint64)])
        self.flat = flat.copy()
        self.view = StatsView()
        rc = self.lib.simka_stats_describe(self.nb_samples, self.dist_flags, self.flat.ctypes.data, self.flat.size,
                                           C.byref(self.view))
        if rc != SIMKA_OK:
            raise SimkaError(rc, "simka_stats_describe")

    def _arr(self, ptr, n):
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def per_sample(self):
        n = self.nb_samples
        t0 = self.layout["tot0"]
        return {"D": self._arr(self.view.nb_distinct, n), "N": self._arr(self.view.nb_kmers, n), "Q": self._arr(self.view.sum_sq, n),
                "D_all": self.flat[t0 + 3 * n: t0 + 4 * n].copy(), "K_occ": self.flat[t0 + 4 * n: t0 + 5 * n].copy()}

    def pairs(self):
        p = int(self.view.nb_pairs)
        v = self.view
        out = {"S_ij": self._arr(v.shared_ij, p), "S_ji": self._arr(v.shared_ji, p), "a": self._arr(v.distinct_shared, p),
               "bc": self._arr(v.bray_curtis, p)}
        if self.dist_flags & DIST_SIMPLE:
            out["chord"] = self._arr(v.chord, p)
            out["hell"] = self._arr(v.hellinger, p)
        if self.dist_flags & DIST_COMPLEX:
            out["whit"] = self._arr(v.whittaker, p)
            out["canb"] = self._arr(v.canberra, p)
            out["kl"] = np.ctypeslib.as_array(v.kl, shape=(p,)).copy()
        return out

    def dense(self, name):
        """pair array -> N x N (upper triangle i<j filled; S_ji goes to the lower triangle of 'S')."""
        n = self.nb_samples
        iu = np.triu_indices(n, 1)
        pr = self.pairs()
        m = np.zeros((n, n), dtype=np.uint64)
        if name == "S":
            m[iu] = pr["S_ij"]
            m.T[iu] = pr["S_ji"]
        else:
            m[iu] = pr[name]
        return m

    def matrix(self, which):
        n = self.nb_samples
        out = np.zeros((n, n), dtype=np.float32)
        rc = self.lib.simka_compute_matrix(C.byref(self.view), which, out.ctypes.data)
        if rc != SIMKA_OK:
            raise SimkaError(rc, "simka_compute_matrix(%d)" % which)
        return out

    def matrices(self):
        lib = self.lib
        return {lib.simka_matrix_name(w).decode(): self.matrix(w) for w in range(lib.simka_nb_matrices())
                if lib.simka_matrix_enabled(w, self.dist_flags)}

    def write_matrices(self, outdir, sample_ids, gz=True):
        """SimkaStatistics::outputMatrix (ref: src/core/SimkaDistance.cpp:603-649)."""
        os.makedirs(outdir, exist_ok=True)
        ids = (C.c_char_p * len(sample_ids))(*[s.encode() for s in sample_ids])
        for name, m in self.matrices().items():
            m = np.ascontiguousarray(m)
            rc = self.lib.simka_write_matrix_csv(outdir.encode(), name.encode(), ids, len(sample_ids), m.ctypes.data, 1 if gz else 0)
            if rc != SIMKA_OK:
                raise SimkaError(rc, "simka_write_matrix_csv(%s)" % name)


COMM_ID_BYTES = 128


class Comm:
    """One RCCL communicator behind the C ABI (simka_comm_*): rank 0 makes the id (Comm.unique_id()), every rank creates.
    The cross-GPU reduction of the accumulators is then ctx.allreduce_stats(comm) -- SimkaStatistics::operator+= over xGMI."""

    @staticmethod
    def library():
        """which librccl serves the collectives of the C ABI: "<file> (<how it was found>)" (simka_comm_library)"""
        buf = C.create_string_buffer(1024)
        lib = load_library()
        rc = lib.simka_comm_library(buf, 1024)
        if rc != SIMKA_OK:
            raise SimkaError(rc, lib.simka_comm_last_error(None).decode())
        return buf.value.decode()

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        lib = load_library()
        rc = lib.simka_comm_unique_id(buf)
        if rc != SIMKA_OK:
            raise SimkaError(rc, lib.simka_comm_last_error(None).decode())
        return bytes(buf)

    def __init__(self, unique_id, nb_ranks, rank, device):
        self.lib = load_library()
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % COMM_ID_BYTES)
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        rc = self.lib.simka_comm_create(buf, nb_ranks, rank, device, C.byref(h))
        if rc != SIMKA_OK:
            raise SimkaError(rc, self.lib.simka_comm_last_error(None).decode())
        self.h, self.rank, self.nb_ranks = h, rank, nb_ranks

    def _check(self, rc):
        if rc != SIMKA_OK:
            raise SimkaError(rc, self.lib.simka_comm_last_error(self.h).decode())

    def allreduce_u64(self, device_ptr, nb_words, stream=None):
        self._check(self.lib.simka_comm_allreduce_u64(self.h, device_ptr, nb_words, stream))

    def alltoallv(self, send_ptr, send_counts, recv_ptr, recv_counts, elem_bytes, stream=None):
        """uneven all-to-all over device buffers; blocks are laid out rank-major on both sides (displacements = prefix sums)"""
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rcn = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(np.uint64)
        rd = np.concatenate([[0], np.cumsum(rcn)[:-1]]).astype(np.uint64)
        self._check(self.lib.simka_comm_alltoallv(self.h, send_ptr, sc.ctypes.data, sd.ctypes.data, recv_ptr, rcn.ctypes.data, rd.ctypes.data,
                                                  elem_bytes, stream))

    def close(self):
        if getattr(self, "h", None):
            self.lib.simka_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SimkaContext:
    def __init__(self, nb_samples, kmer_size=21, abundance_min=2, abundance_max=999999999, simple_dist=False,
                 complex_dist=False, device=0, shard_index=0, shard_count=1, max_kmers_per_sample=0, log2_partitions=0,
                 log2_subranges=0, solid_capacity=0, csr_capacity=0, stream=None):
        self.lib = load_library()
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.nb_samples = nb_samples
        cfg.kmer_size = kmer_size
        cfg.abundance_min = abundance_min
        cfg.abundance_max = abundance_max
        cfg.dist_flags = (DIST_SIMPLE if simple_dist else 0) | (DIST_COMPLEX if complex_dist else 0)
        cfg.device = device
        cfg.shard_index = shard_index
        cfg.shard_count = shard_count
        cfg.max_kmers_per_sample = max_kmers_per_sample
        cfg.log2_partitions = log2_partitions
        cfg.log2_subranges = log2_subranges
        cfg.solid_capacity = solid_capacity
        cfg.csr_capacity = csr_capacity
        cfg.stream = stream
        self.cfg = cfg
        self.nb_samples = nb_samples
        self.dist_flags = cfg.dist_flags
        h = C.c_void_p()
        rc = self.lib.simka_create(C.byref(cfg), C.byref(h))
        if rc != SIMKA_OK:
            raise SimkaError(rc, self.lib.simka_last_error(None).decode())
        self.h = h
        self._keep = []

    def _check(self, rc):
        if rc != SIMKA_OK:
            raise SimkaError(rc, self.lib.simka_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.simka_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- count side -------------------------------------------------------------------------
    def ingest_text(self, index, texts):
        """Device-side ingest (simka_ingest_*): `texts` = the raw bytes of the sample's FASTA / FASTQ files, in order.  Returns
        (nb_bases, nb_reads), or None when a file is irregular (nothing was counted: parse the sample on the host)."""
        self._check(self.lib.simka_ingest_begin(self.h, index))
        for t in texts:
            buf = np.frombuffer(t, dtype=np.uint8)
            fmt = 1 if len(t) and t.lstrip(b"\r\n")[:1] == b"@" else 0
            nr, irr = C.c_uint64(), C.c_int()
            self._check(self.lib.simka_ingest_text(self.h, index, buf.ctypes.data if len(t) else None, len(t), fmt, C.byref(nr), C.byref(irr)))
            if irr.value:
                return None
            if nr.value == 0:
                break          # a file that delivers no read ends the sample (SimkaInputIterator)
        nb, nr = C.c_uint64(), C.c_uint64()
        self._check(self.lib.simka_ingest_count(self.h, index, C.byref(nb), C.byref(nr)))
        return nb.value, nr.value

    def count_sample(self, index, packed, nb_bases, nb_reads, fixed_len=0, offsets=None, on_device=False, nb_input_reads=0, host_pointer=False):
        """`packed`/`offsets`: numpy uint64 arrays (host), integer device pointers (on_device=True) or integer HOST pointers
        (host_pointer=True: e.g. pinned buffers of simka_host_alloc / torch pin_memory)."""
        r = Reads()
        if on_device or host_pointer:
            r.packed = int(packed)
            r.offsets = int(offsets) if offsets is not None else None
        else:
            packed = np.ascontiguousarray(packed, dtype=np.uint64)
            self._keep = [packed]
            r.packed = packed.ctypes.data
            if offsets is not None:
                offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
                self._keep.append(offsets)
                r.offsets = offsets.ctypes.data
        r.nb_bases = int(nb_bases)
        r.nb_reads = int(nb_reads)
        r.fixed_len = int(fixed_len)
        r.on_device = 1 if on_device else 0
        r.nb_input_reads = int(nb_input_reads)
        self._check(self.lib.simka_count_sample(self.h, index, C.byref(r)))

    def sample_totals(self, index):
        t = SampleTotals()
        self._check(self.lib.simka_get_sample_totals(self.h, index, C.byref(t)))
        return {"nb_reads": t.nb_reads, "D": t.nb_distinct, "N": t.nb_kmers, "Q": t.sum_sq, "K_occ": t.kmer_occurrences,
                "D_all": t.distinct_all}

    # -- -keep-tmp: spectra out of / into the context -----------------------------------------
    def export_sample(self, index):
        """(totals, part_counts u32[nparts], keys u64[n], counts u32[n]) of a counted sample -- what -keep-tmp persists."""
        info = SpectrumInfo()
        self._check(self.lib.simka_sample_spectrum_info(self.h, index, C.byref(info)))
        pc = np.zeros(info.nb_partitions, dtype=np.uint32)
        kw = max(int(info.key_words), 1)                 # 2 for kmer_size >= 32: high words, then low words
        keys = np.zeros(max(info.nb_records * kw, 1), dtype=np.uint64)
        counts = np.zeros(max(info.nb_records, 1), dtype=np.uint32)
        self._check(self.lib.simka_export_sample(self.h, index, pc.ctypes.data, keys.ctypes.data, counts.ctypes.data))
        return self.sample_totals(index), pc, keys[:info.nb_records * kw], counts[:info.nb_records]

    def import_sample(self, index, totals, part_counts, keys, counts):
        t = SampleTotals(totals["nb_reads"], totals["D"], totals["N"], totals["Q"], totals["K_occ"], totals["D_all"])
        pc = np.ascontiguousarray(part_counts, dtype=np.uint32)
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        self._check(self.lib.simka_import_sample(self.h, index, C.byref(t), pc.ctypes.data, len(pc), k.ctypes.data if len(k) else None,
                                                 c.ctypes.data if len(c) else None, len(c)))

    def export_sample_device(self, index, device):
        """export_sample with keys (int64) / counts (int32) as torch tensors on the context's GPU (multi-GPU exchange)."""
        import torch
        info = SpectrumInfo()
        self._check(self.lib.simka_sample_spectrum_info(self.h, index, C.byref(info)))
        pc = np.zeros(info.nb_partitions, dtype=np.uint32)
        keys = torch.empty(info.nb_records * max(int(info.key_words), 1), dtype=torch.int64, device=device)
        counts = torch.empty(info.nb_records, dtype=torch.int32, device=device)
        self._check(self.lib.simka_export_sample_device(self.h, index, pc.ctypes.data, keys.data_ptr() if info.nb_records else None,
                                                        counts.data_ptr() if info.nb_records else None))
        return self.sample_totals(index), pc, keys, counts

    def import_sample_device(self, index, totals, part_counts, keys, counts):
        """keys / counts: torch tensors on the context's GPU (or slices of one)."""
        t = SampleTotals(totals["nb_reads"], totals["D"], totals["N"], totals["Q"], totals["K_occ"], totals["D_all"])
        pc = np.ascontiguousarray(part_counts, dtype=np.uint32)
        n = int(counts.numel())
        self._check(self.lib.simka_import_sample_device(self.h, index, C.byref(t), pc.ctypes.data, len(pc), keys.data_ptr() if n else None,
                                                        counts.data_ptr() if n else None, n))

    # batch forms: one synchronisation for many samples (simka_amd/dist.py)
    def spectrum_info(self, index):
        """(nb_records, nb_partitions, key_words) of a counted sample"""
        info = SpectrumInfo()
        self._check(self.lib.simka_sample_spectrum_info(self.h, index, C.byref(info)))
        return int(info.nb_records), int(info.nb_partitions), max(int(info.key_words), 1)

    def nb_partitions(self, sample=None):
        if sample is not None:
            return self.spectrum_info(sample)[1]
        g = self.geometry()
        return 1 << (g["log2_level1"] + g["log2_level2"])

    def samples_spectrum_info(self, samples):
        """-> (part_counts u32 [nb, nparts], totals SampleTotals array) of counted samples."""
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        pc = np.zeros((len(idx), self.nb_partitions(int(idx[0]) if len(idx) else None)), dtype=np.uint32)
        tot = (SampleTotals * max(len(idx), 1))()
        self._check(self.lib.simka_samples_spectrum_info(self.h, idx.ctypes.data, len(idx), pc.ctypes.data, C.addressof(tot)))
        return pc, tot

    def gather_samples_device(self, samples, out_offsets, keys, counts):
        """Copy run (sample j, partition p) to keys/counts[out_offsets[j, p] ...] (torch tensors on this GPU)."""
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        off = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        if len(idx) == 0 or keys.numel() == 0:
            return
        self._check(self.lib.simka_gather_samples_device(self.h, idx.ctypes.data, len(idx), off.ctypes.data, keys.data_ptr(), counts.data_ptr()))

    def gather_samples_device_wide(self, samples, out_offsets, keys_hi, keys_lo, counts):
        """kmer_size >= 32: high and low key words go to separate tensors"""
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        off = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        if len(idx) == 0 or counts.numel() == 0:
            return
        self._check(self.lib.simka_gather_samples_device_wide(self.h, idx.ctypes.data, len(idx), off.ctypes.data, keys_hi.data_ptr(), keys_lo.data_ptr(),
                                                              counts.data_ptr()))

    def import_samples_device_wide(self, samples, totals, sample_offsets, sample_records, keys_hi, keys_lo, counts):
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        so = np.ascontiguousarray(sample_offsets, dtype=np.uint64)
        sr = np.ascontiguousarray(sample_records, dtype=np.uint64)
        n = int(counts.numel())
        self._check(self.lib.simka_import_samples_device_wide(self.h, idx.ctypes.data, len(idx), C.addressof(totals), so.ctypes.data, sr.ctypes.data,
                                                              keys_hi.data_ptr() if n else None, keys_lo.data_ptr() if n else None,
                                                              counts.data_ptr() if n else None))

    def import_samples_device(self, samples, totals, part_lo, part_counts, in_offsets, nb_partitions, keys, counts):
        """part_counts / in_offsets: [nb, width] for the partitions [part_lo, part_lo + width); totals: SampleTotals array."""
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        pc = np.ascontiguousarray(part_counts, dtype=np.uint32)
        off = np.ascontiguousarray(in_offsets, dtype=np.uint64)
        n = int(keys.numel())
        self._check(self.lib.simka_import_samples_device(self.h, idx.ctypes.data, len(idx), C.addressof(totals), part_lo, pc.shape[1], pc.ctypes.data,
                                                         off.ctypes.data, nb_partitions, keys.data_ptr() if n else None,
                                                         counts.data_ptr() if n else None, n))

    # -- merge side -------------------------------------------------------------------------
    # the exchange with its tables on the device (simka_pack_plan / _run, simka_import_block_device)
    def pack_plan(self, samples, nb_ranges):
        """-> records bound for each of nb_ranges ranks (rank g: partitions [P g / nb_ranges, P (g + 1) / nb_ranges))"""
        idx = np.ascontiguousarray(samples, dtype=np.uint32)
        out = np.zeros(nb_ranges, dtype=np.uint64)
        self._keep_plan = idx
        self._check(self.lib.simka_pack_plan(self.h, idx.ctypes.data if len(idx) else None, len(idx), nb_ranges, out.ctypes.data))
        return [int(x) for x in out]

    def pack_run(self, keys, counts, meta):
        """keys (int64) / counts (int32): torch tensors of sum(pack_plan) records; meta: int32 tensor [nb_ranges, nb_slots, width] on this GPU (zeroed)"""
        n = int(counts.numel())
        self._check(self.lib.simka_pack_run(self.h, keys.data_ptr() if n else None, counts.data_ptr() if n else None, meta.data_ptr() if meta is not None else None,
                                            int(meta.shape[1]) if meta is not None else 0, int(meta.shape[2]) if meta is not None else 0))

    def import_block_device(self, slot_samples, totals, part_lo, part_width, meta, nb_partitions, keys, counts):
        """slot_samples: uint32 [nb_slots_total] (0xffffffff: empty slot); totals: SampleTotals array per slot; meta: int32 device tensor [nb_slots_total, width]"""
        ss = np.ascontiguousarray(slot_samples, dtype=np.uint32)
        n = int(counts.numel())
        self._check(self.lib.simka_import_block_device(self.h, ss.ctypes.data, len(ss), C.addressof(totals), int(part_lo), int(part_width), int(meta.shape[-1]),
                                                       meta.data_ptr(), int(nb_partitions), keys.data_ptr() if n else None, counts.data_ptr() if n else None, n))

    def merge(self):
        self._check(self.lib.simka_merge(self.h))

    def sync(self):
        self._check(self.lib.simka_sync(self.h))

    def reset(self):
        self._check(self.lib.simka_reset(self.h))

    def stats_device_buffer(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self.lib.simka_stats_device_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stats_device_ranges(self):
        """((head_ptr, head_words), (totals_ptr, totals_words)) -- see simka_stats_device_ranges."""
        hp, tp = C.c_void_p(), C.c_void_p()
        hn, tn = C.c_uint64(), C.c_uint64()
        self._check(self.lib.simka_stats_device_ranges(self.h, C.byref(hp), C.byref(hn), C.byref(tp), C.byref(tn)))
        return (hp.value, hn.value), (tp.value, tn.value)

    def allreduce_stats(self, comm, which="all"):
        """SimkaStatistics::operator+= across GPUs: ONE RCCL all-reduce of the flat u64 accumulators on the context's stream.
        which: "all" (after merge), "totals" (before merge, -complex-dist), "head" (after merge when the totals are global)."""
        fn = {"all": self.lib.simka_stats_allreduce, "totals": self.lib.simka_totals_allreduce, "head": self.lib.simka_stats_allreduce_head}[which]
        self._check(fn(self.h, comm.h))

    def totals_download(self):
        out = np.zeros(5 * self.nb_samples, dtype=np.uint64)
        self._check(self.lib.simka_totals_download(self.h, out.ctypes.data))
        return out

    def totals_upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self._check(self.lib.simka_totals_upload(self.h, arr.ctypes.data))

    def stats(self):
        n = self.lib.simka_stats_nb_u64(self.nb_samples, self.dist_flags)
        flat = np.zeros(n, dtype=np.uint64)
        self._check(self.lib.simka_stats_download(self.h, flat.ctypes.data, n, None))
        return Stats(self.nb_samples, self.dist_flags, flat)

    # -- profiling --------------------------------------------------------------------------
    def profile_enable(self, on=True, only=None):
        """only: kernel names (keys of profile()) to time; default every kernel"""
        flag = 1 if on else 0
        if on and only:
            names = []
            for w in range(self.lib.simka_profile_nb_kernels(self.h)):
                name = C.c_char_p(); n = C.c_uint64(); ms = C.c_double()
                self._check(self.lib.simka_profile_get(self.h, w, C.byref(name), C.byref(n), C.byref(ms)))
                names.append(name.value.decode())
            flag = 0
            for k in only:
                flag |= 1 << (names.index(k) + 1)
        self._check(self.lib.simka_profile_enable(self.h, flag))

    def profile_reset(self):
        self._check(self.lib.simka_profile_reset(self.h))

    def profile(self):
        out = {}
        for w in range(self.lib.simka_profile_nb_kernels(self.h)):
            name = C.c_char_p()
            n = C.c_uint64()
            ms = C.c_double()
            self._check(self.lib.simka_profile_get(self.h, w, C.byref(name), C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (n.value, ms.value)
        return out

    def geometry(self):
        l1, l2, t = C.c_uint32(), C.c_uint32(), C.c_uint32()
        a, c = C.c_uint64(), C.c_uint64()
        self._check(self.lib.simka_get_geometry(self.h, C.byref(l1), C.byref(l2), C.byref(t), C.byref(a), C.byref(c)))
        return {"log2_level1": l1.value, "log2_level2": l2.value, "log2_subranges": t.value, "arena_capacity": a.value,
                "csr_capacity": c.value}

    def arena_info(self):
        """simka_arena_info: mode (mapped range / plain allocation), records reserved and backed, retired address space of the process"""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.simka_arena_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"mapped_range": bool(a.value), "reserved_records": b.value, "mapped_records": c.value, "retired_va_bytes": d.value}

    def count_paths(self):
        """How the samples counted so far were counted (simka_count_paths)."""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.simka_count_paths(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"partitioned": a.value, "sorted": b.value, "exact_redone": c.value, "full_sorts": d.value}


# ---- host-side ingest (FASTA/FASTQ, plain or gz) for the Python entry points -------------------
def read_sequences(path):
    """Yield the sequences of a FASTA (multi-line) or FASTQ (4-line) file, plain or gzip."""
    import gzip
    with open(path, "rb") as fh:
        magic = fh.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rb") as fh:
        first = fh.read(1)
        if not first:
            return
        rest = fh.read()
    data = first + rest
    lines = data.split(b"\n")
    if first == b">":
        seq = []
        started = False
        for ln in lines:
            ln = ln.rstrip(b"\r")
            if ln.startswith(b">"):
                if started:
                    yield b"".join(seq)
                seq = []
                started = True
            elif started:
                seq.append(ln)
        if started:
            yield b"".join(seq)
    elif first == b"@":
        i = 0
        while i + 1 < len(lines):
            if lines[i].startswith(b"@"):
                yield lines[i + 1].rstrip(b"\r")
                i += 4
            else:
                i += 1
    else:
        raise ValueError("unrecognised sequence file: %s" % path)


def pack_reads(seqs):
    """2-bit pack an iterable of byte strings with simka_pack_read. Returns (packed, offsets, nb_bases, nb_input_reads)."""
    lib = load_library()
    seqs = list(seqs)
    total = sum(len(s) for s in seqs)
    packed = np.zeros(total // 32 + 3, dtype=np.uint64)
    offsets = np.zeros(total + len(seqs) + 2, dtype=np.uint64)
    nb = C.c_uint64(0)
    nfrag = 0
    for s in seqs:
        r = lib.simka_pack_read(s, len(s), packed.ctypes.data, C.byref(nb), offsets.ctypes.data + 8 * nfrag)
        if r < 0:
            raise SimkaError(1, "simka_pack_read")
        nfrag += r
    offsets[nfrag] = nb.value
    return packed[: nb.value // 32 + 3], offsets[: nfrag + 1].copy(), nb.value, len(seqs)


def parse_input_file(path):
    """The -in grammar (ref: src/core/SimkaAlgorithm.cpp:245-351): ID: f1 , f2 ; g1 , g2"""
    base = os.path.dirname(os.path.realpath(path))
    samples = []
    with open(path) as fh:
        for line in fh:
            line = line.replace(" ", "").rstrip("\r\n")
            if not line:
                continue
            parts = line.split(":")
            if len(parts) < 2:
                raise ValueError("Syntax error in input file")
            sid, rest = parts[0], parts[1]
            paired = [p for p in rest.split(";") if p != ""]
            files = []
            for part in paired:
                for fn in [f for f in part.split(",") if f != ""]:
                    files.append(fn if fn.startswith("/") else os.path.join(base, fn))
            samples.append({"id": sid, "files": files, "nb_paired": len(paired)})
    return samples
