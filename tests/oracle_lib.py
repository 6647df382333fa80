"""ctypes wrapper of oracle/_build/liboracle.so -- the CPU checker.  Test-side only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
ACC = {"S": 0, "a": 1, "bc": 2, "chord": 3, "hell": 4, "kul": 5, "whit": 6, "canb": 7}

_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        vp = C.c_void_p
        l.oracle_new.restype = vp
        l.oracle_free.argtypes = [vp]
        l.oracle_error.restype = C.c_char_p
        l.oracle_error.argtypes = [vp]
        l.oracle_load_input.argtypes = [vp, C.c_char_p]
        l.oracle_nb_samples.argtypes = [vp]
        l.oracle_sample_id.restype = C.c_char_p
        l.oracle_sample_id.argtypes = [vp, C.c_int]
        l.oracle_sample_nb_files.argtypes = [vp, C.c_int]
        l.oracle_sample_file.restype = C.c_char_p
        l.oracle_sample_file.argtypes = [vp, C.c_int, C.c_int]
        l.oracle_sample_nb_paired.argtypes = [vp, C.c_int]
        l.oracle_add_sample_mem.argtypes = [vp, C.c_char_p, vp, vp, C.c_uint64]
        l.oracle_set_read_policy.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_double]
        l.oracle_auto_max_reads.restype = C.c_uint64
        l.oracle_auto_max_reads.argtypes = [vp]
        l.oracle_run.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.c_int]
        l.oracle_run_shard.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_uint, C.c_uint]
        l.oracle_get_totals.argtypes = [vp, vp]
        l.oracle_get_shard_totals.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, vp]
        l.oracle_get_global.argtypes = [vp, vp]
        l.oracle_get_acc_u64.argtypes = [vp, C.c_int, vp]
        l.oracle_get_kl.argtypes = [vp, vp]
        l.oracle_matrix_name.restype = C.c_char_p
        l.oracle_matrix_name.argtypes = [C.c_int]
        l.oracle_get_matrix.argtypes = [vp, C.c_int, vp]
        l.oracle_write_matrices.argtypes = [vp, C.c_char_p, C.c_int]
        l.oracle_sample_nsolid.restype = C.c_uint64
        l.oracle_sample_nsolid.argtypes = [vp, C.c_int]
        l.oracle_get_sample_solid.argtypes = [vp, C.c_int, vp, vp]
        _lib = l
    return _lib


class Oracle:
    def __init__(self):
        self.l = lib()
        self.h = C.c_void_p(self.l.oracle_new())
        self._keep = []

    def close(self):
        if self.h:
            self.l.oracle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_input(self, path):
        if self.l.oracle_load_input(self.h, path.encode()) != 0:
            raise RuntimeError(self.l.oracle_error(self.h).decode())

    def add_sample_ascii(self, sid, ascii_u8, offsets):
        a = np.ascontiguousarray(ascii_u8, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._keep += [a, o]
        return self.l.oracle_add_sample_mem(self.h, sid.encode(), a.ctypes.data, o.ctypes.data, len(o) - 1)

    def set_read_policy(self, max_reads=0, min_read_size=0, min_shannon=0.0):
        """-max-reads m (0 = all reads), -min-read-size, -min-shannon-index for the samples read from files (row a2)"""
        self.l.oracle_set_read_policy(self.h, int(max_reads), int(min_read_size), float(min_shannon))

    def auto_max_reads(self):
        """what -max-reads 0 resolves to: (min + mean) / 2 of the reads per sample and paired part"""
        return int(self.l.oracle_auto_max_reads(self.h))

    @property
    def n(self):
        return self.l.oracle_nb_samples(self.h)

    def ids(self):
        return [self.l.oracle_sample_id(self.h, i).decode() for i in range(self.n)]

    def files(self, i):
        return [self.l.oracle_sample_file(self.h, i, f).decode() for f in range(self.l.oracle_sample_nb_files(self.h, i))]

    def nb_paired(self, i):
        return self.l.oracle_sample_nb_paired(self.h, i)

    def run(self, k, amin=2, amax=999999999, simple=False, complex_=False, nparts=1, threads=1, shard_index=0, shard_count=1):
        rc = self.l.oracle_run_shard(self.h, k, amin, amax, int(simple), int(complex_), nparts, threads, shard_index, shard_count)
        if rc != 0:
            raise RuntimeError(self.l.oracle_error(self.h).decode())

    def totals(self):
        out = np.zeros((self.n, 6), dtype=np.uint64)
        self.l.oracle_get_totals(self.h, out.ctypes.data)
        return {"nb_reads": out[:, 0], "K_occ": out[:, 1], "D_all": out[:, 2], "D": out[:, 3], "N": out[:, 4], "Q": out[:, 5]}

    def shard_totals(self, nparts, shard_index, shard_count):
        out = np.zeros((self.n, 3), dtype=np.uint64)
        self.l.oracle_get_shard_totals(self.h, nparts, shard_index, shard_count, out.ctypes.data)
        return {"D": out[:, 0], "N": out[:, 1], "Q": out[:, 2]}

    def global_counts(self):
        out = np.zeros(2, dtype=np.uint64)
        self.l.oracle_get_global(self.h, out.ctypes.data)
        return int(out[0]), int(out[1])

    def acc(self, name):
        n = self.n
        out = np.zeros((n, n), dtype=np.uint64)
        self.l.oracle_get_acc_u64(self.h, ACC[name], out.ctypes.data)
        return out

    def kl(self):
        n = self.n
        out = np.zeros((n, n), dtype=np.float64)
        self.l.oracle_get_kl(self.h, out.ctypes.data)
        return out

    def matrix_names(self):
        return [self.l.oracle_matrix_name(w).decode() for w in range(self.l.oracle_nb_matrices())]

    def matrix(self, which):
        n = self.n
        out = np.zeros((n, n), dtype=np.float32)
        self.l.oracle_get_matrix(self.h, which, out.ctypes.data)
        return out

    def write_matrices(self, outdir, gz=False):
        os.makedirs(outdir, exist_ok=True)
        if self.l.oracle_write_matrices(self.h, outdir.encode(), int(gz)) != 0:
            raise RuntimeError("oracle_write_matrices failed")

    def sample_solid(self, i):
        n = self.l.oracle_sample_nsolid(self.h, i)
        k = np.zeros(max(n, 1), dtype=np.uint64)
        c = np.zeros(max(n, 1), dtype=np.uint32)
        self.l.oracle_get_sample_solid(self.h, i, k.ctypes.data, c.ctypes.data)
        return k[:n], c[:n]

    def flat_stats(self, simple, complex_=False, nparts=1, shard_index=0, shard_count=1):
        """Pack the oracle accumulators into the product's flat u64 layout (include/simka_hip.h, simka_stats_layout):
        [8 header | S_ij S_ji a bc (chord hell) (whit klfix) x P | D N Q D_all K_occ x N | (canb, kl f64 x P: host-derived)]"""
        n = self.n
        iu = np.triu_indices(n, 1)
        P = len(iu[0])
        tot = self.shard_totals(nparts, shard_index, shard_count) if shard_count > 1 else self.totals()
        full = self.totals()
        S = self.acc("S")
        parts = [np.zeros(8, dtype=np.uint64), S[iu], S.T[iu], self.acc("a")[iu], self.acc("bc")[iu]]
        if simple:
            parts += [self.acc("chord")[iu], self.acc("hell")[iu]]
        if complex_:
            # klfix = the both-present part of KL in 2^-60 fixed point (what the device accumulates)
            Nk = full["N"].astype(np.float64)
            kl = self.kl()[iu].astype(np.longdouble)
            one = np.zeros(P, dtype=np.longdouble)
            for c, (i, j) in enumerate(zip(*iu)):
                one[c] = np.log(np.longdouble(2)) * (np.longdouble(int(tot["N"][i]) - int(S[i, j])) / np.longdouble(Nk[i]) +
                                                     np.longdouble(int(tot["N"][j]) - int(S[j, i])) / np.longdouble(Nk[j]))
            klfix = np.rint((kl - one) * np.longdouble(2.0 ** 60)).astype(np.int64).view(np.uint64)
            parts += [self.acc("whit")[iu], klfix]
        parts += [tot["D"], tot["N"], tot["Q"], np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)]
        if complex_:
            parts += [np.zeros(2 * P, dtype=np.uint64)]
        flat = np.concatenate([np.asarray(p, dtype=np.uint64) for p in parts])
        d, s = self.global_counts()
        flat[0], flat[1] = d, s
        return flat
