"""simka_amd -- MI355X (gfx950) implementation of Simka's k-mer counting + ecological-distance hot path.

Layout: csrc/ (HIP kernels + C ABI, built in-tree into lib/libsimka_hip.so, host driver bin/simka),
api.py (ctypes binding / host-side mirror of the reference flow), synth.py (seeded synthetic reads),
dist.py (multi-GPU sharding over torch.distributed).
"""
from .api import (DIST_COMPLEX, DIST_SIMPLE, SimkaContext, SimkaError, Stats, load_library, matrix_names,  # noqa: F401
                  pack_reads, parse_input_file, read_sequences)

__version__ = "0.1.0"
