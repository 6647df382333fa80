"""Builds libsimka_hip.so (gfx950 HIP kernels + C ABI) and the `simka` host driver, in-tree.

`python -m simka_amd.build` or `simka_amd.build.build()`.  hipcc cross-compiles for gfx950
without a GPU, so this also runs in the CPU-only container.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
BIN_DIR = os.path.join(HERE, "bin")
LIB_PATH = os.path.join(LIB_DIR, "libsimka_hip.so")
CLI_PATH = os.path.join(BIN_DIR, "simka")

LIB_SOURCES = ["simka_ctx.hip", "simka_wide.hip", "simka_host.cpp"]
# every file a translation unit of the library #includes (simka_ctx.hip pulls the other .hip files in)
LIB_DEPS = LIB_SOURCES + ["simka_kernels.hip", "simka_skm.hip", "simka_sort.hip", "simka_ingest.hip", "simka_kernels.h",
                          "simka_device.h", "simka_wide.h", "simka_efence.h", "simka_trace.h", "../../include/simka_hip.h"]
CLI_SOURCES = ["simka_cli.cpp"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps if os.path.exists(os.path.join(CSRC, d)))


def _run(cmd):
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(BIN_DIR, exist_ok=True)
    hipcc = _hipcc()
    if force or _stale(LIB_PATH, LIB_DEPS):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wno-unused-result", "-o", LIB_PATH] + LIB_SOURCES + ["-lz", "-ldl"]
        out = _run(cmd)
        if verbose:
            print(out)
    if os.path.exists(os.path.join(CSRC, CLI_SOURCES[0])) and (force or _stale(CLI_PATH, CLI_SOURCES + LIB_DEPS)):
        cmd = [hipcc, "-O2", "-std=c++17", "-o", CLI_PATH] + CLI_SOURCES + \
              ["-L" + LIB_DIR, "-lsimka_hip", "-Wl,-rpath,$ORIGIN/../lib", "-lz", "-lpthread"]
        out = _run(cmd)
        if verbose:
            print(out)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
