"""SIMKA_FAULT_TRACE (simka_amd/csrc/simka_trace.h): the registry of device ranges and the ring of launches are dumped when the process dies
of SIGABRT -- the way the ROCm runtime ends it on a GPU memory access fault -- and scripts/fault_resolve.py names the buffer.  Host
code only: runs without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("trace") / "trace_harness")
    r = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tests", "helpers", "trace_harness.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def _run(exe, addr, tmp_path, on=True):
    env = dict(os.environ, SIMKA_FAULT_TRACE_DIR=str(tmp_path))
    env.pop("SIMKA_FAULT_TRACE", None)
    if on:
        env["SIMKA_FAULT_TRACE"] = "1"
    return subprocess.run([exe, hex(addr)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)


def test_dump_on_abort_and_resolution(harness, tmp_path):
    r = _run(harness, 0x7f0040000000 + (1 << 30), tmp_path)          # first byte behind the one mapped chunk of the arena
    assert r.returncode == -6                                        # the original disposition of SIGABRT still ends the process
    assert "[simka-trace] ==== signal 6" in r.stdout and "==== end of dump ====" in r.stdout
    files = [f for f in os.listdir(tmp_path) if f.startswith("simka_fault_trace.")]
    assert len(files) == 1
    dump = open(os.path.join(tmp_path, files[0])).read()
    assert "arena keys chunk" in dump and "&L.d_skm_a" in dump and "freed@" in dump
    assert dump.count("[simka-trace] launch #") == 256               # the ring keeps the last 256 of 300
    log = os.path.join(tmp_path, "log.txt")
    open(log, "w").write(r.stdout)
    rr = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fault_resolve.py"), log], stdout=subprocess.PIPE, text=True)
    assert rr.returncode == 0, rr.stdout
    assert "inside reserved arena keys: virtual range" in rr.stdout
    assert "NO memory mapped at this address" in rr.stdout
    assert "k_skm_count_fast<true>" in rr.stdout


def test_resolution_of_other_addresses(harness, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fault_resolve
    r = _run(harness, 0x7f0000200010, tmp_path)                      # inside a buffer that was freed
    faults, ranges, launches, t = fault_resolve.parse(r.stdout)
    assert faults == [0x7f0000200010] and len(ranges) == 4 and len(launches) == 256
    text = "\n".join(fault_resolve.resolve(faults[0], ranges, t))
    assert "&L.d_skm_a" in text and "FREED" in text
    text = "\n".join(fault_resolve.resolve(0x7f0000101000 + 64, ranges, t))      # 64 bytes past the end of d_foff
    assert "64 bytes past the end of malloc &ctx->d_foff" in text
    text = "\n".join(fault_resolve.resolve(0x7f0040000000, ranges, t))           # first byte of the mapped chunk
    assert "inside chunk" in text and "NO memory" not in text


def test_off_by_default(harness, tmp_path):
    r = _run(harness, 0x1000, tmp_path, on=False)
    assert r.returncode == 3 and "trace disabled" in r.stdout
    assert not [f for f in os.listdir(tmp_path) if f.startswith("simka_fault_trace.")]
