"""The C++ host mirror of LZ4.LZ4Codec (include/lz4net/LZ4Codec.hpp): builds with g++ against liblz4hip.so;
CPU: host logic (argument checks, loud failure without a device); GPU: the reference's AutoTest."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "lz4codec_selftest")


@pytest.fixture(scope="module")
def exe():
    from lz4net_amd import build
    so = build.build()
    src = os.path.join(ROOT, "tests", "cpp", "lz4codec_selftest.cpp")
    hdr = os.path.join(ROOT, "include", "lz4net", "LZ4Codec.hpp")
    if not os.path.exists(EXE) or any(os.path.getmtime(f) > os.path.getmtime(EXE) for f in (src, hdr, so)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-o", EXE, src, so, "-Wl,-rpath," + os.path.dirname(so),
                        "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_host_logic(exe):
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_autotest_on_gpu(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "self test ok" in r.stdout, r.stdout + r.stderr
