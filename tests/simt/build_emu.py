"""Builds tests/simt/libsimt_kernels.so (TEST INFRASTRUCTURE): the real kernel sources compiled with
g++ against the SIMT emulator.  Rebuilt when any kernel header or emulator file is newer."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lz4net_amd", "csrc")
SO = os.path.join(HERE, "libsimt_kernels.so")


def build() -> str:
    deps = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "*.hpp")) + \
        [os.path.join(HERE, "emu_kernels.cpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        flags = ["-DLZ4HIP_HAVE_HC"] if os.path.exists(os.path.join(CSRC, "lz4hip_hc.hpp")) else []
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused",
                        "-I" + HERE, "-I" + CSRC, *flags, "-o", SO, os.path.join(HERE, "emu_kernels.cpp")],
                       check=True)
    return SO
