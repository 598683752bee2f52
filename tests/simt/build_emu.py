"""Builds tests/simt/libsimt_kernels.so (TEST INFRASTRUCTURE): the real kernel sources compiled with
g++ against the SIMT emulator.  Rebuilt when any kernel header or emulator file is newer."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lz4net_amd", "csrc")
SO = os.path.join(HERE, "libsimt_kernels.so")


def build(starved: bool = False) -> str:
    """starved=True: the same kernels with the lane decoder's cooperative flush cut down to 4 lines per round, so that
    lanes miss flush rounds again and again and run their rings full (and its cooperative staging
    load to 2 pieces per round, so that lanes run out of input) -- the rare states of the lane decoder (a lane that
    cannot append, a far-match chunk fetched but not consumed) become the common ones."""
    so = SO.replace(".so", "_starved.so") if starved else SO
    deps = glob.glob(os.path.join(ROOT, "tools", "ab", "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "*.hpp")) + \
        [os.path.join(HERE, "emu_kernels.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        flags = ["-DLZ4HIP_HAVE_HC", "-DLZ4HIP_TUNING_BUILD"] if os.path.exists(os.path.join(CSRC, "lz4hip_hc.hpp")) else []
        if starved:
            flags += ["-DLZ4HIP_DEC_FLUSH_RECS=4", "-DLZ4HIP_DEC_LOAD_PIECES=2", "-DLZ4HIP_DEC3_FLUSH_RECS=4", "-DLZ4HIP_DEC3_LOAD_PIECES=2", "-DLZ4HIP_DEC4_FLUSH_RECS=2"]
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused", "-Wno-parentheses", "-Wno-unknown-pragmas",
                        "-I" + HERE, "-I" + CSRC, "-I" + os.path.join(ROOT, "tools", "ab"), *flags, "-o", so, os.path.join(HERE, "emu_kernels.cpp")],
                       check=True)
    return so
