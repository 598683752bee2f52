"""Builds lz4net_amd/liblz4hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblz4hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I" + CSRC]   # (-I: tools/ab/ headers of tuning builds include the product headers by name)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) +
                  [os.path.join(os.path.dirname(HERE), "include", "lz4hip.h")])


def is_stale() -> bool:
    return not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        if not os.path.exists(HIPCC):
            raise RuntimeError(f"{HIPCC} not found and {SO} is missing or stale: cannot build the gfx950 library")
        extra = os.environ.get("LZ4HIP_BUILD_FLAGS", "").split()      # e.g. -DLZ4HIP_TUNING_BUILD (extra kernel instantiations for sweeps)
        cmd = [HIPCC, *FLAGS, *extra, os.path.join(CSRC, "lz4hip_api.hip"), "-o", SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
