"""Builds lz4net_amd/liblz4hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries the hash of the kernel sources it was compiled from (lz4hip_build_id(), and the marker string
"LZ4HIP_BUILD_ID=<hash>" in its bytes): a prebuilt .so that travelled to another box is recognised as stale by its CONTENT, not by file times."""
from __future__ import annotations

import glob
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblz4hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I" + CSRC]   # (-I: tools/ab/ headers of tuning builds include the product headers by name)
_MARKER = re.compile(rb"LZ4HIP_BUILD_ID=([0-9a-f]{16}|unknown)(\+tuning)?")


def csrc_sha() -> str:
    """First 16 hex digits of the SHA-256 over lz4net_amd/csrc/ (names + contents, sorted): what committed counter files are keyed on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) +
                  [os.path.join(os.path.dirname(HERE), "include", "lz4hip.h")])


def built_id(so: str = SO):
    """The build id inside a library file (None: no library, or one without the marker)."""
    if not os.path.exists(so):
        return None
    with open(so, "rb") as fh:
        m = _MARKER.search(fh.read())
    return m.group(0)[len(b"LZ4HIP_BUILD_ID="):].decode() if m else None


def wanted_id() -> str:
    return csrc_sha() + ("+tuning" if "-DLZ4HIP_TUNING_BUILD" in os.environ.get("LZ4HIP_BUILD_FLAGS", "").split() else "")


def is_stale() -> bool:
    """The library is missing, was compiled from other kernel sources than the tree holds, or is older than the C header."""
    if built_id() != wanted_id():
        return True
    header = os.path.join(os.path.dirname(HERE), "include", "lz4hip.h")
    return os.path.getmtime(header) > os.path.getmtime(SO)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.environ.get("LZ4HIP_KEEP_LIBRARY") and os.path.exists(SO):
        return SO                                                     # A/B runs copy a prebuilt VARIANT over the library (tools/r06/build_variants.sh): use it as it is
    if force or is_stale():
        if not os.path.exists(HIPCC):
            raise RuntimeError(f"{HIPCC} not found and {SO} is missing or stale (library {built_id()}, sources {wanted_id()}): cannot build the gfx950 library")
        extra = os.environ.get("LZ4HIP_BUILD_FLAGS", "").split()      # e.g. -DLZ4HIP_TUNING_BUILD (extra kernel instantiations for sweeps)
        cmd = [HIPCC, *FLAGS, f'-DLZ4HIP_CSRC_SHA="{csrc_sha()}"', *extra, os.path.join(CSRC, "lz4hip_api.hip"), "-o", SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        if built_id() != wanted_id():
            raise RuntimeError(f"{SO}: built library identifies as {built_id()}, expected {wanted_id()}")
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print("build id:", built_id())
