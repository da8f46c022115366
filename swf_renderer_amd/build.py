"""Builds libswfr.so (HIP kernels + C-ABI host) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["raster2.hip", "renderer.cpp", "geometry.cpp", "shape_decoder.cpp", "frame_builder.cpp", "bitmap_decode.cpp"]
LIB = os.path.join(HERE, "libswfr.so")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libswfr.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "swfr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden",
           "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
           "-DSWFR_BUILD"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
