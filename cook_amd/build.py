"""Builds libcookmatch.so (HIP, gfx950) in-tree: `python -m cook_amd.build`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "engine.hip")
OUT = os.path.join(HERE, "libcookmatch.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(ROOT, "include", "cookmatch.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return OUT
    # exports: the C functions of include/cookmatch.h and nothing else (-fvisibility=hidden for the library's own code, a linker
    # version script for what the C++ runtime's headers force to default visibility: std:: template instantiations)
    vmap = os.path.join(HERE, "csrc", "exports.map")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
           "-ffp-contract=off", "-fvisibility=hidden", "-fvisibility-inlines-hidden", f"-Wl,--version-script={vmap}", "-Wall",
           "-Wno-unused-function", "-o", OUT + f".{os.getpid()}.tmp", SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    try:  # concurrent builders each write their own file; the rename is atomic
        subprocess.check_call(cmd)
        os.replace(cmd[-2], OUT)
    finally:
        if os.path.exists(cmd[-2]):
            os.remove(cmd[-2])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
