"""Builds the SIMT-emulator variant of libcookmatch (TEST INFRASTRUCTURE): the same cook_amd/csrc/*.hip sources
compiled by g++ against tests/simt_emu/hip/hip_runtime.h.  Lets the C-ABI parity tests exercise kernel logic on a
machine without a GPU.  Never shipped, never loaded by cook_amd."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcookmatch_emu.so")
SRC = os.path.join(ROOT, "cook_amd", "csrc", "engine.hip")
DEPS = [os.path.join(ROOT, "cook_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "cook_amd", "csrc"))] + [
    os.path.join(HERE, "emu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "cookmatch.h")]


def build(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-I", HERE, "-x", "c++", SRC,
           os.path.join(HERE, "emu.cpp"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
