"""Builds the SIMT-emulator variant of libcookmatch (TEST INFRASTRUCTURE): the same cook_amd/csrc/*.hip sources
compiled by g++ against tests/simt_emu/hip/hip_runtime.h.  Lets the C-ABI parity tests exercise kernel logic on a
machine without a GPU.  Never shipped, never loaded by cook_amd."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcookmatch_emu.so")
OUT_SHIPPED = os.path.join(HERE, "libcookmatch_emu_shipped.so")  # the shipped launch shapes (window 512, 256 slots, 768-thread resolve ...)
SRC = os.path.join(ROOT, "cook_amd", "csrc", "engine.hip")
DEPS = [os.path.join(ROOT, "cook_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "cook_amd", "csrc"))] + [
    os.path.join(HERE, "emu.cpp"), os.path.join(HERE, "emu_prims.hpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
    os.path.join(ROOT, "include", "cookmatch.h")]


def build(force=False, shipped_shapes=False):
    """shipped_shapes: compile with the launch shapes of the GPU build (cook_amd/csrc/platform.hpp COOK_SHAPE) instead of the small
    ones the emulated suite normally runs with — slower (768 fibers per resolve block), used by tests/test_parity_emu_shipped.py.
    COOK_EMU_DEFS (environment): extra -D definitions for a study build of the whole suite, e.g. COOK_EMU_DEFS=-DCOOK_MV_LM=12
    (the library then gets its own file name)."""
    out = OUT_SHIPPED if shipped_shapes else OUT
    defs = os.environ.get("COOK_EMU_DEFS", "").split()
    if defs:
        tag = "".join(c if c.isalnum() else "_" for c in "".join(defs))
        out = out[:-3] + tag + ".so"
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
        return out
    tmp = f"{out}.{os.getpid()}.tmp"  # several test processes may build at once (pytest -n): each writes its own file, the rename is atomic
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-I", HERE, "-x", "c++", SRC,
           os.path.join(HERE, "emu.cpp"), "-o", tmp]
    if shipped_shapes:
        cmd.insert(1, "-DCOOK_EMU_SHIPPED_SHAPES")
    cmd[1:1] = defs
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, shipped_shapes="--shipped" in sys.argv))
