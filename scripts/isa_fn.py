#!/usr/bin/env python3
"""Compile engine.hip to gfx950 assembly and print the resource usage + memory-instruction counts of the functions whose mangled name
contains the given substring (a build tool for kernel work: spills, flat vs global accesses)."""
import re
import subprocess
import sys

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "match_resolve2"
    extra = sys.argv[2:]
    out = "/tmp/engine_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", out, ROOT + "/cook_amd/csrc/engine.hip"] + extra, stderr=subprocess.DEVNULL)
    txt = open(out).read().split("\n")
    i = 0
    while i < len(txt):
        m = re.match(r"^(_Z\w+):", txt[i])
        if m and pat in m.group(1):
            name = m.group(1)
            j = i
            while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
                j += 1
            body = txt[i:j]
            k = j
            info = {}
            while k < len(txt) and k < j + 60:
                mm = re.match(r";\s*(NumVgprs|ScratchSize|TotalNumSgprs|codeLenInByte|Occupancy|LDSByteSize)\s*[:=]\s*(\d+)", txt[k])
                if mm:
                    info.setdefault(mm.group(1), mm.group(2))
                k += 1
            cnt = lambda s: sum(1 for l in body if s in l)  # noqa: E731
            print(name, info, {s: cnt(s) for s in ("scratch_load", "scratch_store", "flat_load", "flat_store", "global_load", "global_store", "ds_read", "ds_write", "s_waitcnt", "v_readlane", "v_writelane", "s_swappc")})
            open(f"/tmp/{name[:40]}.s", "w").write("\n".join(body))
            i = j
        i += 1


if __name__ == "__main__":
    main()
