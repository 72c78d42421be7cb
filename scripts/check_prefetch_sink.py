#!/usr/bin/env python3
"""Build-time check of the one place where the library relies on a register ALLOCATION: PREFETCH_WORD (gpu_prims.hpp) issues fire-and-forget
loads from inline asm into a "sink" register; the compiler does not know that the data arrives later, so the sink must stay in ONE physical
register, untouched, from the first load to PREFETCH_DRAIN's s_waitcnt.  (Round 5 saw what happens otherwise: a sink whose live range crossed
the way out of the walk's fast loop was moved by the allocator, the data landed in an address register, the GPU raised a memory fault.)
This script compiles engine.hip to gfx950 assembly and, in every function, checks for every run of asm loads up to the asm drain behind it:
  * all the loads of the run write the same vector register,
  * no other instruction between the first load and the drain names that register (no copy, no spill, no reuse),
  * the drain is reached before the function's next barrier or return.
usage: check_prefetch_sink.py [extra hipcc flags, e.g. -DCOOK_WALK_PROF=1]      exit 0 = holds in every function"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    problems, runs = [], 0
    fn = None
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            fn = m.group(1)
        if "#ASMSTART" in lines[i] and i + 1 < len(lines) and re.match(r"\s*global_load_dword v\d+, v\[\d+:\d+\], off", lines[i + 1]):
            sink = re.match(r"\s*global_load_dword (v\d+),", lines[i + 1]).group(1)
            runs += 1
            j = i + 3  # behind #ASMEND
            ok = False
            while j < len(lines):
                t = lines[j].split(";")[0].strip() if "#ASM" not in lines[j] else lines[j].strip()
                if "#ASMSTART" in lines[j]:
                    nxt = lines[j + 1].strip()
                    if nxt.startswith("s_waitcnt vmcnt(0)"):
                        ok = True
                        break
                    mm = re.match(r"global_load_dword (v\d+),", nxt)
                    if mm:
                        if mm.group(1) != sink:
                            problems.append(f"{fn}: asm loads of one run write {sink} and {mm.group(1)} (line {j + 2})")
                        j += 3
                        continue
                if re.match(r"^\.Lfunc_end", lines[j]) or t.startswith("s_endpgm") or t.startswith("s_setpc_b64") or t.startswith("s_barrier"):
                    break
                if t and not t.endswith(":") and not t.startswith(".") and re.search(r"\b" + sink + r"\b", t):
                    problems.append(f"{fn}: `{t}` names the sink {sink} between its loads and the drain (line {j + 1})")
                j += 1
            if not ok:
                problems.append(f"{fn}: asm loads into {sink} (line {i + 2}) reach a barrier / the function's end without a drain")
            # continue scanning behind this run's first load (later loads of the run are re-checked as runs of their own: harmless)
        i += 1
    return runs, problems


def main():
    out = "/tmp/cook_prefetch_check.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "--cuda-device-only",
                           "-S", "-o", out, os.path.join(ROOT, "cook_amd", "csrc", "engine.hip")] + sys.argv[1:], stderr=subprocess.DEVNULL)
    runs, problems = check(open(out).read())
    print(f"{runs} asm-load sites checked, {len(problems)} problems")
    for p in problems[:20]:
        print("  " + p)
    sys.exit(1 if problems or runs == 0 else 0)


if __name__ == "__main__":
    main()
