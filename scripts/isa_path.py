#!/usr/bin/env python3
"""Follows ONE path through a kernel's gfx950 assembly and counts the instructions on it (a wave issues one instruction every fourth
cycle: for a single-wave dependent loop such as the placement walk the count IS the time).

usage: isa_path.py <asm file of one function> <start label> <decisions>
  <asm file>   e.g. the output of scripts/isa_fn.py cut to one function (awk '/^NAME:/,/^.Lfunc_end/')
  <decisions>  a string of T / N: taken or not, for the conditional branches in the order the path meets them; the trace stops
               at the first branch without a decision (printing it) or when it is back at the start label
"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start, dec = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = i
    i, di, n, out = lab[start] + 1, 0, 0, []
    while i < len(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        t = lines[i].split(";")[0].strip()
        i += 1
        if m:
            out.append("  [" + m.group(1) + "]")
            if m.group(1) == start:
                break
            continue
        if not t or t.startswith("."):
            continue
        n += 1
        out.append(t)
        if t.startswith("s_cbranch"):
            tgt = t.split()[-1]
            if di >= len(dec):
                out.append("  ?? no decision for " + t)
                break
            d = dec[di]
            di += 1
            out[-1] += "   <" + d + ">"
            if d == "T":
                if tgt == start:
                    break
                i = lab[tgt] + 1
                out.append("  [" + tgt + "]")
        elif t.startswith("s_branch"):
            tgt = t.split()[-1]
            if tgt == start:
                break
            i = lab[tgt] + 1
            out.append("  [" + tgt + "]")
        elif t.startswith("s_endpgm"):
            break
    print("\n".join(out))
    print("instructions on the path:", n)


if __name__ == "__main__":
    main()
