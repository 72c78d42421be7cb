"""Short names of the library's kernels as rocprofv3 reports them.  The kernels of the batched path (cook_amd/csrc/multi.hpp) are all
instances of ONE __global__ template, cook_multi<&kernel, block, args...>; rocprofv3 prints those mangled (its demangler does not know
the `auto` template parameter) or, demangled, as cook_multi<&kernel<...>, ...>: either way the name that matters is the inner one."""
import re


def short_kernel_name(name: str) -> str:
    n = name.strip()
    m = re.match(r"_Z10cook_multiITnDaXadL_Z(?:L|N12_GLOBAL__N_1)(\d+)", n)
    if m:  # mangled: <length><identifier>, then the inner kernel's own template arguments (I...E), if any
        k = int(m.group(1))
        rest = n[m.end():]
        inner, rest = rest[:k], rest[k:]
        t = re.match(r"I(Li(\d+)E)E", rest)  # one integer template argument, e.g. radix_hist<8>
        if t:
            return f"{inner}<{t.group(2)}>"
        t = re.match(r"I((?:\d+[A-Za-z_0-9]+?)+?)E", rest)  # type arguments, e.g. seg_scan_local<SumU4, LoadU4>
        if t and rest.startswith("I") and not rest.startswith("IL"):
            names = []
            a = t.group(1)
            while a and a[0].isdigit():
                mm = re.match(r"(\d+)", a)
                ln = int(mm.group(1))
                names.append(a[mm.end():mm.end() + ln])
                a = a[mm.end() + ln:]
            if names:
                return f"{inner}<{', '.join(names)}>"
        return inner
    m = re.search(r"cook_multi<&\(?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+(?:<[^(>]*>)?)", n)
    if m:
        return m.group(1)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(short_kernel_name(line))
