"""Prints the revision of the KERNEL SOURCES (a hash over cook_amd/csrc/* and include/cookmatch.h): profiles/ records it next to
the counters, and bench.py refuses to quote PMC traffic taken at another revision."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_rev() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cook_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/cookmatch.h"]:
        p = os.path.normpath(os.path.join(d, f))
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_rev())
