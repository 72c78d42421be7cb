#!/bin/bash
# LLVM's uniformity analysis of ONE kernel of engine.hip (which values / branches / loop exits the compiler takes for per-lane ones:
# a loop that should be wave-uniform and is not runs under execution masks with its counters in vector registers).
# usage: scripts/uniformity.sh <mangled kernel name> [extra hipcc flags]  ->  /tmp/uniformity_<name>.txt
set -e
KFN=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LL=/tmp/uniformity_$KFN.ll
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off --cuda-device-only -emit-llvm -S -o /tmp/uniformity_all.ll \
  "$ROOT/cook_amd/csrc/engine.hip" "$@" 2>/dev/null
KFN=$KFN python3 - <<'P'
import os, re
kfn = os.environ["KFN"]
src = open("/tmp/uniformity_all.ll").read().split("\n")
out, i = [], 0
while i < len(src):
    l = src[i]
    if l.startswith("define "):
        j = i
        while not src[j].startswith("}"):
            j += 1
        if "@" + kfn + "(" in l:
            out += src[i:j + 1]
        else:  # every other function becomes a declaration
            d = l.replace("define ", "declare ", 1)
            for w in ("internal ", "protected ", "hidden ", "linkonce_odr ", "weak_odr ", "dso_local ", "weak "):
                d = d.replace("declare " + w, "declare ").replace("declare " + w, "declare ")
            d = re.sub(r"\s(comdat|personality|align \d+$).*$", "", d[:d.rfind("{")].rstrip())
            out.append(d)
        i = j + 1
        continue
    out.append(l)
    i += 1
open("/tmp/uniformity_%s.ll" % kfn, "w").write("\n".join(out))
P
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output "$LL" 2> /tmp/uniformity_$KFN.txt
echo "/tmp/uniformity_$KFN.txt: $(grep -c 'DIVERGENT:.*br i1' /tmp/uniformity_$KFN.txt) divergent branches, $(sed -n '/CYCLES WITH DIVERGENT EXIT/,/^$/p' /tmp/uniformity_$KFN.txt | grep -c depth=) loops with a divergent exit"
