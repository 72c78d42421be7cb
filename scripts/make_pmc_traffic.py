#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round2.sh (pmc_*_8pools.txt in
gpurun_out/<dir>): HBM bytes per launch of every kernel both passes saw, stamped with the kernel-source revision the passes were
taken at (gpurun_out/<dir>/kernel_rev.txt) so that bench.py only quotes counters of the binary it is timing.
usage: make_pmc_traffic.py gpurun_out/<dir> <tag>"""
import ast
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(name):
    out = {}
    for line in open(os.path.join(src, name)):
        m = re.match(r"^(.*?) (\{.*\}) dispatches (\d+)$", line.strip())
        if m:
            k = re.sub(r"(_multi|_pack)$", "", re.sub(r"^void ", "", m.group(1)).split("<")[0])  # (the lockstep / packed-context launches of a kernel)
            out[k] = (ast.literal_eval(m.group(2)), int(m.group(3)))
    return out


f, w = read("pmc_FETCH_SIZE_8pools.txt"), read("pmc_WRITE_SIZE_8pools.txt")
kern = {}
for k in sorted(set(f) & set(w)):
    fk, wk = f[k][0]["FETCH_SIZE"], w[k][0]["WRITE_SIZE"]
    kern[k] = {"fetch_kb_per_launch": fk, "write_kb_per_launch": wk, "hbm_bytes_per_launch": int((fk + wk) * 1024), "dispatches": f[k][1]}
if "match_resolve2" in kern and "match_walkers" not in kern:
    # the served walkers' ONE launch per cycle = every round of every pool: the resolve launches of the counted cycle (lockstep form,
    # two pools per launch), summed.  Derived, and labelled so.
    r = kern["match_resolve2"]
    kern["match_walkers"] = {"fetch_kb_per_launch": r["fetch_kb_per_launch"] * r["dispatches"], "write_kb_per_launch": r["write_kb_per_launch"] * r["dispatches"],
                             "hbm_bytes_per_launch": r["hbm_bytes_per_launch"] * r["dispatches"], "dispatches": 1,
                             "derived": "match_resolve2 per launch x its dispatches in the counted cycle (the counter passes run the pools in lockstep "
                                        "launches: rocprofv3 serialises counted dispatches, which a walker waiting for a serve launch cannot survive)"}
doc = {"kernel_rev": open(os.path.join(src, "kernel_rev.txt")).read().strip(),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes: the TCC counter slots of gfx950) of `bench.py --steps 1 "
               "--warmup 0 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline` (8 pools on one MI355X), mean per dispatch "
               "over all dispatches of the kernel incl. the early-exit launches of finished rounds (scripts/profile_round2.sh, "
               "scripts/pmc_summary.py).  Values are the counters as reported (KB); the MI355X guide calibrates FETCH_SIZE on gfx950 as HALF "
               "the bytes of wide coalesced streaming reads and leaves other widths and WRITE_SIZE uncalibrated, so the absolute is a lower bound.",
       "source": f"profiles/{tag}_pmc_FETCH_SIZE_8pools.txt + profiles/{tag}_pmc_WRITE_SIZE_8pools.txt", "kernels": kern}
json.dump(doc, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in kern.items()}))
