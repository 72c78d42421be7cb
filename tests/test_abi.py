"""The C-ABI shared library loads and exports every symbol include/cookmatch.h declares (no compute without a GPU);
on a machine without a GPU engine creation must fail loudly instead of falling back to the CPU."""
import ctypes as C
import os
import re

import pytest

from cook_amd import _abi as A
from cook_amd import build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cookmatch.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|void\*?|const char\*)\s+(cook_\w+)\s*\(", txt, flags=re.M)))


def test_header_symbols_exported():
    so = build.build()
    lib = C.CDLL(so)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cookmatch.h but not exported by libcookmatch.so"
    assert sorted(engine.EXPORTS) == syms
    # ... and NOTHING else: the library is built with -fvisibility=hidden, the header's declarations carry default visibility
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TW")
    assert exported == syms, sorted(set(exported) ^ set(syms))[:10]
    # the binding checks the struct-layout version before its first call (cook_amd/engine.py load_library)
    hdr = open(os.path.join(ROOT, "include", "cookmatch.h")).read()
    assert int(re.search(r"#define COOK_ABI_VERSION (\d+)", hdr).group(1)) == A.ABI_VERSION == lib.cook_abi_version()


def test_struct_sizes_match_header(tmp_path):
    # compile a tiny C program against the header and compare sizeof() with the ctypes mirrors
    src = tmp_path / "sz.c"
    names = ["cook_params", "cook_usage", "cook_tasks", "cook_users", "cook_pool_quota", "cook_jobs", "cook_offers",
             "cook_groups", "cook_rebalance_params", "cook_host_spare", "cook_preemption", "cook_queue", "cook_user_state",
             "cook_nodes", "cook_pods", "cook_offer_params", "cook_node_offers", "cook_offer_totals", "cook_resource_stats",
             "cook_cycle_metrics", "cook_cycle_delta"]
    src.write_text('#include <stdio.h>\n#include "cookmatch.h"\nint main(){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [A.CookParams, A.CookUsage, A.CookTasks, A.CookUsers, A.CookPoolQuota, A.CookJobs, A.CookOffers,
               A.CookGroups, A.CookRebalanceParams, A.CookHostSpare, A.CookPreemption, A.CookQueue, A.CookUserState,
               A.CookNodes, A.CookPods, A.CookOfferParams, A.CookNodeOffers, A.CookOfferTotals, A.CookResourceStats,
               A.CookCycleMetrics, A.CookCycleDelta]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    build.build()
    with pytest.raises(engine.CookError):
        engine.Engine(A.default_params())


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cook_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "libcookoracle" not in txt and "cook_oracle" not in txt and "k8s_offers" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f  # no module of oracle/ at all


def test_jni_shim_typechecks_against_header():
    # bindings/jni/cookmatch_jni.c calls every C-ABI function with the header's exact signatures (no JDK here: stub jni.h)
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "jni", "cookmatch_jni.c")])


def test_ctypes_prototypes_are_generated_from_the_header():
    # cook_amd/_protos.py (restype / argtypes of every export) is in sync with include/cookmatch.h
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "gen_protos.py"), "--check"]) == 0, \
        "run python scripts/gen_protos.py after editing include/cookmatch.h"
    lib = engine.load_library(build.build())
    for name, (restype, argtypes) in engine.PROTOS.items():
        fn = getattr(lib, name)
        assert fn.restype == restype and list(fn.argtypes) == list(argtypes), name
    with pytest.raises(C.ArgumentError):
        lib.cook_match_count(C.c_void_p(0), 1.5)  # a float where a pointer belongs is refused before the call
    with pytest.raises(TypeError):
        lib.cook_cycle_run(C.c_void_p(0))  # wrong argument count


def test_jni_shim_binds_the_hot_path_entry_points():
    # every entry point of the rank -> considerable -> match -> rebalance path has a JNI export; buffers are size-checked
    txt = open(os.path.join(ROOT, "bindings", "jni", "cookmatch_jni.c")).read()
    for fn in ("cook_rank", "cook_rank_user_usage", "cook_rank_pool_usage_multi", "cook_considerable", "cook_match", "cook_match_count", "cook_cycle_stage",
               "cook_cycle_update", "cook_cycle_run", "cook_cycle_run_rank", "cook_cycle_run_rank_multi", "cook_cycle_match_multi", "cook_cycle_fetch",
               "cook_rebalance", "cook_offers_build", "cook_match_explain", "cook_match_metrics", "cook_host_alloc", "cook_host_free"):
        assert re.search(r"\b%s\(" % fn, txt), fn
    assert "GetDirectBufferCapacity" in txt and "DeleteLocalRef" in txt
    assert not re.search(r"GetObjectArrayElement\([^;]*;\s*\n\s*return", txt)  # no element reference is leaked


def test_prefetch_sink_register_is_left_alone():
    """The one place where the library relies on a register allocation (gpu_prims.hpp PREFETCH_WORD: fire-and-forget loads from inline asm
    into a "sink" register that has to stay put until the drain): checked in the gfx950 assembly of every kernel, in the shipped build and
    in the measurement build whose ancestor once raised an unexplained GPU memory fault (scripts/check_prefetch_sink.py)."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for flags in ([], ["-DCOOK_WALK_PROF=1"]):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_prefetch_sink.py")] + flags, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        assert "0 problems" in r.stdout and not r.stdout.startswith("0 asm-load")
