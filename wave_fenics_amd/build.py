"""Build libwavehip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwavehip.so")
SOURCES = ["tables.cpp", "mesh_io.cpp", "generic_plan.cpp", "function_space.cpp", "markers.cpp", "kernels.hip", "stiffness_march_idx.hip", "stiffness_march.hip", "stiffness_march_ks.hip", "mass_march.hip", "stiffness_dense.hip", "tsmm.hip", "vector_kernels.hip", "comm.hip", "cg.hip", "api.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "stiffness_core.h"),
           os.path.join(ROOT, "include", "wavehip.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    deps = [path] + HEADERS
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", path, "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1 if verbose else min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), SOURCES))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
