"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/wavehip.h declares.  No compute calls
(there is no GPU here); host-only entry points (tabulation) are exercised."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wlib():
    from wave_fenics_amd import build
    build.build()
    from wave_fenics_amd import _lib
    return _lib


def test_header_symbols_exported(wlib):
    hdr = open(os.path.join(ROOT, "include", "wavehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = wlib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libwavehip.so lacks {name}"
    assert declared == set(wlib.SIGNATURES), "ctypes SIGNATURES out of sync with include/wavehip.h"


def test_host_tabulation_matches_oracle(wlib, oracle):
    import wave_fenics_amd as w
    for p in range(1, 8):
        pts, wts, D = w.tabulate_gll(p)
        po, wo, _, Do = oracle.tabulate_1d_gll(p)
        assert np.abs(pts - po).max() <= 2e-16
        assert np.abs(wts - wo).max() <= 2e-16
        assert np.abs(D - Do).max() <= 1e-13
    for p in (1, 2, 3):
        perm, table = w.tabulate_dense(p)
        _, to = oracle.tabulate_basis_and_permutation(p)
        assert np.abs(table - to).max() <= 1e-13


def test_reorder_dofmap(wlib):
    import ctypes
    rng = np.random.default_rng(0)
    ncells, nd = 7, 27
    dm = rng.integers(0, 1000, size=(ncells, nd)).astype(np.int32)
    perm = rng.permutation(nd).astype(np.int32)
    out = np.zeros_like(dm)
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    wlib.check(wlib.lib().wf_reorder_dofmap(ncells, nd, ip(perm), ip(dm), ip(out)))
    assert np.array_equal(out, dm[:, perm])          # common/permute.hpp:19-26
    bad = perm.copy()
    bad[0] = nd
    with pytest.raises(wlib.WavehipError):
        wlib.check(wlib.lib().wf_reorder_dofmap(ncells, nd, ip(bad), ip(dm), ip(out)))


def test_unsupported_degree_is_an_error(wlib):
    import ctypes
    n = np.zeros(16)
    dp = n.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert wlib.lib().wf_tabulate_gll(9, dp, dp, None) == -2
    assert b"degree" in wlib.lib().wf_last_error()


def test_box_mesh_matches_oracle(oracle):
    import wave_fenics_amd as w
    for p, n, pert in [(2, (3, 2, 4), 0.2), (4, (2, 2, 2), 0.0)]:
        om = oracle.create_box(n, p, perturb=pert)
        m = w.create_box(n, perturb=pert)
        V = w.create_functionspace(m, p)
        assert np.array_equal(m.x, om.x) and np.array_equal(m.geom_dofmap, om.geom_dofmap)
        assert np.array_equal(V.dofmap, om.dofmap) and V.ndofs == om.ndofs


def test_cxx_wrappers_compile_and_link(wlib, tmp_path):
    """include/wavehip.hpp (the reference-named C++ classes) compiles with g++
    and links against libwavehip.so; the host-only calls run."""
    import subprocess
    exe = str(tmp_path / "wrapper_smoke")
    libdir = os.path.join(ROOT, "wave_fenics_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "wrapper_smoke.cpp"), "-o", exe,
                           "-L", libdir, "-lwavehip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    a, b = out.stdout.split()
    assert abs(float(a) - 0.1726731646460114) < 1e-15 and abs(float(b) - 13.513004977448478) < 1e-12


def test_cxx_host_drivers_build(wlib, tmp_path):
    """examples/planar3d.cpp and operator_demo.cpp (the C++ host side over the C
    ABI) compile and link; their usage errors work without a GPU."""
    import subprocess
    out = str(tmp_path / "bin")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), f"OUT={out}", "CXXFLAGS=-O1 -std=c++17 -Wall -Werror"])
    for exe in ("planar3d", "operator_demo", "tsmm_demo", "scatter_demo", "cg_demo"):
        r = subprocess.run([os.path.join(out, exe), "--bogus"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "usage" in r.stderr
