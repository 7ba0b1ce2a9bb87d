"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/madnlp_hip.h declares, and its host-side symbolic analysis (integer work, no
device) equals the oracle's restatement of build_condensed_aug_symbolic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import madnlp_jl_amd as mj
from madnlp_jl_amd import _lib as L
from madnlp_jl_amd.problems import opf_shaped
from oracle.matrixtools import coo_to_csc, force_lower_triangular
from oracle.sparse_condensed import build_condensed_aug_symbolic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = mj.lib()
    assert lib.mnk_version() == 100
    header = open(os.path.join(ROOT, "include", "madnlp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mnk_[A-Za-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    raw = C.CDLL(L.LIBPATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert declared == set(L.SIGNATURES), "ctypes signature table out of sync with the header"


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/madnlp_hip.h against the ctypes table the Python mirror calls through: same number of
    arguments, pointers where the header has pointers, double / int64_t / int where it has those."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "madnlp_hip.h")).read(), flags=re.S)
    protos = re.findall(r"\b(?:int|const char\s*\*|void)\s+(mnk_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert len(protos) >= 100
    for name, args in protos:
        want = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        have = L.SIGNATURES[name][1]
        assert len(want) == len(have), f"{name}: header has {len(want)} arguments, ctypes table {len(have)}"
        for a, t in zip(want, have):
            if "*" in a:
                ok_ = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents")
            elif a.startswith("double"):
                ok_ = t is C.c_double
            elif a.startswith("int64_t"):
                ok_ = t is C.c_int64
            elif a.startswith("int "):
                ok_ = t is C.c_int
            else:
                ok_ = True
            assert ok_, f"{name}: header argument '{a}' bound as {t}"


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIBPATH", "/nonexistent/libmadnlp_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        L.lib()


def _host_only_sc(n, m, jI, jJ, hI, hJ, base=0):
    lib = mj.lib()
    h = C.c_void_p()
    jI32, jJ32 = (np.ascontiguousarray(a + base, dtype=np.int32) for a in (jI, jJ))
    hI32, hJ32 = (np.ascontiguousarray(a + base, dtype=np.int32) for a in (hI, hJ))
    L.check(lib.mnk_sc_create(None, n, m, len(jI32), jI32.ctypes.data, jJ32.ctypes.data, len(hI32),
                              hI32.ctypes.data, hJ32.ctypes.data, base, C.byref(h)), "mnk_sc_create")
    return h


def _fetch(h, n, m, nnzj, nnzh):
    lib = mj.lib()
    s = [C.c_int64() for _ in range(4)]
    L.check(lib.mnk_sc_sizes(h, *[C.byref(v) for v in s]))
    njt, nh, naug, lj = [v.value for v in s]
    out = {"sizes": (njt, nh, naug, lj)}
    for key, which, ncol, nnz in (("jt", 0, m, njt), ("h", 1, n, nh), ("aug", 2, n, naug)):
        cp = np.zeros(ncol + 1, dtype=np.int32)
        rv = np.zeros(max(nnz, 1), dtype=np.int32)
        L.check(lib.mnk_sc_get_structure(h, which, cp.ctypes.data, rv.ctypes.data))
        out[key] = (cp, rv[:nnz])
    for key, which, cnt in (("jt_map", 0, nnzj), ("h_map", 1, nnzh)):
        mp = np.zeros(max(cnt, 1), dtype=np.int64)
        L.check(lib.mnk_sc_get_map(h, which, mp.ctypes.data))
        out[key] = mp[:cnt]
    arrs = [np.zeros(max(k, 1), dtype=np.int32) for k in (n, n, nh, nh, lj, lj, lj, lj)]
    L.check(lib.mnk_sc_get_ptrs(h, *[a.ctypes.data for a in arrs]))
    out["dptr"] = (arrs[0][:n], arrs[1][:n])
    out["hptr"] = (arrs[2][:nh], arrs[3][:nh])
    out["jptr"] = tuple(a[:lj] for a in arrs[4:])
    return out


@pytest.mark.parametrize("case,base", [("case30", 0), ("case118", 1), ("case1354pegase", 1)])
def test_host_symbolic_equals_oracle(case, base):
    P = opf_shaped(case)
    h = _host_only_sc(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, base)
    got = _fetch(h, P.n, P.m, len(P.jac_I), len(P.hess_I))
    mj.lib().mnk_sc_destroy(h)
    hI, hJ = force_lower_triangular(P.hess_I, P.hess_J)
    jt, jt_map = coo_to_csc(P.n, P.m, P.jac_J, P.jac_I)
    hh, h_map = coo_to_csc(P.n, P.n, hI, hJ)
    aug, dptr, hptr, jptr = build_condensed_aug_symbolic(hh, jt)
    assert got["sizes"] == (jt.nnz, hh.nnz, aug.nnz, len(jptr[0]))
    for key, ref in (("jt", jt), ("h", hh), ("aug", aug)):
        np.testing.assert_array_equal(got[key][0], ref.colptr)
        np.testing.assert_array_equal(got[key][1], ref.rowval)
    np.testing.assert_array_equal(got["jt_map"], jt_map)
    np.testing.assert_array_equal(got["h_map"], h_map)
    for key, ref in (("dptr", dptr), ("hptr", hptr), ("jptr", jptr)):
        for a, b in zip(got[key], ref):
            np.testing.assert_array_equal(a, b)


def test_host_symbolic_edge_cases():
    # empty Jacobian / Hessian patterns, duplicate COO entries, upper-triangular Hessian input
    n, m = 5, 3
    h = _host_only_sc(n, m, np.zeros(0, int), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int))
    got = _fetch(h, n, m, 0, 0)
    mj.lib().mnk_sc_destroy(h)
    assert got["sizes"] == (0, 0, n, 0)  # only the diagonal
    np.testing.assert_array_equal(got["aug"][1], np.arange(n))
    jI = np.array([0, 0, 0, 2, 2]); jJ = np.array([4, 1, 4, 0, 0])   # duplicates (0,4) and (2,0)
    hI = np.array([0, 1, 3]); hJ = np.array([2, 1, 0])              # (0,2) is upper -> (2,0)
    h = _host_only_sc(n, m, jI, jJ, hI, hJ)
    got = _fetch(h, n, m, len(jI), len(hI))
    mj.lib().mnk_sc_destroy(h)
    assert got["jt_map"][0] == got["jt_map"][2] and got["jt_map"][3] == got["jt_map"][4]
    assert got["sizes"][0] == 3
    hI2, hJ2 = force_lower_triangular(hI, hJ)
    jt, _ = coo_to_csc(n, m, jJ, jI)
    hh, _ = coo_to_csc(n, n, hI2, hJ2)
    aug, *_ = build_condensed_aug_symbolic(hh, jt)
    np.testing.assert_array_equal(got["aug"][0], aug.colptr)
    np.testing.assert_array_equal(got["aug"][1], aug.rowval)


def test_bad_arguments_are_reported_not_crashed():
    lib = mj.lib()
    h = C.c_void_p()
    bad = np.array([7], dtype=np.int32)
    rc = lib.mnk_sc_create(None, 3, 2, 1, bad.ctypes.data, bad.ctypes.data, 0, None, None, 0, C.byref(h))
    assert rc < 0 and b"out of range" in lib.mnk_last_error_string()


def test_every_solver_option_and_statistic_is_documented():
    """Every key `mnk_ls_set_option` / `mnk_ls_get_stat` accept (csrc/ls.hip) appears in INTEGRATION.md."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "madnlp.jl_amd", "csrc", "ls.hip")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', src)))
    assert len(keys) > 30
    missing = [k for k in keys if f"`{k}`" not in doc]
    assert not missing, missing
