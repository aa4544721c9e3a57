"""CPU: the C-ABI library loads, exports every symbol include/kmcfield.h declares, and its host-side
planning (partition, neighbour discovery, halo lists) matches the oracle.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(km):
    lib = km.lib.load()
    hdr = open(os.path.join(ROOT, "include", "kmcfield.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(kmcf_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), "libkmcfield.so lacks %s" % n
        assert n in km.lib.SIGNATURES, "lib.py lacks a signature for %s" % n
    assert lib.kmcf_version() >= 100


def test_no_oracle_or_cpu_fallback_in_product(km):
    """The product package must not import, link or call anything under oracle/."""
    pkg = km.PKG_DIR
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "kmcf_oracle" not in txt and "orc_" not in txt, f
    out = os.popen("ldd %s" % km.lib.LIB_PATH).read()
    assert "oracle" not in out


def test_error_convention(km):
    lib = km.lib.load()
    h = C.c_void_p()
    assert lib.kmcf_comm_create(C.byref(h), -1, 2, 5) == -1          # KMCF_ERR_ARG, no exit()
    assert b"rank" in lib.kmcf_last_error()
    with pytest.raises(km.lib.KmcfError):
        km.lib.check(-1, "x")


def test_partition_matches_reference_rule(km, oracle):
    for n, P in ((36498, 8), (1597080, 8), (5, 8), (7, 1), (0, 3)):
        c, d = km.solvers.KMC_comm.partition(n, P)
        co, do = oracle.partition(n, P)
        assert np.array_equal(c, co) and np.array_equal(d, do)


def _host_comm(km, P, r):
    lib = km.lib.load()
    h = C.c_void_p()
    km.lib.check(lib.kmcf_comm_create(C.byref(h), -1, P, r), "comm")   # device -1: host-only planning
    return h


class _Comm:
    def __init__(self, handle):
        self.handle = handle


@pytest.mark.parametrize("P", [1, 2, 4, 8])
def test_halo_planning_matches_oracle(km, oracle, ref5, P):
    """Distributed_matrix ctor 1 (dist_matrix.cpp:5-69): neighbours, cols/rows_per_neighbour."""
    S = km.solvers
    ks = ref5["ks"]
    counts, displs = oracle.partition(ks.n, P)
    lib = km.lib.load()
    for r in range(P):
        h = _host_comm(km, P, r)
        r0, nr = int(displs[r]), int(counts[r])
        rp = (ks.row_ptr[r0:r0 + nr + 1] - ks.row_ptr[r0]).astype(np.int32)
        col = ks.col[ks.row_ptr[r0]:ks.row_ptr[r0 + nr]]
        m = S.Distributed_matrix(_Comm(h), ks.n, counts, displs, col, rp, None)
        want = oracle.halo_lists(ks.row_ptr, ks.col, P, r)
        got = m.neighbours()
        assert [g["rank"] for g in got] == [w["rank"] for w in want]
        for g, w in zip(got, want):
            assert g["nnz"] == w["nnz"]
            assert np.array_equal(g["cols"], w["cols"]) and np.array_equal(g["rows"], w["rows"])
        info = m.info()
        assert info["rows_this_rank"] == nr and info["nnz"] == len(col)
        assert info["halo_cols"] == sum(len(w["cols"]) for w in want[1:])
        # compute entry points refuse a host-only matrix instead of falling back
        assert lib.kmcf_spmv(m.handle, None, None) != 0
        m.close()
        lib.kmcf_comm_destroy(h)


def test_matrix_build_rejects_bad_input(km):
    S = km.solvers
    h = _host_comm(km, 1, 0)
    with pytest.raises(km.lib.KmcfError):
        S.Distributed_matrix(_Comm(h), 3, [3], [0], [0, 5], [0, 1, 2, 2], None)      # column out of range
    with pytest.raises(km.lib.KmcfError):
        S.Distributed_matrix(_Comm(h), 3, [2], [0], [0], [0, 1, 1], None)            # counts do not sum to n
    # empty rows / empty rank are fine (ragged input)
    m = S.Distributed_matrix(_Comm(h), 3, [3], [0], [1], [0, 0, 1, 1], None)
    assert m.info()["nnz"] == 1
    m.close()
    km.lib.load().kmcf_comm_destroy(h)


def test_synthetic_crossbar_generator_is_deterministic(km):
    a = km.structure.synth_small(tiles=1)
    b = km.structure.synth_small(tiles=1)
    assert np.array_equal(a["xyz"], b["xyz"]) and np.array_equal(a["element"], b["element"])
    assert a["N"] == 37650 and a["N_contact"] == 576
    assert int((a["element"] == 2).sum()) == 400                 # 5 % of 8000 O
    xx = a["xyz"][:, 0]
    assert np.all(xx[:576] == xx.min()) and np.all(xx[-576:] == xx.max())
    mid_y = a["xyz"][576:-576, 1]
    assert np.all(np.diff(mid_y) >= 0)                           # "bwmin": sorted along y


@pytest.mark.parametrize("long_row", [384, 45])
def test_internal_row_order_is_refined_for_the_row_per_lane_layout(km, oracle, ref5, monkeypatch, long_row):
    """kmcf_matrix_build on the host (no GPU): the internal row order is a permutation, rows beyond KMCF_LONG_ROW come
    last, and inside every tile of the row-per-lane SpMV layout (<= 256 rows, <= 767 columns outside the tile) the
    rows are dealt to wavefronts by length (the 64 longest first, ...), original order inside a wavefront."""
    S = km.solvers
    monkeypatch.setenv("KMCF_LONG_ROW", str(long_row))
    ks = ref5["ks"]
    h = _host_comm(km, 1, 0)
    m = S.Distributed_matrix(_Comm(h), ks.n, [ks.n], [0], ks.col, ks.row_ptr, None)
    perm, n_short, ends = m.row_order()
    n = ks.n
    assert sorted(perm.tolist()) == list(range(n))
    length = np.diff(ks.row_ptr)
    is_long = length[perm] > long_row
    assert n_short == n - int(is_long.sum()) and not is_long[:n_short].any() and is_long[n_short:].all()
    assert len(ends) > 0 and ends[-1] == n_short and np.all(np.diff(np.r_[0, ends]) > 0)
    rp, col = ks.row_ptr, ks.col
    offdiag = np.array([np.count_nonzero(col[rp[r]:rp[r + 1]] != r) for r in range(n)])
    start = 0
    for e in ends:
        rows = perm[start:e]
        assert len(rows) <= 256
        inside = set(rows.tolist())
        outside = {int(c) for r in rows for c in col[rp[r]:rp[r + 1]] if int(c) not in inside}
        assert len(outside) <= 767
        lens = offdiag[rows]
        for w in range(0, len(rows), 64):                       # wave w holds no row shorter than any row of wave w + 1
            if w + 64 < len(rows):
                assert lens[w:w + 64].min() >= lens[w + 64:w + 128].max()
        start = int(e)
    # the halo protocol and the exported pattern are in the caller's order, whatever the internal one
    assert m.info()["rows_this_rank"] == n
    m.close()
    km.lib.load().kmcf_comm_destroy(h)
