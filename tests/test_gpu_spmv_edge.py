"""GPU: edge shapes through every SpMV kernel (forced with KMCF_SPMV_KIND): tiny matrices, empty rows, rows
without a diagonal entry, duplicate entries, a dictionary at and beyond its size limit, fewer tiles than the
kernels' 8-block minimum grid."""
import os
import numpy as np
import pytest

CODED_ON = int(os.environ.get("KMCF_SPMV_CODED", "1") != "0")   # the suite is green under KMCF_SPMV_CODED=0 too

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _csr(rows):
    """rows: list of lists of (col, val) -> indptr, indices, data (entries kept in the given order)."""
    indptr = [0]
    idx, dat = [], []
    for r in rows:
        for c, v in r:
            idx.append(c)
            dat.append(v)
        indptr.append(len(idx))
    return np.array(indptr, np.int32), np.array(idx, np.int32), np.array(dat, np.float64)


def _dense_apply(indptr, indices, data, x):
    y = np.zeros(len(indptr) - 1)
    for i in range(len(y)):
        for j in range(indptr[i], indptr[i + 1]):
            y[i] += data[j] * x[indices[j]]
    return y


CASES = {
    "one_row": [[(0, 2.5)]],
    "diag_only": [[(i, 1.0 + i)] for i in range(7)],
    "empty_rows": [[(0, 2.0), (1, -1.0)], [], [(1, -1.0), (2, 3.0)], [], [(4, 1.0)]],
    "no_diagonal": [[(1, -1.0), (2, -0.5)], [(0, -1.0), (1, 4.0)], [(0, -0.5)]],
    "duplicates": [[(0, 2.0), (1, -1.0), (1, -1.0)], [(0, -1.0), (1, 2.0), (0, -1.0)]],
}


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("name", sorted(CASES))
def test_tiny_matrices(km, torch, monkeypatch, name, kind):
    S = km.solvers
    monkeypatch.setenv("KMCF_SPMV_KIND", str(kind))
    indptr, indices, data = _csr(CASES[name])
    n = len(indptr) - 1
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    mat = S.Distributed_matrix(comm, n, [n], [0], indices, indptr, data)
    info = mat.info()
    assert info["spmv_kind"] == kind, info
    if kind == 2:
        assert (info["spmv_coded"] > 0) == bool(CODED_ON)     # few distinct off-diagonal values in every case
    x = np.linspace(1.0, 2.0, n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), _dense_apply(indptr, indices, data, x), rtol=1e-14, atol=1e-14)
    np.testing.assert_array_equal(mat.get_values(), data)
    mat.close()
    comm.close()


@pytest.mark.parametrize("ndistinct,coded", [(1, 1), (62, 1), (63, 0)])
def test_dictionary_size_limit(km, torch, monkeypatch, ndistinct, coded):
    """62 distinct off-diagonal values are coded, 63 are not (codes 0..61; 63 marks the diagonal)."""
    import scipy.sparse as sp
    S = km.solvers
    monkeypatch.setenv("KMCF_SPMV_KIND", "2")
    rng = np.random.default_rng(5)
    n = 3000
    vals = -(1.0 + np.arange(ndistinct)) / 64.0
    diags = [vals[(np.arange(n - o) * 5 + o) % ndistinct] for o in (1, 2, 3, 30)]     # gcd(5, 62) = gcd(5, 63) = 1
    M = sp.diags(diags, [1, 2, 3, 30], shape=(n, n), format="csr")
    M = (M + M.T + sp.diags(10.0 + rng.random(n))).tocsr()
    M.sort_indices()
    assert len(np.unique(M.data[M.data < 0])) == ndistinct
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    info = mat.info()
    # (one value: row-per-lane kernel, 2; 62 values: coded window kernel, 1; 63: values streamed)
    assert info["spmv_kind"] == 2 and (info["spmv_coded"] > 0) == bool(coded & CODED_ON), info
    if CODED_ON and coded and os.environ.get("KMCF_SPMV_SELL", "1") != "0":
        assert info["spmv_coded"] == (2 if ndistinct <= 3 else 1), info
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.empty_like(p)
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), M @ x, rtol=1e-13, atol=1e-13)
    mat.close()
    comm.close()


@pytest.mark.parametrize("sort", [1, 0])
@pytest.mark.parametrize("nvals", [1, 2, 3])
def test_row_per_lane_kernel_ragged_rows(km, torch, monkeypatch, sort, nvals):
    """spmv_sell_kernel on rows of 0 ... 64 off-diagonal entries (empty rows, rows without a diagonal, a tile cut by
    its window limit), 1-3 dictionary values, with the rows sorted into the internal order (lane t owns row r0 + t)
    and with KMCF_SPMV_SELL_SORT=0 (lane rows read from a table); bit-identical to each other and to the plain
    window kernel's products added in column order."""
    import scipy.sparse as sp
    if not CODED_ON or os.environ.get("KMCF_SPMV_SELL", "1") == "0":
        pytest.skip("row-per-lane kernel switched off")
    S = km.solvers
    monkeypatch.setenv("KMCF_SPMV_KIND", "2")
    monkeypatch.setenv("KMCF_SPMV_SELL_SORT", str(sort))
    rng = np.random.default_rng(17)
    n = 6000
    vals = np.array([-1.0, -0.125, -3.0])[:nvals]
    rows, cols, data = [], [], []
    for i in range(n):
        k = int(rng.integers(0, 65)) if i % 7 else 0                 # every 7th row: diagonal only (or nothing)
        lo, hi = max(0, i - 400), min(n, i + 400)
        if i % 500 < 40:                                             # stretches with far-away columns: window-limited tiles
            cand = rng.choice(n, size=min(k, n - 1), replace=False)
        else:
            cand = rng.choice(np.arange(lo, hi), size=min(k, hi - lo - 1), replace=False)
        cand = cand[cand != i]
        rows += [i] * len(cand)
        cols += list(cand)
        data += list(rng.choice(vals, len(cand)))
        if i % 11:                                                   # some rows have no diagonal entry
            rows.append(i); cols.append(i); data.append(5.0 + rng.random())
    M = sp.csr_matrix((data, (rows, cols)), shape=(n, n))
    M.sum_duplicates()
    M.sort_indices()
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    info = mat.info()
    assert info["spmv_kind"] == 2 and info["spmv_coded"] == 2, info
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.full((n,), 3.0, dtype=torch.float64, device="cuda")
    mat.spmv(p, Ap)
    y = Ap.cpu().numpy()
    want = M @ x
    bound = abs(M) @ np.abs(x)
    assert np.all(np.abs(y - want) <= 1e-14 * bound + 1e-300)
    np.testing.assert_array_equal(mat.get_values(), M.data)
    # the dictionary order of accumulation: off-diagonals in column order, the diagonal last
    ref = np.zeros(n)
    for i in range(n):
        s = 0.0
        dg = 0.0
        for j in range(M.indptr[i], M.indptr[i + 1]):
            if M.indices[j] == i:
                dg = M.data[j]
            else:
                s += M.data[j] * x[M.indices[j]]
        ref[i] = s + dg * x[i]
    short = np.diff(M.indptr) <= int(os.environ.get("KMCF_LONG_ROW", "384"))     # (longer rows: the chunked long-row kernel)
    np.testing.assert_array_equal(y[short], ref[short])
    mat.close()
    comm.close()
