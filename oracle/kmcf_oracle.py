"""ctypes front end of the CPU oracle (oracle/kmcf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VACANCY, OXYGEN_DEFECT = 2, 1

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libkmcf_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libkmcf_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_partition.argtypes = [C.c_int, C.c_int, _ip, _ip]
        L.orc_pattern_brute.restype = C.c_int64
        L.orc_pattern_cells.restype = C.c_int64
        for f in (L.orc_pattern_brute, L.orc_pattern_cells):
            f.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                          _ip, C.c_void_p]
        L.orc_neighbor_list.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, _ip]
        L.orc_update_charge.argtypes = [_ip, _ip, _ip, C.c_int, _ip, C.c_int, C.c_int, C.c_int]
        L.orc_assemble_K.argtypes = [_ip, _ip, _ip, C.c_int, C.c_double, C.c_double, C.c_double,
                                     C.c_int, C.c_int, _ip, _ip, _dp,
                                     C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip,
                                     _ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp]
        L.orc_spmv.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.orc_spmv_omp.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.orc_pcg_jacobi.restype = C.c_int
        L.orc_pcg_jacobi.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                     C.c_int, _ip, _ip, C.POINTER(C.c_double), C.c_void_p]
        L.orc_pcg_jacobi_omp.restype = C.c_int
        L.orc_pcg_jacobi_omp.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                         C.POINTER(C.c_double)]
        L.orc_omp_threads.restype = C.c_int
        L.orc_halo_lists.restype = C.c_int
        L.orc_halo_lists.argtypes = [_ip, _ip, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, C.c_void_p, C.c_void_p]
        L.orc_update_temperature_global.restype = C.c_double
        L.orc_update_temperature_global.argtypes = [_dp, C.c_int, C.c_double, C.c_double, C.c_double,
                                                    C.c_double, C.c_double, C.c_double]
        L.orc_sum_AB_into_A.argtypes = [_dp, _dp, C.c_int]
        L.orc_poisson_gridless.argtypes = [_dp, _dp, _dp, C.c_int, _ip, C.c_double, C.c_double, C.c_double,
                                           C.c_int, C.c_int, _dp]
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def partition(n, P):
    counts = np.zeros(P, np.int32)
    displs = np.zeros(P, np.int32)
    lib().orc_partition(n, P, counts, displs)
    return counts, displs


def pattern(x, y, z, lattice, pbc, cutoff, size_i, size_j, start_i, start_j, brute=False):
    """CSR pattern (row_ptr, col) of one row-block x column-block (block-local columns)."""
    x, y, z, lattice = _f(x), _f(y), _f(z), _f(lattice)
    fn = lib().orc_pattern_brute if brute else lib().orc_pattern_cells
    row_ptr = np.zeros(size_i + 1, np.int32)
    nnz = fn(x, y, z, lattice, int(pbc), float(cutoff), size_i, size_j, start_i, start_j, row_ptr, None)
    col = np.zeros(max(int(nnz), 1), np.int32)
    fn(x, y, z, lattice, int(pbc), float(cutoff), size_i, size_j, start_i, start_j, row_ptr,
       col.ctypes.data_as(C.c_void_p))
    return row_ptr, col[:nnz]


def neighbor_list(x, y, z, nn_dist=3.5, nn=52, count=None, displ=0):
    x, y, z = _f(x), _f(y), _f(z)
    N = len(x)
    count = N if count is None else count
    out = np.empty(count * nn, np.int32)
    lib().orc_neighbor_list(x, y, z, N, float(nn_dist), nn, count, displ, out)
    return out.reshape(count, nn)


def update_charge(element, charge, neigh_idx, metals, row_start=0, row_end=None):
    element, metals = _i(element), _i(metals)
    charge = _i(charge).copy()
    nn = neigh_idx.shape[1]
    row_end = len(element) if row_end is None else row_end
    lib().orc_update_charge(element, charge, _i(neigh_idx).reshape(-1), nn, metals, len(metals), row_start, row_end)
    return charge


class KSystem:
    """Global interface pattern + contact patterns of a device (a3)."""

    def __init__(self, xyz, lattice, pbc, nn_dist, N_left, N_right):
        x, y, z = _f(xyz[:, 0]), _f(xyz[:, 1]), _f(xyz[:, 2])
        self.N = len(x)
        self.N_left, self.N_right = int(N_left), int(N_right)
        self.n = self.N - self.N_left - self.N_right
        n = self.n
        self.row_ptr, self.col = pattern(x, y, z, lattice, pbc, nn_dist, n, n, N_left, N_left)
        self.left_row_ptr, self.left_col = pattern(x, y, z, lattice, pbc, nn_dist, n, N_left, N_left, 0)
        self.right_row_ptr, self.right_col = pattern(x, y, z, lattice, pbc, nn_dist, n, N_right, N_left, N_left + n)
        self.nnz = len(self.col)


def assemble_K(ks, element, charge, metals, high_G, low_G, Vd, P=1):
    """Values + diag/dinv/rhs for all rows, rank by rank with the reference's
    per-block diagonal summation order.  Returns dict of global arrays."""
    n = ks.n
    counts, displs = partition(n, P)
    val = np.zeros(ks.nnz, np.float64)
    diag = np.zeros(n)
    dinv = np.zeros(n)
    rhs = np.zeros(n)
    left = np.zeros(n)
    right = np.zeros(n)
    element, charge, metals = _i(element), _i(charge), _i(metals)
    for r in range(P):
        r0, nr = int(displs[r]), int(counts[r])
        lrp = _i(ks.left_row_ptr[r0:r0 + nr + 1] - ks.left_row_ptr[r0])
        lc = _i(ks.left_col[ks.left_row_ptr[r0]:ks.left_row_ptr[r0 + nr]])
        rrp = _i(ks.right_row_ptr[r0:r0 + nr + 1] - ks.right_row_ptr[r0])
        rc = _i(ks.right_col[ks.right_row_ptr[r0]:ks.right_row_ptr[r0 + nr]])
        if len(lc) == 0:
            lc = np.zeros(1, np.int32)
        if len(rc) == 0:
            rc = np.zeros(1, np.int32)
        d, di, rh, le, ri = (np.zeros(max(nr, 1)) for _ in range(5))
        lib().orc_assemble_K(element, charge, metals, len(metals), high_G, low_G, Vd,
                             ks.N_left, n, ks.row_ptr, ks.col, val,
                             r0, nr, P, r, counts, displs, lrp, lc, rrp, rc, d, di, rh, le, ri)
        diag[r0:r0 + nr], dinv[r0:r0 + nr], rhs[r0:r0 + nr] = d[:nr], di[:nr], rh[:nr]
        left[r0:r0 + nr], right[r0:r0 + nr] = le[:nr], ri[:nr]
    return dict(val=val, diag=diag, dinv=dinv, rhs=rhs, left=left, right=right)


def update_CB_edge(ks, element, metals, high_G, low_G, Vd, x0=None, tol=1e-14, max_it=50000):
    """update_CB_edge_gpu_sparse (src/potential_solver_gpu.cu:673-772): returns (site_CB_edge[N] in J,
    iterations, scaled A values, scaled rhs)."""
    L = lib()
    L.orc_assemble_CB.argtypes = [_ip, _ip, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                  _ip, _ip, _dp, _ip, _ip, _ip, _ip, _dp]
    L.orc_solve_sparse_CG_Jacobi.restype = C.c_int
    L.orc_solve_sparse_CG_Jacobi.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_double, C.c_int]
    n = ks.n
    val = np.zeros(ks.nnz)
    rhs = np.zeros(n)
    lc = _i(ks.left_col) if len(ks.left_col) else np.zeros(1, np.int32)
    rc = _i(ks.right_col) if len(ks.right_col) else np.zeros(1, np.int32)
    L.orc_assemble_CB(_i(element), _i(metals), len(metals), high_G, low_G, Vd, ks.N_left, n,
                      ks.row_ptr, ks.col, val, _i(ks.left_row_ptr), lc, _i(ks.right_row_ptr), rc, rhs)
    val_unscaled, rhs_unscaled = val.copy(), rhs.copy()
    y = np.zeros(n) if x0 is None else _f(x0).copy()
    it = L.orc_solve_sparse_CG_Jacobi(n, ks.row_ptr, ks.col, val, rhs, y, tol, max_it)
    eV_to_J = 1.60217663e-19                                   # potential_solver_gpu.cu:6
    out = np.empty(ks.N)
    out[:ks.N_left] = Vd / 2                                    # :746-749
    out[ks.N_left:ks.N_left + n] = y
    out[ks.N_left + n:] = -Vd / 2
    return out * eV_to_J, it, dict(val=val_unscaled, rhs=rhs_unscaled, val_scaled=val, rhs_scaled=rhs)


def spmv(row_ptr, col, val, x, omp=False):
    n = len(row_ptr) - 1
    y = np.zeros(n)
    (lib().orc_spmv_omp if omp else lib().orc_spmv)(n, _i(row_ptr), _i(col), _f(val), _f(x), y)
    return y


def pcg_jacobi(row_ptr, col, val, rhs, x0, dinv, tol, max_it, P=1, fixed_iters=0, history=False):
    """Returns (x, iters, relres[, rz_hist]); rhs/x0 are not modified."""
    n = len(row_ptr) - 1
    counts, displs = partition(n, P)
    r = _f(rhs).copy()
    x = _f(x0).copy()
    rel = C.c_double(0.0)
    hist = np.zeros(max_it + 2 + fixed_iters) if history else None
    it = lib().orc_pcg_jacobi(n, _i(row_ptr), _i(col), _f(val), r, x, _f(dinv), float(tol), int(max_it),
                              int(fixed_iters), P, counts, displs, C.byref(rel),
                              hist.ctypes.data_as(C.c_void_p) if history else None)
    if history:
        return x, it, rel.value, hist[:it + 1]
    return x, it, rel.value


def pcg_device_order(plan, rhs, x0, dinv, tol, max_it, fixed_iters=0, check_mode=1, val=None, history=False, variant=None):
    """The reference's (P)CG added in the device's order (kmcf_oracle_order.c).  `plan`: what the library exports
    with kmcf_matrix_sum_plan / kmcf_matrix_row_order (a dict: vec_grid, sell_grid, tile_first, tile_rows, row_ptr,
    col, val in the internal row order, perm = caller row of every internal row).  rhs / x0 / dinv in the CALLER's
    row order (dinv None: unpreconditioned).  Returns dict(x, r, iterations, bb, rz, relres, converged[, rz_hist])."""
    L = lib()
    if not hasattr(L, "_order_ready"):
        L.orc_pcg_device_order.restype = C.c_int
        L.orc_pcg_device_order.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.orc_pcg1_device_order.restype = C.c_int
        L.orc_pcg1_device_order.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.orc_spmv_device_order.argtypes = [C.c_int, _ip, _ip, C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.orc_pcg_resident_order.restype = C.c_int
        L.orc_pcg_resident_order.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.orc_pcg1_resident_order.restype = C.c_int
        L.orc_pcg1_resident_order.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L._order_ready = True
    assert plan["sell_active"] and plan["sell_ident"] and plan["n_short"] == plan["rows"] and not plan["halo_cols"] \
        and not plan["sub_grid"], "device-order oracle: single-rank matrix computed by the row-per-lane kernel only"
    perm = _i(plan["perm"])
    n = len(perm)
    r = _f(rhs)[perm].copy()
    x = _f(x0)[perm].copy()
    dv = _f(dinv)[perm].copy() if dinv is not None else np.ones(n)
    bb, rz, done = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
    hist = np.zeros(max(int(max_it), int(fixed_iters)) + 2) if history else None
    # variant: the recurrence the library runs on this matrix (plan["cg_variant"]: 0 classic = the reference's
    # operation order, 1 single-reduction) unless the caller names one
    # ("cg1r-loop" / "classic-loop": the recurrence as the loop of kernels even where the plan names a resident launch)
    cg1r = plan["cg_variant"] == 1 if variant is None else variant in ("cg1r", "cg1r-loop")
    common = (n, _i(plan["row_ptr"]), _i(plan["col"]), _f(plan["val"] if val is None else val), r, x, dv,
              1 if dinv is not None else 0, float(tol), int(max_it), int(fixed_iters))
    tail = (int(plan["vec_grid"]), int(plan["sell_grid"]), len(plan["tile_first"]), _i(plan["tile_first"]),
            _i(plan["tile_rows"]), C.byref(bb), C.byref(rz), C.byref(done), hist.ctypes.data_as(C.c_void_p) if history else None)
    resident = plan.get("resident_tpb", 0) > 0 and variant not in ("cg1r-loop", "classic-loop")
    if resident and not cg1r:
        # the reference's recurrence as ONE register-resident launch: every dot over the resident tree
        assert check_mode == 1
        it = L.orc_pcg_resident_order(*common, int(plan["resident_tpb"]), int(plan["resident_g1"]), *tail[2:])
    elif cg1r and resident:
        # the single-reduction recurrence as ONE register-resident launch (csrc/kmcf_cgr.hip): its own reduction tree
        assert check_mode == 1
        it = L.orc_pcg1_resident_order(*common, int(plan["resident_tpb"]), int(plan["resident_g1"]), *tail[2:])
    elif cg1r:
        assert check_mode == 1
        it = L.orc_pcg1_device_order(*common, *tail)
    else:
        it = L.orc_pcg_device_order(*common, int(check_mode), *tail)
    xo, ro = np.empty(n), np.empty(n)
    xo[perm] = x
    ro[perm] = r
    out = dict(x=xo, r=ro, iterations=it, bb=bb.value, rz=rz.value, converged=bool(done.value),
               relres=float(np.sqrt(rz.value / bb.value)) if bb.value > 0 else 0.0)
    if history:
        out["rz_hist"] = hist[:it + 1]
    return out


def spmv_device_order(plan, x_user):
    """y = A x with every row added in the row-per-lane kernel's order; caller's row order in and out."""
    pcg_device_order.__doc__  # (argtypes are set up there)
    L = lib()
    if not hasattr(L, "_order_ready"):
        L.orc_spmv_device_order.argtypes = [C.c_int, _ip, _ip, C.c_int, _ip, _ip, _dp, _dp, _dp]
    perm = _i(plan["perm"])
    n = len(perm)
    x = _f(x_user)[perm].copy()
    y = np.zeros(n)
    L.orc_spmv_device_order(len(plan["tile_first"]), _i(plan["tile_first"]), _i(plan["tile_rows"]), int(plan["sell_grid"]),
                            _i(plan["row_ptr"]), _i(plan["col"]), _f(plan["val"]), x, y)
    out = np.empty(n)
    out[perm] = y
    return out


class _DevPlan(C.Structure):
    """orc_dev_plan (kmcf_oracle_order.c)"""
    _ipt, _dpt = C.POINTER(C.c_int), C.POINTER(C.c_double)
    _fields_ = [("n", C.c_int), ("n_short", C.c_int), ("n_halo", C.c_int), ("rp", _ipt), ("col", _ipt), ("val", _dpt),
                ("vec_grid", C.c_int), ("sell_grid", C.c_int), ("n_tiles", C.c_int), ("tile_first", _ipt), ("tile_rows", _ipt),
                ("boundary_grid", C.c_int), ("boundary_lpr", C.c_int), ("n_boundary", C.c_int), ("boundary_rows", _ipt),
                ("n_long_items", C.c_int), ("long_items", _ipt), ("sub_grid", C.c_int), ("sub_n", C.c_int), ("sub_rows", _ipt),
                ("sub_rp", _ipt), ("sub_col", _ipt), ("sub_val", _dpt), ("sub_dense", C.c_int), ("sub_n_glob", C.c_int),
                ("sub_strip", C.c_int), ("sub_full", _dpt), ("sub_row0", C.c_int), ("sub_total", _dpt)]


class DeviceRank:
    """One rank of a solve in the device's summation order: the library's exported plan (kmcf_matrix_sum_plan: a dict)
    turned into the arrays kmcf_oracle_order.c walks.  sub: the tunnel sub-block of the T operator on this rank,
    dict(grid, rows = caller LOCAL row of every local sub row, row_ptr, col, val over the gathered sub-vector)."""
    LONG_CHUNK = 2048                        # KMCF_LONG_CHUNK (csrc/kmcf_internal.hpp)

    def __init__(self, plan, sub=None):
        assert plan["sell_active"] and plan["sell_ident"], "device-order oracle: short rows must be computed by the row-per-lane kernel"
        self.plan = plan
        self.n, self.n_halo, self.n_short = int(plan["rows"]), int(plan["halo_cols"]), int(plan["n_short"])
        self.perm = _i(plan["perm"])
        self.inv = np.empty(self.n, np.int32)
        self.inv[self.perm] = np.arange(self.n, dtype=np.int32)
        self.rp, self.col, self.val = _i(plan["row_ptr"]), _i(plan["col"]), _f(plan["val"])
        self.halo_gid = _i(plan.get("halo_gid", np.zeros(0, np.int32)))
        isb = np.zeros(self.n, bool)
        rows_of = np.repeat(np.arange(self.n), np.diff(self.rp))
        isb[rows_of[self.col >= self.n]] = True
        self.boundary = _i(np.nonzero(isb[:self.n_short])[0])
        items = []
        for i in range(self.n_short, self.n):
            first = len(items)
            for j in range(int(self.rp[i]), int(self.rp[i + 1]), self.LONG_CHUNK):
                items.append((i, j, min(j + self.LONG_CHUNK, int(self.rp[i + 1])), first))
        self.long_items = _i(np.array(items, np.int32).reshape(-1)) if items else np.zeros(4, np.int32)
        self.tile_first, self.tile_rows = _i(plan["tile_first"]), _i(plan["tile_rows"])
        ipt, dpt = C.POINTER(C.c_int), C.POINTER(C.c_double)
        pi = lambda a: a.ctypes.data_as(ipt)
        pd = lambda a: a.ctypes.data_as(dpt)
        self.sub = sub
        if sub is not None and len(sub["rows"]):
            self.sub_rows = _i(self.inv[_i(sub["rows"])])
            self.sub_rp, self.sub_col, self.sub_val = _i(sub["row_ptr"]), _i(sub["col"]), _f(sub["val"])
            sub_n, sub_grid = len(self.sub_rows), int(sub["grid"])
        else:
            self.sub_rows = self.sub_rp = self.sub_col = np.zeros(1, np.int32)
            self.sub_val = np.zeros(1)
            sub_n = sub_grid = 0
        # dense symmetric storage of the block (one rank): the full block as a dense array for the tile walk
        self.sub_full = np.zeros(1)
        dense = int(bool(sub is not None and sub.get("dense")))
        self.spread = bool(dense and sub.get("spread"))        # the tiles' strips dealt to the ranks of a group (pcg_device_order_ranks)
        self.sub_strip = int(sub.get("strip", 16)) if dense else 16
        if self.spread:
            dense = 2
        elif dense:
            nt = len(self.sub_rows)
            full = np.zeros((nt, nt))
            rows_of = np.repeat(np.arange(nt), np.diff(self.sub_rp))
            full[rows_of, self.sub_col] = self.sub_val
            self.sub_full = np.ascontiguousarray(full).reshape(-1)
        self.c = _DevPlan(self.n, self.n_short, self.n_halo, pi(self.rp), pi(self.col), pd(self.val), int(plan["vec_grid"]),
                          int(plan["sell_grid"]), len(self.tile_first), pi(self.tile_first), pi(self.tile_rows),
                          int(plan["boundary_grid"]), int(plan["boundary_lpr"]), len(self.boundary), pi(self.boundary),
                          len(items), pi(self.long_items), sub_grid, sub_n, pi(self.sub_rows), pi(self.sub_rp), pi(self.sub_col),
                          pd(self.sub_val), dense, sub_n if dense == 1 else 0, self.sub_strip, pd(self.sub_full), 0, pd(self.sub_full))
        assert len(self.boundary) == int(plan["boundary_rows"]) or self.n_halo == 0
        assert len(items) == int(plan["long_items"])


def _order_lib():
    L = lib()
    if not hasattr(L, "_dev_ready"):
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_dev_spmv.argtypes = [C.POINTER(_DevPlan), _dp, _dp, _dp, C.c_int, dp]
        L.orc_sub_tiles_part.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp]
        L.orc_dev_init.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, dp, dp]
        L.orc_dev_p.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_double]
        L.orc_dev_x.argtypes = [C.c_int, _dp, _dp, C.c_double]
        L.orc_dev_xr.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, _dp, dp]
        L.orc_dev_cg1_update.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                         C.c_int, dp]
        L._dev_ready = True
    return L


def pcg_resident_ranks(ranks, counts, displs, rhs, x0, dinv, tol, max_it, fixed_iters=0):
    """The single-reduction PCG of a group of ranks as every rank's ONE register-resident launch adds it
    (csrc/kmcf_cgr.hip, nranks > 1): rows as the row-per-lane kernel adds them over [own | halo], a rank's sums over
    its tiles / blocks / reduction tree (plan: resident_tpb, resident_g1), the ranks' sums by a butterfly with one lane
    per rank.  ranks: list of DeviceRank; rhs, x0, dinv: GLOBAL vectors in the caller's order.  Returns dict(x, r
    (global), iterations, converged, bb, rz)."""
    L = lib()
    if not hasattr(L, "_resident_ready"):
        L.orc_resident_spmv.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.orc_resident_rank_sum.restype = C.c_double
        L.orc_resident_rank_sum.argtypes = [C.c_int, _ip, _ip, C.c_int, C.c_int, _dp]
        L.orc_wave_sum_n.restype = C.c_double
        L.orc_wave_sum_n.argtypes = [_dp, C.c_int]
        L._resident_ready = True
    P = len(ranks)
    assert all(rk.plan["resident_tpb"] > 0 for rk in ranks), "a rank's plan names no resident launch"
    tol2 = float(tol) * float(tol)
    n_glob = int(displs[-1]) + int(counts[-1])
    S = []
    for q, rk in enumerate(ranks):
        sl = slice(int(displs[q]), int(displs[q]) + int(counts[q]))
        g = lambda v: _f(v)[sl][rk.perm].copy()
        S.append(dict(b=g(rhs), x=g(x0), di=g(dinv) if dinv is not None else np.ones(rk.n), xv=np.zeros(rk.n + rk.n_halo),
                      w=np.zeros(rk.n), tf=_i(rk.plan["tile_first"]), tr=_i(rk.plan["tile_rows"])))

    def spmv(key):
        glob = np.zeros(n_glob)
        for q, rk in enumerate(ranks):
            glob[int(displs[q]) + rk.perm] = S[q][key]
        for q, rk in enumerate(ranks):
            xv = S[q]["xv"]
            xv[:rk.n] = S[q][key]
            if rk.n_halo:
                xv[rk.n:] = glob[rk.halo_gid]
            L.orc_resident_spmv(rk.n, rk.rp, rk.col, rk.val, xv, S[q]["w"])

    def total(key):
        loc = np.zeros(64)
        for q, rk in enumerate(ranks):
            loc[q] = L.orc_resident_rank_sum(len(S[q]["tf"]), S[q]["tf"], S[q]["tr"], int(rk.plan["resident_tpb"]),
                                             int(rk.plan["resident_g1"]), np.ascontiguousarray(S[q][key]))
        return loc[0] if P == 1 else L.orc_wave_sum_n(loc, P)

    for s_ in S:
        s_["z"] = s_["x"].copy()
    spmv("z")
    for s_ in S:
        s_["r"] = s_["b"] + (-1.0) * s_["w"]
        s_["z"] = s_["r"] * s_["di"]
        s_["gp"] = s_["r"] * s_["z"]
        s_["bbp"] = s_["b"] * s_["b"]
        s_["p"] = np.zeros_like(s_["r"])
        s_["s"] = np.zeros_like(s_["r"])
    limit = int(fixed_iters) if fixed_iters > 0 else int(max_it)
    bb = g_old = a_old = rz_last = 0.0
    iters, done = 0, False
    for k in range(1, limit + 1):
        first = k == 1
        spmv("z")
        for s_ in S:
            s_["dp"] = s_["z"] * s_["w"]
        gamma, delta = total("gp"), total("dp")
        if first:
            bb = total("bbp")
        go = True if fixed_iters > 0 else (gamma / bb > tol2)
        rz_last = gamma
        if not go:
            done = True
            break
        if first:
            beta, alpha = 0.0, gamma / delta
        else:
            beta = gamma / g_old
            alpha = gamma / (delta - beta * gamma / a_old)
        g_old, a_old = gamma, alpha
        iters += 1
        na = -alpha
        for s_ in S:
            s_["s"] = s_["w"].copy() if first else s_["w"] + beta * s_["s"]
            s_["r"] = s_["r"] + na * s_["s"]
            zn = s_["r"] * s_["di"]
            s_["p"] = s_["z"].copy() if first else s_["z"] + beta * s_["p"]
            s_["x"] = s_["x"] + alpha * s_["p"]
            s_["z"] = zn
            s_["gp"] = s_["r"] * s_["z"]
    if not done:
        rz_last = total("gp")
    xo, ro = np.zeros(n_glob), np.zeros(n_glob)
    for q, rk in enumerate(ranks):
        xo[int(displs[q]) + rk.perm] = S[q]["x"]
        ro[int(displs[q]) + rk.perm] = S[q]["r"]
    return dict(x=xo, r=ro, iterations=iters, converged=1 if done else 0, bb=bb, rz=rz_last,
                relres=float(np.sqrt(rz_last / bb)) if bb > 0 else 0.0)


def pcg_device_order_ranks(ranks, counts, displs, rhs, x0, dinv, tol, max_it, fixed_iters=0, variant="classic",
                           sub_counts=None, sub_displs=None, history=False):
    """The reference's Jacobi-PCG over a group of ranks in the device's summation order: every rank's kernels by
    kmcf_oracle_order.c, the halo exchange and the all-reduce (one partial per rank, added in rank order) here.
    ranks: list of DeviceRank; counts / displs: the row partition; rhs, x0, dinv: GLOBAL vectors, caller's order
    (dinv None: no preconditioner).  variant "classic" (pcg_loop: the reference's recurrence, two reductions) or
    "cg1r" (pcg1_loop).  Returns dict(x, r (global), iterations, converged, bb, rz[, rz_hist])."""
    L = _order_lib()
    P = len(ranks)
    pre = 1 if dinv is not None else 0
    tol2 = float(tol) * float(tol)
    st = []
    for q, rk in enumerate(ranks):
        sl = slice(int(displs[q]), int(displs[q]) + int(counts[q]))
        g = lambda v: _f(v)[sl][rk.perm].copy()
        st.append(dict(r=g(rhs), x=g(x0), dinv=g(dinv) if pre else np.ones(rk.n), xv=np.zeros(rk.n + rk.n_halo),
                       Ap=np.zeros(rk.n + 1), z=np.zeros(rk.n), p=np.zeros(rk.n), s=np.zeros(rk.n)))
    n_glob = int(displs[-1]) + int(counts[-1])
    n_sub = int(sub_displs[-1]) + int(sub_counts[-1]) if sub_counts is not None else 0

    def allreduce(vals):                       # one partial per rank, rank order (loopback_allreduce / p2p_allreduce_kernel)
        acc = 0.0
        for v in vals:
            acc += v
        return acc

    # tunnel block as dense symmetric tiles whose strips are dealt to the ranks (kmcf_subop::spread): the whole block
    # from the ranks' row slices, once
    spread = any(rk.spread for rk in ranks)
    if spread:
        assert all(rk.spread or not rk.c.sub_n for rk in ranks)
        F = np.zeros((n_sub, n_sub))
        for q, rk in enumerate(ranks):
            if rk.c.sub_n:
                rows_of = np.repeat(np.arange(rk.c.sub_n), np.diff(rk.sub_rp)) + int(sub_displs[q])
                F[rows_of, rk.sub_col] = rk.sub_val
        F = np.ascontiguousarray(F)
        SL = max(rk.sub_strip for rk in ranks)
        sub_total = np.zeros(64 * ((n_sub + 63) // 64))
        ypart = np.zeros_like(sub_total)
        dpt = C.POINTER(C.c_double)
        for q, rk in enumerate(ranks):
            rk.c.sub_row0 = int(sub_displs[q])
            rk.c.sub_total = sub_total.ctypes.data_as(dpt)

    def spmv(key, with_dot):
        """Ap = A v on every rank for v = st[key] (own rows); returns the local p.Ap sums"""
        glob = np.zeros(n_glob)
        for q, rk in enumerate(ranks):
            glob[int(displs[q]) + rk.perm] = st[q][key]
        xsub = np.zeros(max(n_sub, 1))
        if n_sub:
            for q, rk in enumerate(ranks):
                if rk.c.sub_n:
                    xsub[int(sub_displs[q]):int(sub_displs[q]) + rk.c.sub_n] = st[q][key][rk.sub_rows]
        if spread:                                 # every rank's partial of all sums, added in rank order (sub_combine_kernel)
            sub_total[:] = 0.0
            for q in range(P):
                L.orc_sub_tiles_part(n_sub, SL, F, xsub, q, P, ypart)
                sub_total[:] = sub_total + ypart
        out = []
        for q, rk in enumerate(ranks):
            xv = st[q]["xv"]
            xv[:rk.n] = st[q][key]
            if rk.n_halo:
                xv[rk.n:] = glob[rk.halo_gid]
            pap = C.c_double(0.0)
            L.orc_dev_spmv(C.byref(rk.c), xv, xsub, st[q]["Ap"], 1 if with_dot else 0, C.byref(pap))
            out.append(pap.value)
        return out

    hist = []
    limit = int(fixed_iters) if fixed_iters > 0 else int(max_it)
    iters, done, bb, rz_last = 0, False, 0.0, 0.0
    if variant == "classic":
        for q in range(P):
            st[q]["p"][:] = st[q]["x"]
        spmv("p", False)
        loc = []
        for q, rk in enumerate(ranks):
            a, b = C.c_double(), C.c_double()
            L.orc_dev_init(rk.n, rk.c.vec_grid, st[q]["r"], st[q]["Ap"], st[q]["dinv"], pre, st[q]["z"], C.byref(a), C.byref(b))
            loc.append((a.value, b.value))
        rz, bb = allreduce([v[0] for v in loc]), allreduce([v[1] for v in loc])
        rz_par, xa, pending = [0.0, 0.0], 0.0, False
        for k in range(1, limit + 1):
            par, first = k & 1, k == 1
            rz_new = rz
            go = True if fixed_iters > 0 else (rz_new / bb > tol2)
            rz_last = rz_new
            hist.append(rz_new)
            if not go:
                if pending:
                    for q, rk in enumerate(ranks):
                        L.orc_dev_x(rk.n, st[q]["x"], st[q]["p"], xa)
                pending, done = False, True
                break
            rz_par[par] = rz_new
            iters += 1
            beta = 0.0 if first else rz_new / rz_par[par ^ 1]
            for q, rk in enumerate(ranks):
                L.orc_dev_p(rk.n, st[q]["p"], st[q]["z"], st[q]["x"], beta, 1 if first else 0, 1 if pending else 0, xa)
            pAp = allreduce(spmv("p", True))
            a = rz_par[par] / pAp
            loc = []
            for q, rk in enumerate(ranks):
                v = C.c_double()
                L.orc_dev_xr(rk.n, rk.c.vec_grid, st[q]["r"], st[q]["Ap"], st[q]["dinv"], pre, a, st[q]["z"], C.byref(v))
                loc.append(v.value)
            rz = allreduce(loc)
            xa, pending = a, True
        if not done:
            rz_last = rz
            hist.append(rz)
            if pending:
                for q, rk in enumerate(ranks):
                    L.orc_dev_x(rk.n, st[q]["x"], st[q]["p"], xa)
    else:
        for q in range(P):
            st[q]["z"][:] = st[q]["x"]
        spmv("z", False)
        g_loc, b_loc = [], []
        for q, rk in enumerate(ranks):
            a, b = C.c_double(), C.c_double()
            L.orc_dev_init(rk.n, rk.c.vec_grid, st[q]["r"], st[q]["Ap"], st[q]["dinv"], pre, st[q]["z"], C.byref(a), C.byref(b))
            g_loc.append(a.value)
            b_loc.append(b.value)
        g_par, a_par = [0.0, 0.0], [0.0, 0.0]
        for k in range(1, limit + 1):
            par, first = k & 1, k == 1
            d_loc = spmv("z", True)
            gamma, delta = allreduce(g_loc), allreduce(d_loc)
            if first:
                bb = allreduce(b_loc)
            go = True if fixed_iters > 0 else (gamma / bb > tol2)
            if first:
                beta, alpha = 0.0, gamma / delta
            else:
                beta = gamma / g_par[par ^ 1]
                alpha = gamma / (delta - beta * gamma / a_par[par ^ 1])
            rz_last = gamma
            hist.append(gamma)
            if not go:
                done = True
                break
            g_par[par], a_par[par] = gamma, alpha
            iters += 1
            g_loc = []
            for q, rk in enumerate(ranks):
                v = C.c_double()
                L.orc_dev_cg1_update(rk.n, rk.c.vec_grid, st[q]["x"], st[q]["r"], st[q]["p"], st[q]["s"], st[q]["z"], st[q]["Ap"],
                                     st[q]["dinv"], pre, alpha, beta, 1 if first else 0, C.byref(v))
                g_loc.append(v.value)
        if not done:
            rz_last = allreduce(g_loc)
            hist.append(rz_last)
    xo, ro = np.zeros(n_glob), np.zeros(n_glob)
    for q, rk in enumerate(ranks):
        xo[int(displs[q]) + rk.perm] = st[q]["x"]
        ro[int(displs[q]) + rk.perm] = st[q]["r"]
    out = dict(x=xo, r=ro, iterations=iters, converged=done or not ((rz_last / bb if bb else 0.0) > tol2), bb=bb, rz=rz_last,
               relres=float(np.sqrt(rz_last / bb)) if bb > 0 else 0.0)
    if history:
        out["rz_hist"] = np.array(hist)
    return out


def pcg_jacobi_omp(row_ptr, col, val, rhs, x0, dinv, tol, max_it, fixed_iters=0):
    n = len(row_ptr) - 1
    r = _f(rhs).copy()
    x = _f(x0).copy()
    rel = C.c_double(0.0)
    it = lib().orc_pcg_jacobi_omp(n, _i(row_ptr), _i(col), _f(val), r, x, _f(dinv), float(tol), int(max_it),
                                  int(fixed_iters), C.byref(rel))
    return x, it, rel.value


def omp_threads():
    return lib().orc_omp_threads()


def set_threads(n):
    lib().orc_set_threads(int(n))


def halo_lists(row_ptr, col, P, rank):
    """Neighbour list + halo index lists of `rank` (dist_matrix.cpp:237-487)."""
    n = len(row_ptr) - 1
    counts, displs = partition(n, P)
    nb = np.zeros(P, np.int32)
    nnzb = np.zeros(P, np.int32)
    nc = np.zeros(P, np.int32)
    nr = np.zeros(P, np.int32)
    nnb = lib().orc_halo_lists(_i(row_ptr), _i(col), P, rank, counts, displs, nb, nnzb, nc, nr, None, None)
    cols = np.zeros(max(int(nc[:nnb].sum()), 1), np.int32)
    rows = np.zeros(max(int(nr[:nnb].sum()), 1), np.int32)
    lib().orc_halo_lists(_i(row_ptr), _i(col), P, rank, counts, displs, nb, nnzb, nc, nr,
                         cols.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p))
    out = []
    co = ro = 0
    for k in range(nnb):
        out.append(dict(rank=int(nb[k]), nnz=int(nnzb[k]),
                        cols=cols[co:co + nc[k]].copy(), rows=rows[ro:ro + nr[k]].copy()))
        co += nc[k]
        ro += nr[k]
    return out


class MT19937(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def mt_uniform_stream(seed, n):
    """n draws of std::uniform_real_distribution<double>(0,1) over std::mt19937(seed) (src/random_num.h)."""
    L = lib()
    L.orc_mt_uniform.restype = C.c_double
    g = MT19937()
    L.orc_mt_seed(C.byref(g), C.c_uint32(seed))
    return np.array([L.orc_mt_uniform(C.byref(g)) for _ in range(n)])


def mt_raw_stream(seed, n):
    L = lib()
    L.orc_mt_next.restype = C.c_uint32
    g = MT19937()
    L.orc_mt_seed(C.byref(g), C.c_uint32(seed))
    return np.array([L.orc_mt_next(C.byref(g)) for _ in range(n)], dtype=np.uint64)


def kmc_step(xyz, neigh_idx, layer, T_bg, freq, sigma, k, pot, element, charge, layers, rng_state, blk=2048,
             max_events=4096):
    """execute_kmc_step_mpi (src/kmc_events.cu:333-563), one rank.  rng_state: an MT19937 (advanced in place).
    Returns (event_time, n_events, log[n,3], element_after, charge_after)."""
    L = lib()
    L.orc_kmc_step.restype = C.c_int
    N, nn = neigh_idx.shape
    el = _i(element).copy()
    ch = _i(charge).copy()
    t = C.c_double()
    log = np.zeros(3 * max_events, np.int32)
    E = [_f([l[key] for l in layers]) for key in ("E_gen_0", "E_rec_1", "E_diff_2", "E_diff_3")]
    L.orc_kmc_step.argtypes = [C.c_int, C.c_int, _ip, _ip, C.c_double, C.c_double, C.c_double, C.c_double,
                               _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp, C.c_void_p, C.c_int, C.c_int,
                               C.POINTER(C.c_double), _ip]
    n = L.orc_kmc_step(N, nn, _i(neigh_idx).reshape(-1), _i(layer), T_bg, freq, sigma, k,
                       _f(xyz[:, 0]), _f(xyz[:, 1]), _f(xyz[:, 2]), _f(pot), el, ch, E[0], E[1], E[2], E[3],
                       C.cast(C.byref(rng_state), C.c_void_p), blk, max_events, C.byref(t), log)
    return t.value, n, log[:3 * n].reshape(-1, 3).copy(), el, ch


def mt_state(seed):
    g = MT19937()
    lib().orc_mt_seed(C.byref(g), C.c_uint32(seed))
    return g


def update_temperature_global(site_power, T_bg, a, b, number_steps, C_thermal, small_step):
    return lib().orc_update_temperature_global(_f(site_power), len(site_power), T_bg, a, b, number_steps,
                                               C_thermal, small_step)


def poisson_gridless(xyz, charge, sigma, k, cutoff=20.0, count=None, displ=0):
    x, y, z = _f(xyz[:, 0]), _f(xyz[:, 1]), _f(xyz[:, 2])
    N = len(x)
    count = N if count is None else count
    pot = np.zeros(N)
    lib().orc_poisson_gridless(x, y, z, N, _i(charge), sigma, k, cutoff, count, displ, pot)
    return pot


# ---------------------------------------------------------------------------------------------
# T path (current solve): oracle/kmcf_oracle_T.c.  PARITY UNPINNED by any reference fixture.
# ---------------------------------------------------------------------------------------------
class TSystem:
    """Everything the reference's update_power_gpu_sparse_dist assembles for one KMC step, on one rank
    (src/current_solver_gpu.cu:1430-1655): atom arrays, neighbour matrix (pattern, values, diagonal), tunnel
    sub-block (points, pattern, values, diagonal), preconditioner and right-hand side."""

    def __init__(self, xyz, site_element, site_charge, site_CB_edge, metals, nn_dist, n_inj, n_ext,
                 num_layers_contact, Vd, high_G, low_G, loop_G, tol, m_e, V0, x_lo=-4.2, x_hi=52.65):
        L = lib()
        L.orc_T_atoms.restype = C.c_int
        L.orc_T_atoms.argtypes = [_ip, C.c_int, _ip]
        L.orc_T_pattern.restype = C.c_int64
        L.orc_T_pattern.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _ip, C.c_void_p]
        L.orc_T_values.argtypes = [_dp, _dp, _dp, _ip, _ip, _ip, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_int, C.c_int, _ip, _ip, _dp, _dp]
        L.orc_T_tunnel_points.restype = C.c_int
        L.orc_T_tunnel_points.argtypes = [_ip, _dp, C.c_int, C.c_double, C.c_double, _ip]
        L.orc_T_tunnel_pattern.restype = C.c_int64
        L.orc_T_tunnel_pattern.argtypes = [_dp, _dp, _dp, _dp, _ip, _ip, C.c_int, C.c_int, C.c_double, C.c_double,
                                           _ip, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_void_p]
        L.orc_T_tunnel_values.argtypes = [_dp, _dp, _dp, _dp, _ip, _ip, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_double, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip,
                                          _dp, _dp]
        site_element = _i(site_element)
        N = len(site_element)
        atom_site = np.zeros(N, np.int32)
        self.N_atom = L.orc_T_atoms(site_element, N, atom_site)
        self.atom_site = atom_site[:self.N_atom].copy()
        a = self.atom_site
        self.ax, self.ay, self.az = _f(xyz[a, 0]), _f(xyz[a, 1]), _f(xyz[a, 2])
        self.atom_element = _i(site_element[a])
        self.atom_charge = _i(np.asarray(site_charge)[a])
        self.atom_CB_edge = _f(np.asarray(site_CB_edge)[a])
        self.metals = _i(metals)
        Na = self.N_atom
        self.Nsub = Na + 1
        self.params = dict(nn_dist=nn_dist, n_inj=n_inj, n_ext=n_ext, num_layers_contact=num_layers_contact, Vd=Vd,
                           high_G=high_G, low_G=low_G, loop_G=loop_G, tol=tol, m_e=m_e, V0=V0)
        # neighbour matrix
        self.row_ptr = np.zeros(self.Nsub + 1, np.int32)
        nnz = L.orc_T_pattern(self.ax, self.ay, self.az, Na, nn_dist, n_inj, n_ext, self.row_ptr, None)
        self.col = np.zeros(max(int(nnz), 1), np.int32)
        L.orc_T_pattern(self.ax, self.ay, self.az, Na, nn_dist, n_inj, n_ext, self.row_ptr,
                        self.col.ctypes.data_as(C.c_void_p))
        self.col = self.col[:nnz]
        self.val = np.zeros(nnz)
        self.diag_neigh = np.zeros(self.Nsub)
        L.orc_T_values(self.ax, self.ay, self.az, self.atom_element, self.atom_charge, self.metals, len(self.metals),
                       Na, nn_dist, high_G, low_G, loop_G, n_inj, n_ext, self.row_ptr, self.col, self.val,
                       self.diag_neigh)
        # tunnel sub-block (the reference hard-codes num_metals = 2 here, initialize_sparsity_T.cu:800)
        tidx = np.zeros(max(Na, 1), np.int32)
        self.n_t = L.orc_T_tunnel_points(self.atom_element, self.ax, Na, x_lo, x_hi, tidx)
        self.tunnel_idx = tidx[:self.n_t].copy()
        nt = self.n_t
        self.sub_row_ptr = np.zeros(nt + 1, np.int32)
        targs = (self.ax, self.ay, self.az, self.atom_CB_edge, self.atom_element, self.metals, len(self.metals), Na,
                 nn_dist, tol)
        snnz = L.orc_T_tunnel_pattern(*targs, _i(self.tunnel_idx) if nt else np.zeros(1, np.int32), nt,
                                      num_layers_contact, n_inj, n_ext, self.sub_row_ptr, None)
        self.sub_col = np.zeros(max(int(snnz), 1), np.int32)
        L.orc_T_tunnel_pattern(*targs, _i(self.tunnel_idx) if nt else np.zeros(1, np.int32), nt, num_layers_contact,
                               n_inj, n_ext, self.sub_row_ptr, self.sub_col.ctypes.data_as(C.c_void_p))
        self.sub_val = np.zeros(max(int(snnz), 1))
        self.diag_tunnel = np.zeros(max(nt, 1))
        L.orc_T_tunnel_values(*targs, m_e, V0, _i(self.tunnel_idx) if nt else np.zeros(1, np.int32), nt,
                              num_layers_contact, n_inj, n_ext, self.sub_row_ptr, self.sub_col, self.sub_val,
                              self.diag_tunnel)
        self.sub_col, self.sub_val, self.diag_tunnel = self.sub_col[:snnz], self.sub_val[:snnz], self.diag_tunnel[:nt]
        self.sub_rows = _i(self.tunnel_idx + 2)          # shift_vector_by_constant, initialize_sparsity_T.cu:904
        # preconditioner (assemble_preconditioner + invert_diag, current_solver_gpu.cu:1323-1338) and rhs (:1627-1632)
        d = self.diag_neigh.copy()
        d[self.sub_rows] += self.diag_tunnel
        self.diag = d
        self.dinv = 1 / d
        self.rhs = np.zeros(self.Nsub)
        self.rhs[0] = -loop_G * Vd
        self.rhs[1] = loop_G * Vd

    def _sub(self):
        z1 = np.zeros(1, np.int32)
        return (self.n_t, self.sub_row_ptr, self.sub_col if len(self.sub_col) else z1,
                self.sub_val if len(self.sub_val) else np.zeros(1), self.sub_rows if self.n_t else z1)

    def spmv(self, x):
        L = lib()
        L.orc_T_spmv_split.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp, _ip, _dp, _dp]
        y = np.zeros(self.Nsub)
        L.orc_T_spmv_split(self.Nsub, self.row_ptr, self.col, self.val, *self._sub(), _f(x), y)
        return y

    def solve(self, x0, tol, max_it):
        """conjugate_gradient_jacobi_split_sparse: returns (x, iterations, relres)."""
        L = lib()
        L.orc_T_pcg_split.restype = C.c_int
        L.orc_T_pcg_split.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp, _ip, _dp, _dp, _dp, C.c_double,
                                      C.c_int, C.POINTER(C.c_double)]
        r = self.rhs.copy()
        x = _f(x0).copy()
        rel = C.c_double()
        it = L.orc_T_pcg_split(self.Nsub, self.row_ptr, self.col, self.val, *self._sub(), r, x, self.dinv, float(tol),
                               int(max_it), C.byref(rel))
        return x, it, rel.value

    def imacro(self, m):
        L = lib()
        L.orc_T_imacro.restype = C.c_double
        L.orc_T_imacro.argtypes = [_ip, _ip, _dp, _dp]
        return L.orc_T_imacro(self.row_ptr, self.col, self.val, _f(m))

    def power(self, m, alpha, site_power):
        """m: N_atom + 2 potentials already scaled by G0 (shifted in place); writes site_power in place."""
        L = lib()
        L.orc_T_power.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp, _ip, _dp, C.c_double, C.c_double,
                                  _ip, _ip, _ip, C.c_int, _dp]
        n_t, srp, scol, sval, _ = self._sub()
        L.orc_T_power(self.N_atom, self.row_ptr, self.col, self.val, n_t, srp, scol, sval,
                      _i(self.tunnel_idx) if n_t else np.zeros(1, np.int32), m, self.params["Vd"], alpha,
                      self.atom_element, self.atom_site, self.metals, len(self.metals), site_power)
        return site_power

    def merged_csr(self):
        """The monolithic operator A_n + P^T S P as scipy CSR (for dense cross-checks)."""
        import scipy.sparse as sp
        A = sp.csr_matrix((self.val, self.col, self.row_ptr), shape=(self.Nsub, self.Nsub))
        if self.n_t:
            S = sp.csr_matrix((self.sub_val, self.sub_col, self.sub_row_ptr), shape=(self.n_t, self.n_t)).tocoo()
            A = A + sp.coo_matrix((S.data, (self.sub_rows[S.row], self.sub_rows[S.col])), shape=A.shape).tocsr()
        return A
