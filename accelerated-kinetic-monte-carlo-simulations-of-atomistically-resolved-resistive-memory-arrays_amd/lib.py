"""ctypes binding of libkmcfield.so (C ABI: include/kmcfield.h).

The library is the product; this module only declares its entry points.  There
is no CPU fallback: if the shared library is missing or a call fails, a
KmcfError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMCF_LIB_PATH: another build of the SAME library (tools/lab variants with instrumentation compiled in); never a fallback
LIB_PATH = os.environ.get("KMCF_LIB_PATH") or os.path.join(_HERE, "libkmcfield.so")

KMCF_UNIQUE_ID_BYTES = 256


class KmcfError(RuntimeError):
    pass


class MatrixInfo(C.Structure):
    _fields_ = [("matrix_size", C.c_int), ("rows_this_rank", C.c_int), ("nnz", C.c_int64),
                ("number_of_neighbours", C.c_int), ("halo_cols", C.c_int), ("send_rows", C.c_int),
                ("boundary_rows", C.c_int), ("spmv_kind", C.c_int), ("spmv_coded", C.c_int), ("spmv_tiles", C.c_int),
                ("spmv_window_cols", C.c_int64), ("spmv_stream_entries", C.c_int64)]


class SumPlan(C.Structure):
    """kmcf_sum_plan_t"""
    _fields_ = [("rows", C.c_int), ("n_short", C.c_int), ("halo_cols", C.c_int), ("vec_grid", C.c_int),
                ("sell_active", C.c_int), ("sell_ident", C.c_int), ("sell_grid", C.c_int), ("sell_tiles", C.c_int),
                ("boundary_grid", C.c_int), ("boundary_lpr", C.c_int), ("boundary_rows", C.c_int),
                ("long_items", C.c_int), ("sub_grid", C.c_int), ("cg_variant", C.c_int), ("resident_tpb", C.c_int), ("resident_g1", C.c_int),
                ("reserved", C.c_int * 1)]


class TStateInfo(C.Structure):
    _fields_ = [("N_atom", C.c_int), ("Nsub", C.c_int), ("rows_this_rank", C.c_int), ("nnz_neighbour", C.c_int64),
                ("tunnel_points", C.c_int), ("tunnel_points_rank", C.c_int), ("tunnel_first", C.c_int),
                ("nnz_tunnel", C.c_int64), ("tunnel_dense", C.c_int), ("tunnel_bytes", C.c_int64)]


class CurrentParams(C.Structure):
    """kmcf_current_params_t"""
    _fields_ = [("Vd", C.c_double), ("high_G", C.c_double), ("low_G", C.c_double), ("loop_G", C.c_double),
                ("G0", C.c_double), ("tol", C.c_double), ("m_e", C.c_double), ("V0", C.c_double),
                ("alpha_disp", C.c_double), ("contact_x_lo", C.c_double), ("contact_x_hi", C.c_double),
                ("cg_tolerance", C.c_double), ("cg_max_iterations", C.c_int), ("solve_heating", C.c_int)]


class SolveStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("relres", C.c_double), ("bb", C.c_double),
                ("rz", C.c_double), ("ms_solve", C.c_float), ("ms_assembly", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_P = C.c_void_p          # device or opaque pointer
_IP = C.POINTER(C.c_int)
_DP = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/kmcfield.h declares
SIGNATURES = {
    "kmcf_last_error": (C.c_char_p, []),
    "kmcf_version": (C.c_int, []),
    "kmcf_comm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int]),
    "kmcf_comm_unique_id": (C.c_int, [_P]),
    "kmcf_comm_connect": (C.c_int, [_P, _P]),
    "kmcf_comm_destroy": (C.c_int, [_P]),
    "kmcf_comm_create_loopback": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    "kmcf_comm_sync": (C.c_int, [_P]),
    "kmcf_comm_stream": (_P, [_P]),
    "kmcf_comm_set_caller_stream": (C.c_int, [_P, _P]),
    "kmcf_comm_p2p_export": (C.c_int, [_P, _P]),
    "kmcf_comm_p2p_import": (C.c_int, [_P, _P]),
    "kmcf_comm_transport": (C.c_char_p, [_P]),
    "kmcf_comm_select_transport": (C.c_int, [_P, C.c_int]),
    "kmcf_comm_rccl_ranks": (C.c_int, [_P]),
    "kmcf_partition": (C.c_int, [C.c_int, C.c_int, _IP, _IP]),
    "kmcf_matrix_create_csr": (C.c_int, [_P, C.c_int, _IP, _IP, _IP, _IP, _DP, C.POINTER(_P)]),
    "kmcf_matrix_destroy": (C.c_int, [_P]),
    "kmcf_matrix_create_split_sparse": (C.c_int, [_P, C.c_int, _IP, _IP, _IP, _IP, _DP, C.c_int, _IP, _IP, _IP, _IP,
                                                  _IP, _DP, C.POINTER(_P)]),
    "kmcf_matrix_info": (C.c_int, [_P, C.POINTER(MatrixInfo)]),
    "kmcf_matrix_row_order": (C.c_int, [_P, _IP, _IP, _IP, _IP]),
    "kmcf_matrix_halo_columns": (C.c_int, [_P, _IP]),
    "kmcf_matrix_sum_plan": (C.c_int, [_P, C.POINTER(SumPlan), _IP, _IP, _IP, _IP, _DP]),
    "kmcf_matrix_neighbour": (C.c_int, [_P, C.c_int, _IP, _IP, _IP, _IP, _IP, _IP]),
    "kmcf_matrix_set_values": (C.c_int, [_P, _DP]),
    "kmcf_matrix_get_values": (C.c_int, [_P, _DP]),
    "kmcf_spmv": (C.c_int, [_P, _P, _P]),
    "kmcf_spmv_bench": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "kmcf_spmv_replan": (C.c_int, [_P]),
    "kmcf_comm_bench": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "kmcf_pcg_jacobi": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int, C.c_int, C.POINTER(SolveStats)]),
    "kmcf_solve_sparse_CG_Jacobi": (C.c_int, [_P, _P, _P, C.c_double, C.c_int, C.POINTER(SolveStats)]),
    "kmcf_pack": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "kmcf_unpack": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "kmcf_unpack_add": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "kmcf_elementwise_vector_vector": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "kmcf_initialize_sparsity_K": (C.c_int, [_P, _P, _P, _P, _DP, C.c_int, C.c_int, C.c_double, C.c_int,
                                             _IP, _IP, C.POINTER(_P)]),
    "kmcf_kstate_destroy": (C.c_int, [_P]),
    "kmcf_kstate_matrix": (_P, [_P]),
    "kmcf_kstate_pattern": (C.c_int, [_P, C.c_int, _IP, _IP, C.POINTER(C.c_int64)]),
    "kmcf_update_charge": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _IP, _IP]),
    "kmcf_k_assemble": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double]),
    "kmcf_k_get_vectors": (C.c_int, [_P, _DP, _DP, _DP, _DP, _DP]),
    "kmcf_background_potential_sparse": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                                   C.c_double, C.c_double, C.c_double, C.POINTER(SolveStats)]),
    "kmcf_sum_and_gather_potential": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _IP, _IP]),
    "kmcf_update_CB_edge_sparse": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_double, C.POINTER(SolveStats)]),
    "kmcf_compute_cutoff_list": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, C.POINTER(_P)]),
    "kmcf_pairwise_destroy": (C.c_int, [_P]),
    "kmcf_poisson_gridless": (C.c_int, [_P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_int, C.c_int, _P]),
    "kmcf_initialize_sparsity_T": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, _IP, _IP,
                                             C.POINTER(_P)]),
    "kmcf_tstate_destroy": (C.c_int, [_P]),
    "kmcf_tstate_matrix": (_P, [_P]),
    "kmcf_tstate_info": (C.c_int, [_P, C.POINTER(TStateInfo)]),
    "kmcf_tstate_pattern": (C.c_int, [_P, _IP, _IP, C.POINTER(C.c_int64)]),
    "kmcf_tstate_atom_sites": (C.c_int, [_P, _IP]),
    "kmcf_tstate_get_vectors": (C.c_int, [_P, _DP, _DP, _DP]),
    "kmcf_tstate_get_tunnel": (C.c_int, [_P, _IP, _IP, _IP, _DP, _DP]),
    "kmcf_t_assemble": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(CurrentParams)]),
    "kmcf_update_power_sparse": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, _P, C.POINTER(CurrentParams),
                                           C.POINTER(C.c_double), C.POINTER(SolveStats)]),
    "kmcf_update_temperature_global": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_double]),
    "kmcf_rng_create": (C.c_int, [C.c_uint, C.POINTER(_P)]),
    "kmcf_rng_next": (C.c_double, [_P]),
    "kmcf_rng_destroy": (C.c_int, [_P]),
    "kmcf_execute_kmc_step": (C.c_int, [_P, C.c_int, _IP, _IP, C.c_int, _P, _P, C.c_double, C.c_double, C.c_double,
                                        C.c_double, _P, _P, _P, _P, _P, _P, C.c_int, _DP, _DP, _DP, _DP, _P, _P,
                                        C.c_int, _DP, _IP, _IP]),
    "kmcf_neighbor_list":(C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, _P]),
}

_lib = None


def load():
    """Load libkmcfield.so (in-tree).  Raises KmcfError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KmcfError("libkmcfield.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(make -C <package>/csrc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().kmcf_last_error()
        raise KmcfError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
