"""Host-side mirror of the reference's call surface for the field-solve path.

Names, argument meaning and side effects follow src/gpu_solvers.h:36-263,
src/gpu_buffers.h, src/KMC_comm.h and dist_iterative/ so that the parity tests
read like calls into the reference.  Every function forwards to the C ABI of
libkmcfield.so (include/kmcfield.h); torch is used only to own device memory
and (optionally) to bootstrap the RCCL communicator.  No compute happens here
and nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _L


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


def _ia(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def _da(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class KMC_comm:
    """Partition tables of src/KMC_comm.h:225-289 (split=false: every module uses
    all ranks) plus this rank's libkmcfield communicator."""

    def __init__(self, nrows_K, nrows_T, nrows_pairwise, nrows_events, rank=0, size=1, device=0, _handle=None):
        self.lib = _L.load()
        self.rank_K = self.rank_events = self.rank_pairwise = rank
        self.size_K = self.size_events = self.size_pairwise = size
        self.counts_K, self.displs_K = self.partition(nrows_K, size)
        self.rank_T, self.size_T = rank, size                      # the reference forces comm_T off (KMC_comm.h:243)
        self.counts_T, self.displs_T = self.partition(nrows_T, size)
        self.counts_pairwise, self.displs_pairwise = self.partition(nrows_pairwise, size)
        self.counts_events, self.displs_events = self.partition(nrows_events, size)
        self.device = device
        self.loopback = _handle is not None
        if _handle is None:
            h = C.c_void_p()
            _L.check(self.lib.kmcf_comm_create(C.byref(h), device, size, rank), "kmcf_comm_create")
            _handle = h
        self.handle = _handle

    @classmethod
    def loopback_group(cls, nrows_K, nrows_T, nrows_pairwise, nrows_events, size, device=0):
        """All `size` ranks of an in-process test group on one GPU (kmcf_comm_create_loopback); each
        returned KMC_comm must be driven by its own host thread."""
        lib = _L.load()
        arr = (C.c_void_p * size)()
        _L.check(lib.kmcf_comm_create_loopback(arr, device, size), "kmcf_comm_create_loopback")
        return [cls(nrows_K, nrows_T, nrows_pairwise, nrows_events, rank=r, size=size, device=device,
                    _handle=C.c_void_p(arr[r])) for r in range(size)]

    @staticmethod
    def partition(nrows, size):
        lib = _L.load()
        counts = np.zeros(size, np.int32)
        displs = np.zeros(size, np.int32)
        _L.check(lib.kmcf_partition(int(nrows), int(size), counts.ctypes.data_as(C.POINTER(C.c_int)),
                                    displs.ctypes.data_as(C.POINTER(C.c_int))), "kmcf_partition")
        return counts, displs

    def connect(self, dist=None):
        """Bootstrap RCCL: rank 0 creates the unique id, torch.distributed (any
        backend) broadcasts its 128 bytes, every rank joins.  No-op for 1 rank."""
        if self.size_K == 1 or self.loopback:
            _L.check(self.lib.kmcf_comm_connect(self.handle, None), "kmcf_comm_connect")
            return
        assert dist is not None and dist.is_initialized(), "multi-rank groups need torch.distributed for bootstrap"
        buf = (C.c_char * _L.KMCF_UNIQUE_ID_BYTES)()
        if self.rank_K == 0:
            _L.check(self.lib.kmcf_comm_unique_id(buf), "kmcf_comm_unique_id")
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
        use_cuda = dist.get_backend() == "nccl"
        if use_cuda:
            t = t.cuda(self.device)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        _L.check(self.lib.kmcf_comm_connect(self.handle, C.create_string_buffer(raw, len(raw))), "kmcf_comm_connect")

    def connect_p2p(self, dist):
        """Bootstrap the peer-to-peer transport WITHOUT RCCL (kmcf_comm_p2p_export / _import): every rank exports
        the IPC handle of its window, the host program all-gathers them (here: torch.distributed, any backend), every
        rank maps its peers.  Used where RCCL cannot run (several ranks on one GPU) and by tests."""
        _L.check(self.lib.kmcf_comm_connect(self.handle, None), "kmcf_comm_connect")
        nb = 64   # KMCF_P2P_HANDLE_BYTES
        buf = (C.c_char * nb)()
        _L.check(self.lib.kmcf_comm_p2p_export(self.handle, buf), "kmcf_comm_p2p_export")
        mine = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
        parts = [torch.zeros(nb, dtype=torch.uint8) for _ in range(self.size_K)]
        dist.all_gather(parts, mine)
        raw = b"".join(bytes(t.tolist()) for t in parts)
        _L.check(self.lib.kmcf_comm_p2p_import(self.handle, C.create_string_buffer(raw, len(raw))), "kmcf_comm_p2p_import")

    def transport(self):
        return self.lib.kmcf_comm_transport(self.handle).decode()

    def rccl_ranks(self):
        return int(self.lib.kmcf_comm_rccl_ranks(self.handle))

    def select_transport(self, use_p2p):
        _L.check(self.lib.kmcf_comm_select_transport(self.handle, 1 if use_p2p else 0), "kmcf_comm_select_transport")

    def sync(self):
        _L.check(self.lib.kmcf_comm_sync(self.handle), "kmcf_comm_sync")

    def close(self):
        if self.handle:
            self.lib.kmcf_comm_destroy(self.handle)
            self.handle = None


class GPUBuffers:
    """Device SoA of src/gpu_buffers.h:12-162 (the members the K path touches).
    Arrays are torch CUDA tensors; K_distributed is the libkmcfield K state."""

    def __init__(self, N, site_element, site_x, site_y, site_z, nn, sigma, k, lattice, metals, device=0,
                 site_power=None, T_bg=300.0):
        dev = torch.device("cuda", device)
        self.device = dev
        self.N_ = int(N)
        self.nn_ = int(nn)
        self.num_metal_types_ = len(metals)
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.site_element = torch.as_tensor(np.asarray(site_element, np.int32), **i32)
        self.site_x = torch.as_tensor(np.asarray(site_x, np.float64), **f64)
        self.site_y = torch.as_tensor(np.asarray(site_y, np.float64), **f64)
        self.site_z = torch.as_tensor(np.asarray(site_z, np.float64), **f64)
        self.site_charge = torch.zeros(N, **i32)
        self.site_potential_boundary = torch.zeros(N, **f64)
        self.site_potential_charge = torch.zeros(N, **f64)
        self.site_power = torch.zeros(N, **f64) if site_power is None else torch.as_tensor(site_power, **f64)
        self.T_bg = torch.tensor([T_bg], **f64)
        self.metal_types = torch.as_tensor(np.asarray(metals, np.int32), **i32)
        self.lattice_host = np.asarray(lattice, np.float64)
        self.sigma, self.k = float(sigma), float(k)
        self.neigh_idx = None
        self.cutoff_idx = None         # kmcf_pairwise* (replaces cutoff_idx / cutoff_window)
        self.K_distributed = None      # kmcf_kstate* (K_distributed + K_p_distributed + contact patterns)
        self.T_distributed = None      # kmcf_tstate* (T_distributed + T_p_distributed + atom_* arrays)
        self.site_CB_edge = None
        self.atom_virtual_potentials = None   # N_atom + 2 doubles, allocated by initialize_sparsity_T
        self.N_atom_ = 0

    def freeGPUmemory(self):
        if self.T_distributed is not None:
            _L.load().kmcf_tstate_destroy(self.T_distributed)
            self.T_distributed = None
        if self.K_distributed is not None:
            _L.load().kmcf_kstate_destroy(self.K_distributed)
            self.K_distributed = None
        if self.cutoff_idx is not None:
            _L.load().kmcf_pairwise_destroy(self.cutoff_idx)
            self.cutoff_idx = None


# --------------------------------------------------------------------------
# gpu_solvers.h surface
# --------------------------------------------------------------------------

def compute_neighbor_list(kmc_comm, gpubuf, nn_dist=3.5, max_num_neighbors=52):
    """compute_neighbor_list (gpu_solvers.h:43; src/neighbor_lists_gpu.cu:252-292)."""
    lib = _L.load()
    count = int(kmc_comm.counts_events[kmc_comm.rank_events])
    displ = int(kmc_comm.displs_events[kmc_comm.rank_events])
    gpubuf.neigh_idx = torch.empty(max(count, 1) * max_num_neighbors, dtype=torch.int32, device=gpubuf.device)
    gpubuf.nn_ = max_num_neighbors
    _L.check(lib.kmcf_neighbor_list(kmc_comm.handle, _ptr(gpubuf.site_x), _ptr(gpubuf.site_y), _ptr(gpubuf.site_z),
                                    gpubuf.N_, float(nn_dist), max_num_neighbors, count, displ,
                                    _ptr(gpubuf.neigh_idx)), "kmcf_neighbor_list")


def initialize_sparsity_K(gpubuf, pbc, nn_dist, num_atoms_contact, kmc_comm):
    """initialize_sparsity_K (gpu_solvers.h:53; src/iterative_solvers_gpu.cu:262-488)."""
    lib = _L.load()
    h = C.c_void_p()
    lat, latp = _da(gpubuf.lattice_host)
    cnt, cntp = _ia(kmc_comm.counts_K)
    dsp, dspp = _ia(kmc_comm.displs_K)
    _L.check(lib.kmcf_initialize_sparsity_K(kmc_comm.handle, _ptr(gpubuf.site_x), _ptr(gpubuf.site_y),
                                            _ptr(gpubuf.site_z), latp, gpubuf.N_, int(pbc), float(nn_dist),
                                            int(num_atoms_contact), cntp, dspp, C.byref(h)),
             "kmcf_initialize_sparsity_K")
    gpubuf.K_distributed = h


def update_charge_gpu(site_element, site_charge, neigh_idx, N, nn, metals, num_metals, count, displ, kmc_comm):
    """update_charge_gpu (gpu_solvers.h:149; src/potential_solver_gpu.cu:66-85)."""
    lib = _L.load()
    cnt, cntp = _ia(count)
    dsp, dspp = _ia(displ)
    _L.check(lib.kmcf_update_charge(kmc_comm.handle, _ptr(site_element), _ptr(site_charge), _ptr(neigh_idx),
                                    int(N), int(nn), _ptr(metals), int(num_metals), cntp, dspp), "kmcf_update_charge")


def background_potential_gpu_sparse(gpubuf, N, N_left_tot, N_right_tot, Vd, pbc, high_G, low_G, nn_dist,
                                    num_metals, kmc_step_count=0):
    """background_potential_gpu_sparse (gpu_solvers.h:162; src/potential_solver_gpu.cu:846-1128).
    Returns the solve statistics (the reference prints the iteration count on rank 0)."""
    lib = _L.load()
    st = _L.SolveStats()
    _L.check(lib.kmcf_background_potential_sparse(gpubuf.K_distributed, _ptr(gpubuf.site_element),
                                                  _ptr(gpubuf.site_charge), _ptr(gpubuf.metal_types), int(num_metals),
                                                  _ptr(gpubuf.site_potential_boundary), int(N), int(N_left_tot),
                                                  int(N_right_tot), float(Vd), float(high_G), float(low_G),
                                                  C.byref(st)), "kmcf_background_potential_sparse")
    return st.as_dict()


def update_CB_edge_gpu_sparse(gpubuf, N, N_left_tot, N_right_tot, Vd, pbc, high_G, low_G, nn_dist, num_metals):
    """update_CB_edge_gpu_sparse (gpu_solvers.h:143; src/potential_solver_gpu.cu:673-772): writes
    gpubuf.site_CB_edge [J]."""
    lib = _L.load()
    if getattr(gpubuf, "site_CB_edge", None) is None:
        gpubuf.site_CB_edge = torch.zeros(gpubuf.N_, dtype=torch.float64, device=gpubuf.device)
    st = _L.SolveStats()
    _L.check(lib.kmcf_update_CB_edge_sparse(gpubuf.K_distributed, _ptr(gpubuf.site_element), _ptr(gpubuf.site_charge),
                                            _ptr(gpubuf.metal_types), int(num_metals), _ptr(gpubuf.site_CB_edge),
                                            int(N), int(N_left_tot), int(N_right_tot), float(Vd), float(high_G),
                                            float(low_G), C.byref(st)), "kmcf_update_CB_edge_sparse")
    return st.as_dict()


def initialize_sparsity_T(gpubuf, pbc, nn_dist, num_source_inj, num_ground_ext, num_layers_contact, kmc_comm):
    """initialize_sparsity_T (gpu_solvers.h:57; src/initialize_sparsity_T.cu:948-1154).  kmc_comm.counts_T must
    partition N_atom + 1 rows (src/kmc_main.cpp:165-171); pbc is accepted and, like the reference's T kernels,
    not used (they call the non-periodic site_dist_gpu overload)."""
    lib = _L.load()
    if gpubuf.T_distributed is not None:
        lib.kmcf_tstate_destroy(gpubuf.T_distributed)
        gpubuf.T_distributed = None
    h = C.c_void_p()
    cnt, cntp = _ia(kmc_comm.counts_T)
    dsp, dspp = _ia(kmc_comm.displs_T)
    _L.check(lib.kmcf_initialize_sparsity_T(kmc_comm.handle, _ptr(gpubuf.site_x), _ptr(gpubuf.site_y), _ptr(gpubuf.site_z),
                                            _ptr(gpubuf.site_element), gpubuf.N_, float(nn_dist), int(num_source_inj),
                                            int(num_ground_ext), int(num_layers_contact), cntp, dspp, C.byref(h)),
             "kmcf_initialize_sparsity_T")
    gpubuf.T_distributed = h
    info = t_info(gpubuf)
    gpubuf.N_atom_ = info["N_atom"]
    if gpubuf.atom_virtual_potentials is None or gpubuf.atom_virtual_potentials.numel() != info["N_atom"] + 2:
        gpubuf.atom_virtual_potentials = torch.zeros(info["N_atom"] + 2, dtype=torch.float64, device=gpubuf.device)


def current_params(Vd, high_G, low_G, loop_G, G0, tol, m_e, V0, alpha_disp=1.0, solve_heating=False,
                   cg_tolerance=None, cg_max_iterations=100, N_atom=None, contact_x_lo=-4.2, contact_x_hi=52.65):
    """kmcf_current_params_t.  Defaults = what the reference hard-codes: cg tolerance 1e-30 * N_atom and 100
    iterations (src/current_solver_gpu.cu:1455-1456), contact window -4.2 .. 52.65 A (initialize_sparsity_T.cu:645)."""
    if cg_tolerance is None:
        cg_tolerance = 1e-30 * (N_atom or 1)
    return _L.CurrentParams(float(Vd), float(high_G), float(low_G), float(loop_G), float(G0), float(tol), float(m_e),
                            float(V0), float(alpha_disp), float(contact_x_lo), float(contact_x_hi), float(cg_tolerance),
                            int(cg_max_iterations), int(bool(solve_heating)))


def t_assemble(gpubuf, params):
    """Assembly half of update_power_gpu_sparse_dist (src/current_solver_gpu.cu:1496-1632)."""
    _L.check(_L.load().kmcf_t_assemble(gpubuf.T_distributed, _ptr(gpubuf.site_element), _ptr(gpubuf.site_charge),
                                       _ptr(gpubuf.site_CB_edge), _ptr(gpubuf.metal_types), gpubuf.num_metal_types_,
                                       C.byref(params)), "kmcf_t_assemble")


def update_power_gpu_sparse_dist(gpubuf, num_source_inj, num_ground_ext, num_layers_contact, Vd, high_G, low_G, loop_G,
                                 G0, tol, nn_dist, m_e, V0, num_metals, solve_heating_local, solve_heating_global,
                                 alpha_disp, cg_tolerance=None, cg_max_iterations=100, contact_x_lo=-4.2,
                                 contact_x_hi=52.65):
    """update_power_gpu_sparse_dist (gpu_solvers.h:212; src/current_solver_gpu.cu:1430-1855).  Returns
    (imacro, solve statistics); gpubuf.atom_virtual_potentials and gpubuf.site_power are updated in place."""
    lib = _L.load()
    prm = current_params(Vd, high_G, low_G, loop_G, G0, tol, m_e, V0, alpha_disp,
                         bool(solve_heating_local or solve_heating_global), cg_tolerance, cg_max_iterations,
                         gpubuf.N_atom_, contact_x_lo, contact_x_hi)
    st = _L.SolveStats()
    im = C.c_double(0.0)
    _L.check(lib.kmcf_update_power_sparse(gpubuf.T_distributed, _ptr(gpubuf.site_element), _ptr(gpubuf.site_charge),
                                          _ptr(gpubuf.site_CB_edge), _ptr(gpubuf.metal_types), int(num_metals),
                                          _ptr(gpubuf.atom_virtual_potentials), _ptr(gpubuf.site_power), C.byref(prm),
                                          C.byref(im), C.byref(st)), "kmcf_update_power_sparse")
    return im.value, st.as_dict()


def t_info(gpubuf):
    info = _L.TStateInfo()
    _L.check(_L.load().kmcf_tstate_info(gpubuf.T_distributed, C.byref(info)), "kmcf_tstate_info")
    return {k: getattr(info, k) for k, _ in info._fields_}


def t_pattern(gpubuf):
    """(row_ptr, col) of this rank's rows of the T neighbour matrix, global columns."""
    lib = _L.load()
    n = t_info(gpubuf)["rows_this_rank"]
    nnz = C.c_int64()
    rp = np.zeros(n + 1, np.int32)
    _L.check(lib.kmcf_tstate_pattern(gpubuf.T_distributed, rp.ctypes.data_as(C.POINTER(C.c_int)), None, C.byref(nnz)),
             "kmcf_tstate_pattern")
    col = np.zeros(max(nnz.value, 1), np.int32)
    _L.check(lib.kmcf_tstate_pattern(gpubuf.T_distributed, rp.ctypes.data_as(C.POINTER(C.c_int)),
                                     col.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nnz)), "kmcf_tstate_pattern")
    return rp, col[:nnz.value]


def t_atom_sites(gpubuf):
    a = np.zeros(t_info(gpubuf)["N_atom"], np.int32)
    _L.check(_L.load().kmcf_tstate_atom_sites(gpubuf.T_distributed, a.ctypes.data_as(C.POINTER(C.c_int))),
             "kmcf_tstate_atom_sites")
    return a


def t_vectors(gpubuf):
    """After t_assemble: dict(val, diag_neighbour, dinv, rhs) of this rank's rows (caller order)."""
    lib = _L.load()
    info = t_info(gpubuf)
    n = info["rows_this_rank"]
    out = {k: np.zeros(max(n, 1)) for k in ("diag_neighbour", "dinv", "rhs")}
    dp = C.POINTER(C.c_double)
    _L.check(lib.kmcf_tstate_get_vectors(gpubuf.T_distributed, out["diag_neighbour"].ctypes.data_as(dp),
                                         out["dinv"].ctypes.data_as(dp), out["rhs"].ctypes.data_as(dp)),
             "kmcf_tstate_get_vectors")
    out = {k: v[:n] for k, v in out.items()}
    val = np.zeros(max(info["nnz_neighbour"], 1))
    _L.check(lib.kmcf_matrix_get_values(lib.kmcf_tstate_matrix(gpubuf.T_distributed), val.ctypes.data_as(dp)),
             "kmcf_matrix_get_values")
    out["val"] = val[:info["nnz_neighbour"]]
    return out


def t_tunnel(gpubuf):
    """After t_assemble: dict(tunnel_idx, row_ptr, col, val, diag) of this rank's rows of the tunnel sub-block."""
    lib = _L.load()
    info = t_info(gpubuf)
    nt, ns, nnz = info["tunnel_points"], info["tunnel_points_rank"], info["nnz_tunnel"]
    tidx = np.zeros(max(nt, 1), np.int32)
    rp = np.zeros(ns + 1, np.int32)
    col = np.zeros(max(nnz, 1), np.int32)
    val = np.zeros(max(nnz, 1))
    diag = np.zeros(max(ns, 1))
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    _L.check(lib.kmcf_tstate_get_tunnel(gpubuf.T_distributed, tidx.ctypes.data_as(ip), rp.ctypes.data_as(ip),
                                        col.ctypes.data_as(ip), val.ctypes.data_as(dp), diag.ctypes.data_as(dp)),
             "kmcf_tstate_get_tunnel")
    return dict(tunnel_idx=tidx[:nt], row_ptr=rp, col=col[:nnz], val=val[:nnz], diag=diag[:ns], first=info["tunnel_first"])


def sum_and_gather_potential(gpubuf, num_atoms_first_layer, kmc_comm):
    """sum_and_gather_potential (gpu_solvers.h:181; src/potential_solver_gpu.cu:1130-1151) including the
    MPI_Gatherv of the solution done by the caller in the reference (src/kmc_main.cpp:367-384)."""
    lib = _L.load()
    cp = dp = None
    if getattr(gpubuf, "cutoff_idx", None) is not None:     # the pairwise term is in use: gather its rows too
        c_, cp = _ia(kmc_comm.counts_pairwise)
        d_, dp = _ia(kmc_comm.displs_pairwise)
    _L.check(lib.kmcf_sum_and_gather_potential(gpubuf.K_distributed, _ptr(gpubuf.site_potential_boundary),
                                               _ptr(gpubuf.site_potential_charge), gpubuf.N_,
                                               int(num_atoms_first_layer), cp, dp), "kmcf_sum_and_gather_potential")


def compute_cutoff_list(kmc_comm, gpubuf, cutoff_radius=20.0):
    """compute_cutoff_list (gpu_solvers.h:46; src/neighbor_lists_gpu.cu:293-372).  gpubuf.cutoff_idx becomes
    the opaque spatial index that replaces the reference's N x N_cutoff index list."""
    lib = _L.load()
    h = C.c_void_p()
    _L.check(lib.kmcf_compute_cutoff_list(kmc_comm.handle, _ptr(gpubuf.site_x), _ptr(gpubuf.site_y),
                                          _ptr(gpubuf.site_z), gpubuf.N_, float(cutoff_radius), C.byref(h)),
             "kmcf_compute_cutoff_list")
    gpubuf.cutoff_idx = h


def poisson_gridless_gpu(gpubuf, kmc_comm):
    """poisson_gridless_gpu (gpu_solvers.h:173; src/potential_solver_gpu.cu:1620-1655): the rows of this
    rank of site_potential_charge."""
    lib = _L.load()
    r = kmc_comm.rank_pairwise
    _L.check(lib.kmcf_poisson_gridless(gpubuf.cutoff_idx, _ptr(gpubuf.site_x), _ptr(gpubuf.site_y), _ptr(gpubuf.site_z),
                                       _ptr(gpubuf.site_charge), gpubuf.sigma, gpubuf.k,
                                       int(kmc_comm.counts_pairwise[r]), int(kmc_comm.displs_pairwise[r]),
                                       _ptr(gpubuf.site_potential_charge)), "kmcf_poisson_gridless")


def update_temperatureglobal_gpu(site_power, T_bg, N, a_coeff, b_coeff, number_steps, C_thermal, small_step,
                                 kmc_comm):
    """update_temperatureglobal_gpu (gpu_solvers.h:231; src/heat_solver_gpu.cu:53-70)."""
    lib = _L.load()
    _L.check(lib.kmcf_update_temperature_global(kmc_comm.handle, _ptr(site_power), _ptr(T_bg), int(N),
                                                float(a_coeff), float(b_coeff), float(number_steps),
                                                float(C_thermal), float(small_step)),
             "kmcf_update_temperature_global")


class RandomNumberGenerator:
    """RandomNumberGenerator of src/random_num.h: std::mt19937 + uniform_real_distribution<double>(0, 1)."""

    def __init__(self, seed=1):
        self.lib = _L.load()
        h = C.c_void_p()
        _L.check(self.lib.kmcf_rng_create(int(seed), C.byref(h)), "kmcf_rng_create")
        self.handle = h

    def setSeed(self, seed):
        self.lib.kmcf_rng_destroy(self.handle)
        h = C.c_void_p()
        _L.check(self.lib.kmcf_rng_create(int(seed), C.byref(h)), "kmcf_rng_create")
        self.handle = h

    def getRandomNumber(self):
        return self.lib.kmcf_rng_next(self.handle)


def site_layers(site_x, layers):
    """KMCProcess::KMCProcess (src/KMCProcess.cpp:33-50): the LAST layer whose [start_x, end_x] holds x."""
    site_x = np.asarray(site_x)
    lay = np.full(len(site_x), 1000, np.int32)
    for j, l in enumerate(layers):
        lay[(l["start_x"] <= site_x) & (site_x <= l["end_x"])] = j
    if lay.max() >= 1000:
        raise ValueError("a site is not inside the device (KMCProcess.cpp:45-48)")
    return lay


def execute_kmc_step_mpi(kmc_comm, N, count, displs, nn, neigh_idx, site_layer, T_bg, freq, sigma, k, posx, posy, posz,
                         site_potential_charge, site_element, site_charge, rng, layers, max_events=1 << 20,
                         return_log=False):
    """execute_kmc_step_mpi (gpu_solvers.h:250; src/kmc_events.cu:333-563).  T_bg, freq, sigma, k are host
    scalars here.  layers: list of dicts with E_gen_0, E_rec_1, E_diff_2, E_diff_3 (copytoConstMemory).
    Returns the event time (and, with return_log, the number of events and their (i, j, type) rows)."""
    lib = _L.load()
    cnt, cntp = _ia(count)
    dsp, dspp = _ia(displs)
    eg, egp = _da([l["E_gen_0"] for l in layers])
    er, erp = _da([l["E_rec_1"] for l in layers])
    ev, evp = _da([l["E_diff_2"] for l in layers])
    eo, eop = _da([l["E_diff_3"] for l in layers])
    t = C.c_double()
    nev = C.c_int()
    cap = min(int(max_events), 1 << 16) if return_log else 0
    log = np.zeros(max(3 * cap, 1), np.int32)
    if callable(rng):
        # any uniform [0, 1) source, like the reference's RandomNumberGenerator& (one host round trip per event:
        # a foreign generator cannot be drawn ahead and rewound)
        keep = C.CFUNCTYPE(C.c_double, C.c_void_p)(lambda _user: float(rng()))
        fn, user = C.cast(keep, C.c_void_p), None
    else:
        fn, user = C.cast(lib.kmcf_rng_next, C.c_void_p), rng.handle
    _L.check(lib.kmcf_execute_kmc_step(kmc_comm.handle, int(N), cntp, dspp, int(nn), _ptr(neigh_idx), _ptr(site_layer),
                                       float(T_bg), float(freq), float(sigma), float(k), _ptr(posx), _ptr(posy),
                                       _ptr(posz), _ptr(site_potential_charge), _ptr(site_element), _ptr(site_charge),
                                       len(layers), egp, erp, evp, eop, fn, user,
                                       cap if return_log else int(max_events), C.byref(t), C.byref(nev),
                                       log.ctypes.data_as(C.POINTER(C.c_int)) if return_log else None),
             "kmcf_execute_kmc_step")
    if return_log:
        return t.value, nev.value, log[:3 * nev.value].reshape(-1, 3).copy()
    return t.value


# K-state inspection helpers used by the parity tests --------------------------------

def k_assemble(gpubuf, Vd, high_G, low_G):
    lib = _L.load()
    _L.check(lib.kmcf_k_assemble(gpubuf.K_distributed, _ptr(gpubuf.site_element), _ptr(gpubuf.site_charge),
                                 _ptr(gpubuf.metal_types), gpubuf.num_metal_types_, float(Vd), float(high_G),
                                 float(low_G)), "kmcf_k_assemble")


def k_pattern(gpubuf, which=0):
    lib = _L.load()
    mat = Distributed_matrix.from_handle(lib.kmcf_kstate_matrix(gpubuf.K_distributed))
    n = mat.info()["rows_this_rank"]
    nnz = C.c_int64()
    _L.check(lib.kmcf_kstate_pattern(gpubuf.K_distributed, which, None, None, C.byref(nnz)), "kmcf_kstate_pattern")
    rp = np.zeros(n + 1, np.int32)
    col = np.zeros(max(nnz.value, 1), np.int32)
    _L.check(lib.kmcf_kstate_pattern(gpubuf.K_distributed, which, rp.ctypes.data_as(C.POINTER(C.c_int)),
                                     col.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nnz)), "kmcf_kstate_pattern")
    return rp, col[:nnz.value]


def k_vectors(gpubuf):
    lib = _L.load()
    mat = Distributed_matrix.from_handle(lib.kmcf_kstate_matrix(gpubuf.K_distributed))
    n = mat.info()["rows_this_rank"]
    out = {k: np.zeros(max(n, 1)) for k in ("diag", "dinv", "rhs", "left", "right")}
    p = {k: v.ctypes.data_as(C.POINTER(C.c_double)) for k, v in out.items()}
    _L.check(lib.kmcf_k_get_vectors(gpubuf.K_distributed, p["diag"], p["dinv"], p["rhs"], p["left"], p["right"]),
             "kmcf_k_get_vectors")
    out = {k: v[:n] for k, v in out.items()}
    out["val"] = mat.get_values()
    return out


# --------------------------------------------------------------------------
# dist_iterative surface
# --------------------------------------------------------------------------

class Distributed_matrix:
    """Distributed_matrix + its Distributed_vector (dist_iterative/dist_objects.h:11-36, 69-233).
    Construct from the rows of this rank in CSR with GLOBAL column indices (ctor 1, :158-167)."""

    def __init__(self, kmc_comm, matrix_size, counts, displacements, col_indices_in, row_ptr_in, data_in):
        self.lib = _L.load()
        self.owned = True
        h = C.c_void_p()
        cnt, cntp = _ia(counts)
        dsp, dspp = _ia(displacements)
        rp, rpp = _ia(row_ptr_in)
        ci, cip = _ia(col_indices_in)
        dp = None
        if data_in is not None:
            dv, dp = _da(data_in)
        _L.check(self.lib.kmcf_matrix_create_csr(kmc_comm.handle, int(matrix_size), cntp, dspp, rpp, cip, dp,
                                                 C.byref(h)), "kmcf_matrix_create_csr")
        self.handle = h

    @classmethod
    def split_sparse(cls, kmc_comm, matrix_size, counts, displacements, col_indices_in, row_ptr_in, data_in,
                     subblock_size, count_subblock, displ_subblock, subblock_global_rows,
                     sub_row_ptr, sub_col_indices, sub_data):
        """Distributed_matrix + Distributed_subblock_sparse (dist_objects.h:52-65): the operator
        A_neighbour + P^T A_sub P of conjugate_gradient_jacobi_split_sparse, merged at build time."""
        self = cls.__new__(cls)
        self.lib = _L.load()
        self.owned = True
        h = C.c_void_p()
        a = [_ia(v) for v in (counts, displacements, row_ptr_in, col_indices_in, count_subblock, displ_subblock,
                              subblock_global_rows, sub_row_ptr, sub_col_indices)]
        dv, dp = _da(data_in)
        sv, sp_ = _da(sub_data)
        _L.check(self.lib.kmcf_matrix_create_split_sparse(kmc_comm.handle, int(matrix_size), a[0][1], a[1][1], a[2][1],
                                                          a[3][1], dp, int(subblock_size), a[4][1], a[5][1], a[6][1],
                                                          a[7][1], a[8][1], sp_, C.byref(h)),
                 "kmcf_matrix_create_split_sparse")
        self.handle = h
        return self

    @classmethod
    def from_handle(cls, handle):
        self = cls.__new__(cls)
        self.lib = _L.load()
        self.handle = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.owned = False
        return self

    def info(self):
        inf = _L.MatrixInfo()
        _L.check(self.lib.kmcf_matrix_info(self.handle, C.byref(inf)), "kmcf_matrix_info")
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def row_order(self):
        """kmcf_matrix_row_order: (perm, n_short, tile_ends) of the internal row order."""
        n = self.info()["rows_this_rank"]
        perm = np.zeros(max(n, 1), np.int32)
        ns, nt = C.c_int(0), C.c_int(0)
        _L.check(self.lib.kmcf_matrix_row_order(self.handle, perm.ctypes.data_as(C.POINTER(C.c_int)), C.byref(ns), None,
                                                C.byref(nt)), "kmcf_matrix_row_order")
        ends = np.zeros(max(nt.value, 1), np.int32)
        _L.check(self.lib.kmcf_matrix_row_order(self.handle, None, None, ends.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nt)),
                 "kmcf_matrix_row_order")
        return perm[:n], ns.value, ends[:nt.value]

    def sum_plan(self, with_csr=True):
        """kmcf_matrix_sum_plan: the integers that fix the summation order of the solver kernels, the row-per-lane
        tiles and (with_csr) the CSR in the internal order -- what the CPU oracle needs to add in the device's order."""
        pl = _L.SumPlan()
        _L.check(self.lib.kmcf_matrix_sum_plan(self.handle, C.byref(pl), None, None, None, None, None), "kmcf_matrix_sum_plan")
        out = {k: getattr(pl, k) for k, _ in pl._fields_ if k != "reserved"}
        nt, n, nnz = max(pl.sell_tiles, 1), pl.rows, int(self.info()["nnz"])
        first, rows = np.zeros(nt, np.int32), np.zeros(nt, np.int32)
        rp = np.zeros(n + 1, np.int32)
        col = np.zeros(max(nnz, 1), np.int32)
        val = np.zeros(max(nnz, 1))
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        _L.check(self.lib.kmcf_matrix_sum_plan(self.handle, None, ip(first), ip(rows), ip(rp) if with_csr else None,
                                               ip(col) if with_csr else None,
                                               val.ctypes.data_as(C.POINTER(C.c_double)) if with_csr else None),
                 "kmcf_matrix_sum_plan")
        out["tile_first"], out["tile_rows"] = first[:pl.sell_tiles], rows[:pl.sell_tiles]
        if with_csr:
            out["row_ptr"], out["col"], out["val"] = rp, col[:nnz], val[:nnz]
        out["perm"] = self.row_order()[0]
        gid = np.zeros(max(pl.halo_cols, 1), np.int32)
        _L.check(self.lib.kmcf_matrix_halo_columns(self.handle, ip(gid)), "kmcf_matrix_halo_columns")
        out["halo_gid"] = gid[:pl.halo_cols]
        return out

    def neighbours(self):
        """[(rank, nnz, cols_per_neighbour, rows_per_neighbour), ...] starting with self."""
        out = []
        for k in range(self.info()["number_of_neighbours"]):
            r, nnz, nc, nr = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            _L.check(self.lib.kmcf_matrix_neighbour(self.handle, k, C.byref(r), C.byref(nnz), C.byref(nc), None,
                                                    C.byref(nr), None), "kmcf_matrix_neighbour")
            cols = np.zeros(max(nc.value, 1), np.int32)
            rows = np.zeros(max(nr.value, 1), np.int32)
            _L.check(self.lib.kmcf_matrix_neighbour(self.handle, k, None, None, None,
                                                    cols.ctypes.data_as(C.POINTER(C.c_int)), None,
                                                    rows.ctypes.data_as(C.POINTER(C.c_int))), "kmcf_matrix_neighbour")
            out.append(dict(rank=r.value, nnz=nnz.value, cols=cols[:nc.value], rows=rows[:nr.value]))
        return out

    def set_values(self, data):
        dv, dp = _da(data)
        _L.check(self.lib.kmcf_matrix_set_values(self.handle, dp), "kmcf_matrix_set_values")

    def get_values(self):
        out = np.zeros(max(self.info()["nnz"], 1))
        _L.check(self.lib.kmcf_matrix_get_values(self.handle, out.ctypes.data_as(C.POINTER(C.c_double))),
                 "kmcf_matrix_get_values")
        return out[:self.info()["nnz"]]

    def spmv(self, p, Ap):
        """dspmv::gpu_packing_cam (dist_iterative/dist_spmv_gpu_packing.cpp:106-228)."""
        _L.check(self.lib.kmcf_spmv(self.handle, _ptr(p), _ptr(Ap)), "kmcf_spmv")

    def replan(self):
        """kmcf_spmv_replan: re-plan the SpMV from the KMCF_SPMV_* environment (measurement aid)."""
        _L.check(self.lib.kmcf_spmv_replan(self.handle), "kmcf_spmv_replan")
        return self.info()

    def spmv_bench(self, reps, with_dot=True):
        ms = C.c_float()
        _L.check(self.lib.kmcf_spmv_bench(self.handle, int(reps), 1 if with_dot else 0, C.byref(ms)),
                 "kmcf_spmv_bench")
        return ms.value

    def comm_bench(self, kind, reps):
        """ms for `reps` x {0: all-reduce of 3 doubles, 1: halo exchange, 2: SpMV kernels without exchange}."""
        ms = C.c_float()
        _L.check(self.lib.kmcf_comm_bench(self.handle, int(kind), int(reps), C.byref(ms)), "kmcf_comm_bench")
        return ms.value

    def close(self):
        if self.owned and self.handle:
            self.lib.kmcf_matrix_destroy(self.handle)
            self.handle = None


def conjugate_gradient_jacobi(A_distributed, r_local_d, x_local_d, diag_inv_local_d, relative_tolerance,
                              max_iterations, fixed_iters=0):
    """iterative_solver::conjugate_gradient_jacobi (dist_iterative/dist_conjugate_gradient.cpp:149-276).
    r_local_d: rhs in / residual out; x_local_d: start guess in / solution out."""
    st = _L.SolveStats()
    _L.check(A_distributed.lib.kmcf_pcg_jacobi(A_distributed.handle, _ptr(r_local_d), _ptr(x_local_d),
                                               _ptr(diag_inv_local_d), float(relative_tolerance),
                                               int(max_iterations), int(fixed_iters), C.byref(st)), "kmcf_pcg_jacobi")
    return st.as_dict()


def conjugate_gradient(A_distributed, r_local_d, x_local_d, relative_tolerance, max_iterations, fixed_iters=0):
    """iterative_solver::conjugate_gradient (dist_conjugate_gradient.cpp:17-121): no preconditioner."""
    return conjugate_gradient_jacobi(A_distributed, r_local_d, x_local_d, None, relative_tolerance, max_iterations,
                                     fixed_iters)


def solve_sparse_CG_Jacobi(A_distributed, d_rhs, d_x, tol=1e-14, max_iterations=50000):
    """solve_sparse_CG_Jacobi (src/iterative_solvers_gpu.cu:716-887): A and rhs are scaled in place."""
    st = _L.SolveStats()
    _L.check(A_distributed.lib.kmcf_solve_sparse_CG_Jacobi(A_distributed.handle, _ptr(d_rhs), _ptr(d_x), float(tol),
                                                           int(max_iterations), C.byref(st)),
             "kmcf_solve_sparse_CG_Jacobi")
    return st.as_dict()


def pack_gpu(kmc_comm, packed_buffer, unpacked_buffer, indices, number_of_elements):
    _L.check(_L.load().kmcf_pack(kmc_comm.handle, _ptr(packed_buffer), _ptr(unpacked_buffer), _ptr(indices),
                                 int(number_of_elements)), "kmcf_pack")


def unpack_gpu(kmc_comm, unpacked_buffer, packed_buffer, indices, number_of_elements):
    _L.check(_L.load().kmcf_unpack(kmc_comm.handle, _ptr(unpacked_buffer), _ptr(packed_buffer), _ptr(indices),
                                   int(number_of_elements)), "kmcf_unpack")


def unpack_add(kmc_comm, unpacked_buffer, packed_buffer, indices, number_of_elements):
    _L.check(_L.load().kmcf_unpack_add(kmc_comm.handle, _ptr(unpacked_buffer), _ptr(packed_buffer), _ptr(indices),
                                       int(number_of_elements)), "kmcf_unpack_add")


def elementwise_vector_vector(kmc_comm, array1, array2, result, size):
    _L.check(_L.load().kmcf_elementwise_vector_vector(kmc_comm.handle, _ptr(array1), _ptr(array2), _ptr(result),
                                                      int(size)), "kmcf_elementwise_vector_vector")
