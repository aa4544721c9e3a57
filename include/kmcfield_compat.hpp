// kmcfield_compat.hpp -- drop-in definitions of the reference's extern "C" field-solve entry
// points (src/gpu_solvers.h:36-263) on top of libkmcfield's C ABI (kmcfield.h).
//
// How it is used: inside the reference tree, compile ONE translation unit
//
//     // src/kmcfield_backend.cpp
//     #include "kmcfield_compat.hpp"
//
// in place of the bodies of initialize_sparsity_K / initialize_sparsity_CB (src/iterative_solvers_gpu.cu:262-488,
// 199-260), compute_neighbor_list / compute_cutoff_list (src/neighbor_lists_gpu.cu:252-372), update_charge_gpu,
// background_potential_gpu_sparse, update_CB_edge_gpu_sparse, poisson_gridless_gpu, sum_and_gather_potential
// (src/potential_solver_gpu.cu), initialize_sparsity_T (src/initialize_sparsity_T.cu:948-1154),
// update_power_gpu_sparse_dist (src/current_solver_gpu.cu:1430-1855), execute_kmc_step_mpi / copytoConstMemory
// (src/kmc_events.cu) and update_temperatureglobal_gpu (src/heat_solver_gpu.cu:53-70) -- every gpu_solvers.h
// entry point src/kmc_main.cpp calls -- and link
// -lkmcfield.  src/kmc_main.cpp, Device, KMCProcess, GPUBuffers and KMC_comm stay unchanged:
// the signatures below are the reference's own.
//
// This header needs the REFERENCE's headers (gpu_solvers.h -> gpu_buffers.h, KMC_comm.h, utils.h,
// mpi.h); it is not compiled as part of this repository's library.  Error convention restored to
// the reference's: print and exit(1) (gpuErrchk, src/utils.h:145-154).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <map>

#include "gpu_solvers.h"   // reference: GPUBuffers, KMC_comm, ELEMENT, MPI
#include "kmcfield.h"

namespace kmcf_compat {

inline void check(int rc, const char *what)
{
    if (rc != KMCF_OK) {
        std::fprintf(stderr, "kmcfield: %s failed (%d): %s\n", what, rc, kmcf_last_error());
        std::exit(1);
    }
}

// One libkmcfield communicator per GROUP of ranks: comm_K, comm_T, comm_events, comm_pairwise and
// MPI_COMM_WORLD have the same members in the same order when split = false (src/KMC_comm.h:225-243), and
// share one kmcf_comm (two RCCL communicators) instead of one each.  The RCCL unique ids are created on rank
// 0 and broadcast with the MPI the reference already has.
inline kmcf_comm *comm_of(MPI_Comm mpi)
{
    static std::map<MPI_Comm, kmcf_comm *> table;
    auto it = table.find(mpi);
    if (it != table.end()) return it->second;
    for (auto &kv : table) {
        int cmp = MPI_UNEQUAL;
        MPI_Comm_compare(mpi, kv.first, &cmp);
        if (cmp == MPI_IDENT || cmp == MPI_CONGRUENT) { table[mpi] = kv.second; return kv.second; }
    }
    int rank = 0, size = 1, device = 0;
    MPI_Comm_rank(mpi, &rank);
    MPI_Comm_size(mpi, &size);
    if (hipGetDevice(&device) != hipSuccess) { std::fprintf(stderr, "kmcfield: hipGetDevice failed\n"); std::exit(1); }
    kmcf_comm *c = nullptr;
    check(kmcf_comm_create(&c, device, size, rank), "kmcf_comm_create");
    if (size > 1) {
        char id[KMCF_UNIQUE_ID_BYTES];
        if (rank == 0) check(kmcf_comm_unique_id(id), "kmcf_comm_unique_id");
        MPI_Bcast(id, KMCF_UNIQUE_ID_BYTES, MPI_BYTE, 0, mpi);
        check(kmcf_comm_connect(c, id), "kmcf_comm_connect");
    } else {
        check(kmcf_comm_connect(c, nullptr), "kmcf_comm_connect");
    }
    table[mpi] = c;
    return c;
}

// gpubuf.K_distributed is a Distributed_matrix* in the reference; main never dereferences it
// (it only passes gpubuf around), so the slot carries the opaque libkmcfield K state.
inline kmcf_kstate *kstate_of(GPUBuffers &gpubuf) { return reinterpret_cast<kmcf_kstate *>(gpubuf.K_distributed); }
// likewise gpubuf.T_distributed (+ T_p_distributed and the atom_* arrays, which only the T functions touch)
inline kmcf_tstate *tstate_of(GPUBuffers &gpubuf) { return reinterpret_cast<kmcf_tstate *>(gpubuf.T_distributed); }

static_assert(sizeof(ELEMENT) == sizeof(int), "ELEMENT must be a 4-byte enum (src/utils.h:37-44)");

}  // namespace kmcf_compat

extern "C" {
// (plain, non-inline definitions: include this header from exactly ONE translation unit)

// src/neighbor_lists_gpu.cu:252-292
void compute_neighbor_list(MPI_Comm &event_comm, int *counts, int *displ, Device &device, GPUBuffers &gpubuf,
                                  KMCParameters &p)
{
    (void)device; (void)p;
    kmcf_comm *c = kmcf_compat::comm_of(event_comm);
    int rank = 0;
    MPI_Comm_rank(event_comm, &rank);
    const int nn = 52;            // max_num_neighbors, :261
    const double nn_dist = 3.5;   // :262
    if (hipMalloc((void **)&gpubuf.neigh_idx, (size_t)counts[rank] * nn * sizeof(int)) != hipSuccess) std::exit(1);
    kmcf_compat::check(kmcf_neighbor_list(c, gpubuf.site_x, gpubuf.site_y, gpubuf.site_z, gpubuf.N_, nn_dist, nn,
                                          counts[rank], displ[rank], gpubuf.neigh_idx), "kmcf_neighbor_list");
}

// src/iterative_solvers_gpu.cu:262-488
void initialize_sparsity_K(GPUBuffers &gpubuf, int pbc, const double nn_dist, int num_atoms_contact,
                                  KMC_comm &kmc_comm)
{
    kmcf_comm *c = kmcf_compat::comm_of(kmc_comm.comm_K);
    double lattice[3];
    if (hipMemcpy(lattice, gpubuf.lattice, 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) std::exit(1);
    kmcf_kstate *k = nullptr;
    kmcf_compat::check(kmcf_initialize_sparsity_K(c, gpubuf.site_x, gpubuf.site_y, gpubuf.site_z, lattice, gpubuf.N_, pbc,
                                                  nn_dist, num_atoms_contact, kmc_comm.counts_K, kmc_comm.displs_K, &k),
                       "kmcf_initialize_sparsity_K");
    gpubuf.K_distributed = reinterpret_cast<Distributed_matrix *>(k);
}

// src/iterative_solvers_gpu.cu:199-260: the reference builds a second, single-GPU copy of the K pattern for the
// conduction-band-edge solve; update_CB_edge_gpu_sparse below works on the K state's own pattern instead.
void initialize_sparsity_CB(GPUBuffers &gpubuf, int pbc, const double nn_dist, int num_atoms_contact)
{
    (void)pbc; (void)nn_dist; (void)num_atoms_contact;
    if (!kmcf_compat::kstate_of(gpubuf)) {
        std::fprintf(stderr, "kmcfield: initialize_sparsity_CB needs initialize_sparsity_K first\n");
        std::exit(1);
    }
}

// src/initialize_sparsity_T.cu:948-1154, once per bias point (src/kmc_main.cpp:273)
void initialize_sparsity_T(GPUBuffers &gpubuf, int pbc, const double nn_dist, int num_source_inj, int num_ground_ext,
                           int num_layers_contact, KMC_comm &kmc_comm)
{
    (void)pbc;     // the reference's T kernels use the non-periodic distance whatever pbc says (gpu_solvers.h:280-285)
    if (kmcf_compat::tstate_of(gpubuf)) kmcf_compat::check(kmcf_tstate_destroy(kmcf_compat::tstate_of(gpubuf)), "kmcf_tstate_destroy");
    kmcf_tstate *t = nullptr;
    kmcf_compat::check(kmcf_initialize_sparsity_T(kmcf_compat::comm_of(kmc_comm.comm_T), gpubuf.site_x, gpubuf.site_y, gpubuf.site_z,
                                                  reinterpret_cast<const int *>(gpubuf.site_element), gpubuf.N_, nn_dist,
                                                  num_source_inj, num_ground_ext, num_layers_contact, kmc_comm.counts_T,
                                                  kmc_comm.displs_T, &t), "kmcf_initialize_sparsity_T");
    gpubuf.T_distributed = reinterpret_cast<Distributed_matrix *>(t);
    kmcf_tstate_info_t info;
    kmcf_compat::check(kmcf_tstate_info(t, &info), "kmcf_tstate_info");
    gpubuf.N_atom_ = info.N_atom;                                          // update_atom_arrays, current_solver_gpu.cu:1355
}

// src/current_solver_gpu.cu:1430-1855 (gpu_solvers.h:212).  CG settings are the reference's
// (relative_tolerance = 1e-30 * N_atom, max_iterations = 100, :1455-1456); the contact window of the tunnel
// points its hard-coded -4.2 .. 52.65 A (src/initialize_sparsity_T.cu:645).  What the reference has behind its
// benchmark exit(1) -- I_macro and the dissipated power -- is computed (kmcfield.h).
void update_power_gpu_sparse_dist(hipblasHandle_t, hipsolverDnHandle_t, GPUBuffers &gpubuf, const int num_source_inj,
                                  const int num_ground_ext, const int num_layers_contact, const double Vd,
                                  const double high_G, const double low_G, const double loop_G, const double G0,
                                  const double tol, const double nn_dist, const double m_e, const double V0, int num_metals,
                                  double *imacro, const bool solve_heating_local, const bool solve_heating_global,
                                  const double alpha_disp)
{
    (void)num_source_inj; (void)num_ground_ext; (void)num_layers_contact; (void)nn_dist;   // fixed at initialize_sparsity_T
    kmcf_current_params_t p;
    p.Vd = Vd; p.high_G = high_G; p.low_G = low_G; p.loop_G = loop_G; p.G0 = G0; p.tol = tol; p.m_e = m_e; p.V0 = V0;
    p.alpha_disp = alpha_disp;
    p.contact_x_lo = -4.2; p.contact_x_hi = 52.65;
    p.cg_tolerance = 1e-30 * gpubuf.N_atom_;
    p.cg_max_iterations = 100;
    p.solve_heating = (solve_heating_local || solve_heating_global) ? 1 : 0;
    kmcf_solve_stats_t st;
    kmcf_compat::check(kmcf_update_power_sparse(kmcf_compat::tstate_of(gpubuf), reinterpret_cast<const int *>(gpubuf.site_element),
                                                gpubuf.site_charge, gpubuf.site_CB_edge,
                                                reinterpret_cast<const int *>(gpubuf.metal_types), num_metals,
                                                gpubuf.atom_virtual_potentials, gpubuf.site_power, &p, imacro, &st),
                       "kmcf_update_power_sparse");
    int rank = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    if (rank == 0) {   // dist_conjugate_gradient_split_sparse.cpp:170-172, current_solver_gpu.cu:1823
        std::printf("iteration (T) = %d, relative residual = %g\n", st.iterations + 1, st.relres);
        std::printf("I_macro: %g\n", *imacro * (1e6));
    }
}

// src/potential_solver_gpu.cu:66-85
void update_charge_gpu(ELEMENT *d_site_element, int *d_site_charge, int *d_neigh_idx, int N, int nn,
                              const ELEMENT *d_metals, const int num_metals, const int *count, const int *displ,
                              MPI_Comm &comm)
{
    kmcf_compat::check(kmcf_update_charge(kmcf_compat::comm_of(comm), reinterpret_cast<const int *>(d_site_element),
                                          d_site_charge, d_neigh_idx, N, nn, reinterpret_cast<const int *>(d_metals),
                                          num_metals, count, displ), "kmcf_update_charge");
}

// src/potential_solver_gpu.cu:846-1128 (the hipBLAS / hipSOLVER handles are unused there as well)
void background_potential_gpu_sparse(hipblasHandle_t, hipsolverDnHandle_t, GPUBuffers &gpubuf, const int N,
                                            const int N_left_tot, const int N_right_tot, const double Vd,
                                            const int pbc, const double high_G, const double low_G,
                                            const double nn_dist, const int num_metals, int kmc_step_count)
{
    (void)pbc; (void)nn_dist; (void)kmc_step_count;
    kmcf_solve_stats_t st;
    kmcf_compat::check(kmcf_background_potential_sparse(kmcf_compat::kstate_of(gpubuf),
                                                        reinterpret_cast<const int *>(gpubuf.site_element),
                                                        gpubuf.site_charge,
                                                        reinterpret_cast<const int *>(gpubuf.metal_types), num_metals,
                                                        gpubuf.site_potential_boundary, N, N_left_tot, N_right_tot, Vd,
                                                        high_G, low_G, &st), "kmcf_background_potential_sparse");
    // dist_conjugate_gradient.cpp:272-274
    int rank = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    if (rank == 0)
        std::printf("iteration K = %d, relative residual = %g\n", st.iterations + 1, st.relres);
}

// src/potential_solver_gpu.cu:673-772 (single GPU; needs initialize_sparsity_K to have run)
void update_CB_edge_gpu_sparse(hipblasHandle_t, hipsolverDnHandle_t, GPUBuffers &gpubuf, const int N,
                               const int N_left_tot, const int N_right_tot, const double d_Vd, const int pbc,
                               const double d_high_G, const double d_low_G, const double nn_dist, const int num_metals)
{
    (void)pbc; (void)nn_dist;
    kmcf_solve_stats_t st;
    kmcf_compat::check(kmcf_update_CB_edge_sparse(kmcf_compat::kstate_of(gpubuf),
                                                  reinterpret_cast<const int *>(gpubuf.site_element), gpubuf.site_charge,
                                                  reinterpret_cast<const int *>(gpubuf.metal_types), num_metals,
                                                  gpubuf.site_CB_edge, N, N_left_tot, N_right_tot, d_Vd, d_high_G, d_low_G,
                                                  &st), "kmcf_update_CB_edge_sparse");
    std::printf("# CG steps: %d\n", st.iterations);   // src/iterative_solvers_gpu.cu:862
}

// src/potential_solver_gpu.cu:1130-1151.  NB: the reference's main gathers the solution to rank 0
// itself before this call (src/kmc_main.cpp:367-384, MPI on device pointers); libkmcfield's
// all-gather makes that gather redundant but harmless.
void sum_and_gather_potential(GPUBuffers &gpubuf, int num_atoms_first_layer, KMC_comm &kmc_comm)
{
    kmcf_compat::check(kmcf_sum_and_gather_potential(kmcf_compat::kstate_of(gpubuf), gpubuf.site_potential_boundary,
                                                     gpubuf.site_potential_charge, gpubuf.N_, num_atoms_first_layer,
                                                     kmc_comm.counts_pairwise, kmc_comm.displs_pairwise),
                       "kmcf_sum_and_gather_potential");
}

// src/neighbor_lists_gpu.cu:293-372: gpubuf.cutoff_idx (int*) carries the opaque spatial index
void compute_cutoff_list(MPI_Comm &pairwise_comm, int *counts, int *displ, Device &device, GPUBuffers &gpubuf,
                         KMCParameters &p)
{
    (void)counts; (void)displ; (void)device; (void)p;
    kmcf_pairwise *pw = nullptr;
    kmcf_compat::check(kmcf_compute_cutoff_list(kmcf_compat::comm_of(pairwise_comm), gpubuf.site_x, gpubuf.site_y,
                                                gpubuf.site_z, gpubuf.N_, 20.0 /* :298 */, &pw), "kmcf_compute_cutoff_list");
    gpubuf.cutoff_idx = reinterpret_cast<int *>(pw);
    gpubuf.N_cutoff_ = 0;
}

// src/potential_solver_gpu.cu:1620-1655 (sigma and k are device scalars in the reference)
void poisson_gridless_gpu(const int num_atoms_contact, const int pbc, const int N, const double *lattice,
                          const double *sigma, const double *k, const double *posx, const double *posy,
                          const double *posz, const int *site_charge, double *site_potential_charge, const int rank,
                          const int size, const int *count, const int *displ, const int *cutoff_window,
                          const int *cutoff_idx, const int N_cutoff)
{
    (void)num_atoms_contact; (void)pbc; (void)N; (void)lattice; (void)size; (void)cutoff_window; (void)N_cutoff;
    double hs = 0, hk = 0;
    if (hipMemcpy(&hs, sigma, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&hk, k, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) std::exit(1);
    kmcf_pairwise *pw = reinterpret_cast<kmcf_pairwise *>(const_cast<int *>(cutoff_idx));
    kmcf_compat::check(kmcf_poisson_gridless(pw, posx, posy, posz, site_charge, hs, hk, count[rank], displ[rank],
                                             site_potential_charge), "kmcf_poisson_gridless");
}

// src/kmc_events.cu:565-571 + 333-563.  The layer energies are kept on the host; the reference's own
// RandomNumberGenerator is driven through a callback, so its state advances exactly as before.
namespace kmcf_compat {
inline std::vector<double> &layer_E(int which) { static std::vector<double> E[4]; return E[which]; }
inline double next_random_cb(void *user) { return static_cast<RandomNumberGenerator *>(user)->getRandomNumber(); }
}  // namespace kmcf_compat

void copytoConstMemory(std::vector<double> E_gen, std::vector<double> E_rec, std::vector<double> E_Vdiff,
                       std::vector<double> E_Odiff)
{
    kmcf_compat::layer_E(0) = E_gen; kmcf_compat::layer_E(1) = E_rec;
    kmcf_compat::layer_E(2) = E_Vdiff; kmcf_compat::layer_E(3) = E_Odiff;
}

double execute_kmc_step_mpi(MPI_Comm comm, const int N, const int *count, const int *displs, const int nn,
                            const int *neigh_idx, const int *site_layer, const double *lattice, const int pbc,
                            const double *T_bg, const double *freq, const double *sigma, const double *k,
                            const double *posx, const double *posy, const double *posz,
                            const double *site_potential_charge, const double *site_temperature,
                            ELEMENT *site_element, int *site_charge, RandomNumberGenerator &rng)
{
    (void)lattice; (void)pbc; (void)site_temperature;
    double h[4];   // T_bg, freq, sigma, k are device scalars in the reference
    const double *d[4] = {T_bg, freq, sigma, k};
    for (int i = 0; i < 4; ++i)
        if (hipMemcpy(&h[i], d[i], sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) std::exit(1);
    double event_time = 0.0;
    int n_events = 0;
    kmcf_compat::check(kmcf_execute_kmc_step(kmcf_compat::comm_of(comm), N, count, displs, nn, neigh_idx, site_layer, h[0], h[1],
                                             h[2], h[3], posx, posy, posz, site_potential_charge,
                                             reinterpret_cast<int *>(site_element), site_charge,
                                             (int)kmcf_compat::layer_E(0).size(), kmcf_compat::layer_E(0).data(),
                                             kmcf_compat::layer_E(1).data(), kmcf_compat::layer_E(2).data(),
                                             kmcf_compat::layer_E(3).data(), kmcf_compat::next_random_cb, &rng, 1 << 30,
                                             &event_time, &n_events, nullptr), "kmcf_execute_kmc_step");
    int rank = 0;
    MPI_Comm_rank(comm, &rank);
    if (rank == 0) std::printf("Number of KMC events: %d\nEvent time: %g\n", n_events, event_time);   // :551-554
    return event_time;
}

// src/heat_solver_gpu.cu:53-70
void update_temperatureglobal_gpu(const double *site_power, double *T_bg, const int N, const double a_coeff,
                                         const double b_coeff, const double number_steps, const double C_thermal,
                                         const double small_step)
{
    kmcf_compat::check(kmcf_update_temperature_global(kmcf_compat::comm_of(MPI_COMM_WORLD), site_power, T_bg, N, a_coeff,
                                                      b_coeff, number_steps, C_thermal, small_step),
                       "kmcf_update_temperature_global");
}

}  // extern "C"
