/*
 * kmcfield.h -- C ABI of libkmcfield: MI355X-native (gfx950) field solve for
 * DeviceKMC: on-device K-matrix assembly + distributed Jacobi-PCG over a 1-D
 * row-partitioned CSR matrix.
 *
 * Every entry point cites the reference interface it replaces
 * (paths relative to the reference checkout).  All pointers named d_* are
 * DEVICE pointers on the communicator's GPU, h_* are HOST pointers.  Values
 * and vectors are double, indices / charges / ELEMENT are 32-bit int
 * (ELEMENT is a plain enum, src/utils.h:37-44).
 *
 * Error convention: every function returns KMCF_OK (0) or a negative code and
 * records a message retrievable with kmcf_last_error(); nothing calls exit()
 * (the reference aborts: src/utils.h:145-154, dist_iterative/cudaerrchk.h:12-75).
 * The C++ shim in kmcfield_compat.hpp restores abort-on-error for drop-in use.
 *
 * Stream ordering contract: device work of the library runs on its own non-blocking
 * HIP streams.  On entry every compute function orders its stream after everything the
 * caller has queued so far on the caller's stream -- the legacy null stream (what the
 * reference's kernels and plain hipMemcpy use, and PyTorch's default stream) unless
 * kmcf_comm_set_caller_stream() named another -- so buffers written by kernels or
 * asynchronous copies still in flight are complete before the library reads them.  On
 * return every function has synchronised its streams: results are visible to any stream
 * (the reference's contract, hipDeviceSynchronize at dist_conjugate_gradient.cpp:271).
 * Work queued on OTHER streams than the declared one must be synchronised by the caller.
 *
 * Process model: one process per GPU.  A kmcf_comm is this process's member of
 * the solver group (the reference's MPI communicator comm_K, src/KMC_comm.h:
 * 132-289).  Multi-rank groups exchange halos and dot products with RCCL.
 */
#ifndef KMCFIELD_H
#define KMCFIELD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMCF_OK 0
#define KMCF_ERR_ARG (-1)      /* bad argument / shape mismatch                */
#define KMCF_ERR_HIP (-2)      /* a HIP runtime call failed                    */
#define KMCF_ERR_COMM (-3)     /* RCCL failure or communicator not connected   */
#define KMCF_ERR_STATE (-4)    /* call order violated (e.g. solve before assemble) */
#define KMCF_ERR_NOMEM (-5)

#define KMCF_UNIQUE_ID_BYTES 256   /* two RCCL unique ids: halo communicator + reduction communicator */

typedef struct kmcf_comm kmcf_comm;     /* one rank of the solver group                  */
typedef struct kmcf_matrix kmcf_matrix; /* Distributed_matrix + Distributed_vector       */
typedef struct kmcf_kstate kmcf_kstate; /* what initialize_sparsity_K leaves in GPUBuffers */

const char *kmcf_last_error(void);
int kmcf_version(void);

/* ---------------------------------------------------------------------- */
/* Communicator (replaces MPI_Comm comm_K + select_gpu, src/kmc_main.cpp:   */
/* 72-91, src/KMC_comm.h:225-289).                                          */
/* ---------------------------------------------------------------------- */
int kmcf_comm_create(kmcf_comm **out, int device, int nranks, int rank);
/* rank 0 calls kmcf_comm_unique_id, the host program distributes the KMCF_UNIQUE_ID_BYTES
 * (torch.distributed / MPI_Bcast), then every rank calls kmcf_comm_connect.
 * A 1-rank group needs neither. */
int kmcf_comm_unique_id(void *h_id /* KMCF_UNIQUE_ID_BYTES */);
int kmcf_comm_connect(kmcf_comm *c, const void *h_id /* KMCF_UNIQUE_ID_BYTES */);
int kmcf_comm_destroy(kmcf_comm *c);
/* Test transport: all `nranks` members of an in-process group on ONE device (out: array of nranks
 * communicators, each to be driven by its own host thread).  Collectives are host-synchronous
 * device copies; exists because RCCL refuses two ranks on one GPU, so that the multi-rank logic
 * (halo maps, boundary pass, reductions) can be exercised on a 1-GPU box.  Not a performance path. */
int kmcf_comm_create_loopback(kmcf_comm **out, int device, int nranks);
/* Peer-to-peer transport (csrc/kmcf_p2p.hip): the CG's all-reduces and halo exchanges done by kernels over
 * IPC-mapped peer windows instead of one RCCL call each.  KMCF_TRANSPORT=p2p|auto makes kmcf_comm_connect set it up
 * over the RCCL communicator (auto: RCCL stays if set-up or self-test fail).  Without RCCL (kmcf_comm_connect with a
 * NULL id on a multi-rank group): every rank calls kmcf_comm_p2p_export, the host program all-gathers the
 * KMCF_P2P_HANDLE_BYTES of every rank in rank order, every rank calls kmcf_comm_p2p_import.  All waits are bounded
 * (KMCF_P2P_TIMEOUT_MS, default 2000): a peer that never arrives yields KMCF_ERR_COMM, not a hang. */
#define KMCF_P2P_HANDLE_BYTES 64
int kmcf_comm_p2p_export(kmcf_comm *c, void *h_handle /* KMCF_P2P_HANDLE_BYTES */);
int kmcf_comm_p2p_import(kmcf_comm *c, const void *h_handles /* nranks x KMCF_P2P_HANDLE_BYTES */);
const char *kmcf_comm_transport(const kmcf_comm *c);   /* "single", "loopback", "rccl", "p2p ..." */
/* Switch a group that has BOTH transports up (KMCF_TRANSPORT=p2p|auto over RCCL, or an in-process group) between
 * them; collective (every rank, same value), between solves.  Matrices built while the peer-to-peer transport was
 * active work on either. */
int kmcf_comm_select_transport(kmcf_comm *c, int use_p2p);
/* Ranks the RCCL communicator itself reports (ncclCommCount): 0 when no RCCL communicator is connected (one rank,
 * in-process groups, groups bootstrapped through the host program), -1 on an RCCL error.  The reference's benchmark
 * prints MPI_Comm_size the same way (dist_iterative_test/main_test_cg.cpp:94-118). */
int kmcf_comm_rccl_ranks(const kmcf_comm *c);
int kmcf_comm_sync(kmcf_comm *c);            /* wait for the solver streams      */
void *kmcf_comm_stream(kmcf_comm *c);        /* hipStream_t of the compute stream */
/* Declares the hipStream_t the caller queues its own device work on (NULL = legacy null
 * stream, the default); see "Stream ordering contract" above. */
int kmcf_comm_set_caller_stream(kmcf_comm *c, void *hip_stream);

/* Block-row partition rule of the reference (src/KMC_comm.h:249-263,
 * dist_iterative_test/utils.cpp:3-23). */
int kmcf_partition(int nrows, int nranks, int *h_counts, int *h_displs);

/* ---------------------------------------------------------------------- */
/* Distributed CSR matrix (Distributed_matrix ctor 1, dist_iterative/        */
/* dist_objects.h:158-167, dist_matrix.cpp:5-69; Distributed_vector          */
/* dist_vector.cpp:3-42).  Input: the rows of this rank, GLOBAL column ids.  */
/* The matrix must be structurally symmetric (dist_matrix.cpp:3).            */
/* ---------------------------------------------------------------------- */
int kmcf_matrix_create_csr(kmcf_comm *c, int matrix_size, const int *h_counts, const int *h_displs,
                           const int *h_row_ptr, const int *h_col_global, const double *h_val,
                           kmcf_matrix **out);
int kmcf_matrix_destroy(kmcf_matrix *m);

/* "Split sparse" operator of the T-matrix path, A = A_neighbour + P^T A_sub P
 * (conjugate_gradient_jacobi_split_sparse + dspmv_split_sparse::spmm_split_sparse1/2/3,
 * dist_iterative/dist_conjugate_gradient_split_sparse.cpp:18-182, dist_spmv_split_sparse.cpp;
 * Distributed_subblock_sparse, dist_objects.h:52-65).  The sub-block (rows = this rank's
 * count_sub[rank] tunnel rows, columns = GLOBAL sub indices 0..subblock_size) is merged into
 * the row-partitioned CSR at build time; kmcf_spmv / kmcf_pcg_jacobi then apply as usual
 * (no per-SpMV all-gather of the sub-vector).  h_sub_global_rows[s] = global matrix row of
 * sub index s for ALL ranks (the reference all-gathers them once,
 * src/initialize_sparsity_T.cu:752-786). */
int kmcf_matrix_create_split_sparse(kmcf_comm *c, int matrix_size, const int *h_counts, const int *h_displs,
                                    const int *h_row_ptr, const int *h_col_global, const double *h_val,
                                    int subblock_size, const int *h_count_sub, const int *h_displ_sub,
                                    const int *h_sub_global_rows, const int *h_sub_row_ptr,
                                    const int *h_sub_col, const double *h_sub_val, kmcf_matrix **out);

typedef struct {
    int matrix_size;          /* global rows                                        */
    int rows_this_rank;
    int64_t nnz;              /* of this rank, all blocks                           */
    int number_of_neighbours; /* includes self (dist_objects.h:83)                  */
    int halo_cols;            /* sum over k>=1 of nnz_cols_per_neighbour[k]         */
    int send_rows;            /* sum over k>=1 of nnz_rows_per_neighbour[k]         */
    int boundary_rows;        /* local rows that reference a halo column            */
    int spmv_kind;            /* 0 vec, 1 stream, 2 window (LDS-staged x window)    */
    int spmv_coded;           /* values currently dictionary-coded (2 B/nnz): 1 window kernel, 2 row-per-lane kernel */
    int spmv_tiles;           /* window / row-per-lane kernel: number of tiles      */
    int64_t spmv_window_cols; /* the same: sum of the tiles' window sizes           */
    int64_t spmv_stream_entries; /* coded kernels: 16-bit entries streamed per launch; row-per-lane kernels (coded, or
                                    spmv_coded==0 with f64 values): entries per launch, padding included; else 0 */
} kmcf_matrix_info_t;
int kmcf_matrix_info(const kmcf_matrix *m, kmcf_matrix_info_t *info);

/* Halo lists for inspection (cols_per_neighbour / rows_per_neighbour,
 * dist_matrix.cpp:390-487).  k in [0, number_of_neighbours); pass NULL to query
 * sizes.  *neighbour_rank = neighbours[k]. */
int kmcf_matrix_neighbour(const kmcf_matrix *m, int k, int *neighbour_rank, int *nnz_block,
                          int *ncols, int *h_cols, int *nrows, int *h_rows);

/* Internal row order for inspection (no reference counterpart: the reference keeps the caller's order).
 * h_perm[i] = caller's local row stored as internal row i (rows_this_rank entries; NULL: skip).  The rows longer
 * than KMCF_LONG_ROW come last; *n_short = number of rows before them.  h_tile_end: end row (exclusive, internal
 * order) of every tile of the row-per-lane SpMV layout, *n_tiles of them (pass NULL to query the count; 0 when
 * the order was not refined for that layout). */
int kmcf_matrix_row_order(const kmcf_matrix *m, int *h_perm, int *n_short, int *h_tile_end, int *n_tiles);

/* Summation order of the solver kernels for inspection (no reference counterpart).  The CG's dot products and the
 * SpMV's row sums are deterministic: one partial per block, blocks and lanes added in a fixed order that depends
 * only on the quantities below.  tests/ feed them to the CPU oracle, which then adds in the same order and must
 * reproduce the device's iterates bit for bit (oracle/kmcf_oracle_order.c). */
typedef struct {
    int rows, n_short, halo_cols;
    int vec_grid;           /* blocks of the CG's vector kernels = r.z / b.b partials                           */
    int sell_active;        /* 1: the row-per-lane coded kernel computes the short rows (else: see spmv_kind)    */
    int sell_ident;         /* 1: lane t of a tile owns internal row first + t                                   */
    int sell_grid;          /* its blocks = p.Ap partials of the interior pass                                   */
    int sell_tiles;
    int boundary_grid, boundary_lpr, boundary_rows;   /* separate pass over the rows that touch the halo (0: none) */
    int long_items;         /* chunks of the long-row kernel (0: none)                                           */
    int sub_grid;           /* blocks of the tunnel sub-block operator (0: none)                                 */
    int cg_variant;         /* recurrence a solve on this matrix runs now: 0 classic (reference order), 1 single-reduction */
    int resident_tpb;       /* > 0: the solve runs as ONE register-resident launch (kmcf_cgr.hip); tiles per block        */
    int resident_g1;        /* ... blocks per group of its two-stage reduction                                          */
    int reserved[1];
} kmcf_sum_plan_t;
/* h_tile_first / h_tile_rows: first internal row and row count of every row-per-lane tile (sell_tiles entries each;
 * NULL: skip).  h_row_ptr / h_col / h_val: the CSR as stored (internal row order, entries of a row in creation
 * order, own columns as internal row ids, halo columns as rows + halo slot; NULL: skip). */
int kmcf_matrix_sum_plan(const kmcf_matrix *m, kmcf_sum_plan_t *plan, int *h_tile_first, int *h_tile_rows,
                         int *h_row_ptr, int *h_col, double *h_val);

/* Global column of every halo slot (halo_cols entries; column n_loc + h of the stored CSR is global column h_gid[h]). */
int kmcf_matrix_halo_columns(const kmcf_matrix *m, int *h_gid);

/* Overwrite the values (same order as the CSR given at creation). */
int kmcf_matrix_set_values(kmcf_matrix *m, const double *h_val);
/* Copy the values back (creation order). */
int kmcf_matrix_get_values(const kmcf_matrix *m, double *h_val);

/* Ap = A p, distributed: halo pack + exchange overlapped with the interior
 * rows, then boundary rows (dspmv::gpu_packing_cam, dist_iterative/
 * dist_spmv_gpu_packing.cpp:106-228).  d_p, d_Ap: rows_this_rank doubles.
 * Synchronous (returns after the result is visible). */
int kmcf_spmv(kmcf_matrix *m, const double *d_p, double *d_Ap);

/* Timing helper for the roofline line: runs `reps` SpMVs (with_dot != 0: the
 * CG variant with the fused p.Ap partial) on the compute stream bracketed by HIP
 * events on that stream; *ms_total receives the elapsed time. */
int kmcf_spmv_bench(kmcf_matrix *m, int reps, int with_dot, float *ms_total);

/* Re-plans the SpMV of an existing matrix from the KMCF_SPMV_* environment (KIND 0 vec / 1 stream (CSR) /
 * 2 window, CODED 0/1, U, WQ, LPR, LPR2) and re-codes its current values.  Measurement aid: bench.py times
 * the CSR kernel on the same matrix with it (the `roofline_csr` block); tests compare the kernels. */
int kmcf_spmv_replan(kmcf_matrix *m);

/* Diagnostic for multi-rank runs (collective: same arguments on every rank):
 * times `reps` repetitions of one piece of a distributed CG iteration on the
 * compute stream -- kind 0: the all-reduce of the 3 fused scalars, 1: the halo
 * exchange alone (pack, send/recv, wait), 2: the SpMV kernels alone (interior
 * + boundary rows, no exchange), 3 (split operators only): pack + all-gather of
 * the tunnel sub-vector + the wait for it. */
int kmcf_comm_bench(kmcf_matrix *m, int kind, int reps, float *ms_total);

typedef struct {
    int iterations;       /* CG iterations executed (reference prints K = iterations+1) */
    int converged;        /* 1 if the stopping rule was met                             */
    double relres;        /* sqrt(rz/bb), dist_conjugate_gradient.cpp:273                */
    double bb;            /* ||b||^2 (all ranks)                                        */
    double rz;            /* last r.z                                                   */
    float ms_solve;       /* device time (HIP events, compute stream): of the CG loop; kmcf_pcg_jacobi: of everything
                           * the call enqueued (vectors in, r = b - A x0, the iterations, vectors out) */
    float ms_assembly;    /* device time of the assembly kernels (K solve only)         */
} kmcf_solve_stats_t;

/* Jacobi-PCG (iterative_solver::conjugate_gradient_jacobi, dist_iterative/
 * dist_conjugate_gradient.cpp:149-276).  d_r: rhs in, residual out; d_x: start
 * guess in, solution out; d_diag_inv: 1/diag (NULL = unpreconditioned CG,
 * conjugate_gradient :17-121).  Stop when r.z/(b.b) <= tol^2 or after max_it
 * iterations; fixed_iters > 0 runs exactly that many (bench mode). */
int kmcf_pcg_jacobi(kmcf_matrix *m, double *d_r, double *d_x, const double *d_diag_inv,
                    double relative_tolerance, int max_iterations, int fixed_iters,
                    kmcf_solve_stats_t *stats);

/* Single-GPU symmetric-scaled CG (solve_sparse_CG_Jacobi, src/
 * iterative_solvers_gpu.cu:716-887): solves D^-1/2 A D^-1/2 y = D^-1/2 b with
 * an absolute stop ||r||^2 <= tol^2 (tol 1e-14, max 50000 there); A values and
 * rhs are scaled IN PLACE like the reference.  Requires a 1-rank matrix. */
int kmcf_solve_sparse_CG_Jacobi(kmcf_matrix *m, double *d_rhs, double *d_x,
                                double tol, int max_iterations, kmcf_solve_stats_t *stats);

/* Small vector kernels of dist_iterative/utils_cg.cu (:4-111, :323-336). */
int kmcf_pack(kmcf_comm *c, double *d_packed, const double *d_unpacked, const int *d_indices, int n);
int kmcf_unpack(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n);
int kmcf_unpack_add(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n);
int kmcf_elementwise_vector_vector(kmcf_comm *c, const double *d_a, const double *d_b, double *d_out, int n);

/* ---------------------------------------------------------------------- */
/* K path                                                                    */
/* ---------------------------------------------------------------------- */

/* initialize_sparsity_K (src/iterative_solvers_gpu.cu:262-488): pattern of K
 * restricted to the interface sites [N_contact, N - N_contact) for the rows of
 * this rank + left/right contact patterns, built with a cell list instead of
 * the reference's O(n_loc*N) scan.  d_x/d_y/d_z: N site coordinates;
 * h_lattice[3]; counts/displs: row partition of the N - 2*N_contact interface
 * rows (kmc_comm.counts_K / displs_K). */
int kmcf_initialize_sparsity_K(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z,
                               const double *h_lattice, int N, int pbc, double nn_dist, int N_contact,
                               const int *h_counts, const int *h_displs, kmcf_kstate **out);
int kmcf_kstate_destroy(kmcf_kstate *k);
kmcf_matrix *kmcf_kstate_matrix(kmcf_kstate *k);   /* gpubuf.K_distributed */

/* Pattern export for tests (global interface column ids, ascending per row);
 * which: 0 = K rows of this rank, 1 = left contact block, 2 = right contact
 * block.  Pass h_col = NULL to query *nnz. */
int kmcf_kstate_pattern(const kmcf_kstate *k, int which, int *h_row_ptr, int *h_col, int64_t *nnz);

/* update_charge_gpu (src/potential_solver_gpu.cu:12-85): writes the charges of
 * the rows [displ[rank], displ[rank]+count[rank]) and all-gathers them. */
int kmcf_update_charge(kmcf_comm *c, const int *d_site_element, int *d_site_charge, const int *d_neigh_idx,
                       int N, int nn, const int *d_metals, int num_metals,
                       const int *h_count, const int *h_displ);

/* K value assembly only (the first half of background_potential_gpu_sparse,
 * src/potential_solver_gpu.cu:888-1042 with kernels :246-285, :323-367,
 * :438-454, :774-830): fills the CSR values, diagonal, 1/diag and rhs. */
int kmcf_k_assemble(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                    const int *d_metals, int num_metals, double Vd, double high_G, double low_G);
/* Copies of the assembled per-row vectors (rows_this_rank each; NULL to skip). */
int kmcf_k_get_vectors(const kmcf_kstate *k, double *h_diag, double *h_dinv, double *h_rhs,
                       double *h_left, double *h_right);

/* background_potential_gpu_sparse (src/potential_solver_gpu.cu:846-1128):
 * assembly + PCG (tol 1e-14*N_interface, max_it 10000, :885-886), solution
 * written in place into d_site_potential_boundary[N_left + displ ...] whose
 * previous content is the initial guess. */
int kmcf_background_potential_sparse(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                                     const int *d_metals, int num_metals,
                                     double *d_site_potential_boundary,
                                     int N, int N_left_tot, int N_right_tot, double Vd,
                                     double high_G, double low_G, kmcf_solve_stats_t *stats);

/* update_CB_edge_gpu_sparse (src/potential_solver_gpu.cu:575-772): Laplace solve for the
 * conduction-band edge on the K pattern: G = high_G if EITHER site is a metal (:289-319),
 * contacts at +Vd/2 (left) / -Vd/2 (right), solve_sparse_CG_Jacobi (tol 1e-14), boundary
 * fill and scaling by eV_to_J = 1.60217663e-19.  d_site_CB_edge: N doubles, in/out (its
 * interface slice is the start guess, :732).  Single-rank, like the reference.  Overwrites
 * the K values of the state (the next kmcf_k_assemble refills them). */
int kmcf_update_CB_edge_sparse(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                               const int *d_metals, int num_metals, double *d_site_CB_edge, int N,
                               int N_left_tot, int N_right_tot, double Vd, double high_G, double low_G,
                               kmcf_solve_stats_t *stats);

/* The MPI_Gatherv of the solution (src/kmc_main.cpp:367-384) + the two
 * MPI_Bcast + sum_AB_into_A of sum_and_gather_potential
 * (src/potential_solver_gpu.cu:1130-1151): replicates the interface solution
 * on every rank and does site_potential_charge += site_potential_boundary.
 * h_counts_pairwise / h_displs_pairwise (kmc_comm.counts_pairwise, displs_pairwise;
 * may be NULL): the rows of site_potential_charge each rank computed with
 * kmcf_poisson_gridless, all-gathered first (the MPI_Gatherv of src/kmc_main.cpp:
 * 405-425 + the MPI_Bcast of potential_solver_gpu.cu:1139-1142). */
int kmcf_sum_and_gather_potential(kmcf_kstate *k, double *d_site_potential_boundary,
                                  double *d_site_potential_charge, int N, int num_atoms_first_layer,
                                  const int *h_counts_pairwise, const int *h_displs_pairwise);

/* ---------------------------------------------------------------------- */
/* Short-range pairwise Poisson term                           */
/* ---------------------------------------------------------------------- */
typedef struct kmcf_pairwise kmcf_pairwise;

/* compute_cutoff_list (src/neighbor_lists_gpu.cu:293-372; cutoff 20 A there): one-off
 * spatial index of the sites; replaces the N x N_cutoff index list (gpubuf.cutoff_idx). */
int kmcf_compute_cutoff_list(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z, int N,
                             double cutoff_radius, kmcf_pairwise **out);
int kmcf_pairwise_destroy(kmcf_pairwise *p);

/* poisson_gridless_gpu (src/potential_solver_gpu.cu:1620-1655): for the sites
 * [displ, displ+count): site_potential_charge[i] = sum over charged sites j != i within
 * the cutoff of q_j erfc(r/(sigma sqrt 2)) k q / r  (v_solve_gpu, src/gpu_solvers.h:321-329). */
int kmcf_poisson_gridless(kmcf_pairwise *p, const double *d_x, const double *d_y, const double *d_z,
                          const int *d_site_charge, double sigma, double k, int count, int displ,
                          double *d_site_potential_charge);

/* ---------------------------------------------------------------------- */
/* KMC event step                                              */
/* ---------------------------------------------------------------------- */
/* std::mt19937 + std::uniform_real_distribution<double>(0,1): the reference's
 * RandomNumberGenerator (src/random_num.h).  kmcf_rng_next has the callback signature
 * kmcf_execute_kmc_step takes, so a host program may pass its own generator instead. */
typedef struct kmcf_rng kmcf_rng;
int kmcf_rng_create(unsigned int seed, kmcf_rng **out);
double kmcf_rng_next(void *rng);
int kmcf_rng_destroy(kmcf_rng *r);

/* execute_kmc_step_mpi (src/kmc_events.cu:333-563): builds the (site, neighbour) event list of
 * this rank's sites [displs[rank], +count[rank]) (build_event_list_split :128-207), then draws
 * events -- residence-time algorithm: first slot whose cumulative rate exceeds u*total, execute
 * (:284-331), zero the events touching the pair (:237-256), t = -log(u')/total -- until the last
 * drawn t reaches 1/freq; *event_time = that last t (the reference's return value), *n_events the
 * number of executed events, h_event_log (may be NULL; 3*max_events ints) the (i, j, type) triples.
 * d_neigh_idx: this rank's count*nn neighbour slots.  Layer energies: copytoConstMemory
 * (src/kmc_events.cu:565-571).  T_bg, freq, sigma, k: host scalars (device scalars in the reference).
 * ELEMENT / EVENTTYPE codes: src/utils.h:37-60.  Every rank must pass a generator in the same state. */
int kmcf_execute_kmc_step(kmcf_comm *c, int N, const int *h_count, const int *h_displs, int nn,
                          const int *d_neigh_idx, const int *d_site_layer, double T_bg, double freq,
                          double sigma, double k, const double *d_x, const double *d_y, const double *d_z,
                          const double *d_site_potential_charge, int *d_site_element, int *d_site_charge,
                          int num_layers, const double *h_E_gen, const double *h_E_rec,
                          const double *h_E_Vdiff, const double *h_E_Odiff,
                          double (*next_random)(void *), void *rng_user, int max_events,
                          double *event_time, int *n_events, int *h_event_log);

/* ---------------------------------------------------------------------- */
/* T path: current solve (Kirchhoff matrix with two virtual nodes + WKB       */
/* tunnelling sub-block), SURVEY 8 rows a14 / f3.  PARITY UNPINNED: no         */
/* reference fixture exercises it (src/KMC_comm.h:243 disables it from main).  */
/* ---------------------------------------------------------------------- */
typedef struct kmcf_tstate kmcf_tstate; /* gpubuf.T_distributed + T_p_distributed + the atom_* arrays */

/* initialize_sparsity_T (src/initialize_sparsity_T.cu:948-1154), called once per bias point
 * (src/kmc_main.cpp:273): filters the sites into atoms (element != DEFECT, OXYGEN_DEFECT;
 * update_atom_arrays, src/current_solver_gpu.cu:1341-1365 -- the set is invariant under KMC
 * events, which only turn O <-> V and d <-> Od) and builds the pattern of the neighbour matrix
 * over Nsub = N_atom + 1 nodes (0 = extraction, 1 = injection, 2.. = atoms, the last atom --
 * the ground node -- cut) for the rows [displs[rank], +counts[rank]) of counts_T / displs_T
 * (kmc_comm.counts_T).  Cell list instead of the O(n_loc * Nsub) scans; the distance is the
 * non-periodic one the reference's T kernels use whatever pbc says (src/gpu_solvers.h:280-285). */
int kmcf_initialize_sparsity_T(kmcf_comm *c, const double *d_site_x, const double *d_site_y,
                               const double *d_site_z, const int *d_site_element, int N, double nn_dist,
                               int num_source_inj, int num_ground_ext, int num_layers_contact,
                               const int *h_counts_T, const int *h_displs_T, kmcf_tstate **out);
int kmcf_tstate_destroy(kmcf_tstate *t);
kmcf_matrix *kmcf_tstate_matrix(kmcf_tstate *t);   /* gpubuf.T_distributed (neighbour part) */

typedef struct {
    int N_atom;             /* gpubuf.N_atom_                                        */
    int Nsub;               /* matrix size = N_atom + 1                              */
    int rows_this_rank;
    int64_t nnz_neighbour;  /* this rank                                             */
    int tunnel_points;      /* all ranks (num_tunnel_points_global)                  */
    int tunnel_points_rank; /* counts_subblock[rank]                                 */
    int tunnel_first;       /* displ_subblock[rank]                                  */
    int64_t nnz_tunnel;     /* this rank's rows of the tunnel sub-block              */
    int tunnel_dense;       /* 1: stored as dense symmetric 64 x 64 tiles (one rank, block more than a quarter full), 0: bitmap + packed values */
    int64_t tunnel_bytes;   /* bytes one application of the sub-block streams in that storage */
} kmcf_tstate_info_t;
int kmcf_tstate_info(const kmcf_tstate *t, kmcf_tstate_info_t *info);
/* Exports for tests / inspection.  Pattern: rows of this rank, GLOBAL columns ascending (pass
 * h_col = NULL to query *nnz).  atom_site: site index of every atom (N_atom ints). */
int kmcf_tstate_pattern(const kmcf_tstate *t, int *h_row_ptr, int *h_col, int64_t *nnz);
int kmcf_tstate_atom_sites(const kmcf_tstate *t, int *h_atom_site);
/* After kmcf_t_assemble: per-row vectors of this rank (caller row order; NULL to skip):
 * diagonal of the neighbour part, 1/diagonal of the whole operator (the preconditioner), rhs. */
int kmcf_tstate_get_vectors(const kmcf_tstate *t, double *h_diag_neighbour, double *h_dinv, double *h_rhs);
/* After kmcf_t_assemble: the tunnel sub-block of this rank as CSR (columns = global tunnel
 * point ids, ascending; sizes from kmcf_tstate_info): h_tunnel_idx (all tunnel_points atom
 * indices), h_row_ptr (tunnel_points_rank + 1), h_col / h_val (nnz_tunnel), h_diag
 * (tunnel_points_rank).  Any pointer may be NULL. */
int kmcf_tstate_get_tunnel(const kmcf_tstate *t, int *h_tunnel_idx, int *h_row_ptr, int *h_col, double *h_val,
                           double *h_diag);

typedef struct {
    double Vd;
    double high_G, low_G, loop_G;  /* src/kmc_main.cpp:294-296: 1e5*p.high_G, p.low_G, 1e7*p.high_G */
    double G0;                     /* :298                                                        */
    double tol;                    /* [J] barrier-slope tolerance, p.q * 0.01 (:299)              */
    double m_e, V0;                /* effective mass [kg], defect state energy [eV]               */
    double alpha_disp;             /* fraction of the power dissipated as heat (:302)            */
    double contact_x_lo, contact_x_hi;  /* Ti / N atoms with x in this window are tunnel points; the
                                      reference hard-codes -4.2 and 52.65 (initialize_sparsity_T.cu:645) */
    double cg_tolerance;           /* the reference passes 1e-30 * N_atom (current_solver_gpu.cu:1455),
                                      i.e. "run max_iterations"; its commented value is 1e-15 * N_atom */
    int cg_max_iterations;         /* 100 there (:1456)                                           */
    int solve_heating;             /* solve_heating_local || solve_heating_global                */
} kmcf_current_params_t;

/* Assembly half of update_power_gpu_sparse_dist (src/current_solver_gpu.cu:1496-1632): atom
 * arrays, neighbour values + diagonal (populate_T_dist :1051-1247, calc_diagonal_T /
 * insert_diag_T :1279-1321), tunnel sub-block pattern + WKB values + diagonal
 * (assemble_sparse_T_submatrix, src/initialize_sparsity_T.cu:707-946), preconditioner
 * (:1323-1338) and right-hand side (:1627-1632).  d_site_CB_edge: update_CB_edge's output [J]. */
int kmcf_t_assemble(kmcf_tstate *t, const int *d_site_element, const int *d_site_charge,
                    const double *d_site_CB_edge, const int *d_metals, int num_metals,
                    const kmcf_current_params_t *p);

/* update_power_gpu_sparse_dist (src/current_solver_gpu.cu:1430-1855; gpu_solvers.h:212):
 * assembly, conjugate_gradient_jacobi_split_sparse (dist_iterative/
 * dist_conjugate_gradient_split_sparse.cpp:18-182) started from d_atom_virtual_potentials
 * (N_atom + 2 doubles, gpubuf.atom_virtual_potentials) and solved in place, then -- what the
 * reference has behind its benchmark exit(1), restated from update_power_gpu_sparse
 * (:2035-2160) -- the potentials are gathered on every rank and scaled by G0 in place,
 * *imacro = injected current (get_imacro_sparse :501-542), and with solve_heating the
 * potentials are shifted by |min| in place and d_site_power (N doubles) receives
 * -alpha * P of every non-metal atom (semantics of the dense kernels set_ineg / copy_pdisp,
 * :2353-2379, :462-474; DESIGN.md records why set_ineg_sparse is not followed). */
int kmcf_update_power_sparse(kmcf_tstate *t, const int *d_site_element, const int *d_site_charge,
                             const double *d_site_CB_edge, const int *d_metals, int num_metals,
                             double *d_atom_virtual_potentials, double *d_site_power,
                             const kmcf_current_params_t *p, double *imacro, kmcf_solve_stats_t *stats);

/* update_temperatureglobal_gpu (src/heat_solver_gpu.cu:53-70). */
int kmcf_update_temperature_global(kmcf_comm *c, const double *d_site_power, double *d_T_bg, int N,
                                   double a_coeff, double b_coeff, double number_steps,
                                   double C_thermal, double small_step);

/* Site neighbour index list (compute_neighbor_list, src/neighbor_lists_gpu.cu:
 * 55-77, 252-292): nn slots per site, -1 padded, ascending j.  Cell list
 * instead of the O(N^2) scan.  d_neigh_idx: count*nn ints. */
int kmcf_neighbor_list(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z, int N,
                       double nn_dist, int nn, int count, int displ, int *d_neigh_idx);

#ifdef __cplusplus
}
#endif
#endif /* KMCFIELD_H */
