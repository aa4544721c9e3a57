// kmc_loop.cpp -- a plain C++ / HIP host program (no Python, no torch) that drives the per-step loop of the
// reference's main (src/kmc_main.cpp:328-500) through the C ABI of libkmcfield (include/kmcfield.h):
//
//     update_charge_gpu -> background_potential_gpu_sparse -> poisson_gridless_gpu -> sum_and_gather_potential
//     [-> update_power_gpu_sparse_dist with --current] -> execute_kmc_step_mpi
//
// on the reference's shipped 5 nm device (tests/golden/device_5nm.bin: site count, coordinates, elements after
// makeSubstoichiometric -- data extracted from structures/5nm_device/, see tests/golden/make_golden_5nm.py) with
// the parameters of structures/5nm_device/parameters.txt and the layer energies of src/structure_input.h:7-48.
// It prints the cumulative "KMC time is:" line after every step, as src/kmc_main.cpp:519 does; the six values of
// the reference's own run are in structures/5nm_device/expected_output/output1_0.txt and
// tests/test_gpu_cpp_example.py compares against them.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude examples/kmc_loop.cpp -L<package dir> -lkmcfield \
//         -Wl,-rpath,<package dir> -o examples/kmc_loop
//   examples/kmc_loop tests/golden/device_5nm.bin [--current]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kmcfield.h"

#define HIP_OK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)
#define KMCF(call)                                                                                \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != KMCF_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, kmcf_last_error()); std::exit(1); } \
    } while (0)

template <typename T>
static T *to_device(const std::vector<T> &h)
{
    T *d = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d), h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s device_5nm.bin [--current]\n", argv[0]); return 2; }
    const bool solve_current = argc > 2 && std::strcmp(argv[2], "--current") == 0;
    // ---- fixture: int32 N | N x (x, y, z) f64 | N x int32 ELEMENT --------------------------------------------
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    int N = 0;
    if (std::fread(&N, sizeof(int), 1, f) != 1 || N <= 0) { std::fprintf(stderr, "bad header\n"); return 2; }
    std::vector<double> xyz((size_t)3 * N);
    std::vector<int> element((size_t)N);
    if (std::fread(xyz.data(), sizeof(double), xyz.size(), f) != xyz.size() ||
        std::fread(element.data(), sizeof(int), element.size(), f) != element.size()) { std::fprintf(stderr, "short file\n"); return 2; }
    std::fclose(f);
    std::vector<double> x((size_t)N), y((size_t)N), z((size_t)N);
    for (int s = 0; s < N; ++s) { x[s] = xyz[3 * (size_t)s]; y[s] = xyz[3 * (size_t)s + 1]; z[s] = xyz[3 * (size_t)s + 2]; }

    // ---- structures/5nm_device/parameters.txt + src/input_parser.cpp:391-397 ---------------------------------
    const int NL = 576, num_layers_contact = 10, nn = 52;       // num_atoms_first_layer, num_layers_contact, Device.cpp:59
    const double lattice[3] = {108.984220, 51.150000, 51.150000};
    const double nn_dist = 3.5, Vd = 5.0, t_switch = 1e-12, freq = 10e13, T_bg = 300.0;
    const double sigma = 3.5e-10, k_coulomb = 8.987552e9 / 23.0, high_G = 1.0, low_G = 1e-8;
    const int pbc = 0;
    const std::vector<int> metals = {6, 8};                      // Ti, N (src/utils.h:37-44)
    // src/structure_input.h:7-48: start_x, end_x and the four activation energies of the five layers
    const int num_layers = 5;
    const double lay_x0[5] = {-22.0, 0.0, 3.0, 48.1431, 52.6431}, lay_x1[5] = {0.0, 3.0, 48.1431, 52.6431, 90.0};
    const double E_gen[5] = {0.0, 3.93, 3.93, 1.66, 1.73}, E_rec[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const double E_vd[5] = {0.0, 1.09, 1.09, 1.09, 0.0}, E_od[5] = {0.76, 0.76, 0.76, 0.76, 2.8};
    std::vector<int> layer((size_t)N, -1);                      // KMCProcess::KMCProcess, src/KMCProcess.cpp:33-50
    for (int s = 0; s < N; ++s)
        for (int l = 0; l < num_layers; ++l)
            if (lay_x0[l] <= x[s] && x[s] <= lay_x1[l]) layer[s] = l;

    // ---- device buffers (GPUBuffers, src/gpu_buffers.h) --------------------------------------------------------
    HIP_OK(hipSetDevice(0));
    double *d_x = to_device(x), *d_y = to_device(y), *d_z = to_device(z);
    int *d_element = to_device(element), *d_metals = to_device(metals), *d_layer = to_device(layer);
    int *d_charge = nullptr, *d_neigh = nullptr;
    double *d_pot_boundary = nullptr, *d_pot_charge = nullptr, *d_power = nullptr, *d_cb = nullptr, *d_avp = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_charge), (size_t)N * sizeof(int)));
    HIP_OK(hipMemset(d_charge, 0, (size_t)N * sizeof(int)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_neigh), (size_t)N * nn * sizeof(int)));
    for (double **p : {&d_pot_boundary, &d_pot_charge, &d_power, &d_cb}) {
        HIP_OK(hipMalloc(reinterpret_cast<void **>(p), (size_t)N * sizeof(double)));
        HIP_OK(hipMemset(*p, 0, (size_t)N * sizeof(double)));
    }

    // ---- init (src/kmc_main.cpp:165-245) -------------------------------------------------------------------------
    kmcf_comm *comm = nullptr;
    KMCF(kmcf_comm_create(&comm, 0, 1, 0));
    KMCF(kmcf_comm_connect(comm, nullptr));
    int counts_K[1], displs_K[1], counts_N[1], displs_N[1];
    KMCF(kmcf_partition(N - 2 * NL, 1, counts_K, displs_K));
    KMCF(kmcf_partition(N, 1, counts_N, displs_N));
    KMCF(kmcf_neighbor_list(comm, d_x, d_y, d_z, N, nn_dist, nn, N, 0, d_neigh));
    kmcf_pairwise *cutoff = nullptr;
    KMCF(kmcf_compute_cutoff_list(comm, d_x, d_y, d_z, N, 20.0, &cutoff));
    kmcf_kstate *K = nullptr;
    KMCF(kmcf_initialize_sparsity_K(comm, d_x, d_y, d_z, lattice, N, pbc, nn_dist, NL, counts_K, displs_K, &K));
    kmcf_rng *rng = nullptr;
    KMCF(kmcf_rng_create(1 /* rnd_seed_kmc, src/structure_input.h:5 */, &rng));
    kmcf_tstate *T = nullptr;
    kmcf_current_params_t cp;
    if (solve_current) {                                        // src/kmc_main.cpp:270-275, 294-302
        kmcf_solve_stats_t st;
        KMCF(kmcf_update_CB_edge_sparse(K, d_element, d_charge, d_metals, (int)metals.size(), d_cb, N, NL, NL, Vd, high_G, low_G, &st));
        int n_atom = 0;
        for (int s = 0; s < N; ++s) n_atom += (element[s] != 0 && element[s] != 1);
        int counts_T[1], displs_T[1];
        KMCF(kmcf_partition(n_atom + 1, 1, counts_T, displs_T));
        KMCF(kmcf_initialize_sparsity_T(comm, d_x, d_y, d_z, d_element, N, nn_dist, NL, NL, num_layers_contact, counts_T, displs_T, &T));
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_avp), ((size_t)n_atom + 2) * sizeof(double)));
        HIP_OK(hipMemset(d_avp, 0, ((size_t)n_atom + 2) * sizeof(double)));
        cp.Vd = Vd; cp.high_G = 1e5 * high_G; cp.low_G = low_G; cp.loop_G = 1e7 * high_G; cp.G0 = 2 * 3.8612e-5 * 1e-5;
        cp.tol = 1.60217663e-19 * 0.01; cp.m_e = 0.85 * 9.11e-31; cp.V0 = 1.6; cp.alpha_disp = 1.0;
        cp.contact_x_lo = -4.2; cp.contact_x_hi = 52.65;
        cp.cg_tolerance = 1e-15 * n_atom; cp.cg_max_iterations = 2000;   // the reference's commented-out setting (:1454)
        cp.solve_heating = 0;
    }

    // ---- the KMC loop (src/kmc_main.cpp:328-527) ------------------------------------------------------------------
    double kmc_time = 0.0;
    int step = 0;
    while (kmc_time < t_switch && step < 20) {
        kmcf_solve_stats_t st;
        KMCF(kmcf_update_charge(comm, d_element, d_charge, d_neigh, N, nn, d_metals, (int)metals.size(), counts_N, displs_N));
        KMCF(kmcf_background_potential_sparse(K, d_element, d_charge, d_metals, (int)metals.size(), d_pot_boundary, N, NL, NL, Vd,
                                              high_G, low_G, &st));
        KMCF(kmcf_poisson_gridless(cutoff, d_x, d_y, d_z, d_charge, sigma, k_coulomb, N, 0, d_pot_charge));
        if (solve_current) {
            double imacro = 0.0;
            kmcf_solve_stats_t ts;
            KMCF(kmcf_update_power_sparse(T, d_element, d_charge, d_cb, d_metals, (int)metals.size(), d_avp, d_power, &cp, &imacro, &ts));
            std::printf("iteration (T) = %d, relative residual = %g\nI_macro: %g\n", ts.iterations + 1, ts.relres, imacro * 1e6);
        }
        KMCF(kmcf_sum_and_gather_potential(K, d_pot_boundary, d_pot_charge, N, NL, nullptr, nullptr));
        double event_time = 0.0;
        int n_events = 0;
        KMCF(kmcf_execute_kmc_step(comm, N, counts_N, displs_N, nn, d_neigh, d_layer, T_bg, freq, sigma, k_coulomb, d_x, d_y, d_z,
                                   d_pot_charge, d_element, d_charge, num_layers, E_gen, E_rec, E_vd, E_od, kmcf_rng_next, rng,
                                   1 << 20, &event_time, &n_events, nullptr));
        kmc_time += event_time;
        ++step;
        std::printf("iteration K = %d, relative residual = %g\n", st.iterations + 1, st.relres);
        std::printf("Number of KMC events: %d\n", n_events);
        std::printf("KMC time is: %.6g\n", kmc_time);             // src/kmc_main.cpp:519
    }
    std::printf("steps: %d\n", step);
    if (T) KMCF(kmcf_tstate_destroy(T));
    KMCF(kmcf_rng_destroy(rng));
    KMCF(kmcf_pairwise_destroy(cutoff));
    KMCF(kmcf_kstate_destroy(K));
    KMCF(kmcf_comm_destroy(comm));
    return 0;
}
