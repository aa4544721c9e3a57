// K path: sparsity pattern (initialize_sparsity_K), charge rule, fused K value
// assembly, background-potential solve driver, potential gather, global heat.
//
// Reference -> here:
//   calc_nnz_per_row / assemble_K_indices_gpu_off_diagonal_block: O(n_loc * n_cols)
//   distance scans (src/iterative_solvers_gpu.cu:96-157, "20 min" at 40 nm, README.md:13)
//     -> uniform cell list (edge = nn_dist), 27 cells per row, two passes (count, fill+sort).
//   calc_off_diagonal_dist + reduce_rows_into_diag (per block) + 2x reduce_contact_into_diag
//   + insert_into_diag + inverse_diag + calc_rhs_for_A: 5 + 2*nb launches, one thread per
//   row, 7 hipMalloc/hipFree per call (src/potential_solver_gpu.cu:857-1042, 1118-1126)
//     -> one site-class kernel + ONE fused kernel, 16 lanes per row, no allocation per call.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "kmcf_cells.hpp"
#include "kmcf_internal.hpp"

int kmcf_pcg_workspace(kmcf_matrix *m, bool precond, double tol, int max_it, int fixed_iters, kmcf_solve_stats_t *stats);

namespace {

constexpr int EL_OXYGEN_DEFECT = 1, EL_VACANCY = 2;  // src/utils.h:37-44

// populate_neighbor_list, src/neighbor_lists_gpu.cu:55-77: first nn neighbours in
// ascending j, i != j, no pbc in this distance, -1 padding (:277).
constexpr int NL_CAP = 160;
__global__ __launch_bounds__(KMCF_BLOCK) void neighbor_list_kernel(
    cell_grid g, const int *__restrict__ cell_start, const int *__restrict__ cell_items,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
    double nn_dist, int N, int nn, int count, int displ, int *__restrict__ neigh_idx, int *__restrict__ overflow)
{
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const int i = idx + displ;
        int buf[NL_CAP];
        int n = 0;
        for_each_neighbour(g, cell_start, cell_items, x, y, z, i, nn_dist, 1.0, 1.0, 0, 0, N, [&](int j) {
            if (j == i) return;
            if (n >= NL_CAP) { *overflow = 1; return; }
            int k = n;
            while (k > 0 && buf[k - 1] > j) { buf[k] = buf[k - 1]; --k; }
            buf[k] = j;
            ++n;
        });
        for (int t = 0; t < nn; ++t) neigh_idx[(size_t)idx * nn + t] = (t < n) ? buf[t] : -1;
    }
}

// ---------------------------------------------------------------- charge
// update_charge, src/potential_solver_gpu.cu:12-63.  16 lanes per site.
__device__ __forceinline__ bool is_metal_dev(const int *__restrict__ metals, int num_metals, int e)
{
    for (int k = 0; k < num_metals; ++k)
        if (metals[k] == e) return true;
    return false;
}

__global__ __launch_bounds__(KMCF_BLOCK) void update_charge_kernel(
    const int *__restrict__ element, int *__restrict__ charge, const int *__restrict__ neigh_idx, int nn,
    const int *__restrict__ metals, int num_metals, int row_start, int row_count)
{
    constexpr int LPS = 16;
    const int lane = threadIdx.x % LPS;
    const int groups = (row_count + KMCF_BLOCK / LPS - 1) / (KMCF_BLOCK / LPS);
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int idx = grp * (KMCF_BLOCK / LPS) + threadIdx.x / LPS;
        const bool valid = idx < row_count;
        const int i = row_start + idx;
        const int e = valid ? element[i] : -1;
        int vnn = 0, metal_nb = 0;
        if (e == EL_VACANCY || e == EL_OXYGEN_DEFECT) {
            for (int t = lane; t < nn; t += LPS) {
                int nb = neigh_idx[(size_t)idx * nn + t];
                if (nb >= 0) {
                    int en = element[nb];
                    vnn += (en == EL_VACANCY);
                    metal_nb |= is_metal_dev(metals, num_metals, en) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int off = LPS / 2; off >= 1; off >>= 1) {
            vnn += __shfl_xor(vnn, off, 64);
            metal_nb |= __shfl_xor(metal_nb, off, 64);
        }
        if (valid && lane == 0) {
            if (e == EL_VACANCY) charge[i] = (metal_nb || vnn >= 2) ? 0 : 2;
            else if (e == EL_OXYGEN_DEFECT) charge[i] = metal_nb ? 0 : -2;
        }
    }
}

// ---------------------------------------------------------------- K values
// bit0: metal, bit1: uncharged vacancy  ->  G = high_G iff (cls_i & cls_j) != 0
// (calc_off_diagonal_dist, src/potential_solver_gpu.cu:263-280)
__global__ __launch_bounds__(KMCF_BLOCK) void site_class_kernel(const int *__restrict__ element,
                                                                const int *__restrict__ charge,
                                                                const int *__restrict__ metals, int num_metals,
                                                                int N, unsigned char *__restrict__ cls)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < N; s += gridDim.x * blockDim.x) {
        const int e = element[s];
        unsigned char c = is_metal_dev(metals, num_metals, e) ? 1 : 0;
        if (e == EL_VACANCY && charge[s] == 0) c |= 2;
        cls[s] = c;
    }
}

// One pass per row (LPR lanes): off-diagonal values, diagonal = sum of conductances to
// interface neighbours + left + right contact sums, 1/diag, rhs = left*VL + right*VR.
// Row sums are formed from integer counts (n_high*high_G + n_low*low_G): independent of
// lane order and of the rank count (the reference adds the values block by block in
// column order, src/potential_solver_gpu.cu:774-794 -- equal up to a few ulp).
// CB = true: conduction-band-edge Laplace system, G = high_G iff EITHER site is a metal
// (calc_off_diagonal_A_CB_gpu, src/potential_solver_gpu.cu:289-319).
template <bool CB>
__device__ __forceinline__ bool high_rule(unsigned char ci, unsigned char cj)
{
    return CB ? (((ci | cj) & 1) != 0) : ((ci & cj) != 0);
}

// Class of every internal column (own rows in internal order, then the halo slots): one gather per column
// here instead of two dependent gathers (permutation, class) per matrix entry in the assembly.
__global__ __launch_bounds__(KMCF_BLOCK) void cls_col_kernel(int n_cols, int n_loc, int row_site0, int n_left,
                                                             const int *__restrict__ perm, const int *__restrict__ halo_gid,
                                                             const unsigned char *__restrict__ cls,
                                                             unsigned char *__restrict__ cls_col)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += gridDim.x * blockDim.x) {
        const int site = (c < n_loc) ? (row_site0 + (perm ? perm[c] : c)) : (n_left + halo_gid[c - n_loc]);
        cls_col[c] = cls[site];
    }
}

template <int LPR, bool CB>
__global__ __launch_bounds__(KMCF_BLOCK) void k_assemble_kernel(
    int n_loc, int row_site0 /* N_left + displ */, int n_left, int n_interface,
    const int *__restrict__ row_ptr, const int *__restrict__ col, double *__restrict__ val,
    const int *__restrict__ diag_pos, const int *__restrict__ halo_gid,
    const int *__restrict__ perm /* internal -> caller local row, or nullptr */,
    const int *__restrict__ left_row_ptr, const int *__restrict__ left_col,
    const int *__restrict__ right_row_ptr, const int *__restrict__ right_col,
    const unsigned char *__restrict__ cls, double high_G, double low_G, double VL, double VR,
    double *__restrict__ diag_out, double *__restrict__ left_out, double *__restrict__ right_out,
    double *__restrict__ dinv_out, double *__restrict__ rhs_out,
    unsigned short *__restrict__ idx16 /* window SpMV: value codes above the slot bits, or nullptr */,
    double *__restrict__ diagv /* window SpMV: diagonal per row, or nullptr */,
    const unsigned char *__restrict__ cls_col /* class per internal column (cls_col_kernel) */)
{
    constexpr int SLOT_MASK = (1 << KMCF_SLOT_BITS) - 1;
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int groups = (n_loc + RPB - 1) / RPB;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int r = grp * RPB + threadIdx.x / LPR;
        const bool valid = r < n_loc;
        int nh = 0, nl = 0, lh = 0, ll = 0, rh = 0, rl = 0;
        int dpos = -1;
        if (valid) {
            const int ru = perm ? perm[r] : r;            // caller's local row (contact patterns)
            const unsigned char ci = cls_col[r];
            dpos = diag_pos[r];
            for (int j = row_ptr[r] + lane; j < row_ptr[r + 1]; j += LPR) {
                if (j == dpos) continue;
                const bool high = high_rule<CB>(ci, cls_col[col[j]]);
                val[j] = high ? -high_G : -low_G;
                if (idx16) idx16[j] = (unsigned short)((idx16[j] & SLOT_MASK) | ((high ? 0 : 1) << KMCF_SLOT_BITS));
                nh += high; nl += !high;
            }
            for (int j = left_row_ptr[ru] + lane; j < left_row_ptr[ru + 1]; j += LPR) {
                const bool high = high_rule<CB>(ci, cls[left_col[j]]);
                lh += high; ll += !high;
            }
            for (int j = right_row_ptr[ru] + lane; j < right_row_ptr[ru + 1]; j += LPR) {
                const bool high = high_rule<CB>(ci, cls[n_left + n_interface + right_col[j]]);
                rh += high; rl += !high;
            }
        }
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) {
            nh += __shfl_xor(nh, off, 64); nl += __shfl_xor(nl, off, 64);
            lh += __shfl_xor(lh, off, 64); ll += __shfl_xor(ll, off, 64);
            rh += __shfl_xor(rh, off, 64); rl += __shfl_xor(rl, off, 64);
        }
        if (valid && lane == 0) {
            const double d = (double)nh * high_G + (double)nl * low_G;
            const double l = (double)lh * high_G + (double)ll * low_G;
            const double rr = (double)rh * high_G + (double)rl * low_G;
            const double tot = d + l + rr;                 // insert_into_diag :807
            if (dpos >= 0) {
                val[dpos] = tot;
                if (idx16) idx16[dpos] = (unsigned short)((idx16[dpos] & SLOT_MASK) | (KMCF_CODE_DIAG << KMCF_SLOT_BITS));
            }
            if (diagv) diagv[r] = dpos >= 0 ? tot : 0.0;
            diag_out[r] = tot;
            left_out[r] = l;
            right_out[r] = rr;
            dinv_out[r] = 1.0 / (d + l + rr);              // inverse_diag :828
            rhs_out[r] = l * VL + rr * VR;                 // calc_rhs_for_A :452
        }
    }
}

// The same assembly over the tiles of the window SpMV (matrices planned for the coded kernel: U = 8 entries
// per lane, <= 64 rows and <= 1024 window columns per tile).  The row-per-16-lanes kernel above touches col,
// val and idx16 in 64..128-byte pieces and gathers a class per entry from global memory (426 us at 40 nm);
// here a tile's slot stream comes in as one 16-byte load per lane, the classes of its window columns are
// staged in LDS, the row lanes (4 per row) turn slots into codes inside LDS and count the conductances, and
// codes and values go out again as 16- and 64-byte pieces per lane.  Same values, same integer row sums.
template <bool CB>
__global__ __launch_bounds__(KMCF_BLOCK) void k_assemble_tile_kernel(
    int n_tiles, const int2 *__restrict__ tile, const int *__restrict__ row_ptr, const int *__restrict__ wcol,
    unsigned short *__restrict__ idx16, double *__restrict__ val, const int *__restrict__ diag_pos,
    const int *__restrict__ perm, const int *__restrict__ left_row_ptr, const int *__restrict__ left_col,
    const int *__restrict__ right_row_ptr, const int *__restrict__ right_col, const unsigned char *__restrict__ cls,
    const unsigned char *__restrict__ cls_col, int n_left, int n_interface, double high_G, double low_G, double VL,
    double VR, double *__restrict__ diag_out, double *__restrict__ left_out, double *__restrict__ right_out,
    double *__restrict__ dinv_out, double *__restrict__ rhs_out, double *__restrict__ diagv)
{
    constexpr int U = 8, LPR = 4, SLOT_MASK = (1 << KMCF_SLOT_BITS) - 1;   // KMCF_BLOCK / LPR = 64 rows per pass
    typedef unsigned int pack_t __attribute__((ext_vector_type(U / 2)));
    __shared__ unsigned char wcls[1 << KMCF_SLOT_BITS];
    __shared__ pack_t sidx_pk[KMCF_BLOCK];
    const int tid = threadIdx.x, lane = tid % LPR;
    unsigned short *sib = reinterpret_cast<unsigned short *>(sidx_pk);
    for (int c = blockIdx.x; c < n_tiles; c += gridDim.x) {
        const int2 t0 = tile[c], t1 = tile[c + 1];
        const int r0 = t0.x, r1 = t1.x, w0 = t0.y, W = t1.y - w0;
        const int base = row_ptr[r0], cnt = row_ptr[r1] - base;
        const int abase = base & ~(U - 1);                 // aligned start of the block-wide slot load
        for (int w = tid; w < W; w += KMCF_BLOCK) wcls[w] = cls_col[wcol[w0 + w]];
        sidx_pk[tid] = *reinterpret_cast<const pack_t *>(idx16 + abase + U * tid);
        __syncthreads();
        double tot = 0.0;
        int dpos = -1;
        {                                                  // tiles of this plan hold at most RPP rows: one pass
            const int r = r0 + tid / LPR;
            const bool valid = r < r1;
            int nh = 0, nl = 0, lh = 0, ll = 0, rh = 0, rl = 0;
            if (valid) {
                const int ru = perm ? perm[r] : r;        // caller's local row (contact patterns)
                const unsigned char ci = cls_col[r];
                dpos = diag_pos[r];
                for (int j = row_ptr[r] + lane; j < row_ptr[r + 1]; j += LPR) {
                    const int q = j - abase;
                    const int slot = sib[q] & SLOT_MASK;
                    if (j == dpos) { sib[q] = (unsigned short)(slot | (KMCF_CODE_DIAG << KMCF_SLOT_BITS)); continue; }
                    const bool high = high_rule<CB>(ci, wcls[slot]);
                    sib[q] = (unsigned short)(slot | ((high ? 0 : 1) << KMCF_SLOT_BITS));
                    nh += high; nl += !high;
                }
                for (int j = left_row_ptr[ru] + lane; j < left_row_ptr[ru + 1]; j += LPR) {
                    const bool high = high_rule<CB>(ci, cls[left_col[j]]);
                    lh += high; ll += !high;
                }
                for (int j = right_row_ptr[ru] + lane; j < right_row_ptr[ru + 1]; j += LPR) {
                    const bool high = high_rule<CB>(ci, cls[n_left + n_interface + right_col[j]]);
                    rh += high; rl += !high;
                }
            }
#pragma unroll
            for (int off = LPR / 2; off >= 1; off >>= 1) {
                nh += __shfl_xor(nh, off, 64); nl += __shfl_xor(nl, off, 64);
                lh += __shfl_xor(lh, off, 64); ll += __shfl_xor(ll, off, 64);
                rh += __shfl_xor(rh, off, 64); rl += __shfl_xor(rl, off, 64);
            }
            if (valid && lane == 0) {
                const double d = (double)nh * high_G + (double)nl * low_G;
                const double l = (double)lh * high_G + (double)ll * low_G;
                const double rr = (double)rh * high_G + (double)rl * low_G;
                tot = d + l + rr;                          // insert_into_diag :807
                diagv[r] = dpos >= 0 ? tot : 0.0;
                diag_out[r] = tot;
                left_out[r] = l;
                right_out[r] = rr;
                dinv_out[r] = 1.0 / (d + l + rr);          // inverse_diag :828
                rhs_out[r] = l * VL + rr * VR;             // calc_rhs_for_A :452
            } else {
                dpos = -1;
            }
        }
        __syncthreads();
        // codes and values back to global memory: lane t owns entries abase + 8 t .. + 7 (16 bytes of codes,
        // 64 bytes of values); the first and last lanes of a tile share that range with the neighbouring tiles
        // and write entry-wise.  Diagonal entries get a placeholder here and their value after the barrier.
        {
            const int q0 = U * tid, lo = base - abase, hi = lo + cnt;
            if (q0 + U > lo && q0 < hi) {
                const pack_t pk = sidx_pk[tid];
                const unsigned short *e = reinterpret_cast<const unsigned short *>(&pk);
                if (q0 >= lo && q0 + U <= hi) {
                    *reinterpret_cast<pack_t *>(idx16 + abase + q0) = pk;
                    double v[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) v[k] = (e[k] >> KMCF_SLOT_BITS) == 0 ? -high_G : -low_G;
                    double4 *vp = reinterpret_cast<double4 *>(val + abase + q0);
                    vp[0] = make_double4(v[0], v[1], v[2], v[3]);
                    vp[1] = make_double4(v[4], v[5], v[6], v[7]);
                } else {
                    for (int k = 0; k < U; ++k) {
                        const int q = q0 + k;
                        if (q < lo || q >= hi) continue;
                        idx16[abase + q] = e[k];
                        val[abase + q] = (e[k] >> KMCF_SLOT_BITS) == 0 ? -high_G : -low_G;
                    }
                }
            }
        }
        __syncthreads();                                   // orders the placeholder before the value; frees LDS
        if (lane == 0 && dpos >= 0) val[dpos] = tot;
    }
}

// sum_AB_into_A, src/potential_solver_gpu.cu:832-843
__global__ __launch_bounds__(KMCF_BLOCK) void sum_ab_kernel(double *__restrict__ A, const double *__restrict__ B, int N)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) A[i] += B[i];
}

// ---------------------------------------------------------------- heat
__device__ __forceinline__ double block_sum_h(double v, double *lds4)
{
    v = kmcf_wave_sum64(v);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(KMCF_BLOCK) void power_partial_kernel(const double *__restrict__ p, int N, double *__restrict__ part)
{
    __shared__ double lds4[4];
    double s = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) s += p[i];
    double t = block_sum_h(s, lds4);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// reduce + update_temp_global, src/heat_solver_gpu.cu:41-49
__global__ __launch_bounds__(KMCF_BLOCK) void temp_update_kernel(const double *__restrict__ part, int npart,
                                                                 double *__restrict__ T_bg, double a_coeff, double b_coeff,
                                                                 double number_steps, double C_thermal, double small_step)
{
    __shared__ double lds4[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += KMCF_BLOCK) s += part[i];
    double P_tot = block_sum_h(s, lds4);
    if (threadIdx.x == 0) {
        double c_coeff = b_coeff + P_tot / C_thermal * small_step;
        double T_intermediate = *T_bg;
        int step = (int)number_steps;
        *T_bg = c_coeff * (1.0 - pow(a_coeff, (double)step)) / (1.0 - a_coeff) + pow(a_coeff, (double)step) * T_intermediate;
    }
}

}  // namespace

extern "C" int kmcf_initialize_sparsity_K(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z,
                                          const double *h_lattice, int N, int pbc, double nn_dist, int N_contact,
                                          const int *h_counts, const int *h_displs, kmcf_kstate **out)
{
    KMCF_CHECK(c && d_x && d_y && d_z && h_lattice && h_counts && h_displs && out, KMCF_ERR_ARG,
               "kmcf_initialize_sparsity_K: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_initialize_sparsity_K: host-only communicator");
    KMCF_CHECK(N > 2 * N_contact && N_contact >= 0, KMCF_ERR_ARG, "kmcf_initialize_sparsity_K: N=%d, N_contact=%d", N, N_contact);
    KMCF_TRY(kmcf_enter(c));
    const int N_left = N_contact, N_right = N_contact;
    const int N_interface = N - (N_left + N_right);             // iterative_solvers_gpu.cu:271-273
    const int n_loc = h_counts[c->rank], disp = h_displs[c->rank];
    kmcf_kstate *k = new kmcf_kstate();
    k->comm = c; k->N = N; k->N_left = N_left; k->N_right = N_right; k->N_interface = N_interface;

    host_cells hc;
    KMCF_TRY(build_cells(d_x, d_y, d_z, N, h_lattice, pbc, nn_dist, &hc));
    int rc = coords_seen_by_kernels(hc, d_x, d_y, d_z, N, kmcf_setup_stream(c), "kmcf_initialize_sparsity_K");
    if (rc == KMCF_OK)
        rc = build_pattern(hc, d_x, d_y, d_z, h_lattice, pbc, nn_dist, N_left + disp, n_loc, N_left, N_left + N_interface,
                           &k->h_row_ptr, &k->h_col, kmcf_setup_stream(c));
    if (rc == KMCF_OK)
        rc = build_pattern(hc, d_x, d_y, d_z, h_lattice, pbc, nn_dist, N_left + disp, n_loc, 0, N_left,
                           &k->h_left_row_ptr, &k->h_left_col, kmcf_setup_stream(c));          // :449-461
    if (rc == KMCF_OK)
        rc = build_pattern(hc, d_x, d_y, d_z, h_lattice, pbc, nn_dist, N_left + disp, n_loc, N_left + N_interface, N,
                           &k->h_right_row_ptr, &k->h_right_col, kmcf_setup_stream(c));        // :463-474
    hc.release();
    if (rc != KMCF_OK) { delete k; return rc; }

    // Internal row order: this rank's sites sorted into bricks of edge KMCF_BRICK (default 7.7 A, about 60
    // sites) in lexicographic brick order (y, z, x), caller order inside a brick.  Rows that are processed
    // together then gather x from a few neighbouring bricks instead of a +-21 k-column band: measured
    // 161 -> 135 us for the 40 nm SpMV.  The caller never sees this order (vectors are permuted at the ABI).
    std::vector<int> perm;
    {
        double edge = 7.7;
        if (const char *e = getenv("KMCF_BRICK")) edge = atof(e);
        if (edge > 0 && n_loc > 1) {
            std::vector<double> sx(n_loc), sy(n_loc), sz(n_loc);
            const size_t off = (size_t)N_left + disp, bytes = (size_t)n_loc * sizeof(double);
            KMCF_HIP(hipMemcpy(sx.data(), d_x + off, bytes, hipMemcpyDeviceToHost));
            KMCF_HIP(hipMemcpy(sy.data(), d_y + off, bytes, hipMemcpyDeviceToHost));
            KMCF_HIP(hipMemcpy(sz.data(), d_z + off, bytes, hipMemcpyDeviceToHost));
            std::vector<int64_t> key((size_t)n_loc);
            for (int r = 0; r < n_loc; ++r) {
                const int64_t bx = (int64_t)std::floor(sx[r] / edge) + (1 << 19), by = (int64_t)std::floor(sy[r] / edge) + (1 << 19),
                              bz = (int64_t)std::floor(sz[r] / edge) + (1 << 19);
                key[r] = (by << 42) | (bz << 21) | bx;
            }
            perm.resize((size_t)n_loc);
            for (int r = 0; r < n_loc; ++r) perm[r] = r;
            // (sorting the rows of a brick by length, so that the 16 rows of a wavefront in the window SpMV are
            // equally long, was tried: no gain, 48.5 vs 49.7 us)
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key[a] < key[b]; });
        }
    }
    rc = kmcf_matrix_build(c, N_interface, h_counts, h_displs, k->h_row_ptr.data(), k->h_col.data(), nullptr,
                           perm.empty() ? nullptr : perm.data(), &k->K);
    if (rc != KMCF_OK) { delete k; return rc; }
    // position of the diagonal entry (insert_into_diag searches it every call, :795-814), in the
    // internal CSR: entries keep their order inside a row
    // (the matrix may have refined the order it was given: its own h_perm is the one in force)
    const std::vector<int> &perm_m = k->K->h_perm;
    std::vector<int> diag_pos((size_t)n_loc, -1);
    for (int i = 0; i < n_loc; ++i) {
        const int r = perm_m.empty() ? i : perm_m[i];
        for (int j = k->h_row_ptr[r]; j < k->h_row_ptr[r + 1]; ++j)
            if (k->h_col[j] == disp + r) { diag_pos[i] = k->K->h_row_ptr[i] + (j - k->h_row_ptr[r]); break; }
    }
    KMCF_TRY(upload(&k->d_diag_pos, diag_pos));
    KMCF_TRY(upload(&k->d_left_row_ptr, k->h_left_row_ptr));
    KMCF_TRY(upload(&k->d_left_col, k->h_left_col));
    KMCF_TRY(upload(&k->d_right_row_ptr, k->h_right_row_ptr));
    KMCF_TRY(upload(&k->d_right_col, k->h_right_col));
    const size_t nb = std::max(n_loc, 1) * sizeof(double);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_cls), std::max(N, 1)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_cls_col), (size_t)std::max(n_loc + k->K->n_halo, 1)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_diag), nb));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_left), nb));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_right), nb));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&k->d_rhs), nb));
    *out = k;
    return KMCF_OK;
}

extern "C" int kmcf_kstate_destroy(kmcf_kstate *k)
{
    if (!k) return KMCF_OK;
    if (k->comm && k->comm->device >= 0) {
        hipSetDevice(k->comm->device);
        hipStreamSynchronize(k->comm->stream);
        void *ptrs[] = {k->d_left_row_ptr, k->d_left_col, k->d_right_row_ptr, k->d_right_col, k->d_diag_pos, k->d_cls, k->d_cls_col,
                        k->d_diag, k->d_left, k->d_right, k->d_rhs, k->d_gather};
        for (void *p : ptrs)
            if (p) hipFree(p);
    }
    kmcf_matrix_destroy(k->K);
    delete k;
    return KMCF_OK;
}

extern "C" kmcf_matrix *kmcf_kstate_matrix(kmcf_kstate *k) { return k ? k->K : nullptr; }

extern "C" int kmcf_kstate_pattern(const kmcf_kstate *k, int which, int *h_row_ptr, int *h_col, int64_t *nnz)
{
    KMCF_CHECK(k && which >= 0 && which <= 2, KMCF_ERR_ARG, "kmcf_kstate_pattern: bad argument");
    const std::vector<int> &rp = which == 0 ? k->h_row_ptr : (which == 1 ? k->h_left_row_ptr : k->h_right_row_ptr);
    const std::vector<int> &cl = which == 0 ? k->h_col : (which == 1 ? k->h_left_col : k->h_right_col);
    if (nnz) *nnz = (int64_t)cl.size();
    if (h_row_ptr) memcpy(h_row_ptr, rp.data(), rp.size() * sizeof(int));
    if (h_col && !cl.empty()) memcpy(h_col, cl.data(), cl.size() * sizeof(int));
    return KMCF_OK;
}

extern "C" int kmcf_update_charge(kmcf_comm *c, const int *d_site_element, int *d_site_charge, const int *d_neigh_idx,
                                  int N, int nn, const int *d_metals, int num_metals, const int *h_count, const int *h_displ)
{
    KMCF_CHECK(c && d_site_element && d_site_charge && d_neigh_idx && d_metals && h_count && h_displ, KMCF_ERR_ARG,
               "kmcf_update_charge: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_update_charge: host-only communicator");
    KMCF_CHECK(nn > 0 && N >= 0, KMCF_ERR_ARG, "kmcf_update_charge: bad sizes");
    KMCF_TRY(kmcf_enter(c));
    const int count = h_count[c->rank], displ = h_displ[c->rank];
    KMCF_CHECK(displ >= 0 && displ + count <= N, KMCF_ERR_ARG, "kmcf_update_charge: rows [%d,%d) outside N=%d", displ, displ + count, N);
    if (count > 0) {
        update_charge_kernel<<<grid1d((int64_t)count * 16), KMCF_BLOCK, 0, c->stream>>>(
            d_site_element, d_site_charge, d_neigh_idx, nn, d_metals, num_metals, displ, count);
        KMCF_HIP(hipGetLastError());
    }
    KMCF_TRY(kmcf_comm_allgatherv_int(c, d_site_charge, h_count, h_displ));   // MPI_Allgatherv, :82-83
    KMCF_HIP(hipStreamSynchronize(c->stream));                                   // hipDeviceSynchronize, :80
    return kmcf_p2p_check(c);
}

#define KMCF_ASM_ARGS(VL, VR)                                                                                          \
    m->n_loc, k->N_left + m->row0, k->N_left, k->N_interface, m->d_row_ptr, m->d_col, m->d_val, k->d_diag_pos,          \
        m->d_halo_gid, m->d_perm, k->d_left_row_ptr, k->d_left_col, k->d_right_row_ptr, k->d_right_col, k->d_cls,      \
        high_G, low_G, VL, VR, k->d_diag, k->d_left, k->d_right, m->d_dinv, k->d_rhs, code_idx, code_diag, k->d_cls_col

static int k_assemble_async(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                            const int *d_metals, int num_metals, double Vd, double high_G, double low_G,
                            bool cb_rule = false)
{
    kmcf_comm *c = k->comm;
    kmcf_matrix *m = k->K;
    site_class_kernel<<<grid1d(k->N), KMCF_BLOCK, 0, c->stream>>>(d_site_element, d_site_charge, d_metals, num_metals, k->N, k->d_cls);
    KMCF_HIP(hipGetLastError());
    if (m->n_loc > 0) {
        constexpr int LPR = 16;
        const int grid = grid1d((int64_t)m->n_loc * LPR);
        const int n_cols = m->n_loc + m->n_halo;
        cls_col_kernel<<<grid1d(n_cols), KMCF_BLOCK, 0, c->stream>>>(n_cols, m->n_loc, k->N_left + m->row0, k->N_left, m->d_perm,
                                                                    m->d_halo_gid, k->d_cls, k->d_cls_col);
        // window SpMV: the off-diagonals are -high_G / -low_G, so the assembly writes their dictionary codes
        // next to the values and the CG's SpMV streams 2 B/nnz (kmcf_internal.hpp, kmcf_matrix::coded)
        const double dict[2] = {-high_G, -low_G};
        KMCF_TRY(kmcf_matrix_set_dictionary(m, dict, 2));
        unsigned short *code_idx = m->coded ? m->d_idx16 : nullptr;
        double *code_diag = m->coded ? m->d_diagv : nullptr;
        // (the tiles cover the short rows only: a matrix with long rows -- never K in practice -- is assembled row-wise)
        const bool tiled = m->coded && m->spmv_kind == 2 && m->spmv_u == 8 && m->tiles_for_coded && m->n_short == m->n_loc &&
                           !(getenv("KMCF_ASM_TILED") && atoi(getenv("KMCF_ASM_TILED")) == 0);
        if (tiled) {
            const int tgrid = std::min(m->n_tiles, 8 * 256 * 4);
#define KMCF_ASMT_ARGS(VL, VR)                                                                                          \
    m->n_tiles, m->d_tile, m->d_row_ptr, m->d_wcol, m->d_idx16, m->d_val, k->d_diag_pos, m->d_perm, k->d_left_row_ptr,   \
        k->d_left_col, k->d_right_row_ptr, k->d_right_col, k->d_cls, k->d_cls_col, k->N_left, k->N_interface, high_G,   \
        low_G, VL, VR, k->d_diag, k->d_left, k->d_right, m->d_dinv, k->d_rhs, m->d_diagv
            if (!cb_rule) k_assemble_tile_kernel<false><<<tgrid, KMCF_BLOCK, 0, c->stream>>>(KMCF_ASMT_ARGS(-Vd / 2, Vd / 2));
            else k_assemble_tile_kernel<true><<<tgrid, KMCF_BLOCK, 0, c->stream>>>(KMCF_ASMT_ARGS(Vd / 2, -Vd / 2));
#undef KMCF_ASMT_ARGS
        } else if (!cb_rule)
            k_assemble_kernel<LPR, false><<<grid, KMCF_BLOCK, 0, c->stream>>>(KMCF_ASM_ARGS(-Vd / 2, Vd / 2));   // :866-867
        else
            k_assemble_kernel<LPR, true><<<grid, KMCF_BLOCK, 0, c->stream>>>(KMCF_ASM_ARGS(Vd / 2, -Vd / 2));    // :697-698
        KMCF_HIP(hipGetLastError());
    }
    k->assembled = true;
    return KMCF_OK;
}

__global__ __launch_bounds__(KMCF_BLOCK) void cb_finish_kernel(double *__restrict__ cb, int N, int n_left, int n_interface,
                                                               double Vd, double eV_to_J)
{
    // boundary fill (src/potential_solver_gpu.cu:746-749) + hipblasDscal by eV_to_J (:752)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double v = cb[i];
        if (i < n_left) v = Vd / 2;
        else if (i >= n_left + n_interface) v = -Vd / 2;
        cb[i] = v * eV_to_J;
    }
}

int kmcf_scaled_cg_workspace(kmcf_matrix *m, double tol, int max_iterations, double *d_rhs_user, kmcf_solve_stats_t *stats);
int kmcf_jacobi_cg_workspace_absolute(kmcf_matrix *m, double tol, int max_iterations, kmcf_solve_stats_t *stats);

// update_CB_edge_gpu_sparse (src/potential_solver_gpu.cu:673-772): Laplace solve for the conduction-band
// edge on the K pattern (the reference rebuilds an identical single-GPU pattern, initialize_sparsity_CB),
// "either site metal" conductances, contacts at +Vd/2 / -Vd/2, solve_sparse_CG_Jacobi (tol 1e-14),
// result x eV_to_J.  Start guess = current content of site_CB_edge (:732).  Single rank, like the reference.
extern "C" int kmcf_update_CB_edge_sparse(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                                          const int *d_metals, int num_metals, double *d_site_CB_edge, int N,
                                          int N_left_tot, int N_right_tot, double Vd, double high_G, double low_G,
                                          kmcf_solve_stats_t *stats)
{
    KMCF_CHECK(k && d_site_element && d_site_charge && d_metals && d_site_CB_edge, KMCF_ERR_ARG, "kmcf_update_CB_edge_sparse: null argument");
    KMCF_CHECK(N == k->N && N_left_tot == k->N_left && N_right_tot == k->N_right, KMCF_ERR_ARG,
               "kmcf_update_CB_edge_sparse: N/N_left/N_right differ from the pattern's");
    kmcf_comm *c = k->comm;
    kmcf_matrix *m = k->K;
    KMCF_CHECK(c->nranks == 1, KMCF_ERR_ARG, "kmcf_update_CB_edge_sparse: single-rank solve (the reference runs it on one GPU)");
    KMCF_TRY(kmcf_enter(c));
    KMCF_TRY(k_assemble_async(k, d_site_element, d_site_charge, d_metals, num_metals, Vd, high_G, low_G, true));
    // (the matrix now holds the CB system: values, diag, rhs; the next kmcf_k_assemble refills K)
    KMCF_HIP(hipMemcpyAsync(m->d_r, k->d_rhs, (size_t)m->n_loc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double *v_soln = d_site_CB_edge + N_left_tot;
    KMCF_TRY(kmcf_vec_in(m, m->d_x, v_soln));
    // solve_sparse_CG_Jacobi's iteration in its Jacobi-PCG form (kmcf_cg.hip): the CB system is private to this call
    // (the reference builds and frees its own copy, :700-770), so nobody sees A scaled in place, and left unscaled it
    // keeps the two-conductance value codes the coded SpMV runs on.  KMCF_CB_SCALED=1: the literal scaled form.
    const char *e_sc = getenv("KMCF_CB_SCALED");
    if (m->coded && !(e_sc && atoi(e_sc) != 0))
        KMCF_TRY(kmcf_jacobi_cg_workspace_absolute(m, 1e-14 /* :719 */, 50000 /* warning threshold :860 */, stats));
    else
        KMCF_TRY(kmcf_scaled_cg_workspace(m, 1e-14 /* :719 */, 50000 /* warning threshold :860 */, nullptr, stats));
    KMCF_TRY(kmcf_vec_out(m, v_soln, m->d_x));
    cb_finish_kernel<<<grid1d(N), KMCF_BLOCK, 0, c->stream>>>(d_site_CB_edge, N, N_left_tot, k->N_interface, Vd, 1.60217663e-19);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_k_assemble(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                               const int *d_metals, int num_metals, double Vd, double high_G, double low_G)
{
    KMCF_CHECK(k && d_site_element && d_site_charge && d_metals, KMCF_ERR_ARG, "kmcf_k_assemble: null argument");
    KMCF_TRY(kmcf_enter(k->comm));
    KMCF_TRY(k_assemble_async(k, d_site_element, d_site_charge, d_metals, num_metals, Vd, high_G, low_G));
    KMCF_HIP(hipStreamSynchronize(k->comm->stream));
    return KMCF_OK;
}

extern "C" int kmcf_k_get_vectors(const kmcf_kstate *k, double *h_diag, double *h_dinv, double *h_rhs,
                                  double *h_left, double *h_right)
{
    KMCF_CHECK(k, KMCF_ERR_ARG, "kmcf_k_get_vectors: null state");
    KMCF_CHECK(k->assembled, KMCF_ERR_STATE, "kmcf_k_get_vectors: call kmcf_k_assemble first");
    KMCF_TRY(kmcf_enter(k->comm));
    KMCF_HIP(hipStreamSynchronize(k->comm->stream));
    const int n = k->K->n_loc;
    const size_t bytes = (size_t)n * sizeof(double);
    const std::vector<int> &perm = k->K->h_perm;   // device vectors are in the internal row order
    std::vector<double> tmp((size_t)n);
    const double *src[5] = {k->d_diag, k->K->d_dinv, k->d_rhs, k->d_left, k->d_right};
    double *dst[5] = {h_diag, h_dinv, h_rhs, h_left, h_right};
    for (int v = 0; v < 5; ++v) {
        if (!dst[v] || n == 0) continue;
        if (perm.empty()) {
            KMCF_HIP(hipMemcpy(dst[v], src[v], bytes, hipMemcpyDeviceToHost));
        } else {
            KMCF_HIP(hipMemcpy(tmp.data(), src[v], bytes, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) dst[v][perm[i]] = tmp[i];
        }
    }
    return KMCF_OK;
}

extern "C" int kmcf_background_potential_sparse(kmcf_kstate *k, const int *d_site_element, const int *d_site_charge,
                                                const int *d_metals, int num_metals, double *d_site_potential_boundary,
                                                int N, int N_left_tot, int N_right_tot, double Vd,
                                                double high_G, double low_G, kmcf_solve_stats_t *stats)
{
    KMCF_CHECK(k && d_site_element && d_site_charge && d_metals && d_site_potential_boundary, KMCF_ERR_ARG,
               "kmcf_background_potential_sparse: null argument");
    KMCF_CHECK(N == k->N && N_left_tot == k->N_left && N_right_tot == k->N_right, KMCF_ERR_ARG,
               "kmcf_background_potential_sparse: N/N_left/N_right (%d,%d,%d) differ from the pattern's (%d,%d,%d)",
               N, N_left_tot, N_right_tot, k->N, k->N_left, k->N_right);
    kmcf_comm *c = k->comm;
    kmcf_matrix *m = k->K;
    KMCF_CHECK(c->connected, KMCF_ERR_COMM, "kmcf_background_potential_sparse: communicator not connected");
    KMCF_TRY(kmcf_enter(c));
    hipEvent_t a0 = c->ev_a0, a1 = c->ev_a1;   // owned by the communicator: nothing to create or leak per call
    KMCF_HIP(hipEventRecord(a0, c->stream));
    KMCF_TRY(k_assemble_async(k, d_site_element, d_site_charge, d_metals, num_metals, Vd, high_G, low_G));
    KMCF_HIP(hipEventRecord(a1, c->stream));
    const size_t bytes = (size_t)m->n_loc * sizeof(double);
    // the initial guess is the current potential inside the device, solved in place (:861)
    double *v_soln = d_site_potential_boundary + N_left_tot + m->row0;
    const double relative_tolerance = 1e-14 * k->N_interface;   // :885
    const int max_iterations = 10000;                           // :886
    // a solve that runs as ONE resident launch (kmcf_cgr.hip) takes the right-hand side where the assembly left it and
    // the start guess / solution in the caller's array: no copy, no permuting kernel around it
    // (asked on EVERY rank, also one without rows: a group agrees on the resident launch through a collective the first time)
    const bool direct = kmcf_pcg_resident_applies(m) && m->n_loc > 0;
    if (direct) { m->solve_b_src = k->d_rhs; m->solve_x_user = v_soln; }
    else {
        KMCF_HIP(hipMemcpyAsync(m->d_r, k->d_rhs, bytes, hipMemcpyDeviceToDevice, c->stream));   // internal order already
        KMCF_TRY(kmcf_vec_in(m, m->d_x, v_soln));
    }
    const int rc_solve = kmcf_pcg_workspace(m, true, relative_tolerance, max_iterations, 0, stats);
    m->solve_b_src = nullptr; m->solve_x_user = nullptr;
    KMCF_TRY(rc_solve);
    if (!direct) KMCF_TRY(kmcf_vec_out(m, v_soln, m->d_x));
    KMCF_HIP(hipStreamSynchronize(c->stream));
    if (stats) {
        float ms = 0.f;
        KMCF_HIP(hipEventElapsedTime(&ms, a0, a1));
        stats->ms_assembly = ms;
    }
    return KMCF_OK;
}

extern "C" int kmcf_sum_and_gather_potential(kmcf_kstate *k, double *d_site_potential_boundary,
                                             double *d_site_potential_charge, int N, int num_atoms_first_layer,
                                             const int *h_counts_pairwise, const int *h_displs_pairwise)
{
    KMCF_CHECK(k && d_site_potential_boundary && d_site_potential_charge, KMCF_ERR_ARG, "kmcf_sum_and_gather_potential: null argument");
    KMCF_CHECK(N == k->N && num_atoms_first_layer == k->N_left, KMCF_ERR_ARG, "kmcf_sum_and_gather_potential: size mismatch");
    kmcf_comm *c = k->comm;
    KMCF_TRY(kmcf_enter(c));
    // pairwise term: every rank computed its rows (src/kmc_main.cpp:405-425, potential_solver_gpu.cu:1139-1142)
    if (h_counts_pairwise && h_displs_pairwise)
        KMCF_TRY(kmcf_comm_allgatherv_double(c, d_site_potential_charge, h_counts_pairwise, h_displs_pairwise));
    // MPI_Gatherv to rank 0 (src/kmc_main.cpp:367-384) + MPI_Bcast (potential_solver_gpu.cu:1133-1136)
    KMCF_TRY(kmcf_comm_allgatherv_double(c, d_site_potential_boundary + num_atoms_first_layer,
                                         k->K->counts.data(), k->K->displs.data()));
    sum_ab_kernel<<<grid1d(N), KMCF_BLOCK, 0, c->stream>>>(d_site_potential_charge, d_site_potential_boundary, N);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return kmcf_p2p_check(c);
}

extern "C" int kmcf_update_temperature_global(kmcf_comm *c, const double *d_site_power, double *d_T_bg, int N,
                                              double a_coeff, double b_coeff, double number_steps,
                                              double C_thermal, double small_step)
{
    KMCF_CHECK(c && d_site_power && d_T_bg && N >= 0, KMCF_ERR_ARG, "kmcf_update_temperature_global: bad argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_update_temperature_global: host-only communicator");
    KMCF_TRY(kmcf_enter(c));
    double *d_part = c->d_scratch;            // persistent: the reference allocates nothing per call either
    const int g = grid1d(N, 1024);
    power_partial_kernel<<<g, KMCF_BLOCK, 0, c->stream>>>(d_site_power, N, d_part);
    KMCF_HIP(hipGetLastError());
    temp_update_kernel<<<1, KMCF_BLOCK, 0, c->stream>>>(d_part, g, d_T_bg, a_coeff, b_coeff, number_steps, C_thermal, small_step);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_neighbor_list(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z, int N,
                                  double nn_dist, int nn, int count, int displ, int *d_neigh_idx)
{
    KMCF_CHECK(c && d_x && d_y && d_z && d_neigh_idx, KMCF_ERR_ARG, "kmcf_neighbor_list: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_neighbor_list: host-only communicator");
    KMCF_CHECK(nn > 0 && nn <= NL_CAP && count >= 0 && displ >= 0 && displ + count <= N, KMCF_ERR_ARG, "kmcf_neighbor_list: bad sizes");
    KMCF_TRY(kmcf_enter(c));
    host_cells hc;
    const double lattice[3] = {1, 1, 1};
    KMCF_TRY(build_cells(d_x, d_y, d_z, N, lattice, 0, nn_dist, &hc));
    {
        const int rcv = coords_seen_by_kernels(hc, d_x, d_y, d_z, N, kmcf_setup_stream(c), "kmcf_neighbor_list");
        if (rcv != KMCF_OK) { hc.release(); return rcv; }
    }
    int *d_over = nullptr;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_over), sizeof(int)));
    hipStream_t ss = kmcf_setup_stream(c);                       // (the caller's stream: see kmcf_internal.hpp)
    KMCF_HIP(hipMemsetAsync(d_over, 0, sizeof(int), ss));
    if (count > 0) {
        neighbor_list_kernel<<<grid1d(count, 1 << 20), KMCF_BLOCK, 0, ss>>>(
            hc.g, hc.d_cell_start, hc.d_cell_items, d_x, d_y, d_z, nn_dist, N, nn, count, displ, d_neigh_idx, d_over);
        KMCF_HIP(hipGetLastError());
    }
    int over = 0;
    KMCF_HIP(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, ss));
    KMCF_HIP(hipStreamSynchronize(ss));
    hipFree(d_over);
    hc.release();
    KMCF_CHECK(!over, KMCF_ERR_ARG, "kmcf_neighbor_list: a site has more than %d neighbours within nn_dist", NL_CAP);
    return KMCF_OK;
}
