// T path: the current solve of DeviceKMC (Kirchhoff matrix over the atoms + two virtual nodes, WKB
// tunnelling sub-block), SURVEY.md 8 rows a14 / f3.  PARITY UNPINNED by any reference fixture.
//
// Reference -> here:
//   initialize_sparsity_T: O(n_loc * Nsub) distance scans per column block
//   (src/initialize_sparsity_T.cu:10-209, 948-1154)            -> cell list, once per bias point.
//   update_atom_arrays: 7 thrust::copy_if per step (src/current_solver_gpu.cu:1341-1365)
//                                                             -> the atom set is invariant under KMC events
//       (O <-> V, d <-> Od): site indices kept from init, one gather kernel per step, invariance checked.
//   populate_T_dist per block + calc_diagonal_T per block + insert_diag_T (:1051-1321)
//                                                             -> ONE kernel, 16 lanes per row, integer row sums;
//       the off-diagonals are -high_G / -low_G / -loop_G, so the coded window SpMV applies (2 B/nnz); the two
//       virtual-node rows (one entry per contact atom of a layer) are "long rows" (kmcf_internal.hpp).
//   assemble_sparse_T_submatrix: pattern by a thread per row scanning all columns twice, csr2coo, values by a
//   thread per entry with four-fold indirection, rebuilt and re-allocated every step (:707-946)
//                                                             -> the block is dense-ish (36 % at 5 nm, 43 % in the
//       authors' test set), so it is kept as a BITMAP (one 64-bit mask per 64 columns and row, found by a wave
//       per row with one ballot per 64 pairs) + packed f64 values: 8 B/nnz + 1 bit per position instead of the
//       12 B/nnz of CSR; buffers grow only.
//   spmm_split_sparse1: pack, size-1 Isend/Irecv ring of the sub-vector, neighbour SpMV, rocsparse_spmv of the
//   sub-block, unpack_add (dist_iterative/dist_spmv_split_sparse.cpp:5-78)
//                                                             -> pack, all-gather, neighbour SpMV, one bitmap kernel
//       (wave per row, coalesced value stream, x_sub from L2) adding into Ap and into the p.Ap partials.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "kmcf_cells.hpp"
#include "kmcf_internal.hpp"

int kmcf_pcg_workspace(kmcf_matrix *m, bool precond, double tol, int max_it, int fixed_iters, kmcf_solve_stats_t *stats);

struct kmcf_tstate {
    kmcf_comm *comm = nullptr;
    kmcf_matrix *T = nullptr;
    int N = 0, N_atom = 0, Nsub = 0, n_inj = 0, n_ext = 0, n_layers = 0;
    double nn_dist = 0.0;
    std::vector<int> h_atom_site;
    std::vector<int> h_row_ptr, h_col;             // pattern of this rank, global columns, caller row order
    std::vector<int> h_inv_perm;                   // caller local row -> internal row
    // per atom (device)
    int *d_atom_site = nullptr;
    unsigned char *d_site_is_atom = nullptr;       // N: the atom set at init (invariance check)
    double *d_ax = nullptr, *d_ay = nullptr, *d_az = nullptr, *d_acb = nullptr;
    int *d_ael = nullptr, *d_ach = nullptr;
    unsigned char *d_acls = nullptr;               // bit0 metal, bit1 uncharged vacancy
    // per internal row / column of T
    unsigned char *d_cls_col = nullptr;            // 0x80 | node for the virtual nodes, else the atom's class
    int *d_col_node = nullptr;                     // global node of every internal column (own rows | halo slots)
    int *d_diag_pos = nullptr;
    unsigned char *d_ground = nullptr;             // row's atom neighbours the last atom (the cut ground node)
    int *d_inv_perm = nullptr;
    double *d_diag = nullptr, *d_diag_tot = nullptr, *d_rhs = nullptr;
    // tunnel points (all ranks' points on every rank: site arrays are replicated)
    int *d_tflag = nullptr, *d_blk = nullptr, *d_tidx = nullptr, *d_tinfo = nullptr;
    double *d_tx = nullptr, *d_ty = nullptr, *d_tz = nullptr, *d_tcb = nullptr;
    int *d_rowcnt = nullptr;
    double *d_tdiag = nullptr;
    size_t cap_rowcnt = 0, cap_tdiag = 0;
    std::vector<int> h_tidx;
    kmcf_subop sub;
    // post-processing
    double *d_pdisp = nullptr;                     // Nsub
    double *d_scal = nullptr;                      // [0] imacro, [1] min
    int *d_err = nullptr;
    int *h_pin = nullptr;                          // pinned: [0] n_t, [1] err, [2..3] nnz (long long)
    double *d_agree = nullptr, *h_agree = nullptr; // storage vote of a rank group: 2 P on the device; pinned 2 + 2 P
    bool assembled = false;
    kmcf_current_params_t par{};
};

namespace {

constexpr int EL_DEFECT = 0, EL_OXYGEN_DEFECT = 1, EL_VACANCY = 2, EL_TI = 6, EL_N = 8;   // src/utils.h:37-44
constexpr double H_BAR = 1.054571817e-34;        // src/initialize_sparsity_T.cu:6
constexpr double EV_TO_J = 1.60217663e-19;       // :5
constexpr int CODE_HIGH = 0, CODE_LOW = 1, CODE_LOOP = 2;

__device__ __forceinline__ bool in_metals(const int *__restrict__ metals, int nm, int e)
{
    for (int k = 0; k < nm; ++k)
        if (metals[k] == e) return true;
    return false;
}

__device__ __forceinline__ double dist3(double x1, double y1, double z1, double x2, double y2, double z2)
{   // site_dist_gpu, 6-argument overload (src/gpu_solvers.h:280-285)
    const double dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

__device__ __forceinline__ double wave_sum(double v) { return kmcf_wave_sum64(v); }

__device__ __forceinline__ double block_sum4(double v, double *lds4)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------- atoms
__global__ __launch_bounds__(KMCF_BLOCK) void gather_coords_kernel(int n, const int *__restrict__ site, const double *__restrict__ sx,
                                                                   const double *__restrict__ sy, const double *__restrict__ sz,
                                                                   double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az)
{
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
        const int s = site[a];
        ax[a] = sx[s]; ay[a] = sy[s]; az[a] = sz[s];
    }
}

// is_defect filter (src/gpu_solvers.h:331-337) against the set found at init
__global__ __launch_bounds__(KMCF_BLOCK) void check_atom_set_kernel(int N, const int *__restrict__ element,
                                                                    const unsigned char *__restrict__ is_atom, int *__restrict__ err)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < N; s += gridDim.x * blockDim.x) {
        const int e = element[s];
        const bool atom = (e != EL_DEFECT) && (e != EL_OXYGEN_DEFECT);
        if (atom != (is_atom[s] != 0)) *err = 1;
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void gather_atoms_kernel(int n, const int *__restrict__ site, const int *__restrict__ element,
                                                                  const int *__restrict__ charge, const double *__restrict__ cb,
                                                                  const int *__restrict__ metals, int nm, int *__restrict__ ael,
                                                                  int *__restrict__ ach, double *__restrict__ acb,
                                                                  unsigned char *__restrict__ acls)
{
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
        const int s = site[a];
        const int e = element[s], q = charge[s];
        ael[a] = e; ach[a] = q; acb[a] = cb[s];
        unsigned char c = in_metals(metals, nm, e) ? 1 : 0;
        if (e == EL_VACANCY && q == 0) c |= 2;                 // conductive vacancy (:1231-1232)
        acls[a] = c;
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void t_cls_col_kernel(int n_cols, const int *__restrict__ col_node,
                                                               const unsigned char *__restrict__ acls, unsigned char *__restrict__ cls_col)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += gridDim.x * blockDim.x) {
        const int node = col_node[c];
        cls_col[c] = node < 2 ? (unsigned char)(0x80 | node) : acls[node - 2];
    }
}

// ---------------------------------------------------------------- neighbour values
// populate_T_dist (src/current_solver_gpu.cu:1051-1247) + calc_diagonal_T + insert_diag_T (:1279-1321), one
// pass, LPR lanes per row.  Entry rule by the classes of row and column node:
//   virtual row, virtual column (0,1)/(1,0): -loop_G;  virtual row, atom column: -high_G (:1082-1090, :1097-1106)
//   atom row, virtual column: -high_G (:1126-1136);  atom-atom neighbours: -high_G if both metal or both
//   uncharged vacancies, else -low_G (:1224-1241).
// Diagonal = start value + sum of the row's conductances: the start value is what populate_T_dist leaves in the
// diagonal slot -- +high_G in row 0 (:1077-1080) and in rows of atoms that neighbour the cut ground atom
// (:1113-1123), 0 in row 1 and elsewhere.  Row sums from integer counts (order independent).
template <int LPR>
__global__ __launch_bounds__(KMCF_BLOCK) void t_assemble_kernel(
    int n_loc, const int *__restrict__ row_ptr, const int *__restrict__ col, double *__restrict__ val,
    const int *__restrict__ diag_pos, const unsigned char *__restrict__ ground, const unsigned char *__restrict__ cls_col,
    double high_G, double low_G, double loop_G, double *__restrict__ diag_out,
    unsigned short *__restrict__ idx16 /* value codes above the slot bits (rows < n_coded), or nullptr */,
    double *__restrict__ diagv, int n_coded)
{
    constexpr int SLOT_MASK = (1 << KMCF_SLOT_BITS) - 1;
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int groups = (n_loc + RPB - 1) / RPB;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int r = grp * RPB + threadIdx.x / LPR;
        const bool valid = r < n_loc;
        int n_high = 0, n_low = 0, n_loop = 0, dpos = -1;
        unsigned char ci = 0;
        if (valid) {
            ci = cls_col[r];
            dpos = diag_pos[r];
            const bool coded = idx16 && r < n_coded;
            for (int j = row_ptr[r] + lane; j < row_ptr[r + 1]; j += LPR) {
                if (j == dpos) continue;
                const unsigned char cj = cls_col[col[j]];
                int code;
                if (ci & 0x80) code = (cj & 0x80) ? CODE_LOOP : CODE_HIGH;
                else if (cj & 0x80) code = CODE_HIGH;
                else code = (ci & cj & 3) ? CODE_HIGH : CODE_LOW;
                val[j] = code == CODE_HIGH ? -high_G : (code == CODE_LOW ? -low_G : -loop_G);
                if (coded) idx16[j] = (unsigned short)((idx16[j] & SLOT_MASK) | (code << KMCF_SLOT_BITS));
                n_high += code == CODE_HIGH; n_low += code == CODE_LOW; n_loop += code == CODE_LOOP;
            }
        }
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) {
            n_high += __shfl_xor(n_high, off, 64); n_low += __shfl_xor(n_low, off, 64); n_loop += __shfl_xor(n_loop, off, 64);
        }
        if (valid && lane == 0) {
            double d0 = 0.0;
            if (ci == 0x80) d0 = high_G;
            else if (!(ci & 0x80) && ground[r]) d0 = high_G;
            const double off = (double)n_high * high_G + (double)n_low * low_G + (double)n_loop * loop_G;
            const double d = d0 + off;
            if (dpos >= 0) {
                val[dpos] = d;
                if (idx16 && r < n_coded) idx16[dpos] = (unsigned short)((idx16[dpos] & SLOT_MASK) | (KMCF_CODE_DIAG << KMCF_SLOT_BITS));
            }
            if (diagv && r < n_coded) diagv[r] = dpos >= 0 ? d : 0.0;
            diag_out[r] = d;
        }
    }
}

// ---------------------------------------------------------------- tunnel points
// get_is_tunnel_mpi (src/initialize_sparsity_T.cu:618-654) over the atoms 0 .. N_atom - 2 of all ranks; the
// reference's yes * idx / copy_if(is_not_zero) drops atom 0, restated as "atom 0 is never a tunnel point".
constexpr int SCAN_ITEMS = 8;    // per thread: 2048 atoms per block

__device__ __forceinline__ int tunnel_flag(int a, int n_scan, const int *__restrict__ ael, const double *__restrict__ ax,
                                           double x_lo, double x_hi)
{
    if (a < 1 || a >= n_scan) return 0;
    const int e = ael[a];
    return (e == EL_VACANCY || ((e == EL_TI || e == EL_N) && (ax[a] > x_lo && ax[a] < x_hi))) ? 1 : 0;
}

__global__ __launch_bounds__(KMCF_BLOCK) void tunnel_count_kernel(int n_scan, const int *__restrict__ ael, const double *__restrict__ ax,
                                                                  double x_lo, double x_hi, int *__restrict__ blk)
{
    __shared__ int lds4[4];
    const int base = blockIdx.x * KMCF_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    int c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) c += tunnel_flag(base + k, n_scan, ael, ax, x_lo, x_hi);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// one block: exclusive scan of up to any number of ints (sequential over 256-wide segments)
template <typename T>
__global__ __launch_bounds__(KMCF_BLOCK) void scan_exclusive_kernel(int n, const int *__restrict__ in, T *__restrict__ out /* n + 1 */,
                                                                    T *__restrict__ total_pinned)
{
    __shared__ T sh[KMCF_BLOCK];
    __shared__ T carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int s0 = 0; s0 < n; s0 += KMCF_BLOCK) {
        const int i = s0 + threadIdx.x;
        const T v = i < n ? (T)in[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < KMCF_BLOCK; off <<= 1) {          // Hillis-Steele inclusive scan
            const T t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) out[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[KMCF_BLOCK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[n] = carry;
        if (total_pinned) *total_pinned = carry;
    }
}

// scatter: tunnel point list (ascending atom index) + compact per-point data
// info: bit0 vacancy, bit1 "metal_p" (a metal outside the outer contact layers,
// src/initialize_sparsity_T.cu:267-273; num_metals hard-coded to 2 at :800 -- the caller's list is used)
__global__ __launch_bounds__(KMCF_BLOCK) void tunnel_scatter_kernel(
    int n_scan, int N_atom, const int *__restrict__ ael, const double *__restrict__ ax, const double *__restrict__ ay,
    const double *__restrict__ az, const double *__restrict__ acb, const int *__restrict__ metals, int nm,
    double x_lo, double x_hi, int n_layers, int n_inj, int n_ext, const int *__restrict__ blk_off,
    int *__restrict__ tidx, int *__restrict__ tinfo, double *__restrict__ tx, double *__restrict__ ty,
    double *__restrict__ tz, double *__restrict__ tcb)
{
    __shared__ int sh[KMCF_BLOCK];
    const int base = blockIdx.x * KMCF_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS], c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { f[k] = tunnel_flag(base + k, n_scan, ael, ax, x_lo, x_hi); c += f[k]; }
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < KMCF_BLOCK; off <<= 1) {
        const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int pos = blk_off[blockIdx.x] + sh[threadIdx.x] - c;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (!f[k]) continue;
        const int a = base + k;
        const int e = ael[a];
        int info = (e == EL_VACANCY) ? 1 : 0;
        if (in_metals(metals, nm, e) && (a > (n_layers - 1) * n_inj) && (a < (N_atom - (n_layers - 1) * n_ext))) info |= 2;
        tidx[pos] = a; tinfo[pos] = info;
        tx[pos] = ax[a]; ty[pos] = ay[a]; tz[pos] = az[a]; tcb[pos] = acb[a];
        ++pos;
    }
}

// pair rule of the tunnel pattern (src/initialize_sparsity_T.cu:261-284)
__device__ __forceinline__ bool tunnel_pair(int info_i, int info_j, double cb_i, double cb_j, double tol, bool *c2t)
{
    const bool v1 = info_i & 1, v2 = info_j & 1, m1 = info_i & 2, m2 = info_j & 2;
    const bool t2t = v1 && v2, ct = (v1 && m2) || (v2 && m1), cc = m1 && m2;
    *c2t = ct;
    return (t2t || ct || cc) && (fabs(cb_i - cb_j) > tol);
}

// One wave per local sub row: masks of its row (one ballot per 64 columns) and its entry count
// (calc_nnz_per_row_tunnel + assemble_tunnel_col_indices, :212-372).
__global__ __launch_bounds__(KMCF_BLOCK) void tunnel_mask_kernel(
    int n_loc, int s0, int n_glob, int n_groups, const int *__restrict__ tinfo, const double *__restrict__ tx,
    const double *__restrict__ ty, const double *__restrict__ tz, const double *__restrict__ tcb, double nn_dist, double tol,
    unsigned long long *__restrict__ mask, int *__restrict__ rowcnt)
{
    const int lane = threadIdx.x & 63;
    const int wpb = KMCF_BLOCK / 64;
    for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < n_loc; s += gridDim.x * wpb) {
        const int sg = s0 + s;
        const double xi = tx[sg], yi = ty[sg], zi = tz[sg], cbi = tcb[sg];
        const int fi = tinfo[sg];
        int cnt = 0;
        for (int g = 0; g < n_groups; ++g) {
            const int j = g * 64 + lane;
            bool take = false;
            if (j < n_glob) {
                if (j == sg) take = true;                                           // :255-258 diagonal
                else {
                    bool c2t;
                    const double d = dist3(xi, yi, zi, tx[j], ty[j], tz[j]);
                    take = d > nn_dist && tunnel_pair(fi, tinfo[j], cbi, tcb[j], tol, &c2t);   // :261-284
                }
            }
            const unsigned long long mk = __ballot(take);
            if (lane == 0) mask[(size_t)s * n_groups + g] = mk;
            cnt += __popcll(mk);
        }
        if (lane == 0) rowcnt[s] = cnt;
    }
}

// WKB value of an entry (populate_T_tunnel_dist2, src/initialize_sparsity_T.cu:539-611)
__device__ __forceinline__ double wkb_value(double dist_angstrom, double cb_i, double cb_j, bool c2t, double m_e, double V0)
{
    const double local_E_drop = cb_i - cb_j;
    const double prefac = -(sqrt(2 * m_e) / H_BAR) * (2.0 / 3.0);
    const double dist = (1e-10) * dist_angstrom;
    if (c2t) {
        const double energy_window = fabs(local_E_drop);
        const double dV = 0.01;
        const double dE = EV_TO_J * dV * 10000000000;      // :572: the loop below runs once for any physical window
        double T = 0.0;
        for (double iv = 0; iv < energy_window; iv += dE) {
            const double E1 = EV_TO_J * V0 + iv;
            const double E2 = E1 - fabs(local_E_drop);
            if (E2 > 0) T += exp(prefac * (dist / fabs(local_E_drop)) * (pow(E1, 1.5) - pow(E2, 1.5)));
            if (E2 < 0) T += exp(prefac * (dist / fabs(local_E_drop)) * (pow(E1, 1.5)));
        }
        return -T;
    }
    const double E1 = EV_TO_J * V0;
    const double E2 = E1 - fabs(local_E_drop);
    if (E2 > 0) return -exp(prefac * (dist / fabs(E1 - E2)) * (pow(E1, 1.5) - pow(E2, 1.5)));
    if (E2 < 0) return -exp(prefac * (dist / fabs(E1 - E2)) * (pow(E1, 1.5)));
    return 0.0;      // E2 == 0: written by neither branch there (uninitialised memory, :881)
}

// One wave per local sub row: values of the set positions + the row's diagonal = -(sum of the others)
// (populate_T_tunnel_dist2 + calc_diagonal_T_tunnel, :497-614, :669-689).
__global__ __launch_bounds__(KMCF_BLOCK) void tunnel_value_kernel(
    int n_loc, int s0, int n_groups, const int *__restrict__ tinfo, const double *__restrict__ tx, const double *__restrict__ ty,
    const double *__restrict__ tz, const double *__restrict__ tcb, double nn_dist, double tol, double m_e, double V0,
    const unsigned long long *__restrict__ mask, const long long *__restrict__ voff, double *__restrict__ val,
    double *__restrict__ tdiag)
{
    const int lane = threadIdx.x & 63;
    const int wpb = KMCF_BLOCK / 64;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < n_loc; s += gridDim.x * wpb) {
        const int sg = s0 + s;
        const double xi = tx[sg], yi = ty[sg], zi = tz[sg], cbi = tcb[sg];
        const int fi = tinfo[sg];
        long long off = voff[s];
        long long dpos = -1;
        double rowsum = 0.0;
        for (int g = 0; g < n_groups; ++g) {
            const unsigned long long mk = mask[(size_t)s * n_groups + g];
            if (mk == 0ull) continue;                                      // wave-uniform
            if ((mk >> lane) & 1ull) {
                const int j = g * 64 + lane;
                const long long pos = off + __popcll(mk & lt);
                if (j == sg) dpos = pos;
                else {
                    bool c2t;
                    tunnel_pair(fi, tinfo[j], cbi, tcb[j], tol, &c2t);
                    const double v = wkb_value(dist3(xi, yi, zi, tx[j], ty[j], tz[j]), cbi, tcb[j], c2t, m_e, V0);
                    val[pos] = v;
                    rowsum += v;
                }
            }
            off += __popcll(mk);
        }
        rowsum = wave_sum(rowsum);
        if (dpos >= 0) val[dpos] = -rowsum;                                // :685
        if (lane == 0) tdiag[s] = -rowsum;                                 // :680
    }
}

// assemble_preconditioner + invert_diag (src/current_solver_gpu.cu:1323-1338), rhs (:1627-1632)
__global__ __launch_bounds__(KMCF_BLOCK) void copy_diag_rhs_kernel(int n, const double *__restrict__ diag, double *__restrict__ tot,
                                                                   const int *__restrict__ col_node, double loop_G_Vd,
                                                                   double *__restrict__ rhs)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        tot[i] = diag[i];
        const int node = col_node[i];
        rhs[i] = node == 0 ? -loop_G_Vd : (node == 1 ? loop_G_Vd : 0.0);
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void add_tunnel_diag_kernel(int n_sub, const int *__restrict__ rows,
                                                                     const double *__restrict__ tdiag, double *__restrict__ tot)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_sub; s += gridDim.x * blockDim.x) tot[rows[s]] += tdiag[s];
}

__global__ __launch_bounds__(KMCF_BLOCK) void invert_kernel(int n, const double *__restrict__ tot, double *__restrict__ dinv)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dinv[i] = 1 / tot[i];
}

__global__ __launch_bounds__(KMCF_BLOCK) void sub_rows_kernel(int n_sub, int s0, const int *__restrict__ tidx, int row0,
                                                              const int *__restrict__ inv_perm, int *__restrict__ rows)
{
    // shift_vector_by_constant (src/initialize_sparsity_T.cu:904): local row = atom index + 2 - displacement
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_sub; s += gridDim.x * blockDim.x)
        rows[s] = inv_perm[tidx[s0 + s] + 2 - row0];
}

// ---------------------------------------------------------------- sub-block operator
__global__ __launch_bounds__(KMCF_BLOCK) void sub_pack_kernel(int n_sub, const int *__restrict__ rows, const double *__restrict__ p,
                                                              double *__restrict__ xsub_loc, const kmcf_scalars *__restrict__ S, int check_done)
{
    if (check_done && S->done) return;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_sub; s += gridDim.x * blockDim.x) xsub_loc[s] = p[rows[s]];
}

// y[rows[s]] += S[s, :] x_sub: a wave per row walks the row's masks; the lanes of set positions read their
// value from the packed stream (consecutive lanes -> consecutive values) and x_sub[64 g + lane] (coalesced,
// L2 resident: n_glob doubles).  Four mask words are fetched per step so that four value loads are in flight.
template <bool DOT>
__global__ __launch_bounds__(KMCF_BLOCK) void sub_spmv_kernel(
    int n_loc, int n_glob, int n_groups, const unsigned long long *__restrict__ mask, const long long *__restrict__ voff,
    const double *__restrict__ val, const double *__restrict__ xsub, const int *__restrict__ rows,
    const double *__restrict__ p, double *__restrict__ y, double *__restrict__ part, const kmcf_scalars *__restrict__ S,
    int check_done)
{
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    const int lane = threadIdx.x & 63;
    const int wpb = KMCF_BLOCK / 64;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    double dot = 0.0;
    for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < n_loc; s += gridDim.x * wpb) {
        const unsigned long long *mrow = mask + (size_t)s * n_groups;
        long long off = voff[s];
        double acc = 0.0;
        int g = 0;
        for (; g + 4 <= n_groups; g += 4) {
            const unsigned long long m0 = mrow[g], m1 = mrow[g + 1], m2 = mrow[g + 2], m3 = mrow[g + 3];
            const long long o1 = off + __popcll(m0), o2 = o1 + __popcll(m1), o3 = o2 + __popcll(m2);
            const bool b0 = (m0 >> lane) & 1ull, b1 = (m1 >> lane) & 1ull, b2 = (m2 >> lane) & 1ull, b3 = (m3 >> lane) & 1ull;
            const double v0 = b0 ? __builtin_nontemporal_load(val + off + __popcll(m0 & lt)) : 0.0;
            const double v1 = b1 ? __builtin_nontemporal_load(val + o1 + __popcll(m1 & lt)) : 0.0;
            const double v2 = b2 ? __builtin_nontemporal_load(val + o2 + __popcll(m2 & lt)) : 0.0;
            const double v3 = b3 ? __builtin_nontemporal_load(val + o3 + __popcll(m3 & lt)) : 0.0;
            const int j = g * 64 + lane;
            const double x0 = b0 ? xsub[j] : 0.0, x1 = b1 ? xsub[j + 64] : 0.0, x2 = b2 ? xsub[j + 128] : 0.0,
                         x3 = b3 ? xsub[j + 192] : 0.0;
            acc += v0 * x0; acc += v1 * x1; acc += v2 * x2; acc += v3 * x3;
            off = o3 + __popcll(m3);
        }
        for (; g < n_groups; ++g) {
            const unsigned long long mk = mrow[g];
            if ((mk >> lane) & 1ull) acc += val[off + __popcll(mk & lt)] * xsub[g * 64 + lane];
            off += __popcll(mk);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const int r = rows[s];
            y[r] += acc;                                       // unpack_add, dist_spmv_split_sparse.cpp:70-76
            if (DOT) dot += p[r] * acc;
        }
    }
    if (DOT) {
        const double t = block_sum4(dot, lds4);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------- current and power
__global__ __launch_bounds__(KMCF_BLOCK) void scale_kernel(int n, double *__restrict__ m, double g0)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m[i] = m[i] * g0;
}

// get_imacro_sparse (src/current_solver_gpu.cu:501-542), one block on the rank that owns matrix row 1:
// over that row's entries with column node >= 2: sum of X[1][c] (m[c] - m[1]).
__global__ __launch_bounds__(KMCF_BLOCK) void imacro_kernel(int row, const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                            const double *__restrict__ val, const int *__restrict__ col_node,
                                                            const double *__restrict__ m, double *__restrict__ out)
{
    __shared__ double lds4[4];
    double s = 0.0;
    if (row >= 0) {
        const double m1 = m[1];
        for (int j = row_ptr[row] + threadIdx.x; j < row_ptr[row + 1]; j += KMCF_BLOCK) {
            const int node = col_node[col[j]];
            if (node >= 2) s += val[j] * (m[node] - m1);
        }
    }
    const double t = block_sum4(s, lds4);
    if (threadIdx.x == 0) *out = t;
}

// min over m[2 .. n) (thrust::min_element, :2068), one block
__global__ __launch_bounds__(KMCF_BLOCK) void min_kernel(int n, const double *__restrict__ m, double *__restrict__ out)
{
    __shared__ double sh[KMCF_BLOCK];
    double v = m[2];
    for (int i = 2 + threadIdx.x; i < n; i += KMCF_BLOCK) v = fmin(v, m[i]);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = KMCF_BLOCK / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fmin(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// update_m (:449-459) with the minimum read once (there every thread re-reads m[minidx] while it is being updated)
__global__ __launch_bounds__(KMCF_BLOCK) void shift_kernel(int n, double *__restrict__ m, const double *__restrict__ mn)
{
    const double sh = fabs(*mn);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m[i] += sh;
}

__device__ __forceinline__ double ineg_of(double x, double mr, double mc, double Vd)
{   // set_ineg, :2367-2377
    const double ical = x * (mr - mc);
    if ((ical < 0 && Vd > 0) || (ical > 0 && Vd < 0)) return -ical;
    return 0.0;
}

// neighbour part of P[r] = sum_c ineg[r][c] m[c] - (sum_c ineg[r][c]) m[r], rows and columns of atoms only
template <int LPR>
__global__ __launch_bounds__(KMCF_BLOCK) void power_neighbour_kernel(int n_loc, const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                                     const double *__restrict__ val, const int *__restrict__ diag_pos,
                                                                     const int *__restrict__ col_node, const double *__restrict__ m,
                                                                     double Vd, double *__restrict__ psum, double *__restrict__ isum)
{
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int groups = (n_loc + RPB - 1) / RPB;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int r = grp * RPB + threadIdx.x / LPR;
        const bool valid = r < n_loc;
        double a = 0.0, b = 0.0;
        if (valid) {
            const int nr = col_node[r];
            if (nr >= 2) {
                const double mr = m[nr];
                const int dpos = diag_pos[r];
                for (int j = row_ptr[r] + lane; j < row_ptr[r + 1]; j += LPR) {
                    if (j == dpos) continue;
                    const int nc = col_node[col[j]];
                    if (nc < 2) continue;
                    const double mc = m[nc];
                    const double ig = ineg_of(val[j], mr, mc, Vd);
                    a += ig * mc;
                    b += ig;
                }
            }
        }
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if (valid && lane == 0) { psum[r] = a; isum[r] = b; }
    }
}

// tunnel part, added into the same per-row sums (a wave per local sub row)
__global__ __launch_bounds__(KMCF_BLOCK) void power_tunnel_kernel(
    int n_loc, int s0, int n_groups, const unsigned long long *__restrict__ mask, const long long *__restrict__ voff,
    const double *__restrict__ val, const int *__restrict__ tidx, const int *__restrict__ rows, const double *__restrict__ m,
    double Vd, double *__restrict__ psum, double *__restrict__ isum)
{
    const int lane = threadIdx.x & 63;
    const int wpb = KMCF_BLOCK / 64;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < n_loc; s += gridDim.x * wpb) {
        const int sg = s0 + s;
        const double mr = m[tidx[sg] + 2];
        long long off = voff[s];
        double a = 0.0, b = 0.0;
        for (int g = 0; g < n_groups; ++g) {
            const unsigned long long mk = mask[(size_t)s * n_groups + g];
            if (mk == 0ull) continue;
            if ((mk >> lane) & 1ull) {
                const int j = g * 64 + lane;
                if (j != sg) {
                    const double mc = m[tidx[j] + 2];
                    const double ig = ineg_of(val[off + __popcll(mk & lt)], mr, mc, Vd);
                    a += ig * mc;
                    b += ig;
                }
            }
            off += __popcll(mk);
        }
        a = wave_sum(a); b = wave_sum(b);
        if (lane == 0) { psum[rows[s]] += a; isum[rows[s]] += b; }
    }
}

// P[r] into the row-partitioned pdisp (global row order)
__global__ __launch_bounds__(KMCF_BLOCK) void power_finish_kernel(int n_loc, const int *__restrict__ col_node, const double *__restrict__ m,
                                                                  const double *__restrict__ psum, const double *__restrict__ isum,
                                                                  double *__restrict__ pdisp)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_loc; r += gridDim.x * blockDim.x) {
        const int node = col_node[r];
        pdisp[node] = node >= 2 ? psum[r] + (-isum[r]) * m[node] : 0.0;
    }
}

// copy_pdisp (:462-474): site_power[site of atom a] = -alpha P for non-metal atoms
__global__ __launch_bounds__(KMCF_BLOCK) void copy_pdisp_kernel(int n_atoms_in_matrix, const int *__restrict__ atom_site,
                                                                const unsigned char *__restrict__ acls, const double *__restrict__ pdisp,
                                                                double alpha, double *__restrict__ site_power)
{
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < n_atoms_in_matrix; a += gridDim.x * blockDim.x)
        if (!(acls[a] & 1)) site_power[atom_site[a]] = -1 * alpha * pdisp[a + 2];
}


// ---------------------------------------------------------------- dense symmetric sub-block (kmcf_subop::dense)
// Tile (I, J), J >= I, of the upper block triangle: 64 x 64 f64, row-major, at tile index I nb - I (I - 1) / 2 + (J - I).
__host__ __device__ inline long long symm_tile_index(int nb, int I, int J) { return (long long)I * nb - (long long)I * (I - 1) / 2 + (J - I); }

constexpr int SYM_LD = 65;                                   // LDS row stride (doubles): row- AND column-wise reads conflict-free
constexpr int SYM_WAVE_DOUBLES = 64 * SYM_LD + 256;          // tile + four 64-vectors per wave

// Values of the upper tiles (populate_T_tunnel_dist2 on the pairs of the tile; 0 where the pattern has no entry, on
// the diagonal -- set afterwards -- and past the last point).  A wave per strip, lane = column, rows one by one.
__global__ __launch_bounds__(KMCF_BLOCK) void tunnel_dense_fill_kernel(
    int n_strips, const int4 *__restrict__ strips, int n_glob, const int *__restrict__ tinfo, const double *__restrict__ tx,
    const double *__restrict__ ty, const double *__restrict__ tz, const double *__restrict__ tcb, double nn_dist, double tol,
    double m_e, double V0, double *__restrict__ tiles)
{
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < n_strips; s += gridDim.x * 4) {
        const int4 st = strips[s];
        for (int q = 0; q < st.z; ++q) {
            const int J = st.y + q, j = 64 * J + lane;
            const bool jin = j < n_glob;
            const double xj = jin ? tx[j] : 0.0, yj = jin ? ty[j] : 0.0, zj = jin ? tz[j] : 0.0, cbj = jin ? tcb[j] : 0.0;
            const int fj = jin ? tinfo[j] : 0;
            double *tile = tiles + ((size_t)st.w + q) * 4096;
            for (int r = 0; r < 64; ++r) {
                const int i = 64 * st.x + r;
                double v = 0.0;
                if (i < n_glob && jin && i != j) {
                    bool c2t;
                    const double cbi = tcb[i];
                    const double d = dist3(tx[i], ty[i], tz[i], xj, yj, zj);
                    if (d > nn_dist && tunnel_pair(tinfo[i], fj, cbi, cbj, tol, &c2t)) v = wkb_value(d, cbi, cbj, c2t, m_e, V0);
                }
                tile[r * 64 + lane] = v;
            }
        }
    }
}

// One pass over the upper tiles.  MODE 0: parts of y = S x (x: the gathered sub-vector).  MODE 1: parts of the
// dissipated-power sums a_i = sum_j ineg(v_ij, m_i, m_j) m_j, b_i = sum_j ineg(v_ij, m_i, m_j) (x: the potentials of the
// tunnel points; the diagonal is skipped).  A wave per strip of tiles of one block row: the tile goes global ->
// registers -> LDS (the NEXT tile's loads are in flight while this one is used); lane r adds row r's products
// column by column (its row sum runs on through the strip's tiles), lane c adds column c's products row by row
// (the tile's contribution to block row J by symmetry; not for diagonal tiles, which are stored complete).
template <int MODE>
__global__ __launch_bounds__(KMCF_BLOCK) void sub_symm_kernel(int n_strips, const int4 *__restrict__ strips, const double *__restrict__ tiles,
                                                              const double *__restrict__ x, double Vd, double *__restrict__ rowpart,
                                                              double *__restrict__ colpart, const kmcf_scalars *__restrict__ S, int check_done)
{
    extern __shared__ double sym_lds[];
    if (check_done && S->done) return;                  // (iterations enqueued behind the stop: nothing to stream)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *T = sym_lds + (size_t)wv * SYM_WAVE_DOUBLES;
    double *xI = T + 64 * SYM_LD, *xJ = xI + 64;
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    dvec2 reg[32];
    const int s0 = blockIdx.x * 4 + wv, ds = gridDim.x * 4;
    int4 st_next = s0 < n_strips ? strips[s0] : make_int4(0, 0, 0, 0);
    if (s0 < n_strips) {
        const dvec2 *src = reinterpret_cast<const dvec2 *>(tiles + (size_t)st_next.w * 4096);
#pragma unroll
        for (int k = 0; k < 32; ++k) reg[k] = __builtin_nontemporal_load(src + k * 64 + lane);
    }
    for (int s = s0; s < n_strips; s += ds) {
        const int4 st = st_next;
        if (s + ds < n_strips) st_next = strips[s + ds];
        const int I = st.x;
        xI[lane] = x[64 * I + lane];
        double ra = 0.0, rb = 0.0;
        for (int q = 0; q < st.z; ++q) {
            const int J = st.y + q;
            // The tile in the registers goes to LDS pair by pair, and each register, as soon as it is free, requests its
            // share of the NEXT tile (of this strip, or the first of the wave's next strip): a wave always has a whole tile
            // (32 KB) in flight.  (Requesting the next tile only after the last pair of this one had arrived left every
            // wave without a single load in flight for one memory latency per tile: 337 us per application at 40 nm.)
            const bool more = q + 1 < st.z, next_strip = !more && s + ds < n_strips;
            const dvec2 *src = reinterpret_cast<const dvec2 *>(tiles + (more ? (size_t)st.w + q + 1 : (size_t)st_next.w) * 4096);
#pragma unroll
            for (int k = 0; k < 32; ++k) {                   // element pair k 64 + lane = row 2 k + (lane >> 5), columns 2 (lane & 31), + 1
                const int r = 2 * k + (lane >> 5), c = 2 * (lane & 31);
                T[r * SYM_LD + c] = reg[k].x;
                T[r * SYM_LD + c + 1] = reg[k].y;
                if (more || next_strip) reg[k] = __builtin_nontemporal_load(src + k * 64 + lane);
            }
            xJ[lane] = x[64 * J + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool diag = J == I;
            if (MODE == 0) {
                // (x_J / x_I through v_readlane instead of LDS broadcasts: 0.296 against 0.280 ms per iteration -- the kernel is bound by its stream)
#pragma unroll 8
                for (int c = 0; c < 64; ++c) ra += T[lane * SYM_LD + c] * xJ[c];
                if (!diag) {
                    double ca = 0.0;
#pragma unroll 8
                    for (int r = 0; r < 64; ++r) ca += T[r * SYM_LD + lane] * xI[r];
                    colpart[((size_t)st.w + q) * 64 + lane] = ca;
                }
            } else {
                const double mr = xI[lane];
                for (int c = 0; c < 64; ++c) {
                    if (diag && c == lane) continue;
                    const double mc = xJ[c];
                    const double ig = ineg_of(T[lane * SYM_LD + c], mr, mc, Vd);
                    ra += ig * mc;
                    rb += ig;
                }
                if (!diag) {
                    const double mj = xJ[lane];
                    double ca = 0.0, cb = 0.0;
                    for (int r = 0; r < 64; ++r) {
                        const double mi = xI[r];
                        const double ig = ineg_of(T[r * SYM_LD + lane], mj, mi, Vd);
                        ca += ig * mi;
                        cb += ig;
                    }
                    colpart[((size_t)st.w + q) * 128 + lane] = ca;
                    colpart[((size_t)st.w + q) * 128 + 64 + lane] = cb;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (MODE == 0) rowpart[(size_t)s * 64 + lane] = ra;
        else { rowpart[(size_t)s * 128 + lane] = ra; rowpart[(size_t)s * 128 + 64 + lane] = rb; }
    }
}

// ---------------------------------------------------------------- jagged symmetric tiles (kmcf_subop::jagged)
// The upper block triangle again (tile index, strips, row / column parts and their reduction exactly as above), but a
// tile stores only its ENTRIES: the 64 row masks of the pattern (tile-major copy of the bitmap's words) and the values
// in "layers" -- layer k holds the k-th entry of every row that has more than k entries, rows ascending, without gaps
// (a jagged-diagonal layout inside the tile): 4 B per entry of the full block + 1 bit per position, against 4 B per
// POSITION for the dense tiles -- the reference's contact window (44 % full) moves 2.1 x fewer bytes, and the sums are
// the SAME sums: lane r adds row r's products in ascending column order (an absent entry added 0.0 before), lane c
// the tile's column c top-down out of a dense copy of the tile in LDS.  One load instruction per layer, lane r
// reading base_k + (number of rows below r in the layer): consecutive lanes, consecutive addresses.
typedef unsigned long long symj_u64;

// largest of the lanes' values (0 ... 64), through ballots: scalar work only (a shuffle butterfly is six trips through
// the LDS crossbar, ~0.4 us -- per tile)
__device__ __forceinline__ int symj_wave_max(int v)
{
    symj_u64 live = ~0ull;
    int r = 0;
#pragma unroll
    for (int bit = 6; bit >= 0; --bit) {
        const symj_u64 b = __ballot((v >> bit) & 1) & live;
        if (b) { live = b; r |= 1 << bit; }
    }
    return r;
}

// element `l` (a compile-time constant in the unrolled loops) of a 64-vector held one element per lane
__device__ __forceinline__ double symj_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// tile-major masks + entries per tile
__global__ __launch_bounds__(KMCF_BLOCK) void symj_mask_kernel(int n_strips, const int4 *__restrict__ strips, int n_glob, int n_groups,
                                                               const symj_u64 *__restrict__ mask, symj_u64 *__restrict__ jmask, int *__restrict__ tcnt)
{
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < n_strips; s += gridDim.x * 4) {
        const int4 st = strips[s];
        const int i = 64 * st.x + lane;
        for (int q = 0; q < st.z; ++q) {
            const symj_u64 w = i < n_glob ? mask[(size_t)i * n_groups + st.y + q] : 0ull;
            jmask[((size_t)st.w + q) * 64 + lane] = w;
            int c = __popcll(w);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if (lane == 0) tcnt[st.w + q] = c;
        }
    }
}

// values in layer order (populate_T_tunnel_dist2 on the tile's entries; the diagonal entry 0 until symj_set_diag_kernel)
__global__ __launch_bounds__(KMCF_BLOCK) void symj_fill_kernel(
    int n_strips, const int4 *__restrict__ strips, int n_glob, const int *__restrict__ tinfo, const double *__restrict__ tx,
    const double *__restrict__ ty, const double *__restrict__ tz, const double *__restrict__ tcb, double tol, double m_e, double V0,
    const symj_u64 *__restrict__ jmask, const long long *__restrict__ jvoff, double *__restrict__ jval)
{
    const int lane = threadIdx.x & 63;
    const symj_u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < n_strips; s += gridDim.x * 4) {
        const int4 st = strips[s];
        const int i = 64 * st.x + lane;
        const bool iin = i < n_glob;
        const double xi = iin ? tx[i] : 0.0, yi = iin ? ty[i] : 0.0, zi = iin ? tz[i] : 0.0, cbi = iin ? tcb[i] : 0.0;
        const int fi = iin ? tinfo[i] : 0;
        for (int q = 0; q < st.z; ++q) {
            const size_t t = (size_t)st.w + q;
            symj_u64 m = jmask[t * 64 + lane];
            long long base = jvoff[t];
            while (true) {
                const symj_u64 b = __ballot(m != 0);
                if (!b) break;
                if (m) {
                    const int j = 64 * (st.y + q) + __builtin_ctzll(m);
                    m &= m - 1;
                    double v = 0.0;
                    if (j != i) {
                        bool c2t;
                        const double cbj = tcb[j];
                        tunnel_pair(fi, tinfo[j], cbi, cbj, tol, &c2t);
                        v = wkb_value(dist3(xi, yi, zi, tx[j], ty[j], tz[j]), cbi, cbj, c2t, m_e, V0);
                    }
                    jval[base + __popcll(b & lt)] = v;
                }
                base += __popcll(b);
            }
        }
    }
}

// tdiag = -(row sums) into the diagonal tiles' diagonal entries (calc_diagonal_T_tunnel, :669-689); a wave per block row
__global__ __launch_bounds__(KMCF_BLOCK) void symj_set_diag_kernel(int n_glob, int nb, const double *__restrict__ rowsum, double *__restrict__ tdiag,
                                                                   const symj_u64 *__restrict__ jmask, const long long *__restrict__ jvoff,
                                                                   double *__restrict__ jval)
{
    const int lane = threadIdx.x & 63;
    const symj_u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int B = blockIdx.x * 4 + (threadIdx.x >> 6); B < nb; B += gridDim.x * 4) {
        const size_t t = (size_t)symm_tile_index(nb, B, B);
        const symj_u64 m = jmask[t * 64 + lane];
        const int cnt = __popcll(m), kl = __popcll(m & lt), i = 64 * B + lane;
        long long pos = jvoff[t];
        for (int k = 0; k < 64; ++k) {
            const symj_u64 b = __ballot(cnt > k);
            if (!b) break;
            if (k < kl) pos += __popcll(b);
            else if (k == kl) pos += __popcll(b & lt);
        }
        if (i < n_glob && ((m >> lane) & 1ull)) {
            const double d = -rowsum[i];
            tdiag[i] = d;
            jval[pos] = d;
        }
    }
}

// One pass over the stored tiles: the jagged twin of sub_symm_kernel (same parts into rowpart / colpart, MODE as there).
// Per tile ONE wait for memory, at the top, for loads that were all requested a tile earlier, before the passes over
// the dense copy: this tile's layers (a lane past its row's end re-reads the layer's first word: no divergent loads),
// the next tile's masks and x_J.  Control flow is scalar (the strip walk is per wavefront), the scatter into the dense
// copy and its removal are branch-free (idle lanes write a dump word).
template <int MODE>
__global__ __launch_bounds__(KMCF_BLOCK) void sub_symj_kernel(int n_strips, const int4 *__restrict__ strips, const symj_u64 *__restrict__ jmask,
                                                              const long long *__restrict__ jvoff, const double *__restrict__ jval,
                                                              const double *__restrict__ x, double Vd, double *__restrict__ rowpart,
                                                              double *__restrict__ colpart, const kmcf_scalars *__restrict__ S, int check_done)
{
    extern __shared__ double sym_lds[];
    if (check_done && S->done) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const symj_u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    double *T = sym_lds + (size_t)wv * SYM_WAVE_DOUBLES;
    double *xI = T + 64 * SYM_LD, *xJ = xI + 64, *dump = xJ + 64;      // (dump: 64 words of the wave's four spare 64-vectors)
#pragma unroll 8
    for (int c = 0; c < 64; ++c) T[lane * SYM_LD + c] = 0.0;           // the dense copy: entries in, entries out again per tile
    const int ds = gridDim.x * 4;
    int s = blockIdx.x * 4 + wv;                                       // (scalar)
    if (s >= n_strips) return;
    int4 st = strips[s];
    st.x = __builtin_amdgcn_readfirstlane(st.x); st.y = __builtin_amdgcn_readfirstlane(st.y);
    st.z = __builtin_amdgcn_readfirstlane(st.z); st.w = __builtin_amdgcn_readfirstlane(st.w);
    int q = 0;
    double v[64];
    auto request = [&](symj_u64 m, long long base) {                   // the layers of a tile (in eights, up to its deepest)
        const int cnt_ = __popcll(m), kmax_ = symj_wave_max(cnt_);
        const double *src = jval + base;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if ((k & 7) == 0 && k >= kmax_) break;
            const symj_u64 b = __ballot(cnt_ > k);
            v[k] = __builtin_nontemporal_load(src + (cnt_ > k ? __popcll(b & lt) : 0));
            src += __popcll(b);
        }
    };
    // the tile after (s, q) of this wave's walk (scalar): the next of the strip, or the first of the wave's next strip
    auto step = [&](const int4 &a, int sa, int qa, int4 &b, int &sb_, int &qb) -> bool {
        b = a; sb_ = sa; qb = qa + 1;
        if (qb < a.z) return true;
        if (sa + ds >= n_strips) return false;
        sb_ = sa + ds;
        b = strips[sb_];
        b.x = __builtin_amdgcn_readfirstlane(b.x); b.y = __builtin_amdgcn_readfirstlane(b.y);
        b.z = __builtin_amdgcn_readfirstlane(b.z); b.w = __builtin_amdgcn_readfirstlane(b.w);
        qb = 0;
        return true;
    };
    symj_u64 m0 = jmask[(size_t)st.w * 64 + lane];
    int4 st1; int s1, q1;
    bool have1 = step(st, s, q, st1, s1, q1);
    symj_u64 m1 = have1 ? jmask[((size_t)st1.w + q1) * 64 + lane] : 0ull;      // masks run one tile ahead of the values
    request(m0, jvoff[st.w]);
    double xjn = x[64 * st.y + lane];
    xI[lane] = x[64 * st.x + lane];
    double ra = 0.0, rb = 0.0;
    while (true) {
        const int I = st.x, J = st.y + q;
        const size_t t = (size_t)st.w + q;
        const bool diag = J == I;
        const bool more = q + 1 < st.z;
        // ---- this tile's layers are in v (the one wait); entries into the dense copy
        xJ[lane] = xjn;
        const double xi_l = xI[lane], xj_l = xjn;                      // x_I / x_J, one element per lane (columns: readlane)
        const int cnt = __popcll(m0), kmax = symj_wave_max(cnt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            // entries into the dense copy
            symj_u64 m = m0;
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if ((k & 7) == 0 && k >= kmax) break;                  // (scalar; k is a compile-time constant in the unrolled body)
                const bool on = m != 0;
                const int c = on ? __builtin_ctzll(m) : 0;
                m &= m - 1;
                double *dst = on ? T + lane * SYM_LD + c : dump + lane;
                *dst = v[k];
            }
        }
        // ---- the next tile's masks arrived with them: its layers, its x_J and the masks of the tile after it go out now
        // and are in flight during the passes over the dense copy
        int4 st2 = st1; int s2 = s1, q2 = q1;
        const bool have2 = have1 && step(st1, s1, q1, st2, s2, q2);
        const symj_u64 m2 = have2 ? jmask[((size_t)st2.w + q2) * 64 + lane] : 0ull;
        double xj1 = 0.0, xi1 = 0.0;
        if (have1) {
            request(m1, jvoff[(size_t)st1.w + q1]);
            xj1 = x[64 * (st1.y + q1) + lane];
            if (s1 != s) xi1 = x[64 * st1.x + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kmax > 0) {                                                // the row, as the dense tiles add it (x_J by readlane: no LDS broadcasts)
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 64; ++c) ra += T[lane * SYM_LD + c] * symj_lane(xj_l, c);
            } else {
#pragma unroll
                for (int c = 0; c < 64; ++c) {
                    const double mc = symj_lane(xj_l, c);
                    const double ig = ineg_of(T[lane * SYM_LD + c], xi_l, mc, Vd);
                    if (!(diag && c == lane)) { ra += ig * mc; rb += ig; }
                }
            }
        }
        if (!diag) {
            double ca = 0.0, cb = 0.0;
            if (kmax > 0) {
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 64; ++r) ca += T[r * SYM_LD + lane] * symj_lane(xi_l, r);
                } else {
#pragma unroll
                    for (int r = 0; r < 64; ++r) {
                        const double mi = symj_lane(xi_l, r);
                        const double ig = ineg_of(T[r * SYM_LD + lane], xj_l, mi, Vd);
                        ca += ig * mi;
                        cb += ig;
                    }
                }
            }
            if (MODE == 0) colpart[t * 64 + lane] = ca;
            else { colpart[t * 128 + lane] = ca; colpart[t * 128 + 64 + lane] = cb; }
        }
        __builtin_amdgcn_wave_barrier();
        if (kmax > 0) {   // entries out again: the lane's row (64 plain stores; walking the mask again costs ten instructions per entry)
#pragma unroll
            for (int c = 0; c < 64; ++c) T[lane * SYM_LD + c] = 0.0;
        }
        if (!more) {                                   // the strip's row parts
            if (MODE == 0) rowpart[(size_t)s * 64 + lane] = ra;
            else { rowpart[(size_t)s * 128 + lane] = ra; rowpart[(size_t)s * 128 + 64 + lane] = rb; }
            ra = rb = 0.0;
        }
        if (!have1) break;
        xjn = xj1;
        if (s1 != s) {
            __builtin_amdgcn_wave_barrier();
            xI[lane] = xi1;
        }
        st = st1; s = s1; q = q1; m0 = m1;
        st1 = st2; s1 = s2; q1 = q2; m1 = m2; have1 = have2;
    }
}

// Adds the parts of one block row B (one block, four waves) in a fixed order: wave w takes the column sums of tiles
// (K, B) for K = w, w + 4, ... < B, then the strips first + w, first + w + 4, ... of its block row, each list in
// ascending order; the four waves' sums are joined as ((c0 + c1) + (c2 + c3)) + ((r0 + r1) + (r2 + r3)).  Into
// y[rows[i]] (MODE 0; rows == nullptr: y[i]) or psum / isum (MODE 1); fused p.Ap partial, one per block (MODE 0).
// (A single lane walking all ~300 parts of its row one dependent load after the other took 83 us at 40 nm.)
template <int MODE, bool DOT>
__global__ __launch_bounds__(KMCF_BLOCK) void sub_symm_reduce_kernel(int n_glob, int nb, const int *__restrict__ strip_first,
                                                                     const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                                     const int *__restrict__ rows, const double *__restrict__ p,
                                                                     double *__restrict__ y, double *__restrict__ y2, double *__restrict__ part,
                                                                     const kmcf_scalars *__restrict__ S, int check_done,
                                                                     const int *__restrict__ tile_local = nullptr, double *__restrict__ ypart = nullptr)
{
    // Spread over a rank group (tile_local, ypart != nullptr): the same walk over THIS rank's tiles and strips (another
    // rank's tile (K, B) is skipped in place: wave w keeps K = w, w + 4, ...; the strips are this rank's list), the sums
    // of all 64 nb points go to ypart (and ypart + 64 nb, MODE 1) for sub_combine_kernel instead of y.
    __shared__ double sh[2][2][4][64];                  // [a / b][column / row parts][wave][lane]
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    constexpr int W = MODE == 0 ? 64 : 128;
    const int B = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    double ca = 0.0, cb = 0.0, ra = 0.0, rb = 0.0;
#pragma unroll 4
    for (int K = w; K < B; K += 4) {
        size_t t = (size_t)symm_tile_index(nb, K, B);
        if (tile_local) {
            const int tl = tile_local[t];
            if (tl < 0) continue;
            t = (size_t)tl;
        }
        ca += colpart[t * W + l];
        if (MODE == 1) cb += colpart[t * W + 64 + l];
    }
    for (int s = strip_first[B] + w; s < strip_first[B + 1]; s += 4) {
        ra += rowpart[(size_t)s * W + l];
        if (MODE == 1) rb += rowpart[(size_t)s * W + 64 + l];
    }
    sh[0][0][w][l] = ca; sh[0][1][w][l] = ra;
    if (MODE == 1) { sh[1][0][w][l] = cb; sh[1][1][w][l] = rb; }
    __syncthreads();
    double dot = 0.0;
    const int i = 64 * B + l;
    if (ypart) {
        if (w == 0) {
            ypart[i] = ((sh[0][0][0][l] + sh[0][0][1][l]) + (sh[0][0][2][l] + sh[0][0][3][l])) +
                       ((sh[0][1][0][l] + sh[0][1][1][l]) + (sh[0][1][2][l] + sh[0][1][3][l]));
            if (MODE == 1)
                ypart[64 * nb + i] = ((sh[1][0][0][l] + sh[1][0][1][l]) + (sh[1][0][2][l] + sh[1][0][3][l])) +
                                     ((sh[1][1][0][l] + sh[1][1][1][l]) + (sh[1][1][2][l] + sh[1][1][3][l]));
        }
        return;
    }
    if (w == 0 && i < n_glob) {
        const double a = ((sh[0][0][0][l] + sh[0][0][1][l]) + (sh[0][0][2][l] + sh[0][0][3][l])) +
                         ((sh[0][1][0][l] + sh[0][1][1][l]) + (sh[0][1][2][l] + sh[0][1][3][l]));
        const int r = rows ? rows[i] : i;
        y[r] += a;
        if (MODE == 0) {
            if (DOT) dot = p[r] * a;
        } else {
            const double b = ((sh[1][0][0][l] + sh[1][0][1][l]) + (sh[1][0][2][l] + sh[1][0][3][l])) +
                             ((sh[1][1][0][l] + sh[1][1][1][l]) + (sh[1][1][2][l] + sh[1][1][3][l]));
            y2[r] += b;
        }
    }
    if (DOT) {
        const double t = block_sum4(dot, lds4);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

// The sums of a rank group's partials (spread storage): point row0 + s = the partials of ranks 0, 1, ... in that order;
// into y[rows[s]] (rows == nullptr: y[s]) and y2 (MODE 1); fused p.Ap partial, one per block of 256 points (MODE 0).
template <int MODE, bool DOT>
__global__ __launch_bounds__(KMCF_BLOCK) void sub_combine_kernel(int ns, int row0, int P, int npad, const double *__restrict__ ypart_all,
                                                                 const int *__restrict__ rows, const double *__restrict__ p,
                                                                 double *__restrict__ y, double *__restrict__ y2, double *__restrict__ part,
                                                                 const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    constexpr int W = MODE == 0 ? 1 : 2;
    const int s = blockIdx.x * KMCF_BLOCK + threadIdx.x;
    double dot = 0.0;
    if (s < ns) {
        const size_t i = (size_t)row0 + s;
        double a = 0.0, b = 0.0;
        for (int q = 0; q < P; ++q) {
            a += ypart_all[(size_t)q * W * npad + i];
            if (MODE == 1) b += ypart_all[(size_t)q * W * npad + npad + i];
        }
        const int r = rows ? rows[s] : s;
        y[r] += a;
        if (MODE == 1) y2[r] += b;
        else if (DOT) dot = p[r] * a;
    }
    if (DOT) {
        const double t = block_sum4(dot, lds4);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

// tdiag = -(row sums), the diagonal entries of the diagonal tiles (calc_diagonal_T_tunnel, :669-689); tile_local: the
// spread storage's map (only this rank's diagonal tiles exist here)
__global__ __launch_bounds__(KMCF_BLOCK) void symm_set_diag_kernel(int n_glob, int nb, const double *__restrict__ rowsum, double *__restrict__ tdiag,
                                                                   double *__restrict__ tiles, const int *__restrict__ tile_local)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_glob; i += gridDim.x * blockDim.x) {
        const double d = -rowsum[i];
        tdiag[i] = d;
        const int B = i >> 6, l = i & 63;
        long long tl = symm_tile_index(nb, B, B);
        if (tile_local) tl = tile_local[tl];
        if (tl >= 0) tiles[(size_t)tl * 4096 + l * 64 + l] = d;
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void fill_kernel(int n, double *__restrict__ a, double v)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = v;
}

__global__ __launch_bounds__(KMCF_BLOCK) void gather_tunnel_pot_kernel(int n_glob, const int *__restrict__ tidx, const double *__restrict__ m,
                                                                       double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_glob; i += gridDim.x * blockDim.x) out[i] = m[tidx[i] + 2];
}

template <typename T>
int ensure(T **d, size_t *cap, size_t need)
{
    if (need <= *cap && *d) return KMCF_OK;
    if (*d) KMCF_HIP(hipFree(*d));
    *d = nullptr;
    const size_t n = std::max<size_t>(need + need / 4, 64);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(d), n * sizeof(T)));
    *cap = n;
    return KMCF_OK;
}

template <typename T>
int dalloc(T **d, size_t n)
{
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(d), std::max<size_t>(n, 1) * sizeof(T)));
    KMCF_HIP(hipMemset(*d, 0, std::max<size_t>(n, 1) * sizeof(T)));
    return KMCF_OK;
}

}  // namespace

extern "C" int kmcf_tstate_destroy(kmcf_tstate *t)
{
    if (!t) return KMCF_OK;
    if (t->comm && t->comm->device >= 0) {
        hipSetDevice(t->comm->device);
        hipStreamSynchronize(t->comm->stream);
        void *ptrs[] = {t->d_atom_site, t->d_site_is_atom, t->d_ax, t->d_ay, t->d_az, t->d_acb, t->d_ael, t->d_ach, t->d_acls,
                        t->d_cls_col, t->d_col_node, t->d_diag_pos, t->d_ground, t->d_inv_perm, t->d_diag, t->d_diag_tot, t->d_rhs,
                        t->d_tflag, t->d_blk, t->d_tidx, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, t->d_rowcnt, t->d_tdiag,
                        t->sub.d_rows, t->sub.d_mask, t->sub.d_voff, t->sub.d_val, t->sub.d_xsub, t->d_pdisp, t->d_scal, t->d_err,
                        t->sub.d_tiles, t->sub.d_strips, t->sub.d_strip_first, t->sub.d_rowpart, t->sub.d_colpart,
                        t->sub.d_jmask, t->sub.d_jvoff, t->sub.d_jval, t->sub.d_jcnt, t->sub.d_tile_local, t->sub.d_ypart, t->d_agree};
        for (void *p : ptrs)
            if (p) hipFree(p);
        if (t->h_pin) hipHostFree(t->h_pin);
        if (t->h_agree) hipHostFree(t->h_agree);
    }
    if (t->T) { t->T->sub = nullptr; kmcf_matrix_destroy(t->T); }
    delete t;
    return KMCF_OK;
}

extern "C" kmcf_matrix *kmcf_tstate_matrix(kmcf_tstate *t) { return t ? t->T : nullptr; }

extern "C" int kmcf_initialize_sparsity_T(kmcf_comm *c, const double *d_site_x, const double *d_site_y, const double *d_site_z,
                                          const int *d_site_element, int N, double nn_dist, int num_source_inj,
                                          int num_ground_ext, int num_layers_contact, const int *h_counts_T,
                                          const int *h_displs_T, kmcf_tstate **out)
{
    KMCF_CHECK(c && d_site_x && d_site_y && d_site_z && d_site_element && h_counts_T && h_displs_T && out, KMCF_ERR_ARG,
               "kmcf_initialize_sparsity_T: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_initialize_sparsity_T: host-only communicator");
    KMCF_CHECK(N > 0 && num_source_inj > 0 && num_ground_ext > 0 && num_layers_contact > 0 && nn_dist > 0, KMCF_ERR_ARG,
               "kmcf_initialize_sparsity_T: bad sizes");
    KMCF_TRY(kmcf_enter(c));
    hipStream_t st = c->stream;
    KMCF_HIP(hipStreamSynchronize(st));
    // atoms = sites that are not interstitials (is_defect, src/gpu_solvers.h:331-337), in site order
    std::vector<int> el((size_t)N);
    KMCF_HIP(hipMemcpy(el.data(), d_site_element, (size_t)N * sizeof(int), hipMemcpyDeviceToHost));
    kmcf_tstate *t = new kmcf_tstate();
    struct guard_t { kmcf_tstate *t; ~guard_t() { if (t) kmcf_tstate_destroy(t); } } guard{t};
    t->comm = c; t->N = N; t->n_inj = num_source_inj; t->n_ext = num_ground_ext; t->n_layers = num_layers_contact;
    t->nn_dist = nn_dist;
    std::vector<unsigned char> is_atom((size_t)N, 0);
    for (int s = 0; s < N; ++s)
        if (el[s] != EL_DEFECT && el[s] != EL_OXYGEN_DEFECT) { t->h_atom_site.push_back(s); is_atom[s] = 1; }
    const int Na = t->N_atom = (int)t->h_atom_site.size();
    const int Nsub = t->Nsub = Na + 1;
    KMCF_CHECK(Na >= 4 && num_source_inj + 2 < Nsub && num_ground_ext < Na, KMCF_ERR_ARG,
               "kmcf_initialize_sparsity_T: %d atoms for %d injection / %d extraction atoms", Na, num_source_inj, num_ground_ext);
    const int P = c->nranks, rank = c->rank;
    {
        int64_t tot = 0;
        for (int q = 0; q < P; ++q) tot += h_counts_T[q];
        KMCF_CHECK(tot == Nsub, KMCF_ERR_ARG, "kmcf_initialize_sparsity_T: counts_T sum to %lld, the matrix has N_atom + 1 = %d rows",
                   (long long)tot, Nsub);
    }
    const int n_loc = h_counts_T[rank], row0 = h_displs_T[rank];
    KMCF_TRY(upload(&t->d_atom_site, t->h_atom_site));
    KMCF_TRY(upload(&t->d_site_is_atom, is_atom));
    KMCF_TRY(dalloc(&t->d_ax, (size_t)Na)); KMCF_TRY(dalloc(&t->d_ay, (size_t)Na)); KMCF_TRY(dalloc(&t->d_az, (size_t)Na));
    KMCF_TRY(dalloc(&t->d_acb, (size_t)Na)); KMCF_TRY(dalloc(&t->d_ael, (size_t)Na)); KMCF_TRY(dalloc(&t->d_ach, (size_t)Na));
    KMCF_TRY(dalloc(&t->d_acls, (size_t)Na));
    // On the CALLER's stream (kmcf_setup_stream, kmcf_internal.hpp): the first kernel that reads the caller's arrays.
    gather_coords_kernel<<<grid1d(Na), KMCF_BLOCK, 0, kmcf_setup_stream(c)>>>(Na, t->d_atom_site, d_site_x, d_site_y, d_site_z, t->d_ax, t->d_ay, t->d_az);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(kmcf_setup_stream(c)));
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(st));
    std::vector<double> ax((size_t)Na), ay((size_t)Na), az((size_t)Na);
    KMCF_HIP(hipMemcpy(ax.data(), t->d_ax, (size_t)Na * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(ay.data(), t->d_ay, (size_t)Na * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(az.data(), t->d_az, (size_t)Na * sizeof(double), hipMemcpyDeviceToHost));
    {
        // what the gather kernel read must be what a synchronous copy of the caller's arrays reads (the failure this
        // guards against built a pattern from coordinates whose y and z were still zero: silently)
        std::vector<double> sx((size_t)N);
        const double *src[3] = {d_site_x, d_site_y, d_site_z};
        const std::vector<double> *got[3] = {&ax, &ay, &az};
        for (int q = 0; q < 3; ++q) {
            KMCF_HIP(hipMemcpy(sx.data(), src[q], (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
            for (int a = 0; a < Na; ++a)
                KMCF_CHECK(memcmp(&(*got[q])[(size_t)a], &sx[(size_t)t->h_atom_site[(size_t)a]], sizeof(double)) == 0, KMCF_ERR_STATE,
                           "kmcf_initialize_sparsity_T: the gather kernel does not see the site coordinates the host copy sees (atom %d, "
                           "coordinate %d: %.17g against %.17g): an upload of the caller's arrays has not completed for this stream -- "
                           "upload from pinned memory, or synchronise the upload's stream, before the call", a, q, (*got[q])[(size_t)a],
                           sx[(size_t)t->h_atom_site[(size_t)a]]);
        }
    }

    // atom-atom pattern of this rank's atom rows (calc_nnz_per_row_T / assemble_T_col_indices, "direct terms",
    // src/initialize_sparsity_T.cu:62-72, 166-177; the diagonal comes with it: dist 0 < nn_dist)
    const int a0 = std::max(row0, 2) - 2, a1 = std::max(row0 + n_loc, 2) - 2;      // atoms [a0, a1) = rows [a0 + 2, a1 + 2)
    std::vector<int> arp, acol;
    {
        host_cells hc;
        const double lattice[3] = {1, 1, 1};
        KMCF_TRY(build_cells(t->d_ax, t->d_ay, t->d_az, Na, lattice, 0, nn_dist, &hc));
        const int rc = build_pattern(hc, t->d_ax, t->d_ay, t->d_az, lattice, 0, nn_dist, a0, a1 - a0, 0, Na - 1, &arp, &acol, st);
        hc.release();
        if (rc != KMCF_OK) return rc;
    }
    // rows of this rank with GLOBAL columns (src/initialize_sparsity_T.cu:27-60, 126-164)
    std::vector<int> &rp = t->h_row_ptr, &col = t->h_col;
    rp.assign((size_t)n_loc + 1, 0);
    col.clear();
    for (int r = 0; r < n_loc; ++r) {
        const int i = row0 + r;
        if (i == 0) {
            col.push_back(0); col.push_back(1);
            for (int j = std::max(2, (Nsub + 1) - num_ground_ext + 1); j < Nsub; ++j) col.push_back(j);      // :37
        } else if (i == 1) {
            col.push_back(0); col.push_back(1);
            for (int j = 2; j < std::min(num_source_inj + 2, Nsub); ++j) col.push_back(j);                  // :42
        } else {
            if (i > (Nsub + 1) - num_ground_ext) col.push_back(0);                                           // :51
            if (i < num_source_inj + 2) col.push_back(1);                                                    // :56
            const int ar = i - 2 - a0;
            for (int q = arp[ar]; q < arp[ar + 1]; ++q) col.push_back(acol[q] + 2);
        }
        KMCF_CHECK(col.size() < (size_t)INT32_MAX, KMCF_ERR_ARG, "kmcf_initialize_sparsity_T: pattern exceeds int32 indexing");
        rp[r + 1] = (int)col.size();
    }
    // internal row order: virtual nodes first, atoms in bricks like K (kmcf_kstate.hip); long rows are moved
    // behind by the matrix builder
    std::vector<int> perm;
    {
        double edge = 7.7;
        if (const char *e = getenv("KMCF_BRICK")) edge = atof(e);
        if (edge > 0 && n_loc > 1) {
            std::vector<int64_t> key((size_t)n_loc);
            for (int r = 0; r < n_loc; ++r) {
                const int i = row0 + r;
                if (i < 2) { key[r] = -1; continue; }
                const int a = i - 2;
                const int64_t bx = (int64_t)std::floor(ax[a] / edge) + (1 << 19), by = (int64_t)std::floor(ay[a] / edge) + (1 << 19),
                              bz = (int64_t)std::floor(az[a] / edge) + (1 << 19);
                key[r] = (by << 42) | (bz << 21) | bx;
            }
            perm.resize((size_t)n_loc);
            for (int r = 0; r < n_loc; ++r) perm[r] = r;
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key[a] < key[b]; });
        }
    }
    KMCF_TRY(kmcf_matrix_build(c, Nsub, h_counts_T, h_displs_T, rp.data(), col.data(), nullptr, perm.empty() ? nullptr : perm.data(), &t->T));
    kmcf_matrix *m = t->T;
    // per internal row: caller row, diagonal position, ground flag; per internal column: global node
    std::vector<int> iperm((size_t)n_loc);                      // internal -> caller local row
    for (int i = 0; i < n_loc; ++i) iperm[i] = m->h_perm.empty() ? i : m->h_perm[i];
    t->h_inv_perm.assign((size_t)n_loc, 0);
    for (int i = 0; i < n_loc; ++i) t->h_inv_perm[iperm[i]] = i;
    std::vector<int> diag_pos((size_t)n_loc, -1), col_node((size_t)n_loc + m->n_halo);
    std::vector<unsigned char> ground((size_t)n_loc, 0);
    std::vector<int> halo_gid((size_t)m->n_halo);
    if (m->n_halo > 0) KMCF_HIP(hipMemcpy(halo_gid.data(), m->d_halo_gid, (size_t)m->n_halo * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_loc; ++i) {
        const int r = iperm[i], node = row0 + r;
        col_node[i] = node;
        for (int j = rp[r]; j < rp[r + 1]; ++j)
            if (col[j] == node) { diag_pos[i] = m->h_row_ptr[i] + (j - rp[r]); break; }
        if (node >= 2) {
            const int a = node - 2, g = Na - 1;
            const double dx = ax[g] - ax[a], dy = ay[g] - ay[a], dz = az[g] - az[a];
            ground[i] = std::sqrt(dx * dx + dy * dy + dz * dz) < nn_dist ? 1 : 0;      // current_solver_gpu.cu:1115-1117
        }
    }
    for (int h = 0; h < m->n_halo; ++h) col_node[(size_t)n_loc + h] = halo_gid[h];
    KMCF_TRY(upload(&t->d_diag_pos, diag_pos));
    KMCF_TRY(upload(&t->d_ground, ground));
    KMCF_TRY(upload(&t->d_col_node, col_node));
    KMCF_TRY(upload(&t->d_inv_perm, t->h_inv_perm));
    KMCF_TRY(dalloc(&t->d_cls_col, (size_t)n_loc + m->n_halo));
    KMCF_TRY(dalloc(&t->d_diag, (size_t)n_loc)); KMCF_TRY(dalloc(&t->d_diag_tot, (size_t)n_loc)); KMCF_TRY(dalloc(&t->d_rhs, (size_t)n_loc + 2));
    // tunnel workspace sized by the atom count (grow-only buffers for the block itself)
    const int nblk = (Na + KMCF_BLOCK * SCAN_ITEMS - 1) / (KMCF_BLOCK * SCAN_ITEMS);
    KMCF_TRY(dalloc(&t->d_tflag, 1)); KMCF_TRY(dalloc(&t->d_blk, (size_t)2 * nblk + 2));
    KMCF_TRY(dalloc(&t->d_tidx, (size_t)Na)); KMCF_TRY(dalloc(&t->d_tinfo, (size_t)Na));
    KMCF_TRY(dalloc(&t->d_tx, (size_t)Na)); KMCF_TRY(dalloc(&t->d_ty, (size_t)Na)); KMCF_TRY(dalloc(&t->d_tz, (size_t)Na));
    KMCF_TRY(dalloc(&t->d_tcb, (size_t)Na));
    KMCF_TRY(dalloc(&t->d_pdisp, (size_t)Nsub + 2)); KMCF_TRY(dalloc(&t->d_scal, 4)); KMCF_TRY(dalloc(&t->d_err, 1));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&t->h_pin), 8 * sizeof(int), hipHostMallocDefault));
    memset(t->h_pin, 0, 8 * sizeof(int));
    KMCF_TRY(dalloc(&t->d_agree, (size_t)2 * P));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&t->h_agree), (size_t)(2 + 2 * P) * sizeof(double), hipHostMallocDefault));
    memset(t->h_agree, 0, (size_t)(2 + 2 * P) * sizeof(double));
    t->sub.counts.assign(P, 0);
    t->sub.displs.assign(P, 0);
    guard.t = nullptr;
    *out = t;
    return KMCF_OK;
}

extern "C" int kmcf_tstate_info(const kmcf_tstate *t, kmcf_tstate_info_t *info)
{
    KMCF_CHECK(t && info, KMCF_ERR_ARG, "kmcf_tstate_info: null argument");
    info->N_atom = t->N_atom;
    info->Nsub = t->Nsub;
    info->rows_this_rank = t->T->n_loc;
    info->nnz_neighbour = t->T->nnz;
    info->tunnel_points = t->assembled ? t->sub.n_glob : 0;
    info->tunnel_points_rank = t->assembled ? t->sub.n_loc : 0;
    info->tunnel_first = t->assembled ? t->sub.row0 : 0;
    info->nnz_tunnel = t->assembled ? t->sub.nnz : 0;
    info->tunnel_dense = t->assembled && t->sub.dense ? (t->sub.jagged ? 2 : 1) : 0;
    info->tunnel_bytes = !t->assembled ? 0
                         : t->sub.jagged ? (int64_t)t->sub.jnnz * 8 + (int64_t)t->sub.n_tiles * 520
                         : t->sub.dense ? (int64_t)t->sub.n_tiles * 4096 * 8
                                        : (int64_t)t->sub.nnz * 8 + (int64_t)t->sub.n_loc * ((t->sub.n_glob + 63) / 64) * 8;
    return KMCF_OK;
}

extern "C" int kmcf_tstate_pattern(const kmcf_tstate *t, int *h_row_ptr, int *h_col, int64_t *nnz)
{
    KMCF_CHECK(t, KMCF_ERR_ARG, "kmcf_tstate_pattern: null state");
    if (nnz) *nnz = (int64_t)t->h_col.size();
    if (h_row_ptr) memcpy(h_row_ptr, t->h_row_ptr.data(), t->h_row_ptr.size() * sizeof(int));
    if (h_col && !t->h_col.empty()) memcpy(h_col, t->h_col.data(), t->h_col.size() * sizeof(int));
    return KMCF_OK;
}

extern "C" int kmcf_tstate_atom_sites(const kmcf_tstate *t, int *h_atom_site)
{
    KMCF_CHECK(t && h_atom_site, KMCF_ERR_ARG, "kmcf_tstate_atom_sites: null argument");
    memcpy(h_atom_site, t->h_atom_site.data(), t->h_atom_site.size() * sizeof(int));
    return KMCF_OK;
}


// launches of the dense symmetric operator (dynamic LDS beyond 64 KB needs the attribute once per kernel)
template <int MODE>
static int symm_launch(kmcf_subop &sb, const double *x, double Vd, hipStream_t st, const kmcf_scalars *S = nullptr, int check_done = 0)
{
    // (the attribute is set per launch: it is per device, and in-process groups launch from several host threads)
    const size_t lds = (size_t)4 * SYM_WAVE_DOUBLES * sizeof(double);
    const int grid = std::max(1, std::min((sb.n_strips + 3) / 4, 8192));
    if (sb.jagged) {
        KMCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sub_symj_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        sub_symj_kernel<MODE><<<grid, KMCF_BLOCK, lds, st>>>(sb.n_strips, sb.d_strips, sb.d_jmask, sb.d_jvoff, sb.d_jval, x, Vd, sb.d_rowpart, sb.d_colpart,
                                                             S, check_done);
        KMCF_HIP(hipGetLastError());
        return KMCF_OK;
    }
    KMCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sub_symm_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sub_symm_kernel<MODE><<<grid, KMCF_BLOCK, lds, st>>>(sb.n_strips, sb.d_strips, sb.d_tiles, x, Vd, sb.d_rowpart, sb.d_colpart, S, check_done);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

// Spread storage: this rank's partial sums -> every rank's partials (one all-gather on the compute stream) -> the sums
// of the ns points from row0 on, added in rank order (MODE / DOT / rows / y / y2 / part as in sub_symm_reduce_kernel).
// Every rank of the group calls it, also one without points of its own (ns == 0): it holds strips like the others.
template <int MODE, bool DOT>
static int symm_spread_finish(kmcf_comm *c, kmcf_subop &sb, int ns, int row0, const int *rows, const double *p, double *y, double *y2,
                              double *part, const kmcf_scalars *S, int chk)
{
    hipStream_t st = c->stream;
    constexpr int W = MODE == 0 ? 1 : 2;
    const int P = c->nranks, npad = 64 * sb.nb;
    sub_symm_reduce_kernel<MODE, false><<<sb.nb, KMCF_BLOCK, 0, st>>>(sb.n_glob, sb.nb, sb.d_strip_first, sb.d_rowpart, sb.d_colpart, nullptr, nullptr,
                                                                      nullptr, nullptr, nullptr, S, chk, sb.d_tile_local,
                                                                      sb.d_ypart + (size_t)c->rank * W * npad);
    KMCF_HIP(hipGetLastError());
    KMCF_TRY(kmcf_comm_allgatherv_double(c, sb.d_ypart, sb.y_counts[W - 1].data(), sb.y_displs[W - 1].data()));
    if (ns > 0) {
        sub_combine_kernel<MODE, DOT><<<(ns + KMCF_BLOCK - 1) / KMCF_BLOCK, KMCF_BLOCK, 0, st>>>(ns, row0, P, npad, sb.d_ypart, rows, p, y, y2, part, S, chk);
        KMCF_HIP(hipGetLastError());
    }
    return KMCF_OK;
}

// strips, tiles, values, diagonal of the dense symmetric storage (after the tunnel points are known): of one rank, or
// (sb.spread) this rank's share of a group's
static int symm_setup(kmcf_tstate *t)
{
    kmcf_subop &sb = t->sub;
    kmcf_comm *c = t->comm;
    hipStream_t st = c->stream;
    const kmcf_current_params_t *p = &t->par;
    const int n_t = sb.n_glob, nb = (n_t + 63) / 64;
    const int P = sb.spread ? c->nranks : 1, rank = sb.spread ? c->rank : 0;
    sb.nb = nb;
    sb.n_tiles_glob = (long long)nb * (nb + 1) / 2;
    KMCF_CHECK(sb.n_tiles_glob < (long long)INT32_MAX, KMCF_ERR_ARG, "tunnel block of %d points: tile index exceeds int32", n_t);
    int strip_len = 16;
    if (const char *e = getenv("KMCF_SUB_STRIP")) strip_len = std::max(1, atoi(e));
    std::vector<int4> strips;
    std::vector<int> first((size_t)nb + 1, 0), tile_local;
    if (sb.spread) tile_local.assign((size_t)sb.n_tiles_glob, -1);
    // the deal: strip after strip (block rows ascending, then columns) to the rank that holds the fewest tiles so far,
    // the lowest such rank -- within a strip's length of even, whatever the mix of full and ragged strips
    std::vector<long long> held((size_t)P, 0);
    long long n_mine = 0;
    for (int I = 0; I < nb; ++I) {
        first[I] = (int)strips.size();
        for (int J = I; J < nb; J += strip_len) {
            const int len = std::min(strip_len, nb - J);
            const int owner = (int)(std::min_element(held.begin(), held.end()) - held.begin());
            held[owner] += len;
            if (owner != rank) continue;
            strips.push_back(make_int4(I, J, len, (int)(sb.spread ? n_mine : symm_tile_index(nb, I, J))));
            if (sb.spread)
                for (int q = 0; q < len; ++q) tile_local[(size_t)symm_tile_index(nb, I, J + q)] = (int)(n_mine + q);
            n_mine += len;
        }
    }
    first[nb] = (int)strips.size();
    sb.n_tiles = n_mine;
    sb.n_strips = (int)strips.size();
    if (!sb.jagged) KMCF_TRY(ensure(&sb.d_tiles, &sb.cap_tiles, (size_t)std::max<long long>(sb.n_tiles, 1) * 4096));
    KMCF_TRY(ensure(&sb.d_strips, &sb.cap_strips, strips.size() + 1));
    KMCF_TRY(ensure(&sb.d_strip_first, &sb.cap_sf, first.size()));
    KMCF_TRY(ensure(&sb.d_rowpart, &sb.cap_rowpart, (size_t)(sb.n_strips + 1) * 128));
    KMCF_TRY(ensure(&sb.d_colpart, &sb.cap_colpart, (size_t)std::max<long long>(sb.n_tiles, 1) * 128));
    if (!strips.empty()) KMCF_HIP(hipMemcpyAsync(sb.d_strips, strips.data(), strips.size() * sizeof(int4), hipMemcpyHostToDevice, st));
    KMCF_HIP(hipMemcpyAsync(sb.d_strip_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (sb.spread) {
        KMCF_TRY(ensure(&sb.d_tile_local, &sb.cap_tile_local, tile_local.size() + 1));
        KMCF_TRY(ensure(&sb.d_ypart, &sb.cap_ypart, (size_t)2 * P * 64 * nb));
        KMCF_HIP(hipMemcpyAsync(sb.d_tile_local, tile_local.data(), tile_local.size() * sizeof(int), hipMemcpyHostToDevice, st));
        for (int w = 1; w <= 2; ++w) {                              // slices of the partials' all-gather: one / two sums per point
            sb.y_counts[w - 1].assign((size_t)P, w * 64 * nb);
            sb.y_displs[w - 1].resize((size_t)P);
            for (int q = 0; q < P; ++q) sb.y_displs[w - 1][q] = q * w * 64 * nb;
        }
    }
    KMCF_HIP(hipStreamSynchronize(st));                            // (the host vectors go out of scope)
    const int grid = std::max(1, std::min((sb.n_strips + 3) / 4, 8192));
    if (sb.jagged) {
        // masks tile-major, entries per tile -> first value of every tile -> values in layer order
        KMCF_TRY(ensure(&sb.d_jmask, &sb.cap_jmask, (size_t)sb.n_tiles * 64));
        KMCF_TRY(ensure(&sb.d_jcnt, &sb.cap_jcnt, (size_t)sb.n_tiles + 1));
        KMCF_TRY(ensure(&sb.d_jvoff, &sb.cap_jvoff, (size_t)sb.n_tiles + 2));
        symj_mask_kernel<<<grid, KMCF_BLOCK, 0, st>>>(sb.n_strips, sb.d_strips, n_t, sb.n_groups, sb.d_mask, sb.d_jmask, sb.d_jcnt);
        long long *pin_j = reinterpret_cast<long long *>(t->h_pin + 6);
        scan_exclusive_kernel<long long><<<1, KMCF_BLOCK, 0, st>>>((int)sb.n_tiles, sb.d_jcnt, sb.d_jvoff, nullptr);
        KMCF_HIP(hipGetLastError());
        KMCF_HIP(hipMemcpyAsync(pin_j, sb.d_jvoff + sb.n_tiles, sizeof(long long), hipMemcpyDeviceToHost, st));
        KMCF_HIP(hipStreamSynchronize(st));
        sb.jnnz = *pin_j;
        KMCF_TRY(ensure(&sb.d_jval, &sb.cap_jval, (size_t)sb.jnnz + 64));
        symj_fill_kernel<<<grid, KMCF_BLOCK, 0, st>>>(sb.n_strips, sb.d_strips, n_t, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, p->tol, p->m_e,
                                                     p->V0, sb.d_jmask, sb.d_jvoff, sb.d_jval);
    } else {
        tunnel_dense_fill_kernel<<<grid, KMCF_BLOCK, 0, st>>>(sb.n_strips, sb.d_strips, n_t, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, t->nn_dist,
                                                             p->tol, p->m_e, p->V0, sb.d_tiles);
    }
    KMCF_HIP(hipGetLastError());
    // diagonal = -(row sums): one application to the vector of ones (the diagonal entries are still 0); spread: of ALL
    // points on every rank (a rank sets the diagonal of the diagonal tiles it holds)
    fill_kernel<<<grid1d(64 * nb), KMCF_BLOCK, 0, st>>>(64 * nb, sb.d_xsub, 1.0);
    KMCF_HIP(hipMemsetAsync(t->d_tdiag, 0, (size_t)n_t * sizeof(double), st));
    KMCF_TRY(symm_launch<0>(sb, sb.d_xsub, 0.0, st));
    if (sb.spread)
        KMCF_TRY((symm_spread_finish<0, false>(c, sb, n_t, 0, nullptr, nullptr, t->d_tdiag, nullptr, nullptr, nullptr, 0)));
    else
        sub_symm_reduce_kernel<0, false><<<nb, KMCF_BLOCK, 0, st>>>(n_t, nb, sb.d_strip_first, sb.d_rowpart, sb.d_colpart, nullptr, nullptr,
                                                                            t->d_tdiag, nullptr, nullptr, nullptr, 0);
    if (sb.jagged) symj_set_diag_kernel<<<std::max(1, (nb + 3) / 4), KMCF_BLOCK, 0, st>>>(n_t, nb, t->d_tdiag, t->d_tdiag, sb.d_jmask, sb.d_jvoff, sb.d_jval);
    else symm_set_diag_kernel<<<grid1d(n_t), KMCF_BLOCK, 0, st>>>(n_t, nb, t->d_tdiag, t->d_tdiag, sb.d_tiles, sb.spread ? sb.d_tile_local : nullptr);
    KMCF_HIP(hipMemsetAsync(sb.d_xsub, 0, (size_t)64 * nb * sizeof(double), st));      // the pad behind the last point stays 0
    KMCF_HIP(hipGetLastError());
    // partials of the p.Ap sum: one per block row (reduce kernel) / per 256 own points (combine kernel)
    sb.grid = sb.spread ? (sb.n_loc + KMCF_BLOCK - 1) / KMCF_BLOCK : nb;
    return KMCF_OK;
}

static int t_assemble_async(kmcf_tstate *t, const int *d_site_element, const int *d_site_charge, const double *d_site_CB_edge,
                            const int *d_metals, int num_metals, const kmcf_current_params_t *p)
{
    kmcf_comm *c = t->comm;
    kmcf_matrix *m = t->T;
    hipStream_t st = c->stream;
    const int Na = t->N_atom, n_loc = m->n_loc, P = c->nranks, rank = c->rank;
    t->par = *p;
    t->assembled = false;
    // 1. atom arrays (update_atom_arrays, :1341-1365) + invariance of the atom set
    KMCF_HIP(hipMemsetAsync(t->d_err, 0, sizeof(int), st));
    check_atom_set_kernel<<<grid1d(t->N), KMCF_BLOCK, 0, st>>>(t->N, d_site_element, t->d_site_is_atom, t->d_err);
    gather_atoms_kernel<<<grid1d(Na), KMCF_BLOCK, 0, st>>>(Na, t->d_atom_site, d_site_element, d_site_charge, d_site_CB_edge, d_metals,
                                                           num_metals, t->d_ael, t->d_ach, t->d_acb, t->d_acls);
    KMCF_HIP(hipGetLastError());
    // 2. neighbour values + diagonal (one pass)
    if (n_loc > 0) {
        const int n_cols = n_loc + m->n_halo;
        t_cls_col_kernel<<<grid1d(n_cols), KMCF_BLOCK, 0, st>>>(n_cols, t->d_col_node, t->d_acls, t->d_cls_col);
        const double dict[3] = {-p->high_G, -p->low_G, -p->loop_G};
        KMCF_TRY(kmcf_matrix_set_dictionary(m, dict, 3));
        constexpr int LPR = 16;
        t_assemble_kernel<LPR><<<grid1d((int64_t)n_loc * LPR), KMCF_BLOCK, 0, st>>>(
            n_loc, m->d_row_ptr, m->d_col, m->d_val, t->d_diag_pos, t->d_ground, t->d_cls_col, p->high_G, p->low_G, p->loop_G,
            t->d_diag, m->coded ? m->d_idx16 : nullptr, m->coded ? m->d_diagv : nullptr, m->n_short);
        KMCF_HIP(hipGetLastError());
    }
    // 3. tunnel points of all ranks (get_is_tunnel_mpi + copy_if + MPI_Allgatherv, initialize_sparsity_T.cu:739-787)
    const int n_scan = Na - 1;                                        // atoms 0 .. N_atom - 2 are matrix rows
    const int nblk = (Na + KMCF_BLOCK * SCAN_ITEMS - 1) / (KMCF_BLOCK * SCAN_ITEMS);
    tunnel_count_kernel<<<nblk, KMCF_BLOCK, 0, st>>>(n_scan, t->d_ael, t->d_ax, p->contact_x_lo, p->contact_x_hi, t->d_blk);
    scan_exclusive_kernel<int><<<1, KMCF_BLOCK, 0, st>>>(nblk, t->d_blk, t->d_blk + nblk, nullptr);
    tunnel_scatter_kernel<<<nblk, KMCF_BLOCK, 0, st>>>(n_scan, Na, t->d_ael, t->d_ax, t->d_ay, t->d_az, t->d_acb, d_metals, num_metals,
                                                       p->contact_x_lo, p->contact_x_hi, t->n_layers, t->n_inj, t->n_ext, t->d_blk + nblk,
                                                       t->d_tidx, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipMemcpyAsync(t->h_pin, t->d_blk + 2 * nblk, sizeof(int), hipMemcpyDeviceToHost, st));   // total = n_t
    KMCF_HIP(hipMemcpyAsync(t->h_pin + 1, t->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    KMCF_HIP(hipStreamSynchronize(st));
    KMCF_CHECK(t->h_pin[1] == 0, KMCF_ERR_STATE,
               "kmcf_t_assemble: the set of atom sites differs from the one at kmcf_initialize_sparsity_T (call it again)");
    const int n_t = t->h_pin[0];
    t->h_tidx.resize((size_t)n_t);
    if (n_t > 0) KMCF_HIP(hipMemcpy(t->h_tidx.data(), t->d_tidx, (size_t)n_t * sizeof(int), hipMemcpyDeviceToHost));
    // counts_subblock / displ_subblock (:752-758): tunnel points by owner of their matrix row (atom index + 2)
    kmcf_subop &sb = t->sub;
    for (int q = 0; q < P; ++q) {
        const int lo = m->displs[q], hi = m->displs[q] + m->counts[q];
        const int b = (int)(std::lower_bound(t->h_tidx.begin(), t->h_tidx.end(), lo - 2) - t->h_tidx.begin());
        const int e = (int)(std::lower_bound(t->h_tidx.begin(), t->h_tidx.end(), hi - 2) - t->h_tidx.begin());
        sb.displs[q] = b;
        sb.counts[q] = e - b;
    }
    sb.n_glob = n_t;
    sb.n_loc = sb.counts[rank];
    sb.row0 = sb.displs[rank];
    sb.n_groups = (n_t + 63) / 64;
    sb.nnz = 0;
    const int ns = sb.n_loc, ng = sb.n_groups;
    KMCF_TRY(ensure(&sb.d_rows, &sb.cap_rows, (size_t)ns + 1));
    KMCF_TRY(ensure(&sb.d_voff, &sb.cap_voff, (size_t)ns + 2));
    KMCF_HIP(hipMemsetAsync(sb.d_voff, 0, ((size_t)ns + 2) * sizeof(long long), st));
    KMCF_TRY(ensure(&sb.d_mask, &sb.cap_mask, (size_t)ns * ng + 1));
    KMCF_TRY(ensure(&sb.d_xsub, &sb.cap_x, (size_t)n_t + 320));          // + 256: the 4-group unroll reads x_sub[j + 192] only when set
    KMCF_TRY(ensure(&t->d_rowcnt, &t->cap_rowcnt, (size_t)ns + 1));
    KMCF_TRY(ensure(&t->d_tdiag, &t->cap_tdiag, (size_t)n_t + 1));       // (per LOCAL point; the spread tiles: per point of all ranks)
    const int wpb = KMCF_BLOCK / 64;
    const int wgrid = std::max(1, std::min((ns + wpb - 1) / wpb, KMCF_MAX_PARTIALS));
    sb.grid = ns > 0 ? wgrid : 0;
    sb.dense = sb.jagged = sb.spread = false;
    if (ns > 0) {
        sub_rows_kernel<<<grid1d(ns), KMCF_BLOCK, 0, st>>>(ns, sb.row0, t->d_tidx, m->row0, t->d_inv_perm, sb.d_rows);
        tunnel_mask_kernel<<<wgrid, KMCF_BLOCK, 0, st>>>(ns, sb.row0, n_t, ng, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, t->nn_dist,
                                                         p->tol, sb.d_mask, t->d_rowcnt);
        long long *pin_nnz = reinterpret_cast<long long *>(t->h_pin + 2);
        scan_exclusive_kernel<long long><<<1, KMCF_BLOCK, 0, st>>>(ns, t->d_rowcnt, sb.d_voff, nullptr);
        KMCF_HIP(hipGetLastError());
        KMCF_HIP(hipMemcpyAsync(pin_nnz, sb.d_voff + ns, sizeof(long long), hipMemcpyDeviceToHost, st));
        KMCF_HIP(hipStreamSynchronize(st));
        sb.nnz = *pin_nnz;
    }
    // Storage of the block: dense symmetric tiles when the block is more than a QUARTER full, else the bitmap + packed
    // values.  In bytes the tiles (4 n^2) win from half full on (8 d n^2 + n^2 / 8 for the bitmap form); in time from a
    // quarter on: the tile kernel streams at 5.3 TB/s, the bitmap kernel -- a load of 64 x d values per mask word -- at 2.7
    // (the reference's contact window at 40 nm, 44 % full: 3.8 against 6.3 ms per application).  Not when the tiles would
    // take more than 60 % of the free device memory.
    // Round 4: the same tiles holding only their ENTRIES (jagged: 4 B per entry of the full block + 1 bit per position,
    // the same sums bit for bit, 2.2 x less memory at 44 % -- but bound by its instruction count, not its bytes: the
    // reference's window on the 4 x 4-cell device, 17 722 points, 44 % full: dense tiles 0.28, jagged tiles 0.41,
    // bitmap 0.49 ms per iteration) where the dense tiles do not fit the device's memory (one rank).
    // A rank GROUP (round 4): the dense tiles' strips dealt to the ranks (kmcf_subop::spread) under the same rule,
    // weighed on the whole block -- the ranks agree through one small all-gather (entries of their rows, and whether
    // their share of the tiles fits) so that all of them take the same branch.
    // KMCF_SUB_DENSE = 0 bitmap / 1 dense tiles / 2 jagged tiles (one rank; a group: dense tiles) overrides.
    if (n_t > 0) {
        const long long nbl = (n_t + 63) / 64, all_tiles = nbl * (nbl + 1) / 2;
        const double nn2 = (double)n_t * (double)n_t, nbt = nn2 / 8192;
        const char *env = getenv("KMCF_SUB_DENSE");
        if (P == 1) {
            sb.dense = n_t >= 2048 && 4.0 * (double)sb.nnz > nn2;
            if (sb.dense && sb.cap_tiles < (size_t)all_tiles * 4096) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && 4.0 * nn2 + 1024.0 * nbt > 0.6 * (double)fr) {
                    sb.jagged = 4.0 * (double)sb.nnz + 1544.0 * nbt <= 0.6 * (double)fr + 8.0 * (double)sb.cap_jval;
                    sb.dense = sb.jagged;
                }
            }
            if (env) { sb.dense = atoi(env) != 0; sb.jagged = sb.dense && atoi(env) == 2; }
            if (nbl > KMCF_MAX_PARTIALS) sb.dense = sb.jagged = false;       // (one p.Ap partial per block row)
        } else if (!(env && atoi(env) == 0)) {
            // what this rank would hold: every P-th strip; 32 KB a tile + its parts
            const double share = ((double)all_tiles / P + nbl) * (32768.0 + 1024.0);
            size_t fr = 0, tot = 0;
            const bool fits = sb.cap_tiles * sizeof(double) >= share || (hipMemGetInfo(&fr, &tot) == hipSuccess && share <= 0.6 * (double)fr);
            double *h_ag = reinterpret_cast<double *>(t->h_agree);
            h_ag[0] = (double)sb.nnz; h_ag[1] = fits ? 1.0 : 0.0;
            KMCF_HIP(hipMemcpyAsync(t->d_agree + 2 * rank, h_ag, 2 * sizeof(double), hipMemcpyHostToDevice, st));
            std::vector<int> cnt((size_t)P, 2), dsp((size_t)P);
            for (int q = 0; q < P; ++q) dsp[q] = 2 * q;
            KMCF_TRY(kmcf_comm_allgatherv_double(c, t->d_agree, cnt.data(), dsp.data()));
            KMCF_HIP(hipMemcpyAsync(h_ag + 2, t->d_agree, (size_t)2 * P * sizeof(double), hipMemcpyDeviceToHost, st));
            KMCF_HIP(hipStreamSynchronize(st));
            double nnz_all = 0.0;
            bool fit_all = true;
            for (int q = 0; q < P; ++q) { nnz_all += h_ag[2 + 2 * q]; fit_all = fit_all && h_ag[3 + 2 * q] != 0.0; }
            sb.dense = sb.spread = fit_all && (ns + KMCF_BLOCK - 1) / KMCF_BLOCK <= KMCF_MAX_PARTIALS &&
                                   (env ? atoi(env) != 0 : n_t >= 2048 && 4.0 * nnz_all > nn2);
        }
        if (sb.dense) {
            KMCF_TRY(symm_setup(t));
        } else if (ns > 0) {
            KMCF_TRY(ensure(&sb.d_val, &sb.cap_val, (size_t)sb.nnz + 64));
            tunnel_value_kernel<<<wgrid, KMCF_BLOCK, 0, st>>>(ns, sb.row0, ng, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, t->nn_dist, p->tol,
                                                              p->m_e, p->V0, sb.d_mask, sb.d_voff, sb.d_val, t->d_tdiag);
            KMCF_HIP(hipGetLastError());
        }
    }
    // 4. preconditioner and right-hand side
    if (n_loc > 0) {
        copy_diag_rhs_kernel<<<grid1d(n_loc), KMCF_BLOCK, 0, st>>>(n_loc, t->d_diag, t->d_diag_tot, t->d_col_node, p->loop_G * p->Vd, t->d_rhs);
        if (ns > 0) add_tunnel_diag_kernel<<<grid1d(ns), KMCF_BLOCK, 0, st>>>(ns, sb.d_rows, t->d_tdiag + (sb.spread ? sb.row0 : 0), t->d_diag_tot);
        invert_kernel<<<grid1d(n_loc), KMCF_BLOCK, 0, st>>>(n_loc, t->d_diag_tot, m->d_dinv);
        KMCF_HIP(hipGetLastError());
    }
    m->sub = &t->sub;
    t->assembled = true;
    return KMCF_OK;
}

// Sub-block part of a split SpMV.  begin (called by kmcf_spmv_device BEFORE the CSR part): pack the local part of the
// sub-vector (pack_gpu, dist_spmv_split_sparse.cpp:29-33) and start gathering the rest (:36-48) -- for a group on RCCL
// or the peer-to-peer transport on the COMM stream, so that the exchange runs underneath the neighbour part
// (spmm_split_sparse2 / 3, :123-192, :246-337); finish (after the CSR part): wait for it, y[sub rows] += S x_sub.
int kmcf_subop_begin(kmcf_matrix *m, bool skip_if_done)
{
    kmcf_subop *sb = m->sub;
    kmcf_comm *c = m->comm;
    hipStream_t st = c->stream;
    sb->gather_pending = false;
    if (sb->n_glob == 0) return KMCF_OK;
    const int chk = skip_if_done ? 1 : 0;
    if (sb->n_loc > 0) {
        sub_pack_kernel<<<grid1d(sb->n_loc), KMCF_BLOCK, 0, st>>>(sb->n_loc, sb->d_rows, m->d_p, sb->d_xsub + sb->row0, m->d_S, chk);
        KMCF_HIP(hipGetLastError());
    }
    if (c->nranks == 1 && !c->force_collectives) return KMCF_OK;
    static const bool no_overlap = getenv("KMCF_SUB_OVERLAP") && atoi(getenv("KMCF_SUB_OVERLAP")) == 0;
    if (!no_overlap && (c->p2p_active || (!c->group && c->nccl))) {
        KMCF_HIP(hipEventRecord(c->ev_subpack, st));
        KMCF_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_subpack, 0));
        const int rc = kmcf_comm_allgatherv_double_comm_stream(c, sb->d_xsub, sb->counts.data(), sb->displs.data());
        if (rc == KMCF_OK) {
            KMCF_HIP(hipEventRecord(c->ev_sub, c->comm_stream));
            sb->gather_pending = true;
            return KMCF_OK;
        }
        if (rc != KMCF_ERR_STATE) return rc;
    }
    return kmcf_comm_allgatherv_double(c, sb->d_xsub, sb->counts.data(), sb->displs.data());     // in order, on the compute stream
}

int kmcf_subop_finish(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    kmcf_subop *sb = m->sub;
    kmcf_comm *c = m->comm;
    hipStream_t st = c->stream;
    if (sb->n_glob == 0) return KMCF_OK;
    const int chk = skip_if_done ? 1 : 0;
    if (sb->gather_pending) {
        KMCF_HIP(hipStreamWaitEvent(st, c->ev_sub, 0));
        sb->gather_pending = false;
    }
    double *part = m->d_part_a + 3 * KMCF_MAX_PARTIALS;
    if (sb->spread) {                                                          // (every rank: also one without points of its own)
        KMCF_TRY(symm_launch<0>(*sb, sb->d_xsub, 0.0, st, m->d_S, chk));
        if (with_dot) return symm_spread_finish<0, true>(c, *sb, sb->n_loc, sb->row0, sb->d_rows, m->d_p, m->d_Ap, nullptr, part, m->d_S, chk);
        return symm_spread_finish<0, false>(c, *sb, sb->n_loc, sb->row0, sb->d_rows, m->d_p, m->d_Ap, nullptr, part, m->d_S, chk);
    }
    if (sb->n_loc == 0) return KMCF_OK;
    if (sb->dense) {
        KMCF_TRY(symm_launch<0>(*sb, sb->d_xsub, 0.0, st, m->d_S, chk));       // (behind the stop: neither pass runs)
        if (with_dot)
            sub_symm_reduce_kernel<0, true><<<sb->grid, KMCF_BLOCK, 0, st>>>(sb->n_glob, sb->nb, sb->d_strip_first, sb->d_rowpart, sb->d_colpart,
                                                                            sb->d_rows, m->d_p, m->d_Ap, nullptr, part, m->d_S, chk);
        else
            sub_symm_reduce_kernel<0, false><<<sb->grid, KMCF_BLOCK, 0, st>>>(sb->n_glob, sb->nb, sb->d_strip_first, sb->d_rowpart, sb->d_colpart,
                                                                             sb->d_rows, m->d_p, m->d_Ap, nullptr, part, m->d_S, chk);
        KMCF_HIP(hipGetLastError());
        return KMCF_OK;
    }
    if (with_dot)
        sub_spmv_kernel<true><<<sb->grid, KMCF_BLOCK, 0, st>>>(sb->n_loc, sb->n_glob, sb->n_groups, sb->d_mask, sb->d_voff, sb->d_val, sb->d_xsub,
                                                              sb->d_rows, m->d_p, m->d_Ap, part, m->d_S, chk);
    else
        sub_spmv_kernel<false><<<sb->grid, KMCF_BLOCK, 0, st>>>(sb->n_loc, sb->n_glob, sb->n_groups, sb->d_mask, sb->d_voff, sb->d_val, sb->d_xsub,
                                                               sb->d_rows, m->d_p, m->d_Ap, part, m->d_S, chk);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

static int check_params(const kmcf_current_params_t *p)
{
    KMCF_CHECK(p, KMCF_ERR_ARG, "null parameter block");
    KMCF_CHECK(p->high_G > 0 && p->low_G > 0 && p->loop_G > 0 && p->m_e > 0 && p->tol >= 0, KMCF_ERR_ARG, "non-positive conductance / mass");
    KMCF_CHECK(p->cg_max_iterations >= 0, KMCF_ERR_ARG, "negative iteration limit");
    return KMCF_OK;
}

extern "C" int kmcf_t_assemble(kmcf_tstate *t, const int *d_site_element, const int *d_site_charge, const double *d_site_CB_edge,
                               const int *d_metals, int num_metals, const kmcf_current_params_t *p)
{
    KMCF_CHECK(t && d_site_element && d_site_charge && d_site_CB_edge && d_metals, KMCF_ERR_ARG, "kmcf_t_assemble: null argument");
    KMCF_TRY(check_params(p));
    KMCF_TRY(kmcf_enter(t->comm));
    KMCF_TRY(t_assemble_async(t, d_site_element, d_site_charge, d_site_CB_edge, d_metals, num_metals, p));
    KMCF_HIP(hipStreamSynchronize(t->comm->stream));
    return KMCF_OK;
}

extern "C" int kmcf_tstate_get_vectors(const kmcf_tstate *t, double *h_diag_neighbour, double *h_dinv, double *h_rhs)
{
    KMCF_CHECK(t, KMCF_ERR_ARG, "kmcf_tstate_get_vectors: null state");
    KMCF_CHECK(t->assembled, KMCF_ERR_STATE, "kmcf_tstate_get_vectors: call kmcf_t_assemble first");
    KMCF_TRY(kmcf_enter(t->comm));
    KMCF_HIP(hipStreamSynchronize(t->comm->stream));
    const int n = t->T->n_loc;
    std::vector<double> tmp((size_t)std::max(n, 1));
    const double *src[3] = {t->d_diag, t->T->d_dinv, t->d_rhs};
    double *dst[3] = {h_diag_neighbour, h_dinv, h_rhs};
    for (int v = 0; v < 3; ++v) {
        if (!dst[v] || n == 0) continue;
        KMCF_HIP(hipMemcpy(tmp.data(), src[v], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) dst[v][t->T->h_perm.empty() ? i : t->T->h_perm[i]] = tmp[i];
    }
    return KMCF_OK;
}

extern "C" int kmcf_tstate_get_tunnel(const kmcf_tstate *t, int *h_tunnel_idx, int *h_row_ptr, int *h_col, double *h_val, double *h_diag)
{
    KMCF_CHECK(t, KMCF_ERR_ARG, "kmcf_tstate_get_tunnel: null state");
    KMCF_CHECK(t->assembled, KMCF_ERR_STATE, "kmcf_tstate_get_tunnel: call kmcf_t_assemble first");
    KMCF_TRY(kmcf_enter(t->comm));
    KMCF_HIP(hipStreamSynchronize(t->comm->stream));
    const kmcf_subop &sb = t->sub;
    if (h_tunnel_idx && sb.n_glob) memcpy(h_tunnel_idx, t->h_tidx.data(), (size_t)sb.n_glob * sizeof(int));
    const int ns = sb.n_loc, ng = sb.n_groups;
    if (h_diag && ns) KMCF_HIP(hipMemcpy(h_diag, t->d_tdiag + (sb.spread ? sb.row0 : 0), (size_t)ns * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> tiles;
    if (h_val && sb.nnz && sb.jagged) {                      // the same out of the layers of the jagged tiles
        tiles.assign((size_t)sb.n_tiles * 4096, 0.0);
        std::vector<unsigned long long> jm((size_t)sb.n_tiles * 64);
        std::vector<long long> jo((size_t)sb.n_tiles + 1);
        std::vector<double> jv((size_t)sb.jnnz + 1);
        KMCF_HIP(hipMemcpy(jm.data(), sb.d_jmask, jm.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        KMCF_HIP(hipMemcpy(jo.data(), sb.d_jvoff, jo.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (sb.jnnz) KMCF_HIP(hipMemcpy(jv.data(), sb.d_jval, (size_t)sb.jnnz * sizeof(double), hipMemcpyDeviceToHost));
        for (long long tl = 0; tl < sb.n_tiles; ++tl) {
            unsigned long long m[64];
            for (int r = 0; r < 64; ++r) m[r] = jm[(size_t)tl * 64 + r];
            long long pos = jo[(size_t)tl];
            for (bool any = true; any;) {                    // layer by layer, rows ascending
                any = false;
                for (int r = 0; r < 64; ++r)
                    if (m[r]) {
                        tiles[(size_t)tl * 4096 + (size_t)r * 64 + __builtin_ctzll(m[r])] = jv[(size_t)pos++];
                        m[r] &= m[r] - 1;
                        any = true;
                    }
            }
        }
    } else if (h_val && sb.nnz && sb.spread) {
        // the tiles of this rank's rows lie on all ranks: the rows' entries are computed afresh as the bitmap storage
        // would (the same values: one symmetric expression per pair), the diagonal entries are the ones in use
        double *d_v = nullptr, *d_dg = nullptr;
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_v), ((size_t)sb.nnz + 64) * sizeof(double)));
        if (hipMalloc(reinterpret_cast<void **>(&d_dg), ((size_t)ns + 1) * sizeof(double)) != hipSuccess) { hipFree(d_v); KMCF_HIP(hipErrorOutOfMemory); }
        const int wgrid = std::max(1, std::min((ns + 3) / 4, KMCF_MAX_PARTIALS));
        tunnel_value_kernel<<<wgrid, KMCF_BLOCK, 0, t->comm->stream>>>(ns, sb.row0, ng, t->d_tinfo, t->d_tx, t->d_ty, t->d_tz, t->d_tcb, t->nn_dist,
                                                                      t->par.tol, t->par.m_e, t->par.V0, sb.d_mask, sb.d_voff, d_v, d_dg);
        hipError_t e1 = hipGetLastError();
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(t->comm->stream);
        if (e1 == hipSuccess) e1 = hipMemcpy(h_val, d_v, (size_t)sb.nnz * sizeof(double), hipMemcpyDeviceToHost);
        hipFree(d_v); hipFree(d_dg);
        KMCF_HIP(e1);
    } else if (h_val && sb.nnz && sb.dense) {                // values out of the upper tiles (test-sized blocks)
        tiles.resize((size_t)sb.n_tiles * 4096);
        KMCF_HIP(hipMemcpy(tiles.data(), sb.d_tiles, tiles.size() * sizeof(double), hipMemcpyDeviceToHost));
    } else if (h_val && sb.nnz) {
        KMCF_HIP(hipMemcpy(h_val, sb.d_val, (size_t)sb.nnz * sizeof(double), hipMemcpyDeviceToHost));   // packed = CSR order
    }
    if (h_row_ptr || h_col || !tiles.empty() || (sb.spread && h_val)) {
        std::vector<double> tdiag_now;
        if (sb.spread && h_val && ns) {
            tdiag_now.resize((size_t)ns);
            KMCF_HIP(hipMemcpy(tdiag_now.data(), t->d_tdiag + sb.row0, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost));
        }
        std::vector<unsigned long long> mk((size_t)ns * ng + 1);
        if (ns) KMCF_HIP(hipMemcpy(mk.data(), sb.d_mask, (size_t)ns * ng * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        int64_t pos = 0;
        if (h_row_ptr) h_row_ptr[0] = 0;
        for (int s = 0; s < ns; ++s) {
            for (int g = 0; g < ng; ++g) {
                unsigned long long w = mk[(size_t)s * ng + g];
                while (w) {
                    const int b = __builtin_ctzll(w);
                    if (h_col) h_col[pos] = g * 64 + b;
                    if (sb.spread && h_val && g * 64 + b == sb.row0 + s) h_val[pos] = tdiag_now[(size_t)s];
                    if (!tiles.empty()) {
                        const int i = sb.row0 + s, j = g * 64 + b, lo = std::min(i, j), hi = std::max(i, j);
                        h_val[pos] = tiles[(size_t)symm_tile_index(sb.nb, lo >> 6, hi >> 6) * 4096 + (size_t)(lo & 63) * 64 + (hi & 63)];
                    }
                    ++pos;
                    w &= w - 1;
                }
            }
            if (h_row_ptr) h_row_ptr[s + 1] = (int)pos;
        }
    }
    return KMCF_OK;
}

extern "C" int kmcf_update_power_sparse(kmcf_tstate *t, const int *d_site_element, const int *d_site_charge,
                                        const double *d_site_CB_edge, const int *d_metals, int num_metals,
                                        double *d_atom_virtual_potentials, double *d_site_power,
                                        const kmcf_current_params_t *p, double *imacro, kmcf_solve_stats_t *stats)
{
    KMCF_CHECK(t && d_site_element && d_site_charge && d_site_CB_edge && d_metals && d_atom_virtual_potentials, KMCF_ERR_ARG,
               "kmcf_update_power_sparse: null argument");
    KMCF_TRY(check_params(p));
    KMCF_CHECK(!p->solve_heating || d_site_power, KMCF_ERR_ARG, "kmcf_update_power_sparse: solve_heating needs d_site_power");
    kmcf_comm *c = t->comm;
    kmcf_matrix *m = t->T;
    KMCF_CHECK(c->connected, KMCF_ERR_COMM, "kmcf_update_power_sparse: communicator not connected");
    KMCF_TRY(kmcf_enter(c));
    hipStream_t st = c->stream;
    const int n_loc = m->n_loc, Na = t->N_atom;
    KMCF_HIP(hipEventRecord(c->ev_a0, st));
    KMCF_TRY(t_assemble_async(t, d_site_element, d_site_charge, d_site_CB_edge, d_metals, num_metals, p));
    KMCF_HIP(hipEventRecord(c->ev_a1, st));
    // the initial guess is the current content of atom_virtual_potentials, solved in place (:1463)
    double *x_user = d_atom_virtual_potentials + m->row0;
    if (n_loc > 0) KMCF_HIP(hipMemcpyAsync(m->d_r, t->d_rhs, (size_t)n_loc * sizeof(double), hipMemcpyDeviceToDevice, st));
    KMCF_TRY(kmcf_vec_in(m, m->d_x, x_user));
    KMCF_TRY(kmcf_pcg_workspace(m, true, p->cg_tolerance, p->cg_max_iterations, 0, stats));     // :1674
    KMCF_TRY(kmcf_vec_out(m, x_user, m->d_x));
    // every rank gets all Nsub potentials (the reference gathers them on rank 0 only, :1782)
    KMCF_TRY(kmcf_comm_allgatherv_double(c, d_atom_virtual_potentials, m->counts.data(), m->displs.data()));
    // scaled by G0 in place, all N_atom + 2 entries (:2038-2040)
    scale_kernel<<<grid1d(Na + 2), KMCF_BLOCK, 0, st>>>(Na + 2, d_atom_virtual_potentials, p->G0);
    // I_macro on the owner of row 1, then shared
    const bool own1 = m->row0 <= 1 && 1 < m->row0 + n_loc;
    imacro_kernel<<<1, KMCF_BLOCK, 0, st>>>(own1 ? t->h_inv_perm[1 - m->row0] : -1, m->d_row_ptr, m->d_col, m->d_val, t->d_col_node,
                                            d_atom_virtual_potentials, t->d_scal);
    KMCF_HIP(hipGetLastError());
    KMCF_TRY(kmcf_comm_allreduce_sum(c, t->d_scal, 1));
    // (into pinned memory that outlives this frame: the error returns below leave before the synchronisation)
    double *h_im = reinterpret_cast<double *>(t->h_pin + 4);
    KMCF_HIP(hipMemcpyAsync(h_im, t->d_scal, sizeof(double), hipMemcpyDeviceToHost, st));
    if (p->solve_heating) {
        // shift (:2068-2071), forward currents and their row sums (:2086-2098), P = I_neg m (:2100-2131), copy_pdisp (:2137)
        min_kernel<<<1, KMCF_BLOCK, 0, st>>>(Na + 2, d_atom_virtual_potentials, t->d_scal + 1);
        shift_kernel<<<grid1d(Na + 2), KMCF_BLOCK, 0, st>>>(Na + 2, d_atom_virtual_potentials, t->d_scal + 1);
        double *psum = m->d_r, *isum = m->d_Ap;                          // workspace vectors, free after the solve
        if (n_loc > 0) {
            power_neighbour_kernel<16><<<grid1d((int64_t)n_loc * 16), KMCF_BLOCK, 0, st>>>(
                n_loc, m->d_row_ptr, m->d_col, m->d_val, t->d_diag_pos, t->d_col_node, d_atom_virtual_potentials, p->Vd, psum, isum);
            KMCF_HIP(hipGetLastError());
        }
        if (t->sub.spread) {                                               // (every rank of the group: its strips, then the gather)
            kmcf_subop &sb = t->sub;
            gather_tunnel_pot_kernel<<<grid1d(sb.n_glob), KMCF_BLOCK, 0, st>>>(sb.n_glob, t->d_tidx, d_atom_virtual_potentials, sb.d_xsub);
            KMCF_TRY(symm_launch<1>(sb, sb.d_xsub, p->Vd, st));
            KMCF_TRY((symm_spread_finish<1, false>(c, sb, sb.n_loc, sb.row0, sb.d_rows, nullptr, psum, isum, nullptr, nullptr, 0)));
            KMCF_HIP(hipMemsetAsync(sb.d_xsub + sb.n_glob, 0, (size_t)(64 * sb.nb - sb.n_glob) * sizeof(double), st));
        }
        if (n_loc > 0) {
            if (t->sub.n_loc > 0 && t->sub.dense && !t->sub.spread) {
                kmcf_subop &sb = t->sub;
                gather_tunnel_pot_kernel<<<grid1d(sb.n_glob), KMCF_BLOCK, 0, st>>>(sb.n_glob, t->d_tidx, d_atom_virtual_potentials, sb.d_xsub);
                KMCF_TRY(symm_launch<1>(sb, sb.d_xsub, p->Vd, st));
                sub_symm_reduce_kernel<1, false><<<sb.nb, KMCF_BLOCK, 0, st>>>(sb.n_glob, sb.nb, sb.d_strip_first, sb.d_rowpart, sb.d_colpart,
                                                                                          sb.d_rows, nullptr, psum, isum, nullptr, nullptr, 0);
                KMCF_HIP(hipMemsetAsync(sb.d_xsub + sb.n_glob, 0, (size_t)(64 * sb.nb - sb.n_glob) * sizeof(double), st));
            } else if (t->sub.n_loc > 0 && !t->sub.dense)
                power_tunnel_kernel<<<t->sub.grid, KMCF_BLOCK, 0, st>>>(t->sub.n_loc, t->sub.row0, t->sub.n_groups, t->sub.d_mask, t->sub.d_voff,
                                                                       t->sub.d_val, t->d_tidx, t->sub.d_rows, d_atom_virtual_potentials, p->Vd,
                                                                       psum, isum);
            power_finish_kernel<<<grid1d(n_loc), KMCF_BLOCK, 0, st>>>(n_loc, t->d_col_node, d_atom_virtual_potentials, psum, isum, t->d_pdisp);
        }
        KMCF_HIP(hipGetLastError());
        KMCF_TRY(kmcf_comm_allgatherv_double(c, t->d_pdisp, m->counts.data(), m->displs.data()));
        copy_pdisp_kernel<<<grid1d(Na - 1), KMCF_BLOCK, 0, st>>>(Na - 1, t->d_atom_site, t->d_acls, t->d_pdisp, p->alpha_disp, d_site_power);
        KMCF_HIP(hipGetLastError());
    }
    KMCF_HIP(hipStreamSynchronize(st));
    KMCF_TRY(kmcf_p2p_check(c));
    if (imacro) *imacro = *h_im;
    if (stats) {
        float ms = 0.f;
        KMCF_HIP(hipEventElapsedTime(&ms, c->ev_a0, c->ev_a1));
        stats->ms_assembly = ms;
    }
    return KMCF_OK;
}
