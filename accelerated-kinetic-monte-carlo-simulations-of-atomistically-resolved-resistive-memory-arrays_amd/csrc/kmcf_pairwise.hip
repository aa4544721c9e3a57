// Short-range pairwise ("gridless") Poisson term: potential of the charged sites within a cutoff
// (SURVEY.md 8f-1).  Replaces compute_cutoff_list (src/neighbor_lists_gpu.cu:293-372) and
// poisson_gridless_gpu / calculate_pairwise_interaction_indexed (src/potential_solver_gpu.cu:1525-1564,
// 1620-1655).
//
// The reference stores, per site, the indices of ALL possibly-charged sites within 20 A (N x N_cutoff
// ints: ~0.6 GB at 5 nm) and walks the whole list every step, skipping the uncharged ones.  Only a few
// percent of those sites carry a charge, so here the structure is inverted:
//   init:      sites sorted by 20 A cell (positions never change)            -> kmcf_pairwise
//   each step: flags of the charged sites in cell order -> exclusive scan -> compacted charged list per
//              cell (deterministic, all on the device), then 16 lanes per site sum the charged sites of
//              the 27 surrounding cells.
// Result per site = sum over charged j != i with dist < cutoff of  q_j erfc(r/(sigma sqrt 2)) k q / r
// (v_solve_gpu, src/gpu_solvers.h:321-329), the same set the reference's list yields as long as charged
// sites are V / Od (update_charge only ever charges those, potential_solver_gpu.cu:27-60).
#include <algorithm>
#include <cmath>
#include <vector>

#include "kmcf_internal.hpp"

struct kmcf_pairwise {
    kmcf_comm *comm = nullptr;
    int N = 0;
    double cutoff = 20.0;
    double x0 = 0, y0 = 0, z0 = 0, inv = 0;
    int ncx = 1, ncy = 1, ncz = 1;
    int ncell = 1;
    int *d_cell_order = nullptr;   // sites sorted by cell (ascending site id inside a cell), N
    int *d_cell_start = nullptr;   // ncell + 1 offsets into d_cell_order
    int *d_flag_pos = nullptr;     // N + 1: exclusive scan of the charged flags (cell order)
    int *d_block_sum = nullptr;    // scan scratch
    int *d_clist = nullptr;        // compacted charged sites (cell order), N
    int n_blocks = 0;
};

namespace {

constexpr int SCAN_ITEMS = 8;                         // per thread
constexpr int SCAN_TILE = KMCF_BLOCK * SCAN_ITEMS;    // 2048 flags per block

__device__ __forceinline__ int block_excl_scan(int v, int *lds, int *total)
{
    // exclusive scan of one int per thread over 256 threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(s, off, 64);
        if (lane >= off) s += t;
    }
    if (lane == 63) lds[w] = s;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += lds[i];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + s - v;
}

// phase 1: per-tile counts of charged sites (cell order)
__global__ __launch_bounds__(KMCF_BLOCK) void flag_count_kernel(int N, const int *__restrict__ cell_order,
                                                                const int *__restrict__ charge, int *__restrict__ block_sum)
{
    __shared__ int lds[4];
    const int t0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int c = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int t = t0 + i;
        if (t < N) c += (charge[cell_order[t]] != 0);
    }
    int total;
    block_excl_scan(c, lds, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

// phase 2: exclusive scan of the tile counts (one block; tiles <= 256 * 64)
__global__ __launch_bounds__(KMCF_BLOCK) void block_sum_scan_kernel(int nb, int *__restrict__ block_sum)
{
    __shared__ int lds[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += KMCF_BLOCK) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? block_sum[b] : 0;
        int total;
        const int ex = block_excl_scan(v, lds, &total);
        if (b < nb) block_sum[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sum[nb] = carry;   // total number of charged sites
}

// phase 3: positions + compaction
__global__ __launch_bounds__(KMCF_BLOCK) void flag_scatter_kernel(int N, const int *__restrict__ cell_order,
                                                                  const int *__restrict__ charge,
                                                                  const int *__restrict__ block_sum,
                                                                  int *__restrict__ flag_pos, int *__restrict__ clist)
{
    __shared__ int lds[4];
    const int t0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS], site[SCAN_ITEMS], c = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int t = t0 + i;
        site[i] = t < N ? cell_order[t] : -1;
        f[i] = (t < N) ? (charge[site[i]] != 0) : 0;
        c += f[i];
    }
    int pos = block_sum[blockIdx.x] + block_excl_scan(c, lds, nullptr);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int t = t0 + i;
        if (t < N) {
            flag_pos[t] = pos;
            if (f[i]) clist[pos] = site[i];
            pos += f[i];
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == KMCF_BLOCK - 1) flag_pos[N] = block_sum[gridDim.x];
}

struct pw_grid {
    double x0, y0, z0, inv;
    int ncx, ncy, ncz;
};

__device__ __forceinline__ int pw_coord(double v, double v0, double inv, int nc)
{
    int c = (int)floor((v - v0) * inv);
    return c < 0 ? 0 : (c >= nc ? nc - 1 : c);
}

// calculate_pairwise_interaction_indexed (potential_solver_gpu.cu:1525-1564): potential[i] = sum, written not added
__global__ __launch_bounds__(KMCF_BLOCK) void pairwise_kernel(
    pw_grid g, const int *__restrict__ cell_start, const int *__restrict__ flag_pos, const int *__restrict__ clist,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
    const int *__restrict__ charge, double sigma, double k, double cutoff, int count, int displ,
    double *__restrict__ potential)
{
    constexpr int LPS = 16, SPB = KMCF_BLOCK / LPS;
    const double q = 1.60217663e-19;            // gpu_solvers.h:323
    const int lane = threadIdx.x % LPS;
    const int groups = (count + SPB - 1) / SPB;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int idx = grp * SPB + threadIdx.x / LPS;
        const bool valid = idx < count;
        const int i = displ + idx;
        double acc = 0.0;
        if (valid) {
            const double xi = x[i], yi = y[i], zi = z[i];
            const int cx = pw_coord(xi, g.x0, g.inv, g.ncx), cy = pw_coord(yi, g.y0, g.inv, g.ncy),
                      cz = pw_coord(zi, g.z0, g.inv, g.ncz);
            for (int ax = max(cx - 1, 0); ax <= min(cx + 1, g.ncx - 1); ++ax)
                for (int ay = max(cy - 1, 0); ay <= min(cy + 1, g.ncy - 1); ++ay) {
                    // cells (ax, ay, az-1..az+1) are contiguous in the cell order: one run
                    const int c_lo = (ax * g.ncy + ay) * g.ncz + max(cz - 1, 0);
                    const int c_hi = (ax * g.ncy + ay) * g.ncz + min(cz + 1, g.ncz - 1);
                    const int b = flag_pos[cell_start[c_lo]], e = flag_pos[cell_start[c_hi + 1]];
                    for (int t = b + lane; t < e; t += LPS) {
                        const int j = clist[t];
                        if (j == i) continue;
                        const double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
                        const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                        if (dist < cutoff) {
                            const double r = 1e-10 * dist;                                    // :1554
                            acc += (double)charge[j] * erfc(r / (sigma * sqrt(2.0))) * k * q / r;  // gpu_solvers.h:325
                        }
                    }
                }
        }
#pragma unroll
        for (int off = LPS / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (valid && lane == 0) potential[i] = acc;
    }
}

}  // namespace

extern "C" int kmcf_compute_cutoff_list(kmcf_comm *c, const double *d_x, const double *d_y, const double *d_z, int N,
                                        double cutoff_radius, kmcf_pairwise **out)
{
    KMCF_CHECK(c && d_x && d_y && d_z && out && N > 0 && cutoff_radius > 0, KMCF_ERR_ARG, "kmcf_compute_cutoff_list: bad argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_compute_cutoff_list: host-only communicator");
    KMCF_TRY(kmcf_enter(c));
    std::vector<double> x(N), y(N), z(N);
    KMCF_HIP(hipMemcpy(x.data(), d_x, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(y.data(), d_y, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(z.data(), d_z, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    kmcf_pairwise *p = new kmcf_pairwise();
    p->comm = c; p->N = N; p->cutoff = cutoff_radius;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int s = 0; s < N; ++s) {
        lo[0] = std::min(lo[0], x[s]); hi[0] = std::max(hi[0], x[s]);
        lo[1] = std::min(lo[1], y[s]); hi[1] = std::max(hi[1], y[s]);
        lo[2] = std::min(lo[2], z[s]); hi[2] = std::max(hi[2], z[s]);
    }
    p->x0 = lo[0]; p->y0 = lo[1]; p->z0 = lo[2]; p->inv = 1.0 / cutoff_radius;
    p->ncx = (int)std::floor((hi[0] - lo[0]) * p->inv) + 1;
    p->ncy = (int)std::floor((hi[1] - lo[1]) * p->inv) + 1;
    p->ncz = (int)std::floor((hi[2] - lo[2]) * p->inv) + 1;
    p->ncell = p->ncx * p->ncy * p->ncz;
    auto coord = [&](double v, double v0, int nc) {
        int cc = (int)std::floor((v - v0) * p->inv);
        return cc < 0 ? 0 : (cc >= nc ? nc - 1 : cc);
    };
    std::vector<int> start((size_t)p->ncell + 1, 0), cid((size_t)N), order((size_t)N);
    for (int s = 0; s < N; ++s) {
        cid[s] = (coord(x[s], p->x0, p->ncx) * p->ncy + coord(y[s], p->y0, p->ncy)) * p->ncz + coord(z[s], p->z0, p->ncz);
        start[cid[s] + 1]++;
    }
    for (int cc = 0; cc < p->ncell; ++cc) start[cc + 1] += start[cc];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int s = 0; s < N; ++s) order[fill[cid[s]]++] = s;
    p->n_blocks = (N + SCAN_TILE - 1) / SCAN_TILE;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_cell_order), (size_t)N * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_cell_start), ((size_t)p->ncell + 1) * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_flag_pos), ((size_t)N + 1) * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_block_sum), ((size_t)p->n_blocks + 1) * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_clist), (size_t)N * sizeof(int)));
    KMCF_HIP(hipMemcpy(p->d_cell_order, order.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(p->d_cell_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
    *out = p;
    return KMCF_OK;
}

extern "C" int kmcf_pairwise_destroy(kmcf_pairwise *p)
{
    if (!p) return KMCF_OK;
    hipSetDevice(p->comm->device);
    hipStreamSynchronize(p->comm->stream);
    void *ptrs[] = {p->d_cell_order, p->d_cell_start, p->d_flag_pos, p->d_block_sum, p->d_clist};
    for (void *q : ptrs)
        if (q) hipFree(q);
    delete p;
    return KMCF_OK;
}

extern "C" int kmcf_poisson_gridless(kmcf_pairwise *p, const double *d_x, const double *d_y, const double *d_z,
                                     const int *d_site_charge, double sigma, double k, int count, int displ,
                                     double *d_site_potential_charge)
{
    KMCF_CHECK(p && d_x && d_y && d_z && d_site_charge && d_site_potential_charge, KMCF_ERR_ARG, "kmcf_poisson_gridless: null argument");
    KMCF_CHECK(count >= 0 && displ >= 0 && displ + count <= p->N, KMCF_ERR_ARG, "kmcf_poisson_gridless: rows [%d,%d) outside N=%d",
               displ, displ + count, p->N);
    kmcf_comm *c = p->comm;
    KMCF_TRY(kmcf_enter(c));
    hipStream_t st = c->stream;
    flag_count_kernel<<<p->n_blocks, KMCF_BLOCK, 0, st>>>(p->N, p->d_cell_order, d_site_charge, p->d_block_sum);
    block_sum_scan_kernel<<<1, KMCF_BLOCK, 0, st>>>(p->n_blocks, p->d_block_sum);
    flag_scatter_kernel<<<p->n_blocks, KMCF_BLOCK, 0, st>>>(p->N, p->d_cell_order, d_site_charge, p->d_block_sum,
                                                          p->d_flag_pos, p->d_clist);
    KMCF_HIP(hipGetLastError());
    if (count > 0) {
        pw_grid g{p->x0, p->y0, p->z0, p->inv, p->ncx, p->ncy, p->ncz};
        int64_t grid = ((int64_t)count * 16 + KMCF_BLOCK - 1) / KMCF_BLOCK;
        if (grid > 8192) grid = 8192;
        pairwise_kernel<<<(int)grid, KMCF_BLOCK, 0, st>>>(g, p->d_cell_start, p->d_flag_pos, p->d_clist, d_x, d_y, d_z,
                                                         d_site_charge, sigma, k, p->cutoff, count, displ,
                                                         d_site_potential_charge);
        KMCF_HIP(hipGetLastError());
    }
    KMCF_HIP(hipStreamSynchronize(st));
    return KMCF_OK;
}
