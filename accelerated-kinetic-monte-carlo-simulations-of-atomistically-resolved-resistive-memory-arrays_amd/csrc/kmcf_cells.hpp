// Uniform cell list + CSR pattern builder shared by the K path (kmcf_kstate.hip) and the T path
// (kmcf_tstate.hip).  Replaces the reference's O(n_rows * n_cols) distance scans
// (src/iterative_solvers_gpu.cu:96-157, src/initialize_sparsity_T.cu:10-209; "20 min" at 40 nm, README.md:13)
// with 27-cell queries; the output is what those scans produce (columns ascending, diagonal included).
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <cstring>
#include <algorithm>
#include <cmath>
#include <vector>

#include "kmcf_internal.hpp"

namespace {

// src/gpu_solvers.h:274-319
__device__ __forceinline__ double site_dist_dev(double x1, double y1, double z1, double x2, double y2, double z2,
                                                double ly, double lz, int pbc)
{
    if (pbc == 1) {
        double dist_x = x1 - x2;
        double fy = (y1 - y2) / ly;
        fy -= round(fy);
        double fz = (z1 - z2) / lz;
        fz -= round(fz);
        double dy = fy * ly, dz = fz * lz;
        return sqrt(dist_x * dist_x + dy * dy + dz * dz);
    }
    double dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

struct cell_grid {
    double x0, y0, z0, inv_x, inv_y, inv_z;
    int ncx, ncy, ncz;
    int wrap_y, wrap_z;  // periodic wrap of neighbour cells (pbc)
};

__device__ __forceinline__ int cell_coord(double v, double v0, double inv, int nc)
{
    int c = (int)floor((v - v0) * inv);
    return c < 0 ? 0 : (c >= nc ? nc - 1 : c);
}

// Visit every site j in the 27 cells around site i with col_lo <= j < col_hi and
// dist(i,j) < cutoff; F(j) is called in cell order (NOT ascending j).
template <typename F>
__device__ __forceinline__ void for_each_neighbour(const cell_grid &g, const int *__restrict__ cell_start,
                                                   const int *__restrict__ cell_items,
                                                   const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, int i, double cutoff,
                                                   double ly, double lz, int pbc, int col_lo, int col_hi, F f)
{
    const double xi = x[i], yi = y[i], zi = z[i];
    const int cx = cell_coord(xi, g.x0, g.inv_x, g.ncx);
    const int cy = cell_coord(yi, g.y0, g.inv_y, g.ncy);
    const int cz = cell_coord(zi, g.z0, g.inv_z, g.ncz);
    for (int ax = cx - 1; ax <= cx + 1; ++ax) {
        if (ax < 0 || ax >= g.ncx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            int ay = cy + dy;
            if (g.wrap_y) ay = (ay + g.ncy) % g.ncy;
            else if (ay < 0 || ay >= g.ncy) continue;
            for (int dz = -1; dz <= 1; ++dz) {
                int az = cz + dz;
                if (g.wrap_z) az = (az + g.ncz) % g.ncz;
                else if (az < 0 || az >= g.ncz) continue;
                const int cidx = (ax * g.ncy + ay) * g.ncz + az;
                for (int t = cell_start[cidx]; t < cell_start[cidx + 1]; ++t) {
                    const int j = cell_items[t];
                    if (j < col_lo || j >= col_hi) continue;
                    double d = site_dist_dev(xi, yi, zi, x[j], y[j], z[j], ly, lz, pbc);
                    if (d < cutoff) f(j);
                }
            }
        }
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void pattern_count_kernel(
    cell_grid g, const int *__restrict__ cell_start, const int *__restrict__ cell_items,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
    double cutoff, double ly, double lz, int pbc, int row_site0, int n_rows, int col_lo, int col_hi,
    int *__restrict__ nnz_per_row)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        int cnt = 0;
        for_each_neighbour(g, cell_start, cell_items, x, y, z, row_site0 + r, cutoff, ly, lz, pbc, col_lo, col_hi,
                           [&](int) { ++cnt; });
        nnz_per_row[r] = cnt;
    }
}

// Fill columns (block-local: j - col_lo) and sort each row ascending, as the reference's
// col = 0..size_j-1 scan produces them (src/iterative_solvers_gpu.cu:143-155).
__global__ __launch_bounds__(KMCF_BLOCK) void pattern_fill_kernel(
    cell_grid g, const int *__restrict__ cell_start, const int *__restrict__ cell_items,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
    double cutoff, double ly, double lz, int pbc, int row_site0, int n_rows, int col_lo, int col_hi,
    const int *__restrict__ row_ptr, int *__restrict__ col)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        const int b = row_ptr[r];
        int n = 0;
        for_each_neighbour(g, cell_start, cell_items, x, y, z, row_site0 + r, cutoff, ly, lz, pbc, col_lo, col_hi,
                           [&](int j) {
                               // insertion into the sorted prefix col[b .. b+n)
                               int v = j - col_lo, k = n;
                               while (k > 0 && col[b + k - 1] > v) { col[b + k] = col[b + k - 1]; --k; }
                               col[b + k] = v;
                               ++n;
                           });
    }
}

// Brute-force variants = the reference loops verbatim in structure; used when the
// periodic cell grid would have fewer than 3 cells along y or z.
__global__ __launch_bounds__(KMCF_BLOCK) void pattern_brute_kernel(
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
    double cutoff, double ly, double lz, int pbc, int row_site0, int n_rows, int col_lo, int col_hi,
    const int *__restrict__ row_ptr, int *__restrict__ nnz_per_row, int *__restrict__ col)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        const int i = row_site0 + r;
        int n = 0;
        for (int j = col_lo; j < col_hi; ++j) {
            double d = site_dist_dev(x[i], y[i], z[i], x[j], y[j], z[j], ly, lz, pbc);
            if (d < cutoff) {
                if (col) col[row_ptr[r] + n] = j - col_lo;
                ++n;
            }
        }
        if (nnz_per_row) nnz_per_row[r] = n;
    }
}

int grid1d(int64_t n, int cap = 2048)
{
    int64_t g = (n + KMCF_BLOCK - 1) / KMCF_BLOCK;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------------------------------------------------------- host-side cell list
struct host_cells {
    cell_grid g;
    int *d_cell_start = nullptr;
    int *d_cell_items = nullptr;
    bool usable = true;   // false: fall back to brute force (tiny periodic grids)
    unsigned long long bits_sum = 0;   // checksum of the coordinates as build_cells' synchronous host copy saw them (coords_seen_by_kernels)
    void release()
    {
        if (d_cell_start) hipFree(d_cell_start);
        if (d_cell_items) hipFree(d_cell_items);
        d_cell_start = d_cell_items = nullptr;
    }
};

int build_cells(const double *d_x, const double *d_y, const double *d_z, int N, const double *lattice, int pbc,
                double edge, host_cells *hc)
{
    std::vector<double> x(N), y(N), z(N);
    KMCF_HIP(hipMemcpy(x.data(), d_x, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(y.data(), d_y, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    KMCF_HIP(hipMemcpy(z.data(), d_z, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int s = 0; s < N; ++s) {
        lo[0] = std::min(lo[0], x[s]); hi[0] = std::max(hi[0], x[s]);
        lo[1] = std::min(lo[1], y[s]); hi[1] = std::max(hi[1], y[s]);
        lo[2] = std::min(lo[2], z[s]); hi[2] = std::max(hi[2], z[s]);
    }
    if (N == 0) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 0; }
    {
        unsigned long long acc = 0, w;
        for (int s = 0; s < N; ++s) {
            memcpy(&w, &x[s], 8); acc += w;
            memcpy(&w, &y[s], 8); acc += 3 * w;
            memcpy(&w, &z[s], 8); acc += 5 * w;
        }
        hc->bits_sum = acc;
    }
    cell_grid &g = hc->g;
    g.wrap_y = g.wrap_z = 0;
    g.x0 = lo[0]; g.inv_x = 1.0 / edge; g.ncx = (int)std::floor((hi[0] - lo[0]) / edge) + 1;
    if (pbc == 1) {
        // periodic in y and z (src/gpu_solvers.h:290-309): cells tile [y0, y0 + L) exactly
        int ny = (int)std::floor(lattice[1] / edge), nz = (int)std::floor(lattice[2] / edge);
        if (ny < 3 || nz < 3 || hi[1] - lo[1] >= lattice[1] || hi[2] - lo[2] >= lattice[2]) { hc->usable = false; return KMCF_OK; }
        g.y0 = lo[1]; g.ncy = ny; g.inv_y = ny / lattice[1]; g.wrap_y = 1;
        g.z0 = lo[2]; g.ncz = nz; g.inv_z = nz / lattice[2]; g.wrap_z = 1;
    } else {
        g.y0 = lo[1]; g.inv_y = 1.0 / edge; g.ncy = (int)std::floor((hi[1] - lo[1]) / edge) + 1;
        g.z0 = lo[2]; g.inv_z = 1.0 / edge; g.ncz = (int)std::floor((hi[2] - lo[2]) / edge) + 1;
    }
    const int64_t ncell = (int64_t)g.ncx * g.ncy * g.ncz;
    KMCF_CHECK(ncell < (int64_t)1 << 30, KMCF_ERR_ARG, "cell grid too large (%lld cells)", (long long)ncell);
    auto coord = [](double v, double v0, double inv, int nc) {
        int c = (int)std::floor((v - v0) * inv);
        return c < 0 ? 0 : (c >= nc ? nc - 1 : c);
    };
    std::vector<int> start((size_t)ncell + 1, 0), cid((size_t)N), items((size_t)std::max(N, 1));
    for (int s = 0; s < N; ++s) {
        int cx = coord(x[s], g.x0, g.inv_x, g.ncx), cy = coord(y[s], g.y0, g.inv_y, g.ncy), cz = coord(z[s], g.z0, g.inv_z, g.ncz);
        cid[s] = (cx * g.ncy + cy) * g.ncz + cz;
        start[cid[s] + 1]++;
    }
    for (int64_t cidx = 0; cidx < ncell; ++cidx) start[cidx + 1] += start[cidx];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int s = 0; s < N; ++s) items[fill[cid[s]]++] = s;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&hc->d_cell_start), start.size() * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&hc->d_cell_items), items.size() * sizeof(int)));
    KMCF_HIP(hipMemcpy(hc->d_cell_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(hc->d_cell_items, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice));
    return KMCF_OK;
}

// The set-up kernels must see the coordinates the host copy above saw.  A kernel on stream `st` adds up the same
// checksum (integer adds of the raw bits: order-free); a mismatch means the kernels would build their pattern from
// arrays whose upload has not landed for them -- seen once as a silent wrong pattern (y and z still zero: DESIGN 11, "the
// flake of the in-process groups"), now a loud KMCF_ERR_STATE.
__global__ __launch_bounds__(KMCF_BLOCK) void coords_sum_kernel(int N, const double *__restrict__ x, const double *__restrict__ y,
                                                                const double *__restrict__ z, unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < N; s += gridDim.x * blockDim.x)
        acc += (unsigned long long)__double_as_longlong(x[s]) + 3ull * (unsigned long long)__double_as_longlong(y[s]) +
               5ull * (unsigned long long)__double_as_longlong(z[s]);
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int coords_seen_by_kernels(const host_cells &hc, const double *d_x, const double *d_y, const double *d_z, int N, hipStream_t st, const char *who)
{
    unsigned long long *d_sum = nullptr, h_sum = 0;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_sum), sizeof(unsigned long long)));
    KMCF_HIP(hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), st));
    coords_sum_kernel<<<grid1d(N, 1024), KMCF_BLOCK, 0, st>>>(N, d_x, d_y, d_z, d_sum);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipMemcpyAsync(&h_sum, d_sum, sizeof(h_sum), hipMemcpyDeviceToHost, st));
    KMCF_HIP(hipStreamSynchronize(st));
    hipFree(d_sum);
    KMCF_CHECK(h_sum == hc.bits_sum, KMCF_ERR_STATE,
               "%s: the set-up kernels do not see the site coordinates the host copy sees (checksum %llx against %llx): an upload of "
               "the caller's arrays has not completed for this stream -- upload from pinned memory, or synchronise the upload's stream, "
               "before the call (INTEGRATION.md, \"Arrays handed to the library\")", who, h_sum, hc.bits_sum);
    return KMCF_OK;
}

// CSR pattern of rows (sites row_site0 .. +n_rows) against columns (sites col_lo .. col_hi).
int build_pattern(const host_cells &hc, const double *d_x, const double *d_y, const double *d_z,
                  const double *lattice, int pbc, double cutoff, int row_site0, int n_rows, int col_lo, int col_hi,
                  std::vector<int> *row_ptr, std::vector<int> *col, hipStream_t st)
{
    row_ptr->assign((size_t)n_rows + 1, 0);
    col->clear();
    if (n_rows == 0) return KMCF_OK;
    int *d_cnt = nullptr, *d_rp = nullptr, *d_col = nullptr;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_cnt), (size_t)n_rows * sizeof(int)));
    const int grid = grid1d(n_rows, 1 << 20);
    if (hc.usable)
        pattern_count_kernel<<<grid, KMCF_BLOCK, 0, st>>>(hc.g, hc.d_cell_start, hc.d_cell_items, d_x, d_y, d_z, cutoff,
                                                          lattice[1], lattice[2], pbc, row_site0, n_rows, col_lo, col_hi, d_cnt);
    else
        pattern_brute_kernel<<<grid, KMCF_BLOCK, 0, st>>>(d_x, d_y, d_z, cutoff, lattice[1], lattice[2], pbc, row_site0,
                                                          n_rows, col_lo, col_hi, nullptr, d_cnt, nullptr);
    KMCF_HIP(hipGetLastError());
    std::vector<int> cnt((size_t)n_rows);
    KMCF_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n_rows * sizeof(int), hipMemcpyDeviceToHost, st));
    KMCF_HIP(hipStreamSynchronize(st));
    int64_t nnz = 0;
    for (int r = 0; r < n_rows; ++r) { nnz += cnt[r]; (*row_ptr)[r + 1] = (int)nnz; }
    KMCF_CHECK(nnz < (int64_t)INT32_MAX, KMCF_ERR_ARG, "pattern has %lld nnz: exceeds int32 indexing", (long long)nnz);
    col->resize((size_t)nnz);
    if (nnz > 0) {
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_rp), ((size_t)n_rows + 1) * sizeof(int)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_col), (size_t)nnz * sizeof(int)));
        KMCF_HIP(hipMemcpyAsync(d_rp, row_ptr->data(), ((size_t)n_rows + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (hc.usable)
            pattern_fill_kernel<<<grid, KMCF_BLOCK, 0, st>>>(hc.g, hc.d_cell_start, hc.d_cell_items, d_x, d_y, d_z, cutoff,
                                                             lattice[1], lattice[2], pbc, row_site0, n_rows, col_lo, col_hi, d_rp, d_col);
        else
            pattern_brute_kernel<<<grid, KMCF_BLOCK, 0, st>>>(d_x, d_y, d_z, cutoff, lattice[1], lattice[2], pbc, row_site0,
                                                              n_rows, col_lo, col_hi, d_rp, nullptr, d_col);
        KMCF_HIP(hipGetLastError());
        KMCF_HIP(hipMemcpyAsync(col->data(), d_col, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, st));
        KMCF_HIP(hipStreamSynchronize(st));
        hipFree(d_rp);
        hipFree(d_col);
    }
    hipFree(d_cnt);
    return KMCF_OK;
}

template <typename T>
int upload(T **d, const std::vector<T> &h)
{
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(d), std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) KMCF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return KMCF_OK;
}

}  // namespace
