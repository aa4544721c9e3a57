// CSR SpMV for gfx950 (wave64), replacing rocsparse_spmv csr_adaptive/csr_stream
// (dist_iterative/dist_spmv_gpu_packing.cpp:161-194).
//
// HBM-bound: per launch the kernel streams 12 B/nnz (f64 value + i32 column) +
// 20 B/row (row_ptr, x once, y once); x gathers are served by L2/Infinity Cache.
// No MFMA: 2 flop per 12 streamed bytes.
//
// Two kernels, chosen per matrix in kmcf_spmv_plan():
//
//  "stream" (default for short rows, e.g. K: 4..53 nnz/row): the nnz range is cut
//    into chunks of whole rows (<= 256*U nnz).  A 256-thread block streams a chunk's
//    values and columns with fully coalesced loads that do not depend on row_ptr (U
//    independent loads per lane in flight -> memory-level parallelism instead of the
//    row_ptr -> col -> x dependency chain per row), gathers x, parks the products in
//    LDS and then reduces them per row out of LDS.
//  "vec<LPR>": LPR lanes cooperate on one row (64/LPR rows per wavefront), wave-shuffle
//    reduction.  Used for the boundary-row pass (row list) and for matrices with rows
//    longer than a chunk.
//
// Both fuse the p.Ap dot product of CG (one partial per block, reduced in a fixed order
// by the consumer kernel) and map blocks to rows XCD-aware: blocks with equal
// blockIdx % 8 (same XCD, same L2) walk one contiguous eighth of the matrix, so each L2
// holds one window of x.
#include <cstdlib>

#include "kmcf_internal.hpp"

namespace {

__device__ __forceinline__ double wave_sum_width(double v, int width)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        if (off < width) v += __shfl_xor(v, off, 64);
    return v;
}

// Deterministic block sum (256 threads): shuffle inside each wavefront, then LDS.
__device__ __forceinline__ double block_sum_256(double v, double *lds4)
{
    v = wave_sum_width(v, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    double t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------ stream kernel
template <int U, int LPR2, bool DOT, bool SKIP_BOUNDARY>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_stream_kernel(
    int n_chunks, const int *__restrict__ chunk_row, const int *__restrict__ row_ptr,
    const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, const unsigned char *__restrict__ is_boundary,
    double *__restrict__ part, const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ double prod[KMCF_BLOCK * U];
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Cx = (n_chunks + 7) >> 3;  // chunks per XCD
    double dot = 0.0;
    for (int g0 = bi; g0 < Cx + nb8; g0 += nb8) {
        // check_done bit 1 (lab): co-resident blocks of a CU (assumed bi % 32 == CU) take adjacent chunks
        int g = g0;
        if ((check_done & 2) && nb8 == 256) g = (g0 & ~255) + ((bi & 31) << 3) + (bi >> 5);
        const int c = xcd * Cx + g;
        if (g >= Cx || c >= n_chunks) { if (check_done & 2) continue; else break; }   // block-uniform
        const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
        const int base = row_ptr[r0];
        const int cnt = row_ptr[r1] - base;
        // phase 1: stream values/columns, gather x, park products
        double v[U];
        int ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * KMCF_BLOCK + tid;
            const bool in = i < cnt;
            // streamed once: nontemporal loads keep the vector L1 for the x lines
            v[u] = in ? __builtin_nontemporal_load(val + base + i) : 0.0;
            ci[u] = in ? __builtin_nontemporal_load(col + base + i) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * KMCF_BLOCK + tid;
            if (i < cnt) prod[i] = v[u] * x[ci[u]];
        }
        __syncthreads();
        // phase 2: per-row sums out of LDS, LPR2 lanes per row
        constexpr int RPP = KMCF_BLOCK / LPR2;  // rows per pass
        const int lane = tid % LPR2;
        const int nrows = r1 - r0;
        const int passes = (nrows + RPP - 1) / RPP;
        for (int ps = 0; ps < passes; ++ps) {
            const int rr = r0 + ps * RPP + tid / LPR2;
            const bool valid = rr < r1;
            double s = 0.0;
            if (valid) {
                const int b = row_ptr[rr] - base, e = row_ptr[rr + 1] - base;
                for (int j = b + lane; j < e; j += LPR2) s += prod[j];
            }
            if (LPR2 > 1) s = wave_sum_width(s, LPR2);
            if (valid && lane == 0 && !(SKIP_BOUNDARY && is_boundary[rr])) {
                y[rr] = s;
                if (DOT) dot += x[rr] * s;
            }
        }
        __syncthreads();
    }
    if (DOT) {
        double t = block_sum_256(dot, lds4);
        if (tid == 0) part[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ vector kernel
template <int LPR, bool DOT, bool SKIP_BOUNDARY, bool ROW_LIST>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_vec_kernel(
    int n_rows, const int *__restrict__ row_ptr, const int *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const unsigned char *__restrict__ is_boundary, const int *__restrict__ row_list,
    double *__restrict__ part, const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    constexpr int RPB = KMCF_BLOCK / LPR;  // rows per block per step
    const int lane_in_row = threadIdx.x % LPR;
    const int row_in_block = threadIdx.x / LPR;
    const int G = (n_rows + RPB - 1) / RPB;       // row groups
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Gx = (G + 7) >> 3;                  // groups per XCD
    double dot = 0.0;
    for (int g = bi; g < Gx; g += nb8) {
        const int grp = xcd * Gx + g;
        const int row = grp * RPB + row_in_block;
        bool valid = (grp < G) && (row < n_rows);
        int r = row;
        if (ROW_LIST && valid) r = row_list[row];
        if (SKIP_BOUNDARY && valid) valid = (is_boundary[r] == 0);
        double s = 0.0;
        if (valid) {
            const int b = row_ptr[r], e = row_ptr[r + 1];
            for (int j = b + lane_in_row; j < e; j += LPR) s += val[j] * x[col[j]];
        }
        s = wave_sum_width(s, LPR);
        if (valid && lane_in_row == 0) {
            y[r] = s;
            if (DOT) dot += x[r] * s;
        }
    }
    if (DOT) {
        double t = block_sum_256(dot, lds4);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(KMCF_BLOCK) void pack_kernel(double *__restrict__ packed, const double *__restrict__ src,
                                                          const int *__restrict__ idx, int n,
                                                          const kmcf_scalars *__restrict__ S, int check_done)
{
    // _pack_gpu, dist_iterative/utils_cg.cu:4-15 (there: 32-thread blocks)
    if (check_done && S->done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) packed[i] = src[idx[i]];
}

__global__ __launch_bounds__(KMCF_BLOCK) void unpack_kernel(double *__restrict__ dst, const double *__restrict__ packed,
                                                            const int *__restrict__ idx, int n)
{
    // _unpack_gpu, utils_cg.cu:52-63
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[idx[i]] = packed[i];
}

__global__ __launch_bounds__(KMCF_BLOCK) void unpack_add_kernel(double *__restrict__ dst, const double *__restrict__ packed,
                                                                const int *__restrict__ idx, int n)
{
    // _unpack_add, utils_cg.cu:100-111 (indices are unique per call in the reference)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[idx[i]] += packed[i];
}

__global__ __launch_bounds__(KMCF_BLOCK) void hadamard_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                              double *__restrict__ out, int n)
{
    // _elementwise_vector_vector, utils_cg.cu:323-336
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

int grid_for(int64_t work_items, int per_block)
{
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 8) g = 8;
    if (g > KMCF_MAX_PARTIALS) g = KMCF_MAX_PARTIALS;
    return (int)((g + 7) / 8 * 8);
}

#define KMCF_VEC_ARGS(nrows, isb, rl, part) \
    nrows, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, isb, rl, part, m->d_S, chk

template <int LPR>
void launch_vec(kmcf_matrix *m, bool with_dot, bool skip_if_done, bool boundary_pass)
{
    hipStream_t st = m->comm->stream;
    const int chk = skip_if_done ? 1 : 0;
    if (!boundary_pass) {
        const int grid = m->spmv_grid;
        const bool skipb = (m->n_halo > 0);
        if (with_dot) {
            if (skipb)
                spmv_vec_kernel<LPR, true, true, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_loc, m->d_is_boundary, nullptr, m->d_part_a));
            else
                spmv_vec_kernel<LPR, true, false, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_loc, nullptr, nullptr, m->d_part_a));
        } else {
            if (skipb)
                spmv_vec_kernel<LPR, false, true, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_loc, m->d_is_boundary, nullptr, nullptr));
            else
                spmv_vec_kernel<LPR, false, false, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_loc, nullptr, nullptr, nullptr));
        }
    } else {
        const int grid = m->spmv_grid_b;
        // partials of the boundary pass live behind the interior ones
        if (with_dot)
            spmv_vec_kernel<LPR, true, false, true><<<grid, KMCF_BLOCK, 0, st>>>(
                KMCF_VEC_ARGS(m->n_boundary_rows, nullptr, m->d_boundary_rows, m->d_part_a + KMCF_MAX_PARTIALS));
        else
            spmv_vec_kernel<LPR, false, false, true><<<grid, KMCF_BLOCK, 0, st>>>(
                KMCF_VEC_ARGS(m->n_boundary_rows, nullptr, m->d_boundary_rows, nullptr));
    }
}

void launch_vec_any(kmcf_matrix *m, bool with_dot, bool skip_if_done, bool boundary_pass)
{
    switch (m->spmv_lpr) {
        case 4: launch_vec<4>(m, with_dot, skip_if_done, boundary_pass); break;
        case 8: launch_vec<8>(m, with_dot, skip_if_done, boundary_pass); break;
        case 32: launch_vec<32>(m, with_dot, skip_if_done, boundary_pass); break;
        case 64: launch_vec<64>(m, with_dot, skip_if_done, boundary_pass); break;
        default: launch_vec<16>(m, with_dot, skip_if_done, boundary_pass); break;
    }
}

#define KMCF_STREAM_ARGS(isb, part) \
    m->n_chunks, m->d_chunk_row, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, isb, part, m->d_S, chk

template <int U, int LPR2>
void launch_stream(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    hipStream_t st = m->comm->stream;
    const int chk = (skip_if_done ? 1 : 0) | (getenv("KMCF_SPMV_MAP") ? 2 : 0);
    const int grid = m->spmv_grid;
    const bool skipb = (m->n_halo > 0);
    if (with_dot) {
        if (skipb) spmv_stream_kernel<U, LPR2, true, true><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(m->d_is_boundary, m->d_part_a));
        else spmv_stream_kernel<U, LPR2, true, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(nullptr, m->d_part_a));
    } else {
        if (skipb) spmv_stream_kernel<U, LPR2, false, true><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(m->d_is_boundary, nullptr));
        else spmv_stream_kernel<U, LPR2, false, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(nullptr, nullptr));
    }
}

void launch_interior(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    if (m->spmv_kind == 1) {
        const int key = m->spmv_u * 100 + m->spmv_lpr2;
        switch (key) {
            case 401: launch_stream<4, 1>(m, with_dot, skip_if_done); break;
            case 404: launch_stream<4, 4>(m, with_dot, skip_if_done); break;
            case 801: launch_stream<8, 1>(m, with_dot, skip_if_done); break;
            case 808: launch_stream<8, 8>(m, with_dot, skip_if_done); break;
            case 1604: launch_stream<16, 4>(m, with_dot, skip_if_done); break;
            default: launch_stream<8, 4>(m, with_dot, skip_if_done); break;
        }
    } else {
        launch_vec_any(m, with_dot, skip_if_done, false);
    }
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

}  // namespace

int kmcf_spmv_plan(kmcf_matrix *m)
{
    // vec kernel: lanes per row from the mean row length (K rows hold 4..53 entries, mean 25.8)
    const double mean = m->n_loc > 0 ? double(m->nnz) / m->n_loc : 0.0;
    int lpr = 4;
    while (lpr < 64 && lpr * 2 <= mean) lpr *= 2;  // 25.8 -> 16
    {
        int v = env_int("KMCF_SPMV_LPR", 0);
        if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) lpr = v;
    }
    m->spmv_lpr = lpr;
    m->spmv_grid_b = m->n_boundary_rows > 0 ? grid_for(m->n_boundary_rows, KMCF_BLOCK / lpr) : 0;

    // stream kernel: chunks of whole rows with at most 256*U nnz
    int u = env_int("KMCF_SPMV_U", 8);
    if (u != 4 && u != 8 && u != 16) u = 8;
    m->spmv_u = u;
    m->spmv_lpr2 = env_int("KMCF_SPMV_LPR2", 4);
    const int cap = KMCF_BLOCK * u;
    int kind = env_int("KMCF_SPMV_KIND", 1);
    std::vector<int> chunk_row;
    if (kind == 1) {
        const std::vector<int> &rp = m->h_row_ptr;
        chunk_row.push_back(0);
        int r = 0;
        while (r < m->n_loc) {
            const int start = rp[r];
            int e = r;
            while (e < m->n_loc && rp[e + 1] - start <= cap) ++e;
            if (e == r) { kind = 0; break; }   // a single row exceeds a chunk: vector kernel
            chunk_row.push_back(e);
            r = e;
        }
        if (m->n_loc == 0) kind = 0;
    }
    m->spmv_kind = kind;
    if (kind == 1) {
        m->n_chunks = (int)chunk_row.size() - 1;
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_chunk_row), chunk_row.size() * sizeof(int)));
        KMCF_HIP(hipMemcpy(m->d_chunk_row, chunk_row.data(), chunk_row.size() * sizeof(int), hipMemcpyHostToDevice));
        m->spmv_grid = grid_for(m->n_chunks, 1);
    } else {
        m->spmv_grid = grid_for(m->n_loc, KMCF_BLOCK / lpr);
    }
    return KMCF_OK;
}

int kmcf_halo_exchange_begin(kmcf_matrix *m)
{
    kmcf_comm *c = m->comm;
    // in a loopback group every rank takes part in the (host-synchronous) exchange, neighbours or not
    const bool loopback = c->group && c->group->nranks > 1;
    if (m->number_of_neighbours <= 1 && !loopback) return KMCF_OK;
    if (m->n_send > 0) {
        const int grid = grid_for(m->n_send, KMCF_BLOCK);
        pack_kernel<<<grid, KMCF_BLOCK, 0, c->stream>>>(m->d_send_buf, m->d_p, m->d_send_idx, m->n_send, m->d_S, 0);
        KMCF_HIP(hipGetLastError());
    }
    KMCF_HIP(hipEventRecord(c->ev_packed, c->stream));
    KMCF_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
    KMCF_TRY(kmcf_comm_send_recv_halo(m));
    KMCF_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
    return KMCF_OK;
}

int kmcf_halo_exchange_end(kmcf_matrix *m)
{
    if (m->number_of_neighbours <= 1) return KMCF_OK;
    KMCF_HIP(hipStreamWaitEvent(m->comm->stream, m->comm->ev_halo, 0));
    return KMCF_OK;
}

// Distributed Ap = A p on the matrix workspace (d_p local part already filled):
// halo exchange on the comm stream overlapped with the interior rows, then the
// boundary rows (dspmv::gpu_packing_cam, dist_spmv_gpu_packing.cpp:106-228).
int kmcf_spmv_device(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    KMCF_TRY(kmcf_halo_exchange_begin(m));
    launch_interior(m, with_dot, skip_if_done);
    KMCF_HIP(hipGetLastError());
    if (m->n_halo > 0) {
        KMCF_TRY(kmcf_halo_exchange_end(m));
        if (m->n_boundary_rows > 0) {
            launch_vec_any(m, with_dot, skip_if_done, true);
            KMCF_HIP(hipGetLastError());
        }
    }
    return KMCF_OK;
}

extern "C" int kmcf_spmv(kmcf_matrix *m, const double *d_p, double *d_Ap)
{
    KMCF_CHECK(m && d_p && d_Ap, KMCF_ERR_ARG, "kmcf_spmv: null argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_spmv: host-only matrix");
    kmcf_comm *c = m->comm;
    KMCF_HIP(hipSetDevice(c->device));
    KMCF_TRY(kmcf_vec_in(m, m->d_p, d_p));
    KMCF_TRY(kmcf_spmv_device(m, false, false));
    KMCF_TRY(kmcf_vec_out(m, d_Ap, m->d_Ap));
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_spmv_bench(kmcf_matrix *m, int reps, int with_dot, float *ms_total)
{
    KMCF_CHECK(m && reps > 0 && ms_total, KMCF_ERR_ARG, "kmcf_spmv_bench: bad argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_spmv_bench: host-only matrix");
    kmcf_comm *c = m->comm;
    KMCF_HIP(hipSetDevice(c->device));
    KMCF_HIP(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; ++i) KMCF_TRY(kmcf_spmv_device(m, with_dot != 0, false));
    KMCF_HIP(hipEventRecord(c->ev_t1, c->stream));
    KMCF_HIP(hipEventSynchronize(c->ev_t1));
    KMCF_HIP(hipEventElapsedTime(ms_total, c->ev_t0, c->ev_t1));
    return KMCF_OK;
}

// Re-plan the SpMV of an existing matrix from the KMCF_SPMV_* environment (tuning aid).
extern "C" int kmcf_spmv_replan(kmcf_matrix *m)
{
    KMCF_CHECK(m && m->d_val, KMCF_ERR_ARG, "kmcf_spmv_replan: bad matrix");
    KMCF_HIP(hipSetDevice(m->comm->device));
    KMCF_HIP(hipStreamSynchronize(m->comm->stream));
    if (m->d_chunk_row) { hipFree(m->d_chunk_row); m->d_chunk_row = nullptr; }
    return kmcf_spmv_plan(m);
}

extern "C" int kmcf_pack(kmcf_comm *c, double *d_packed, const double *d_unpacked, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_pack: bad argument");
    if (n == 0) return KMCF_OK;
    pack_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_packed, d_unpacked, d_indices, n, nullptr, 0);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_unpack(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_unpack: bad argument");
    if (n == 0) return KMCF_OK;
    unpack_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_unpacked, d_packed, d_indices, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_unpack_add(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_unpack_add: bad argument");
    if (n == 0) return KMCF_OK;
    unpack_add_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_unpacked, d_packed, d_indices, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_elementwise_vector_vector(kmcf_comm *c, const double *d_a, const double *d_b, double *d_out, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_elementwise_vector_vector: bad argument");
    if (n == 0) return KMCF_OK;
    hadamard_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_a, d_b, d_out, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}
