// CSR SpMV for gfx950 (wave64), replacing rocsparse_spmv csr_adaptive/csr_stream
// (dist_iterative/dist_spmv_gpu_packing.cpp:161-194).
//
// HBM-bound: per launch the kernel streams 12 B/nnz (f64 value + i32 column) +
// 20 B/row (row_ptr, x once, y once); x gathers are served by L2/Infinity Cache.
// No MFMA: 2 flop per 12 streamed bytes.
//
// Kernel "vec<LPR>": LPR lanes cooperate on one row (64/LPR rows per wavefront),
// strided walk of the row so a wavefront touches one contiguous nnz span per
// step, wave-shuffle (DPP) reduction, and the p.Ap dot product of CG fused in
// (one partial per block, reduced deterministically by the consumer kernel).
// Block -> row mapping is XCD-aware: blocks with equal blockIdx % 8 (same XCD, hence
// same L2) walk one contiguous eighth of the rows, so each L2 holds one x window.
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
                spmv_vec_kernel<LPR, true, true, false><<<grid, KMCF_BLOCK, 0, st>>>(
                    m->n_loc, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, m->d_is_boundary, nullptr, m->d_part_a, m->d_S, chk);
            else
                spmv_vec_kernel<LPR, true, false, false><<<grid, KMCF_BLOCK, 0, st>>>(
                    m->n_loc, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, nullptr, nullptr, m->d_part_a, m->d_S, chk);
        } else {
            if (skipb)
                spmv_vec_kernel<LPR, false, true, false><<<grid, KMCF_BLOCK, 0, st>>>(
                    m->n_loc, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, m->d_is_boundary, nullptr, nullptr, m->d_S, chk);
            else
                spmv_vec_kernel<LPR, false, false, false><<<grid, KMCF_BLOCK, 0, st>>>(
                    m->n_loc, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, nullptr, nullptr, nullptr, m->d_S, chk);
        }
    } else {
        const int grid = m->spmv_grid_b;
        // partials of the boundary pass live behind the interior ones
        if (with_dot)
            spmv_vec_kernel<LPR, true, false, true><<<grid, KMCF_BLOCK, 0, st>>>(
                m->n_boundary_rows, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, nullptr, m->d_boundary_rows,
                m->d_part_a + KMCF_MAX_PARTIALS, m->d_S, chk);
        else
            spmv_vec_kernel<LPR, false, false, true><<<grid, KMCF_BLOCK, 0, st>>>(
                m->n_boundary_rows, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, nullptr, m->d_boundary_rows,
                nullptr, m->d_S, chk);
    }
}

void launch_any(kmcf_matrix *m, bool with_dot, bool skip_if_done, bool boundary_pass)
{
    switch (m->spmv_lpr) {
        case 4: launch_vec<4>(m, with_dot, skip_if_done, boundary_pass); break;
        case 8: launch_vec<8>(m, with_dot, skip_if_done, boundary_pass); break;
        case 32: launch_vec<32>(m, with_dot, skip_if_done, boundary_pass); break;
        case 64: launch_vec<64>(m, with_dot, skip_if_done, boundary_pass); break;
        default: launch_vec<16>(m, with_dot, skip_if_done, boundary_pass); break;
    }
}

}  // namespace

int kmcf_spmv_plan(kmcf_matrix *m)
{
    // lanes per row from the mean row length (K rows hold 4..53 entries, mean 25.8)
    double mean = m->n_loc > 0 ? double(m->nnz) / m->n_loc : 0.0;
    int lpr = 4;
    while (lpr < 64 && lpr * 2 <= mean) lpr *= 2;  // 25.8 -> 16
    if (const char *e = getenv("KMCF_SPMV_LPR")) {
        int v = atoi(e);
        if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) lpr = v;
    }
    m->spmv_lpr = lpr;
    m->spmv_grid = grid_for(m->n_loc, KMCF_BLOCK / lpr);
    m->spmv_grid_b = m->n_boundary_rows > 0 ? grid_for(m->n_boundary_rows, KMCF_BLOCK / lpr) : 0;
    // zero the partial slots once: grids never shrink below what a consumer reads
    return KMCF_OK;
}

// Number of pAp partials a consumer has to reduce (interior grid + boundary grid slots).
// Partials are laid out [0, KMCF_MAX_PARTIALS) interior, [KMCF_MAX_PARTIALS, 2x) boundary;
// unused slots stay zero (allocated zeroed, grids are fixed per matrix).

int kmcf_halo_exchange_begin(kmcf_matrix *m)
{
    if (m->number_of_neighbours <= 1) return KMCF_OK;
    kmcf_comm *c = m->comm;
    const int grid = grid_for(m->n_send, KMCF_BLOCK);
    pack_kernel<<<grid, KMCF_BLOCK, 0, c->stream>>>(m->d_send_buf, m->d_p, m->d_send_idx, m->n_send, m->d_S, 0);
    KMCF_HIP(hipGetLastError());
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
    launch_any(m, with_dot, skip_if_done, false);
    KMCF_HIP(hipGetLastError());
    if (m->n_halo > 0) {
        KMCF_TRY(kmcf_halo_exchange_end(m));
        if (m->n_boundary_rows > 0) {
            launch_any(m, with_dot, skip_if_done, true);
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
    KMCF_HIP(hipMemcpyAsync(m->d_p, d_p, (size_t)m->n_loc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    KMCF_TRY(kmcf_spmv_device(m, false, false));
    KMCF_HIP(hipMemcpyAsync(d_Ap, m->d_Ap, (size_t)m->n_loc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
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
