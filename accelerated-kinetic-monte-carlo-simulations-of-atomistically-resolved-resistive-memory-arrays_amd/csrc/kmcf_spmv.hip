// CSR SpMV for gfx950 (wave64), replacing rocsparse_spmv csr_adaptive/csr_stream
// (dist_iterative/dist_spmv_gpu_packing.cpp:161-194).
//
// HBM-bound work: a CSR launch streams 12 B/nnz (f64 value + i32 column) + 20 B/row (row_ptr, x once,
// y once); x gathers are served by L2/Infinity Cache.  No MFMA: 2 flop per 12 streamed bytes.
//
// Four kernels, chosen per matrix in kmcf_spmv_plan() (DESIGN.md 3.1 has the measurements behind each):
//
//  "window" / "window, coded" (default where columns are local, e.g. K in its internal brick order):
//    tiles of whole rows whose distinct columns are staged once in LDS; entries carry a 16-bit window
//    slot instead of a column, and -- where the off-diagonal values come from a small dictionary, as K's
//    two conductances do -- a 6-bit value code instead of an f64 value: 2 B/nnz.  See the kernels below.
//  "stream" (short rows with scattered columns): the nnz range is cut
//    into chunks of whole rows (<= 256*U nnz).  A 256-thread block streams a chunk's
//    values and columns with fully coalesced loads that do not depend on row_ptr (U
//    independent loads per lane in flight -> memory-level parallelism instead of the
//    row_ptr -> col -> x dependency chain per row), gathers x, parks the products in
//    LDS and then reduces them per row out of LDS.
//  "vec<LPR>": LPR lanes cooperate on one row (64/LPR rows per wavefront), wave-shuffle
//    reduction.  Used for the boundary-row pass (row list) and for matrices with rows
//    longer than a chunk.
//
// All fuse the p.Ap dot product of CG (one partial per block, reduced in a fixed order
// by the consumer kernel) and map blocks to rows XCD-aware: blocks with equal
// blockIdx % 8 (same XCD, same L2) walk one contiguous eighth of the matrix, so each L2
// holds one window of x.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <climits>

#include "kmcf_p2p_dev.hpp"

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
    v = kmcf_wave_sum64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    double t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

// Matrix data a kernel reads once per launch.  Marked nontemporal (streaming: do not keep it in the caches) only
// where the matrix is larger than the caches anyway: measured on the 40 nm K matrix (round 2), 534 MB of CSR
// stream 141 us nontemporal / 148 us plain, 481 MB of window format 129 / 133 -- but the 2 B/nnz formats, which
// FIT the 256 MiB Infinity Cache together with the CG's vectors, 57 / 42 us (coded window kernel) and
// 33.7 / 25.0 us (row-per-lane kernel): the hint evicts what the next iteration would have found in the cache.
template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// ------------------------------------------------------------------ stream kernel
template <int U, int LPR2, bool DOT, bool SKIP_BOUNDARY, bool NT>
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
    for (int g = bi; g < Cx; g += nb8) {
        const int c = xcd * Cx + g;
        if (c >= n_chunks) break;                      // block-uniform
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
            v[u] = in ? stream_load<NT>(val + base + i) : 0.0;
            ci[u] = in ? stream_load<NT>(col + base + i) : 0;
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

// ------------------------------------------------------------------ window kernel
// Tiles of whole rows (<= 256*U nnz, <= 8*U rows) whose distinct columns (<= 256*WQ of them, found at plan
// time) are staged once in LDS: the tile's window map is read coalesced, x is fetched run by run (neighbouring
// lanes read neighbouring addresses, so the texture path merges them), and the per-entry gather happens in
// LDS through 16-bit window slots instead of one L1 tag lookup per lane (PMC: 45.6 M TCP accesses per launch
// for the stream kernel, most of them single-lane gathers).  Values keep their CSR order and stay f64.
template <int U, int WQ, int LPR2, bool DOT, bool SKIP_BOUNDARY, bool NT>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_window_kernel(
    int n_tiles, const int2 *__restrict__ tile, const int *__restrict__ row_ptr, const int *__restrict__ wcol,
    const unsigned short *__restrict__ idx16, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, const unsigned char *__restrict__ is_boundary, double *__restrict__ part,
    const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ double xw[KMCF_BLOCK * WQ];
    __shared__ double prod[KMCF_BLOCK * U];
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Cx = (n_tiles + 7) >> 3;  // tiles per XCD
    double dot = 0.0;
    for (int g = bi; g < Cx; g += nb8) {
        const int c = xcd * Cx + g;
        if (c >= n_tiles) break;                       // block-uniform
        const int2 t0 = tile[c], t1 = tile[c + 1];
        const int r0 = t0.x, r1 = t1.x, w0 = t0.y, W = t1.y - w0;
        const int base = row_ptr[r0];
        const int cnt = row_ptr[r1] - base;
        // every global load of the tile is issued before the first wait: window map (its dependent x
        // fetch is the longest chain), then the value / slot streams
        int wc[WQ];
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int w = q * KMCF_BLOCK + tid;
            wc[q] = w < W ? stream_load<NT>(wcol + w0 + w) : -1;
        }
        double v[U];
        unsigned short ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * KMCF_BLOCK + tid;
            const bool in = i < cnt;
            v[u] = in ? stream_load<NT>(val + base + i) : 0.0;
            ci[u] = in ? stream_load<NT>(idx16 + base + i) : (unsigned short)0;
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            if (wc[q] >= 0) xw[q * KMCF_BLOCK + tid] = x[wc[q]];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * KMCF_BLOCK + tid;
            if (i < cnt) prod[i] = v[u] * xw[ci[u] & ((1 << KMCF_SLOT_BITS) - 1)];   // code bits may be stale
        }
        __syncthreads();
        constexpr int RPP = KMCF_BLOCK / LPR2;  // rows per pass
        const int lane = tid % LPR2;
        const int passes = (r1 - r0 + RPP - 1) / RPP;
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

// ------------------------------------------------------------------ coded window kernel
// Same tiles, for matrices whose off-diagonal values are all one of <= 62 doubles (K and the CB-edge system:
// two, -high_G and -low_G): idx16 carries the value's dictionary code above the slot bits and the kernel
// never reads val -- 2 B/nnz of matrix stream.  Only the 16-bit stream is parked in LDS (not products): the
// row lanes read slot+code, look the value up in an LDS dictionary (entry 63 = 0: the diagonal entry, which
// is skipped in the stream), fetch x from the window and accumulate in the same order as the other kernels;
// the diagonal product comes from diagv and is added after the off-diagonal sum.
//
// Software-pipelined over the block's tiles: with 2 B/nnz the kernel is no longer bound by HBM but by the
// latency of one tile's load -> gather -> barrier -> reduce chain at 8 waves per SIMD (measured: loads alone
// 36 us, reduction alone 28 us, one after the other 58 us).  So the slot stream and window map of tile t+1 are
// requested before tile t is reduced, and LDS is double-buffered, which also leaves one barrier per tile:
// buffer b is rewritten for tile t+2 only by threads that passed the barrier of tile t+1, i.e. after every
// thread finished reducing tile t.
template <int U>
struct slot_pack {
    typedef unsigned int type __attribute__((ext_vector_type(U / 2)));   // U 16-bit entries
};

// Stages of the pipeline, each one tile iteration apart so that no load waits for another load of the same
// iteration (the first version fetched tile descriptor -> row_ptr[r0] -> slot stream in a dependent chain and
// the x gather of a tile inside that tile's own iteration: ~3 exposed memory latencies per tile and block, with
// ~14 tiles per block the whole kernel time):
//   A(k): tile descriptor (first row, rows, first window slot, window size) and entry base -- address-only loads
//   B(k): window map, needs A(k)
//   C(k): x gather through the window map, slot stream, row data (row_ptr, x, diagonal) -- needs A(k), B(k)
//   D(k): LDS staging, barrier, row reduction
// iteration k runs D(k), C(k+1), B(k+2), A(k+3).  Loads are unconditional (clamped addresses): a branch per
// load would make the compiler drain every outstanding load at the join.  Past the block's last tile the stages
// reload that last tile.
template <int U, int WQ, bool DOT, bool SKIP_BOUNDARY>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_wcode_kernel(
    int n_tiles, const int4 *__restrict__ tile4, const int *__restrict__ tbase, const int *__restrict__ row_ptr,
    const int *__restrict__ wcol, const unsigned short *__restrict__ idx16, const double *__restrict__ x,
    double *__restrict__ y, const unsigned char *__restrict__ is_boundary, double *__restrict__ part,
    const kmcf_scalars *__restrict__ S, int check_done, const double *__restrict__ dict, const double *__restrict__ diagv)
{
    constexpr int LPR2 = 4, RPP = KMCF_BLOCK / LPR2, SLOT_MASK = (1 << KMCF_SLOT_BITS) - 1, UN = 4;
    typedef typename slot_pack<U>::type pack_t;
    __shared__ double xw[2][KMCF_BLOCK * WQ];
    __shared__ pack_t sidx_pk[2][KMCF_BLOCK];
    __shared__ double sdict[64];
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    const int tid = threadIdx.x;
    if (tid < 64) sdict[tid] = dict[tid];               // visible after the first tile's barrier
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Cx = (n_tiles + 7) >> 3;                  // tiles per XCD
    const int lane = tid % LPR2;
    double dot = 0.0;
    // this block's tiles: c(k) = xcd Cx + bi + k nb8 for k < nt
    const int gmax = min(Cx, n_tiles - xcd * Cx);
    const int nt = gmax > bi ? (gmax - bi + nb8 - 1) / nb8 : 0;
    if (nt > 0) {
        const int c_first = xcd * Cx + bi;
#define KMCF_TILE_OF(k) (c_first + min((k), nt - 1) * nb8)
        // prologue: A(0..2), B(0..1), C(0)
        int4 d0 = tile4[KMCF_TILE_OF(0)], d1 = tile4[KMCF_TILE_OF(1)], d2 = tile4[KMCF_TILE_OF(2)];
        int b0 = tbase[KMCF_TILE_OF(0)], b1 = tbase[KMCF_TILE_OF(1)], b2 = tbase[KMCF_TILE_OF(2)];
        int wc0[WQ], wc1[WQ];
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            wc0[q] = stream_load<false>(wcol + d0.z + min(q * KMCF_BLOCK + tid, max(d0.w - 1, 0)));
            wc1[q] = stream_load<false>(wcol + d1.z + min(q * KMCF_BLOCK + tid, max(d1.w - 1, 0)));
        }
        pack_t pk = stream_load<false>(reinterpret_cast<const pack_t *>(idx16 + (b0 & ~(U - 1))) + tid);
        int nb, ne;
        double nxrow, ndg, xr[WQ];
        {
            const int rc = d0.x + min(tid / LPR2, d0.y - 1);
            nb = row_ptr[rc]; ne = row_ptr[rc + 1]; nxrow = x[rc]; ndg = diagv[rc];
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) xr[q] = x[wc0[q]];
        int buf = 0;
        for (int k = 0; k < nt; ++k) {
            // ---- D(k), first half: stage the tile in LDS
            const int cr0 = d0.x, cr1 = d0.x + d0.y, cbase = b0 & ~(U - 1);
            int rr = cr0 + tid / LPR2;
            int b = nb - cbase, e = ne - cbase;
            double xrow = nxrow, dg = ndg;
            if (rr >= cr1) b = e = 0;
            sidx_pk[buf][tid] = pk;
#pragma unroll
            for (int q = 0; q < WQ; ++q) xw[buf][q * KMCF_BLOCK + tid] = xr[q];
            // ---- C(k+1): everything tile k+1 needs is requested now and lands while tile k is reduced
            pk = stream_load<false>(reinterpret_cast<const pack_t *>(idx16 + (b1 & ~(U - 1))) + tid);
            {
                const int rc = d1.x + min(tid / LPR2, d1.y - 1);
                nb = row_ptr[rc]; ne = row_ptr[rc + 1]; nxrow = x[rc]; ndg = diagv[rc];
            }
#pragma unroll
            for (int q = 0; q < WQ; ++q) xr[q] = x[wc1[q]];
            // ---- B(k+2), A(k+3)
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                wc0[q] = wc1[q];
                wc1[q] = stream_load<false>(wcol + d2.z + min(q * KMCF_BLOCK + tid, max(d2.w - 1, 0)));
            }
            d0 = d1; b0 = b1;
            d1 = d2; b1 = b2;
            d2 = tile4[KMCF_TILE_OF(k + 3)];
            b2 = tbase[KMCF_TILE_OF(k + 3)];
            __syncthreads();
            // ---- D(k), second half: row sums out of LDS
            const double *xwb = xw[buf];
            const unsigned short *sib = reinterpret_cast<const unsigned short *>(sidx_pk[buf]);
            // One pass of the row lanes over <= RPP rows.  Tiles planned for U <= 8 hold at most 8 U <= RPP rows:
            // a single pass whose row data came with the prefetch.  (With a second pass in the same loop the
            // compiler must assume xrow / dg may come from that pass's fresh loads and waits for ALL outstanding
            // loads -- the prefetch of the next tiles included -- before the row's last add.)
            auto row_pass = [&](int rr_, int b_, int e_, double xrow_, double dg_) {
                const bool valid = rr_ < cr1;
                double s = 0.0;
                // UN entries per step with independent LDS reads; entries past the row end read as the diagonal
                // code, whose dictionary value is 0.
                for (int j0 = b_ + lane; j0 < e_; j0 += LPR2 * UN) {
                    int cc[UN];
#pragma unroll
                    for (int q = 0; q < UN; ++q) {
                        // unconditional LDS read, then a select: a branch per entry (what the compiler makes of a
                        // guarded read) costs more issue slots than the read itself.  Reads past the row's end
                        // stay inside the block's LDS (at most 3 LPR2 entries behind a tile of <= 256 U - U
                        // entries; the second buffer and the dictionary follow) and are discarded by the select.
                        const int j = j0 + LPR2 * q;
                        const int raw = (int)sib[j];
                        cc[q] = j < e_ ? raw : (KMCF_CODE_DIAG << KMCF_SLOT_BITS);
                    }
                    double xv[UN], vv[UN];
#pragma unroll
                    for (int q = 0; q < UN; ++q) {
                        xv[q] = xwb[cc[q] & SLOT_MASK];
                        vv[q] = sdict[cc[q] >> KMCF_SLOT_BITS];
                    }
#pragma unroll
                    for (int q = 0; q < UN; ++q) s = __builtin_fma(vv[q], xv[q], s);     // one rounding per entry
                }
                s = wave_sum_width(s, LPR2);
                if (valid && lane == 0 && !(SKIP_BOUNDARY && is_boundary[rr_])) {
                    s += dg_ * xrow_;
                    y[rr_] = s;
                    if (DOT) dot += xrow_ * s;
                }
            };
            row_pass(rr, b, e, xrow, dg);
            if constexpr (8 * U > RPP) {
                const int passes = (cr1 - cr0 + RPP - 1) / RPP;
                for (int ps = 1; ps < passes; ++ps) {
                    const int r2 = cr0 + ps * RPP + tid / LPR2;
                    const int rc = min(r2, cr1 - 1);
                    int b2_ = row_ptr[rc] - cbase, e2_ = row_ptr[rc + 1] - cbase;
                    if (r2 >= cr1) b2_ = e2_ = 0;
                    row_pass(r2, b2_, e2_, x[rc], diagv[rc]);
                }
            }
            buf ^= 1;
        }
#undef KMCF_TILE_OF
    }
    if (DOT) {
        double t = block_sum_256(dot, lds4);
        if (tid == 0) part[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ coded row-per-lane kernel
// The coded window kernel above spends its time on instructions, not bytes (measured: ~213 VALU instructions per
// wave and tile, two LDS lookups and a slot decode per entry, 4 lanes per row running to the longest of a wave's
// 16 rows).  This layout removes the per-entry work instead of tuning it:
//   * the value lookup moves out of the entry loop: the x window is staged ND times, once per dictionary value,
//     already multiplied (xs[c][slot] = dict[c] * x[wcol[slot]]), and an entry IS the LDS byte offset of its
//     product: ((code << LW) | slot) << 3.  Per entry: one 16-bit extract, one ds_read_b64, one add -- the
//     same products, added in the row's column order;
//   * one row per lane, rows of a tile (<= 256 rows, one shared window) dealt to waves by length (sell_lane_order),
//     so a wave's lanes run about equally long; the wave's stream is [step][lane][4 entries], padded to its
//     longest row with the offset of a slot that holds 0.0.  A step is one coalesced 8-byte load per lane straight
//     into registers: the stream never touches LDS, the trip count is a scalar, nothing in the loop is divergent.
//     The lane order is part of the matrix's internal row order (kmcf_sell_refine_order), so lane t of a tile owns
//     row r0 + t and x, the diagonal and y are contiguous (IDENT); a matrix ordered otherwise reads its lane
//     rows from lrow;
//   * window slots 0 .. 255 are the tile's own rows (staged from the row's x load), the others the columns outside
//     the tile, gathered through the window map wcol;
//   * the register a step frees receives the same step of the next tile at once: one register set holds the
//     stream of two tiles in flight (with a second set: 106 VGPRs, 4 blocks per CU, 41.9 us; with one: 81,
//     5 blocks, 37.6 us on the 40 nm K matrix).
// Stages per iteration k as in the kernel above: D(k) stage + reduce (+ stream of k+1), C(k+1) gathers and row
// data, B(k+2) window map (+ lane rows), A(k+3) descriptors.
typedef unsigned int sell_pair __attribute__((ext_vector_type(2)));   // 4 entries
// the stream is loaded with plain (cacheable) loads -- see stream_load above; -DKMCF_SELL_NT_LAB is the
// nontemporal variant of the measurements in tools/lab/big_lab.sh
// (NT: the matrix's own stream beyond the Infinity Cache, kmcf_matrix::sell_nt -- the hint keeps the stream from
// evicting x there: 12 x 12-cell device, 350 MB of format, 67.5 us against 71.0; never inside the cache: 31.9 against 25.2)
#define KMCF_SELL_LD(p) stream_load<NT>(p)
// (the window map and the diagonal -- read once per launch, too -- under the same hint: 77.6 us against 68.2, dropped)
#define KMCF_SELL_LD2(p) (*(p))

template <int WQ>
struct sell_regs {
    double xr[WQ];
    double xrow, xown, dg;             // x of the lane's row; x of row r0 + t (the same thing when IDENT)
    int row, nq, wn;
    bool valid;
};

template <int NQ, int LW, int ND, bool DOT, bool SKIP_BOUNDARY, bool IDENT, bool NT = false>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_sell_kernel(
    int n_tiles, const int4 *__restrict__ tile4, const int2 *__restrict__ swave, const int *__restrict__ lrow,
    const int *__restrict__ wcol, const sell_pair *__restrict__ stream, const double *__restrict__ x, double *__restrict__ y,
    const unsigned char *__restrict__ is_boundary, double *__restrict__ part, const kmcf_scalars *__restrict__ S,
    int check_done, const double *__restrict__ dict, const double *__restrict__ diagv)
{
    // window slots [0, 256): the tile's own rows (slot t = x[r0 + t], which the lane loads anyway); the rest:
    // the other columns the tile references, gathered through wcol
    constexpr int W = 1 << LW, WQ = W / KMCF_BLOCK - 1, BUF = ND * W;
    typedef sell_regs<WQ> regs_t;
    __shared__ double xs[2 * BUF];      // (exactly 32 KB for two values and 1024 slots: five blocks fill a CU's LDS)
    if (check_done && S->done) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double dv[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) dv[c] = dict[c];
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Cx = (n_tiles + 7) >> 3;
    const int gmax = min(Cx, n_tiles - xcd * Cx);
    const int nt = gmax > bi ? (gmax - bi + nb8 - 1) / nb8 : 0;
    double dot = 0.0;
    if (nt > 0) {
        const int c_first = xcd * Cx + bi;
#define KMCF_TILE_OF(k) (c_first + min((k), nt - 1) * nb8)
        // stage B: window map (and lane rows) of a tile
        auto load_b = [&](int c, const int4 &d, int (&wc)[WQ], int &lr) {
#pragma unroll
            for (int q = 0; q < WQ; ++q) wc[q] = KMCF_SELL_LD2(wcol + d.z + min(q * KMCF_BLOCK + tid, max(d.w - 1, 0)));
            if (!IDENT) lr = lrow[(size_t)c * KMCF_BLOCK + tid];
        };
        // the same with (wave-scope, relaxed) atomic loads, which stay where they are written and cost nothing
        // extra: the prologue must issue in the loop body's order (B before C) or the wait counts derived for the
        // loop head are the prologue's, and the compiler sinks ordinary loads of __restrict__ data past anything
        auto load_b_pinned = [&](int c, const int4 &d, int (&wc)[WQ], int &lr) {
#pragma unroll
            for (int q = 0; q < WQ; ++q) wc[q] = __hip_atomic_load(wcol + d.z + min(q * KMCF_BLOCK + tid, max(d.w - 1, 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (!IDENT) lr = __hip_atomic_load(lrow + (size_t)c * KMCF_BLOCK + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        };
        // stage C: x gathers and row data (issue order = the order they are waited for)
        auto load_c = [&](const int4 &d, const int2 &sw, const int (&wc)[WQ], int lr, regs_t &t) {
#pragma unroll
            for (int q = 0; q < WQ; ++q) t.xr[q] = x[wc[q]];
            if (IDENT) {
                t.valid = tid < d.y;
                t.row = d.x + min(tid, d.y - 1);
            } else {
                t.valid = lr >= 0;
                t.row = d.x + (t.valid ? lr : 0);
            }
            t.xrow = x[t.row];
            t.xown = IDENT ? t.xrow : x[d.x + min(tid, d.y - 1)];
            t.dg = KMCF_SELL_LD2(diagv + t.row);
            if (SKIP_BOUNDARY) t.valid = t.valid && is_boundary[t.row] == 0;
            t.nq = sw.y;
            t.wn = d.w;
        };
        int4 d1 = tile4[KMCF_TILE_OF(1)], d2 = tile4[KMCF_TILE_OF(2)];
        int2 s1 = swave[KMCF_TILE_OF(1) * 4 + wv], s2 = swave[KMCF_TILE_OF(2) * 4 + wv];
        int wca[WQ], wcb[WQ], lra = 0, lrb = 0;
        regs_t ta, tb;
        sell_pair pk[NQ];
        {   // prologue: window maps of tiles 0 and 1, then everything of tile 0 -- the issue order of the loop
            // body (B before C), so that the wait counts the compiler derives for the loop head are the loop's own
            const int4 d0 = tile4[c_first];
            const int2 s0 = swave[c_first * 4 + wv];
            load_b(c_first, d0, wca, lra);
            load_b_pinned(KMCF_TILE_OF(1), d1, wcb, lrb);
            __builtin_amdgcn_sched_barrier(0);
            load_c(d0, s0, wca, lra, ta);
            const sell_pair *sp = stream + s0.x + lane;
            const int last = max(s0.y - 1, 0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) pk[q] = KMCF_SELL_LD(sp + min(q, last) * 64);
        }
        // wu / lu: window map and lane rows of tile k+1 (loaded an iteration ago); wl / ll: receive tile k+2's.
        // B is issued before C so that waiting for tile k+1's map next iteration leaves this iteration's later
        // loads in flight (waits are in issue order).
        auto body = [&](int k, regs_t &cur, regs_t &nxt, double *xb, int (&wu)[WQ], int &lu, int (&wl)[WQ], int &ll) {
            // ---- D(k), first half: products of the window into LDS (slots past the window: 0.0, the padding target)
#pragma unroll
            for (int c = 0; c < ND; ++c) xb[c * W + tid] = dv[c] * cur.xown;
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int slot = q * KMCF_BLOCK + tid;
                const double v = slot < cur.wn ? cur.xr[q] : 0.0;
#pragma unroll
                for (int c = 0; c < ND; ++c) xb[c * W + KMCF_BLOCK + slot] = dv[c] * v;
            }
            // ---- B(k+2), C(k+1), A(k+3)
            load_b(KMCF_TILE_OF(k + 2), d2, wl, ll);
            __builtin_amdgcn_sched_barrier(0);
            load_c(d1, s1, wu, lu, nxt);
            const sell_pair *spn = stream + s1.x + lane;     // tile k+1's stream, requested step by step below
            const int lastn = max(s1.y - 1, 0);              // (past the wave's end: re-read, a cache hit)
            d1 = d2; s1 = s2;
            d2 = tile4[KMCF_TILE_OF(k + 3)];
            s2 = swave[KMCF_TILE_OF(k + 3) * 4 + wv];
            __syncthreads();
            // ---- D(k), second half: the lane's row
            const char *base = reinterpret_cast<const char *>(xb);
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q < cur.nq) {
                    const sell_pair e = pk[q];
                    const double a0 = *reinterpret_cast<const double *>(base + (e.x & 0xffffu));
                    const double a1 = *reinterpret_cast<const double *>(base + (e.x >> 16));
                    const double a2 = *reinterpret_cast<const double *>(base + (e.y & 0xffffu));
                    const double a3 = *reinterpret_cast<const double *>(base + (e.y >> 16));
                    s += a0; s += a1; s += a2; s += a3;
                }
                pk[q] = KMCF_SELL_LD(spn + min(q, lastn) * 64);
            }
            if (cur.valid) {
                s += cur.dg * cur.xrow;
                y[cur.row] = s;                       // (a nontemporal store here under NT: no gain, 68.5-74.6 against 67.1-70.3 us)
                if (DOT) dot += cur.xrow * s;
            }
        };
        // Tiles in pairs (the LDS buffers and map registers swap roles), an odd last tile after the loop: with a
        // conditional second half inside the loop the compiler sees a path from the first half back to the loop
        // head and waits there for loads it has just requested.
        int k = 0;
        for (; k + 1 < nt; k += 2) {
            body(k, ta, tb, xs, wcb, lrb, wca, lra);
            body(k + 1, tb, ta, xs + BUF, wca, lra, wcb, lrb);
        }
        if (k < nt) body(k, ta, tb, xs, wcb, lrb, wca, lra);
#undef KMCF_TILE_OF
    }
    if (DOT) {
        __syncthreads();                    // every wave is done with xs
        double t = block_sum_256(dot, xs);
        if (tid == 0) part[blockIdx.x] = t;
    }
}


// ------------------------------------------------------------------ row-per-lane kernel, f64 values
// The row-per-lane layout for matrices whose values are NOT dictionary-coded (general CSR input; the symmetrically
// scaled CB-edge system): same tiles, same lane order, same window (the tile's own rows + the outside columns through
// wcol, staged once in LDS -- plain x here, one copy), same 16-bit entries (LDS byte offsets; the code bits are masked
// off), and next to them the entries' f64 VALUES in the same [step][lane][4] order: 10 B per entry.  Per entry one
// ds_read_b64, one multiply, one add -- no products parked in LDS, no second pass over them, no row_ptr: the window
// kernel (10 B/nnz too) needs both (131 us at 40 nm; this one: see DESIGN 3.1).  Row sum = the off-diagonal products in
// stored order, then the diagonal product (d_diagv): the order of the coded kernel.  Latency is hidden by occupancy
// (16 KB of LDS and <= 64 VGPRs per block: 8 blocks per CU), not by a hand-built pipeline: a tile moves 5 x the bytes of
// a coded tile.
template <int LW, bool DOT, bool SKIP_BOUNDARY>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_sellv_kernel(
    int n_tiles, const int4 *__restrict__ tile4, const int2 *__restrict__ swave, const int *__restrict__ wcol,
    const sell_pair *__restrict__ stream, const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y,
    const unsigned char *__restrict__ is_boundary, double *__restrict__ part, const kmcf_scalars *__restrict__ S, int check_done,
    const double *__restrict__ diagv)
{
    constexpr int W = 1 << LW, WQ = W / KMCF_BLOCK - 1;
    constexpr unsigned int OFFMASK = (unsigned int)((W - 1) << 3);
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    __shared__ double xs[2][W];
    if (check_done && S->done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Cx = (n_tiles + 7) >> 3;
    const int gmax = min(Cx, n_tiles - xcd * Cx);
    const int nt = gmax > bi ? (gmax - bi + nb8 - 1) / nb8 : 0;
    double dot = 0.0;
    int buf = 0;
    for (int k = 0; k < nt; ++k, buf ^= 1) {
        const int c = xcd * Cx + bi + k * nb8;
        const int4 d = tile4[c];                             // (first row, rows, first window slot, outside columns)
        const int2 sw = swave[c * 4 + wv];                   // (first 8-byte group of this wave's stream, steps)
        const bool has_row = tid < d.y;
        const int row = d.x + min(tid, d.y - 1);
        const double xrow = x[row], dg = diagv[row];
        double *xb = xs[buf];
        xb[tid] = has_row ? xrow : 0.0;                      // window slots 0 .. 255: the tile's own rows
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int slot = q * KMCF_BLOCK + tid;
            xb[KMCF_BLOCK + slot] = slot < d.w ? x[wcol[d.z + slot]] : 0.0;       // (slots past the window, the padding target W - 1 among them: 0.0)
        }
        __syncthreads();                                     // (buffer `buf` was last read two tiles ago: every thread has passed the barrier in between)
        const char *base = reinterpret_cast<const char *>(xb);
        const sell_pair *sp = stream + sw.x + lane;
        const dvec2 *vp = reinterpret_cast<const dvec2 *>(vals + ((size_t)sw.x + lane) * 4);
        double s = 0.0;
        int q = 0;
        for (; q + 2 <= sw.y; q += 2) {                      // two steps per trip: their loads in flight together
            const sell_pair e0 = sp[q * 64], e1 = sp[(q + 1) * 64];
            const dvec2 a0 = vp[(size_t)q * 128], a1 = vp[(size_t)q * 128 + 1], b0 = vp[(size_t)(q + 1) * 128], b1 = vp[(size_t)(q + 1) * 128 + 1];
            s += a0.x * *reinterpret_cast<const double *>(base + ((e0.x & 0xffffu) & OFFMASK));
            s += a0.y * *reinterpret_cast<const double *>(base + ((e0.x >> 16) & OFFMASK));
            s += a1.x * *reinterpret_cast<const double *>(base + ((e0.y & 0xffffu) & OFFMASK));
            s += a1.y * *reinterpret_cast<const double *>(base + ((e0.y >> 16) & OFFMASK));
            s += b0.x * *reinterpret_cast<const double *>(base + ((e1.x & 0xffffu) & OFFMASK));
            s += b0.y * *reinterpret_cast<const double *>(base + ((e1.x >> 16) & OFFMASK));
            s += b1.x * *reinterpret_cast<const double *>(base + ((e1.y & 0xffffu) & OFFMASK));
            s += b1.y * *reinterpret_cast<const double *>(base + ((e1.y >> 16) & OFFMASK));
        }
        if (q < sw.y) {
            const sell_pair e0 = sp[q * 64];
            const dvec2 a0 = vp[(size_t)q * 128], a1 = vp[(size_t)q * 128 + 1];
            s += a0.x * *reinterpret_cast<const double *>(base + ((e0.x & 0xffffu) & OFFMASK));
            s += a0.y * *reinterpret_cast<const double *>(base + ((e0.x >> 16) & OFFMASK));
            s += a1.x * *reinterpret_cast<const double *>(base + ((e0.y & 0xffffu) & OFFMASK));
            s += a1.y * *reinterpret_cast<const double *>(base + ((e0.y >> 16) & OFFMASK));
        }
        bool valid = has_row;
        if (SKIP_BOUNDARY) valid = valid && is_boundary[row] == 0;
        if (valid) {
            s += dg * xrow;
            y[row] = s;
            if (DOT) dot += xrow * s;
        }
    }
    if (DOT) {
        __syncthreads();
        double t = block_sum_256(dot, &xs[0][0]);
        if (tid == 0) part[blockIdx.x] = t;
    }
}

// values of d_val into the row-per-lane order (and the diagonal into d_diagv): whenever the values changed
template <int LPR>
__global__ __launch_bounds__(KMCF_BLOCK) void sellv_refresh_kernel(int n, const int *__restrict__ row_ptr, const int *__restrict__ diag_pos,
                                                                   const double *__restrict__ val, const int *__restrict__ sell_pos,
                                                                   double *__restrict__ sellv, double *__restrict__ diagv)
{
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    for (int r = blockIdx.x * RPB + threadIdx.x / LPR; r < n; r += gridDim.x * RPB) {
        const int b = row_ptr[r], e = row_ptr[r + 1], dp = diag_pos[r];
        const int len = e - b - (dp >= 0 ? 1 : 0);
        const int pos0 = sell_pos[r];
        if (lane == 0) diagv[r] = dp >= 0 ? val[dp] : 0.0;
        for (int k = lane; k < len; k += LPR) {
            const int j = b + k + ((dp >= 0 && b + k >= dp) ? 1 : 0);
            sellv[(size_t)pos0 + (size_t)(k >> 2) * 256 + (k & 3)] = val[j];
        }
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

// The boundary-row pass of the "direct" peer-to-peer protocol (kmcf_p2p_dev.hpp): the same rows, lanes and sums as
// spmv_vec_kernel<LPR, DOT, false, true>, but the block first waits (bounded) for the flags of this SpMV's halo and
// then reads halo columns in place from buffer (seq & 1) of the landing zone in this rank's window -- no wait / copy
// kernel, no second stream, no event between the interior rows and these.
template <int LPR, bool DOT>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_vec_halo_kernel(
    int n_rows, const int *__restrict__ row_ptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ row_list, double *__restrict__ part,
    const kmcf_scalars *__restrict__ S, int check_done, int n_loc, kmcf_p2p_dev pd, u64 seq)
{
    __shared__ double lds4[4];
    if (check_done && S->done) return;
    if ((int)threadIdx.x < pd.n_nb) wait_ge(&pd.flags[threadIdx.x * P2P_FS], seq, pd.timeout, pd.d_err, pd.h_err, 2);
    __syncthreads();
    const bool ok = __hip_atomic_load(pd.d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    const double *land = pd.landing + (size_t)(seq & 1) * pd.n_halo;
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane_in_row = threadIdx.x % LPR;
    const int row_in_block = threadIdx.x / LPR;
    const int G = (n_rows + RPB - 1) / RPB;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
    const int Gx = (G + 7) >> 3;
    double dot = 0.0;
    for (int g = bi; g < Gx && ok; g += nb8) {
        const int grp = xcd * Gx + g;
        const int row = grp * RPB + row_in_block;
        const bool valid = (grp < G) && (row < n_rows);
        const int r = valid ? row_list[row] : 0;
        double s = 0.0;
        if (valid) {
            const int b = row_ptr[r], e = row_ptr[r + 1];
            for (int j = b + lane_in_row; j < e; j += LPR) {
                const int c = col[j];
                const double xv = c >= n_loc ? load_system(land + (c - n_loc)) : x[c];
                s += val[j] * xv;
            }
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

// ------------------------------------------------------------------ long rows
// Rows [n_short, n_loc): one block per chunk of KMCF_LONG_CHUNK entries; the last block to finish (atomic
// counter) adds each row's chunk sums in chunk order and writes y -- deterministic, one launch.  Values always
// come from val (long rows are never dictionary-coded).
template <bool DOT>
__global__ __launch_bounds__(KMCF_BLOCK) void spmv_long_kernel(
    int n_items, const int4 *__restrict__ items, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, double *__restrict__ lpart, unsigned int *__restrict__ ctr,
    double *__restrict__ part, const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ double lds4[4];
    __shared__ int s_last;
    if (check_done && S->done) return;
    const int4 it = items[blockIdx.x];
    double s = 0.0;
    for (int j = it.y + threadIdx.x; j < it.z; j += KMCF_BLOCK) s += val[j] * x[col[j]];
    s = block_sum_256(s, lds4);
    if (threadIdx.x == 0) {
        lpart[blockIdx.x] = s;
        __threadfence();
        s_last = (atomicAdd(ctr, 1u) == (unsigned int)(n_items - 1));
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double dot = 0.0;
    for (int q = threadIdx.x; q < n_items; q += KMCF_BLOCK) {
        const int4 a = items[q];
        if (a.w != q) continue;                              // first chunk of its row: this thread sums the row
        double t = 0.0;
        for (int c = q; c < n_items && items[c].x == a.x; ++c)
            t += __hip_atomic_load(lpart + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written by other blocks
        y[a.x] = t;
        if (DOT) dot += x[a.x] * t;
    }
    if (DOT) {
        const double t = block_sum_256(dot, lds4);
        if (threadIdx.x == 0) part[0] = t;
    }
    if (threadIdx.x == 0) *ctr = 0u;
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
                spmv_vec_kernel<LPR, true, true, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_short, m->d_is_boundary, nullptr, m->d_part_a));
            else
                spmv_vec_kernel<LPR, true, false, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_short, nullptr, nullptr, m->d_part_a));
        } else {
            if (skipb)
                spmv_vec_kernel<LPR, false, true, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_short, m->d_is_boundary, nullptr, nullptr));
            else
                spmv_vec_kernel<LPR, false, false, false><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_VEC_ARGS(m->n_short, nullptr, nullptr, nullptr));
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

template <int LPR>
void launch_vec_halo(kmcf_matrix *m, bool with_dot, bool skip_if_done, u64 seq)
{
    const kmcf_p2p_dev pd = kmcf_p2p_dev_of(m);
    const int chk = skip_if_done ? 1 : 0;
    if (with_dot)
        spmv_vec_halo_kernel<LPR, true><<<m->spmv_grid_b, KMCF_BLOCK, 0, m->comm->stream>>>(
            m->n_boundary_rows, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, m->d_boundary_rows, m->d_part_a + KMCF_MAX_PARTIALS,
            m->d_S, chk, m->n_loc, pd, seq);
    else
        spmv_vec_halo_kernel<LPR, false><<<m->spmv_grid_b, KMCF_BLOCK, 0, m->comm->stream>>>(
            m->n_boundary_rows, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, m->d_boundary_rows, nullptr, m->d_S, chk, m->n_loc, pd, seq);
}

void launch_vec_halo_any(kmcf_matrix *m, bool with_dot, bool skip_if_done, u64 seq)
{
    switch (m->spmv_lpr) {
        case 4: launch_vec_halo<4>(m, with_dot, skip_if_done, seq); break;
        case 8: launch_vec_halo<8>(m, with_dot, skip_if_done, seq); break;
        case 32: launch_vec_halo<32>(m, with_dot, skip_if_done, seq); break;
        case 64: launch_vec_halo<64>(m, with_dot, skip_if_done, seq); break;
        default: launch_vec_halo<16>(m, with_dot, skip_if_done, seq); break;
    }
}

#define KMCF_STREAM_ARGS(isb, part) \
    m->n_chunks, m->d_chunk_row, m->d_row_ptr, m->d_col, m->d_val, m->d_p, m->d_Ap, isb, part, m->d_S, chk

template <int U, int LPR2>
void launch_stream(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    hipStream_t st = m->comm->stream;
    const int chk = skip_if_done ? 1 : 0;
    const int grid = kmcf_interior_grid(m);
    const bool skipb = (m->n_halo > 0);
#define KMCF_STREAM_LAUNCH(NT)                                                                                                       \
    if (with_dot) {                                                                                                                  \
        if (skipb) spmv_stream_kernel<U, LPR2, true, true, NT><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(m->d_is_boundary, m->d_part_a)); \
        else spmv_stream_kernel<U, LPR2, true, false, NT><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(nullptr, m->d_part_a));  \
    } else {                                                                                                                         \
        if (skipb) spmv_stream_kernel<U, LPR2, false, true, NT><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(m->d_is_boundary, nullptr)); \
        else spmv_stream_kernel<U, LPR2, false, false, NT><<<grid, KMCF_BLOCK, 0, st>>>(KMCF_STREAM_ARGS(nullptr, nullptr));      \
    }
    if (m->stream_nt) { KMCF_STREAM_LAUNCH(true) } else { KMCF_STREAM_LAUNCH(false) }
#undef KMCF_STREAM_LAUNCH
}

#define KMCF_WINDOW_ARGS(isb, part) \
    m->n_tiles, m->d_tile, m->d_row_ptr, m->d_wcol, m->d_idx16, m->d_val, m->d_p, m->d_Ap, isb, part, m->d_S, chk
#define KMCF_WCODE_ARGS(isb, part) \
    m->n_tiles, m->d_tile4, m->d_tbase, m->d_row_ptr, m->d_wcol, m->d_idx16, m->d_p, m->d_Ap, isb, part, m->d_S, chk, m->d_dict, m->d_diagv

template <typename K, typename... A>
void run_or_query(K kernel, bool launch, int *per_cu, int grid, hipStream_t st, A... args)
{
    if (launch) kernel<<<grid, KMCF_BLOCK, 0, st>>>(args...);
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kernel, KMCF_BLOCK, 0) != hipSuccess) *per_cu = 0;
}

// One place that names every window-kernel instance.  which: 0 plain (values streamed), 1 dictionary-coded.
// launch = false: return the resident blocks per CU of that instance instead of launching it.
template <int U, int WQ>
int window_dispatch(kmcf_matrix *m, int which, bool launch, bool with_dot, bool skip_if_done)
{
    hipStream_t st = m->comm->stream;
    const int chk = skip_if_done ? 1 : 0;
    const bool skipb = (m->n_halo > 0);
    const unsigned char *isb = skipb ? m->d_is_boundary : nullptr;
    double *part = with_dot ? m->d_part_a : nullptr;
    const int grid = launch ? kmcf_interior_grid(m) : 0;
    int pc = 0;
    if (which == 0) {
#define KMCF_WINDOW_LAUNCH(NT)                                                                                                            \
        if (with_dot) {                                                                                                                   \
            if (skipb) run_or_query(spmv_window_kernel<U, WQ, 4, true, true, NT>, launch, &pc, grid, st, KMCF_WINDOW_ARGS(isb, part));    \
            else run_or_query(spmv_window_kernel<U, WQ, 4, true, false, NT>, launch, &pc, grid, st, KMCF_WINDOW_ARGS(isb, part));          \
        } else {                                                                                                                          \
            if (skipb) run_or_query(spmv_window_kernel<U, WQ, 4, false, true, NT>, launch, &pc, grid, st, KMCF_WINDOW_ARGS(isb, part));   \
            else run_or_query(spmv_window_kernel<U, WQ, 4, false, false, NT>, launch, &pc, grid, st, KMCF_WINDOW_ARGS(isb, part));         \
        }
        if (m->stream_nt) { KMCF_WINDOW_LAUNCH(true) } else { KMCF_WINDOW_LAUNCH(false) }
#undef KMCF_WINDOW_LAUNCH
    } else {
        if (with_dot) {
            if (skipb) run_or_query(spmv_wcode_kernel<U, WQ, true, true>, launch, &pc, grid, st, KMCF_WCODE_ARGS(isb, part));
            else run_or_query(spmv_wcode_kernel<U, WQ, true, false>, launch, &pc, grid, st, KMCF_WCODE_ARGS(isb, part));
        } else {
            if (skipb) run_or_query(spmv_wcode_kernel<U, WQ, false, true>, launch, &pc, grid, st, KMCF_WCODE_ARGS(isb, part));
            else run_or_query(spmv_wcode_kernel<U, WQ, false, false>, launch, &pc, grid, st, KMCF_WCODE_ARGS(isb, part));
        }
    }
    return pc;
}

#define KMCF_SELL_ARGS(isb, part) \
    m->n_sell_tiles, m->d_sell_tile, m->d_sell_wave, m->d_sell_lrow, m->d_sell_wcol, reinterpret_cast<const sell_pair *>(m->d_sell), \
        m->d_p, m->d_Ap, isb, part, m->d_S, chk, m->d_dict, m->d_diagv

template <int NQ, int LW, int ND, bool IDENT>
int sell_dispatch1(kmcf_matrix *m, bool launch, bool with_dot, bool skip_if_done)
{
    hipStream_t st = m->comm->stream;
    const int chk = skip_if_done ? 1 : 0;
    const bool skipb = (m->n_halo > 0);
    const unsigned char *isb = skipb ? m->d_is_boundary : nullptr;
    double *part = with_dot ? m->d_part_a : nullptr;
    const int grid = launch ? m->sell_grid : 0;
    int pc = 0;
    if constexpr (IDENT) {
        if (!skipb && m->sell_nt && launch) {            // (same registers and LDS: the occupancy query of the plain instance holds)
            if (with_dot) run_or_query(spmv_sell_kernel<NQ, LW, ND, true, false, true, true>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
            else run_or_query(spmv_sell_kernel<NQ, LW, ND, false, false, true, true>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
            return pc;
        }
    }
    if (with_dot) {
        if (skipb) run_or_query(spmv_sell_kernel<NQ, LW, ND, true, true, IDENT>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
        else run_or_query(spmv_sell_kernel<NQ, LW, ND, true, false, IDENT>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
    } else {
        if (skipb) run_or_query(spmv_sell_kernel<NQ, LW, ND, false, true, IDENT>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
        else run_or_query(spmv_sell_kernel<NQ, LW, ND, false, false, IDENT>, launch, &pc, grid, st, KMCF_SELL_ARGS(isb, part));
    }
    return pc;
}

template <int NQ, int LW, int ND>
int sell_dispatch(kmcf_matrix *m, bool launch, bool with_dot, bool skip_if_done)
{
    return m->sell_ident ? sell_dispatch1<NQ, LW, ND, true>(m, launch, with_dot, skip_if_done)
                         : sell_dispatch1<NQ, LW, ND, false>(m, launch, with_dot, skip_if_done);
}

// instantiated (steps, log2 window, dictionary size) triples; dictionaries of one value run as two (second = 0)
constexpr int KMCF_SELL_NQ[] = {8, 13, 16};

int sell_dispatch_any(kmcf_matrix *m, bool launch, bool with_dot, bool skip_if_done)
{
    constexpr int LW = 10;
    const int nd = m->dict_n <= 2 ? 2 : 3;
    switch (m->sell_nq * 10 + nd) {
        case 82: return sell_dispatch<8, LW, 2>(m, launch, with_dot, skip_if_done);
        case 83: return sell_dispatch<8, LW, 3>(m, launch, with_dot, skip_if_done);
        case 132: return sell_dispatch<13, LW, 2>(m, launch, with_dot, skip_if_done);
        case 133: return sell_dispatch<13, LW, 3>(m, launch, with_dot, skip_if_done);
        case 162: return sell_dispatch<16, LW, 2>(m, launch, with_dot, skip_if_done);
        default: return sell_dispatch<16, LW, 3>(m, launch, with_dot, skip_if_done);
    }
}

inline bool sell_active(const kmcf_matrix *m) { return m->spmv_kind == 2 && m->coded && m->sell_ok && m->dict_n <= 3; }

#define KMCF_SELLV_ARGS(isb, part) \
    m->n_sell_tiles, m->d_sell_tile, m->d_sell_wave, m->d_sell_wcol, reinterpret_cast<const sell_pair *>(m->d_sell), m->d_sellv, m->d_p, m->d_Ap, \
        isb, part, m->d_S, chk, m->d_diagv

int sellv_dispatch(kmcf_matrix *m, bool launch, bool with_dot, bool skip_if_done)
{
    constexpr int LW = 10;
    hipStream_t st = m->comm->stream;
    const int chk = skip_if_done ? 1 : 0;
    const bool skipb = (m->n_halo > 0);
    const unsigned char *isb = skipb ? m->d_is_boundary : nullptr;
    double *part = with_dot ? m->d_part_a : nullptr;
    const int grid = launch ? m->sellv_grid : 0;
    int pc = 0;
    if (with_dot) {
        if (skipb) run_or_query(spmv_sellv_kernel<LW, true, true>, launch, &pc, grid, st, KMCF_SELLV_ARGS(isb, part));
        else run_or_query(spmv_sellv_kernel<LW, true, false>, launch, &pc, grid, st, KMCF_SELLV_ARGS(isb, part));
    } else {
        if (skipb) run_or_query(spmv_sellv_kernel<LW, false, true>, launch, &pc, grid, st, KMCF_SELLV_ARGS(isb, part));
        else run_or_query(spmv_sellv_kernel<LW, false, false>, launch, &pc, grid, st, KMCF_SELLV_ARGS(isb, part));
    }
    return pc;
}

// the f64 row-per-lane stream exists and holds the current values (allocated on first use, refreshed when dirty)
int sellv_prepare(kmcf_matrix *m)
{
    hipStream_t st = m->comm->stream;
    if (!m->d_sellv) {
        const size_t n = (size_t)m->n_sell_entries + 128 * 4;
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sellv), n * sizeof(double)));
        KMCF_HIP(hipMemsetAsync(m->d_sellv, 0, n * sizeof(double), st));              // the padding entries stay 0.0
        m->sellv_dirty = true;
    }
    if (m->sellv_dirty) {
        constexpr int LPR = 4;
        sellv_refresh_kernel<LPR><<<grid_for(m->n_short, KMCF_BLOCK / LPR), KMCF_BLOCK, 0, st>>>(m->n_short, m->d_row_ptr, m->d_diag_pos, m->d_val,
                                                                                                m->d_sell_pos, m->d_sellv, m->d_diagv);
        KMCF_HIP(hipGetLastError());
        m->sellv_dirty = false;
    }
    return KMCF_OK;
}


// The stream's value codes follow d_idx16's (which the assembly kernels write): one pass whenever they changed.
template <int LPR>
__global__ __launch_bounds__(KMCF_BLOCK) void sell_refresh_kernel(int n, const int *__restrict__ row_ptr,
                                                                  const int *__restrict__ diag_pos,
                                                                  const unsigned short *__restrict__ idx16,
                                                                  const int *__restrict__ sell_pos,
                                                                  unsigned short *__restrict__ sell, int lw)
{
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const unsigned int slot_part = (unsigned int)(((1 << lw) - 1) << 3);
    for (int r = blockIdx.x * RPB + threadIdx.x / LPR; r < n; r += gridDim.x * RPB) {
        const int b = row_ptr[r], e = row_ptr[r + 1], dp = diag_pos[r];
        const int len = e - b - (dp >= 0 ? 1 : 0);
        const int pos0 = sell_pos[r];
        for (int k = lane; k < len; k += LPR) {
            const int j = b + k + ((dp >= 0 && b + k >= dp) ? 1 : 0);
            const unsigned int code = (unsigned int)idx16[j] >> KMCF_SLOT_BITS;
            const int pos = pos0 + (k >> 2) * 256 + (k & 3);
            sell[pos] = (unsigned short)((sell[pos] & slot_part) | (code << (lw + 3)));
        }
    }
}

int sell_refresh(kmcf_matrix *m)
{
    if (!m->sell_dirty) return KMCF_OK;
    constexpr int LPR = 4;
    sell_refresh_kernel<LPR><<<grid_for(m->n_short, KMCF_BLOCK / LPR), KMCF_BLOCK, 0, m->comm->stream>>>(
        m->n_short, m->d_row_ptr, m->d_diag_pos, m->d_idx16, m->d_sell_pos, m->d_sell, m->sell_lw);
    KMCF_HIP(hipGetLastError());
    m->sell_dirty = false;
    return KMCF_OK;
}

int window_dispatch_any(kmcf_matrix *m, int which, bool launch, bool with_dot, bool skip_if_done)
{
    switch (m->spmv_u * 100 + m->spmv_wmax / KMCF_BLOCK) {
        case 402: return window_dispatch<4, 2>(m, which, launch, with_dot, skip_if_done);
        case 802: return window_dispatch<8, 2>(m, which, launch, with_dot, skip_if_done);
        case 804: return window_dispatch<8, 4>(m, which, launch, with_dot, skip_if_done);
        case 1603: return window_dispatch<16, 3>(m, which, launch, with_dot, skip_if_done);
        case 1604: return window_dispatch<16, 4>(m, which, launch, with_dot, skip_if_done);
        default: return window_dispatch<8, 3>(m, which, launch, with_dot, skip_if_done);
    }
}

// The window kernels walk their tiles in a static loop, so the grid must be exactly what the chip holds at
// once: with more blocks than resident slots the surplus ones start only after a first-wave block has
// finished its whole loop (measured 116 us at 2048 blocks vs 102 us at 7 blocks x 256 CUs; 108 vs 72 us
// for the coded kernel when 512 B more LDS dropped it from 7 to 6 blocks per CU under an unchanged grid).
int window_grid(kmcf_matrix *m, int which)
{
    int cus = 0;
    const int per_cu = window_dispatch_any(m, which, false, true, false);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->comm->device) != hipSuccess) cus = 0;
    const int resident = per_cu * cus / kmcf_device_share();
    int g = grid_for(m->n_tiles, 1);
    if (resident >= 8 && g > resident) g = resident / 8 * 8;
    return g;
}

int sell_grid(kmcf_matrix *m)
{
    int cus = 0;
    const int per_cu = sell_dispatch_any(m, false, true, false);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->comm->device) != hipSuccess) cus = 0;
    const int resident = per_cu * cus / kmcf_device_share();
    int g = grid_for(m->n_sell_tiles, 1);
    if (resident >= 8 && g > resident) g = resident / 8 * 8;
    return g;
}

void launch_interior(kmcf_matrix *m, bool with_dot, bool skip_if_done)
{
    if (sell_active(m)) {
        if (sell_refresh(m) != KMCF_OK) return;
        sell_dispatch_any(m, true, with_dot, skip_if_done);
    } else if (kmcf_sellv_usable(m)) {
        if (sellv_prepare(m) != KMCF_OK) return;
        sellv_dispatch(m, true, with_dot, skip_if_done);
    } else if (m->spmv_kind == 2) {
        window_dispatch_any(m, m->coded ? 1 : 0, true, with_dot, skip_if_done);
    } else if (m->spmv_kind == 1) {
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

// Cuts the rows into tiles for the window kernels: whole rows, at most 256*u entries, 8*u rows (u/8 full
// passes of the 4-lanes-per-row reduction) and 256*wq distinct columns per tile (compact-halo column ids, so
// halo slots are window columns like any other).  *ok stays false (nothing allocated) if a row does not fit
// a tile or, with `judge`, if a window column is used by fewer than two entries on average: columns too
// scattered for a window to pay off, the stream kernel's direct gathers serve those better (K: ~5 entries
// per window column).
int plan_sell(kmcf_matrix *m, const std::vector<int> &col);

int plan_window(kmcf_matrix *m, int u, int wq, bool judge, bool *ok)
{
    *ok = false;
    const int n = m->n_short;                        // long rows have their own kernel
    const std::vector<int> &rp = m->h_row_ptr;
    if (n == 0 || rp[n] == 0) return KMCF_OK;
    // the row limit serves the coded kernel (full passes of its row lanes); the plain kernel, whose cost is
    // the value stream, prefers tiles filled to the entry limit
    const char *ce = getenv("KMCF_SPMV_CODED");
    const bool for_coded = m->expect_coded && !(ce && atoi(ce) == 0);
    // (the coded kernel reads a tile's slot stream from an aligned start up to u - 1 entries early: spmv_wcode_kernel)
    const int cap = KMCF_BLOCK * u - (for_coded ? u : 0), wmax = KMCF_BLOCK * wq, row_cap = for_coded ? 8 * u : n;
    m->tiles_for_coded = for_coded;
    std::vector<int> col((size_t)m->nnz);
    KMCF_HIP(hipMemcpy(col.data(), m->d_col, col.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> slot((size_t)m->n_loc + m->n_halo, -1);
    std::vector<int2> tiles;
    std::vector<int> wcol, uniq;
    std::vector<unsigned short> idx((size_t)m->nnz);
    int r = 0;
    while (r < n) {
        uniq.clear();
        int e = r;
        while (e < n && e - r < row_cap && rp[e + 1] - rp[r] <= cap) {
            const size_t before = uniq.size();
            for (int j = rp[e]; j < rp[e + 1]; ++j)
                if (slot[col[j]] < 0) { slot[col[j]] = 0; uniq.push_back(col[j]); }
            if ((int)uniq.size() > wmax) {
                for (size_t q = before; q < uniq.size(); ++q) slot[uniq[q]] = -1;
                uniq.resize(before);
                break;
            }
            ++e;
        }
        if (e == r) return KMCF_OK;              // one row alone exceeds a tile
        std::sort(uniq.begin(), uniq.end());
        for (size_t q = 0; q < uniq.size(); ++q) slot[uniq[q]] = (int)q;
        for (int j = rp[r]; j < rp[e]; ++j) idx[j] = (unsigned short)slot[col[j]];
        for (int cj : uniq) slot[cj] = -1;
        tiles.push_back(make_int2(r, (int)wcol.size()));
        wcol.insert(wcol.end(), uniq.begin(), uniq.end());
        r = e;
    }
    tiles.push_back(make_int2(n, (int)wcol.size()));
    const int nt = (int)tiles.size() - 1;
    if (judge && double(rp[n]) < 2.0 * double(wcol.size())) return KMCF_OK;
    if (getenv("KMCF_SPMV_VERBOSE"))
        fprintf(stderr, "kmcf window plan: %d tiles, %.1f rows, %.1f nnz, %.1f window columns per tile\n", nt, double(n) / nt,
                double(rp[n]) / nt, double(wcol.size()) / nt);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_tile), tiles.size() * sizeof(int2)));
    wcol.push_back(0);                                    // spare elements: wcode_issue_loads clamps, never branches
    idx.resize(idx.size() + KMCF_BLOCK * 16 + 16, 0);    // the coded kernel's block-wide slot load may run past the last tile
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_wcol), wcol.size() * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_idx16), idx.size() * sizeof(unsigned short)));
    KMCF_HIP(hipMemcpy(m->d_tile, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    {   // self-contained descriptors for the coded kernel's prefetch pipeline (no dependent loads)
        std::vector<int4> t4((size_t)nt + 1);
        std::vector<int> tb((size_t)nt + 1);
        for (int c = 0; c < nt; ++c) {
            t4[c] = make_int4(tiles[c].x, tiles[c + 1].x - tiles[c].x, tiles[c].y, tiles[c + 1].y - tiles[c].y);
            tb[c] = rp[tiles[c].x];
        }
        t4[nt] = make_int4(n, 1, 0, 0);
        tb[nt] = rp[n];
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_tile4), t4.size() * sizeof(int4)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_tbase), tb.size() * sizeof(int)));
        KMCF_HIP(hipMemcpy(m->d_tile4, t4.data(), t4.size() * sizeof(int4), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(m->d_tbase, tb.data(), tb.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    KMCF_HIP(hipMemcpy(m->d_wcol, wcol.data(), wcol.size() * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_idx16, idx.data(), idx.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    m->n_tiles = nt;
    m->n_wcols = (int64_t)wcol.size() - 1;              // without the spare element
    m->spmv_wmax = wmax;
    // diagonal positions and the buffers of the dictionary-coded variant (codes are written later, by
    // kmcf_matrix_encode_values or by the K assembly)
    m->h_diag_pos.assign((size_t)n, -1);
    for (int i = 0; i < n; ++i)
        for (int j = rp[i]; j < rp[i + 1]; ++j)
            if (col[j] == i) { m->h_diag_pos[i] = j; break; }
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_diag_pos), (size_t)n * sizeof(int)));
    KMCF_HIP(hipMemcpy(m->d_diag_pos, m->h_diag_pos.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_dict), 64 * sizeof(double)));
    KMCF_HIP(hipMemset(m->d_dict, 0, 64 * sizeof(double)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_diagv), (size_t)std::max(m->n_loc, 1) * sizeof(double)));   // (every row: the row-wise K assembly writes all of them)
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_code_fail), sizeof(int)));
    m->coded = false;
    *ok = true;
    KMCF_TRY(plan_sell(m, col));                        // (the coded kernel when the values get a dictionary, the f64 one otherwise)
    m->sellv_grid = 0;
    if (m->sell_ok && m->sell_ident && m->sell_lw == 10 && env_int("KMCF_SPMV_SELLV", 1) != 0) {
        int cus = 0;
        const int per_cu = sellv_dispatch(m, false, true, false);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->comm->device) != hipSuccess) cus = 0;
        const int resident = per_cu * cus / kmcf_device_share();
        int g = grid_for(m->n_sell_tiles, 1);
        if (resident >= 8 && g > resident) g = resident / 8 * 8;
        m->sellv_grid = g;
    }
    return KMCF_OK;
}

// Row-per-lane layout for the coded kernel (spmv_sell_kernel).  Declines (sell_ok stays false, the coded window
// kernel runs) when a row holds more off-diagonal entries than the largest instantiated register file (4 x 16)
// or when padding would add more than half to the stream.
// Tile limits of the row-per-lane layout: window slots per class (the first 256 are the tile's own rows, slot
// W - 1 stays empty: the padding entries' target) and rows per tile -- 256 where that still gives every CU a tile, fewer on small matrices.
struct sell_params { int lw, ecap, row_cap; };
sell_params sell_plan_params(int n)
{
    sell_params p;
    p.lw = 10;
    p.ecap = (1 << p.lw) - KMCF_BLOCK - 1;          // columns outside the tile's own rows
    // (a rank's share of the 40 nm matrix in an 8-rank group, 225 k rows: 7.0 us with 256-row tiles, 10.8 with
    // 128, 13.8 with 64 -- large tiles win as long as every CU gets one)
    int row_cap = KMCF_BLOCK;
    while (row_cap > 64 && n / row_cap < 256) row_cap /= 2;
    p.row_cap = std::min(KMCF_BLOCK, std::max(64, env_int("KMCF_SPMV_SELL_ROWS", row_cap) / 64 * 64));
    return p;
}

// Lane order of a tile's rows: the 64 longest rows form wave 0, the next 64 wave 1, ... (a wave's stream is padded
// to ITS longest row), and inside a wave the rows keep their original order -- neighbours in the row order stay
// neighbours, which the gathers of every kernel like (a full sort by length scatters them over the tile:
// 309 instead of 224 runs per window on the 40 nm K matrix, and the CSR stream kernel 7 % slower).
template <class Len>
void sell_lane_order(int nr, Len len, std::vector<int> &ord)
{
    ord.resize((size_t)nr);
    for (int i = 0; i < nr; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return len(a) > len(b); });
    for (int t0 = 0; t0 < nr; t0 += 64) std::sort(ord.begin() + t0, ord.begin() + std::min(t0 + 64, nr));
}

// One tile of the row-per-lane layout, cut greedily from position `start`: rows are added while the tile holds at
// most row_cap rows and references at most ecap columns OUTSIDE itself (its own rows are window slots 0..255 for
// free).  row_id(e): column id of position e's row; cols(e, f): calls f(c) for the off-diagonal columns of
// position e.  mark: per column id, bit 0 = referenced by the tile, bit 1 = row of the tile; `touched` lists
// the referenced ids.  The caller clears the marks of `touched` and of the tile's rows afterwards.
template <class RowId, class Cols>
int sell_cut_tile(int start, int e_max, int row_cap, int ecap, RowId row_id, Cols cols, std::vector<unsigned char> &mark,
                  std::vector<int> &touched)
{
    touched.clear();
    int e = start, ext = 0;
    while (e < e_max && e - start < row_cap) {
        const int id = row_id(e);
        const size_t before = touched.size();
        int next = ext - ((mark[id] & 1) ? 1 : 0);          // referenced from inside so far: now one of the tile's own
        mark[id] |= 2;
        cols(e, [&](int c) {
            if (!(mark[c] & 1)) {
                mark[c] |= 1;
                touched.push_back(c);
                if (!(mark[c] & 2)) ++next;
            }
        });
        if (next > ecap) {
            for (size_t q = before; q < touched.size(); ++q) mark[touched[q]] &= (unsigned char)~1;
            touched.resize(before);
            mark[id] &= (unsigned char)~2;
            break;
        }
        ext = next;
        ++e;
    }
    return e;
}

int plan_sell(kmcf_matrix *m, const std::vector<int> &col)
{
    m->sell_ok = false;
    if (env_int("KMCF_SPMV_SELL", 1) == 0) return KMCF_OK;
    const int n = m->n_short;
    const std::vector<int> &rp = m->h_row_ptr;
    const std::vector<int> &dpos = m->h_diag_pos;
    const sell_params sp = sell_plan_params(n);
    const int lw = sp.lw, W = 1 << lw, row_cap = sp.row_cap;
    int maxlen = 0;
    for (int i = 0; i < n; ++i) maxlen = std::max(maxlen, rp[i + 1] - rp[i] - (dpos[i] >= 0 ? 1 : 0));
    int nq = 0;
    for (int v : KMCF_SELL_NQ)
        if (4 * v >= maxlen) { nq = v; break; }
    if (nq == 0) return KMCF_OK;
    const unsigned short pad = (unsigned short)((W - 1) << 3);
    std::vector<int> slot((size_t)m->n_loc + m->n_halo, -1), uniq, touched, wcol, ord, pos((size_t)n, 0);
    std::vector<unsigned char> mark((size_t)m->n_loc + m->n_halo, 0);
    std::vector<int4> tiles;
    std::vector<int2> waves;
    std::vector<int> lrow;
    std::vector<unsigned short> st;
    int64_t real = 0;
    bool ident = true;                               // rows already sorted inside every tile (kmcf_sell_refine_order)
    auto len_of = [&](int i) { return rp[i + 1] - rp[i] - (dpos[i] >= 0 ? 1 : 0); };
    int r = 0;
    size_t next_cut = 0;
    const std::vector<int> &cuts = m->h_sell_cuts;   // tile ends fixed when the row order was refined (or empty)
    while (r < n) {
        while (next_cut < cuts.size() && cuts[next_cut] <= r) ++next_cut;
        const bool given = next_cut < cuts.size() && cuts[next_cut] <= n && cuts[next_cut] - r <= row_cap;
        const int e_max = given ? cuts[next_cut] : n;
        // a given cut is taken whole: the running count of outside columns depends on the order in which the rows
        // are added (a column stops being "outside" when its row joins), and the rows have been reordered since the
        // cut was made -- only the count of the complete tile is the same, and is checked
        const int e = sell_cut_tile(r, e_max, row_cap, given ? INT_MAX : sp.ecap, [](int i) { return i; },
                                    [&](int i, auto f) {
                                        for (int j = rp[i]; j < rp[i + 1]; ++j)
                                            if (j != dpos[i]) f(col[j]);
                                    },
                                    mark, touched);
        if (e == r) return KMCF_OK;                  // (cannot happen: a row holds <= 64 entries)
        uniq.clear();
        for (int c : touched) {
            if (c >= r && c < e) slot[c] = c - r;    // one of the tile's rows: its slot is its lane
            else uniq.push_back(c);
            mark[c] = 0;
        }
        for (int i = r; i < e; ++i) mark[i] = 0;
        if ((int)uniq.size() > sp.ecap) {            // (a cut that does not fit: made for another pattern)
            for (int c : touched) slot[c] = -1;
            return KMCF_OK;
        }
        std::sort(uniq.begin(), uniq.end());
        for (size_t q = 0; q < uniq.size(); ++q) slot[uniq[q]] = KMCF_BLOCK + (int)q;
        const int nr = e - r;
        sell_lane_order(nr, [&](int a) { return len_of(r + a); }, ord);
        for (int i = 0; i < nr; ++i) ident = ident && ord[i] == i;
        tiles.push_back(make_int4(r, nr, (int)wcol.size(), (int)uniq.size()));
        for (int t = 0; t < KMCF_BLOCK; ++t) lrow.push_back(t < nr ? ord[t] : -1);
        for (int w = 0; w < KMCF_BLOCK / 64; ++w) {
            const int t0 = 64 * w;
            int wlen = 0;
            for (int t = t0; t < std::min(t0 + 64, nr); ++t) wlen = std::max(wlen, len_of(r + ord[t]));
            const int wq = (wlen + 3) / 4;                                  // steps of the wave's longest row
            const size_t base = st.size() / 4;                              // in 8-byte groups
            waves.push_back(make_int2((int)base, wq));
            st.resize(st.size() + (size_t)wq * 256, pad);
            for (int t = t0; t < std::min(t0 + 64, nr); ++t) {
                const int row = r + ord[t];
                pos[row] = (int)((base + (t - t0)) * 4);
                int k = 0;
                for (int j = rp[row]; j < rp[row + 1]; ++j) {
                    if (j == dpos[row]) continue;
                    st[(size_t)pos[row] + (size_t)(k >> 2) * 256 + (k & 3)] = (unsigned short)(slot[col[j]] << 3);
                    ++k;
                }
                real += k;
            }
        }
        for (int cj : touched) slot[cj] = -1;
        wcol.insert(wcol.end(), uniq.begin(), uniq.end());
        r = e;
    }
    if (st.size() / 4 > (size_t)0x7fffff00 || (double)st.size() > 1.5 * (double)real + 4096.0 * tiles.size()) return KMCF_OK;
    const int nt = (int)tiles.size();
    if (getenv("KMCF_SPMV_VERBOSE"))
        fprintf(stderr, "kmcf row-per-lane plan: %d tiles of <= %d rows%s, %.1f rows, %.1f window columns per tile, %lld entries + %.1f %% padding, %d steps\n",
                nt, row_cap, ident ? " (sorted in place)" : "", double(n) / nt, double(wcol.size()) / nt, (long long)real, 100.0 * (double(st.size()) / double(std::max<int64_t>(real, 1)) - 1.0), nq);
    if (getenv("KMCF_SPMV_VERBOSE")) {
        std::vector<long long> hist(20, 0), rows(20, 0);
        for (const int2 &w : waves) ++hist[std::min(w.y, 19)];
        for (int i = 0; i < n; ++i) ++rows[std::min((len_of(i) + 3) / 4, 19)];
        long long runs = 0, lines = 0;
        for (const int4 &t : tiles)
            for (int q = 0; q < t.w; ++q) {
                if (q == 0 || wcol[t.z + q] != wcol[t.z + q - 1] + 1) ++runs;
                if (q == 0 || wcol[t.z + q] / 8 != wcol[t.z + q - 1] / 8) ++lines;
            }
        fprintf(stderr, "  window: %.1f runs of consecutive columns, %.1f 64-byte lines of x per tile\n", double(runs) / nt, double(lines) / nt);
        fprintf(stderr, "  steps: waves / rows ");
        for (int q = 0; q < 20; ++q)
            if (hist[q] || rows[q]) fprintf(stderr, " %d: %lld / %lld;", q, hist[q], rows[q]);
        fprintf(stderr, "\n");
    }
    m->n_sell_entries = (int64_t)st.size();
    m->n_sell_wcols = (int64_t)wcol.size();
    st.resize(st.size() + 128 * 4, pad);              // idle waves read one group per lane at their (empty) stream's start
    wcol.push_back(0);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell_tile), tiles.size() * sizeof(int4)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell_wave), waves.size() * sizeof(int2)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell_lrow), lrow.size() * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell_wcol), wcol.size() * sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell), st.size() * sizeof(unsigned short)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_sell_pos), (size_t)n * sizeof(int)));
    KMCF_HIP(hipMemcpy(m->d_sell_tile, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_sell_wave, waves.data(), waves.size() * sizeof(int2), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_sell_lrow, lrow.data(), lrow.size() * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_sell_wcol, wcol.data(), wcol.size() * sizeof(int), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_sell, st.data(), st.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(m->d_sell_pos, pos.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    m->n_sell_tiles = nt;
    m->sell_lw = lw;
    m->sell_nq = nq;
    m->sell_dirty = true;
    m->sell_ident = ident;
    m->sell_ok = true;
    m->sell_grid = 0;                                 // with the dictionary (its size selects the instance)
    // beyond the Infinity Cache (256 MiB; 10 x 10 cells = 243 MB of format still run 40.2 us plain against 51.9
    // nontemporal) the entry stream is loaded nontemporal; KMCF_SELL_NT overrides
    m->sell_nt = 2.0 * (double)st.size() + 4.0 * (double)wcol.size() + 28.0 * (double)n > 300e6;
    if (const char *e = getenv("KMCF_SELL_NT")) m->sell_nt = atoi(e) != 0;
    return KMCF_OK;
}

// Writes the dictionary code of every entry above its slot bits and the per-row diagonal value.
template <int LPR>
__global__ __launch_bounds__(KMCF_BLOCK) void encode_values_kernel(int n, const int *__restrict__ row_ptr,
                                                                   const int *__restrict__ diag_pos,
                                                                   const double *__restrict__ val,
                                                                   const double *__restrict__ dict, int nd,
                                                                   unsigned short *__restrict__ idx16,
                                                                   double *__restrict__ diagv, int *__restrict__ fail)
{
    __shared__ double sdict[64];
    if (threadIdx.x < 64) sdict[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    constexpr int RPB = KMCF_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    for (int r = blockIdx.x * RPB + threadIdx.x / LPR; r < n; r += gridDim.x * RPB) {
        const int dp = diag_pos[r];
        if (lane == 0) diagv[r] = dp >= 0 ? val[dp] : 0.0;
        for (int j = row_ptr[r] + lane; j < row_ptr[r + 1]; j += LPR) {
            int code = KMCF_CODE_DIAG;
            if (j != dp) {
                const double v = val[j];
                code = -1;
                for (int k = 0; k < nd; ++k)       // bit pattern, not ==: -0.0 and NaNs stay what they are
                    if (__double_as_longlong(sdict[k]) == __double_as_longlong(v)) { code = k; break; }
                if (code < 0) { *fail = 1; code = 0; }
            }
            idx16[j] = (unsigned short)((idx16[j] & ((1 << KMCF_SLOT_BITS) - 1)) | (code << KMCF_SLOT_BITS));
        }
    }
}

}  // namespace

int kmcf_spmv_plan(kmcf_matrix *m)
{
    // f64-value kernels: nontemporal matrix loads once the CSR stream alone is beyond what the caches can keep
    // between two launches (stream_load; KMCF_SPMV_NT = 0 / 1 overrides)
    m->stream_nt = 12.0 * (double)m->nnz > 192e6;
    if (const char *e = getenv("KMCF_SPMV_NT")) m->stream_nt = atoi(e) != 0;
    // vec kernel: lanes per row from the mean row length (K rows hold 4..53 entries, mean 25.8)
    const double mean = m->n_short > 0 ? double(m->h_row_ptr[m->n_short]) / m->n_short : 0.0;
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
    // kind: window kernel unless the plan declines (scattered columns, very long rows), then stream, then vec
    int kind = env_int("KMCF_SPMV_KIND", -1);
    const bool judge = kind < 0;
    if (kind < 0 || kind > 2) kind = 2;
    if (kind == 2) {
        int wq = env_int("KMCF_SPMV_WQ", 2);             // 512 window columns: K tiles need ~300 (measured best)
        if (wq != 2 && wq != 3 && wq != 4) wq = 2;
        // instantiated (U, WQ) pairs: window_dispatch_any
        const int uw = (u == 4 && wq == 2) ? 4 : ((u == 16 && wq >= 3) ? 16 : 8);
        bool ok = false;
        KMCF_TRY(plan_window(m, uw, wq, judge, &ok));
        if (ok) {
            m->spmv_u = uw;
            m->spmv_kind = 2;
            m->spmv_grid = window_grid(m, 0);
            m->spmv_grid_coded = 0;                  // set with the dictionary (kmcf_matrix_set_dictionary)
            return KMCF_OK;
        }
        kind = 1;
    }
    std::vector<int> chunk_row;
    if (kind == 1) {
        const std::vector<int> &rp = m->h_row_ptr;
        chunk_row.push_back(0);
        int r = 0;
        while (r < m->n_short) {
            const int start = rp[r];
            int e = r;
            while (e < m->n_short && rp[e + 1] - start <= cap) ++e;
            if (e == r) { kind = 0; break; }   // a single row exceeds a chunk: vector kernel
            chunk_row.push_back(e);
            r = e;
        }
        if (m->n_short == 0) kind = 0;
    }
    m->spmv_kind = kind;
    if (kind == 1) {
        m->n_chunks = (int)chunk_row.size() - 1;
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_chunk_row), chunk_row.size() * sizeof(int)));
        KMCF_HIP(hipMemcpy(m->d_chunk_row, chunk_row.data(), chunk_row.size() * sizeof(int), hipMemcpyHostToDevice));
        m->spmv_grid = grid_for(m->n_chunks, 1);
    } else {
        m->spmv_grid = grid_for(m->n_short, KMCF_BLOCK / lpr);
    }
    return KMCF_OK;
}

static bool coding_enabled(const kmcf_matrix *m)
{
    if (m->spmv_kind != 2 || !m->d_idx16 || !m->tiles_for_coded) return false;
    const char *e = getenv("KMCF_SPMV_CODED");
    return !(e && atoi(e) == 0);
}

int kmcf_matrix_set_dictionary(kmcf_matrix *m, const double *h_dict, int nd)
{
    m->coded = false;
    m->sellv_dirty = true;                               // (the caller is about to write new values)
    if (!coding_enabled(m)) return KMCF_OK;
    KMCF_CHECK(nd >= 0 && nd <= KMCF_DICT_MAX, KMCF_ERR_ARG, "value dictionary of %d entries", nd);
    bool same = true;
    for (int k = 0; k < 64; ++k) {
        const double v = k < nd ? h_dict[k] : 0.0;
        if (memcmp(&m->h_dict[k], &v, sizeof(v)) != 0) { same = false; m->h_dict[k] = v; }
    }
    if (!same || !m->dict_uploaded) {
        KMCF_HIP(hipMemcpyAsync(m->d_dict, m->h_dict, sizeof(m->h_dict), hipMemcpyHostToDevice, m->comm->stream));
        m->dict_uploaded = true;
    }
    if (m->spmv_grid_coded <= 0) m->spmv_grid_coded = window_grid(m, 1);
    if (m->sell_ok && (m->sell_grid <= 0 || (m->dict_n <= 2) != (nd <= 2))) {
        m->dict_n = nd;
        m->sell_grid = sell_grid(m);
        if (getenv("KMCF_SPMV_VERBOSE")) fprintf(stderr, "kmcf row-per-lane kernel: grid %d for %d tiles\n", m->sell_grid, m->n_sell_tiles);
    }
    m->dict_n = nd;
    m->sell_dirty = true;                               // the caller is about to write (or has just written) the codes
    m->coded = true;
    return KMCF_OK;
}

int kmcf_matrix_encode_values(kmcf_matrix *m, const double *h_dict, int nd)
{
    m->coded = false;
    m->sellv_dirty = true;                               // d_val holds new values
    if (!coding_enabled(m) || m->n_short == 0) return KMCF_OK;
    hipStream_t st = m->comm->stream;
    KMCF_TRY(kmcf_matrix_set_dictionary(m, h_dict, nd));
    m->coded = false;
    KMCF_HIP(hipMemsetAsync(m->d_code_fail, 0, sizeof(int), st));
    constexpr int LPR = 16;
    encode_values_kernel<LPR><<<grid_for(m->n_short, KMCF_BLOCK / LPR), KMCF_BLOCK, 0, st>>>(
        m->n_short, m->d_row_ptr, m->d_diag_pos, m->d_val, m->d_dict, nd, m->d_idx16, m->d_diagv, m->d_code_fail);
    KMCF_HIP(hipGetLastError());
    int fail = 1;
    KMCF_HIP(hipMemcpyAsync(&fail, m->d_code_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    KMCF_HIP(hipStreamSynchronize(st));
    m->coded = (fail == 0);
    return KMCF_OK;
}

int kmcf_matrix_encode_from_host(kmcf_matrix *m, const double *h_val_internal)
{
    m->coded = false;
    m->sellv_dirty = true;
    if (!coding_enabled(m) || m->n_short == 0) return KMCF_OK;
    // distinct off-diagonal values (bit patterns); give up at the first one beyond the dictionary size
    long long dict[KMCF_DICT_MAX];
    int nd = 0;
    const std::vector<int> &rp = m->h_row_ptr;
    int last = -1;
    for (int i = 0; i < m->n_short; ++i) {
        const int dp = m->h_diag_pos[i];
        for (int j = rp[i]; j < rp[i + 1]; ++j) {
            if (j == dp) continue;
            long long b;
            memcpy(&b, &h_val_internal[j], sizeof(b));
            if (last >= 0 && dict[last] == b) continue;
            int k = 0;
            while (k < nd && dict[k] != b) ++k;
            if (k == nd) {
                if (nd == KMCF_DICT_MAX) return KMCF_OK;     // too many distinct values: plain window kernel
                dict[nd++] = b;
            }
            last = k;
        }
    }
    double d[KMCF_DICT_MAX];
    memcpy(d, dict, sizeof(double) * nd);
    return kmcf_matrix_encode_values(m, d, nd);
}

int kmcf_halo_exchange_begin(kmcf_matrix *m)
{
    kmcf_comm *c = m->comm;
    // in a loopback group every rank takes part in the (host-synchronous) exchange, neighbours or not
    const bool loopback = c->group && c->group->nranks > 1 && !c->p2p_active;
    if (m->number_of_neighbours <= 1 && !loopback) return KMCF_OK;
    if (m->n_send > 0 && !c->p2p_active) {          // (the peer-to-peer transport packs inside its put kernel)
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
int kmcf_spmv_device(kmcf_matrix *m, bool with_dot, bool skip_if_done, int flags)
{
    if (kmcf_p2p_direct(m)) {
        // "direct" peer-to-peer protocol: ONE stream, no pack / wait / copy kernels.  The halo of this SpMV (sequence
        // seq) was put by the kernel that produced d_p (flags bit 0 unset and nothing put yet: a stand-alone put now);
        // interior rows; boundary rows after a bounded wait for the flags, halo columns read in place; then the
        // acknowledgement, by the caller's next kernel (flags bit 0) or a small kernel of its own.
        u64 &seq = *kmcf_p2p_halo_seq(m, 0), &seq_put = *kmcf_p2p_halo_seq(m, 1);
        ++seq;
        if (seq_put < seq) {
            KMCF_TRY(kmcf_p2p_direct_put(m, seq, skip_if_done));
            seq_put = seq;
        }
        launch_interior(m, with_dot, skip_if_done);
        KMCF_HIP(hipGetLastError());
        if (m->n_halo > 0 && m->n_boundary_rows > 0) {
            launch_vec_halo_any(m, with_dot, skip_if_done, seq);
            KMCF_HIP(hipGetLastError());
        }
        if (!(flags & 1)) KMCF_TRY(kmcf_p2p_direct_ack(m, seq, skip_if_done));
        return KMCF_OK;
    }
    if (m->sub) KMCF_TRY(kmcf_subop_begin(m, skip_if_done));     // sub-vector: packed, its all-gather under way (comm stream)
    KMCF_TRY(kmcf_halo_exchange_begin(m));
    launch_interior(m, with_dot, skip_if_done);
    KMCF_HIP(hipGetLastError());
    // the compute stream always waits for the exchange it started: a rank that only SENDS (n_send > 0, no halo
    // of its own) must not repack d_send_buf while the previous send is still reading it
    if (m->n_halo > 0 || m->n_send > 0) KMCF_TRY(kmcf_halo_exchange_end(m));
    if (m->n_halo > 0) {
        if (m->n_boundary_rows > 0) {
            launch_vec_any(m, with_dot, skip_if_done, true);
            KMCF_HIP(hipGetLastError());
        }
    }
    if (m->n_long_items > 0) {
        hipStream_t st = m->comm->stream;
        const int chk = skip_if_done ? 1 : 0;
        double *part = m->d_part_a + 2 * KMCF_MAX_PARTIALS;
        if (with_dot)
            spmv_long_kernel<true><<<m->n_long_items, KMCF_BLOCK, 0, st>>>(m->n_long_items, m->d_long_items, m->d_col, m->d_val, m->d_p,
                                                                          m->d_Ap, m->d_long_part, m->d_long_ctr, part, m->d_S, chk);
        else
            spmv_long_kernel<false><<<m->n_long_items, KMCF_BLOCK, 0, st>>>(m->n_long_items, m->d_long_items, m->d_col, m->d_val, m->d_p,
                                                                           m->d_Ap, m->d_long_part, m->d_long_ctr, part, m->d_S, chk);
        KMCF_HIP(hipGetLastError());
    }
    if (m->sub) KMCF_TRY(kmcf_subop_finish(m, with_dot, skip_if_done));
    return KMCF_OK;
}

bool kmcf_sell_coded_active(const kmcf_matrix *m) { return sell_active(m); }
int kmcf_sell_ready(kmcf_matrix *m) { return sell_refresh(m); }

kmcf_part4 kmcf_spmv_partials(const kmcf_matrix *m)
{
    kmcf_part4 q;
    q.p[0] = m->d_part_a;                           q.n[0] = kmcf_interior_grid(m);
    q.p[1] = m->d_part_a + KMCF_MAX_PARTIALS;       q.n[1] = m->n_halo > 0 ? m->spmv_grid_b : 0;
    q.p[2] = m->d_part_a + 2 * KMCF_MAX_PARTIALS;   q.n[2] = m->n_long_items > 0 ? 1 : 0;
    q.p[3] = m->d_part_a + 3 * KMCF_MAX_PARTIALS;   q.n[3] = m->sub ? m->sub->grid : 0;
    return q;
}

extern "C" int kmcf_matrix_sum_plan(const kmcf_matrix *m, kmcf_sum_plan_t *plan, int *h_tile_first, int *h_tile_rows,
                                    int *h_row_ptr, int *h_col, double *h_val)
{
    KMCF_CHECK(m, KMCF_ERR_ARG, "kmcf_matrix_sum_plan: null matrix");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_matrix_sum_plan: host-only matrix");
    const bool sell = sell_active(m) || kmcf_sellv_usable(m);
    if (plan) {
        memset(plan, 0, sizeof(*plan));
        plan->rows = m->n_loc;
        plan->n_short = m->n_short;
        plan->halo_cols = m->n_halo;
        plan->vec_grid = kmcf_vec_grid(m->n_loc);
        plan->sell_active = sell ? 1 : 0;
        plan->sell_ident = m->sell_ident ? 1 : 0;
        plan->sell_grid = sell ? kmcf_interior_grid(m) : 0;
        plan->sell_tiles = sell ? m->n_sell_tiles : 0;
        plan->boundary_grid = m->n_halo > 0 && m->n_boundary_rows > 0 ? m->spmv_grid_b : 0;
        plan->boundary_lpr = m->spmv_lpr;
        plan->boundary_rows = m->n_boundary_rows;
        plan->long_items = m->n_long_items;
        plan->sub_grid = m->sub ? m->sub->grid : 0;
        plan->cg_variant = kmcf_cg_single_reduction(m) ? 1 : 0;
        if (plan->cg_variant == 1 || (m->comm->nranks == 1 && kmcf_cgr_classic_applies(m)))   // (the reference's recurrence: one rank, small matrices)
            kmcf_cgr_plan_info(const_cast<kmcf_matrix *>(m), &plan->resident_tpb, &plan->resident_g1, nullptr);
    }
    KMCF_HIP(hipSetDevice(m->comm->device));
    KMCF_HIP(hipStreamSynchronize(m->comm->stream));
    if ((h_tile_first || h_tile_rows) && sell && m->n_sell_tiles > 0) {
        std::vector<int4> t((size_t)m->n_sell_tiles);
        KMCF_HIP(hipMemcpy(t.data(), m->d_sell_tile, t.size() * sizeof(int4), hipMemcpyDeviceToHost));
        for (int i = 0; i < m->n_sell_tiles; ++i) {
            if (h_tile_first) h_tile_first[i] = t[i].x;
            if (h_tile_rows) h_tile_rows[i] = t[i].y;
        }
    }
    if (h_row_ptr) memcpy(h_row_ptr, m->h_row_ptr.data(), ((size_t)m->n_loc + 1) * sizeof(int));
    if (h_col && m->nnz) KMCF_HIP(hipMemcpy(h_col, m->d_col, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost));
    if (h_val && m->nnz) KMCF_HIP(hipMemcpy(h_val, m->d_val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return KMCF_OK;
}

extern "C" int kmcf_matrix_halo_columns(const kmcf_matrix *m, int *h_gid)
{
    KMCF_CHECK(m && (h_gid || m->n_halo == 0), KMCF_ERR_ARG, "kmcf_matrix_halo_columns: null argument");
    for (int k = 1; k < m->number_of_neighbours; ++k)
        for (size_t s = 0; s < m->cols_per_neighbour[k].size(); ++s)
            h_gid[m->halo_offset[k] + (int)s] = m->displs[m->neighbours[k]] + m->cols_per_neighbour[k][s];
    return KMCF_OK;
}

extern "C" int kmcf_spmv(kmcf_matrix *m, const double *d_p, double *d_Ap)
{
    // a rank that owns no rows (fewer rows than ranks) still takes part in the exchange; its vectors may be null
    KMCF_CHECK(m && ((d_p && d_Ap) || m->n_loc == 0), KMCF_ERR_ARG, "kmcf_spmv: null argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_spmv: host-only matrix");
    kmcf_comm *c = m->comm;
    KMCF_TRY(kmcf_enter(c));
    KMCF_TRY(kmcf_group_rendezvous(c));
    KMCF_TRY(kmcf_vec_in(m, m->d_p, d_p));
    KMCF_TRY(kmcf_spmv_device(m, false, false));
    KMCF_TRY(kmcf_vec_out(m, d_Ap, m->d_Ap));
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return kmcf_p2p_check(c);
}

extern "C" int kmcf_spmv_bench(kmcf_matrix *m, int reps, int with_dot, float *ms_total)
{
    KMCF_CHECK(m && reps > 0 && ms_total, KMCF_ERR_ARG, "kmcf_spmv_bench: bad argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_spmv_bench: host-only matrix");
    kmcf_comm *c = m->comm;
    KMCF_TRY(kmcf_enter(c));
    KMCF_TRY(kmcf_group_rendezvous(c));
    c->in_solve = true;
    struct leave_t { kmcf_comm *c; ~leave_t() { c->in_solve = false; } } leave{c};
    KMCF_HIP(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; ++i) KMCF_TRY(kmcf_spmv_device(m, with_dot != 0, false));
    KMCF_HIP(hipEventRecord(c->ev_t1, c->stream));
    KMCF_HIP(hipEventSynchronize(c->ev_t1));
    KMCF_HIP(hipEventElapsedTime(ms_total, c->ev_t0, c->ev_t1));
    return KMCF_OK;
}

// Times the pieces of one distributed CG iteration separately (diagnostic for bench.py at N > 1; every
// rank must call it with the same arguments): kind 0 = all-reduce of 3 doubles, 1 = halo exchange alone
// (pack, send/recv, wait), 2 = the SpMV kernels alone (interior + boundary rows, no exchange).
extern "C" int kmcf_comm_bench(kmcf_matrix *m, int kind, int reps, float *ms_total)
{
    KMCF_CHECK(m && reps > 0 && ms_total && kind >= 0 && kind <= 3, KMCF_ERR_ARG, "kmcf_comm_bench: bad argument");
    KMCF_CHECK(kind != 3 || m->sub, KMCF_ERR_ARG, "kmcf_comm_bench: kind 3 (sub-vector all-gather) needs a split operator");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_comm_bench: host-only matrix");
    kmcf_comm *c = m->comm;
    KMCF_TRY(kmcf_enter(c));
    KMCF_TRY(kmcf_group_rendezvous(c));
    c->in_solve = true;
    struct leave_t { kmcf_comm *c; ~leave_t() { c->in_solve = false; } } leave{c};
    KMCF_HIP(hipMemsetAsync(&m->d_S->red[0], 0, 3 * sizeof(double), c->stream));
    KMCF_HIP(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; ++i) {
        if (kind == 0) {
            KMCF_TRY(kmcf_comm_allreduce_sum(c, &m->d_S->red[0], 3));
        } else if (kind == 1) {
            KMCF_TRY(kmcf_halo_exchange_begin(m));
            KMCF_TRY(kmcf_halo_exchange_end(m));
        } else if (kind == 3) {                       // pack + all-gather of the tunnel sub-vector + the wait for it
            KMCF_TRY(kmcf_subop_begin(m, false));
            if (m->sub->gather_pending) { KMCF_HIP(hipStreamWaitEvent(c->stream, c->ev_sub, 0)); m->sub->gather_pending = false; }
        } else {
            launch_interior(m, true, false);
            if (m->n_halo > 0 && m->n_boundary_rows > 0) launch_vec_any(m, true, false, true);
            KMCF_HIP(hipGetLastError());
        }
    }
    KMCF_HIP(hipEventRecord(c->ev_t1, c->stream));
    KMCF_HIP(hipEventSynchronize(c->ev_t1));
    KMCF_HIP(hipEventElapsedTime(ms_total, c->ev_t0, c->ev_t1));
    return KMCF_OK;
}

// Re-plan the SpMV of an existing matrix from the KMCF_SPMV_* environment (tuning aid).
// Refines an internal row order for the row-per-lane layout: the rows of each of its tiles (cut exactly as
// plan_sell will cut them: the cut depends on the rows' column SETS, not on their order inside a tile) are
// put into lane order (sell_lane_order: waves of similar length).  Lane t of a tile then owns row r0 + t: x, diagonal and y of a
// tile are contiguous for the kernel.  cuts receives the end row of every tile: plan_sell must cut there (its
// own greedy cut would try a DIFFERENT next row after a window-limited tile, the next tile's longest).  rp / col: the caller-ordered local pattern (own columns < n_loc, halo
// slots above); perm[i] = caller row of internal row i; only the first n_short entries are touched.
void kmcf_sell_refine_order(int n_short, int n_cols, const int *rp, const int *col, std::vector<int> &perm, std::vector<int> &cuts)
{
    cuts.clear();
    if (env_int("KMCF_SPMV_SELL", 1) == 0 || env_int("KMCF_SPMV_SELL_SORT", 1) == 0 || n_short < 2) return;
    const sell_params sp = sell_plan_params(n_short);
    std::vector<unsigned char> mark((size_t)n_cols, 0);
    std::vector<int> touched, len((size_t)n_short);
    for (int i = 0; i < n_short; ++i) {
        const int r = perm[i];
        bool diag = false;
        for (int j = rp[r]; j < rp[r + 1] && !diag; ++j) diag = col[j] == r;
        len[i] = rp[r + 1] - rp[r] - (diag ? 1 : 0);
    }
    std::vector<int> ord, tmp;
    int i = 0;
    while (i < n_short) {
        const int e = sell_cut_tile(i, n_short, sp.row_cap, sp.ecap, [&](int q) { return perm[q]; },
                                    [&](int q, auto f) {
                                        const int r = perm[q];
                                        bool diag = false;
                                        for (int j = rp[r]; j < rp[r + 1]; ++j) {
                                            if (col[j] == r && !diag) { diag = true; continue; }
                                            f(col[j]);
                                        }
                                    },
                                    mark, touched);
        for (int c : touched) mark[c] = 0;
        for (int q = i; q < e; ++q) mark[perm[q]] = 0;
        if (e == i) { cuts.clear(); return; }         // a row alone exceeds a window: no such layout
        cuts.push_back(e);
        const int nr = e - i;
        sell_lane_order(nr, [&](int a) { return len[i + a]; }, ord);
        tmp.resize((size_t)nr);
        for (int t = 0; t < nr; ++t) tmp[t] = perm[i + ord[t]];
        std::copy(tmp.begin(), tmp.end(), perm.begin() + i);
        for (int t = 0; t < nr; ++t) tmp[t] = len[i + ord[t]];
        std::copy(tmp.begin(), tmp.end(), len.begin() + i);
        i = e;
    }
}

void kmcf_sell_free(kmcf_matrix *m)
{
    void *ptrs[] = {m->d_sell_tile, m->d_sell_wave, m->d_sell_lrow, m->d_sell_wcol, m->d_sell, m->d_sell_pos, m->d_sellv};
    for (void *p : ptrs)
        if (p) hipFree(p);
    m->d_sell_tile = nullptr; m->d_sell_wave = nullptr; m->d_sell_lrow = nullptr;
    m->d_sell_wcol = nullptr; m->d_sell = nullptr; m->d_sell_pos = nullptr; m->d_sellv = nullptr;
    m->sellv_dirty = true; m->sellv_grid = 0;
    m->sell_ok = false;
    m->n_sell_tiles = 0;
}

extern "C" int kmcf_spmv_replan(kmcf_matrix *m)
{
    KMCF_CHECK(m && m->d_val, KMCF_ERR_ARG, "kmcf_spmv_replan: bad matrix");
    KMCF_TRY(kmcf_enter(m->comm));
    KMCF_HIP(hipStreamSynchronize(m->comm->stream));
    if (m->d_chunk_row) { hipFree(m->d_chunk_row); m->d_chunk_row = nullptr; }
    if (m->d_tile) { hipFree(m->d_tile); m->d_tile = nullptr; }
    if (m->d_tile4) { hipFree(m->d_tile4); m->d_tile4 = nullptr; }
    if (m->d_tbase) { hipFree(m->d_tbase); m->d_tbase = nullptr; }
    if (m->d_wcol) { hipFree(m->d_wcol); m->d_wcol = nullptr; }
    if (m->d_idx16) { hipFree(m->d_idx16); m->d_idx16 = nullptr; }
    if (m->d_dict) { hipFree(m->d_dict); m->d_dict = nullptr; }
    if (m->d_diagv) { hipFree(m->d_diagv); m->d_diagv = nullptr; }
    if (m->d_diag_pos) { hipFree(m->d_diag_pos); m->d_diag_pos = nullptr; }
    if (m->d_code_fail) { hipFree(m->d_code_fail); m->d_code_fail = nullptr; }
    kmcf_sell_free(m);
    kmcf_cgr_free(m);                 // (the resident solve's plan follows the row-per-lane layout)
    m->n_tiles = 0;
    m->coded = false;
    m->dict_uploaded = false;
    KMCF_TRY(kmcf_spmv_plan(m));
    // re-code the values now in d_val (tuning aid: a host round trip is fine here)
    std::vector<double> v((size_t)m->nnz);
    KMCF_HIP(hipMemcpy(v.data(), m->d_val, v.size() * sizeof(double), hipMemcpyDeviceToHost));
    return kmcf_matrix_encode_from_host(m, v.data());
}

extern "C" int kmcf_pack(kmcf_comm *c, double *d_packed, const double *d_unpacked, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_pack: bad argument");
    if (n == 0) return KMCF_OK;
    KMCF_TRY(kmcf_enter(c));
    pack_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_packed, d_unpacked, d_indices, n, nullptr, 0);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_unpack(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_unpack: bad argument");
    if (n == 0) return KMCF_OK;
    KMCF_TRY(kmcf_enter(c));
    unpack_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_unpacked, d_packed, d_indices, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_unpack_add(kmcf_comm *c, double *d_unpacked, const double *d_packed, const int *d_indices, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_unpack_add: bad argument");
    if (n == 0) return KMCF_OK;
    KMCF_TRY(kmcf_enter(c));
    unpack_add_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_unpacked, d_packed, d_indices, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" int kmcf_elementwise_vector_vector(kmcf_comm *c, const double *d_a, const double *d_b, double *d_out, int n)
{
    KMCF_CHECK(c && n >= 0, KMCF_ERR_ARG, "kmcf_elementwise_vector_vector: bad argument");
    if (n == 0) return KMCF_OK;
    KMCF_TRY(kmcf_enter(c));
    hadamard_kernel<<<grid_for(n, KMCF_BLOCK), KMCF_BLOCK, 0, c->stream>>>(d_a, d_b, d_out, n);
    KMCF_HIP(hipGetLastError());
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}
