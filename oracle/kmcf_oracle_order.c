/*
 * kmcf_oracle_order.c -- CPU oracle, part 3: the reference's Jacobi-PCG in the SUMMATION ORDER of libkmcfield.
 *
 * TEST INFRASTRUCTURE ONLY (see kmcf_oracle.c): imported by tests/ and __graft_entry__.smoke(), never by the
 * product package.
 *
 * Why this file exists.  The algorithm is the reference's (iterative_solver::conjugate_gradient_jacobi,
 * dist_iterative/dist_conjugate_gradient.cpp:149-276; conjugate_gradient :17-121; the absolute rule of
 * solve_sparse_CG_Jacobi, src/iterative_solvers_gpu.cu:838-858): same recurrence, same operation order, same
 * stopping rule -- orc_pcg_jacobi in kmcf_oracle.c restates it with the rows in the caller's order and pairwise
 * dots.  But the K system spans conductances 1 ... 1e-8 and is solved to 1e-14 N: the iteration count at which
 * r.z/b.b crosses tol^2 moves by +-3 % with the ORDER in which a dot product's terms are added (316 ... 328 on the
 * 5 nm device), so two correct implementations cannot be compared by their counts unless they add in the same
 * order.  The device's order is deterministic (one partial per block, lanes / waves / blocks added in a fixed
 * tree, kmcf_cg.hip + kmcf_spmv.hip) and is a function of a few integers that the library exports
 * (kmcf_matrix_sum_plan).  This file adds in exactly that order; the result must then agree with the device
 * BIT FOR BIT -- every iterate, every r.z, the iteration count -- which is what tests/test_gpu_parity.py checks.
 *
 * Each function names the device code whose order it follows (csrc/...), and the reference lines of the
 * operation it performs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLK 256

/* wave_sum64 (csrc/kmcf_cg.hip): xor butterfly over 64 lanes, offsets 32, 16 ... 1; every lane ends with the total */
static double wave_sum64(const double *v)
{
    double a[64], b[64];
    memcpy(a, v, sizeof(a));
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ off];
        memcpy(a, b, sizeof(a));
    }
    return a[0];
}

/* the same over groups of `width` lanes (wave_sum_width, csrc/kmcf_spmv.hip): offsets < width only; returns lane 0 of
 * the group that starts at v */
static double group_sum(const double *v, int width)
{
    double a[64], b[64];
    for (int l = 0; l < width; ++l) a[l] = v[l];
    for (int off = 32; off >= 1; off >>= 1) {
        if (off >= width) continue;
        for (int l = 0; l < width; ++l) b[l] = a[l] + a[l ^ off];
        for (int l = 0; l < width; ++l) a[l] = b[l];
    }
    return a[0];
}

/* block_sum / block_sum_256: four wave sums, then (w0 + w1) + (w2 + w3) */
static double block_sum(const double *v /* BLK */)
{
    double w[4];
    for (int k = 0; k < 4; ++k) w[k] = wave_sum64(v + 64 * k);
    return (w[0] + w[1]) + (w[2] + w[3]);
}

/* reduce_partials (csrc/kmcf_cg.hip): thread t adds p[q][t], p[q][t + 256] ... for q = 0 .. 3, then block_sum */
static double reduce_partials(int narr, const double *const *p, const int *n)
{
    double v[BLK];
    for (int t = 0; t < BLK; ++t) {
        double s = 0.0;
        for (int q = 0; q < narr; ++q)
            for (int i = t; i < n[q]; i += BLK) s += p[q][i];
        v[t] = s;
    }
    return block_sum(v);
}

static double reduce1(const double *p, int n)
{
    const double *pp[1] = {p};
    return reduce_partials(1, pp, &n);
}

/* One row of the row-per-lane coded kernel (spmv_sell_kernel, csrc/kmcf_spmv.hip): the off-diagonal products in
 * the row's stored order, one add each, then the diagonal product (first entry whose column is the row). */
static double row_sum_diag_last(int i, const int *rp, const int *col, const double *val, const double *x)
{
    int dp = -1;
    for (int j = rp[i]; j < rp[i + 1]; ++j)
        if (col[j] == i) { dp = j; break; }
    double s = 0.0;
    for (int j = rp[i]; j < rp[i + 1]; ++j)
        if (j != dp) s += val[j] * x[col[j]];
    const double dg = dp >= 0 ? val[dp] : 0.0;
    s += dg * x[i];
    return s;
}

/* Ap = A p over the tiles of the row-per-lane kernel + the p.Ap partial of every block: block b walks tiles
 * c = xcd Cx + bi + k nb8 (xcd = b & 7, bi = b >> 3, nb8 = grid / 8, Cx = ceil(tiles / 8)); lane t of a tile owns
 * row first + t and adds x[row] * Ap[row] to its dot after every tile. */
static void spmv_sell_order(int n_tiles, const int *tile_first, const int *tile_rows, int grid, const int *rp,
                            const int *col, const double *val, const double *x, double *y, double *part /* grid or NULL */)
{
    const int nb8 = grid >> 3, Cx = (n_tiles + 7) >> 3;
    for (int b = 0; b < grid; ++b) {
        const int xcd = b & 7, bi = b >> 3;
        int gmax = n_tiles - xcd * Cx;
        if (gmax > Cx) gmax = Cx;
        const int nt = gmax > bi ? (gmax - bi + nb8 - 1) / nb8 : 0;
        double dot[BLK];
        for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
        for (int k = 0; k < nt; ++k) {
            const int c = xcd * Cx + bi + k * nb8;
            for (int t = 0; t < tile_rows[c]; ++t) {
                const int row = tile_first[c] + t;
                const double s = row_sum_diag_last(row, rp, col, val, x);
                y[row] = s;
                dot[t] += x[row] * s;
            }
        }
        if (part) part[b] = block_sum(dot);
    }
}

/* r.z partials as cg_xr_kernel forms them: two rows per lane and step (double2), thread (b, t) takes pairs
 * b 256 + t, + grid 256, ...; the odd last row goes to thread 0 of block 0 after its pairs. */
static void rz_partials_pairs(int n, int grid, const double *r, const double *z, double *part)
{
    const int n2 = n >> 1, T = grid * BLK;
    double *acc = (double *)calloc((size_t)T, sizeof(double));
    for (int i = 0; i < n2; ++i) {
        double *a = &acc[i % T];
        *a += r[2 * i] * z[2 * i];
        *a += r[2 * i + 1] * z[2 * i + 1];
    }
    if (n & 1) acc[0] += r[n - 1] * z[n - 1];
    for (int b = 0; b < grid; ++b) part[b] = block_sum(acc + (size_t)b * BLK);
    free(acc);
}

/*
 * Classic loop of one rank (pcg_loop, csrc/kmcf_cg.hip) on a matrix whose short rows are all computed by the
 * row-per-lane kernel.  All arrays in the library's INTERNAL row order (kmcf_matrix_sum_plan exports the CSR in
 * that order; the caller permutes the vectors with kmcf_matrix_row_order's perm).
 *   r  in: rhs, out: residual;  x  in: start guess, out: solution;  dinv: 1/diag (ignored when !precond)
 *   check_mode 1: r.z/b.b > tol^2 (dist_conjugate_gradient.cpp:217)
 *              2: first test sqrt(r.z) > tol^2, later ones r.z > tol^2 (src/iterative_solvers_gpu.cu:838-858)
 *   fixed_iters > 0: exactly that many iterations, no test
 * rz_hist (may be NULL, max_it + 2 entries): r.z as seen by the loop head of iteration k (index k - 1).
 * Returns the iterations executed; *bb_out, *rz_out as kmcf_solve_stats_t reports them.
 */
int orc_pcg_device_order(int n, const int *rp, const int *col, const double *val, double *r, double *x, const double *dinv,
                         int precond, double tol, int max_it, int fixed_iters, int check_mode, int vec_grid,
                         int sell_grid, int n_tiles, const int *tile_first, const int *tile_rows, double *bb_out,
                         double *rz_out, int *done_out, double *rz_hist)
{
    const int G = vec_grid, T = G * BLK;
    double *p = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *Ap = (double *)calloc((size_t)n + 1, sizeof(double));
    double *z = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *part_a = (double *)calloc((size_t)sell_grid + 1, sizeof(double));
    double *part_b = (double *)calloc((size_t)G, sizeof(double));
    double *part_c = (double *)calloc((size_t)G, sizeof(double));
    double *acc1 = (double *)malloc((size_t)T * sizeof(double)), *acc2 = (double *)malloc((size_t)T * sizeof(double));
    const double tol2 = tol * tol;
    /* p <- x0 ; Ap = A x0 (:178-191) */
    memcpy(p, x, (size_t)n * sizeof(double));
    spmv_sell_order(n_tiles, tile_first, tile_rows, sell_grid, rp, col, val, p, Ap, NULL);
    /* cg_init_kernel: r = b - A x0, z, partial r.z and b.b (:187, 201-212); one row per lane and step */
    for (int i = 0; i < T; ++i) acc1[i] = acc2[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double b = r[i];
        acc2[i % T] += b * b;
        const double ri = b + (-1.0) * Ap[i];
        r[i] = ri;
        z[i] = precond ? ri * dinv[i] : ri;
        acc1[i % T] += ri * z[i];
    }
    for (int b = 0; b < G; ++b) { part_b[b] = block_sum(acc1 + (size_t)b * BLK); part_c[b] = block_sum(acc2 + (size_t)b * BLK); }

    double bb = 0.0, rz_par[2] = {0.0, 0.0}, rz_last = 0.0, xa = 0.0;
    int pending = 0, iters = 0, done = 0;
    const int limit = fixed_iters > 0 ? fixed_iters : max_it;
    for (int k = 1; k <= limit; ++k) {
        const int parity = k & 1, first = k == 1;
        /* ---- cg_p_kernel: loop head (:217-227) + the x += alpha p of iteration k - 1 (:246) */
        const double rz_new = reduce1(part_b, G);
        if (first) bb = reduce1(part_c, G);
        int go = 1;
        if (fixed_iters <= 0) {
            if (check_mode == 1) go = rz_new / bb > tol2;
            else go = (first ? sqrt(rz_new) : rz_new) > tol2;
        }
        rz_last = rz_new;
        if (rz_hist) rz_hist[k - 1] = rz_new;
        if (!go) {
            if (pending) for (int i = 0; i < n; ++i) x[i] = x[i] + xa * p[i];
            pending = 0;
            done = 1;
            break;
        }
        rz_par[parity] = rz_new;
        iters += 1;
        if (first) {
            memcpy(p, z, (size_t)n * sizeof(double));                       /* :226 */
        } else {
            const double beta = rz_new / rz_par[parity ^ 1];                /* :220 */
            for (int i = 0; i < n; ++i) {
                if (pending) x[i] = x[i] + xa * p[i];
                p[i] = beta * p[i] + z[i];                                  /* :221-222 */
            }
        }
        /* ---- SpMV + p.Ap partials (:232-241) */
        spmv_sell_order(n_tiles, tile_first, tile_rows, sell_grid, rp, col, val, p, Ap, part_a);
        /* ---- cg_xr_kernel: alpha, r, z, r.z partials (:243-265); x stays pending */
        const double pAp = reduce1(part_a, sell_grid);
        const double a = rz_par[parity] / pAp, na = -a;
        for (int i = 0; i < n; ++i) {
            r[i] = r[i] + na * Ap[i];
            z[i] = precond ? r[i] * dinv[i] : r[i];
        }
        rz_partials_pairs(n, G, r, z, part_b);
        xa = a;
        pending = 1;
    }
    if (!done) {
        /* cg_tail_kernel + cg_x_kernel: the loop condition once more (:217), the pending update */
        rz_last = reduce1(part_b, G);
        if (rz_hist) rz_hist[limit] = rz_last;
        if (pending) for (int i = 0; i < n; ++i) x[i] = x[i] + xa * p[i];
    }
    *bb_out = bb;
    *rz_out = rz_last;
    *done_out = done;
    free(p); free(Ap); free(z); free(part_a); free(part_b); free(part_c); free(acc1); free(acc2);
    return iters;
}

/*
 * Single-reduction loop of one rank (pcg1_loop, csrc/kmcf_cg.hip; Chronopoulos & Gear): the same Krylov iterates as
 * the reference's recurrence in exact arithmetic, gamma = (r, z) and delta = (A z, z) reduced at one point.  The
 * library runs it for multi-rank groups (one all-reduce per iteration) and under KMCF_CG_VARIANT=cg1r.
 *   init : r = b - A x0 ; z = r .* dinv ; gamma = (r, z)                                  (cg1_init_kernel)
 *   loop : w = A z, delta = (w, z) ; stop unless gamma / b.b > tol^2 ;
 *          beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old) ;
 *          p = z + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; z = r .* dinv ; gamma' = (r, z)   (cg1_update_kernel)
 * Arguments as orc_pcg_device_order.
 */
int orc_pcg1_device_order(int n, const int *rp, const int *col, const double *val, double *r, double *x, const double *dinv,
                          int precond, double tol, int max_it, int fixed_iters, int vec_grid, int sell_grid, int n_tiles,
                          const int *tile_first, const int *tile_rows, double *bb_out, double *rz_out, int *done_out,
                          double *rz_hist)
{
    const int G = vec_grid, T = G * BLK;
    double *z = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *w = (double *)calloc((size_t)n + 1, sizeof(double));
    double *p = (double *)calloc((size_t)n + 1, sizeof(double));
    double *sv = (double *)calloc((size_t)n + 1, sizeof(double));
    double *part_a = (double *)calloc((size_t)sell_grid + 1, sizeof(double));
    double *part_b = (double *)calloc((size_t)G, sizeof(double));
    double *part_c = (double *)calloc((size_t)G, sizeof(double));
    double *acc1 = (double *)malloc((size_t)T * sizeof(double)), *acc2 = (double *)malloc((size_t)T * sizeof(double));
    const double tol2 = tol * tol;
    memcpy(z, x, (size_t)n * sizeof(double));
    spmv_sell_order(n_tiles, tile_first, tile_rows, sell_grid, rp, col, val, z, w, NULL);      /* A x0 */
    for (int i = 0; i < T; ++i) acc1[i] = acc2[i] = 0.0;
    for (int i = 0; i < n; ++i) {                                                              /* cg1_init_kernel */
        const double b = r[i];
        acc2[i % T] += b * b;
        const double ri = b + (-1.0) * w[i];
        r[i] = ri;
        z[i] = precond ? ri * dinv[i] : ri;
        acc1[i % T] += ri * z[i];
    }
    for (int b = 0; b < G; ++b) { part_b[b] = block_sum(acc1 + (size_t)b * BLK); part_c[b] = block_sum(acc2 + (size_t)b * BLK); }
    double bb = 0.0, g_par[2] = {0.0, 0.0}, a_par[2] = {0.0, 0.0}, rz_last = 0.0;
    int iters = 0, done = 0;
    const int limit = fixed_iters > 0 ? fixed_iters : max_it;
    for (int k = 1; k <= limit; ++k) {
        const int parity = k & 1, first = k == 1;
        spmv_sell_order(n_tiles, tile_first, tile_rows, sell_grid, rp, col, val, z, w, part_a);   /* w = A z, delta partials */
        const double gamma = reduce1(part_b, G);
        const double delta = reduce1(part_a, sell_grid);
        if (first) bb = reduce1(part_c, G);
        const int go = fixed_iters > 0 ? 1 : (gamma / bb > tol2);
        double beta = 0.0, alpha;
        if (first) alpha = gamma / delta;
        else {
            beta = gamma / g_par[parity ^ 1];
            alpha = gamma / (delta - beta * gamma / a_par[parity ^ 1]);
        }
        rz_last = gamma;
        if (rz_hist) rz_hist[k - 1] = gamma;
        if (!go) { done = 1; break; }
        g_par[parity] = gamma;
        a_par[parity] = alpha;
        iters += 1;
        const double na = -alpha;
        for (int i = 0; i < n; ++i) {
            const double zi = z[i], wi = w[i];
            const double pi = first ? zi : zi + beta * p[i];
            const double si = first ? wi : wi + beta * sv[i];
            p[i] = pi;
            sv[i] = si;
            x[i] = x[i] + alpha * pi;
            const double ri = r[i] + na * si;
            r[i] = ri;
            z[i] = precond ? ri * dinv[i] : ri;
        }
        rz_partials_pairs(n, G, r, z, part_b);                      /* two rows per lane and step, like cg_xr_kernel */
    }
    if (!done) {
        rz_last = reduce1(part_b, G);                                                          /* cg_tail_kernel */
        if (rz_hist) rz_hist[limit] = rz_last;
    }
    *bb_out = bb;
    *rz_out = rz_last;
    *done_out = done;
    free(z); free(w); free(p); free(sv); free(part_a); free(part_b); free(part_c); free(acc1); free(acc2);
    return iters;
}

/* The same recurrence as ONE register-resident launch (csrc/kmcf_cgr.hip: cgr_kernel): rows, row sums and the
 * recurrence as above; what differs is how the dot products are added.  Lane t of tile c holds row tile_first[c] + t
 * (ONE row per lane; lanes past the tile's rows hold 0); a tile's sum is block_sum of its 256 lanes; block b owns tiles
 * b tpb ... b tpb + tpb - 1 and adds their sums as T0, T0 + T1 or (T0 + T1) + (T2 + T3); the leader of every group of g1
 * blocks adds its group with one lane per block (wave_sum64, idle lanes 0); every block adds the group sums with one
 * lane per group (wave_sum64).  The last r.z of a loop that ends on its iteration limit is formed the same way. */
static double resident_block_sum(const double *lane_val, int n_tiles, int tpb, int b)
{
    double T[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < tpb; ++j) {
        const int c = b * tpb + j;
        if (c < n_tiles) T[j] = block_sum(lane_val + (size_t)c * BLK);
        else {                                           /* a block's idle tile: 256 lanes of 0.0 */
            double zero[BLK];
            for (int t = 0; t < BLK; ++t) zero[t] = 0.0;
            T[j] = block_sum(zero);
        }
    }
    return tpb == 1 ? T[0] : (tpb == 2 ? T[0] + T[1] : (T[0] + T[1]) + (T[2] + T[3]));
}

static double resident_sum(const double *lane_val /* n_tiles x BLK */, int n_tiles, int tpb, int g1)
{
    const int nblocks = (n_tiles + tpb - 1) / tpb;
    if (g1 == 0) {
        /* flat reduction (<= 256 blocks): lane l of wave w holds block 64 w + l (idle lanes 0); wave_sum64 each, then
         * (w0 + w1) + (w2 + w3) */
        double ws[4];
        for (int w = 0; w < 4; ++w) {
            double blk[64];
            for (int l = 0; l < 64; ++l) blk[l] = 64 * w + l < nblocks ? resident_block_sum(lane_val, n_tiles, tpb, 64 * w + l) : 0.0;
            ws[w] = wave_sum64(blk);
        }
        return (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
    const int ngroups = (nblocks + g1 - 1) / g1;
    double grp[64];
    for (int l = 0; l < 64; ++l) grp[l] = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        double blk[64];
        for (int l = 0; l < 64; ++l) blk[l] = 0.0;
        for (int l = 0; l < g1 && g * g1 + l < nblocks; ++l) blk[l] = resident_block_sum(lane_val, n_tiles, tpb, g * g1 + l);
        grp[g] = wave_sum64(blk);
    }
    return wave_sum64(grp);
}

int orc_pcg1_resident_order(int n, const int *rp, const int *col, const double *val, double *r, double *x, const double *dinv,
                            int precond, double tol, int max_it, int fixed_iters, int tpb, int g1, int n_tiles,
                            const int *tile_first, const int *tile_rows, double *bb_out, double *rz_out, int *done_out,
                            double *rz_hist)
{
    double *z = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *w = (double *)calloc((size_t)n + 1, sizeof(double));
    double *p = (double *)calloc((size_t)n + 1, sizeof(double));
    double *sv = (double *)calloc((size_t)n + 1, sizeof(double));
    double *la = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    double *lb = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    double *lc = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    const double tol2 = tol * tol;
#define FOR_LANES(...)                                                           \
    for (int c = 0; c < n_tiles; ++c)                                            \
        for (int t = 0; t < tile_rows[c]; ++t) {                                 \
            const int i = tile_first[c] + t;                                     \
            const size_t L = (size_t)c * BLK + t;                                \
            __VA_ARGS__                                                          \
        }
    for (int i = 0; i < n; ++i) w[i] = row_sum_diag_last(i, rp, col, val, x);                 /* A x0 */
    FOR_LANES({
        const double b = r[i];
        lc[L] = b * b;
        const double ri = b + (-1.0) * w[i];
        r[i] = ri;
        z[i] = ri * (precond ? dinv[i] : 1.0);
        la[L] = ri * z[i];
    })
    double bb = 0.0, g_old = 0.0, a_old = 0.0, rz_last = 0.0;
    int iters = 0, done = 0;
    const int limit = fixed_iters > 0 ? fixed_iters : max_it;
    for (int k = 1; k <= limit; ++k) {
        const int first = k == 1;
        for (int i = 0; i < n; ++i) w[i] = row_sum_diag_last(i, rp, col, val, z);
        FOR_LANES({ lb[L] = z[i] * w[i]; })
        const double gamma = resident_sum(la, n_tiles, tpb, g1);
        const double delta = resident_sum(lb, n_tiles, tpb, g1);
        if (first) bb = resident_sum(lc, n_tiles, tpb, g1);
        const int go = fixed_iters > 0 ? 1 : (gamma / bb > tol2);
        rz_last = gamma;
        if (rz_hist) rz_hist[k - 1] = gamma;
        if (!go) { done = 1; break; }
        double beta = 0.0, alpha;
        if (first) alpha = gamma / delta;
        else {
            beta = gamma / g_old;
            alpha = gamma / (delta - beta * gamma / a_old);
        }
        g_old = gamma;
        a_old = alpha;
        iters += 1;
        const double na = -alpha;
        FOR_LANES({
            const double zi = z[i], wi = w[i];
            const double pi = first ? zi : zi + beta * p[i];
            const double si = first ? wi : wi + beta * sv[i];
            p[i] = pi;
            sv[i] = si;
            x[i] = x[i] + alpha * pi;
            const double ri = r[i] + na * si;
            r[i] = ri;
            z[i] = ri * (precond ? dinv[i] : 1.0);
            la[L] = ri * z[i];
        })
    }
    if (!done) {
        rz_last = resident_sum(la, n_tiles, tpb, g1);
        if (rz_hist) rz_hist[limit] = rz_last;
    }
#undef FOR_LANES
    *bb_out = bb;
    *rz_out = rz_last;
    *done_out = done;
    free(z); free(w); free(p); free(sv); free(la); free(lb); free(lc);
    return iters;
}

/* The reference's recurrence (orc_pcg_device_order above: :217-266, p.Ap and r.z reduced separately) as ONE register-resident
 * launch (csrc/kmcf_cgr.hip, classic = 1): rows and row sums as the row-per-lane kernel's, every dot product over the
 * resident tree (resident_sum), x += alpha p applied where alpha is formed (the kernels apply it one kernel later: the same
 * operation on the same operands).  Arguments as orc_pcg1_resident_order. */
int orc_pcg_resident_order(int n, const int *rp, const int *col, const double *val, double *r, double *x, const double *dinv,
                           int precond, double tol, int max_it, int fixed_iters, int tpb, int g1, int n_tiles,
                           const int *tile_first, const int *tile_rows, double *bb_out, double *rz_out, int *done_out,
                           double *rz_hist)
{
    double *z = (double *)malloc(((size_t)n + 1) * sizeof(double));
    double *w = (double *)calloc((size_t)n + 1, sizeof(double));
    double *p = (double *)calloc((size_t)n + 1, sizeof(double));
    double *la = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    double *lc = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    const double tol2 = tol * tol;
#define FOR_LANES(...)                                                           \
    for (int c = 0; c < n_tiles; ++c)                                            \
        for (int t = 0; t < tile_rows[c]; ++t) {                                 \
            const int i = tile_first[c] + t;                                     \
            const size_t L = (size_t)c * BLK + t;                                \
            __VA_ARGS__                                                          \
        }
    for (int i = 0; i < n; ++i) w[i] = row_sum_diag_last(i, rp, col, val, x);                 /* A x0 */
    FOR_LANES({
        const double b = r[i];
        lc[L] = b * b;
        const double ri = b + (-1.0) * w[i];
        r[i] = ri;
        z[i] = ri * (precond ? dinv[i] : 1.0);
        la[L] = ri * z[i];
    })
    double rz = resident_sum(la, n_tiles, tpb, g1);
    const double bb = resident_sum(lc, n_tiles, tpb, g1);
    double rz_prev = 0.0, rz_last = 0.0;
    int iters = 0, done = 0;
    const int limit = fixed_iters > 0 ? fixed_iters : max_it;
    for (int k = 1; k <= limit; ++k) {
        const int first = k == 1;
        const int go = fixed_iters > 0 ? 1 : (rz / bb > tol2);
        rz_last = rz;
        if (rz_hist) rz_hist[k - 1] = rz;
        if (!go) { done = 1; break; }
        iters += 1;
        if (first) { for (int i = 0; i < n; ++i) p[i] = z[i]; }
        else {
            const double beta = rz / rz_prev;
            for (int i = 0; i < n; ++i) p[i] = beta * p[i] + z[i];
        }
        for (int i = 0; i < n; ++i) w[i] = row_sum_diag_last(i, rp, col, val, p);
        FOR_LANES({ la[L] = p[i] * w[i]; })
        const double pAp = resident_sum(la, n_tiles, tpb, g1);
        const double a = rz / pAp, na = -a;
        FOR_LANES({
            x[i] = x[i] + a * p[i];
            const double ri = r[i] + na * w[i];
            r[i] = ri;
            z[i] = ri * (precond ? dinv[i] : 1.0);
            la[L] = ri * z[i];
        })
        rz_prev = rz;
        rz = resident_sum(la, n_tiles, tpb, g1);
    }
    if (!done) {
        rz_last = rz;
        if (rz_hist) rz_hist[limit] = rz;
    }
#undef FOR_LANES
    *bb_out = bb;
    *rz_out = rz_last;
    *done_out = done;
    free(z); free(w); free(p); free(la); free(lc);
    return iters;
}

/* Pieces of the register-resident solve for groups of ranks (csrc/kmcf_cgr.hip with nranks > 1), driven rank by rank
 * from kmcf_oracle.py: pcg_resident_ranks.  Row sums over a rank's rows with xv = [own | halo] (every row, boundary
 * rows included, is added like the row-per-lane kernel adds it); a rank's sum of one value per row over its tiles,
 * blocks and reduction tree; the butterfly over the ranks' sums (one lane per rank). */
void orc_resident_spmv(int n, const int *rp, const int *col, const double *val, const double *xv, double *y)
{
    for (int i = 0; i < n; ++i) y[i] = row_sum_diag_last(i, rp, col, val, xv);
}

double orc_resident_rank_sum(int n_tiles, const int *tile_first, const int *tile_rows, int tpb, int g1, const double *row_val)
{
    double *lv = (double *)calloc((size_t)n_tiles * BLK, sizeof(double));
    for (int c = 0; c < n_tiles; ++c)
        for (int t = 0; t < tile_rows[c]; ++t) lv[(size_t)c * BLK + t] = row_val[tile_first[c] + t];
    const double r = resident_sum(lv, n_tiles, tpb, g1);
    free(lv);
    return r;
}

double orc_wave_sum_n(const double *v, int n)
{
    double a[64];
    for (int l = 0; l < 64; ++l) a[l] = l < n ? v[l] : 0.0;
    return wave_sum64(a);
}

/* y = A x in the row-per-lane kernel's order (for SpMV parity at bit level) */
void orc_spmv_device_order(int n_tiles, const int *tile_first, const int *tile_rows, int sell_grid, const int *rp,
                           const int *col, const double *val, const double *x, double *y)
{
    spmv_sell_order(n_tiles, tile_first, tile_rows, sell_grid, rp, col, val, x, y, NULL);
}

/* ====================================================================================================== */
/* Building blocks for groups of ranks and for the split T operator: ONE rank's kernels each, in the device's   */
/* order; the loop over iterations and ranks (halo exchange, rank-ordered sums of the all-reduce) is driven   */
/* from kmcf_oracle.py (pcg_device_order_ranks).                                                               */
/* ====================================================================================================== */

typedef struct {
    int n, n_short, n_halo;                       /* rows of this rank, rows before the long ones, halo slots        */
    const int *rp, *col;                          /* CSR as stored: internal row order, halo columns = n + slot      */
    const double *val;
    int vec_grid;                                 /* blocks of the vector kernels                                    */
    int sell_grid, n_tiles;                       /* row-per-lane kernel: blocks, tiles                              */
    const int *tile_first, *tile_rows;
    int boundary_grid, boundary_lpr, n_boundary;  /* separate pass over the short rows that touch the halo           */
    const int *boundary_rows;
    int n_long_items;                             /* long-row kernel: chunks (row, first entry, end entry, first chunk of the row) */
    const int *long_items;
    int sub_grid, sub_n;                          /* tunnel sub-block operator: blocks, local sub rows                */
    const int *sub_rows;                          /* internal row of every local sub row                             */
    const int *sub_rp, *sub_col;                  /* its CSR over the gathered sub-vector (ascending columns)        */
    const double *sub_val;
    int sub_dense;                                /* 1: the block is applied from dense symmetric 64 x 64 tiles       */
    int sub_n_glob, sub_strip;                    /* points of all ranks (= this rank's: one rank only); tiles per strip */
    const double *sub_full;                       /* sub_n_glob x sub_n_glob, row-major, zeros where there is no entry */
    int sub_row0;                                 /* sub_dense == 2 (tiles spread over a rank group): first point of this rank */
    const double *sub_total;                      /* ... and the sums of ALL points, the ranks' partials added in rank order
                                                   * (orc_sub_tiles_part per rank, then the caller)                    */
} orc_dev_plan;

/* One rank's partial sums of the tile walk when the strips of the upper block triangle are dealt to P ranks (each strip
 * of the global list -- block rows ascending, J0 = I, I + SL, ... -- to the rank holding the fewest tiles so far, the
 * lowest such rank; csrc/kmcf_tstate.hip: symm_setup, sub_symm_reduce_kernel with tile_local / ypart): for block row B,
 * wave w adds the column parts of THIS rank's tiles (K, B) among K = w, w + 4, ... < B and the row parts of this rank's
 * strips of block row B, the k-th of them by wave k mod 4; joined as ((c0 + c1) + (c2 + c3)) + ((r0 + r1) + (r2 + r3))
 * into ypart[64 B + l] (all 64 nb of them). */
void orc_sub_tiles_part(int nt, int SL, const double *F, const double *xsub, int q, int P, double *ypart)
{
    const int nb = (nt + 63) / 64;
    int *gfirst = (int *)calloc((size_t)nb + 1, sizeof(int));       /* global index of the first strip of every block row */
    for (int I = 0; I < nb; ++I) gfirst[I + 1] = gfirst[I] + (nb - I + SL - 1) / SL;
    int *owner = (int *)calloc((size_t)gfirst[nb] + 1, sizeof(int));
    {
        long long *held = (long long *)calloc((size_t)P, sizeof(long long));
        int s = 0;
        for (int I = 0; I < nb; ++I)
            for (int J = I; J < nb; J += SL, ++s) {
                int o = 0;
                for (int r = 1; r < P; ++r)
                    if (held[r] < held[o]) o = r;
                owner[s] = o;
                held[o] += J + SL < nb ? SL : nb - J;
            }
        free(held);
    }
#define FV(i, j) (((i) < nt && (j) < nt) ? F[(size_t)(i) * nt + (j)] : 0.0)
#define XS(j) ((j) < nt ? xsub[j] : 0.0)
    for (int B = 0; B < nb; ++B) {
        double cw[4][64], rw[4][64];
        memset(cw, 0, sizeof(cw)); memset(rw, 0, sizeof(rw));
        for (int wv = 0; wv < 4; ++wv)
            for (int K = wv; K < B; K += 4) {
                if (owner[gfirst[K] + (B - K) / SL] != q) continue; /* tile (K, B) lies in another rank's strip */
                for (int c = 0; c < 64; ++c) {
                    double ca = 0.0;
                    for (int r = 0; r < 64; ++r) ca += FV(64 * K + r, 64 * B + c) * XS(64 * K + r);
                    cw[wv][c] += ca;
                }
            }
        int k = 0, sidx = 0;
        for (int J0 = B; J0 < nb; J0 += SL, ++sidx) {
            if (owner[gfirst[B] + sidx] != q) continue;
            const int wv = k++ & 3, J1 = J0 + SL < nb ? J0 + SL : nb;
            for (int r = 0; r < 64; ++r) {
                double ra = 0.0;
                for (int J = J0; J < J1; ++J)
                    for (int c = 0; c < 64; ++c) ra += FV(64 * B + r, 64 * J + c) * XS(64 * J + c);
                rw[wv][r] += ra;
            }
        }
        for (int l = 0; l < 64; ++l)
            ypart[64 * B + l] = ((cw[0][l] + cw[1][l]) + (cw[2][l] + cw[3][l])) + ((rw[0][l] + rw[1][l]) + (rw[2][l] + rw[3][l]));
    }
#undef FV
#undef XS
    free(gfirst); free(owner);
}

/* One distributed SpMV of one rank, y = A x with x = [own | halo] (+ S x_sub on the sub rows), every row and every
 * p.Ap partial in the order of the kernel that computes it on the device (kmcf_spmv_device, csrc/kmcf_spmv.hip):
 *   short rows without halo columns : spmv_sell_kernel (diagonal last; lanes of a tile; SKIP_BOUNDARY)
 *   short rows with halo columns    : spmv_vec_kernel<LPR, ROW_LIST> (LPR strided partial sums + butterfly)
 *   long rows                       : spmv_long_kernel (256-strided chunk sums, chunks added in order)
 *   sub rows                        : sub_spmv_kernel (lane = column mod 64, groups ascending, butterfly; y += )
 * *pap_local = reduce_partials over the four partial arrays (cg_finalize_kernel / the consumer's reduce). */
void orc_dev_spmv(const orc_dev_plan *pl, const double *x, const double *xsub, double *y, int with_dot, double *pap_local)
{
    const int n = pl->n;
    double *pa = (double *)calloc((size_t)pl->sell_grid + 1, sizeof(double));
    double *pb = (double *)calloc((size_t)pl->boundary_grid + 1, sizeof(double));
    double *pc = (double *)calloc(2, sizeof(double));
    double *pd = (double *)calloc((size_t)pl->sub_grid + 1, sizeof(double));
    /* ---- interior: row-per-lane kernel, rows that touch the halo skipped when this rank has a halo */
    {
        const int grid = pl->sell_grid, nb8 = grid >> 3, Cx = (pl->n_tiles + 7) >> 3;
        for (int b = 0; b < grid; ++b) {
            const int xcd = b & 7, bi = b >> 3;
            int gmax = pl->n_tiles - xcd * Cx;
            if (gmax > Cx) gmax = Cx;
            const int nt = gmax > bi ? (gmax - bi + nb8 - 1) / nb8 : 0;
            double dot[BLK];
            for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
            for (int k = 0; k < nt; ++k) {
                const int c = xcd * Cx + bi + k * nb8;
                for (int t = 0; t < pl->tile_rows[c]; ++t) {
                    const int row = pl->tile_first[c] + t;
                    if (pl->n_halo > 0) {
                        int isb = 0;
                        for (int j = pl->rp[row]; j < pl->rp[row + 1]; ++j) isb |= pl->col[j] >= n;
                        if (isb) continue;
                    }
                    const double sv = row_sum_diag_last(row, pl->rp, pl->col, pl->val, x);
                    y[row] = sv;
                    dot[t] += x[row] * sv;
                }
            }
            pa[b] = block_sum(dot);
        }
    }
    /* ---- boundary rows: spmv_vec_kernel<LPR, DOT, false, ROW_LIST> */
    if (pl->n_halo > 0 && pl->n_boundary > 0) {
        const int LPR = pl->boundary_lpr, RPB = BLK / LPR, grid = pl->boundary_grid;
        const int G = (pl->n_boundary + RPB - 1) / RPB, Gx = (G + 7) >> 3, nb8 = grid >> 3;
        for (int b = 0; b < grid; ++b) {
            const int xcd = b & 7, bi = b >> 3;
            double dot[BLK];
            for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
            for (int g = bi; g < Gx; g += nb8) {
                const int grp = xcd * Gx + g;
                for (int rib = 0; rib < RPB; ++rib) {
                    const int row = grp * RPB + rib;
                    if (!(grp < G && row < pl->n_boundary)) continue;
                    const int r = pl->boundary_rows[row];
                    double lane[64];
                    for (int l = 0; l < LPR; ++l) {
                        double sv = 0.0;
                        for (int j = pl->rp[r] + l; j < pl->rp[r + 1]; j += LPR) sv += pl->val[j] * x[pl->col[j]];
                        lane[l] = sv;
                    }
                    const double sv = group_sum(lane, LPR);
                    y[r] = sv;
                    dot[rib * LPR] += x[r] * sv;
                }
            }
            pb[b] = block_sum(dot);
        }
    }
    /* ---- long rows: spmv_long_kernel */
    if (pl->n_long_items > 0) {
        const int ni = pl->n_long_items;
        double *lpart = (double *)calloc((size_t)ni, sizeof(double));
        for (int q = 0; q < ni; ++q) {
            const int *it = pl->long_items + 4 * q;
            double th[BLK];
            for (int t = 0; t < BLK; ++t) {
                double sv = 0.0;
                for (int j = it[1] + t; j < it[2]; j += BLK) sv += pl->val[j] * x[pl->col[j]];
                th[t] = sv;
            }
            lpart[q] = block_sum(th);
        }
        double dot[BLK];
        for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
        for (int q = 0; q < ni; ++q) {
            const int *a = pl->long_items + 4 * q;
            if (a[3] != q) continue;
            double tsum = 0.0;
            for (int c = q; c < ni && pl->long_items[4 * c] == a[0]; ++c) tsum += lpart[c];
            y[a[0]] = tsum;
            dot[q % BLK] += x[a[0]] * tsum;
        }
        pc[0] = block_sum(dot);
        free(lpart);
    }
    /* ---- sub-block, dense symmetric storage: sub_symm_kernel + sub_symm_reduce_kernel (csrc/kmcf_tstate.hip).  Upper
     * tiles (I, J >= I) in strips of sub_strip tiles of one block row: lane r adds row r's products column by column and
     * runs on through the strip's tiles; lane c adds column c's products row by row (not for diagonal tiles).  Block row
     * B: wave w adds the column parts of tiles (K, B), K = w, w + 4, ... < B and the strips first + w, + 4, ...; joined
     * as ((c0 + c1) + (c2 + c3)) + ((r0 + r1) + (r2 + r3)); one p.Ap partial per block row. */
    if (pl->sub_grid > 0 && pl->sub_n > 0 && pl->sub_dense == 2) {
        /* tiles spread over the rank group: sub_combine_kernel -- a lane per own point, one p.Ap partial per 256 of them */
        for (int b = 0; b < pl->sub_grid; ++b) {
            double dot[BLK];
            for (int t = 0; t < BLK; ++t) {
                const int sr = b * BLK + t;
                dot[t] = 0.0;
                if (sr >= pl->sub_n) continue;
                const double a = pl->sub_total[pl->sub_row0 + sr];
                const int r = pl->sub_rows[sr];
                y[r] += a;
                dot[t] = x[r] * a;
            }
            pd[b] = block_sum(dot);
        }
    } else
    if (pl->sub_grid > 0 && pl->sub_n > 0 && pl->sub_dense) {
        const int nt = pl->sub_n_glob, nb = (nt + 63) / 64, SL = pl->sub_strip;
        const double *F = pl->sub_full;
#define FV(i, j) (((i) < nt && (j) < nt) ? F[(size_t)(i) * nt + (j)] : 0.0)
#define XS(j) ((j) < nt ? xsub[j] : 0.0)
        for (int B = 0; B < nb; ++B) {
            double cw[4][64], rw[4][64];
            memset(cw, 0, sizeof(cw)); memset(rw, 0, sizeof(rw));
            for (int wv = 0; wv < 4; ++wv) {
                for (int K = wv; K < B; K += 4)                      /* column sums of tile (K, B): lane c over rows r */
                    for (int c = 0; c < 64; ++c) {
                        double ca = 0.0;
                        for (int r = 0; r < 64; ++r) ca += FV(64 * K + r, 64 * B + c) * XS(64 * K + r);
                        cw[wv][c] += ca;
                    }
                int sidx = 0;
                for (int J0 = B; J0 < nb; J0 += SL, ++sidx) {        /* strips of block row B */
                    if ((sidx & 3) != wv) continue;
                    const int J1 = J0 + SL < nb ? J0 + SL : nb;
                    for (int r = 0; r < 64; ++r) {
                        double ra = 0.0;
                        for (int J = J0; J < J1; ++J)
                            for (int c = 0; c < 64; ++c) ra += FV(64 * B + r, 64 * J + c) * XS(64 * J + c);
                        rw[wv][r] += ra;
                    }
                }
            }
            double dot[BLK];
            for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
            for (int l = 0; l < 64; ++l) {
                const int i = 64 * B + l;
                if (i >= nt) continue;
                const double a = ((cw[0][l] + cw[1][l]) + (cw[2][l] + cw[3][l])) + ((rw[0][l] + rw[1][l]) + (rw[2][l] + rw[3][l]));
                const int r = pl->sub_rows[i];
                y[r] += a;
                dot[l] = x[r] * a;
            }
            pd[B] = block_sum(dot);
        }
#undef FV
#undef XS
    } else
    /* ---- sub-block: sub_spmv_kernel (a wave per local sub row; rows s = b 4 + w, + grid 4, ...) */
    if (pl->sub_grid > 0 && pl->sub_n > 0) {
        const int grid = pl->sub_grid;
        for (int b = 0; b < grid; ++b) {
            double dot[BLK];
            for (int t = 0; t < BLK; ++t) dot[t] = 0.0;
            for (int wv = 0; wv < 4; ++wv)
                for (int sr = b * 4 + wv; sr < pl->sub_n; sr += grid * 4) {
                    double lane[64];
                    for (int l = 0; l < 64; ++l) lane[l] = 0.0;
                    for (int j = pl->sub_rp[sr]; j < pl->sub_rp[sr + 1]; ++j) {     /* ascending columns: per lane, ascending groups */
                        const int cj = pl->sub_col[j];
                        lane[cj & 63] += pl->sub_val[j] * xsub[cj];
                    }
                    const double acc = wave_sum64(lane);
                    const int r = pl->sub_rows[sr];
                    y[r] += acc;
                    dot[wv * 64] += x[r] * acc;
                }
            pd[b] = block_sum(dot);
        }
    }
    if (with_dot && pap_local) {
        const double *pp[4] = {pa, pb, pc, pd};
        const int nn[4] = {pl->sell_grid, pl->n_halo > 0 && pl->n_boundary > 0 ? pl->boundary_grid : 0, pl->n_long_items > 0 ? 1 : 0,
                           pl->sub_grid};
        *pap_local = reduce_partials(4, pp, nn);
    }
    free(pa); free(pb); free(pc); free(pd);
}

/* cg_init_kernel / cg1_init_kernel of one rank: r = b - A x0, z = r .* dinv; local sums of r.z and b.b */
void orc_dev_init(int n, int vec_grid, double *r, const double *Ap, const double *dinv, int precond, double *z, double *rz_local,
                  double *bb_local)
{
    const int G = vec_grid, T = G * BLK;
    double *acc1 = (double *)calloc((size_t)T, sizeof(double)), *acc2 = (double *)calloc((size_t)T, sizeof(double));
    double *p1 = (double *)calloc((size_t)G, sizeof(double)), *p2 = (double *)calloc((size_t)G, sizeof(double));
    for (int i = 0; i < n; ++i) {
        const double b = r[i];
        acc2[i % T] += b * b;
        const double ri = b + (-1.0) * Ap[i];
        r[i] = ri;
        z[i] = precond ? ri * dinv[i] : ri;
        acc1[i % T] += ri * z[i];
    }
    for (int b = 0; b < G; ++b) { p1[b] = block_sum(acc1 + (size_t)b * BLK); p2[b] = block_sum(acc2 + (size_t)b * BLK); }
    *rz_local = reduce1(p1, G);
    *bb_local = reduce1(p2, G);
    free(acc1); free(acc2); free(p1); free(p2);
}

/* cg_p_kernel of one rank (after the stopping test): x += xa p (pending), p = beta p + z, or p = z in the first iteration */
void orc_dev_p(int n, double *p, const double *z, double *x, double beta, int first, int pending, double xa)
{
    for (int i = 0; i < n; ++i) {
        if (first) { p[i] = z[i]; continue; }
        if (pending) x[i] = x[i] + xa * p[i];
        p[i] = beta * p[i] + z[i];
    }
}

void orc_dev_x(int n, double *x, const double *p, double xa)
{
    for (int i = 0; i < n; ++i) x[i] = x[i] + xa * p[i];
}

/* cg_xr_kernel of one rank: r -= alpha Ap, z = r .* dinv, local r.z (pairs of rows per lane) */
void orc_dev_xr(int n, int vec_grid, double *r, const double *Ap, const double *dinv, int precond, double alpha, double *z,
                double *rz_local)
{
    const double na = -alpha;
    double *part = (double *)calloc((size_t)vec_grid, sizeof(double));
    for (int i = 0; i < n; ++i) {
        r[i] = r[i] + na * Ap[i];
        z[i] = precond ? r[i] * dinv[i] : r[i];
    }
    rz_partials_pairs(n, vec_grid, r, z, part);
    *rz_local = reduce1(part, vec_grid);
    free(part);
}

/* cg1_update_kernel of one rank */
void orc_dev_cg1_update(int n, int vec_grid, double *x, double *r, double *p, double *sv, double *z, const double *w,
                        const double *dinv, int precond, double alpha, double beta, int first, double *rz_local)
{
    const int G = vec_grid;
    const double na = -alpha;
    double *part = (double *)calloc((size_t)G, sizeof(double));
    for (int i = 0; i < n; ++i) {
        const double zi = z[i], wi = w[i];
        const double pi = first ? zi : zi + beta * p[i];
        const double si = first ? wi : wi + beta * sv[i];
        p[i] = pi;
        sv[i] = si;
        x[i] = x[i] + alpha * pi;
        const double ri = r[i] + na * si;
        r[i] = ri;
        z[i] = precond ? ri * dinv[i] : ri;
    }
    rz_partials_pairs(n, G, r, z, part);                            /* two rows per lane and step, like cg_xr_kernel */
    *rz_local = reduce1(part, G);
    free(part);
}
