// Distributed CSR matrix: host-side planning (neighbour discovery, halo lists,
// compact-halo column remap) and device upload.
//
// Reference behaviour mirrored (dist_iterative/dist_matrix.cpp):
//   neighbours        cyclic order starting at own rank, only ranks whose column
//                     range holds a nonzero of our rows            (:237-278)
//   cols_per_neighbour sorted unique block-local columns = what we receive (:451-487)
//   rows_per_neighbour local rows with a nonzero in the block = what we send;
//                     valid because the matrix is structurally symmetric (:3, :418-448)
// MI355X layout (differs from the reference on purpose):
//   ONE CSR per rank whose column ids are remapped to [0,n_loc) (own block) and
//   n_loc + compact halo slot (neighbour blocks), so the SpMV is a single kernel
//   over a single x vector [p_local | p_halo] and received halos need no unpack
//   (the reference keeps one CSR + one full-length dense vector per neighbour and
//   scatters every received buffer, dist_vector.cpp:3-42, utils_cg.cu:52-98).
#include <algorithm>
#include <cstring>

#include "kmcf_internal.hpp"

namespace {

template <typename T>
int dev_upload(T **d, const std::vector<T> &h)
{
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(d), bytes));
    if (!h.empty()) KMCF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return KMCF_OK;
}

template <typename T>
int dev_alloc(T **d, size_t n)
{
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(d), std::max<size_t>(n, 1) * sizeof(T)));
    KMCF_HIP(hipMemset(*d, 0, std::max<size_t>(n, 1) * sizeof(T)));
    return KMCF_OK;
}

}  // namespace

int kmcf_matrix_build(kmcf_comm *c, int matrix_size, const int *counts, const int *displs,
                      const int *h_row_ptr, const int *h_col_global, const double *h_val,
                      const int *h_perm, kmcf_matrix **out)
{
    KMCF_CHECK(c && counts && displs && h_row_ptr && out, KMCF_ERR_ARG, "kmcf_matrix_build: null argument");
    const int P = c->nranks, rank = c->rank;
    {
        int64_t tot = 0;
        for (int q = 0; q < P; ++q) {
            KMCF_CHECK(counts[q] >= 0 && displs[q] == (int)tot, KMCF_ERR_ARG,
                       "kmcf_matrix_build: counts/displs are not a contiguous partition at rank %d", q);
            tot += counts[q];
        }
        KMCF_CHECK(tot == matrix_size, KMCF_ERR_ARG, "kmcf_matrix_build: counts sum %lld != matrix_size %d",
                   (long long)tot, matrix_size);
    }
    kmcf_matrix *m = new kmcf_matrix();
    // every error return below frees what was built so far (host state and device buffers)
    struct build_guard {
        kmcf_matrix *m;
        ~build_guard() { if (m) kmcf_matrix_destroy(m); }
    } guard{m};
    m->comm = c;
    m->matrix_size = matrix_size;
    m->counts.assign(counts, counts + P);
    m->displs.assign(displs, displs + P);
    m->n_loc = counts[rank];
    m->row0 = displs[rank];
    const int n_loc = m->n_loc;
    KMCF_CHECK(h_row_ptr[0] == 0, KMCF_ERR_ARG, "kmcf_matrix_build: row_ptr[0] != 0");
    m->nnz = h_row_ptr[n_loc];
    const int64_t nnz = m->nnz;
    KMCF_CHECK(nnz == 0 || h_col_global, KMCF_ERR_ARG, "kmcf_matrix_build: null column array");

    // owner of a global column (ranks with zero rows are skipped by upper_bound on displs)
    auto owner = [&](int col) {
        int q = int(std::upper_bound(displs, displs + P, col) - displs) - 1;
        while (q > 0 && counts[q] == 0) --q;  // displs repeats for empty ranks
        return q;
    };

    // pass 1: per-owner flags of referenced block-local columns, nnz per owner
    std::vector<int64_t> nnz_owner(P, 0);
    std::vector<std::vector<unsigned char>> col_flag(P);
    std::vector<int> own(nnz);
    for (int r = 0; r < n_loc; ++r) {
        KMCF_CHECK(h_row_ptr[r + 1] >= h_row_ptr[r], KMCF_ERR_ARG, "kmcf_matrix_build: row_ptr not monotone at row %d", r);
        for (int j = h_row_ptr[r]; j < h_row_ptr[r + 1]; ++j) {
            int cg = h_col_global[j];
            KMCF_CHECK(cg >= 0 && cg < matrix_size, KMCF_ERR_ARG, "kmcf_matrix_build: column %d out of range at nnz %d", cg, j);
            int q = owner(cg);
            own[j] = q;
            if (col_flag[q].empty()) col_flag[q].assign((size_t)counts[q], 0);
            col_flag[q][cg - displs[q]] = 1;
            nnz_owner[q]++;
        }
    }
    // neighbours in cyclic order from self; self is always block 0 (the reference's
    // initialize_sparsity_K always has the diagonal in the pattern)
    m->neighbours.clear();
    for (int k = 0; k < P; ++k) {
        int q = (rank + k) % P;
        if (k == 0 || nnz_owner[q] > 0) {
            m->neighbours.push_back(q);
            m->nnz_per_neighbour.push_back((int)nnz_owner[q]);
        }
    }
    m->number_of_neighbours = (int)m->neighbours.size();
    const int nnb = m->number_of_neighbours;
    std::vector<int> nb_index(P, -1);
    for (int k = 0; k < nnb; ++k) nb_index[m->neighbours[k]] = k;

    // receive lists (sorted unique block-local columns) and compact slot maps
    m->cols_per_neighbour.assign(nnb, {});
    m->rows_per_neighbour.assign(nnb, {});
    m->halo_offset.assign(nnb, 0);
    m->send_offset.assign(nnb, 0);
    std::vector<std::vector<int>> slot_of(nnb);  // block-local column -> slot in block
    int halo = 0;
    for (int k = 0; k < nnb; ++k) {
        int q = m->neighbours[k];
        auto &flag = col_flag[q];
        auto &cols = m->cols_per_neighbour[k];
        for (int cidx = 0; cidx < (int)flag.size(); ++cidx)
            if (flag[cidx]) cols.push_back(cidx);
        if (k >= 1) {
            m->halo_offset[k] = halo;
            slot_of[k].assign((size_t)counts[q], -1);
            for (int s = 0; s < (int)cols.size(); ++s) slot_of[k][cols[s]] = s;
            halo += (int)cols.size();
        }
    }
    m->n_halo = halo;

    // remapped CSR + send lists + boundary rows
    std::vector<int> col_local(nnz);
    std::vector<unsigned char> is_boundary((size_t)n_loc, 0);
    std::vector<int> last_row_seen(nnb, -1);
    for (int r = 0; r < n_loc; ++r) {
        for (int j = h_row_ptr[r]; j < h_row_ptr[r + 1]; ++j) {
            int q = own[j], k = nb_index[q];
            int cl = h_col_global[j] - displs[q];
            if (k == 0) {
                col_local[j] = cl;
            } else {
                col_local[j] = n_loc + m->halo_offset[k] + slot_of[k][cl];
                is_boundary[r] = 1;
            }
            if (last_row_seen[k] != r) {
                last_row_seen[k] = r;
                m->rows_per_neighbour[k].push_back(r);
            }
        }
    }
    std::vector<int> boundary_rows;
    for (int r = 0; r < n_loc; ++r)
        if (is_boundary[r]) boundary_rows.push_back(r);
    m->n_boundary_rows = (int)boundary_rows.size();
    std::vector<int> send_idx;
    int so = 0;
    for (int k = 1; k < nnb; ++k) {
        m->send_offset[k] = so;
        send_idx.insert(send_idx.end(), m->rows_per_neighbour[k].begin(), m->rows_per_neighbour[k].end());
        so += (int)m->rows_per_neighbour[k].size();
    }
    m->n_send = so;
    std::vector<int> halo_gid((size_t)m->n_halo);
    for (int k = 1; k < nnb; ++k)
        for (int s = 0; s < (int)m->cols_per_neighbour[k].size(); ++s)
            halo_gid[m->halo_offset[k] + s] = displs[m->neighbours[k]] + m->cols_per_neighbour[k][s];

    // ---- internal row order: internal row i = caller row perm[i] -----------------------------
    // the caller's locality order (if any), then long rows moved behind all others (kmcf_internal.hpp)
    std::vector<int> rp(h_row_ptr, h_row_ptr + n_loc + 1);
    m->h_row_ptr_user = rp;
    std::vector<double> val_int;
    if (h_val) val_int.assign(h_val, h_val + nnz);
    std::vector<int> perm_eff;
    if (h_perm && n_loc > 0) perm_eff.assign(h_perm, h_perm + n_loc);
    int long_thr = 384;
    if (const char *e = getenv("KMCF_LONG_ROW")) long_thr = atoi(e);
    int n_long = 0;
    if (long_thr > 0)
        for (int r = 0; r < n_loc; ++r) n_long += (rp[r + 1] - rp[r] > long_thr);
    if (n_long > 0) {
        if (perm_eff.empty()) { perm_eff.resize((size_t)n_loc); for (int i = 0; i < n_loc; ++i) perm_eff[i] = i; }
        for (int i = 0; i < n_loc; ++i)
            if (perm_eff[i] < 0 || perm_eff[i] >= n_loc) {
                kmcf_set_error("kmcf_matrix_build: perm is not a permutation of the local rows (entry %d = %d)", i, perm_eff[i]);
                return KMCF_ERR_ARG;
            }
        std::stable_partition(perm_eff.begin(), perm_eff.end(), [&](int r) { return rp[r + 1] - rp[r] <= long_thr; });
    }
    m->n_short = n_loc - n_long;
    // rows of a row-per-lane tile sorted by length (kmcf_spmv.hip: lane t of a tile owns row r0 + t)
    if (m->n_short >= 2) {
        if (perm_eff.empty()) { perm_eff.resize((size_t)n_loc); for (int i = 0; i < n_loc; ++i) perm_eff[i] = i; }
        bool valid = true;
        {
            std::vector<unsigned char> seen((size_t)n_loc, 0);
            for (int i = 0; i < n_loc && valid; ++i) {
                const int r = perm_eff[i];
                valid = r >= 0 && r < n_loc && !seen[r];
                if (valid) seen[r] = 1;
            }
        }
        if (valid) kmcf_sell_refine_order(m->n_short, n_loc + m->n_halo, rp.data(), col_local.data(), perm_eff, m->h_sell_cuts);
    }
    if (!perm_eff.empty()) {
        const int *hp = perm_eff.data();
        std::vector<int> inv((size_t)n_loc, -1);
        for (int i = 0; i < n_loc; ++i) {
            const int r = hp[i];
            if (r < 0 || r >= n_loc || inv[r] != -1) {
                kmcf_set_error("kmcf_matrix_build: perm is not a permutation of the local rows (entry %d = %d)", i, r);
                return KMCF_ERR_ARG;
            }
            inv[r] = i;
        }
        m->h_perm = perm_eff;
        std::vector<int> rp_new((size_t)n_loc + 1, 0), col_new((size_t)nnz);
        std::vector<double> val_new(h_val ? (size_t)nnz : 0);
        std::vector<unsigned char> isb_new((size_t)n_loc, 0);
        for (int i = 0; i < n_loc; ++i) rp_new[i + 1] = rp_new[i] + (rp[hp[i] + 1] - rp[hp[i]]);
        for (int i = 0; i < n_loc; ++i) {
            const int r = hp[i];
            int dst = rp_new[i];
            for (int j = rp[r]; j < rp[r + 1]; ++j, ++dst) {
                const int cl = col_local[j];
                col_new[dst] = cl < n_loc ? inv[cl] : cl;     // own block: internal index; halo slots unchanged
                if (h_val) val_new[dst] = h_val[j];
            }
            isb_new[i] = is_boundary[r];
        }
        rp.swap(rp_new);
        col_local.swap(col_new);
        val_int.swap(val_new);
        is_boundary.swap(isb_new);
        for (int &s : send_idx) s = inv[s];                   // gather positions; packed order = protocol order
    }
    // boundary pass = short rows that reference the halo (long rows wait for the halo in their own kernel)
    boundary_rows.clear();
    for (int i = 0; i < m->n_short; ++i)
        if (is_boundary[i]) boundary_rows.push_back(i);
    m->n_boundary_rows = (int)boundary_rows.size();
    // chunks of the long rows
    std::vector<int4> long_items;
    for (int i = m->n_short; i < n_loc; ++i) {
        const int first = (int)long_items.size();
        for (int j = rp[i]; j < rp[i + 1]; j += KMCF_LONG_CHUNK)
            long_items.push_back(make_int4(i, j, std::min(j + KMCF_LONG_CHUNK, rp[i + 1]), first));
    }
    m->n_long_items = (int)long_items.size();
    m->h_row_ptr = rp;

    if (c->device < 0) {  // host-only planning communicator: no device state
        guard.m = nullptr;
        *out = m;
        return KMCF_OK;
    }

    KMCF_TRY(kmcf_enter(c));
    KMCF_TRY(dev_upload(&m->d_row_ptr, rp));
    KMCF_TRY(dev_upload(&m->d_col, col_local));
    if (h_val) {
        KMCF_TRY(dev_upload(&m->d_val, val_int));
    } else {
        KMCF_TRY(dev_alloc(&m->d_val, (size_t)nnz));
    }
    if (!m->h_perm.empty()) KMCF_TRY(dev_upload(&m->d_perm, m->h_perm));
    if (m->n_long_items > 0) {
        KMCF_TRY(dev_upload(&m->d_long_items, long_items));
        KMCF_TRY(dev_alloc(&m->d_long_part, (size_t)m->n_long_items));
        KMCF_TRY(dev_alloc(&m->d_long_ctr, 1));
    }
    if (m->n_halo > 0) {
        KMCF_TRY(dev_upload(&m->d_is_boundary, is_boundary));
        KMCF_TRY(dev_upload(&m->d_boundary_rows, boundary_rows));
        KMCF_TRY(dev_upload(&m->d_send_idx, send_idx));
        KMCF_TRY(dev_alloc(&m->d_send_buf, (size_t)m->n_send));
        KMCF_TRY(dev_upload(&m->d_halo_gid, halo_gid));
    }
    KMCF_TRY(dev_alloc(&m->d_p, (size_t)n_loc + m->n_halo + 2));
    KMCF_TRY(dev_alloc(&m->d_Ap, (size_t)n_loc + 2));
    KMCF_TRY(dev_alloc(&m->d_r, (size_t)n_loc + 2));
    KMCF_TRY(dev_alloc(&m->d_x, (size_t)n_loc + 2));
    KMCF_TRY(dev_alloc(&m->d_dinv, (size_t)n_loc + 2));
    KMCF_TRY(dev_alloc(&m->d_part_a, (size_t)4 * KMCF_MAX_PARTIALS));
    KMCF_TRY(dev_alloc(&m->d_part_b, (size_t)KMCF_MAX_PARTIALS));
    KMCF_TRY(dev_alloc(&m->d_part_c, (size_t)KMCF_MAX_PARTIALS));
    KMCF_TRY(dev_alloc(&m->d_S, 1));
    // Plan hint: a matrix created without values is filled by the K assembly (two off-diagonal values, always
    // coded); one created with values is coded iff its off-diagonals take few distinct values.
    m->expect_coded = true;
    if (h_val) {
        long long dict[KMCF_DICT_MAX];
        int nd = 0, last = -1;
        for (int i = 0; i < m->n_short && m->expect_coded; ++i)
            for (int j = rp[i]; j < rp[i + 1]; ++j) {
                if (col_local[j] == i) continue;
                long long b;
                memcpy(&b, &val_int[j], sizeof(b));
                if (last >= 0 && dict[last] == b) continue;
                int k = 0;
                while (k < nd && dict[k] != b) ++k;
                last = k;
                if (k < nd) continue;
                if (nd == KMCF_DICT_MAX) { m->expect_coded = false; break; }
                dict[nd++] = b;
            }
    }
    KMCF_TRY(kmcf_spmv_plan(m));
    if (h_val) KMCF_TRY(kmcf_matrix_encode_from_host(m, val_int.data()));
    // What every rank sends to a neighbour must be what that neighbour expects, or the grouped ncclSend / ncclRecv
    // of the first exchange hangs instead of failing (a structurally asymmetric matrix; the host loopback transport
    // checks the same at exchange time).  One small all-gather at build time; with the peer-to-peer transport the
    // same table tells every sender where in the receiver's window its data and its flag go.
    if (P > 1 && c->connected && (c->p2p_active || !c->group)) {
        // per rank: sent to q | expected from q | landing offset | flag offset | acknowledgement offset (8-byte units, in
        // this rank's window, for what q sends) | this rank's halo size (= distance between its two landing buffers)
        // | first granule of q's values in this rank's granule zone | this rank's reduction zone (register-resident solve)
        const int W = 8 * P;
        std::vector<int> tab((size_t)W * P, 0), cnt(P, W), dsp(P);
        for (int q = 0; q < P; ++q) dsp[q] = W * q;
        int land8 = 0, flag8 = 0, ack8 = 0, ll8 = 0, red8 = 0;
        if (c->p2p_active) KMCF_TRY(kmcf_p2p_matrix_alloc(m, &land8, &flag8, &ack8, &ll8, &red8));
        for (int q = 0; q < P; ++q) tab[(size_t)W * rank + 7 * P + q] = red8;
        for (int k = 1; k < nnb; ++k) {
            const int q = m->neighbours[k];
            tab[(size_t)W * rank + q] = (int)m->rows_per_neighbour[k].size();
            tab[(size_t)W * rank + P + q] = (int)m->cols_per_neighbour[k].size();
            tab[(size_t)W * rank + 2 * P + q] = land8 + m->halo_offset[k];       // neighbour q's values land at its halo slots
            tab[(size_t)W * rank + 3 * P + q] = flag8 + (k - 1) * 16;               // (one 128-byte line per flag: P2P_FS)
            tab[(size_t)W * rank + 4 * P + q] = ack8 + (k - 1) * 16;                // q acknowledges MY puts here
            tab[(size_t)W * rank + 5 * P + q] = std::max(m->n_halo, 1);
            tab[(size_t)W * rank + 6 * P + q] = ll8 + 2 * m->halo_offset[k];
        }
        // (kept until the matrix is destroyed: hipFree waits for EVERY stream of the process, and in an in-process group
        // another rank's kernel may already be waiting, on the device, for this rank's next exchange)
        int *&d_tab = m->d_build_tab;
        KMCF_TRY(dev_upload(&d_tab, tab));
        int rc = kmcf_comm_allgatherv_int(c, d_tab, cnt.data(), dsp.data());
        if (rc == KMCF_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = KMCF_ERR_HIP;
        if (rc == KMCF_OK && c->p2p_active) rc = kmcf_p2p_check(c);
        if (rc == KMCF_OK && hipMemcpy(tab.data(), d_tab, tab.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = KMCF_ERR_HIP;
        if (rc != KMCF_OK) return rc;
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) {
                if (tab[(size_t)W * a + b] == tab[(size_t)W * b + P + a]) continue;
                // the lists themselves, from the rank that holds them (seen twice in in-process test groups, never reproduced
                // on purpose: whoever meets it next gets the evidence with the message)
                std::string lst;
                for (int k = 1; k < nnb; ++k) {
                    if (rank == b && m->neighbours[k] == a) {
                        lst = " columns expected:";
                        for (size_t q = 0; q < m->cols_per_neighbour[k].size() && q < 48; ++q) lst += " " + std::to_string(m->cols_per_neighbour[k][q]);
                    }
                    if (rank == a && m->neighbours[k] == b) {
                        lst = " rows sent:";
                        for (size_t q = 0; q < m->rows_per_neighbour[k].size() && q < 48; ++q) lst += " " + std::to_string(m->rows_per_neighbour[k][q] + m->row0);
                    }
                }
                KMCF_CHECK(false, KMCF_ERR_COMM,
                           "kmcf_matrix_build: rank %d sends %d halo values to rank %d, which expects %d (matrix not structurally symmetric?)"
                           " [seen by rank %d; its own neighbour lists: %d / %d; halo sizes in the table: %d, %d;%s]",
                           a, tab[(size_t)W * a + b], b, tab[(size_t)W * b + P + a], rank, (int)m->rows_per_neighbour.size(),
                           (int)m->cols_per_neighbour.size(), tab[(size_t)W * a + 5 * P + b], tab[(size_t)W * b + 5 * P + a], lst.c_str());
            }
        if (c->p2p_active) {
            std::vector<long long> r_land((size_t)nnb, 0), r_flag((size_t)nnb, 0), r_ack((size_t)nnb, 0), r_halo((size_t)nnb, 0), r_ll((size_t)nnb, 0);
            std::vector<long long> r_red((size_t)P, 0);
            for (int k = 1; k < nnb; ++k) {
                const int q = m->neighbours[k];
                r_land[k] = tab[(size_t)W * q + 2 * P + rank];
                r_flag[k] = tab[(size_t)W * q + 3 * P + rank];
                r_ack[k] = tab[(size_t)W * q + 4 * P + rank];
                r_halo[k] = tab[(size_t)W * q + 5 * P + rank];
                r_ll[k] = tab[(size_t)W * q + 6 * P + rank];
            }
            for (int q = 0; q < P; ++q) r_red[(size_t)q] = tab[(size_t)W * q + 7 * P + rank];
            KMCF_TRY(kmcf_p2p_matrix_connect(m, r_land, r_flag, r_ack, r_halo, r_ll, r_red));
        }
    }
    guard.m = nullptr;
    *out = m;
    return KMCF_OK;
}

extern "C" int kmcf_matrix_create_csr(kmcf_comm *c, int matrix_size, const int *h_counts, const int *h_displs,
                                      const int *h_row_ptr, const int *h_col_global, const double *h_val,
                                      kmcf_matrix **out)
{
    return kmcf_matrix_build(c, matrix_size, h_counts, h_displs, h_row_ptr, h_col_global, h_val, nullptr, out);
}

// "Split sparse" operator of the T-matrix path: A = A_neighbour + P^T A_sub P, where A_sub acts on the
// sub-vector of the `subblock_size` tunnel rows of ALL ranks (conjugate_gradient_jacobi_split_sparse +
// dspmv_split_sparse::spmm_split_sparse1/2/3, dist_iterative/dist_conjugate_gradient_split_sparse.cpp:18-182,
// dist_spmv_split_sparse.cpp; Distributed_subblock_sparse, dist_objects.h:52-65).  The reference keeps the
// two pieces apart and all-gathers the sub-vector on every SpMV (ring Isend/Irecv or MPI_Iallgatherv);
// here the sub-block is merged into the row-partitioned CSR at build time, so its off-rank columns simply
// become halo columns of the one compact-halo SpMV and the PCG entry points are the ordinary ones.
// h_sub_global_rows[s] = global row of sub index s (the reference all-gathers these once,
// src/initialize_sparsity_T.cu:752-786); this rank owns sub indices [displ_sub[rank], +count_sub[rank]).
// h_sub_col holds GLOBAL sub indices.  Duplicate (row, col) pairs of the two pieces are kept as two entries.
extern "C" int kmcf_matrix_create_split_sparse(kmcf_comm *c, int matrix_size, const int *h_counts, const int *h_displs,
                                               const int *h_row_ptr, const int *h_col_global, const double *h_val,
                                               int subblock_size, const int *h_count_sub, const int *h_displ_sub,
                                               const int *h_sub_global_rows, const int *h_sub_row_ptr,
                                               const int *h_sub_col, const double *h_sub_val, kmcf_matrix **out)
{
    KMCF_CHECK(c && h_counts && h_displs && h_row_ptr && h_count_sub && h_displ_sub && h_sub_global_rows && h_sub_row_ptr && out,
               KMCF_ERR_ARG, "kmcf_matrix_create_split_sparse: null argument");
    const int rank = c->rank, n_loc = h_counts[rank], row0 = h_displs[rank];
    const int ns_loc = h_count_sub[rank], s0 = h_displ_sub[rank];
    KMCF_CHECK(s0 >= 0 && s0 + ns_loc <= subblock_size, KMCF_ERR_ARG, "kmcf_matrix_create_split_sparse: sub-block partition out of range");
    // extra entries per local row
    std::vector<int> extra((size_t)n_loc, 0);
    for (int s = 0; s < ns_loc; ++s) {
        const int r = h_sub_global_rows[s0 + s] - row0;
        KMCF_CHECK(r >= 0 && r < n_loc, KMCF_ERR_ARG, "kmcf_matrix_create_split_sparse: sub index %d maps to row %d outside this rank", s0 + s, r + row0);
        extra[r] += h_sub_row_ptr[s + 1] - h_sub_row_ptr[s];
    }
    std::vector<int> rp((size_t)n_loc + 1, 0);
    for (int r = 0; r < n_loc; ++r) rp[r + 1] = rp[r] + (h_row_ptr[r + 1] - h_row_ptr[r]) + extra[r];
    std::vector<int> col((size_t)rp[n_loc]);
    std::vector<double> val((size_t)rp[n_loc]);
    std::vector<int> fill(rp.begin(), rp.end() - 1);
    for (int r = 0; r < n_loc; ++r)
        for (int j = h_row_ptr[r]; j < h_row_ptr[r + 1]; ++j) { col[fill[r]] = h_col_global[j]; val[fill[r]++] = h_val ? h_val[j] : 0.0; }
    for (int s = 0; s < ns_loc; ++s) {
        const int r = h_sub_global_rows[s0 + s] - row0;
        for (int j = h_sub_row_ptr[s]; j < h_sub_row_ptr[s + 1]; ++j) {
            KMCF_CHECK(h_sub_col[j] >= 0 && h_sub_col[j] < subblock_size, KMCF_ERR_ARG, "kmcf_matrix_create_split_sparse: sub column out of range");
            col[fill[r]] = h_sub_global_rows[h_sub_col[j]];
            val[fill[r]++] = h_sub_val ? h_sub_val[j] : 0.0;
        }
    }
    return kmcf_matrix_build(c, matrix_size, h_counts, h_displs, rp.data(), col.data(), val.data(), nullptr, out);
}

extern "C" int kmcf_matrix_destroy(kmcf_matrix *m)
{
    if (!m) return KMCF_OK;
    if (m->comm && m->comm->device >= 0) {
        hipSetDevice(m->comm->device);
        hipStreamSynchronize(m->comm->stream);
        hipStreamSynchronize(m->comm->comm_stream);
        kmcf_p2p_matrix_free(m);
        kmcf_sell_free(m);
        kmcf_cgr_free(m);
        void *ptrs[] = {m->d_row_ptr, m->d_col, m->d_val, m->d_boundary_rows, m->d_is_boundary, m->d_send_idx,
                        m->d_send_buf, m->d_halo_gid, m->d_p, m->d_Ap, m->d_r, m->d_x, m->d_dinv,
                        m->d_part_a, m->d_part_b, m->d_part_c, m->d_S, m->d_chunk_row, m->d_perm, m->d_pd, m->d_s,
                        m->d_tile, m->d_wcol, m->d_idx16, m->d_dict, m->d_diagv, m->d_diag_pos, m->d_code_fail,
                        m->d_long_items, m->d_long_part, m->d_long_ctr, m->d_tile4, m->d_tbase, m->d_build_tab};
        for (void *p : ptrs)
            if (p) hipFree(p);
    }
    delete m;
    return KMCF_OK;
}

extern "C" int kmcf_matrix_row_order(const kmcf_matrix *m, int *h_perm, int *n_short, int *h_tile_end, int *n_tiles)
{
    KMCF_CHECK(m, KMCF_ERR_ARG, "kmcf_matrix_row_order: null matrix");
    if (h_perm)
        for (int i = 0; i < m->n_loc; ++i) h_perm[i] = m->h_perm.empty() ? i : m->h_perm[i];
    if (n_short) *n_short = m->n_short;
    if (n_tiles) *n_tiles = (int)m->h_sell_cuts.size();
    if (h_tile_end)
        for (size_t t = 0; t < m->h_sell_cuts.size(); ++t) h_tile_end[t] = m->h_sell_cuts[t];
    return KMCF_OK;
}

extern "C" int kmcf_matrix_info(const kmcf_matrix *m, kmcf_matrix_info_t *info)
{
    KMCF_CHECK(m && info, KMCF_ERR_ARG, "kmcf_matrix_info: null argument");
    info->matrix_size = m->matrix_size;
    info->rows_this_rank = m->n_loc;
    info->nnz = m->nnz;
    info->number_of_neighbours = m->number_of_neighbours;
    info->halo_cols = m->n_halo;
    info->send_rows = m->n_send;
    info->boundary_rows = m->n_boundary_rows;
    info->spmv_kind = m->spmv_kind;
    const bool sellc = m->spmv_kind == 2 && m->coded && m->sell_ok && m->dict_n <= 3;
    const bool sell = sellc || kmcf_sellv_usable(m);      // either row-per-lane kernel: the lane stream's figures
    info->spmv_coded = (m->spmv_kind == 2 && m->coded) ? (sellc ? 2 : 1) : 0;
    info->spmv_tiles = m->spmv_kind == 2 ? (sell ? m->n_sell_tiles : m->n_tiles) : 0;
    info->spmv_window_cols = m->spmv_kind == 2 ? (sell ? m->n_sell_wcols : m->n_wcols) : 0;
    info->spmv_stream_entries = sell ? m->n_sell_entries : (info->spmv_coded ? (int64_t)m->h_row_ptr[m->n_short] : 0);
    return KMCF_OK;
}

extern "C" int kmcf_matrix_neighbour(const kmcf_matrix *m, int k, int *neighbour_rank, int *nnz_block,
                                     int *ncols, int *h_cols, int *nrows, int *h_rows)
{
    KMCF_CHECK(m && k >= 0 && k < m->number_of_neighbours, KMCF_ERR_ARG, "kmcf_matrix_neighbour: k=%d out of range", k);
    if (neighbour_rank) *neighbour_rank = m->neighbours[k];
    if (nnz_block) *nnz_block = m->nnz_per_neighbour[k];
    if (ncols) *ncols = (int)m->cols_per_neighbour[k].size();
    if (nrows) *nrows = (int)m->rows_per_neighbour[k].size();
    if (h_cols) memcpy(h_cols, m->cols_per_neighbour[k].data(), m->cols_per_neighbour[k].size() * sizeof(int));
    if (h_rows) memcpy(h_rows, m->rows_per_neighbour[k].data(), m->rows_per_neighbour[k].size() * sizeof(int));
    return KMCF_OK;
}

extern "C" int kmcf_matrix_set_values(kmcf_matrix *m, const double *h_val)
{
    KMCF_CHECK(m && h_val, KMCF_ERR_ARG, "kmcf_matrix_set_values: null argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_matrix_set_values: host-only matrix");
    KMCF_TRY(kmcf_enter(m->comm));
    KMCF_HIP(hipStreamSynchronize(m->comm->stream));
    if (m->h_perm.empty()) {
        KMCF_HIP(hipMemcpy(m->d_val, h_val, (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice));
        KMCF_TRY(kmcf_matrix_encode_from_host(m, h_val));
    } else {
        // creation order -> internal order, row by row (entries keep their order inside a row)
        std::vector<double> v((size_t)m->nnz);
        for (int i = 0; i < m->n_loc; ++i) {
            const int r = m->h_perm[i];
            const int len = m->h_row_ptr_user[r + 1] - m->h_row_ptr_user[r];
            if (len) memcpy(&v[m->h_row_ptr[i]], &h_val[m->h_row_ptr_user[r]], (size_t)len * sizeof(double));
        }
        KMCF_HIP(hipMemcpy(m->d_val, v.data(), (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice));
        KMCF_TRY(kmcf_matrix_encode_from_host(m, v.data()));
    }
    return KMCF_OK;
}

extern "C" int kmcf_matrix_get_values(const kmcf_matrix *m, double *h_val)
{
    KMCF_CHECK(m && h_val, KMCF_ERR_ARG, "kmcf_matrix_get_values: null argument");
    KMCF_CHECK(m->d_val, KMCF_ERR_STATE, "kmcf_matrix_get_values: host-only matrix");
    KMCF_TRY(kmcf_enter(m->comm));
    KMCF_HIP(hipStreamSynchronize(m->comm->stream));
    if (m->h_perm.empty()) {
        KMCF_HIP(hipMemcpy(h_val, m->d_val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        std::vector<double> v((size_t)m->nnz);
        KMCF_HIP(hipMemcpy(v.data(), m->d_val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < m->n_loc; ++i) {
            const int r = m->h_perm[i];
            const int len = m->h_row_ptr_user[r + 1] - m->h_row_ptr_user[r];
            if (len) memcpy(&h_val[m->h_row_ptr_user[r]], &v[m->h_row_ptr[i]], (size_t)len * sizeof(double));
        }
    }
    return KMCF_OK;
}
