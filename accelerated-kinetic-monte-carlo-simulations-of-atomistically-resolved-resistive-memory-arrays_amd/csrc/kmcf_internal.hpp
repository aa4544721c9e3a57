// Internal declarations of libkmcfield (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "kmcfield.h"

// ---------------------------------------------------------------- errors
void kmcf_set_error(const char *fmt, ...);

#define KMCF_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            kmcf_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return KMCF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define KMCF_CHECK(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            kmcf_set_error(__VA_ARGS__);   \
            return (code);                 \
        }                                  \
    } while (0)

#define KMCF_TRY(call)            \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != KMCF_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------- tuning
constexpr int KMCF_BLOCK = 256;          // 4 wavefronts
constexpr int KMCF_MAX_PARTIALS = 2048;  // = 256 CUs x 8 resident blocks
constexpr int KMCF_CHUNK_ITERS = 32;     // CG iterations enqueued between convergence read-backs
constexpr int KMCF_DICT_MAX = 62;        // value dictionary of the coded window SpMV (codes 0..61)
constexpr int KMCF_CODE_DIAG = 63;       // code of a row's diagonal entry (value in d_diagv)
constexpr int KMCF_SLOT_BITS = 10;       // window slots < 1024

// Device-resident CG scalars (never round-trip through the host inside the loop).
struct kmcf_scalars {
    double bb;        // ||b||^2
    double rz[2];     // r.z ping-pong by iteration parity
    double rz_last;   // most recent r.z
    double pAp;
    double red[4];    // all-reduce staging (multi-rank)
    double alpha[2];  // single-reduction CG: step length ping-pong by iteration parity
    int done;         // stopping rule met
    int iters;        // iterations executed
    int x_pending;    // classic loop: x += xa * p of the last iteration is still to be applied (cg_p / cg_x / cg_out)
    int stop_k;       // iteration whose loop-head kernel set `done` (0: none).  That kernel's own blocks test THIS word,
                      // never `done`: block 0 writes `done` while other blocks of the same launch may not have started
    double xa;        // ... its step length
};

struct kmcf_matrix;


// In-process "loopback" group: the P ranks are P host threads of ONE process sharing one GPU.
// Transport for testing the multi-rank logic (halo maps, boundary pass, reductions) on a 1-GPU
// box, where RCCL refuses two ranks on one device.  Host-synchronous, not a performance path.
struct kmcf_group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    bool broken = false;               // a barrier timed out: the group is unusable
    std::vector<void *> slot;            // per-rank published pointer
    std::vector<kmcf_matrix *> mat;      // per-rank matrix taking part in the current halo exchange
    int refs = 0;
};

struct kmcf_p2p;        // peer-to-peer transport of a communicator (kmcf_p2p.hip)
struct kmcf_p2p_halo;   // ... and the halo protocol state of one matrix

struct kmcf_comm {
    kmcf_group *group = nullptr;        // loopback transport (nullptr: RCCL)
    kmcf_p2p *p2p = nullptr;            // mapped peer windows; used for the exchanges while p2p_active
    bool p2p_active = false;
    bool in_solve = false;              // inside a CG loop: its per-iteration exchanges never meet on the host (kmcf_group_rendezvous)
    int device = 0;
    int nranks = 1;
    int rank = 0;
    hipStream_t stream = nullptr;       // compute stream
    hipStream_t comm_stream = nullptr;  // halo exchange stream
    hipEvent_t ev_packed = nullptr;     // compute -> comm
    hipEvent_t ev_halo = nullptr;       // comm -> compute
    hipEvent_t ev_subpack = nullptr, ev_sub = nullptr;   // tunnel sub-vector: packed (compute -> comm), gathered (comm -> compute)
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_a0 = nullptr, ev_a1 = nullptr;   // assembly timing (kmcf_background_potential_sparse)
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr;   // first / last device work of a kmcf_pcg_jacobi call
    hipEvent_t ev_entry = nullptr;      // caller's stream -> compute stream at every entry point (kmcf_enter)
    hipStream_t caller_stream = nullptr;  // stream the caller's own work is queued on (default: legacy null stream)
    double *d_scratch = nullptr;        // 1024 doubles of persistent scratch (heat reduction partials)
    void *nccl = nullptr;               // ncclComm_t: halo send/recv (comm stream)
    void *nccl_red = nullptr;           // ncclComm_t: all-reduce / gathers (compute stream)
    bool force_collectives = false;     // KMCF_FORCE_COMM: 1-rank group still runs the collectives
    bool connected = false;
    int mark_seq = 0;                   // number of the last chunk check (cg_chunk_check)
    int *h_pinned = nullptr;            // 16 ints pinned host (done/iters read-back)
    kmcf_scalars *h_scal = nullptr;     // pinned host copy of a solve's scalars (read after the call's one sync)
    // event-step workspace kept between KMC steps (kmcf_execute_kmc_step, kmcf_events.hip)
    struct kmcf_event_cache *ev_cache = nullptr;
};
void kmcf_event_cache_free(kmcf_comm *c);

struct kmcf_subop;   // sub-block operator of the split T matrix (kmcf_tstate.hip)

constexpr int KMCF_LONG_CHUNK = 2048;    // entries of a long row handled by one block (spmv_long_kernel)

struct kmcf_matrix {
    kmcf_comm *comm = nullptr;
    int matrix_size = 0;
    int n_loc = 0;
    // Long rows (more than KMCF_LONG_ROW entries, default 384: the T matrix's two virtual-node rows hold
    // one entry per contact atom of a layer, 578 at 5 nm and 33 602 at 40 nm) do not fit the tiles of the
    // window / stream kernels.  They are moved to the END of the internal row order: the tiled kernels see
    // rows [0, n_short), spmv_long_kernel handles rows [n_short, n_loc) in chunks of KMCF_LONG_CHUNK entries
    // with a deterministic two-stage sum.  Vector kernels (CG updates) run over all n_loc rows.
    int n_short = 0;
    int n_long_items = 0;
    int4 *d_long_items = nullptr;      // (row, first entry, end entry, first item of this row) per chunk
    double *d_long_part = nullptr;     // one partial per chunk
    unsigned int *d_long_ctr = nullptr;
    kmcf_subop *sub = nullptr;         // optional: y[sub rows] += S x_sub after the CSR part (T matrix)
    kmcf_p2p_halo *p2p = nullptr;      // halo landing zone / flags in the peer windows (p2p transport)
    // optional, for the length of one kmcf_pcg_workspace call that runs resident (kmcf_pcg_resident_applies): the right-hand
    // side where it lies (internal order, read-only) instead of d_r, and the start guess / solution in the CALLER's order and
    // place instead of d_x -- the K solve of a KMC step then needs no copy and no permuting kernel around its one launch
    const double *solve_b_src = nullptr;
    double *solve_x_user = nullptr;
    bool last_solve_resident = false;  // the solve enqueued last ran as ONE resident launch (its scalars are complete: no tail for the output kernel to form)
    struct kmcf_cgr *cgr = nullptr;    // plan and buffers of the register-resident solve (kmcf_cgr.hip); tpb == 0: does not qualify
    int row0 = 0;                 // displs[rank]
    int64_t nnz = 0;
    std::vector<int> counts, displs;

    // neighbour bookkeeping (host) -- mirrors Distributed_matrix
    int number_of_neighbours = 1;
    std::vector<int> neighbours;              // [0] = self, cyclic order
    std::vector<int> nnz_per_neighbour;
    std::vector<std::vector<int>> cols_per_neighbour;  // block-local columns to receive
    std::vector<std::vector<int>> rows_per_neighbour;  // local rows to send
    std::vector<int> halo_offset;             // start of neighbour k's slots in the halo (k>=1)
    int n_halo = 0;
    int n_send = 0;
    std::vector<int> send_offset;
    int n_boundary_rows = 0;

    // creation-order -> internal nnz position (identity today; kept for set/get_values)
    // device CSR with compact-halo column ids: [0,n_loc) own, [n_loc, n_loc+n_halo) halo slots
    int *d_row_ptr = nullptr;
    int *d_col = nullptr;
    double *d_val = nullptr;
    int *d_boundary_rows = nullptr;    // list of local rows touching the halo
    unsigned char *d_is_boundary = nullptr;  // per-row flag (nullptr if no halo)
    int *d_build_tab = nullptr;        // exchange table of the build (freed with the matrix)
    int *d_send_idx = nullptr;         // concatenated rows_per_neighbour[k>=1]
    double *d_send_buf = nullptr;
    int *d_halo_gid = nullptr;         // global column id of each halo slot

    // Internal row order.  Rows (and own-block columns, and every workspace vector) may be stored in a
    // locality-improving order: internal row i = caller's local row perm[i].  Empty = identity.
    // Vectors crossing the C ABI are permuted on the way in / out (kmcf_vec_in / kmcf_vec_out); the
    // halo protocol (send / receive lists) stays in the caller's order.
    std::vector<int> h_perm;           // internal -> caller local row
    std::vector<int> h_row_ptr_user;   // row_ptr in the caller's order (set/get_values)
    int *d_perm = nullptr;

    // CG workspace (allocated once; the reference mallocs Ap/z in create_cg_overhead)
    double *d_p = nullptr;             // n_loc + n_halo   (Distributed_vector)
    double *d_Ap = nullptr;
    double *d_r = nullptr;
    double *d_x = nullptr;
    double *d_dinv = nullptr;
    double *d_pd = nullptr;            // single-reduction CG: search direction p (d_p then carries z)
    double *d_s = nullptr;             // single-reduction CG: s = A p by recurrence
    double *d_part_a = nullptr;        // pAp partials: [0, MAXP) interior rows | [MAXP, 2 MAXP) boundary rows |
                                       // [2 MAXP] long rows | [3 MAXP, 4 MAXP) sub-block operator
    double *d_part_b = nullptr;        // rz partials
    double *d_part_c = nullptr;        // bb partials
    kmcf_scalars *d_S = nullptr;

    // SpMV launch plan
    int spmv_grid = 0;                 // interior pass grid (= number of pAp partials it writes)
    int spmv_grid_b = 0;               // boundary pass grid
    int spmv_lpr = 16;                 // lanes per row (vec kernel)
    bool stream_nt = false;            // the f64-value kernels mark their matrix loads nontemporal (matrix larger than the caches)
    int spmv_kind = 0;                 // 0: vec<LPR>, 1: stream (nnz-chunked, LDS row reduction), 2: window
    // window kernel (kind 2): tiles of whole rows whose distinct columns (<= spmv_wmax) are staged in LDS
    int n_tiles = 0;
    int64_t n_wcols = 0;               // sum of the tiles' window sizes
    int spmv_wmax = 0;
    int2 *d_tile = nullptr;            // (first row, first window slot) per tile, n_tiles + 1
    int4 *d_tile4 = nullptr;           // (first row, rows, first window slot, window size) per tile: the coded kernel's view
    int *d_tbase = nullptr;            // first entry of each tile
    int *d_wcol = nullptr;             // column of each window slot (ascending inside a tile)
    unsigned short *d_idx16 = nullptr; // per nnz: window slot of its column inside the tile (bits 0-9) | value code (10-15)
    // Dictionary-coded values (window kernel only): when every off-diagonal value of the matrix is one of
    // <= KMCF_DICT_MAX distinct doubles (K and the CB-edge system hold two: -high_G, -low_G), the slot stream
    // carries the value's dictionary code and the SpMV does not read d_val at all: 2 B/nnz instead of 10.
    // Diagonal entries carry code KMCF_CODE_DIAG and are taken from d_diagv.  Lossless: same doubles, same
    // products; only the diagonal product is added last instead of in column order.
    bool coded = false;                // codes + d_diagv + d_dict currently match d_val
    bool expect_coded = true;          // plan hint: tiles sized for the coded kernel (<= 8*U rows: full passes)
    bool tiles_for_coded = false;      // the plan followed the hint (the coded kernel requires its entry limit)
    double *d_dict = nullptr;          // KMCF_DICT_MAX + 1 doubles
    double *d_diagv = nullptr;         // diagonal value per row (0 where a row has no diagonal entry)
    int *d_diag_pos = nullptr;         // nnz index of each row's diagonal entry, -1 if none
    int *d_code_fail = nullptr;        // encode kernel: set when a value is not in the dictionary
    std::vector<int> h_diag_pos;
    double h_dict[64] = {0.0};         // host copy of d_dict (lives as long as the matrix: async upload source)
    bool dict_uploaded = false;
    int dict_n = 0;                    // dictionary entries in use
    int spmv_grid_coded = 0;           // interior grid while coded (its kernel's residency differs)
    // Row-per-lane layout of the coded stream (spmv_sell_kernel, kmcf_spmv.hip): tiles of <= 256 rows sharing one
    // x window; inside a tile the rows are sorted by length and dealt to the lanes of 4 waves; a wave's stream is
    // [step q][lane][4 entries] padded to the wave's longest row.  Entry = LDS byte offset of value-times-x:
    // ((code << sell_lw) | slot) << 3.  Derived from d_idx16's codes by sell_refresh_kernel whenever they change.
    bool sell_ok = false;              // the layout exists (plan accepted it)
    bool sell_dirty = true;            // codes in d_idx16 are newer than d_sell
    std::vector<int> h_sell_cuts;      // end row of every tile, fixed by kmcf_sell_refine_order (empty: plan_sell cuts greedily)
    bool sell_ident = false;           // every tile's rows are already sorted by length: lane t owns row r0 + t
    bool sell_nt = false;              // the coded entry stream is larger than the Infinity Cache: loaded nontemporal
    int sell_lw = 10, sell_nq = 0;     // log2 of the window slots per class; steps of 4 entries held in registers
    int n_sell_tiles = 0, sell_grid = 0;
    int64_t n_sell_wcols = 0, n_sell_entries = 0;
    int4 *d_sell_tile = nullptr;       // (first row, rows, first window slot, window size) per tile
    int2 *d_sell_wave = nullptr;       // per tile and wave: (first 8-byte group of its stream, steps)
    int *d_sell_lrow = nullptr;        // per tile and lane: row - first row, -1 = idle lane
    int *d_sell_wcol = nullptr;        // column of each window slot
    unsigned short *d_sell = nullptr;  // the entry stream
    int *d_sell_pos = nullptr;         // per row: position of its first entry in d_sell
    // ... the same layout with f64 VALUES streamed next to the 16-bit slots (spmv_sellv_kernel: matrices whose values are
    // not dictionary-coded -- general CSR input, the symmetrically scaled CB-edge system): d_sellv[pos] = value of entry pos
    // of d_sell (0.0 in the padding), refreshed from d_val whenever the values changed; allocated on first use
    double *d_sellv = nullptr;
    bool sellv_dirty = true;
    int sellv_grid = 0;
    int spmv_u = 8;                    // stream: nnz per thread per chunk
    int spmv_lpr2 = 4;                 // stream: lanes per row in the LDS reduction
    int n_chunks = 0;
    int *d_chunk_row = nullptr;        // stream: first row of each chunk (n_chunks + 1)
    std::vector<int> h_row_ptr;        // host copy of row_ptr (launch planning)
};

struct kmcf_kstate {
    kmcf_comm *comm = nullptr;
    kmcf_matrix *K = nullptr;
    int N = 0, N_left = 0, N_right = 0, N_interface = 0;
    // host copies of the pattern in global interface column ids (for export)
    std::vector<int> h_row_ptr, h_col;
    std::vector<int> h_left_row_ptr, h_left_col, h_right_row_ptr, h_right_col;
    // contact patterns on device (gpubuf.left_row_ptr_d etc.)
    int *d_left_row_ptr = nullptr, *d_left_col = nullptr;
    int *d_right_row_ptr = nullptr, *d_right_col = nullptr;
    int *d_diag_pos = nullptr;         // nnz index of the diagonal entry per row
    unsigned char *d_cls = nullptr;    // per-site class (N): 1 metal, 2 uncharged vacancy, 0 other
    unsigned char *d_cls_col = nullptr;  // the same per internal column (own rows in internal order | halo slots)
    double *d_diag = nullptr, *d_left = nullptr, *d_right = nullptr, *d_rhs = nullptr;
    bool assembled = false;
    // replicated interface solution for sum_and_gather
    double *d_gather = nullptr;
};

void kmcf_sell_free(kmcf_matrix *m);       // frees the row-per-lane layout (kmcf_spmv.hip)
void kmcf_sell_refine_order(int n_short, int n_cols, const int *rp, const int *col, std::vector<int> &perm, std::vector<int> &cuts);

// grid of the CG's vector kernels (= r.z / b.b partials): 8 blocks per CU at most, one partial per block.  One step of
// two rows per lane where that fits (KMCF_VEC_ROWS rows per lane, default 2): the kernels request a block's first
// step together with its scalars, so with a single step a block's life is ONE memory round trip (a rank's eighth of
// the 40 nm matrix: update kernel 7.2 us with two steps per lane).
// KMCF_DEVICE_SHARE = s: s ranks share this GPU (rehearsals of an N-rank run on fewer GPUs): every grid that is sized to
// fill the chip takes 1/s of it, so that the ranks' kernels -- which wait for each other on the device -- are resident
// together (two whole-chip grids of waiting blocks on one GPU starve each other: seen as all-reduce time-outs at 40 nm)
inline int kmcf_device_share()
{
    static const int s = getenv("KMCF_DEVICE_SHARE") ? std::max(1, atoi(getenv("KMCF_DEVICE_SHARE"))) : 1;
    return s;
}

inline int kmcf_vec_grid(int n)
{
    static const int rows = getenv("KMCF_VEC_ROWS") ? std::max(2, atoi(getenv("KMCF_VEC_ROWS"))) : 2;
    int64_t g = ((int64_t)n + KMCF_BLOCK * rows - 1) / (KMCF_BLOCK * rows);
    if (g < 1) g = 1;
    if (g > KMCF_MAX_PARTIALS / kmcf_device_share()) g = KMCF_MAX_PARTIALS / kmcf_device_share();
    return (int)g;
}

// recurrence of a solve on this matrix: false = classic (the reference's operation order; default for one rank),
// true = single-reduction variant (default for multi-rank groups); KMCF_CG_VARIANT=classic|cg1r overrides
inline bool kmcf_cg_single_reduction(const kmcf_matrix *m);

// grid of the interior SpMV pass = number of p.Ap partials it writes
inline bool kmcf_sellv_usable(const kmcf_matrix *m)
{
    return m->spmv_kind == 2 && !m->coded && m->sell_ok && m->sell_ident && m->sellv_grid > 0;      // (sellv_grid: 0 with KMCF_SPMV_SELLV=0 at plan time)
}

inline int kmcf_interior_grid(const kmcf_matrix *m)
{
    if (m->spmv_kind == 2 && m->coded) return (m->sell_ok && m->dict_n <= 3) ? m->sell_grid : m->spmv_grid_coded;
    if (kmcf_sellv_usable(m)) return m->sellv_grid;
    return m->spmv_grid;
}

inline bool kmcf_cg_single_reduction(const kmcf_matrix *m)
{
    bool cg1r = m->comm->nranks > 1;
    if (const char *e = getenv("KMCF_CG_VARIANT")) cg1r = (e[0] == 'c' && e[1] == 'g');
    return cg1r;
}

// Sub-block operator of a split matrix A = A_csr + P^T S P (the T matrix's tunnel block): S acts on the
// sub-vector of the n_glob sub points of ALL ranks; this rank owns rows [displs[rank], +counts[rank]).
// Stored as a bitmap over the n_glob columns (one 64-bit mask per 64 columns and row) plus the packed f64
// values of the set positions: 8 B/nnz + 1 bit per position, against 12 B/nnz for CSR -- the block is
// dense-ish (36 % at 5 nm here, 43 % in the authors' 40 nm test set, main_test_cg_split.cpp:1030-1035).
struct kmcf_subop {
    int n_glob = 0, n_loc = 0, row0 = 0;     // sub points: all ranks, this rank, first of this rank
    int n_groups = 0;                        // ceil(n_glob / 64)
    std::vector<int> counts, displs;
    int *d_rows = nullptr;                   // internal local row of each local sub point
    unsigned long long *d_mask = nullptr;    // n_loc x n_groups
    long long *d_voff = nullptr;             // n_loc + 1: first value of each row
    double *d_val = nullptr;
    double *d_xsub = nullptr;                // n_glob: gathered sub-vector
    long long nnz = 0;
    size_t cap_mask = 0, cap_val = 0, cap_rows = 0, cap_x = 0, cap_voff = 0;
    int grid = 0;                            // blocks of the operator kernel = partials it writes
    // Dense symmetric storage (kmcf_tstate.hip: sub_symm_kernel), chosen per assembly for one rank when the block is
    // more than half full (the vacancy-vacancy block of a large device is ~100 % dense): the upper block triangle
    // as 64 x 64 tiles of f64 (zeros where the pattern has no entry), diagonal tiles complete.  Half the bytes of the
    // packed full block per application; every tile serves its block row (row sums) and, transposed out of LDS, its
    // block column (column sums); partial sums are written per strip / tile and added in a fixed order.
    bool gather_pending = false;             // the sub-vector all-gather of the current SpMV runs on the comm stream
    bool dense = false;
    int nb = 0;                              // ceil(n_glob / 64)
    int n_strips = 0;
    long long n_tiles = 0;
    double *d_tiles = nullptr;               // n_tiles x 4096
    int4 *d_strips = nullptr;                // (block row I, first block column J0, tiles, first tile index)
    int *d_strip_first = nullptr;            // nb + 1: first strip of every block row
    double *d_rowpart = nullptr, *d_colpart = nullptr;   // 2 x 64 per strip / 2 x 64 per tile (the power pass needs two sums)
    size_t cap_tiles = 0, cap_strips = 0, cap_sf = 0, cap_rowpart = 0, cap_colpart = 0;
    // ... spread over a rank group (round 4): the strips of the upper block triangle dealt to the ranks (each strip of the
    // global list -- block rows ascending, then columns -- to the rank holding the fewest tiles so far, the lowest such
    // rank; n_strips / n_tiles / d_strips / d_tiles then count and hold THIS rank's), every rank forms its partial of ALL 64 nb sums, the partials are all-gathered and added in rank order
    // by the owner of each point (sub_combine_kernel).  One more all-gather per application than the row-sliced bitmap
    // form -- for half the bytes per entry and the tile kernel's rate.
    bool spread = false;
    int *d_tile_local = nullptr;             // global tile index -> this rank's tile, -1: another rank's
    double *d_ypart = nullptr;               // P x (1 or 2) x 64 nb: every rank's partial sums (rank q's at q W 64 nb)
    long long n_tiles_glob = 0;
    size_t cap_tile_local = 0, cap_ypart = 0;
    std::vector<int> y_counts[2], y_displs[2];   // the all-gather of the partials: one sum per point (the operator) / two (the power pass)
    // ... or (jagged, with dense set: strips, parts and their reduction are shared) the same tiles holding only their
    // ENTRIES: tile-major row masks + the values in layers (kmcf_tstate.hip: sub_symj_kernel) -- 4 B per entry of the
    // full block + 1 bit per position instead of 4 B per position
    bool jagged = false;
    unsigned long long *d_jmask = nullptr;   // n_tiles x 64
    long long *d_jvoff = nullptr;            // n_tiles + 1: first value of every tile
    double *d_jval = nullptr;
    int *d_jcnt = nullptr;                   // entries per tile (scan input)
    long long jnnz = 0;                      // stored entries (upper tiles + complete diagonal tiles)
    size_t cap_jmask = 0, cap_jvoff = 0, cap_jval = 0, cap_jcnt = 0;
};

// ---------------------------------------------------------------- internal entry points
// spmv.hip
int kmcf_spmv_plan(kmcf_matrix *m);
// Ap = A*p on m->d_p (already holding local p), writes m->d_Ap and pAp partials.
// flags bit 0 ("direct" peer-to-peer protocol only): the caller's NEXT kernel acknowledges this SpMV's halo
int kmcf_spmv_device(kmcf_matrix *m, bool with_dot, bool skip_if_done, int flags = 0);
// up to four arrays of per-block partial sums (an SpMV writes one per pass: interior rows, boundary rows, long
// rows, sub-block), added in a fixed order by whoever consumes them
struct kmcf_part4 { const double *p[4]; int n[4]; };
kmcf_part4 kmcf_spmv_partials(const kmcf_matrix *m);
// tstate.hip: y[sub rows] += S x_sub (+ dot partials).  begin: pack the local part of x_sub out of m->d_p and start the
// all-gather of the ranks' parts -- on the comm stream where the transport allows, so that it runs underneath the
// neighbour part of the SpMV (spmm_split_sparse2/3 post the exchange first and poll it, dist_spmv_split_sparse.cpp:
// 123-192, 246-337); finish: wait for it, then the block's kernels on the compute stream
int kmcf_subop_begin(kmcf_matrix *m, bool skip_if_done);
int kmcf_subop_finish(kmcf_matrix *m, bool with_dot, bool skip_if_done);
// Dictionary-code the values now in d_val (window kernel only; no-op otherwise).  h_dict/nd: the distinct
// off-diagonal values; m->coded is set iff every off-diagonal value was found.  Synchronous.
int kmcf_matrix_encode_values(kmcf_matrix *m, const double *h_dict, int nd);
// Host scan of internal-order values for their distinct off-diagonal values, then kmcf_matrix_encode_values.
int kmcf_matrix_encode_from_host(kmcf_matrix *m, const double *h_val_internal);
// For producers that write codes and d_diagv themselves (K assembly): install the dictionary, mark coded.
int kmcf_matrix_set_dictionary(kmcf_matrix *m, const double *h_dict, int nd);
// cg.hip
int kmcf_halo_exchange_begin(kmcf_matrix *m);   // pack + send/recv on the comm stream
int kmcf_halo_exchange_end(kmcf_matrix *m);     // compute stream waits for the halo
// comm.hip
// First call of every compute entry point: selects the device and orders the library's compute stream after
// the work the caller has queued so far (buffers filled by kernels or async copies still in flight on the
// caller's stream are complete before any library kernel reads them).
int kmcf_enter(kmcf_comm *c);
int kmcf_comm_allreduce_sum(kmcf_comm *c, double *d_buf, int count);
// In-process groups on the peer-to-peer transport only (ranks = host threads sharing ONE GPU, a test device): the
// members meet on the host before an exchange whose kernels wait for each other on the device.  A member that is
// still inside a call that waits for the whole device (hipFree does) while another member's kernel already waits,
// on that device, for this member's next kernel, would otherwise block until the bounded wait expires (seen as
// intermittent time-outs of the set-up all-gathers).  Between processes -- one device each -- no such coupling exists.
int kmcf_group_rendezvous(kmcf_comm *c);
int kmcf_comm_send_recv_halo(kmcf_matrix *m);
int kmcf_comm_allgatherv_double(kmcf_comm *c, double *d_buf, const int *counts, const int *displs);
// the same on the COMM stream (RCCL: the halo communicator), for gathers that overlap with work on the compute stream;
// not for host-synchronous loopback groups (returns KMCF_ERR_STATE there: the caller keeps the in-order path)
int kmcf_comm_allgatherv_double_comm_stream(kmcf_comm *c, double *d_buf, const int *counts, const int *displs);
int kmcf_comm_allgatherv_int(kmcf_comm *c, int *d_buf, const int *counts, const int *displs);
// p2p.hip
int kmcf_p2p_create(kmcf_comm *c);
int kmcf_p2p_destroy(kmcf_comm *c);
int kmcf_p2p_set_peers_direct(kmcf_comm *c, char *const *bases);
char *kmcf_p2p_window(kmcf_comm *c);
bool kmcf_p2p_fits(kmcf_comm *c, size_t gather_bytes);
int kmcf_p2p_check(kmcf_comm *c);          // KMCF_ERR_COMM if a bounded wait expired (call after a synchronisation)
int kmcf_p2p_allreduce(kmcf_comm *c, double *d_buf, int count);
// finalize + all-reduce in ONE 1-block kernel: red[i] = sum over ranks of (sum of the partial arrays of part[i]);
// returns at once (on every rank alike) when skip_if_done and S->done
int kmcf_p2p_allreduce_parts(kmcf_comm *c, const kmcf_part4 *part, int count, kmcf_scalars *d_S, int skip_if_done);
int kmcf_p2p_allgatherv(kmcf_comm *c, void *d_buf, const int *counts, const int *displs, size_t elem, hipStream_t st = nullptr);
int kmcf_p2p_matrix_alloc(kmcf_matrix *m, int *land_off8, int *flag_off8, int *ack_off8, int *ll_off8, int *red_off8);
int kmcf_p2p_matrix_connect(kmcf_matrix *m, const std::vector<long long> &r_land8, const std::vector<long long> &r_flag8,
                            const std::vector<long long> &r_ack8, const std::vector<long long> &r_halo,
                            const std::vector<long long> &r_ll8, const std::vector<long long> &r_red8);
void kmcf_p2p_matrix_free(kmcf_matrix *m);
int kmcf_p2p_halo_exchange(kmcf_matrix *m);
// "direct" protocol (kmcf_p2p_dev.hpp): usable for this matrix?  (group on the p2p transport, per-row put table built)
bool kmcf_p2p_direct(const kmcf_matrix *m);
int kmcf_p2p_direct_put(kmcf_matrix *m, unsigned long long seq, bool skip_if_done);      // standalone put of d_p's sent rows (compute stream)
int kmcf_p2p_direct_ack(kmcf_matrix *m, unsigned long long seq, bool skip_if_done);      // standalone acknowledgement (compute stream)
// cgr.hip: register-resident PCG (one launch per solve) for matrices whose tiles are all resident at once
bool kmcf_cgr_usable(kmcf_matrix *m);
int kmcf_cgr_solve(kmcf_matrix *m, bool precond, double tol, int max_it, int fixed_iters, bool classic);
bool kmcf_pcg_resident_applies(kmcf_matrix *m);      // kmcf_cg.hip: the next kmcf_pcg_workspace call on m runs as a resident launch
int kmcf_cgr_check(kmcf_matrix *m);           // after the synchronisation: KMCF_ERR_STATE if a bounded wait expired
int kmcf_cgr_plan_info(kmcf_matrix *m, int *tpb, int *g1, int *nblocks);
void kmcf_cgr_free(kmcf_matrix *m);
// the reference's recurrence runs as a resident launch on one rank, wherever a resident launch fits (kmcf_cg.hip:
// pcg_workspace_run; KMCF_CGR_CLASSIC_TILES lowers the limit)
inline bool kmcf_cgr_classic_applies(const kmcf_matrix *m)
{
    static const int classic_tiles = getenv("KMCF_CGR_CLASSIC_TILES") ? atoi(getenv("KMCF_CGR_CLASSIC_TILES")) : 1024;
    return m->comm->nranks == 1 && !m->comm->force_collectives && m->n_sell_tiles <= classic_tiles;
}
// matrix.hip
int kmcf_matrix_build(kmcf_comm *c, int matrix_size, const int *counts, const int *displs,
                      const int *h_row_ptr, const int *h_col_global, const double *h_val,
                      const int *h_perm /* internal -> caller local row, or nullptr */, kmcf_matrix **out);
// caller-order vector -> internal order (dst internal) and back, on the compute stream
int kmcf_vec_in(kmcf_matrix *m, double *d_internal, const double *d_user);
int kmcf_vec_out(kmcf_matrix *m, double *d_user, const double *d_internal);

// ---------------------------------------------------------------- stream of the set-up kernels that read the caller's arrays
// Set-up functions (pattern builds, neighbour lists) launch the kernels that read the CALLER's coordinate arrays on the
// caller's own stream, not on the communicator's.  Found with in-process test groups (several host threads, each with
// its torch tensors uploaded a moment ago from pageable memory): a kernel on the communicator's non-blocking stream saw
// such an array as it was BEFORE the upload (zeros of the fresh allocation) although the upload call had returned, a
// synchronous copy of the same array at entry saw the uploaded values, and neither an event on the caller's stream nor
// a hipDeviceSynchronize() at entry changed that; the same kernel on the caller's stream, or after a synchronous
// device-to-device copy into a buffer of the library's, always saw the upload (tools/tmulti_stress.py: 5 wrong builds
// in 40 rounds before, 0 in 60 after).  Whatever orders an upload from pageable memory in this runtime, it is the stream
// the upload was issued on.
inline hipStream_t kmcf_setup_stream(const kmcf_comm *c) { return c->caller_stream; }

// ---------------------------------------------------------------- wavefront sum without the LDS crossbar
// v += __shfl_xor(v, off) for off = 32 ... 1 costs twelve ds_bpermute_b32 in a dependent chain (~130 cycles each way)
// and sits in front of, or behind, the stream of every CG kernel and SpMV.  The same butterfly through the VALU: gfx950's
// v_permlane32_swap / v_permlane16_swap for lane ^ 32 / ^ 16, DPP row_ror:8 for ^ 8, quad_perm for ^ 2 / ^ 1; for
// ^ 4 DPP has no pattern, but after the ^ 8 step lanes l and l ^ 8 hold the same bits, so row_ror:4 -- lane
// (l + 4) mod 16, which is l ^ 4 or l ^ 12 -- delivers the value lane l ^ 4 holds.  Operand for operand the sum of the
// shuffle loop: the results (and with them the device-order restatement the tests compare with) do not change.
// All 64 lanes must be active.  (tools/lab/xlane_lab.hip checks the patterns against __shfl_xor.)
#if defined(__HIPCC__)
typedef unsigned int kmcf_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double kmcf_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kmcf_lane_xor32(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const kmcf_u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = (threadIdx.x & 32) != 0;
    return __hiloint2double((int)(up ? b.x : b.y), (int)(up ? a.x : a.y));
}
__device__ __forceinline__ double kmcf_lane_xor16(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const kmcf_u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (threadIdx.x & 16) != 0;
    return __hiloint2double((int)(odd ? b.x : b.y), (int)(odd ? a.x : a.y));
}
__device__ __forceinline__ double kmcf_wave_sum64(double v)
{
    v += kmcf_lane_xor32(v);
    v += kmcf_lane_xor16(v);
    v += kmcf_dpp_f64<0x128>(v);      // row_ror:8  = lane ^ 8
    v += kmcf_dpp_f64<0x124>(v);      // row_ror:4  : the bits of lane ^ 4 (see above)
    v += kmcf_dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1] = lane ^ 2
    v += kmcf_dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
#endif
