// Register-resident Jacobi-PCG: a whole solve in ONE launch, for matrices small enough that every 256-row tile of the
// row-per-lane layout (kmcf_spmv.hip, spmv_sell_kernel) can have its own 256 lanes resident at the same time -- a rank's
// share of the 40 nm crossbar from four ranks on (<= 2048 tiles = 524 288 rows per GPU), the whole 5 nm device.
// Replaces, for those matrices, the loop of iterative_solver::conjugate_gradient_jacobi
// (dist_iterative/dist_conjugate_gradient.cpp:149-276) that kmcf_cg.hip runs as 2-3 kernels per iteration.
//
// Why.  A rank's eighth of the 40 nm matrix iterates in 14.7 us with two kernels per iteration, of which ~9 us are
// not bytes but kernel boundaries and the chain of dependent memory round trips every short kernel starts with
// ("scaling expectation").  MI355X has 128 MB of vector registers and 40 MB of LDS: the rank's matrix
// stream (2 B per entry), its five CG vectors and the tile's window map all FIT in the registers of the lanes that
// own the rows.  So nothing is re-read: lane t of tile c keeps row r0 + t's x, r, p, s, 1/diag, diagonal, its slice
// of the entry stream and its window columns in registers for the whole solve, and per iteration only two things
// cross block boundaries:
//   (1) z of the neighbouring tiles (the tile's window), and
//   (2) the iteration's dot products (single-reduction recurrence of Chronopoulos & Gear, as pcg1_loop: one reduction
//       point per iteration).
// Both go AROUND the L2s (which are per XCD and not coherent with each other inside a kernel) with agent-scope
// accesses, and both carry their own sequence number, so that no flag, fence or wait-for-stores sits between
// writing a value and somebody else using it ("LL" encoding, as RCCL's low-latency protocol does over xGMI):
//   a double travels as two 8-byte words, {low 32 bits | seq << 32} and {high 32 bits | seq << 32}; a reader polls
//   the two words until both carry the sequence number it expects.  8-byte aligned accesses are single-copy atomic,
//   so a word is either the old or the new one.  Buffers are double-buffered by the parity of the sequence number;
//   the reduction in every iteration keeps any block from running two versions ahead of a reader.
// Reduction: every block publishes its sums in a 128-byte line of its own; the leader of each group of g1 blocks adds
// its group (one lane per block, wavefront butterfly) and publishes the group's sums; every block adds the group sums
// (one lane per group, butterfly).  No read-modify-write on a shared word (256 same-address atomics cost ~10 us,
// tools/lab/gridbar_lab.hip), two memory hops (one where the grid is at most 256 blocks: every block reads every block's
// line itself), the same numbers in the same order in every block -- and in the tests' CPU restatement of this order,
// which must agree bit for bit (kmcf_matrix_sum_plan exports resident_tpb / resident_g1 for it).
// Measured before it was written: tools/lab/resident_lab.hip (881 tiles: 6.7 us per iteration with one block per CU).
//
// Every wait is bounded by the wall clock (KMCF_CGR_TIMEOUT_MS, default 4000): expiry sets an error word, every other
// wait returns at once, the host returns KMCF_ERR_STATE.  All blocks of the launch must be resident together: the
// grid is checked against the occupancy of the kernel (times the CUs, divided by KMCF_DEVICE_SHARE).
#include <cstring>

#include "kmcf_p2p_dev.hpp"

namespace {

typedef unsigned int sell_pair __attribute__((ext_vector_type(2)));   // 4 entries of the stream (kmcf_spmv.hip)
constexpr int CGR_W = 1 << KMCF_SLOT_BITS;     // window slots per dictionary value
constexpr int CGR_WQ = CGR_W / KMCF_BLOCK - 1; // outside columns per lane
constexpr int CGR_LINE = 16;                   // 8-byte words per 128-byte line
constexpr int CGR_NV = 3;                      // sums per reduction (gamma, delta, b.b)

}  // namespace

struct kmcf_cgr {
    int tpb = 0, nblocks = 0, g1 = 0, ngroups = 0;
    u64 *d_zll = nullptr;          // [2][2 * (n_loc + n_halo)]: LL words of z, by parity of the sequence number
    u64 *d_slot = nullptr;         // [2][nblocks][CGR_LINE]
    u64 *d_gslot = nullptr;        // [2][ngroups][CGR_LINE]
    unsigned int *d_seq = nullptr; // last sequence number the previous solve used (the kernel reads and advances it)
    int *d_err = nullptr, *h_err = nullptr;
    unsigned long long seq_bound = 1;   // host's upper bound of *d_seq (wrap protection)
    size_t zwords = 0;
};

namespace {

struct cgr_args {
    int n_tiles, nblocks, g1, ngroups;
    const int4 *tile4;
    const int2 *swave;
    const int *wcol;
    const sell_pair *stream;
    const double *dict, *diagv;
    double *r, *x;
    const double *dinv;              // nullptr: unpreconditioned
    const double *b_src;             // right-hand side (internal order); r receives the residual
    double *x_user;                  // nullptr, or: start guess in / solution out in the caller's order (through perm), instead of x
    const int *perm;                 // internal row -> caller's row (nullptr: identity)
    kmcf_scalars *S;
    u64 *zll;
    long long zwords;                // words per parity buffer
    u64 *slot, *gslot;
    unsigned int *seq;
    int *d_err, *h_err;
    long long timeout;
    int limit, check_tol;
    double tol2;
    int classic;                     // 1: the reference's recurrence (two reduction points per iteration); 0: the single-reduction form
    // A poll that comes too early costs a whole round trip (~1 us: the loop waits for its answer before asking again), so a
    // wavefront sleeps before the FIRST poll of a gather / of a reduction's collection: units of ~0.1 us (s_sleep 4), start
    // values (KMCF_CGR_DELAY, KMCF_CGR_RDELAY; default 6), then adapted per wavefront -- one more after a first poll that
    // failed, one less after adapt_delay in a row that did not (KMCF_CGR_ADAPT, default 16; 0: fixed).  Measured, us per
    // iteration, none / fixed / adapted: a rank's eighth of the 40 nm matrix 11.2 / 10.5 / 10.2, the 5 nm device 6.4 / 5.8 /
    // 5.7 (with units of 0.2 us long streaks suited the eighth and hurt the 5 nm device -- 10.2 / 7.3 at 16: its first polls
    // also fail for skew between blocks, which no delay cures, and a coarse delay creeps up; the best fixed pair per size reads
    // 10.1 / 5.6).
    int gather_delay, reduce_delay, adapt_delay;
    int sibling_lds;                 // 1: window columns owned by a sibling tile of the block come out of LDS (KMCF_CGR_SIB=0: through the granules)
    // groups of ranks (peer-to-peer transport; kmcf_p2p_dev.hpp): nranks == 1 -> everything below unused
    int nranks, rank, n_loc;
    const int *put_row, *putr_ptr;             // per internal row: its entry list (-1: not sent) | entries of a sent row
    u64 *const *putr_ll;                       // per entry: the granule (parity 0) of that row's value in the RECEIVER's zone
    const long long *putr_ll_stride;           // ... and the 8-byte words to its parity 1
    const u64 *halo_ll;                        // my granule zone: halo slot h, parity p at [p * halo_stride + 2 h]
    long long halo_stride;
    u64 *const *red_peer;                      // per rank: its reduction zone; mine: red_mine ([2][P2P_MAXR][P2P_FS] words)
    const u64 *red_mine;
};

// the same granules with system-scope accesses: a peer's device writes / reads them (fine-grained window memory)
__device__ __forceinline__ void ll_store_sys(u64 *p, double v, unsigned int seq)
{
    const u64 b = (u64)__double_as_longlong(v), s = (u64)seq << 32;
    __hip_atomic_store(p, (b & 0xffffffffull) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, (b >> 32) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and as ONE 16-byte store (the granule is 16-byte aligned): half the packets over a link, and lanes that own
// consecutive rows fill whole 64-byte requests.  Tear-safe like ll_store16: each 8-byte half carries the sequence number.
typedef unsigned int cgr_u4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_store_sys16(u64 *p, double v, unsigned int seq)
{
    const u64 b = (u64)__double_as_longlong(v);
    const cgr_u4s d = {(unsigned int)b, seq, (unsigned int)(b >> 32), seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ bool ll_try_sys(const u64 *p, unsigned int seq, double &v)
{
    const u64 a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)(a >> 32) != seq || (unsigned int)(b >> 32) != seq) return false;
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return true;
}


__device__ __forceinline__ u64 cgr_ld(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cgr_st(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void ll_store(u64 *p, double v, unsigned int seq)
{
    const u64 b = (u64)__double_as_longlong(v), s = (u64)seq << 32;
    cgr_st(p, (b & 0xffffffffull) | s);
    cgr_st(p + 1, (b >> 32) | s);
}
__device__ __forceinline__ bool ll_try(const u64 *p, unsigned int seq, double &v)
{
    const u64 a = cgr_ld(p), b = cgr_ld(p + 1);
    if ((unsigned int)(a >> 32) != seq || (unsigned int)(b >> 32) != seq) return false;
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return true;
}

// The same two words as ONE 16-byte access through a buffer descriptor with the sc1 bit (aux = 16): consecutive lanes
// that read consecutive values are merged by the texture path into one request per run (8-byte accesses of a 16-byte
// pitch are not: one request per lane and word; the z gather of a rank's eighth is bound by that request count).
// Tear-safe without any assumption about 16-byte atomicity: each 8-byte half carries the sequence number.
typedef unsigned int cgr_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_store16(__amdgpu_buffer_rsrc_t rs, unsigned int byte_off, double v, unsigned int seq)
{
    const u64 b = (u64)__double_as_longlong(v);
    const cgr_u4 d = {(unsigned int)b, seq, (unsigned int)(b >> 32), seq};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ bool ll_try16(__amdgpu_buffer_rsrc_t rs, unsigned int byte_off, unsigned int seq, double &v)
{
    const cgr_u4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    if (d.y != seq || d.w != seq) return false;
    v = __longlong_as_double((long long)((u64)d.x | ((u64)d.z << 32)));
    return true;
}

// ... a granule of the rank's halo zone (fine-grained window memory, written by a peer's device): the same 16-byte load
// at system scope (sc0 sc1) through a descriptor of the zone -- one request instead of two, a 32-bit offset instead of a
// 64-bit address per lane
__device__ __forceinline__ bool ll_try16_sys(__amdgpu_buffer_rsrc_t rs, unsigned int byte_off, unsigned int seq, double &v)
{
    const cgr_u4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 17);
    if (d.y != seq || d.w != seq) return false;
    v = __longlong_as_double((long long)((u64)d.x | ((u64)d.z << 32)));
    return true;
}

struct cgr_wait {
    long long t0, timeout;
    int *d_err, *h_err;
    int spins;
    bool failed;
    // called after an unsuccessful poll: true = give up (this wait or somebody else's has expired)
    __device__ __forceinline__ bool give_up(int code)
    {
        if ((++spins & 15) == 0) {
            if (__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { failed = true; return true; }
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(d_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                failed = true;
                return true;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        return false;
    }
};

// One solve.  Block b owns tiles b TPB ... b TPB + TPB - 1 (256 threads each); blocks of one launch are all resident.
// (REC: the recurrence -- 0 single-reduction, 1 the reference's; MULTI: a group of ranks.  Compile-time, so that an
// instance carries only its own paths: the one-rank single-reduction instance with four tiles per block keeps 32 instead
// of 124 bytes per lane in scratch under its 128-register bound and iterates 5 % faster -- 11.86 -> 11.23 us on a rank's
// eighth of the 40 nm matrix, 6.77 -> 6.46 on the 5 nm device, same box, alternating.)
template <int NQ, int ND, int TPB, int REC, bool MULTI>
__global__ __launch_bounds__(KMCF_BLOCK * TPB) void cgr_kernel(const cgr_args A)
{
    extern __shared__ double cgr_lds[];
    double *red = cgr_lds + (size_t)TPB * ND * CGR_W;      // [CGR_NV][16] wavefront sums
    double *bc = red + CGR_NV * 16;                        // [CGR_NV] reduced sums, for every thread; flat reduction: [CGR_NV][4] wave sums
    double *zblk = bc + 16;                                // [TPB * 256] the block's own rows of the vector being multiplied
    const int tid = threadIdx.x, sub = tid >> 8, t = tid & 255, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), gw = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *xs = cgr_lds + (size_t)sub * ND * CGR_W;
    const int c = blockIdx.x * TPB + sub;
    const bool active = c < A.n_tiles;
    const int4 d = active ? A.tile4[c] : make_int4(0, 0, 0, 0);          // (first row, rows, first window slot, outside columns)
    const int2 sw = active ? A.swave[c * 4 + wv] : make_int2(0, 0);      // (first 8-byte group of the wave's stream, steps)
    const bool has_row = t < d.y;
    const int row = d.x + t;
    int wc[CGR_WQ];
#pragma unroll
    for (int q = 0; q < CGR_WQ; ++q) wc[q] = (q * KMCF_BLOCK + t < d.w) ? A.wcol[d.z + q * KMCF_BLOCK + t] : -1;
    // Window columns that are rows of a sibling tile of this block (consecutive tiles are neighbours in space: a third
    // of a tile's outside columns, tools/lab/window_stats.py) come out of LDS, where every lane leaves its own value
    // (zblk, indexed by row - the block's first row), not through the granules: fewer requests around the L2s.
    int rb0 = 0, rb1 = 0;                          // the block's rows [rb0, rb1)
    if (TPB > 1) {
        const int cf = blockIdx.x * TPB, cl = min(cf + TPB, A.n_tiles) - 1;
        const int4 df = A.tile4[cf], dl = A.tile4[cl];
        rb0 = __builtin_amdgcn_readfirstlane(df.x);
        rb1 = __builtin_amdgcn_readfirstlane(dl.x + dl.y);
    }
    const int nwq = __builtin_amdgcn_readfirstlane((d.w + KMCF_BLOCK - 1) / KMCF_BLOCK);   // 256-slot stretches of the window that hold outside columns
#pragma unroll
    for (int q = 0; q < CGR_WQ; ++q)               // (the others are written once, here)
#pragma unroll
        for (int k = 0; k < ND; ++k) xs[k * CGR_W + KMCF_BLOCK + q * KMCF_BLOCK + t] = 0.0;
    bool sib[CGR_WQ];
#pragma unroll
    for (int q = 0; q < CGR_WQ; ++q) sib[q] = TPB > 1 && A.sibling_lds && wc[q] >= rb0 && wc[q] < rb1;
    sell_pair pk[NQ];
    {
        const sell_pair *sp = A.stream + sw.x + lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q) pk[q] = q < sw.y ? sp[q * 64] : sell_pair{0u, 0u};
    }
    double dv[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) dv[k] = A.dict[k];
    const double dg = has_row ? A.diagv[row] : 0.0;
    const double di = has_row ? (A.dinv ? A.dinv[row] : 1.0) : 0.0;
    const double b_i = has_row ? A.b_src[row] : 0.0;
    const int urow = has_row && A.x_user ? (A.perm ? A.perm[row] : row) : 0;
    double x = has_row ? (A.x_user ? A.x_user[urow] : A.x[row]) : 0.0;
    unsigned int seq = *A.seq;                   // (read by every block before block 0 overwrites it at the very end)
    // Every publication and every reduction takes the next sequence number (unique over the life of the buffers); which
    // of its two buffers a publication / a reduction uses alternates on its own count (zpar / rpar), the same in every
    // block and on every rank: both recurrences below publish and reduce in a fixed pattern.
    int zpar = 1, rpar = 1;
    cgr_wait W{wall_clock64(), A.timeout, A.d_err, A.h_err, 0, false};
#ifdef KMCF_CGR_PROFILE
    long long tp_gather = 0, tp_row = 0, tp_red1 = 0, tp_red2 = 0, tp_red3 = 0, tp_mark = 0;
#define CGR_T0() tp_mark = wall_clock64()
#define CGR_T(acc) do { const long long n_ = wall_clock64(); acc += n_ - tp_mark; tp_mark = n_; } while (0)
#else
#define CGR_T0() do { } while (0)
#define CGR_T(acc) do { } while (0)
#endif

    // z (or x0) of this lane's row, version `seq`, to whoever has the row in its window
    const __amdgpu_buffer_rsrc_t zrs[2] = {__builtin_amdgcn_make_buffer_rsrc(A.zll, 0, (int)(A.zwords * 8), 0x00020000),
                                           __builtin_amdgcn_make_buffer_rsrc(A.zll + A.zwords, 0, (int)(A.zwords * 8), 0x00020000)};
    const __amdgpu_buffer_rsrc_t srs[2] = {__builtin_amdgcn_make_buffer_rsrc(A.slot, 0, A.nblocks * 128, 0x00020000),
                                           __builtin_amdgcn_make_buffer_rsrc(A.slot + (size_t)A.nblocks * CGR_LINE, 0, A.nblocks * 128, 0x00020000)};
    constexpr bool multi = MULTI;
    // (the halo zone: halo slot h, parity p at byte p * 8 halo_stride + 16 h)
    const __amdgpu_buffer_rsrc_t hrs[2] = {
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(A.halo_ll), 0, multi ? (int)(A.halo_stride * 8) : 0, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(A.halo_ll) + (multi ? A.halo_stride : 0), 0, multi ? (int)(A.halo_stride * 8) : 0, 0x00020000)};
    int put0 = 0, put1 = 0;                        // this row's entries in the put table (a row a neighbour rank needs)
    if (multi && has_row) {
        const int pr = A.put_row[row];
        if (pr >= 0) { put0 = A.putr_ptr[pr]; put1 = A.putr_ptr[pr + 1]; }
    }
    auto publish = [&](double v) {
        ++seq;
        zpar ^= 1;
        if (has_row) {
            if (zpar) ll_store16(zrs[1], 16u * (unsigned int)row, v, seq);
            else ll_store16(zrs[0], 16u * (unsigned int)row, v, seq);
            for (int e = put0; e < put1; ++e) ll_store_sys16(A.putr_ll[e] + zpar * A.putr_ll_stride[e], v, seq);
        }
    };
    // y_row = sum of the row's products with version `seq` of the vector whose own entry is `own`
    // (trail: a barrier behind the row sums.  xs is written again by the NEXT product; every reduction in between -- and,
    // with several tiles per block, the next product's own first barrier -- already orders those writes behind these reads,
    // so only a product that is followed directly by another one in a one-tile block asks for it)
    int gdel = A.gather_delay, gstreak = 0, rdel = A.reduce_delay, rstreak = 0, kdel = A.reduce_delay, kstreak = 0;
    auto spmv = [&](double own, bool trail = false) -> double {
        const __amdgpu_buffer_rsrc_t zb = zpar ? zrs[1] : zrs[0];
        CGR_T0();
        double g[CGR_WQ];
        bool need[CGR_WQ];
#pragma unroll
        for (int q = 0; q < CGR_WQ; ++q) { g[q] = 0.0; need[q] = wc[q] >= 0 && !sib[q]; }
        if (TPB > 1 && has_row) zblk[row - rb0] = own;
        for (int k = 0; k < gdel; ++k) __builtin_amdgcn_s_sleep(4);
        int passes = 0;
        while (true) {
            ++passes;
            bool all = true;
#pragma unroll
            for (int q = 0; q < CGR_WQ; ++q)
                if (need[q]) {
                    bool got;
                    if (multi && wc[q] >= A.n_loc) got = ll_try16_sys(zpar ? hrs[1] : hrs[0], 16u * (unsigned int)(wc[q] - A.n_loc), seq, g[q]);
                    else got = ll_try16(zb, 16u * (unsigned int)wc[q], seq, g[q]);
                    if (got) need[q] = false;
                    else all = false;
                }
            if (all) break;
            if (W.give_up(11)) {
#ifdef KMCF_CGR_DEBUG
                for (int q = 0; q < CGR_WQ; ++q)
                    if (need[q] && (tid & 63) == __ffsll((long long)__ballot(need[q])) - 1) {
                        const bool hal = multi && wc[q] >= A.n_loc;
                        const u64 *pp = hal ? A.halo_ll + zpar * A.halo_stride + 2 * (size_t)(wc[q] - A.n_loc) : A.zll + (size_t)zpar * A.zwords + 2 * (size_t)wc[q];
                        printf("cgr gather timeout: rank %d block %d tid %d seq %u col %d (%s, n_loc %d) words %llx %llx\n", A.rank, (int)blockIdx.x, tid, seq, wc[q],
                               hal ? "halo" : "own", A.n_loc, (unsigned long long)__hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                               (unsigned long long)__hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                    }
#endif
                break;
            }
            asm volatile("" ::: "memory");                 // (the next pass loads again)
        }
        if (A.adapt_delay) {                               // (wavefront-uniform)
            if (__ballot(passes > 1) != 0ull) { gdel = min(gdel + 1, 24); gstreak = 0; }
            else if (++gstreak >= A.adapt_delay) { gdel = max(gdel - 1, 0); gstreak = 0; }
        }
        CGR_T(tp_gather);
        if (TPB > 1) {
            __syncthreads();                               // (every lane's own value is in zblk)
#pragma unroll
            for (int q = 0; q < CGR_WQ; ++q)
                if (sib[q]) g[q] = zblk[wc[q] - rb0];
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) xs[k * CGR_W + t] = dv[k] * own;                 // (lanes without a row: own = 0)
#pragma unroll
        for (int q = 0; q < CGR_WQ; ++q) {
            if (q >= nwq) break;                           // (tile-uniform: the stretches past the tile's window keep their 0.0 -- the padding target, slot W - 1, among them)
#pragma unroll
            for (int k = 0; k < ND; ++k) xs[k * CGR_W + KMCF_BLOCK + q * KMCF_BLOCK + t] = dv[k] * g[q];   // (past the window: 0.0, the padding target)
        }
        __syncthreads();
        const char *base = reinterpret_cast<const char *>(xs);
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < sw.y) {
                const sell_pair e = pk[q];
                const double a0 = *reinterpret_cast<const double *>(base + (e.x & 0xffffu));
                const double a1 = *reinterpret_cast<const double *>(base + (e.x >> 16));
                const double a2 = *reinterpret_cast<const double *>(base + (e.y & 0xffffu));
                const double a3 = *reinterpret_cast<const double *>(base + (e.y >> 16));
                s += a0; s += a1; s += a2; s += a3;
            }
        }
        s += dg * own;
        if (trail) __syncthreads();
        CGR_T(tp_row);
        return has_row ? s : 0.0;
    };
    // sums over all rows of up to CGR_NV per-lane values, number `rs`; the same value in every thread of every block
    // (returns false -- in every thread of the block alike -- when a wait of this block has given up)
    auto reduce_rank = [&](int nv, double v0, double v1, double v2, double (&out)[CGR_NV], unsigned int rs) -> bool {
        CGR_T0();
        v0 = kmcf_wave_sum64(v0); v1 = kmcf_wave_sum64(v1);
        if (nv > 2) v2 = kmcf_wave_sum64(v2);
        if (lane == 0) { red[gw] = v0; red[16 + gw] = v1; red[32 + gw] = v2; }
        __syncthreads();
        u64 *sl = A.slot + ((size_t)rpar * A.nblocks + blockIdx.x) * CGR_LINE;
        if (tid < nv) {                                    // tile sums (w0 + w1) + (w2 + w3), tiles in pairs
            const double *w = red + 16 * tid;
            double tsum[TPB];
#pragma unroll
            for (int j = 0; j < TPB; ++j) tsum[j] = (w[4 * j] + w[4 * j + 1]) + (w[4 * j + 2] + w[4 * j + 3]);
            double bsum = tsum[0];
            if (TPB == 2) bsum = tsum[0] + tsum[1];
            if (TPB == 4) bsum = (tsum[0] + tsum[1]) + (tsum[2] + tsum[3]);
            if (A.g1 == 0) ll_store16(rpar ? srs[1] : srs[0], 128u * blockIdx.x + 16u * tid, bsum, rs);
            else ll_store(sl + 2 * tid, bsum, rs);
        }
        CGR_T(tp_red1);
        if (A.g1 != 0 || gw < 4)
            for (int k = 0; k < rdel; ++k) __builtin_amdgcn_s_sleep(4);
        if (A.g1 == 0) {
            // flat (<= 256 blocks): ONE hop -- every block reads every block's line itself, lane l of wave w that of block
            // 64 w + l; a wave adds its 64 (butterfly), the four wave sums as (w0 + w1) + (w2 + w3)
            if (gw < 4) {
                const int q = 64 * gw + lane;
                const bool mine = q < A.nblocks;
                const __amdgpu_buffer_rsrc_t sb = rpar ? srs[1] : srs[0];
                double g[CGR_NV] = {0.0, 0.0, 0.0};
                bool need[CGR_NV] = {mine, mine, mine && nv > 2};
                int passes = 0;
                while (true) {
                    ++passes;
                    bool all = true;
#pragma unroll
                    for (int i = 0; i < CGR_NV; ++i)
                        if (need[i]) {
                            if (ll_try16(sb, 128u * (unsigned int)q + 16u * i, rs, g[i])) need[i] = false;
                            else all = false;
                        }
                    if (all || W.give_up(12)) break;
                    asm volatile("" ::: "memory");
                }
                if (A.adapt_delay) {
                    if (__ballot(passes > 1) != 0ull) { rdel = min(rdel + 1, 24); rstreak = 0; }
                    else if (++rstreak >= A.adapt_delay) { rdel = max(rdel - 1, 0); rstreak = 0; }
                }
                g[0] = kmcf_wave_sum64(g[0]); g[1] = kmcf_wave_sum64(g[1]);
                if (nv > 2) g[2] = kmcf_wave_sum64(g[2]);
                if (lane == 0) { bc[gw] = g[0]; bc[4 + gw] = g[1]; bc[8 + gw] = g[2]; }
            }
            const int bad = __syncthreads_or(W.failed ? 1 : 0);
            CGR_T(tp_red3);
#pragma unroll
            for (int i = 0; i < CGR_NV; ++i) out[i] = (bc[4 * i] + bc[4 * i + 1]) + (bc[4 * i + 2] + bc[4 * i + 3]);
            // (bc / red are written again by the next reduction -- behind ITS first barrier, which every wavefront reaches
            // only after these reads; a group of ranks writes bc right away, in the rank stage)
            if (multi) __syncthreads();
            return bad == 0;
        }
        if (gw == 0 && blockIdx.x % A.g1 == 0) {           // leader of a group: one lane per block of the group
            const int q = blockIdx.x + lane;
            const bool mine = lane < A.g1 && q < A.nblocks;
            const u64 *sq = A.slot + ((size_t)rpar * A.nblocks + (mine ? q : blockIdx.x)) * CGR_LINE;
            double g[CGR_NV] = {0.0, 0.0, 0.0};
            bool need[CGR_NV] = {mine, mine, mine && nv > 2};
            while (true) {
                bool all = true;
#pragma unroll
                for (int i = 0; i < CGR_NV; ++i)
                    if (need[i]) {
                        if (ll_try(sq + 2 * i, rs, g[i])) need[i] = false;
                        else all = false;
                    }
                if (all || W.give_up(12)) break;
            }
            g[0] = kmcf_wave_sum64(g[0]); g[1] = kmcf_wave_sum64(g[1]);
            if (nv > 2) g[2] = kmcf_wave_sum64(g[2]);
            u64 *gs = A.gslot + ((size_t)rpar * A.ngroups + blockIdx.x / A.g1) * CGR_LINE;
            if (lane < nv) ll_store(gs + 2 * lane, lane == 0 ? g[0] : (lane == 1 ? g[1] : g[2]), rs);
            CGR_T(tp_red2);
        }
        if (gw == 1) {                                     // every block: one lane per group
            const bool mine = lane < A.ngroups;
            const u64 *gq = A.gslot + ((size_t)rpar * A.ngroups + (mine ? lane : 0)) * CGR_LINE;
            double g[CGR_NV] = {0.0, 0.0, 0.0};
            bool need[CGR_NV] = {mine, mine, mine && nv > 2};
            while (true) {
                bool all = true;
#pragma unroll
                for (int i = 0; i < CGR_NV; ++i)
                    if (need[i]) {
                        if (ll_try(gq + 2 * i, rs, g[i])) need[i] = false;
                        else all = false;
                    }
                if (all || W.give_up(13)) break;
            }
            g[0] = kmcf_wave_sum64(g[0]); g[1] = kmcf_wave_sum64(g[1]);
            if (nv > 2) g[2] = kmcf_wave_sum64(g[2]);
            if (lane == 0) { bc[0] = g[0]; bc[1] = g[1]; bc[2] = g[2]; }
        }
        const int bad = __syncthreads_or(W.failed ? 1 : 0);
        CGR_T(tp_red3);
        out[0] = bc[0]; out[1] = bc[1]; out[2] = bc[2];
        if (multi) __syncthreads();                        // (as above)
        return bad == 0;
    };

    // ... and over the ranks of a group: this rank's sums (the same in every block) go to a line per rank in every peer's
    // reduction zone; every block adds the P lines of its own zone, one lane per rank (butterfly): the same numbers in the
    // same order on every rank
    auto reduce = [&](int nv, double v0, double v1, double v2, double (&out)[CGR_NV]) -> bool {
        ++seq;
        rpar ^= 1;
        const unsigned int rs = seq;
        const bool ok = reduce_rank(nv, v0, v1, v2, out, rs);
        if (!multi) return ok;
        const size_t par = (size_t)rpar;
        if (blockIdx.x == 0 && gw == 0 && lane < A.nranks) {
            u64 *dst = A.red_peer[lane] + (par * P2P_MAXR + A.rank) * P2P_FS;
            ll_store_sys(dst, out[0], rs);
            ll_store_sys(dst + 2, out[1], rs);
            if (nv > 2) ll_store_sys(dst + 4, out[2], rs);
        }
        if (gw == 1) {
            const bool mine = lane < A.nranks;
            const u64 *src = A.red_mine + (par * P2P_MAXR + (mine ? lane : 0)) * P2P_FS;
            double g[CGR_NV] = {0.0, 0.0, 0.0};
            bool need[CGR_NV] = {mine, mine, mine && nv > 2};
            for (int k = 0; k < kdel; ++k) __builtin_amdgcn_s_sleep(4);      // (the peers' lines: delayed and adapted like the local polls)
            int passes = 0;
            while (true) {
                ++passes;
                bool all = true;
#pragma unroll
                for (int i = 0; i < CGR_NV; ++i)
                    if (need[i]) {
                        if (ll_try_sys(src + 2 * i, rs, g[i])) need[i] = false;
                        else all = false;
                    }
                if (all || W.give_up(14)) break;
            }
            if (A.adapt_delay) {
                if (__ballot(passes > 1) != 0ull) { kdel = min(kdel + 1, 48); kstreak = 0; }
                else if (++kstreak >= A.adapt_delay) { kdel = max(kdel - 1, 0); kstreak = 0; }
            }
            g[0] = kmcf_wave_sum64(g[0]); g[1] = kmcf_wave_sum64(g[1]);
            if (nv > 2) g[2] = kmcf_wave_sum64(g[2]);
            if (lane == 0) { bc[0] = g[0]; bc[1] = g[1]; bc[2] = g[2]; }
        }
        const int bad = __syncthreads_or(W.failed ? 1 : 0);
        out[0] = bc[0]; out[1] = bc[1]; out[2] = bc[2];
        __syncthreads();
        return ok && bad == 0;
    };

    // ---- r = b - A x0 ; z = r .* dinv ; (r, z) ; b.b                          (dist_conjugate_gradient.cpp:178-213)
    publish(x);
    const double ax = spmv(has_row ? x : 0.0, TPB == 1 && REC != 1);     // (single-reduction form: the next product follows directly)
    double r = b_i + (-1.0) * ax;
    double z = r * di;
    double gp = r * z, bbp = b_i * b_i;
    double p = 0.0, s = 0.0;
    double bb = 0.0, g_old = 0.0, a_old = 0.0, rz_last = 0.0, pAp = 0.0;
    int iters = 0, done = 0;
    bool ok = true;
    if (REC == 1) {
        // ---- the reference's recurrence and operation order (:217-266; kmcf_cg.hip: cg_p_kernel, cg_xr_kernel): two
        // reduction points per iteration -- p.Ap, then r.z
        double sums[CGR_NV];
        ok = reduce(2, gp, bbp, 0.0, sums);
        double rz = sums[0], rz_prev = 0.0;
        bb = sums[1];
        for (int k = 1; k <= A.limit && ok; ++k) {
            const bool first = k == 1;
            const bool go = A.check_tol ? (rz / bb > A.tol2) : true;                       // :217
            rz_last = rz;
            if (!go) { done = 1; break; }
            ++iters;
            if (first) p = z;                                                                  // :226 dcopy(z -> p)
            else { const double beta = rz / rz_prev; p = beta * p + z; }                      // :220-222 dscal, daxpy
            publish(p);
            const double Ap = spmv(has_row ? p : 0.0);
            ok = reduce(2, p * Ap, 0.0, 0.0, sums);
            if (!ok) break;
            pAp = sums[0];
            const double a = rz / pAp, na = -a;                                               // :240
            x = x + a * p;                                                                     // :243
            r = r + na * Ap;                                                                   // :246
            z = r * di;
            gp = r * z;
            ok = reduce(2, gp, 0.0, 0.0, sums);
            rz_prev = rz;
            rz = sums[0];
            g_old = rz_prev; a_old = a;
        }
        if (!done && ok) rz_last = rz;          // the loop condition once more after the last iteration (:217, :273)
    } else {
        // ---- the single-reduction form (kmcf_cg.hip: cg1_update_kernel): gamma = (r, z) and delta = (A z, z) at one point
        publish(z);
        for (int k = 1; k <= A.limit; ++k) {
            const bool first = k == 1;
            const double w = spmv(has_row ? z : 0.0);
            double sums[CGR_NV];
            ok = reduce(first ? 3 : 2, gp, z * w, bbp, sums);
            if (!ok) break;
            const double gamma = sums[0], delta = sums[1];
            if (first) bb = sums[2];
            const bool go = A.check_tol ? (gamma / bb > A.tol2) : true;
            rz_last = gamma;
            if (!go) { done = 1; break; }
            double beta = 0.0, alpha;
            if (first) alpha = gamma / delta;
            else {
                beta = gamma / g_old;
                alpha = gamma / (delta - beta * gamma / a_old);
            }
            g_old = gamma; a_old = alpha; pAp = delta;
            ++iters;
            const double na = -alpha;
            s = first ? w : w + beta * s;
            r = r + na * s;
            const double zn = r * di;
            publish(zn);                 // (on its way before the rest of the update)
            p = first ? z : z + beta * p;
            x = x + alpha * p;
            z = zn;
            gp = r * z;
        }
        if (!done && ok) {               // the loop condition once more after the last iteration (:217, :273): r.z only
            double sums[CGR_NV];
            reduce(2, gp, 0.0, 0.0, sums);
            rz_last = sums[0];
        }
    }
#ifdef KMCF_CGR_PROFILE
    if ((blockIdx.x == 0 || blockIdx.x == 17 || blockIdx.x == A.nblocks - 1) && (tid == 0 || tid == 64 || tid == 700))
        printf("cgr profile block %d thread %d: %d iterations, ticks per iteration: gather %.1f row %.1f red-publish %.1f red-leader %.1f red-wait %.1f (tick = 10 ns)\n",
               (int)blockIdx.x, tid, iters, (double)tp_gather / max(iters, 1), (double)tp_row / max(iters, 1), (double)tp_red1 / max(iters, 1),
               (double)tp_red2 / max(iters, 1), (double)tp_red3 / max(iters, 1));
#endif
    if (has_row) {
        if (A.x_user) A.x_user[urow] = x;
        else A.x[row] = x;
        A.r[row] = r;
    }
    if (blockIdx.x == 0 && tid == 0) {
        kmcf_scalars h;
        h.bb = bb; h.rz[0] = h.rz[1] = g_old; h.rz_last = rz_last; h.pAp = pAp;
        h.red[0] = h.red[1] = h.red[2] = h.red[3] = 0.0;
        h.alpha[0] = h.alpha[1] = a_old;
        h.done = done; h.iters = iters; h.x_pending = 0; h.stop_k = 0; h.xa = 0.0;
        *A.S = h;
        *A.seq = seq + 1;
    }
}

struct cgr_launch_info { int per_cu; size_t lds; };

template <int NQ, int ND, int TPB>
int cgr_run(const cgr_args &A, bool launch, hipStream_t st, cgr_launch_info *info)
{
    const size_t lds = ((size_t)TPB * ND * CGR_W + CGR_NV * 16 + 16 + (TPB > 1 ? TPB * KMCF_BLOCK : 0)) * sizeof(double);
    typedef void (*kern_t)(const cgr_args);
    // [recurrence][group of ranks]
    const kern_t kerns[2][2] = {{cgr_kernel<NQ, ND, TPB, 0, false>, cgr_kernel<NQ, ND, TPB, 0, true>},
                                {cgr_kernel<NQ, ND, TPB, 1, false>, cgr_kernel<NQ, ND, TPB, 1, true>}};
    if (info) {                                    // planning: what fits whichever instance a solve will launch
        info->lds = lds;
        info->per_cu = 1 << 30;
        for (int r = 0; r < 2; ++r)
            for (int g = 0; g < 2; ++g) {
                int per_cu = 0;
                KMCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kerns[r][g]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                KMCF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kerns[r][g], KMCF_BLOCK * TPB, lds));
                info->per_cu = std::min(info->per_cu, per_cu);
            }
    }
    if (launch) {
        const kern_t kern = kerns[A.classic == 1 ? 1 : 0][A.nranks > 1 ? 1 : 0];
        KMCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<A.nblocks, KMCF_BLOCK * TPB, lds, st>>>(A);
        KMCF_HIP(hipGetLastError());
    }
    return KMCF_OK;
}

template <int NQ, int ND>
int cgr_run_tpb(int tpb, const cgr_args &A, bool launch, hipStream_t st, cgr_launch_info *info)
{
    switch (tpb) {
        case 1: return cgr_run<NQ, ND, 1>(A, launch, st, info);
        case 2: return cgr_run<NQ, ND, 2>(A, launch, st, info);
        default: return cgr_run<NQ, ND, 4>(A, launch, st, info);
    }
}

int cgr_run_any(const kmcf_matrix *m, int tpb, const cgr_args &A, bool launch, hipStream_t st, cgr_launch_info *info)
{
    const int nd = m->dict_n <= 2 ? 2 : 3;
    switch (m->sell_nq * 10 + nd) {          // the instantiated (steps, dictionary size) pairs of the row-per-lane kernel
        case 82: return cgr_run_tpb<8, 2>(tpb, A, launch, st, info);
        case 83: return cgr_run_tpb<8, 3>(tpb, A, launch, st, info);
        case 132: return cgr_run_tpb<13, 2>(tpb, A, launch, st, info);
        case 133: return cgr_run_tpb<13, 3>(tpb, A, launch, st, info);
        case 162: return cgr_run_tpb<16, 2>(tpb, A, launch, st, info);
        default: return cgr_run_tpb<16, 3>(tpb, A, launch, st, info);
    }
}

int cgr_mode()
{
    const char *e = getenv("KMCF_CG_RESIDENT");
    return e ? atoi(e) : 1;                  // 0 off, 1 where a matrix qualifies
}

}  // namespace

bool kmcf_sell_coded_active(const kmcf_matrix *m);   // kmcf_spmv.hip
int kmcf_sell_ready(kmcf_matrix *m);                  // ... its stream holds the current value codes

void kmcf_cgr_free(kmcf_matrix *m)
{
    kmcf_cgr *g = m->cgr;
    if (!g) return;
    if (g->d_zll) hipFree(g->d_zll);
    if (g->d_slot) hipFree(g->d_slot);
    if (g->d_gslot) hipFree(g->d_gslot);
    if (g->d_seq) hipFree(g->d_seq);
    if (g->d_err) hipFree(g->d_err);
    if (g->h_err) hipHostFree(g->h_err);
    delete g;
    m->cgr = nullptr;
}

// Plans the resident solve of this matrix (once; the answer is kept): tiles per block and grid such that every block is
// resident, reduction groups, buffers.  cgr->tpb == 0 afterwards: the matrix does not qualify.
static int cgr_plan(kmcf_matrix *m)
{
    if (m->cgr) return KMCF_OK;
    kmcf_cgr *g = new kmcf_cgr();
    m->cgr = g;
    const kmcf_comm *c = m->comm;
    // short rows only, no tunnel block, coded row-per-lane stream with lane t = row r0 + t; one rank, or a group on the
    // peer-to-peer transport whose every rank qualifies (agreed below)
    const bool group = c->nranks > 1;
    bool ok = !(c->force_collectives || m->sub || m->n_short != m->n_loc || m->n_loc == 0 || !m->sell_ok || !m->sell_ident ||
                m->n_sell_tiles <= 0 || m->sell_lw != KMCF_SLOT_BITS || !kmcf_sell_coded_active(m));
    if (group) ok = ok && c->p2p_active && m->p2p && m->p2p->d_putr_ll && m->n_long_items == 0;
    else ok = ok && m->n_halo == 0;
    if (!ok && !group) return KMCF_OK;
    int cus = 0;
    KMCF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    // (the ranks of an in-process group -- a test device -- share ONE GPU: their launches wait for each other, so all of
    // them must be resident together)
    const int share = std::max(kmcf_device_share(), c->group ? c->nranks : 1);
    cgr_args A{};
    int pick = 0;
    const int forced = getenv("KMCF_CGR_TPB") ? atoi(getenv("KMCF_CGR_TPB")) : 0;
    // Tiles per block: the smallest that keeps the grid within 256 blocks -- the flat one-hop reduction needs that, and
    // more blocks mean more CUs whose LDS pipes share the row sums (measured, us per iteration at 1 / 2 / 4 tiles per
    // block: 5 nm device, 286 tiles: 8.1 / 6.8 / 8.3; a rank's eighth of the 40 nm matrix, 881 tiles: 14.5 / 14.5 / 11.4);
    // beyond 1024 tiles: four per block and the two-hop reduction.
    for (int pass = 0; pass < 2 && !pick && ok; ++pass)
        for (int tpb : {1, 2, 4}) {
            if (forced && tpb != forced) continue;
            const int nb = (m->n_sell_tiles + tpb - 1) / tpb;
            if (pass == 0 && !forced && nb > 256) continue;
            if (pass == 1 && !forced && tpb != 4) continue;
            cgr_launch_info info{0, 0};
            KMCF_TRY(cgr_run_any(m, tpb, A, false, nullptr, &info));
            if (info.per_cu >= 1 && nb <= (long long)info.per_cu * cus / share) { pick = tpb; break; }
        }
    if (group) {
        // every rank runs the resident launch or none does (the launches wait for each other): the ranks that could, counted
        double *d_v = c->d_scratch;
        const double mine = pick ? 1.0 : 0.0;
        double all = 0.0;
        KMCF_HIP(hipMemcpyAsync(d_v, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
        KMCF_HIP(hipStreamSynchronize(c->stream));
        KMCF_TRY(kmcf_comm_allreduce_sum(const_cast<kmcf_comm *>(c), d_v, 1));
        KMCF_HIP(hipStreamSynchronize(c->stream));
        KMCF_TRY(kmcf_p2p_check(const_cast<kmcf_comm *>(c)));
        KMCF_HIP(hipMemcpy(&all, d_v, sizeof(double), hipMemcpyDeviceToHost));
        if (all != (double)c->nranks) pick = 0;
    }
    if (getenv("KMCF_TRACE")) fprintf(stderr, "cgr_plan rank %d: ok %d tiles %d forced %d pick %d share %d\n", c->rank, (int)ok, m->n_sell_tiles, forced, pick, share);
    if (!pick) return KMCF_OK;
    g->tpb = pick;
    g->nblocks = (m->n_sell_tiles + pick - 1) / pick;
    // reduction: flat (g1 = 0: every block reads every block's sums itself, one hop) up to 256 blocks, else by groups of
    // g1 blocks (two hops); KMCF_CGR_G1 = n forces groups of n
    int g1 = getenv("KMCF_CGR_G1") ? atoi(getenv("KMCF_CGR_G1")) : (g->nblocks <= 256 ? 0 : 16);
    if (g1 != 0 || g->nblocks > 256) {
        g1 = std::max(2, std::min(64, g1));
        while ((g->nblocks + g1 - 1) / g1 > 64) g1 *= 2;            // one lane per group in the second stage
        if (g1 > 64) { g->tpb = 0; return KMCF_OK; }
    }
    g->g1 = g1;
    g->ngroups = g1 ? (g->nblocks + g1 - 1) / g1 : 1;
    g->zwords = 2 * ((size_t)m->n_loc + m->n_halo + 2);
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_zll), 2 * g->zwords * sizeof(u64)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_slot), 2 * (size_t)g->nblocks * CGR_LINE * sizeof(u64)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_gslot), 2 * (size_t)g->ngroups * CGR_LINE * sizeof(u64)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_seq), sizeof(unsigned int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_err), sizeof(int)));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->h_err), sizeof(int), hipHostMallocDefault));
    *g->h_err = 0;
    // granules never carry sequence number 0: everything starts cleared (the zones in the peer-to-peer window were
    // cleared when the matrix was built, kmcf_p2p_matrix_alloc)
    KMCF_HIP(hipMemset(g->d_zll, 0, 2 * g->zwords * sizeof(u64)));
    KMCF_HIP(hipMemset(g->d_slot, 0, 2 * (size_t)g->nblocks * CGR_LINE * sizeof(u64)));
    KMCF_HIP(hipMemset(g->d_gslot, 0, 2 * (size_t)g->ngroups * CGR_LINE * sizeof(u64)));
    KMCF_HIP(hipMemset(g->d_seq, 0, sizeof(unsigned int)));
    KMCF_HIP(hipMemset(g->d_err, 0, sizeof(int)));
    g->seq_bound = 0;
    return KMCF_OK;
}

bool kmcf_cgr_usable(kmcf_matrix *m)
{
    if (cgr_mode() == 0) return false;
    // a group agrees on the launch inside cgr_plan (a collective): every rank must get there or none -- so what decides
    // here is the same on every rank (the transport); what differs from rank to rank is weighed inside
    if (m->comm->nranks > 1 ? !m->comm->p2p_active : !kmcf_sell_coded_active(m)) return false;
    if (cgr_plan(m) != KMCF_OK) return false;
    return m->cgr && m->cgr->tpb > 0;
}

int kmcf_cgr_plan_info(kmcf_matrix *m, int *tpb, int *g1, int *nblocks)
{
    // (a group's plan is agreed among its ranks at the first solve: asking must not start that exchange)
    const bool ok = (m->comm->nranks > 1 && !m->cgr) ? false : kmcf_cgr_usable(m);
    if (tpb) *tpb = ok ? m->cgr->tpb : 0;
    if (g1) *g1 = ok ? m->cgr->g1 : 0;
    if (nblocks) *nblocks = ok ? m->cgr->nblocks : 0;
    return KMCF_OK;
}

// Enqueues one resident solve on the workspace (m->d_r: b in, r out; m->d_x: x0 in, x out; m->d_dinv), scalars into
// m->d_S.  The caller synchronises and then calls kmcf_cgr_check.
int kmcf_cgr_solve(kmcf_matrix *m, bool precond, double tol, int max_it, int fixed_iters, bool classic)
{
    kmcf_cgr *g = m->cgr;
    KMCF_CHECK(g && g->tpb > 0, KMCF_ERR_STATE, "kmcf_cgr_solve: the matrix has no resident plan");
    kmcf_comm *c = m->comm;
    hipStream_t st = c->stream;
    KMCF_TRY(kmcf_sell_ready(m));
    const int limit = fixed_iters > 0 ? fixed_iters : max_it;
    // sequence numbers are 32 bits in the LL words and must never repeat within the life of the buffers: well before
    // the counter could wrap, the buffers are cleared and the counter starts again
    if (g->seq_bound + 3ull * (unsigned long long)limit + 32 > 0xf0000000ull) {
        // (a group: every rank reaches this point in the same solve -- the bound advances identically -- and clears its
        // zones between two collectives: nobody still writes granules of the solve before, nobody publishes early)
        if (c->nranks > 1) {
            KMCF_HIP(hipStreamSynchronize(st));
            KMCF_TRY(kmcf_comm_allreduce_sum(c, c->d_scratch, 1));
            KMCF_HIP(hipStreamSynchronize(st));
        }
        KMCF_HIP(hipMemsetAsync(g->d_zll, 0, 2 * g->zwords * sizeof(u64), st));
        KMCF_HIP(hipMemsetAsync(g->d_slot, 0, 2 * (size_t)g->nblocks * CGR_LINE * sizeof(u64), st));
        KMCF_HIP(hipMemsetAsync(g->d_gslot, 0, 2 * (size_t)g->ngroups * CGR_LINE * sizeof(u64), st));
        KMCF_HIP(hipMemsetAsync(g->d_seq, 0, sizeof(unsigned int), st));
        KMCF_HIP(hipMemsetAsync(g->d_err, 0, sizeof(int), st));
        if (c->nranks > 1) {
            char *win = kmcf_p2p_window(c);
            KMCF_HIP(hipMemsetAsync(win + m->p2p->ll_off, 0, 2 * (size_t)std::max(m->n_halo, 1) * 2 * sizeof(u64), st));
            KMCF_HIP(hipMemsetAsync(win + m->p2p->red_off, 0, 2 * (size_t)P2P_MAXR * P2P_FS * sizeof(u64), st));
            KMCF_HIP(hipStreamSynchronize(st));
            KMCF_TRY(kmcf_comm_allreduce_sum(c, c->d_scratch, 1));
            KMCF_HIP(hipStreamSynchronize(st));
        }
        g->seq_bound = 0;
    }
    g->seq_bound += 3ull * (unsigned long long)limit + 16;          // (publications + reductions: at most three numbers per iteration)
    static const long long timeout_ms = getenv("KMCF_CGR_TIMEOUT_MS") ? atoll(getenv("KMCF_CGR_TIMEOUT_MS")) : 4000;
    int rate_khz = 0;
    KMCF_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c->device));
    cgr_args A{};
    A.n_tiles = m->n_sell_tiles; A.nblocks = g->nblocks; A.g1 = g->g1; A.ngroups = g->ngroups;
    A.tile4 = m->d_sell_tile; A.swave = m->d_sell_wave; A.wcol = m->d_sell_wcol;
    A.stream = reinterpret_cast<const sell_pair *>(m->d_sell);
    A.dict = m->d_dict; A.diagv = m->d_diagv;
    A.r = m->d_r; A.x = m->d_x; A.dinv = precond ? m->d_dinv : nullptr;
    A.b_src = m->solve_b_src ? m->solve_b_src : m->d_r;
    A.x_user = m->solve_x_user;
    A.perm = m->d_perm;
    A.S = m->d_S;
    A.zll = g->d_zll; A.zwords = (long long)g->zwords;
    A.slot = g->d_slot; A.gslot = g->d_gslot; A.seq = g->d_seq;
    A.d_err = g->d_err; A.h_err = g->h_err;
    A.timeout = (long long)rate_khz * timeout_ms;
    if (c->nranks > 1 && c->p2p && !getenv("KMCF_CGR_TIMEOUT_MS")) A.timeout = c->p2p->timeout_ticks;      // (a group: the transport's bound, KMCF_P2P_TIMEOUT_MS)
    A.limit = limit; A.check_tol = fixed_iters > 0 ? 0 : 1; A.tol2 = tol * tol;
    {
        static const int gd = getenv("KMCF_CGR_DELAY") ? atoi(getenv("KMCF_CGR_DELAY")) : 6;
        static const int rd = getenv("KMCF_CGR_RDELAY") ? atoi(getenv("KMCF_CGR_RDELAY")) : 6;
        static const int ad = getenv("KMCF_CGR_ADAPT") ? atoi(getenv("KMCF_CGR_ADAPT")) : 16;
        A.gather_delay = gd; A.reduce_delay = rd; A.adapt_delay = ad;
    }
    A.classic = classic ? 1 : 0;
    A.sibling_lds = !(getenv("KMCF_CGR_SIB") && atoi(getenv("KMCF_CGR_SIB")) == 0);
    A.nranks = c->nranks; A.rank = c->rank; A.n_loc = m->n_loc;
    if (c->nranks > 1) {
        const kmcf_p2p_halo *h = m->p2p;
        char *win = kmcf_p2p_window(c);
        A.put_row = h->d_put_row; A.putr_ptr = h->d_putr_ptr; A.putr_ll = h->d_putr_ll; A.putr_ll_stride = h->d_putr_ll_stride;
        A.halo_ll = reinterpret_cast<const u64 *>(win + h->ll_off);
        A.halo_stride = 2 * (long long)std::max(m->n_halo, 1);
        A.red_peer = h->d_red_peer;
        A.red_mine = reinterpret_cast<const u64 *>(win + h->red_off);
    }
    return cgr_run_any(m, g->tpb, A, true, st, nullptr);
}

int kmcf_cgr_check(kmcf_matrix *m)
{
    kmcf_cgr *g = m->cgr;
    if (!g || !g->h_err || *g->h_err == 0) return KMCF_OK;
    const int code = *g->h_err;
    *g->h_err = 0;
    hipMemsetAsync(g->d_err, 0, sizeof(int), m->comm->stream);
    g->seq_bound = 0xffffffffull;            // whatever the buffers hold now: cleared before the next solve
    kmcf_set_error("resident CG: a bounded wait expired (%s; KMCF_CGR_TIMEOUT_MS) -- not all blocks of the launch were resident, "
                   "or the device is shared with a kernel that never ends", code == 11 ? "window gather" : code == 12 ? "group sums" : "final sums");
    return KMCF_ERR_STATE;
}
