// KMC event step: residence-time selection over the (site, neighbour) event list.
// Replaces execute_kmc_step_mpi (src/kmc_events.cu:333-563) and its kernels build_event_list_split
// (:128-207), zero_out_events_split (:237-256), read_out_event (:257-269), execute_event (:284-331).
//
// Reference per EVENT: thrust::inclusive_scan over all count*nn slots (16 B/slot read + 8 B/slot written),
// a device upper_bound, a full zero-out pass, three host syncs.  Here the list keeps a two-level sum tree
// (one sum per 2048-slot tile, one per group of 256 tiles), no scan array at all:
//   per step : build kernel (type u8 + probability f64 per slot), tile sums, symmetry check of the lists
//   per event: select = three block-wide searches (group sums, the <= 256 tile sums of one group, the <= 2048
//              slots of one tile) -> execute -> zero the events touching the pair and refresh the affected
//              tile and group sums.  One rank with symmetric neighbour lists: only the rows of i, j and of
//              their listed neighbours are visited (<= 2 + 2 nn rows, <= 4 nn + 4 tiles), select + execute +
//              zero-out are one launch, and the whole event needs ONE host sync; otherwise a full pass over
//              this rank's slots (the reference's way).
// Same selection rule (first slot whose inclusive cumulative sum exceeds u * total), same event rules,
// same loop (events are drawn until the LAST drawn residence time reaches 1/freq; that last draw is the
// returned time).  Ranks own contiguous site ranges; the partial totals are all-gathered and the rank
// whose range holds the drawn number selects (MPI_Allgather + MPI_Bcast in the reference, :423-470).
#include <climits>
#include <cstring>
#include <chrono>
#include <cmath>
#include <random>
#include <vector>

#include "kmcf_internal.hpp"

struct kmcf_rng {
    std::mt19937 rng{0};                                            // src/random_num.h
    std::uniform_real_distribution<double> distribution{0.0, 1.0};
};

// Workspace of kmcf_execute_kmc_step, owned by the communicator (kmcf_comm::ev_cache).
struct kmcf_event_cache {
    size_t M = 0;                       // event slots of this rank (count * nn)
    int P = 0, nn = 0;
    unsigned char *d_type = nullptr;
    double *d_prob = nullptr, *d_tsum = nullptr, *d_gsum = nullptr, *d_tot = nullptr;
    int *d_ij = nullptr, *d_aff = nullptr, *d_asym = nullptr;
    double *d_u = nullptr, *d_totlog = nullptr;     // batch: uniforms in, totals out
    int *d_evlog = nullptr;
    void *d_state = nullptr;
    char *h_pin = nullptr;                          // persistent batches: state | event log | totals | uniforms, pinned, written / read by the kernel itself
    int batch_number = 0;
    // row-aligned sum tree of the persistent batch kernel: one sum per row (nn slots), per tile of EV_RT rows, per
    // group of EV_GROUP tiles
    double *d_rsum = nullptr, *d_tsum2 = nullptr, *d_gsum2 = nullptr;
    const int *sym_key = nullptr;       // neighbour list the symmetry verdict belongs to
    int sym_N = 0;
    bool symmetric = false;
    // replicated step of a multi-rank group: the neighbour lists of ALL sites, gathered once
    int *d_neigh_full = nullptr;
    const int *full_key = nullptr;      // caller's list the gathered copy belongs to
    int full_N = 0;
};

void kmcf_event_cache_free(kmcf_comm *c)
{
    if (!c || !c->ev_cache) return;
    kmcf_event_cache *w = c->ev_cache;
    void *ptrs[] = {w->d_type, w->d_prob, w->d_tsum, w->d_gsum, w->d_tot, w->d_ij, w->d_aff, w->d_asym,
                    w->d_u, w->d_totlog, w->d_evlog, w->d_state, w->d_neigh_full, w->d_rsum, w->d_tsum2, w->d_gsum2};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (w->h_pin) hipHostFree(w->h_pin);
    delete w;
    c->ev_cache = nullptr;
}

extern "C" int kmcf_rng_create(unsigned int seed, kmcf_rng **out)
{
    KMCF_CHECK(out, KMCF_ERR_ARG, "kmcf_rng_create: null argument");
    kmcf_rng *r = new kmcf_rng();
    r->rng.seed(seed);                                              // RandomNumberGenerator::setSeed
    *out = r;
    return KMCF_OK;
}

extern "C" double kmcf_rng_next(void *rng) { kmcf_rng *r = static_cast<kmcf_rng *>(rng); return r->distribution(r->rng); }

extern "C" int kmcf_rng_destroy(kmcf_rng *r) { delete r; return KMCF_OK; }

namespace {

constexpr int EL_DEFECT = 0, EL_OXYGEN_DEFECT = 1, EL_VACANCY = 2, EL_O = 3;       // src/utils.h:37-44
constexpr int EV_GEN = 0, EV_REC = 1, EV_VDIFF = 2, EV_ODIFF = 3, EV_NULL = 4;     // EVENTTYPE, src/utils.h:53-60
constexpr int EV_TILE = 2048;                                                       // slots per tile sum
constexpr int EV_GROUP = 256;                                                       // tiles per group sum
constexpr int MAX_LAYERS = 8;

struct layer_energies { double gen[MAX_LAYERS], rec[MAX_LAYERS], vdiff[MAX_LAYERS], odiff[MAX_LAYERS]; };

__device__ __forceinline__ double v_solve_dev(double r_dist, int charge, double sigma, double k)
{
    const double q = 1.60217663e-19;                                // src/gpu_solvers.h:321-329
    return (double)charge * erfc(r_dist / (sigma * sqrt(2.0))) * k * q / r_dist;
}

__device__ __forceinline__ double block_sum_ev(double v, double *lds4)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

// build_event_list_split, src/kmc_events.cu:128-207
__global__ __launch_bounds__(KMCF_BLOCK) void build_event_list_kernel(
    int N, int size_i, int start_i, int nn, const int *__restrict__ neigh_idx, const int *__restrict__ layer,
    double T_bg, double freq, double sigma, double k, const double *__restrict__ x, const double *__restrict__ y,
    const double *__restrict__ z, const double *__restrict__ pot, const int *__restrict__ element,
    const int *__restrict__ charge, layer_energies E, unsigned char *__restrict__ event_type,
    double *__restrict__ event_prob)
{
    const double kB = 8.617333262e-5, epsilon = 1e-200;
    const size_t M = (size_t)size_i * nn;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < M; id += (size_t)gridDim.x * blockDim.x) {
        int et = EV_NULL;
        double P = 0.0;
        const int i = (int)(id / nn) + start_i;
        const int j = neigh_idx[id];
        if (j >= 0 && j < N) {
            const double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
            const double dist = 1e-10 * sqrt(dx * dx + dy * dy + dz * dz);
            const int ei = element[i], ej = element[j];
            const double dpot = pot[i] - pot[j];
            double EA = 0.0;
            if (ei == EL_DEFECT && ej == EL_O) {
                const double Eg = 2 * dpot;
                EA = E.gen[layer[j]] - Eg - 0;
                et = EV_GEN;
            }
            if (ei == EL_OXYGEN_DEFECT && ej == EL_VACANCY) {
                const double self_int_V = v_solve_dev(dist, 2, sigma, k);
                const int charge_state = charge[i] - charge[j];
                const double Er = charge_state * (dpot + (charge_state / 2) * self_int_V);
                EA = E.rec[layer[j]] - Er - 0;
                et = EV_REC;
            }
            if (ei == EL_VACANCY && ej == EL_O) {
                double self_int_V = 0.0;
                if (charge[i] != 0) self_int_V = v_solve_dev(dist, charge[i], sigma, k);
                const double Ev = (charge[i] - charge[j]) * (dpot + self_int_V);
                EA = E.vdiff[layer[j]] - Ev - 0;
                et = EV_VDIFF;
            }
            if (ei == EL_OXYGEN_DEFECT && ej == EL_DEFECT) {
                double self_int_V = 0.0;
                if (charge[i] != 0) self_int_V = v_solve_dev(dist, 2, sigma, k);
                const double Eo = (charge[i] - charge[j]) * (dpot - self_int_V);
                EA = E.odiff[layer[j]] - Eo - 0;
                et = EV_ODIFF;
            }
            if (et != EV_NULL) P = freq * (1 / (exp(EA / (kB * T_bg)) + epsilon));
        }
        event_type[id] = (unsigned char)et;
        event_prob[id] = P;
    }
}

// Full pass: zero the events touching the executed pair (zero_out_events_split, :237-256) and refresh every
// tile sum.  i_del < 0: plain tile sums (first call of a step).
__global__ __launch_bounds__(KMCF_BLOCK) void zero_and_sum_kernel(size_t M, int start_i, int nn,
                                                                  const int *__restrict__ neigh_idx,
                                                                  unsigned char *__restrict__ event_type,
                                                                  double *__restrict__ event_prob, int i_del, int j_del,
                                                                  double *__restrict__ tsum)
{
    __shared__ double lds4[4];
    const size_t base = (size_t)blockIdx.x * EV_TILE;
    double s = 0.0;
    for (int t = threadIdx.x; t < EV_TILE; t += KMCF_BLOCK) {
        const size_t id = base + t;
        if (id >= M) break;
        double p = event_prob[id];
        if (i_del >= 0) {
            const int i = (int)(id / nn) + start_i, j = neigh_idx[id];
            if (j >= 0 && (i == i_del || j == j_del || i == j_del || j == i_del)) {
                event_type[id] = (unsigned char)EV_NULL;
                event_prob[id] = 0.0;
                p = 0.0;
            }
        }
        s += p;
    }
    const double tot = block_sum_ev(s, lds4);
    if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}

// refresh the tile sums listed in aff (one block per entry)
__global__ __launch_bounds__(KMCF_BLOCK) void tile_sum_kernel(size_t M, const double *__restrict__ event_prob,
                                                              const int *__restrict__ aff, double *__restrict__ tsum)
{
    __shared__ double lds4[4];
    const int tile = aff[blockIdx.x];
    if (tile < 0) return;
    const size_t base = (size_t)tile * EV_TILE;
    double s = 0.0;
    for (int t = threadIdx.x; t < EV_TILE; t += KMCF_BLOCK) {
        const size_t id = base + t;
        if (id >= M) break;
        s += event_prob[id];
    }
    const double tot = block_sum_ev(s, lds4);
    if (threadIdx.x == 0) tsum[tile] = tot;
}

// group sums (sequential over the <= 256 tiles of a group, one thread per group) and, by thread 0 of the
// last block to finish ... kept simple: a second 1-thread kernel adds the groups up.
__global__ __launch_bounds__(KMCF_BLOCK) void group_sum_kernel(int nb, const double *__restrict__ tsum, int ng, double *__restrict__ gsum)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const int b0 = g * EV_GROUP, b1 = min(b0 + EV_GROUP, nb);
    double s = 0.0;
    for (int b = b0; b < b1; ++b) s += tsum[b];
    gsum[g] = s;
}

__global__ void total_kernel(int ng, const double *__restrict__ gsum, double *__restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int g = 0; g < ng; ++g) s += gsum[g];
    *total = s;
}

// refresh the sums of the groups that hold a tile listed in aff: one block per entry, one thread per tile of
// the group (duplicates rewrite the same value; a one-thread walk over the 256 tile sums took 20 us)
__global__ __launch_bounds__(KMCF_BLOCK) void group_sum_aff_kernel(const int *__restrict__ aff, int nb,
                                                                   const double *__restrict__ tsum, double *__restrict__ gsum)
{
    static_assert(EV_GROUP == KMCF_BLOCK, "one thread per tile of a group");
    __shared__ double lds4[4];
    const int tile = aff[blockIdx.x];
    if (tile < 0) return;                                      // block-uniform
    const int g = tile / EV_GROUP;
    const int b = g * EV_GROUP + threadIdx.x;
    const double s = block_sum_ev(b < nb ? tsum[b] : 0.0, lds4);
    if (threadIdx.x == 0) gsum[g] = s;
}

// Wave-wide (64 lanes, no barrier) search for the first index k in [0, L) with  acc + (a[0] + ... + a[k]) >
// number.  A lane owns PER consecutive entries: running sums inside the lane, a shuffle scan of the lane totals
// on top, so every cumulative sum is formed in one fixed order.  (It is not the order of a one-thread walk:
// the two can pick different slots only when `number` lies within rounding of a slot boundary.)  If the sums
// never exceed `number` -- rounding at the very end of the list -- the last entry with a positive value is
// taken.  Returns the index, or -1 if no entry is positive; *acc becomes the cumulative sum before it.
// Every lane of the (fully active) wave gets both.
// Lane exchanges of the search through the VALU (DPP) and scalar reads instead of ds_bpermute (see ev_partner below;
// checked in tools/lab/scan_lab.hip).  All lanes of the wavefront must be active.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double ev_dpp0(double v)        // lanes without a source, or outside ROWMASK, receive 0.0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// inclusive scan over the 64 lanes: inside each row of 16 lanes the steps 1, 2, 4, 8, then the totals of the rows before
__device__ __forceinline__ double ev_scan_incl(double v)
{
    v += ev_dpp0<0x111, 0xf>(v);      // row_shr:1
    v += ev_dpp0<0x112, 0xf>(v);      // row_shr:2
    v += ev_dpp0<0x114, 0xf>(v);      // row_shr:4
    v += ev_dpp0<0x118, 0xf>(v);      // row_shr:8
    v += ev_dpp0<0x142, 0xa>(v);      // row_bcast15: lane 15 of rows 0 / 2 into rows 1 / 3
    v += ev_dpp0<0x143, 0xc>(v);      // row_bcast31: lane 31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double ev_readlane(double v, int src)   // src: the same in every lane
{
    const int sl = __builtin_amdgcn_readfirstlane(src);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), sl), __builtin_amdgcn_readlane(__double2loint(v), sl));
}

template <int PER, class F>
__device__ __forceinline__ int wave_search_f(F value_at, int L, double number, double *acc)
{
    const int lane = threadIdx.x & 63;
    double base = *acc;
    int last = -1;
    double last_acc = 0.0;
    for (int s0 = 0; s0 < L; s0 += 64 * PER) {
        double p[PER];
        double run = 0.0;
        int my_last = -1;
        double my_last_acc = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = s0 + lane * PER + k;
            const double v = idx < L ? value_at(idx) : 0.0;
            if (v > 0.0) { my_last = idx; my_last_acc = run; }
            run += v;
            p[k] = run;
        }
        const double incl = ev_scan_incl(run);
        const double excl = ev_dpp0<0x138, 0xf>(incl);        // wave_shr:1 (lane 0: 0.0)
        const double seg_total = ev_readlane(incl, 63);
        const double mine = base + excl;                      // cumulative sum before this lane's entries
        int cand = INT_MAX;
        double cand_acc = 0.0;
#pragma unroll
        for (int k = PER - 1; k >= 0; --k) {
            const int idx = s0 + lane * PER + k;
            if (idx < L && mine + p[k] > number) { cand = idx; cand_acc = mine + (k ? p[k - 1] : 0.0); }
        }
        const unsigned long long hit = __ballot(cand != INT_MAX);
        if (hit) {                                            // lanes own ascending ranges: the lowest lane wins
            const int src = __ffsll((long long)hit) - 1;
            *acc = ev_readlane(cand_acc, src);
            return __builtin_amdgcn_readlane(cand, __builtin_amdgcn_readfirstlane(src));
        }
        const unsigned long long pos = __ballot(my_last >= 0);
        if (pos) {
            const int src = 63 - __clzll((long long)pos);
            last = __builtin_amdgcn_readlane(my_last, __builtin_amdgcn_readfirstlane(src));
            last_acc = ev_readlane(mine + my_last_acc, src);
        }
        base += seg_total;
    }
    if (last >= 0) *acc = last_acc;
    return last;
}
template <int PER>
__device__ int wave_search(const double *__restrict__ a, int L, double number, double *acc)
{
    return wave_search_f<PER>([a](int idx) { return a[idx]; }, L, number, acc);
}

// First slot whose inclusive cumulative sum exceeds the drawn number (thrust::upper_bound on the scan,
// :444): search the group sums, then the tile sums of that group, then the slots of that tile -- three
// searches by the first wavefront of the block, no barrier in between.  number < 0: the number is u * total
// with u given and the total is formed here (single-rank path).  FUSED: lane 0 then executes the event
// (execute_event, :284-331) and the whole block zeroes the events of the pair through the neighbour lists --
// one launch instead of three.
//
// FUSED launches come in batches without a host round trip in between (the host pre-draws the random numbers):
// event `ev` of the batch takes its two uniforms from batch_u[2 ev], [2 ev + 1], logs (i, j, type) and the
// total, and decides itself whether the step goes on (residence time -log(u2)/total < 1/freq, :418, :479);
// once the step has ended the remaining launches of the batch return at once.
struct event_batch_state {
    int done;      // 0 running, 1 the last executed event ended the step, 2 no event could be selected
    int n_exec;    // events executed in this batch
    int seq;       // batch kernel: the number of the batch, written last (the host polls it: pinned host memory)
    int pad_;
};

template <bool FUSED>
__global__ __launch_bounds__(KMCF_BLOCK) void select_event_kernel(
    size_t M, int nb, int ng, int start_i, int nn, double number, double u, const double *__restrict__ gsum,
    const double *__restrict__ tsum, double *__restrict__ event_prob, unsigned char *__restrict__ event_type,
    const int *__restrict__ neigh_idx, int *__restrict__ ijevent, double *__restrict__ total_out,
    int *__restrict__ site_element, int *__restrict__ site_charge, int *__restrict__ aff, int ev,
    const double *__restrict__ batch_u, event_batch_state *__restrict__ state, double inv_freq)
{
    __shared__ int s_ij[3];
    __shared__ int s_done;
    double total = 0.0;
    if (FUSED) {
        // one read of the flag for the whole block: lane 0 writes it further down in this same launch, and a
        // wave scheduled late must not see that write and skip its share of the zero-out
        if (threadIdx.x == 0) s_done = state->done;
        __syncthreads();
        if (s_done) return;
        u = batch_u[2 * ev];
        ijevent += 3 * ev;
        total_out += 2 * ev;                                   // (total rate, residence time) per event
    }
    if (threadIdx.x < 64) {                                    // the first wavefront selects
        const int lane = threadIdx.x;
        if (number < 0) {
            double s = 0.0;
            for (int g = lane; g < ng; g += 64) s += gsum[g];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            total = s;
            if (lane == 0) *total_out = total;
            number = u * total;
        }
        double acc = 0.0;
        int g = wave_search<4>(gsum, ng, number, &acc);
        if (g < 0) g = 0;
        const int b0 = g * EV_GROUP;
        int b = wave_search<4>(tsum + b0, min(EV_GROUP, nb - b0), number, &acc);
        b = b0 + (b < 0 ? 0 : b);
        const size_t id0 = (size_t)b * EV_TILE;
        const int L = (int)(id0 + EV_TILE < M ? (size_t)EV_TILE : M - id0);
        int k = wave_search<EV_TILE / 64>(event_prob + id0, L, number, &acc);
        if (k < 0) k = 0;
        const size_t id = id0 + k;
        if (lane == 0) {
            const int i = (int)(id / nn) + start_i, j = neigh_idx[id], et = (int)event_type[id];
            ijevent[0] = s_ij[0] = i;
            ijevent[1] = s_ij[1] = j;
            ijevent[2] = s_ij[2] = et;
            if (FUSED && j >= 0) {
                if (et == EV_GEN) { site_element[i] = EL_OXYGEN_DEFECT; site_element[j] = EL_VACANCY; site_charge[i] = -2; site_charge[j] = 2; }
                else if (et == EV_REC) { site_element[i] = EL_DEFECT; site_element[j] = EL_O; site_charge[i] = 0; site_charge[j] = 0; }
                else if (et == EV_VDIFF || et == EV_ODIFF) {
                    const int te = site_element[i]; site_element[i] = site_element[j]; site_element[j] = te;
                    const int tc = site_charge[i]; site_charge[i] = site_charge[j]; site_charge[j] = tc;
                }
            }
            if (FUSED) {
                if (j < 0) {
                    state->done = 2;                           // nothing selectable: the host reports it
                } else {
                    state->n_exec = ev + 1;
                    // the device alone decides whether the step goes on; the host takes this t_res as the event
                    // time it returns (no second evaluation with std::log that could disagree by an ulp)
                    const double t_res = -log(batch_u[2 * ev + 1]) / total;
                    total_out[1] = t_res;
                    if (!(t_res < inv_freq)) state->done = 1;  // this event was the step's last
                }
            }
        }
    }
    if (!FUSED) return;
    __syncthreads();
    const int i_del = s_ij[0], j_del = s_ij[1];
    for (int t = threadIdx.x; t < 4 * nn + 4; t += KMCF_BLOCK) aff[t] = -1;
    if (j_del < 0) return;
    __syncthreads();
    // rows i_del and j_del lose all their events; a row n that lists i_del or j_del is, by symmetry, listed
    // by them (one rank, symmetric lists).  Affected tile ids -> aff (duplicates are harmless).
    for (int t = threadIdx.x; t < 2 * nn; t += KMCF_BLOCK) {
        const int s = t < nn ? i_del : j_del;
        const size_t own = (size_t)s * nn + (t % nn);
        const int n = neigh_idx[own];
        if (n >= 0) {
            event_type[own] = (unsigned char)EV_NULL;
            event_prob[own] = 0.0;
            const size_t rb = (size_t)n * nn;
            aff[2 * t] = (int)(rb / EV_TILE);
            aff[2 * t + 1] = (int)((rb + nn - 1) / EV_TILE);
        }
    }
    // the 2 nn x nn slots of the listed neighbours' rows, spread over the whole block
    for (int idx = threadIdx.x; idx < 2 * nn * nn; idx += KMCF_BLOCK) {
        const int t = idx / nn, q = idx - t * nn;
        const int n = neigh_idx[(size_t)(t < nn ? i_del : j_del) * nn + (t % nn)];
        if (n < 0) continue;
        const size_t sl = (size_t)n * nn + q;
        const int jj = neigh_idx[sl];
        if (jj == i_del || jj == j_del) { event_type[sl] = (unsigned char)EV_NULL; event_prob[sl] = 0.0; }
    }
    if (threadIdx.x < 2) {
        const size_t rb = (size_t)(threadIdx.x == 0 ? i_del : j_del) * nn;
        aff[4 * nn + 2 * threadIdx.x] = (int)(rb / EV_TILE);
        aff[4 * nn + 2 * threadIdx.x + 1] = (int)((rb + nn - 1) / EV_TILE);
    }
}


// ------------------------------------------------------------------------------------------------------------
// Persistent batch kernel (one rank holding all rows, symmetric neighbour lists): ONE block of 16 wavefronts runs a
// whole batch of events -- select, execute, zero what the event invalidates, refresh the sums -- with no launch and
// no host round trip in between.  What makes one block enough is a sum tree aligned to ROWS: one sum per row of nn
// slots, per tile of EV_RT rows, per group of EV_GROUP tiles.  An event empties the rows of i and j (their sums
// become 0) and removes the slots that point to them from the rows of their <= 2 nn neighbours: a wavefront per such
// row reads its nn neighbour ids and probabilities (two coalesced loads), zeroes, and adds the row up again on the
// spot (44 KB per event at nn = 52; the slot-aligned tree of the three-launch path re-adds <= 4 nn + 4 tiles of 2048
// slots, 1.6 MB, which took 212 blocks); then half a wavefront per touched tile adds its EV_RT row sums and a
// wavefront per touched group its EV_GROUP tile sums.  Fixed orders (the step's build kernel uses the same
// functions): row = butterfly over the nn lanes; tile = butterfly over 32 lanes; group = four consecutive tile sums
// per lane, then the butterfly.
constexpr int EV_RT = 32;
constexpr int EV_PB = 1024;                                 // threads of the persistent block
constexpr int EV_GLDS = 512;                                // group sums the block keeps in LDS (4.2 M rows; beyond: read from gsum)
constexpr int EV_ST = 4;                                    // tiles per "supertile": one lane's share of a group sum (ev_group_sum)
constexpr int EV_STMAX = 12800;                             // supertile sums the block keeps in (dynamic) LDS: 100 KB = 1.64 M rows
constexpr int EV_BMAX = 512;                                // events per batch
constexpr int EV_AFF = 2 * 63 + 2;                          // rows an event touches at most (nn <= 63: the entries of two wavefronts)
constexpr int EV_TREL = 2048;                               // range of the tile claim table (tiles above the smallest touched one)
constexpr int EV_OWN = 128;                                 // owner lists: two wavefronts of entries
constexpr int EV_TP = EV_OWN / (2 * (EV_PB / 64));          // tile passes at most: half a wavefront per claimed tile
static_assert(EV_AFF <= 255 && EV_RT == 32, "slab ids are bytes, a slab's row mask is one word");

// Cross-lane partners through the VALU (DPP, and gfx950's v_permlane16/32_swap) instead of ds_bpermute: a double from
// another lane in ~10 cycles instead of the ~130 of a trip through the LDS crossbar -- the sums below are chains of six
// of them.  Checked against __shfl_xor in tools/lab/xlane_lab.hip.  All lanes of the wavefront must be active.
typedef unsigned int ev_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ int ev_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int ev_x16_i(int v)             // lane ^ 16
{
    const ev_u2 a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 16) ? a.x : a.y);
}
__device__ __forceinline__ int ev_x32_i(int v)             // lane ^ 32
{
    const ev_u2 a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32) ? a.x : a.y);
}
// STEP 32, 16, 8, 2, 1: the value of lane ^ STEP; STEP 4: of lane ^ 7 (row_half_mirror -- DPP has no lane ^ 4)
template <int STEP>
__device__ __forceinline__ int ev_partner_i(int v)
{
    if constexpr (STEP == 32) return ev_x32_i(v);
    else if constexpr (STEP == 16) return ev_x16_i(v);
    else if constexpr (STEP == 8) return ev_dpp_i<0x128>(v);       // row_ror:8
    else if constexpr (STEP == 4) return ev_dpp_i<0x141>(v);       // row_half_mirror
    else if constexpr (STEP == 2) return ev_dpp_i<0x4E>(v);        // quad_perm [2,3,0,1]
    else return ev_dpp_i<0xB1>(v);                                 // quad_perm [1,0,3,2]
}
template <int STEP>
__device__ __forceinline__ double ev_partner(double v)
{
    return __hiloint2double(ev_partner_i<STEP>(__double2hiint(v)), ev_partner_i<STEP>(__double2loint(v)));
}
// The fixed order of every sum of the row-aligned tree: a butterfly whose steps pair lane l with l^32, l^16, l^8, l^7,
// l^2, l^1 (every lane ends with the same bits: the pairs are symmetric and + commutes).
__device__ __forceinline__ double ev_wave_sum(double v)
{
    v += ev_partner<32>(v); v += ev_partner<16>(v); v += ev_partner<8>(v);
    v += ev_partner<4>(v); v += ev_partner<2>(v); v += ev_partner<1>(v);
    return v;
}
__device__ __forceinline__ double ev_half_sum(double v)    // over the 32 lanes of a half wavefront
{
    v += ev_partner<16>(v); v += ev_partner<8>(v);
    v += ev_partner<4>(v); v += ev_partner<2>(v); v += ev_partner<1>(v);
    return v;
}
__device__ __forceinline__ int ev_wave_min(int v)
{
    v = min(v, ev_partner_i<32>(v)); v = min(v, ev_partner_i<16>(v)); v = min(v, ev_partner_i<8>(v));
    v = min(v, ev_partner_i<4>(v)); v = min(v, ev_partner_i<2>(v)); v = min(v, ev_partner_i<1>(v));
    return v;
}
// sum of a group's tile sums (a whole wavefront)
__device__ __forceinline__ double ev_group_sum(const double *__restrict__ tsum, long long g, long long n_tiles)
{
    const int lane = threadIdx.x & 63;
    const long long t0 = g * EV_GROUP + 4 * lane;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += t0 + k < n_tiles ? tsum[t0 + k] : 0.0;
    return ev_wave_sum(s);
}

// ... the same, leaving every lane's share (the sum of its supertile) in st[g * 64 + lane] when st != nullptr
__device__ __forceinline__ double ev_group_sum_st(const double *__restrict__ tsum, long long g, long long n_tiles, double *st)
{
    const int lane = threadIdx.x & 63;
    const long long t0 = g * EV_GROUP + 4 * lane;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += t0 + k < n_tiles ? tsum[t0 + k] : 0.0;
    if (st && t0 < n_tiles) st[g * (EV_GROUP / EV_ST) + lane] = s;
    return ev_wave_sum(s);
}

// level 0: row sums (a wavefront per row), 1: tile sums (half a wavefront per tile), 2: group sums (a wavefront per group)
__global__ __launch_bounds__(KMCF_BLOCK) void ev_tree_level_kernel(int level, long long n_out, long long n_in, int nn,
                                                                   const double *__restrict__ in, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * KMCF_BLOCK + threadIdx.x) >> 6, n_waves = (long long)gridDim.x * (KMCF_BLOCK / 64);
    if (level == 0) {
        for (long long r = wave; r < n_out; r += n_waves) {
            const double v = ev_wave_sum(lane < nn ? in[r * nn + lane] : 0.0);
            if (lane == 0) out[r] = v;
        }
    } else if (level == 1) {
        for (long long tp = wave; 2 * tp < n_out; tp += n_waves) {
            const long long tile = 2 * tp + (lane >> 5), row = tile * EV_RT + (lane & 31);
            const double v = ev_half_sum(tile < n_out && row < n_in ? in[row] : 0.0);
            if ((lane & 31) == 0 && tile < n_out) out[tile] = v;
        }
    } else {
        for (long long g = wave; g < n_out; g += n_waves) {
            const double v = ev_group_sum(in, g, n_in);
            if (lane == 0) out[g] = v;
        }
    }
}

struct event_batch_args {
    int count, nn, nbatch, trel_max, number;
    long long n_tiles, n_groups;
    double inv_freq;
    int n_st;      // > 0: the sums of all n_st supertiles (EV_ST tiles = 128 rows each) are kept in LDS for the batch
};

__global__ __launch_bounds__(EV_PB) void event_batch_kernel(
    event_batch_args A, double *__restrict__ prob, unsigned char *__restrict__ type, const int *__restrict__ neigh,
    double *__restrict__ rsum, double *__restrict__ tsum, double *__restrict__ gsum, int *__restrict__ site_element,
    int *__restrict__ site_charge, int *__restrict__ evlog, double *__restrict__ totlog, const double *__restrict__ batch_u,
    event_batch_state *__restrict__ state)
{
    // Per event, five dependent global round trips (tile sums of the chosen group -> row sums of the chosen tile -> the
    // row's slots with their ids and types -> j's neighbour ids -> the <= 2 nn neighbour rows) and six barriers; the
    // sums an event changes are refreshed WITHOUT a further round trip: the row sums of every touched tile and the tile
    // sums of every touched group are requested together with the neighbour rows, and what the event changes is
    // patched into them through LDS (the changed rows' new sums pushed into their tile's slab, the new tile sums looked
    // up by the group's lanes).  Every sum is formed from the same operands in the same order as a fresh build.
    __shared__ int s_ij[3];
    __shared__ int s_stop;
    __shared__ int s_rows[EV_AFF];                         // rows whose events change: the neighbours of i, of j, then i, j
    __shared__ int s_uniq[EV_AFF];                         // bit 0 / 1: first entry of its tile / group in s_rows
    __shared__ int s_min[2];                               // smallest neighbour of i, of j
    // claims (no two lanes may hammer ONE LDS word with atomics: same-address LDS atomics serialise at ~40 cycles each,
    // a wavefront of them is a microsecond): tiles through a table with a word per tile above the smallest touched
    // one (-1, or the entry that owns the tile: conflicts only among the entries of one tile); groups through a
    // wavefront-local vote, then one atomic per distinct group
    __shared__ int s_tent[EV_TREL];
    __shared__ unsigned long long s_gmask[2];              // (two sets used alternately: set ev & 1 is cleared while event ev + 1 runs)
    __shared__ int s_slow[2];                              // 1: a touched tile or group is out of the claim range
    __shared__ unsigned int s_slabmask[EV_AFF];            // per owning entry: rows of its tile whose sum this event changed
    __shared__ double s_slab[EV_AFF][EV_RT];               // ... and their new sums
    __shared__ double s_tnew[EV_AFF];                      // per owning entry: new sum of its tile
    __shared__ int s_own[EV_OWN], s_ocnt[2];               // the owning entries, listed (two lists: see the claims), and how many
    __shared__ double s_g[EV_GLDS], s_u[EV_BMAX], s_nlog[EV_BMAX];
    // Round 4: one level of the selection walk out of LDS instead of memory.  A group sum is formed as four consecutive
    // tile sums per lane, then the butterfly (ev_group_sum): a lane's share -- the sum of a "supertile" of 4 tiles = 128
    // rows -- is kept here for ALL supertiles (100 KB at 40 nm), refreshed for free wherever a group sum is re-formed.
    // The walk is then groups (LDS) -> supertiles of the group (LDS) -> the 128 row sums of the supertile (memory) -> the
    // row's slots (memory): two dependent round trips instead of three (the group's 256 tile sums came from memory).
    extern __shared__ double s_st[];
    const bool st_lds = A.n_st > 0;
    const int nn = A.nn, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int NW = EV_PB / 64;
    const int n_aff = 2 * nn + 2;
#ifdef KMCF_EV_PROFILE
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = wall_clock64();
#define EV_TICK(k) { const long long now_ = wall_clock64(); tk[k] += now_ - t_prev; t_prev = now_; }
#else
#define EV_TICK(k)
#endif
    if (t < 2) { s_gmask[t] = 0ull; s_slow[t] = 0; s_ocnt[t] = 0; }
    for (int q = t; q < EV_TREL; q += EV_PB) s_tent[q] = -1;
    int my_trel = -1;                                      // the table word this thread's entry claimed in the previous event
    // Held in LDS for the whole batch: the group sums (the top of every selection walk; written through to gsum), the
    // batch's uniforms and -log(u) of the residence times.  Each is one global round trip less in front of an event.
    const bool g_lds = A.n_groups <= EV_GLDS;
    if (g_lds) for (int g = t; g < (int)A.n_groups; g += EV_PB) s_g[g] = gsum[g];
    if (st_lds)
        for (int q = t; q < A.n_st; q += EV_PB) {              // (the order of ev_group_sum's per-lane part)
            const long long t0 = (long long)EV_ST * q;
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < EV_ST; ++k) sm += t0 + k < A.n_tiles ? tsum[t0 + k] : 0.0;
            s_st[q] = sm;
        }
    const double *gs = g_lds ? s_g : gsum;
    if (t < A.nbatch) { s_u[t] = batch_u[2 * t]; s_nlog[t] = -log(batch_u[2 * t + 1]); }    // (:479; A.nbatch <= EV_BMAX)
    const int trel_max = A.trel_max;                       // EV_TREL (tests: smaller, to reach the out-of-range path)
    __syncthreads();
    for (int ev = 0; ev < A.nbatch; ++ev) {
        const int par = ev & 1;
        // ---- select (first wavefront): groups -> tiles of the group -> rows of the tile -> slots of the row
        if (t < 64) {
            double sum = 0.0;
            for (long long g = t; g < A.n_groups; g += 64) sum += gs[g];
            const double total = ev_wave_sum(sum);
            const double number = s_u[ev] * total;
            double acc = 0.0;
            int g = wave_search<4>(gs, (int)A.n_groups, number, &acc);
            if (g < 0) g = 0;
            long long row;
            if (st_lds) {
                const int q0 = g * (EV_GROUP / EV_ST);
                int q = wave_search<1>(s_st + q0, min(EV_GROUP / EV_ST, A.n_st - q0), number, &acc);
                const long long r0 = (long long)(q0 + (q < 0 ? 0 : q)) * (EV_ST * EV_RT);
                int r = wave_search<2>(rsum + r0, (int)(r0 + EV_ST * EV_RT < A.count ? EV_ST * EV_RT : A.count - r0), number, &acc);
                row = r0 + (r < 0 ? 0 : r);
            } else {
                const long long b0 = (long long)g * EV_GROUP;
                int b = wave_search<4>(tsum + b0, (int)(b0 + EV_GROUP < A.n_tiles ? EV_GROUP : A.n_tiles - b0), number, &acc);
                const long long tile = b0 + (b < 0 ? 0 : b);
                const long long r0 = tile * EV_RT;
                int r = wave_search<1>(rsum + r0, (int)(r0 + EV_RT < A.count ? EV_RT : A.count - r0), number, &acc);
                row = r0 + (r < 0 ? 0 : r);
            }
            // the row's slots together with their neighbour ids and event types: the chosen slot's j and type arrive with
            // the probabilities instead of one round trip after them -- and the ids ARE row i's share of s_rows
            const long long sl = row * nn + (lane < nn ? lane : 0);
            const double pv_l = prob[sl];
            const int nj_l = neigh[sl], ty_l = (int)type[sl];
            int k = wave_search_f<1>([pv_l](int) { return pv_l; }, nn, number, &acc);
            if (k < 0) k = 0;
            const int ks = __builtin_amdgcn_readfirstlane(k);
            const int j = __builtin_amdgcn_readlane(nj_l, ks), et = __builtin_amdgcn_readlane(ty_l, ks);
            const double t_res = s_nlog[ev] / total;                                  // :479; the device decides whether the step goes on
            const bool last = !(t_res < A.inv_freq);                                  // this event is the step's last
            if (j >= 0 && !last) {
                // row i loses all its events (zero_out_events_split, :237-256, through the symmetric lists)
                const bool mine = lane < nn && nj_l >= 0;
                if (lane < nn) s_rows[lane] = nj_l;
                if (mine) { type[sl] = (unsigned char)EV_NULL; prob[sl] = 0.0; }
                const int mi = ev_wave_min(mine ? nj_l : INT_MAX);
                if (lane == 0) s_min[0] = mi;
            }
            if (t == 0) {
                const int i = (int)row;
                evlog[3 * ev] = s_ij[0] = i;
                evlog[3 * ev + 1] = s_ij[1] = j;
                evlog[3 * ev + 2] = s_ij[2] = et;
                totlog[2 * ev] = total;
                int stop = 0;
                if (j < 0) {
                    state->done = 2;                           // nothing selectable: the host reports it
                    stop = 1;
                } else {
                    state->n_exec = ev + 1;
                    totlog[2 * ev + 1] = t_res;
                    if (last) { state->done = 1; stop = 1; }
                }
                s_stop = stop;
            }
        }
        __syncthreads();
        EV_TICK(0)
        const int i_del = s_ij[0], j_del = s_ij[1];
        // execute_event (:284-331) by the last thread, beside the others' work: nothing else in this kernel reads the site
        // arrays, and the same thread executes every event of the batch (program order between two events' swaps)
        if (t == EV_PB - 1 && j_del >= 0) {
            const int i = i_del, j = j_del, et = s_ij[2];
            if (et == EV_GEN) { site_element[i] = EL_OXYGEN_DEFECT; site_element[j] = EL_VACANCY; site_charge[i] = -2; site_charge[j] = 2; }
            else if (et == EV_REC) { site_element[i] = EL_DEFECT; site_element[j] = EL_O; site_charge[i] = 0; site_charge[j] = 0; }
            else if (et == EV_VDIFF || et == EV_ODIFF) {
                const int te = site_element[i]; site_element[i] = site_element[j]; site_element[j] = te;
                const int tc = site_charge[i]; site_charge[i] = site_charge[j]; site_charge[j] = tc;
            }
        }
        if (s_stop) break;                                     // (the sums are rebuilt by the next step's build)
        // ---- row j loses all its events too; the other set of bookkeeping words is cleared for the next event
        if (t < 64) {
            const bool in = t < nn;
            const long long own = (long long)j_del * nn + (in ? t : 0);
            const int n = in ? neigh[own] : -1;
            if (in) s_rows[nn + t] = n;
            if (n >= 0) { type[own] = (unsigned char)EV_NULL; prob[own] = 0.0; }
            const int mj = ev_wave_min(n >= 0 ? n : INT_MAX);
            if (t == 0) s_min[1] = mj;
        }
        if (t == 64) { s_rows[2 * nn] = i_del; s_rows[2 * nn + 1] = j_del; rsum[i_del] = 0.0; rsum[j_del] = 0.0; }
        if (t == EV_PB - 1) { s_gmask[par ^ 1] = 0ull; s_slow[par ^ 1] = 0; }
        if (my_trel >= 0) { s_tent[my_trel] = -1; my_trel = -1; }   // (last read before the previous event's final barrier)
        __syncthreads();
        EV_TICK(1)
        // ---- requests, first part: eight neighbour rows per wavefront (their ids and probabilities) -- in flight while the
        // claims below are sorted out (requesting the rows of i's neighbours even earlier, beside j's neighbour ids, made
        // an event slower: 11.1 against 9.5 us)
        int nq[8], jj[8];
        double pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = wv + NW * q;
            nq[q] = e < 2 * nn ? s_rows[e] : -1;
            jj[q] = -1; pv[q] = 0.0;
            if (nq[q] >= 0) {                                   // (wavefront-uniform: no requests for empty entries)
                const long long sl = (long long)nq[q] * nn + (lane < nn ? lane : 0);
                jj[q] = neigh[sl];
                pv[q] = prob[sl];
            }
        }
        // ---- claims: the first entry of a tile / group owns it (the others would only repeat the same sums): a bit per
        // tile / group relative to the smallest touched row's.  An owner of a tile takes a slab, an owner of a group a slot.
        const int tmin = min(min(s_min[0], s_min[1]), min(i_del, j_del)) / EV_RT;
        if (wv <= (n_aff - 1) / 64) {                          // (whole wavefronts: the group vote needs every lane)
            const int row = t < n_aff ? s_rows[t] : -1;
            const bool valid = row >= 0;
            const int tile = valid ? row / EV_RT : 0;
            const int trel = tile - tmin, grel = tile / EV_GROUP - tmin / EV_GROUP;
            int u = valid ? 3 : 0;
            bool owner = false;
            if (valid) {
                if (trel < trel_max) {
                    if (atomicCAS(&s_tent[trel], -1, t) != -1) u &= ~1;
                    else { s_slabmask[t] = 0u; my_trel = trel; owner = true; }
                } else s_slow[par] = 1;
                if (grel >= 64) s_slow[par] = 1;
            }
            // the owners in a list without gaps (the tile passes below walk it instead of all entries): the first
            // wavefront's from the bottom of s_own, the second's from the top -- no count has to cross between the two
            {
                const unsigned long long ob = __ballot(owner);
                if (owner) { const int lp = __popcll(ob & ((1ull << lane) - 1ull)); s_own[wv == 0 ? lp : EV_OWN - 1 - lp] = t; }
                if (lane == 0) s_ocnt[wv] = __popcll(ob);
            }
            // groups: a bit per touched group above the smallest touched row's, OR-ed over the wavefront through the VALU,
            // one atomic per wavefront; the group phase walks the set bits (entries whose group is out of the mask's
            // range keep bit 1 of u and re-add their group themselves on the out-of-range path)
            {
                const bool gv = valid && grel < 64;
                int glo = gv && grel < 32 ? 1 << grel : 0, ghi = gv && grel >= 32 ? 1 << (grel - 32) : 0;
                glo |= ev_partner_i<32>(glo); ghi |= ev_partner_i<32>(ghi);
                glo |= ev_partner_i<16>(glo); ghi |= ev_partner_i<16>(ghi);
                glo |= ev_partner_i<8>(glo); ghi |= ev_partner_i<8>(ghi);
                glo |= ev_partner_i<4>(glo); ghi |= ev_partner_i<4>(ghi);
                glo |= ev_partner_i<2>(glo); ghi |= ev_partner_i<2>(ghi);
                glo |= ev_partner_i<1>(glo); ghi |= ev_partner_i<1>(ghi);
                if (lane == 0 && (glo | ghi)) atomicOr(&s_gmask[par], ((unsigned long long)(unsigned int)ghi << 32) | (unsigned int)glo);
                if (gv) u &= ~2;
            }
            if (t < n_aff) s_uniq[t] = u;
        }
        EV_TICK(7)
        __syncthreads();
        EV_TICK(2)
        const bool fastp = s_slow[par] == 0;
        // the touched groups: the set bits of the mask, the k-th for wavefront k (then k + 16, ...: those are read again)
        const unsigned long long gmask = s_gmask[par];
        const int ngrp = __popcll(gmask), gbase = tmin / EV_GROUP;
        auto group_of = [&](int k) {                           // k < ngrp, wavefront-uniform
            unsigned long long mm = gmask;
            for (int q = 0; q < k; ++q) mm &= mm - 1;
            return (long long)gbase + (__ffsll((long long)mm) - 1);
        };
        // ---- requests, second part: the row sums of the claimed tiles (half a wavefront per tile) and the tile sums of the
        // claimed groups (a wavefront per group)
        double tp[EV_TP], gp[4] = {0.0, 0.0, 0.0, 0.0};
        const int c0 = s_ocnt[0], c1 = s_ocnt[1], c0r = (c0 + 31) & ~31, cr = c0r + ((c1 + 31) & ~31);
        // owner of half-wavefront h in pass p: index v = 32 p + h into the two owner lists, each padded to whole passes
        auto owner_of = [&](int v) {
            if (v < c0r) return v < c0 ? s_own[v] : -1;
            const int uu = v - c0r;
            return uu < c1 ? s_own[EV_OWN - 1 - uu] : -1;
        };
#pragma unroll
        for (int p = 0; p < EV_TP; ++p) {
            tp[p] = 0.0;
            if (fastp && p * 2 * NW < cr) {                    // (wavefront-uniform: passes past the lists are skipped)
                const int e = owner_of(p * 2 * NW + 2 * wv + (lane >> 5));
                if (e >= 0) {
                    const long long rr = (long long)(s_rows[e] / EV_RT) * EV_RT + (lane & 31);
                    if (rr < A.count) tp[p] = rsum[rr];
                }
            }
        }
        if (fastp && wv < ngrp) {
            const long long t0 = group_of(wv) * EV_GROUP + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) gp[k] = t0 + k < A.n_tiles ? tsum[t0 + k] : 0.0;
        }
#ifdef KMCF_EV_PROFILE
        asm volatile("" ::"v"(jj[0]), "v"(jj[7]), "v"(pv[0]), "v"(pv[7]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EV_TICK(5)
#endif
        // ---- the neighbour rows: drop the slots that point to i or j, add each row up again.  Eight rows in one
        // butterfly: at offset 32 a lane keeps four rows and hands the other four to its partner, at 16 two and two, at 8
        // one and one -- every row is added in exactly the pairs of ev_wave_sum (lane l with l ^ 32, then ^ 16, ...), with
        // 10 exchanges instead of 48; lanes 8 q .. 8 q + 7 end up with row q's sum.
        double v8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool mine = nq[q] >= 0 && lane < nn;
            double v = mine ? pv[q] : 0.0;
            if (mine && (jj[q] == i_del || jj[q] == j_del)) {
                const long long sl = (long long)nq[q] * nn + lane;
                type[sl] = (unsigned char)EV_NULL; prob[sl] = 0.0; v = 0.0;
            }
            if (nq[q] == i_del || nq[q] == j_del) v = 0.0;  // (their own rows were emptied above)
            v8[q] = v;
        }
        double v4[4], v2[2], v1;
        {
            const bool hi = (lane & 32) != 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double keep = hi ? v8[k + 4] : v8[k], send = hi ? v8[k] : v8[k + 4];
                v4[k] = keep + ev_partner<32>(send);
            }
            const bool h16 = (lane & 16) != 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double keep = h16 ? v4[k + 2] : v4[k], send = h16 ? v4[k] : v4[k + 2];
                v2[k] = keep + ev_partner<16>(send);
            }
            const bool h8 = (lane & 8) != 0;
            const double keep = h8 ? v2[1] : v2[0], send = h8 ? v2[0] : v2[1];
            v1 = keep + ev_partner<8>(send);
            v1 += ev_partner<4>(v1);
            v1 += ev_partner<2>(v1);
            v1 += ev_partner<1>(v1);
        }
        if ((lane & 7) == 0) {
            const int e = wv + NW * (lane >> 3);
            const int nrow = e < 2 * nn ? s_rows[e] : -1;
            if (nrow >= 0) {
                rsum[nrow] = v1;
                if (fastp) {
                    const int slot = s_tent[nrow / EV_RT - tmin];
                    s_slab[slot][nrow & (EV_RT - 1)] = v1;
                    atomicOr(&s_slabmask[slot], 1u << (nrow & (EV_RT - 1)));
                }
            }
        }
        if (fastp && t >= 64 && t < 66) {                      // rows i and j themselves: sum 0
            const int r = t == 64 ? i_del : j_del;
            const int slot = s_tent[r / EV_RT - tmin];
            s_slab[slot][r & (EV_RT - 1)] = 0.0;
            atomicOr(&s_slabmask[slot], 1u << (r & (EV_RT - 1)));
        }
        __syncthreads();
        EV_TICK(6)
        // ---- tile sums: half a wavefront per claimed tile, its rows' sums as requested above with this event's changes
        // patched in (out-of-range path: per entry, read again)
        if (fastp) {
#pragma unroll
            for (int p = 0; p < EV_TP; ++p) {
                if (p * 2 * NW >= cr) continue;                // (wavefront-uniform)
                const int e = owner_of(p * 2 * NW + 2 * wv + (lane >> 5)), l5 = lane & 31;
                const bool on = e >= 0;
                double val = 0.0;
                if (on) val = ((s_slabmask[e] >> l5) & 1u) ? s_slab[e][l5] : tp[p];
                const double v = ev_half_sum(val);
                if (on && l5 == 0) { tsum[s_rows[e] / EV_RT] = v; s_tnew[e] = v; }
            }
            EV_TICK(7)
        } else {
            for (int e = 2 * wv + (lane >> 5); e < n_aff; e += 2 * NW) {
                const int row = s_rows[e];
                const bool on = row >= 0 && (s_uniq[e] & 1);
                const long long tile = on ? row / EV_RT : 0, rr = tile * EV_RT + (lane & 31);
                const double v = ev_half_sum(on && rr < A.count ? rsum[rr] : 0.0);
                if (on && (lane & 31) == 0) tsum[tile] = v;
            }
        }
        __syncthreads();
        EV_TICK(3)
        // ---- group sums: a wavefront per claimed group, the new tile sums looked up through the claim table
        if (fastp) {
            for (int k = wv + NW; k < ngrp; k += NW) {         // (more than 16 touched groups: the tile sums are in memory by now)
                const long long g = group_of(k);
                const double v = ev_group_sum_st(tsum, g, A.n_tiles, st_lds ? s_st : nullptr);
                if (lane == 0) { gsum[g] = v; if (g_lds) s_g[g] = v; }
            }
            if (wv < ngrp) {
                const long long g = group_of(wv);
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long long trel = g * EV_GROUP + 4 * lane + k - tmin;
                    if (trel >= 0 && trel < trel_max) { const int e = s_tent[trel]; if (e >= 0) gp[k] = s_tnew[e]; }
                    s += gp[k];
                }
                if (st_lds && g * EV_GROUP + 4 * lane < A.n_tiles) s_st[g * (EV_GROUP / EV_ST) + lane] = s;
                const double v = ev_wave_sum(s);
                if (lane == 0) { gsum[g] = v; if (g_lds) s_g[g] = v; }
            }
        } else {
            for (int k = wv; k < ngrp; k += NW) {              // the groups in the mask's range ...
                const long long g = group_of(k);
                const double v = ev_group_sum_st(tsum, g, A.n_tiles, st_lds ? s_st : nullptr);
                if (lane == 0) { gsum[g] = v; if (g_lds) s_g[g] = v; }
            }
            for (int e = wv; e < n_aff; e += NW) {             // ... and, entry by entry, those beyond it
                const int row = s_rows[e];
                if (row < 0 || !(s_uniq[e] & 2)) continue;        // wavefront-uniform
                const long long g = (row / EV_RT) / EV_GROUP;
                const double v = ev_group_sum_st(tsum, g, A.n_tiles, st_lds ? s_st : nullptr);
                if (lane == 0) { gsum[g] = v; if (g_lds) s_g[g] = v; }
            }
        }
        __syncthreads();
        EV_TICK(4)
    }
    // the batch's results are in the host's memory (evlog, totlog, state: pinned, written by this thread): its number last
    if (t == 0 && A.number) {
        __threadfence_system();
        __hip_atomic_store(&state->seq, A.number, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef KMCF_EV_PROFILE
    if (t == 0) printf("event batch of %d: ticks select %lld row-j %lld claims %lld nbr-loads %lld nbr-rows %lld tiles %lld groups %lld | before barriers (claims + tiles) %lld\n", A.nbatch, tk[0], tk[1], tk[2], tk[5], tk[6], tk[3], tk[4], tk[7]);
#endif
}

// execute_event, :284-331
__global__ void execute_event_kernel(int *__restrict__ site_element, int *__restrict__ site_charge, const int *__restrict__ ijevent)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int i = ijevent[0], j = ijevent[1], et = ijevent[2];
    if (et == EV_GEN) { site_element[i] = EL_OXYGEN_DEFECT; site_element[j] = EL_VACANCY; site_charge[i] = -2; site_charge[j] = 2; }
    else if (et == EV_REC) { site_element[i] = EL_DEFECT; site_element[j] = EL_O; site_charge[i] = 0; site_charge[j] = 0; }
    else if (et == EV_VDIFF || et == EV_ODIFF) {
        const int te = site_element[i]; site_element[i] = site_element[j]; site_element[j] = te;
        const int tc = site_charge[i]; site_charge[i] = site_charge[j]; site_charge[j] = tc;
    }
}

// every listed neighbour j of i lists i back?  (one rank holds all rows)
__global__ __launch_bounds__(KMCF_BLOCK) void check_symmetry_kernel(int N, int nn, const int *__restrict__ neigh_idx, int *__restrict__ asym)
{
    const size_t M = (size_t)N * nn;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < M; id += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(id / nn), j = neigh_idx[id];
        if (j < 0) continue;
        if (j >= N) { *asym = 1; continue; }
        bool found = false;
        for (int q = 0; q < nn; ++q) found |= (neigh_idx[(size_t)j * nn + q] == i);
        if (!found) *asym = 1;
    }
}

}  // namespace

extern "C" int kmcf_execute_kmc_step(kmcf_comm *c, int N, const int *h_count, const int *h_displs, int nn,
                                     const int *d_neigh_idx, const int *d_site_layer, double T_bg, double freq,
                                     double sigma, double k, const double *d_x, const double *d_y, const double *d_z,
                                     const double *d_site_potential_charge, int *d_site_element, int *d_site_charge,
                                     int num_layers, const double *h_E_gen, const double *h_E_rec,
                                     const double *h_E_Vdiff, const double *h_E_Odiff,
                                     double (*next_random)(void *), void *rng_user, int max_events,
                                     double *event_time, int *n_events, int *h_event_log)
{
    KMCF_CHECK(c && h_count && h_displs && d_neigh_idx && d_site_layer && d_x && d_y && d_z && d_site_potential_charge &&
                   d_site_element && d_site_charge && h_E_gen && h_E_rec && h_E_Vdiff && h_E_Odiff && next_random && event_time,
               KMCF_ERR_ARG, "kmcf_execute_kmc_step: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_execute_kmc_step: host-only communicator");
    KMCF_CHECK(num_layers > 0 && num_layers <= MAX_LAYERS && nn > 0 && freq > 0, KMCF_ERR_ARG, "kmcf_execute_kmc_step: bad sizes");
    KMCF_TRY(kmcf_enter(c));
    hipStream_t st = c->stream;
    // Multi-rank groups: the reference partitions the event list and pays a zero-out pass, an MPI_Allgather, an
    // MPI_Bcast and two host synchronisations PER EVENT (:423-459; ~1e4 events per 40 nm step).  The site arrays
    // are replicated on every rank anyway, so here every rank runs the whole step on the whole list: the neighbour
    // lists are gathered once, the generators are in the same state by contract, and the ranks execute the same
    // events with no collective and no extra synchronisation at all.  KMCF_EVENTS_PARTITIONED=1 keeps the
    // reference's scheme.
    const bool replicate = c->nranks > 1 && !getenv("KMCF_EVENTS_PARTITIONED");
    const int P = replicate ? 1 : c->nranks, rank = replicate ? 0 : c->rank;
    const int count = replicate ? N : h_count[rank], start_i = replicate ? 0 : h_displs[rank];
    const size_t M = (size_t)count * nn;
    const int nb = (int)((M + EV_TILE - 1) / EV_TILE);
    const int ng = std::max((nb + EV_GROUP - 1) / EV_GROUP, 1);
    layer_energies E;
    for (int l = 0; l < MAX_LAYERS; ++l) {
        E.gen[l] = l < num_layers ? h_E_gen[l] : 0.0; E.rec[l] = l < num_layers ? h_E_rec[l] : 0.0;
        E.vdiff[l] = l < num_layers ? h_E_Vdiff[l] : 0.0; E.odiff[l] = l < num_layers ? h_E_Odiff[l] : 0.0;
    }
    const int n_aff = 4 * nn + 4;
    // workspace kept on the communicator between steps: the reference allocates and frees its event arrays on
    // every call (:352-361); at 40 nm that is 0.8 GB per step
    if (c->ev_cache && (c->ev_cache->M != M || c->ev_cache->P != P || c->ev_cache->nn != nn)) kmcf_event_cache_free(c);
    if (!c->ev_cache) {
        kmcf_event_cache *w = new kmcf_event_cache();
        w->M = M; w->P = P; w->nn = nn;
        c->ev_cache = w;
        if (hipMalloc(reinterpret_cast<void **>(&w->d_type), std::max<size_t>(M, 1)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_prob), std::max<size_t>(M, 1) * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_tsum), (size_t)std::max(nb, 1) * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_gsum), (size_t)ng * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_tot), (size_t)P * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_ij), (size_t)3 * P * sizeof(int)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_aff), (size_t)n_aff * sizeof(int)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&w->d_asym), sizeof(int)) != hipSuccess) {
            kmcf_event_cache_free(c);
            kmcf_set_error("kmcf_execute_kmc_step: out of device memory for %zu event slots", M);
            return KMCF_ERR_HIP;
        }
    }
    kmcf_event_cache *w = c->ev_cache;
    if (replicate) {
        if (w->full_key != d_neigh_idx || w->full_N != N) {     // the lists are built once per run (kmc_main.cpp:199)
            if (w->d_neigh_full) { hipFree(w->d_neigh_full); w->d_neigh_full = nullptr; }
            KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&w->d_neigh_full), (size_t)N * nn * sizeof(int)));
            const int R = c->nranks;
            std::vector<int> cn(R), dn(R);
            for (int q = 0; q < R; ++q) {
                KMCF_CHECK((int64_t)h_count[q] * nn < INT32_MAX && (int64_t)h_displs[q] * nn < INT32_MAX, KMCF_ERR_ARG,
                           "kmcf_execute_kmc_step: neighbour list too large for the int32 gather");
                cn[q] = h_count[q] * nn; dn[q] = h_displs[q] * nn;
            }
            if (h_count[c->rank] > 0)
                KMCF_HIP(hipMemcpyAsync(w->d_neigh_full + (size_t)dn[c->rank], d_neigh_idx, (size_t)cn[c->rank] * sizeof(int),
                                        hipMemcpyDeviceToDevice, st));
            KMCF_TRY(kmcf_comm_allgatherv_int(c, w->d_neigh_full, cn.data(), dn.data()));
            w->full_key = d_neigh_idx;
            w->full_N = N;
        }
        d_neigh_idx = w->d_neigh_full;
    }
    unsigned char *d_type = w->d_type;
    double *d_prob = w->d_prob, *d_tsum = w->d_tsum, *d_gsum = w->d_gsum, *d_tot = w->d_tot;
    int *d_ij = w->d_ij, *d_aff = w->d_aff, *d_asym = w->d_asym;
    KMCF_HIP(hipMemsetAsync(d_gsum, 0, (size_t)ng * sizeof(double), st));
    KMCF_HIP(hipMemsetAsync(d_asym, 0, sizeof(int), st));
    std::vector<int> ones(P, 1), iota(P), threes(P, 3), iota3(P);
    for (int q = 0; q < P; ++q) { iota[q] = q; iota3[q] = 3 * q; }
    int rc = KMCF_OK;
    auto fail = [&](int code) { rc = code; };
    bool fast = false;   // neighbour-list zero-out + one host sync per event
    if (M > 0) {
        int64_t g = ((int64_t)M + KMCF_BLOCK * 4 - 1) / (KMCF_BLOCK * 4);
        if (g > 16384) g = 16384;
        build_event_list_kernel<<<(int)g, KMCF_BLOCK, 0, st>>>(N, count, start_i, nn, d_neigh_idx, d_site_layer, T_bg, freq, sigma, k,
                                                               d_x, d_y, d_z, d_site_potential_charge, d_site_element,
                                                               d_site_charge, E, d_type, d_prob);
        zero_and_sum_kernel<<<nb, KMCF_BLOCK, 0, st>>>(M, start_i, nn, d_neigh_idx, d_type, d_prob, -1, -1, d_tsum);
        if (P == 1 && count == N && !getenv("KMCF_EVENTS_FULLSCAN")) {
            // the neighbour lists are built once per run (kmc_main.cpp:199): the verdict on their symmetry is
            // kept with the workspace, keyed by the list's address and shape
            if (w->sym_key != d_neigh_idx || w->sym_N != N) {
                check_symmetry_kernel<<<(int)g, KMCF_BLOCK, 0, st>>>(N, nn, d_neigh_idx, d_asym);
                int asym = 1;
                if (hipMemcpyAsync(&asym, d_asym, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) fail(KMCF_ERR_HIP);
                w->sym_key = d_neigh_idx;
                w->sym_N = N;
                w->symmetric = (asym == 0);
            }
            fast = w->symmetric;
        }
    }
    double t = 0.0;
    int nev = 0;
    std::vector<double> totals(P);
    const int ggrid = (ng + KMCF_BLOCK - 1) / KMCF_BLOCK;
    if (fast && M > 0 && rc == KMCF_OK) {
        // Fast path.  All group sums once, afterwards only the groups of the tiles an event touched; three
        // launches per event (select + execute + zero-out, tile sums, group sums); events are enqueued in
        // batches with ONE host round trip per batch: the two uniforms of every event of the batch are drawn
        // ahead, the device decides after each event whether the step goes on, and the launches behind the
        // step's last event return at once.  With the library's own generator the draws are made on a copy of
        // its state and the real generator is advanced by what was consumed, so the caller's stream is used
        // exactly as by the reference (two draws per executed event); a foreign callback cannot be rewound,
        // so it gets batches of one.
        // persistent batch kernel (default; KMCF_EVENTS_PERSISTENT=0: three launches per event) with its row-aligned sums
        const bool persistent = !(getenv("KMCF_EVENTS_PERSISTENT") && atoi(getenv("KMCF_EVENTS_PERSISTENT")) == 0) && nn <= 63;
        const long long n_tiles2 = ((long long)count + EV_RT - 1) / EV_RT, n_groups2 = (n_tiles2 + EV_GROUP - 1) / EV_GROUP;
        if (persistent) {
            if (!w->d_rsum &&
                (hipMalloc(reinterpret_cast<void **>(&w->d_rsum), (size_t)std::max(count, 1) * sizeof(double)) != hipSuccess ||
                 hipMalloc(reinterpret_cast<void **>(&w->d_tsum2), (size_t)std::max<long long>(n_tiles2, 1) * sizeof(double)) != hipSuccess ||
                 hipMalloc(reinterpret_cast<void **>(&w->d_gsum2), (size_t)std::max<long long>(n_groups2, 1) * sizeof(double)) != hipSuccess)) fail(KMCF_ERR_HIP);
            if (rc == KMCF_OK) {
                auto lgrid = [](long long n_waves) { return (int)std::min<long long>(std::max<long long>((n_waves + 3) / 4, 1), 65536); };
                ev_tree_level_kernel<<<lgrid(count), KMCF_BLOCK, 0, st>>>(0, count, (long long)M, nn, d_prob, w->d_rsum);
                ev_tree_level_kernel<<<lgrid((n_tiles2 + 1) / 2), KMCF_BLOCK, 0, st>>>(1, n_tiles2, count, nn, w->d_rsum, w->d_tsum2);
                ev_tree_level_kernel<<<lgrid(n_groups2), KMCF_BLOCK, 0, st>>>(2, n_groups2, n_tiles2, nn, w->d_tsum2, w->d_gsum2);
            }
        } else {
            group_sum_kernel<<<ggrid, KMCF_BLOCK, 0, st>>>(nb, d_tsum, ng, d_gsum);
        }
        const int BMAX = persistent ? EV_BMAX : 128;
        // (KMCF_EV_TREL: tests shrink the claim range to drive the kernel's out-of-range path)
        const int trel_max = getenv("KMCF_EV_TREL") ? std::min(std::max(atoi(getenv("KMCF_EV_TREL")), 1), EV_TREL) : EV_TREL;
        const bool own_rng = (next_random == kmcf_rng_next);
        if (!w->d_u &&
            (hipMalloc(reinterpret_cast<void **>(&w->d_u), 2 * 512 * sizeof(double)) != hipSuccess ||
             hipMalloc(reinterpret_cast<void **>(&w->d_totlog), 2 * 512 * sizeof(double)) != hipSuccess ||
             hipMalloc(reinterpret_cast<void **>(&w->d_evlog), 3 * 512 * sizeof(int)) != hipSuccess ||
             hipMalloc(reinterpret_cast<void **>(&w->d_state), sizeof(event_batch_state)) != hipSuccess)) fail(KMCF_ERR_HIP);
        double *d_u = w->d_u, *d_totlog = w->d_totlog;
        int *d_evlog = w->d_evlog;
        event_batch_state *d_state = static_cast<event_batch_state *>(w->d_state);
        // Persistent batches hand their results over in pinned host memory, which the kernel writes itself and the host
        // polls (state.seq): no copies in either direction, no sleep in hipStreamSynchronize per batch (a one-event step
        // of the 5 nm device: 0.15 -> see DESIGN 8).  KMCF_EVENTS_PINNED=0: device buffers and copies.
        const bool pinned = persistent && !(getenv("KMCF_EVENTS_PINNED") && atoi(getenv("KMCF_EVENTS_PINNED")) == 0);
        constexpr size_t PIN_LOG = 64, PIN_TOT = PIN_LOG + 3 * EV_BMAX * sizeof(int), PIN_U = PIN_TOT + 2 * EV_BMAX * sizeof(double),
                         PIN_END = PIN_U + 2 * EV_BMAX * sizeof(double);
        if (pinned && !w->h_pin) {
            if (hipHostMalloc(reinterpret_cast<void **>(&w->h_pin), PIN_END, hipHostMallocDefault) != hipSuccess) fail(KMCF_ERR_HIP);
            else { memset(w->h_pin, 0, PIN_END); w->batch_number = 0; }      // the polled number (state.seq) starts below the first one waited for
        }
        std::vector<double> h_u_v(2 * BMAX), h_tot_v(2 * BMAX);
        std::vector<int> h_log_v(3 * BMAX);
        const bool pin_ok = pinned && w->h_pin;
        volatile event_batch_state *p_state = pin_ok ? reinterpret_cast<volatile event_batch_state *>(w->h_pin) : nullptr;
        double *h_u = pin_ok ? reinterpret_cast<double *>(w->h_pin + PIN_U) : h_u_v.data();
        double *h_tot = pin_ok ? reinterpret_cast<double *>(w->h_pin + PIN_TOT) : h_tot_v.data();
        int *h_log = pin_ok ? reinterpret_cast<int *>(w->h_pin + PIN_LOG) : h_log_v.data();
        int B = own_rng ? 4 : 1;
        while (rc == KMCF_OK && t < 1 / freq && nev < max_events) {                      // :418
            const int nbatch = std::min(B, max_events - nev);
            if (own_rng) {
                kmcf_rng peek = *static_cast<kmcf_rng *>(rng_user);
                for (int q = 0; q < 2 * nbatch; ++q) h_u[q] = peek.distribution(peek.rng);
            } else {
                h_u[0] = next_random(rng_user);                                           // :430
                h_u[1] = next_random(rng_user);                                           // :479
            }
            event_batch_state hs = {0, 0, 0, 0};
            if (pin_ok) {
                if (w->batch_number == 0x7fffffff) w->batch_number = 0;
                p_state->done = 0; p_state->n_exec = 0; p_state->seq = 0;      // never equal to the number about to be waited for
            } else if (hipMemcpyAsync(d_u, h_u, 2 * nbatch * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
                       hipMemsetAsync(d_state, 0, sizeof(event_batch_state), st) != hipSuccess) { fail(KMCF_ERR_HIP); break; }
            if (persistent) {
                event_batch_args A;
                A.count = count; A.nn = nn; A.nbatch = nbatch; A.trel_max = trel_max; A.n_tiles = n_tiles2; A.n_groups = n_groups2; A.inv_freq = 1 / freq;
                A.number = pin_ok ? ++w->batch_number : 0;
                // the supertile sums in LDS where they fit beside the kernel's own arrays (KMCF_EVENTS_ST=0: the round-3 walk)
                A.n_st = 0;
                size_t dyn = 0;
                {
                    static const bool st_on = !(getenv("KMCF_EVENTS_ST") && atoi(getenv("KMCF_EVENTS_ST")) == 0);
                    const long long n_st = (n_tiles2 + EV_ST - 1) / EV_ST;
                    hipFuncAttributes fa;
                    if (st_on && n_groups2 <= EV_GLDS && n_st <= EV_STMAX &&
                        hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(event_batch_kernel)) == hipSuccess &&
                        fa.sharedSizeBytes + (size_t)n_st * sizeof(double) <= (size_t)160 * 1024 &&
                        hipFuncSetAttribute(reinterpret_cast<const void *>(event_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(n_st * sizeof(double))) == hipSuccess) {
                        A.n_st = (int)n_st;
                        dyn = (size_t)n_st * sizeof(double);
                    }
                    (void)hipGetLastError();
                }
                if (pin_ok)
                    event_batch_kernel<<<1, EV_PB, dyn, st>>>(A, d_prob, d_type, d_neigh_idx, w->d_rsum, w->d_tsum2, w->d_gsum2, d_site_element,
                                                                  d_site_charge, h_log, h_tot, h_u, const_cast<event_batch_state *>(p_state));
                else
                    event_batch_kernel<<<1, EV_PB, dyn, st>>>(A, d_prob, d_type, d_neigh_idx, w->d_rsum, w->d_tsum2, w->d_gsum2, d_site_element,
                                                                  d_site_charge, d_evlog, d_totlog, d_u, d_state);
            }
            for (int ev = 0; ev < nbatch && !persistent; ++ev) {
                select_event_kernel<true><<<1, KMCF_BLOCK, 0, st>>>(M, nb, ng, start_i, nn, -1.0, 0.0, d_gsum, d_tsum, d_prob, d_type,
                                                                    d_neigh_idx, d_evlog, d_totlog, d_site_element, d_site_charge,
                                                                    d_aff, ev, d_u, d_state, 1 / freq);
                tile_sum_kernel<<<n_aff, KMCF_BLOCK, 0, st>>>(M, d_prob, d_aff, d_tsum);
                group_sum_aff_kernel<<<n_aff, KMCF_BLOCK, 0, st>>>(d_aff, nb, d_tsum, d_gsum);
            }
            if (pin_ok) {
                if (hipGetLastError() != hipSuccess) { fail(KMCF_ERR_HIP); break; }
                const int number = w->batch_number;
                const auto t_poll = std::chrono::steady_clock::now();
                int spins = 0;
                bool synced = false;
                while (__atomic_load_n(const_cast<const int *>(&p_state->seq), __ATOMIC_ACQUIRE) != number) {
                    if ((++spins & 255) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count() > 10e-3) {
                        if (hipStreamSynchronize(st) != hipSuccess) rc = KMCF_ERR_HIP;     // (then the number is there, or the kernel failed)
                        synced = true;
                        break;
                    }
                }
                if (rc != KMCF_OK || (synced && p_state->seq != number)) { fail(KMCF_ERR_HIP); break; }
                hs.done = p_state->done; hs.n_exec = p_state->n_exec;
            } else if (hipMemcpyAsync(&hs, d_state, sizeof(hs), hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipMemcpyAsync(h_log, d_evlog, 3 * nbatch * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipMemcpyAsync(h_tot, d_totlog, 2 * nbatch * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                       hipStreamSynchronize(st) != hipSuccess) { fail(KMCF_ERR_HIP); break; }
            const int n = hs.n_exec;
            for (int e = 0; e < n; ++e)
                if (h_event_log) for (int q = 0; q < 3; ++q) h_event_log[3 * (nev + e) + q] = h_log[3 * e + q];
            // residence time of the last executed event as the device computed it (:479); whether it ends the
            // step was decided there too (hs.done), so the two can never disagree
            if (n > 0) t = h_tot[2 * (n - 1) + 1];
            if (own_rng) for (int q = 0; q < 2 * n; ++q) next_random(rng_user);           // consume what was used
            nev += n;
            if (rc == KMCF_OK && (hs.done == 2 || n == 0)) {
                kmcf_set_error("kmcf_execute_kmc_step: no event could be selected (total rate %g)", n < nbatch ? h_tot[2 * n] : 0.0);
                fail(KMCF_ERR_STATE);
            }
            if (own_rng && n == nbatch && hs.done == 0 && B < BMAX) B *= 2;
        }
    }
    while (!fast && rc == KMCF_OK && t < 1 / freq && nev < max_events) {                 // :418
        if (M > 0) group_sum_kernel<<<ggrid, KMCF_BLOCK, 0, st>>>(nb, d_tsum, ng, d_gsum);
        double total = 0.0;
        int ij[3];
        int source_rank = 0;
        {
            total_kernel<<<1, 64, 0, st>>>(ng, d_gsum, d_tot + rank);
            if (kmcf_comm_allgatherv_double(c, d_tot, ones.data(), iota.data()) != KMCF_OK) { fail(KMCF_ERR_COMM); break; }   // MPI_Allgather :423
            if (hipMemcpyAsync(totals.data(), d_tot, (size_t)P * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { fail(KMCF_ERR_HIP); break; }
            for (int q = 1; q < P; ++q) totals[q] += totals[q - 1];                       // :425-427
            total = totals[P - 1];
            double number = next_random(rng_user) * total;                                // :430
            source_rank = P - 1;
            for (int q = 0; q < P; ++q)
                if (number < totals[q]) { source_rank = q; break; }                       // :432-437
            if (hipMemsetAsync(d_ij + 3 * rank, 0xff, 3 * sizeof(int), st) != hipSuccess) { fail(KMCF_ERR_HIP); break; }
            if (rank == source_rank && M > 0) {
                if (rank > 0) number -= totals[rank - 1];                                 // :440-442
                if (number < 0) number = 0;
                select_event_kernel<false><<<1, KMCF_BLOCK, 0, st>>>(M, nb, ng, start_i, nn, number, 0.0, d_gsum, d_tsum, d_prob,
                                                                     d_type, d_neigh_idx, d_ij + 3 * rank, d_tot + rank,
                                                                     nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0.0);
            }
            if (kmcf_comm_allgatherv_int(c, d_ij, threes.data(), iota3.data()) != KMCF_OK) { fail(KMCF_ERR_COMM); break; }  // MPI_Bcast :455-459
            if (hipMemcpyAsync(ij, d_ij + 3 * source_rank, 3 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { fail(KMCF_ERR_HIP); break; }
            if (ij[0] >= 0 && ij[0] < N && ij[1] >= 0 && ij[1] < N) {
                execute_event_kernel<<<1, 64, 0, st>>>(d_site_element, d_site_charge, d_ij + 3 * source_rank);
                if (M > 0) zero_and_sum_kernel<<<nb, KMCF_BLOCK, 0, st>>>(M, start_i, nn, d_neigh_idx, d_type, d_prob, ij[0], ij[1], d_tsum);
            }
        }
        if (ij[0] < 0 || ij[0] >= N || ij[1] < 0 || ij[1] >= N) {
            kmcf_set_error("kmcf_execute_kmc_step: no event could be selected (total rate %g)", total);
            fail(KMCF_ERR_STATE);
            break;
        }
        if (h_event_log) { h_event_log[3 * nev] = ij[0]; h_event_log[3 * nev + 1] = ij[1]; h_event_log[3 * nev + 2] = ij[2]; }
        t = -std::log(next_random(rng_user)) / total;                                     // :479
        ++nev;
    }
    if (rc == KMCF_OK && hipStreamSynchronize(st) != hipSuccess) rc = KMCF_ERR_HIP;
    if (rc == KMCF_ERR_HIP) kmcf_set_error("kmcf_execute_kmc_step: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    *event_time = t;
    if (n_events) *n_events = nev;
    return rc;
}
