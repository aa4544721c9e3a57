// Peer-to-peer transport: the CG's per-iteration exchanges done by the kernels themselves over mapped peer
// memory (xGMI between the GPUs of a node), instead of one RCCL operation per exchange.
//
// Why: at 8 ranks one rank's share of the 40 nm iteration is ~20 us of kernels; an ncclAllReduce of 3 doubles plus
// a grouped ncclSend/ncclRecv per iteration cost more than that.  What the exchanges need is tiny
// -- 8..24 bytes to every peer for the dot products, ~100 KB to two slab neighbours for the halo -- so each rank
// exposes one WINDOW of device memory (IPC-mapped by every peer, fine-grained so that remote stores are coherent),
// and
//   * all-reduce  = one 1-block kernel: store my partials into slot [parity][my rank] of EVERY peer's window, raise
//                   that peer's flag, wait for the P flags of my own window, add the P slots in rank order
//                   (deterministic: every rank adds the same numbers in the same order);
//   * halo        = the pack kernel stores p[row] straight into the neighbour's landing zone and its last block
//                   raises the neighbour's flag; a small kernel on the receiving side waits for its flags and moves
//                   the landing zone behind p_local (the halo slots of the one SpMV vector);
//   * all-gatherv = (once per KMC step: charges, solution, tunnel sub-vector) stage my slice in my window, raise
//                   flags, pull the peers' slices, acknowledge.
// Flags are sequence numbers (monotonic, never reset): store data -> system-scope fence -> flag store (release);
// poll with acquire loads.  All-reduce slots are reused by parity: a rank can be at most one all-reduce ahead of
// another (it needs every peer's contribution to finish one).  The halo protocol assumes NOTHING about what runs
// between two exchanges: landing zones are double-buffered by sequence parity and every consumer acknowledges the
// sequence it has finished reading into the SENDER's window; put(s) first waits (on its own window: local memory)
// for ack >= s - 2, so buffer s & 1 is rewritten only after its previous content (s - 2) was consumed, and a flag
// that already reads s + 1 when the consumer of s looks at it still means "buffer s & 1 holds s" (s + 2 cannot have
// been put).  Back-to-back SpMVs, the CG's initial A x0 followed by A z, benches: all safe.  EVERY wait is bounded (wall clock): on
// expiry the kernel records an error and returns, later waits return at once, and the host reports KMCF_ERR_COMM
// -- no hang.  RCCL stays available as the fallback transport and as the cross-check (tests compare iterates).
//
// Bootstrap: through RCCL (handles all-gathered on the existing communicator) or, without RCCL, through the host
// program (kmcf_comm_p2p_export / kmcf_comm_p2p_import; tests use torch.distributed gloo and two processes on one
// GPU, where RCCL refuses to run).  Members of an in-process loopback group exchange plain pointers.
#include <cstring>

#include "kmcf_p2p_dev.hpp"

namespace {

// One block.  buf[0..count): my partials in, the sums over all ranks out.
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_allreduce_kernel(char *const *__restrict__ peer, int P, int rank, double *__restrict__ buf,
                                                                   int count, u64 seq, long long timeout, int *d_err, int *h_err)
{
    const int t = threadIdx.x, parity = (int)(seq & 1);
    if (t < P) {
        double *slot = reinterpret_cast<double *>(peer[t] + P2P_OFF_RED_SLOT) + ((size_t)parity * P2P_MAXR + rank) * P2P_FS;
        for (int i = 0; i < count; ++i)
            __hip_atomic_store(reinterpret_cast<u64 *>(&slot[i]), (u64)__double_as_longlong(buf[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        store_release_system(reinterpret_cast<u64 *>(peer[t] + P2P_OFF_RED_FLAG) + ((size_t)parity * P2P_MAXR + rank) * P2P_FS, seq);
    }
    __syncthreads();
    if (t < P) wait_ge(reinterpret_cast<const u64 *>(peer[rank] + P2P_OFF_RED_FLAG) + ((size_t)parity * P2P_MAXR + t) * P2P_FS, seq, timeout, d_err, h_err, 1);
    __syncthreads();
    if (t < count && __hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const double *slots = reinterpret_cast<const double *>(peer[rank] + P2P_OFF_RED_SLOT) + (size_t)parity * P2P_MAXR * P2P_FS;
        double s = 0.0;
        for (int q = 0; q < P; ++q)            // rank order on every rank: identical sums everywhere
            s += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(&slots[(size_t)q * P2P_FS + t]), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_SYSTEM));
        buf[t] = s;
    }
}

// The same with the block-level finalize in front: value i = sum of the partial arrays of part[i] (fixed order),
// result into S->red[i].  One launch instead of a finalize kernel plus an exchange kernel per reduction.
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_allreduce_parts_kernel(char *const *__restrict__ peer, int P, int rank, kmcf_part4 p0, kmcf_part4 p1,
                                                                         kmcf_part4 p2, int count, kmcf_scalars *__restrict__ S, int skip_if_done,
                                                                         u64 seq, long long timeout, int *d_err, int *h_err)
{
    __shared__ double lds4[4];
    __shared__ double mine[4];
    if (skip_if_done && S->done) return;
    const int t = threadIdx.x, parity = (int)(seq & 1);
    for (int i = 0; i < count; ++i) {
        const kmcf_part4 &r = i == 0 ? p0 : (i == 1 ? p1 : p2);
        double v = 0.0;
        for (int q = 0; q < 4; ++q)
            for (int j = t; j < r.n[q]; j += KMCF_BLOCK) v += r.p[q][j];
        v = kmcf_wave_sum64(v);
        if ((t & 63) == 0) lds4[t >> 6] = v;
        __syncthreads();
        if (t == 0) mine[i] = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
        __syncthreads();
    }
    if (t < P) {
        double *slot = reinterpret_cast<double *>(peer[t] + P2P_OFF_RED_SLOT) + ((size_t)parity * P2P_MAXR + rank) * P2P_FS;
        for (int i = 0; i < count; ++i)
            __hip_atomic_store(reinterpret_cast<u64 *>(&slot[i]), (u64)__double_as_longlong(mine[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        store_release_system(reinterpret_cast<u64 *>(peer[t] + P2P_OFF_RED_FLAG) + ((size_t)parity * P2P_MAXR + rank) * P2P_FS, seq);
    }
    __syncthreads();
    if (t < P) wait_ge(reinterpret_cast<const u64 *>(peer[rank] + P2P_OFF_RED_FLAG) + ((size_t)parity * P2P_MAXR + t) * P2P_FS, seq, timeout, d_err, h_err, 1);
    __syncthreads();
    if (t < count && __hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const double *slots = reinterpret_cast<const double *>(peer[rank] + P2P_OFF_RED_SLOT) + (size_t)parity * P2P_MAXR * P2P_FS;
        double s = 0.0;
        for (int q = 0; q < P; ++q)
            s += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(&slots[(size_t)q * P2P_FS + t]), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_SYSTEM));
        S->red[t] = s;
    }
}

// pack + put: p[send_idx[i]] -> buffer (seq & 1) of the neighbour's landing zone, once every neighbour has
// acknowledged seq - 2 (the buffer's previous content); the last block to finish raises the neighbours' flags
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_put_kernel(int n_send, const int *__restrict__ send_idx, const double *__restrict__ p,
                                                             double *const *__restrict__ put_ptr, const long long *__restrict__ put_stride,
                                                             int n_nb, u64 *const *__restrict__ put_flag, const u64 *__restrict__ acks,
                                                             u64 seq, long long timeout, int *d_err, int *h_err,
                                                             unsigned int *__restrict__ ctr, const kmcf_scalars *__restrict__ S, int check_done)
{
    __shared__ int s_last;
    if (check_done && S->done) return;
    if ((int)threadIdx.x < n_nb && seq > 2) wait_ge(&acks[threadIdx.x * P2P_FS], seq - 2, timeout, d_err, h_err, 5);
    __syncthreads();
    const long long par = (long long)(seq & 1);
    if (__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)      // (never overwrite an unconsumed buffer)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_send; i += gridDim.x * blockDim.x)
            __hip_atomic_store(reinterpret_cast<u64 *>(put_ptr[i] + par * put_stride[i]), (u64)__double_as_longlong(p[send_idx[i]]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if ((int)threadIdx.x < n_nb) store_release_system(put_flag[threadIdx.x], seq);
    if (threadIdx.x == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait for the flags of all neighbours, move buffer (seq & 1) of the landing zone behind p_local, then -- the last
// block to finish -- acknowledge seq to every sender (into ITS window)
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_halo_wait_kernel(int n_nb, const u64 *__restrict__ flags, u64 seq, long long timeout,
                                                                   int *d_err, int *h_err, int n_halo, const double *__restrict__ landing,
                                                                   double *__restrict__ halo_dst, u64 *const *__restrict__ ack_ptr,
                                                                   unsigned int *__restrict__ ctr)
{
    __shared__ int s_last;
    if ((int)threadIdx.x < n_nb) wait_ge(&flags[threadIdx.x * P2P_FS], seq, timeout, d_err, h_err, 2);
    __syncthreads();
    if (__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const double *src = landing + (size_t)(seq & 1) * n_halo;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_halo; i += gridDim.x * blockDim.x)
        halo_dst[i] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(&src[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    __syncthreads();                                     // every load of this block has returned (its value was stored)
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    if ((int)threadIdx.x < n_nb) store_release_system(ack_ptr[threadIdx.x], seq);
    if (threadIdx.x == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acknowledgement alone (direct protocol: an SpMV whose successor kernel does not acknowledge)
__global__ __launch_bounds__(64) void p2p_ack_kernel(int n_nb, u64 *const *__restrict__ ack_ptr, u64 seq, const kmcf_scalars *__restrict__ S,
                                                    int check_done)
{
    if (check_done && S->done) return;
    if ((int)threadIdx.x < n_nb) store_release_system(ack_ptr[threadIdx.x], seq);
}

// gather, step 1: my slice into my staging half (after every peer has consumed what was there two gathers ago),
// then my sequence number to every peer
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_gather_stage_kernel(char *const *__restrict__ peer, int P, int rank, const unsigned int *__restrict__ src,
                                                                      unsigned int *__restrict__ stage, size_t n_words, u64 seq, long long timeout,
                                                                      int *d_err, int *h_err, unsigned int *__restrict__ ctr)
{
    __shared__ int s_last;
    if ((int)threadIdx.x < P && seq > 2)
        wait_ge(reinterpret_cast<const u64 *>(peer[rank] + P2P_OFF_G_ACK) + threadIdx.x * P2P_FS, seq - 2, timeout, d_err, h_err, 4);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(&stage[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if ((int)threadIdx.x < P) store_release_system(reinterpret_cast<u64 *>(peer[threadIdx.x] + P2P_OFF_G_FLAG) + rank * P2P_FS, seq);
#ifdef KMCF_P2P_DEBUG
    if ((int)threadIdx.x < P) printf("gather stage: rank %d wrote seq %llu to %p (peer %d)\n", rank, seq, (void *)(reinterpret_cast<u64 *>(peer[threadIdx.x] + P2P_OFF_G_FLAG) + rank * P2P_FS), (int)threadIdx.x);
#endif
    if (threadIdx.x == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct p2p_word_offsets { size_t off[P2P_MAXR + 1]; };

// gather, step 2: pull every peer's slice out of its staging half, then acknowledge
__global__ __launch_bounds__(KMCF_BLOCK) void p2p_gather_pull_kernel(char *const *__restrict__ peer, int P, int rank, size_t stage_byte_off,
                                                                     p2p_word_offsets wo /* P + 1, in 4-byte words of the gathered buffer */,
                                                                     unsigned int *__restrict__ dst, u64 seq, long long timeout, int *d_err, int *h_err,
                                                                     unsigned int *__restrict__ ctr)
{
    __shared__ int s_last;
    if ((int)threadIdx.x < P) wait_ge(reinterpret_cast<const u64 *>(peer[rank] + P2P_OFF_G_FLAG) + threadIdx.x * P2P_FS, seq, timeout, d_err, h_err, 3);
    __syncthreads();
    if (__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        for (int q = 0; q < P; ++q) {
            if (q == rank) continue;
            const unsigned int *src = reinterpret_cast<const unsigned int *>(peer[q] + stage_byte_off);
            for (size_t i = wo.off[q] + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < wo.off[q + 1]; i += (size_t)gridDim.x * blockDim.x)
                dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if ((int)threadIdx.x < P) store_release_system(reinterpret_cast<u64 *>(peer[threadIdx.x] + P2P_OFF_G_ACK) + rank * P2P_FS, seq);
    if (threadIdx.x == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

int kmcf_p2p_destroy(kmcf_comm *c)
{
    kmcf_p2p *w = c->p2p;
    if (!w) return KMCF_OK;
    for (int q = 0; q < (int)w->peer.size(); ++q)
        if (q != w->rank && w->ipc_opened[q] && w->peer[q]) hipIpcCloseMemHandle(w->peer[q]);
    if (w->win) hipFree(w->win);
    if (w->d_peer) hipFree(w->d_peer);
    if (w->d_err) hipFree(w->d_err);
    if (w->d_ctr) hipFree(w->d_ctr);
    if (w->h_err) hipHostFree(w->h_err);
    delete w;
    c->p2p = nullptr;
    c->p2p_active = false;
    return KMCF_OK;
}

// window + bookkeeping of this rank (peers come later)
int kmcf_p2p_create(kmcf_comm *c)
{
    if (c->p2p) return KMCF_OK;
    KMCF_CHECK(c->nranks <= P2P_MAXR, KMCF_ERR_ARG, "p2p transport: %d ranks exceed %d", c->nranks, P2P_MAXR);
    KMCF_HIP(hipSetDevice(c->device));
    kmcf_p2p *w = new kmcf_p2p();
    c->p2p = w;
    w->nranks = c->nranks;
    w->rank = c->rank;
    size_t mb = 96;
    if (const char *e = getenv("KMCF_P2P_WINDOW_MB")) mb = (size_t)std::max(8, atoi(e));
    w->win_bytes = mb << 20;
    // fine-grained device memory: stores arriving over xGMI are coherent with the owner's reads; a plain
    // allocation serves where the runtime refuses (same-device tests)
    if (hipExtMallocWithFlags(reinterpret_cast<void **>(&w->win), w->win_bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        w->fine_grained = true;
    } else {
        (void)hipGetLastError();
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&w->win), w->win_bytes));
    }
    if (getenv("KMCF_P2P_VERBOSE")) fprintf(stderr, "kmcfield p2p: rank %d window %zu MB at %p, %s\n", c->rank, w->win_bytes >> 20, (void *)w->win,
                                            w->fine_grained ? "fine-grained" : "COARSE-grained (hipExtMallocWithFlags refused)");
    KMCF_HIP(hipMemset(w->win, 0, P2P_OFF_BUMP));
    w->stage_half = align_up((w->win_bytes - P2P_OFF_BUMP) / 4, 4096);       // half of the window for the two staging halves
    w->stage_off = P2P_OFF_BUMP;
    w->bump = w->stage_off + 2 * w->stage_half;
    w->peer.assign((size_t)c->nranks, nullptr);
    w->ipc_opened.assign((size_t)c->nranks, false);
    w->peer[c->rank] = w->win;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&w->d_peer), (size_t)c->nranks * sizeof(char *)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&w->d_err), sizeof(int)));
    KMCF_HIP(hipMemset(w->d_err, 0, sizeof(int)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&w->d_ctr), 4 * sizeof(unsigned int)));
    KMCF_HIP(hipMemset(w->d_ctr, 0, 4 * sizeof(unsigned int)));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&w->h_err), sizeof(int), hipHostMallocDefault));
    *w->h_err = 0;
    double ms = 10000.0;                               // (ranks may enter a solve seconds apart: set-up, IO)
    if (const char *e = getenv("KMCF_P2P_TIMEOUT_MS")) ms = atof(e);
    int khz = 0;                                       // wall_clock64() tick rate of this device
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    w->timeout_ticks = (long long)(ms * khz);
    return KMCF_OK;
}

static int p2p_finish_peers(kmcf_comm *c)
{
    kmcf_p2p *w = c->p2p;
    KMCF_HIP(hipMemcpy(w->d_peer, w->peer.data(), (size_t)c->nranks * sizeof(char *), hipMemcpyHostToDevice));
    KMCF_HIP(hipDeviceSynchronize());
    c->p2p_active = true;
    return KMCF_OK;
}

int kmcf_p2p_set_peers_direct(kmcf_comm *c, char *const *bases)
{
    for (int q = 0; q < c->nranks; ++q) c->p2p->peer[q] = bases[q];
    return p2p_finish_peers(c);
}

char *kmcf_p2p_window(kmcf_comm *c) { return c->p2p ? c->p2p->win : nullptr; }

bool kmcf_p2p_fits(kmcf_comm *c, size_t gather_bytes) { return c->p2p && gather_bytes <= c->p2p->stage_half; }

extern "C" int kmcf_comm_p2p_export(kmcf_comm *c, void *h_handle)
{
    KMCF_CHECK(c && h_handle, KMCF_ERR_ARG, "kmcf_comm_p2p_export: null argument");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_comm_p2p_export: host-only communicator");
    static_assert(sizeof(hipIpcMemHandle_t) == KMCF_P2P_HANDLE_BYTES, "IPC handle size");
    KMCF_TRY(kmcf_p2p_create(c));
    hipIpcMemHandle_t h;
    KMCF_HIP(hipIpcGetMemHandle(&h, c->p2p->win));
    memcpy(h_handle, &h, sizeof(h));
    return KMCF_OK;
}

extern "C" int kmcf_comm_p2p_import(kmcf_comm *c, const void *h_handles)
{
    KMCF_CHECK(c && h_handles && c->p2p, KMCF_ERR_ARG, "kmcf_comm_p2p_import: export first");
    kmcf_p2p *w = c->p2p;
    KMCF_HIP(hipSetDevice(c->device));
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char *>(h_handles) + (size_t)q * sizeof(h), sizeof(h));
        void *p = nullptr;
        KMCF_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        w->peer[q] = static_cast<char *>(p);
        w->ipc_opened[q] = true;
    }
    KMCF_TRY(p2p_finish_peers(c));
    c->connected = true;          // a group may run on this transport alone (no RCCL: two processes on one GPU)
    return KMCF_OK;
}

// error word -> status (after a stream synchronisation)
int kmcf_p2p_check(kmcf_comm *c)
{
    if (!c || !c->p2p || *c->p2p->h_err == 0) return KMCF_OK;
    // (a transport that failed and was switched off -- bench.py's trial, KMCF_TRANSPORT=auto -- must not fail the calls
    // that now run over RCCL; kmcf_comm_select_transport refuses to switch it on again)
    if (!c->p2p_active) return KMCF_OK;
    static const char *what[] = {"", "all-reduce", "halo exchange", "all-gather (data)", "all-gather (acknowledgement)",
                                 "halo exchange (acknowledgement of the put before last)"};
    const int code = *c->p2p->h_err;
    kmcf_set_error("p2p transport: rank %d timed out in %s -- a peer did not arrive within the bound (KMCF_P2P_TIMEOUT_MS)", c->rank,
                   code >= 1 && code <= 5 ? what[code] : "a wait");
    return KMCF_ERR_COMM;
}

int kmcf_p2p_allreduce(kmcf_comm *c, double *d_buf, int count)
{
    kmcf_p2p *w = c->p2p;
    KMCF_CHECK(count >= 1 && count <= 4, KMCF_ERR_ARG, "p2p all-reduce of %d doubles (1..4)", count);
    ++w->seq_red;
    p2p_allreduce_kernel<<<1, KMCF_BLOCK, 0, c->stream>>>(w->d_peer, c->nranks, c->rank, d_buf, count, w->seq_red, w->timeout_ticks, w->d_err, w->h_err);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

int kmcf_p2p_allreduce_parts(kmcf_comm *c, const kmcf_part4 *part, int count, kmcf_scalars *d_S, int skip_if_done)
{
    kmcf_p2p *w = c->p2p;
    KMCF_CHECK(count >= 1 && count <= 3, KMCF_ERR_ARG, "p2p all-reduce of %d partial sets (1..3)", count);
    ++w->seq_red;
    const kmcf_part4 none{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};
    p2p_allreduce_parts_kernel<<<1, KMCF_BLOCK, 0, c->stream>>>(w->d_peer, c->nranks, c->rank, part[0], count > 1 ? part[1] : none,
                                                                count > 2 ? part[2] : none, count, d_S, skip_if_done, w->seq_red,
                                                                w->timeout_ticks, w->d_err, w->h_err);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

// In-place all-gather with uneven counts on the compute stream (elements of 4 or 8 bytes).
int kmcf_p2p_allgatherv(kmcf_comm *c, void *d_buf, const int *counts, const int *displs, size_t elem, hipStream_t st)
{
    if (!st) st = c->stream;
    kmcf_p2p *w = c->p2p;
    const int P = c->nranks, rank = c->rank;
    KMCF_CHECK(elem == 4 || elem == 8, KMCF_ERR_ARG, "p2p all-gather: element size %zu", elem);
    const size_t wpe = elem / 4;
    p2p_word_offsets wo;
    size_t *woff = wo.off;
    for (int q = 0; q < P; ++q) woff[q] = (size_t)displs[q] * wpe;
    woff[P] = ((size_t)displs[P - 1] + counts[P - 1]) * wpe;
    for (int q = 0; q + 1 < P; ++q)
        KMCF_CHECK(woff[q] + (size_t)counts[q] * wpe == woff[q + 1], KMCF_ERR_ARG, "p2p all-gather: slices must be contiguous");
    KMCF_CHECK(woff[P] * 4 <= w->stage_half, KMCF_ERR_NOMEM, "p2p all-gather of %zu bytes exceeds the staging area (%zu; KMCF_P2P_WINDOW_MB)",
               woff[P] * 4, w->stage_half);
    ++w->seq_gather;
    const size_t half_off = w->stage_off + (w->seq_gather & 1) * w->stage_half;
    const size_t my_words = (size_t)counts[rank] * wpe;
    // Few blocks: every block of these kernels spins on flags, and ranks that share one GPU (the in-process test
    // groups, several processes on one device) must never fill it with waiting blocks while the kernel that would
    // release them cannot be scheduled (seen once: 4 ranks x 512 waiting blocks = every block slot of the chip).
    // The link, not the block count, bounds a gather anyway.
    const int g1 = (int)std::min<size_t>(std::max<size_t>(my_words / (KMCF_BLOCK * 4), 1), 48);
    const int g2 = (int)std::min<size_t>(std::max<size_t>(woff[P] / (KMCF_BLOCK * 4), 1), 48);
    unsigned int *buf32 = static_cast<unsigned int *>(d_buf);
    p2p_gather_stage_kernel<<<g1, KMCF_BLOCK, 0, st>>>(w->d_peer, P, rank, buf32 + woff[rank],
                                                              reinterpret_cast<unsigned int *>(w->win + half_off) + woff[rank], my_words,
                                                              w->seq_gather, w->timeout_ticks, w->d_err, w->h_err, w->d_ctr);
    p2p_gather_pull_kernel<<<g2, KMCF_BLOCK, 0, st>>>(w->d_peer, P, rank, half_off, wo, buf32, w->seq_gather, w->timeout_ticks,
                                                             w->d_err, w->h_err, w->d_ctr + 1);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;              // asynchronous like the RCCL path; a timeout shows in kmcf_p2p_check after the next sync
}

// ---------------------------------------------------------------- halo protocol of one matrix
int kmcf_p2p_matrix_alloc(kmcf_matrix *m, int *land_off8, int *flag_off8, int *ack_off8, int *ll_off8, int *red_off8)
{
    kmcf_p2p *w = m->comm->p2p;
    kmcf_p2p_halo *h = new kmcf_p2p_halo();
    m->p2p = h;
    const int n_nb = m->number_of_neighbours - 1;
    h->land_off = align_up(w->bump, 256);
    h->flag_off = align_up(h->land_off + 2 * (size_t)std::max(m->n_halo, 1) * sizeof(double), 256);     // two buffers (sequence parity)
    h->ack_off = align_up(h->flag_off + (size_t)std::max(n_nb, 1) * P2P_FS * sizeof(u64), 256);      // (one line per flag)
    // granule zone and reduction zone of the register-resident solve (kmcf_cgr.hip): zeroed with the flags -- a granule's
    // sequence number is never 0
    h->ll_off = align_up(h->ack_off + (size_t)std::max(n_nb, 1) * P2P_FS * sizeof(u64), 256);
    h->red_off = align_up(h->ll_off + 2 * (size_t)std::max(m->n_halo, 1) * 2 * sizeof(u64), 256);
    const size_t end = h->red_off + 2 * (size_t)P2P_MAXR * P2P_FS * sizeof(u64);
    KMCF_CHECK(end <= w->win_bytes, KMCF_ERR_NOMEM, "p2p window exhausted (%zu of %zu bytes; KMCF_P2P_WINDOW_MB)", end, w->win_bytes);
    w->bump = end;
    KMCF_HIP(hipMemset(w->win + h->flag_off, 0, end - h->flag_off));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_ctr), 2 * sizeof(unsigned int)));
    KMCF_HIP(hipMemset(h->d_ctr, 0, 2 * sizeof(unsigned int)));
    *land_off8 = (int)(h->land_off / 8);
    *flag_off8 = (int)(h->flag_off / 8);
    *ack_off8 = (int)(h->ack_off / 8);
    *ll_off8 = (int)(h->ll_off / 8);
    *red_off8 = (int)(h->red_off / 8);
    return KMCF_OK;
}

// r_land8[k], r_flag8[k], r_ack8[k] (k >= 1): where neighbour k wants MY data / my flag / my acknowledgement of ITS
// puts inside ITS window (units of 8 bytes); r_halo[k]: its halo size = distance between its two landing buffers
int kmcf_p2p_matrix_connect(kmcf_matrix *m, const std::vector<long long> &r_land8, const std::vector<long long> &r_flag8,
                            const std::vector<long long> &r_ack8, const std::vector<long long> &r_halo,
                            const std::vector<long long> &r_ll8 /* per neighbour: first granule of MY values in ITS zone */,
                            const std::vector<long long> &r_red8 /* per RANK: its reduction zone */)
{
    kmcf_p2p *w = m->comm->p2p;
    kmcf_p2p_halo *h = m->p2p;
    const int nnb = m->number_of_neighbours;
    std::vector<double *> put((size_t)std::max(m->n_send, 1), nullptr);
    std::vector<long long> stride((size_t)std::max(m->n_send, 1), 0);
    std::vector<u64 *> flg((size_t)std::max(nnb - 1, 1), nullptr), ack((size_t)std::max(nnb - 1, 1), nullptr);
    std::vector<u64 *> put_ll((size_t)std::max(m->n_send, 1), nullptr);
    for (int k = 1; k < nnb; ++k) {
        char *base = w->peer[m->neighbours[k]];
        double *land = reinterpret_cast<double *>(base) + r_land8[k];
        u64 *ll = reinterpret_cast<u64 *>(base) + r_ll8[k];
        for (size_t i = 0; i < m->rows_per_neighbour[k].size(); ++i) {
            put[(size_t)m->send_offset[k] + i] = land + i;
            stride[(size_t)m->send_offset[k] + i] = r_halo[k];
            put_ll[(size_t)m->send_offset[k] + i] = ll + 2 * i;
        }
        flg[(size_t)k - 1] = reinterpret_cast<u64 *>(base) + r_flag8[k];
        ack[(size_t)k - 1] = reinterpret_cast<u64 *>(base) + r_ack8[k];
    }
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_put_ptr), put.size() * sizeof(double *)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_put_stride), stride.size() * sizeof(long long)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_put_flag), flg.size() * sizeof(u64 *)));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_ack_ptr), ack.size() * sizeof(u64 *)));
    KMCF_HIP(hipMemcpy(h->d_put_ptr, put.data(), put.size() * sizeof(double *), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(h->d_put_stride, stride.data(), stride.size() * sizeof(long long), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(h->d_put_flag, flg.data(), flg.size() * sizeof(u64 *), hipMemcpyHostToDevice));
    KMCF_HIP(hipMemcpy(h->d_ack_ptr, ack.data(), ack.size() * sizeof(u64 *), hipMemcpyHostToDevice));
    // the same entries by internal row: the kernel that computes a row puts it (direct protocol)
    {
        const int n = m->n_loc;
        std::vector<int> inv((size_t)std::max(n, 1), 0);
        for (int i = 0; i < n; ++i) inv[m->h_perm.empty() ? i : m->h_perm[i]] = i;
        std::vector<std::vector<int>> per_row((size_t)std::max(n, 1));        // entries (index into put / stride) of every internal row
        for (int k = 1; k < nnb; ++k)
            for (size_t i = 0; i < m->rows_per_neighbour[k].size(); ++i)
                per_row[(size_t)inv[m->rows_per_neighbour[k][i]]].push_back(m->send_offset[k] + (int)i);
        std::vector<int> put_row((size_t)std::max(n, 1), -1), rptr(1, 0);
        std::vector<double *> raddr;
        std::vector<long long> rstride, rllstride;
        std::vector<u64 *> rll;
        for (int i = 0; i < n; ++i) {
            if (per_row[i].empty()) continue;
            put_row[i] = (int)rptr.size() - 1;
            for (int e : per_row[i]) {
                raddr.push_back(put[(size_t)e]); rstride.push_back(stride[(size_t)e]);
                rll.push_back(put_ll[(size_t)e]); rllstride.push_back(2 * stride[(size_t)e]);
            }
            rptr.push_back((int)raddr.size());
        }
        if (raddr.empty()) { raddr.push_back(nullptr); rstride.push_back(0); rll.push_back(nullptr); rllstride.push_back(0); }
        std::vector<u64 *> redp((size_t)m->comm->nranks, nullptr);
        for (int q = 0; q < m->comm->nranks; ++q) redp[(size_t)q] = reinterpret_cast<u64 *>(w->peer[q]) + r_red8[(size_t)q];
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_putr_ll), rll.size() * sizeof(u64 *)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_putr_ll_stride), rllstride.size() * sizeof(long long)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_red_peer), redp.size() * sizeof(u64 *)));
        KMCF_HIP(hipMemcpy(h->d_putr_ll, rll.data(), rll.size() * sizeof(u64 *), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(h->d_putr_ll_stride, rllstride.data(), rllstride.size() * sizeof(long long), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(h->d_red_peer, redp.data(), redp.size() * sizeof(u64 *), hipMemcpyHostToDevice));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_put_row), put_row.size() * sizeof(int)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_putr_ptr), rptr.size() * sizeof(int)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_putr_addr), raddr.size() * sizeof(double *)));
        KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_putr_stride), rstride.size() * sizeof(long long)));
        KMCF_HIP(hipMemcpy(h->d_put_row, put_row.data(), put_row.size() * sizeof(int), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(h->d_putr_ptr, rptr.data(), rptr.size() * sizeof(int), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(h->d_putr_addr, raddr.data(), raddr.size() * sizeof(double *), hipMemcpyHostToDevice));
        KMCF_HIP(hipMemcpy(h->d_putr_stride, rstride.data(), rstride.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    return KMCF_OK;
}

bool kmcf_p2p_direct(const kmcf_matrix *m)
{
    const char *e = getenv("KMCF_P2P_DIRECT");              // (read per call: bench.py times the protocols against each other)
    const bool off = e && atoi(e) == 0;
    const kmcf_comm *c = m->comm;
    // long rows and the tunnel sub-block read the halo behind p_local: they keep the copying protocol
    return !off && c->p2p_active && c->nranks > 1 && m->p2p && m->p2p->d_put_row && m->n_long_items == 0 && !m->sub;
}

kmcf_p2p_dev kmcf_p2p_dev_of(const kmcf_matrix *m)
{
    const kmcf_p2p *w = m->comm->p2p;
    const kmcf_p2p_halo *h = m->p2p;
    kmcf_p2p_dev d;
    d.peer = w->d_peer; d.P = m->comm->nranks; d.rank = m->comm->rank; d.timeout = w->timeout_ticks; d.d_err = w->d_err; d.h_err = w->h_err;
    d.n_nb = m->number_of_neighbours - 1;
    d.flags = reinterpret_cast<const u64 *>(w->win + h->flag_off);
    d.acks = reinterpret_cast<const u64 *>(w->win + h->ack_off);
    d.ack_ptr = h->d_ack_ptr; d.put_flag = h->d_put_flag;
    d.landing = reinterpret_cast<const double *>(w->win + h->land_off);
    d.n_halo = std::max(m->n_halo, 1);
    d.put_row = h->d_put_row; d.putr_ptr = h->d_putr_ptr; d.putr_addr = h->d_putr_addr; d.putr_stride = h->d_putr_stride;
    d.ctr = h->d_ctr;
    return d;
}

unsigned long long kmcf_p2p_next_red_seq(kmcf_comm *c) { return ++c->p2p->seq_red; }
unsigned long long kmcf_p2p_red_seq(const kmcf_comm *c) { return c->p2p->seq_red; }
void kmcf_p2p_set_red_seq(kmcf_comm *c, unsigned long long v) { c->p2p->seq_red = v; }
unsigned long long *kmcf_p2p_halo_seq(kmcf_matrix *m, int which) { return which ? &m->p2p->seq_put : &m->p2p->seq; }

int kmcf_p2p_direct_put(kmcf_matrix *m, unsigned long long seq, bool skip_if_done)
{
    kmcf_comm *c = m->comm;
    kmcf_p2p *w = c->p2p;
    kmcf_p2p_halo *h = m->p2p;
    const int n_nb = m->number_of_neighbours - 1;
    if (n_nb <= 0) return KMCF_OK;
    const int g = std::max(1, std::min((m->n_send + KMCF_BLOCK - 1) / KMCF_BLOCK, 64));
    p2p_put_kernel<<<g, KMCF_BLOCK, 0, c->stream>>>(m->n_send, m->d_send_idx, m->d_p, h->d_put_ptr, h->d_put_stride, n_nb, h->d_put_flag,
                                                    reinterpret_cast<const u64 *>(w->win + h->ack_off), seq, w->timeout_ticks, w->d_err, w->h_err,
                                                    h->d_ctr, m->d_S, skip_if_done ? 1 : 0);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

int kmcf_p2p_direct_ack(kmcf_matrix *m, unsigned long long seq, bool skip_if_done)
{
    const int n_nb = m->number_of_neighbours - 1;
    if (n_nb <= 0) return KMCF_OK;
    p2p_ack_kernel<<<1, 64, 0, m->comm->stream>>>(n_nb, m->p2p->d_ack_ptr, seq, m->d_S, skip_if_done ? 1 : 0);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}

void kmcf_p2p_matrix_free(kmcf_matrix *m)
{
    if (!m->p2p) return;
    if (m->p2p->d_put_ptr) hipFree(m->p2p->d_put_ptr);
    if (m->p2p->d_put_stride) hipFree(m->p2p->d_put_stride);
    if (m->p2p->d_put_flag) hipFree(m->p2p->d_put_flag);
    if (m->p2p->d_ack_ptr) hipFree(m->p2p->d_ack_ptr);
    if (m->p2p->d_put_row) hipFree(m->p2p->d_put_row);
    if (m->p2p->d_putr_ptr) hipFree(m->p2p->d_putr_ptr);
    if (m->p2p->d_putr_addr) hipFree(m->p2p->d_putr_addr);
    if (m->p2p->d_putr_stride) hipFree(m->p2p->d_putr_stride);
    if (m->p2p->d_putr_ll) hipFree(m->p2p->d_putr_ll);
    if (m->p2p->d_putr_ll_stride) hipFree(m->p2p->d_putr_ll_stride);
    if (m->p2p->d_red_peer) hipFree(m->p2p->d_red_peer);
    if (m->p2p->d_ctr) hipFree(m->p2p->d_ctr);
    delete m->p2p;
    m->p2p = nullptr;
}

// on the comm stream, after the compute stream's packed event: put my rows, wait for my halo
int kmcf_p2p_halo_exchange(kmcf_matrix *m)
{
    kmcf_comm *c = m->comm;
    kmcf_p2p *w = c->p2p;
    kmcf_p2p_halo *h = m->p2p;
    const int n_nb = m->number_of_neighbours - 1;
    if (n_nb <= 0) return KMCF_OK;
    KMCF_CHECK(h != nullptr, KMCF_ERR_STATE, "p2p halo: matrix was built before the transport was up");
    ++h->seq;
    h->seq_put = h->seq;
    const int g = std::max(1, std::min((m->n_send + KMCF_BLOCK - 1) / KMCF_BLOCK, 64));
    p2p_put_kernel<<<g, KMCF_BLOCK, 0, c->comm_stream>>>(m->n_send, m->d_send_idx, m->d_p, h->d_put_ptr, h->d_put_stride, n_nb, h->d_put_flag,
                                                         reinterpret_cast<const u64 *>(w->win + h->ack_off), h->seq, w->timeout_ticks, w->d_err,
                                                         w->h_err, h->d_ctr, nullptr, 0);
    const int g2 = std::max(1, std::min((m->n_halo + KMCF_BLOCK * 4 - 1) / (KMCF_BLOCK * 4), 32));
    p2p_halo_wait_kernel<<<g2, KMCF_BLOCK, 0, c->comm_stream>>>(n_nb, reinterpret_cast<const u64 *>(w->win + h->flag_off), h->seq, w->timeout_ticks,
                                                               w->d_err, w->h_err, m->n_halo, reinterpret_cast<const double *>(w->win + h->land_off),
                                                               m->d_p + m->n_loc, h->d_ack_ptr, h->d_ctr + 1);
    KMCF_HIP(hipGetLastError());
    return KMCF_OK;
}
