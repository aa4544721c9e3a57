// Communicator: one rank (= one process = one GPU) of the solver group.
// Replaces the reference's MPI usage on the hot path:
//   MPI_Allreduce of the CG dots      -> ncclAllReduce on device scalars, compute stream
//   MPI_Isend/Irecv of packed halos   -> grouped ncclSend/ncclRecv on the comm stream,
//                                        receiving straight into the compact halo slots
//   MPI_Gatherv/Bcast/Allgatherv      -> grouped ncclBroadcast (uneven counts)
// RCCL is resolved with dlopen at connect time so that 1-rank use needs no RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "kmcf_internal.hpp"

namespace {

thread_local char g_err[1024] = "";

struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return KMCF_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    KMCF_CHECK(h != nullptr, KMCF_ERR_COMM, "cannot dlopen librccl.so.1: %s", dlerror());
#define KMCF_SYM(field, name)                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));      \
    KMCF_CHECK(g_rccl.field != nullptr, KMCF_ERR_COMM, "librccl lacks %s", name)
    KMCF_SYM(GetUniqueId, "ncclGetUniqueId");
    KMCF_SYM(CommInitRank, "ncclCommInitRank");
    KMCF_SYM(CommDestroy, "ncclCommDestroy");
    KMCF_SYM(AllReduce, "ncclAllReduce");
    KMCF_SYM(Broadcast, "ncclBroadcast");
    KMCF_SYM(Send, "ncclSend");
    KMCF_SYM(Recv, "ncclRecv");
    KMCF_SYM(GroupStart, "ncclGroupStart");
    KMCF_SYM(GroupEnd, "ncclGroupEnd");
    KMCF_SYM(CommCount, "ncclCommCount");
    KMCF_SYM(GetErrorString, "ncclGetErrorString");
#undef KMCF_SYM
    g_rccl.handle = h;
    return KMCF_OK;
}

#define KMCF_NCCL(call)                                                                      \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            kmcf_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
            return KMCF_ERR_COMM;                                                            \
        }                                                                                    \
    } while (0)

}  // namespace

void kmcf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *kmcf_last_error(void) { return g_err; }
extern "C" int kmcf_version(void) { return 100; }

extern "C" int kmcf_partition(int nrows, int nranks, int *h_counts, int *h_displs)
{
    // src/KMC_comm.h:249-263
    KMCF_CHECK(nrows >= 0 && nranks > 0 && h_counts && h_displs, KMCF_ERR_ARG, "kmcf_partition: bad arguments");
    int per = nrows / nranks;
    for (int i = 0; i < nranks; ++i) h_counts[i] = (i < nrows % nranks) ? per + 1 : per;
    h_displs[0] = 0;
    for (int i = 1; i < nranks; ++i) h_displs[i] = h_displs[i - 1] + h_counts[i - 1];
    return KMCF_OK;
}

extern "C" int kmcf_comm_create(kmcf_comm **out, int device, int nranks, int rank)
{
    KMCF_CHECK(out && nranks >= 1 && rank >= 0 && rank < nranks, KMCF_ERR_ARG, "kmcf_comm_create: bad rank/nranks");
    if (device < 0) {
        // host-only planning communicator: partition / halo-list logic without a GPU;
        // every compute entry point refuses it (KMCF_ERR_STATE)
        kmcf_comm *c = new kmcf_comm();
        c->device = -1;
        c->nranks = nranks;
        c->rank = rank;
        c->connected = false;
        *out = c;
        return KMCF_OK;
    }
    int ndev = 0;
    KMCF_HIP(hipGetDeviceCount(&ndev));
    KMCF_CHECK(device >= 0 && device < ndev, KMCF_ERR_ARG, "kmcf_comm_create: device %d of %d", device, ndev);
    KMCF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    KMCF_HIP(hipGetDeviceProperties(&prop, device));
    KMCF_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, KMCF_ERR_HIP,
               "libkmcfield is built for gfx950 only, device %d is %s", device, prop.gcnArchName);
    kmcf_comm *c = new kmcf_comm();
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    KMCF_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    KMCF_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    KMCF_HIP(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    KMCF_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    KMCF_HIP(hipEventCreateWithFlags(&c->ev_subpack, hipEventDisableTiming));
    KMCF_HIP(hipEventCreateWithFlags(&c->ev_sub, hipEventDisableTiming));
    KMCF_HIP(hipEventCreate(&c->ev_t0));
    KMCF_HIP(hipEventCreate(&c->ev_t1));
    KMCF_HIP(hipEventCreate(&c->ev_a0));
    KMCF_HIP(hipEventCreate(&c->ev_a1));
    KMCF_HIP(hipEventCreate(&c->ev_call0));
    KMCF_HIP(hipEventCreate(&c->ev_call1));
    KMCF_HIP(hipEventCreateWithFlags(&c->ev_entry, hipEventDisableTiming));
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_scratch), 1024 * sizeof(double)));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_pinned), 16 * sizeof(int), hipHostMallocDefault));
    KMCF_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_scal), sizeof(kmcf_scalars), hipHostMallocDefault));
    // the host polls h_pinned[2] for the number of a chunk check (cg_chunk_check): a recycled pinned block must not
    // already hold the first number that will be waited for
    memset(c->h_pinned, 0, 16 * sizeof(int));
    memset(c->h_scal, 0, sizeof(kmcf_scalars));
    c->mark_seq = 0;
    c->connected = (nranks == 1);
    *out = c;
    return KMCF_OK;
}

extern "C" int kmcf_comm_unique_id(void *h_id128)
{
    KMCF_CHECK(h_id128, KMCF_ERR_ARG, "kmcf_comm_unique_id: null buffer");
    KMCF_TRY(load_rccl());
    // two RCCL communicators: [0] halo send/recv on the comm stream, [1] reductions and gathers on
    // the compute stream -- one communicator must not be driven from two streams concurrently
    static_assert(2 * sizeof(ncclUniqueId) == KMCF_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id[2];
    KMCF_NCCL(g_rccl.GetUniqueId(&id[0]));
    KMCF_NCCL(g_rccl.GetUniqueId(&id[1]));
    memcpy(h_id128, id, sizeof(id));
    return KMCF_OK;
}

// IPC handles of the windows all-gathered through the RCCL communicator, peers opened, then a self-test
static int p2p_bootstrap_over_rccl(kmcf_comm *c)
{
    const int P = c->nranks;
    std::vector<char> mine(KMCF_P2P_HANDLE_BYTES), all((size_t)P * KMCF_P2P_HANDLE_BYTES);
    KMCF_TRY(kmcf_comm_p2p_export(c, mine.data()));
    int *d_h = nullptr;
    const int wi = KMCF_P2P_HANDLE_BYTES / 4;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_h), all.size()));
    KMCF_HIP(hipMemcpy(reinterpret_cast<char *>(d_h) + (size_t)c->rank * KMCF_P2P_HANDLE_BYTES, mine.data(), mine.size(), hipMemcpyHostToDevice));
    std::vector<int> cnt(P, wi), dsp(P);
    for (int q = 0; q < P; ++q) dsp[q] = q * wi;
    int rc = kmcf_comm_allgatherv_int(c, d_h, cnt.data(), dsp.data());        // over RCCL: p2p is not active yet
    if (rc == KMCF_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = KMCF_ERR_HIP;
    if (rc == KMCF_OK && hipMemcpy(all.data(), d_h, all.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = KMCF_ERR_HIP;
    hipFree(d_h);
    if (rc != KMCF_OK) return rc;
    KMCF_TRY(kmcf_comm_p2p_import(c, all.data()));
    // self-test: sums of (rank + 1) * i over a few rounds, both parities
    double *d_t = nullptr;
    KMCF_HIP(hipMalloc(reinterpret_cast<void **>(&d_t), 4 * sizeof(double)));
    for (int round = 1; round <= 4 && rc == KMCF_OK; ++round) {
        double h[3] = {(c->rank + 1.0) * round, 1.0, -0.5 * c->rank}, want[3] = {0.5 * P * (P + 1.0) * round, (double)P, -0.25 * P * (P - 1.0)};
        if (hipMemcpy(d_t, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { rc = KMCF_ERR_HIP; break; }
        rc = kmcf_p2p_allreduce(c, d_t, 3);
        if (rc == KMCF_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = KMCF_ERR_HIP;
        if (rc == KMCF_OK) rc = kmcf_p2p_check(c);
        if (rc == KMCF_OK && hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = KMCF_ERR_HIP;
        if (rc == KMCF_OK && (h[0] != want[0] || h[1] != want[1] || h[2] != want[2])) {
            kmcf_set_error("p2p self-test: all-reduce gave %g %g %g, expected %g %g %g", h[0], h[1], h[2], want[0], want[1], want[2]);
            rc = KMCF_ERR_COMM;
        }
    }
    hipFree(d_t);
    return rc;
}

extern "C" int kmcf_comm_connect(kmcf_comm *c, const void *h_id128)
{
    KMCF_CHECK(c, KMCF_ERR_ARG, "kmcf_comm_connect: null comm");
    KMCF_CHECK(c->device >= 0, KMCF_ERR_STATE, "kmcf_comm_connect: host-only communicator");
    if (c->group) return KMCF_OK;   // loopback groups are connected at creation
    // KMCF_FORCE_COMM=1 (test aid): a 1-rank group also creates its RCCL communicators and runs
    // every collective of the multi-rank code path (all-reduce of the dots, gathers)
    const bool force = getenv("KMCF_FORCE_COMM") != nullptr;
    if (c->nranks == 1 && !force) { c->connected = true; return KMCF_OK; }
    if (c->nranks > 1 && !h_id128 && !force) {
        // no RCCL id: the group will run on the peer-to-peer transport alone; connected once
        // kmcf_comm_p2p_export / kmcf_comm_p2p_import have mapped the peers' windows
        return KMCF_OK;
    }
    KMCF_TRY(load_rccl());
    KMCF_HIP(hipSetDevice(c->device));
    ncclUniqueId id[2];
    if (c->nranks == 1) {
        KMCF_NCCL(g_rccl.GetUniqueId(&id[0]));
        KMCF_NCCL(g_rccl.GetUniqueId(&id[1]));
    } else {
        KMCF_CHECK(h_id128, KMCF_ERR_ARG, "kmcf_comm_connect: null id");
        memcpy(id, h_id128, sizeof(id));
    }
    ncclComm_t comm;
    KMCF_NCCL(g_rccl.CommInitRank(&comm, c->nranks, id[0], c->rank));
    c->nccl = comm;
    KMCF_NCCL(g_rccl.CommInitRank(&comm, c->nranks, id[1], c->rank));
    c->nccl_red = comm;
    c->force_collectives = force;
    c->connected = true;
    // KMCF_TRANSPORT=p2p|auto: map the peers' windows (IPC handles all-gathered over RCCL) and run the exchanges of
    // the CG loop over them; RCCL stays connected as the fallback.  "auto" keeps RCCL if the set-up or a short
    // self-test (all-reduces of known values through the windows) fails.
    const char *tr = getenv("KMCF_TRANSPORT");
    if (c->nranks > 1 && tr && (strcmp(tr, "p2p") == 0 || strcmp(tr, "auto") == 0)) {
        const bool must = strcmp(tr, "p2p") == 0;
        int rc = p2p_bootstrap_over_rccl(c);
        if (rc != KMCF_OK) {
            kmcf_p2p_destroy(c);
            if (must) return rc;
            fprintf(stderr, "kmcfield: p2p transport unavailable (%s); using RCCL\n", kmcf_last_error());
        }
    }
    return KMCF_OK;
}

extern "C" int kmcf_comm_destroy(kmcf_comm *c)
{
    if (!c) return KMCF_OK;
    if (c->device < 0) { delete c; return KMCF_OK; }
    if (c->group) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->group->mu); last = (--c->group->refs == 0); }
        if (last) delete c->group;
        c->group = nullptr;
    }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    kmcf_event_cache_free(c);
    kmcf_p2p_destroy(c);
    if (c->nccl) g_rccl.CommDestroy(static_cast<ncclComm_t>(c->nccl));
    if (c->nccl_red) g_rccl.CommDestroy(static_cast<ncclComm_t>(c->nccl_red));
    if (c->ev_packed) hipEventDestroy(c->ev_packed);
    if (c->ev_halo) hipEventDestroy(c->ev_halo);
    if (c->ev_subpack) hipEventDestroy(c->ev_subpack);
    if (c->ev_sub) hipEventDestroy(c->ev_sub);
    if (c->ev_t0) hipEventDestroy(c->ev_t0);
    if (c->ev_t1) hipEventDestroy(c->ev_t1);
    if (c->ev_a0) hipEventDestroy(c->ev_a0);
    if (c->ev_a1) hipEventDestroy(c->ev_a1);
    if (c->ev_call0) hipEventDestroy(c->ev_call0);
    if (c->ev_call1) hipEventDestroy(c->ev_call1);
    if (c->ev_entry) hipEventDestroy(c->ev_entry);
    if (c->d_scratch) hipFree(c->d_scratch);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->comm_stream) hipStreamDestroy(c->comm_stream);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_scal) hipHostFree(c->h_scal);
    delete c;
    return KMCF_OK;
}

extern "C" int kmcf_comm_sync(kmcf_comm *c)
{
    KMCF_CHECK(c, KMCF_ERR_ARG, "kmcf_comm_sync: null comm");
    if (c->device < 0) return KMCF_OK;
    KMCF_HIP(hipStreamSynchronize(c->comm_stream));
    KMCF_HIP(hipStreamSynchronize(c->stream));
    return KMCF_OK;
}

extern "C" void *kmcf_comm_stream(kmcf_comm *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

extern "C" int kmcf_comm_set_caller_stream(kmcf_comm *c, void *stream)
{
    KMCF_CHECK(c, KMCF_ERR_ARG, "kmcf_comm_set_caller_stream: null comm");
    c->caller_stream = static_cast<hipStream_t>(stream);
    return KMCF_OK;
}

// The library's streams are non-blocking (they must overlap with each other and must not serialise against
// unrelated null-stream traffic), so nothing orders them after the caller's queued work implicitly.  Every
// entry point therefore records an event on the caller's stream (the legacy null stream unless
// kmcf_comm_set_caller_stream named another) and makes the compute stream wait for it.  The way back needs
// no event: every entry point synchronises its streams before it returns.
int kmcf_enter(kmcf_comm *c)
{
    KMCF_CHECK(c && c->device >= 0, KMCF_ERR_STATE, "host-only communicator");
    KMCF_HIP(hipSetDevice(c->device));
    // nothing queued on the caller's stream (the usual case: the caller has just synchronised): nothing to order
    // against -- an event on the legacy null stream is not cheap
    if (!getenv("KMCF_ENTER_ALWAYS") && hipStreamQuery(c->caller_stream) == hipSuccess) return KMCF_OK;
    (void)hipGetLastError();                          // (hipErrorNotReady is the expected answer otherwise)
    // (Seen on ROCm 7.2: after an upload of some MB from PAGEABLE host memory the null stream keeps answering "not
    // ready" -- also after hipDeviceSynchronize, also when asked again -- until an event has been recorded on it: the
    // next call pays this branch once, 40-100 us until its first kernel starts.  Uploads from pinned memory do not.)
    if (getenv("KMCF_TRACE")) fprintf(stderr, "kmcf_enter: the caller's stream is busy (ordering behind an event on it)\n");
    KMCF_HIP(hipEventRecord(c->ev_entry, c->caller_stream));
    KMCF_HIP(hipStreamWaitEvent(c->stream, c->ev_entry, 0));
    return KMCF_OK;
}

// ------------------------------------------------------------------ loopback transport (tests)
namespace {

// Bounded: a rank that never arrives (its thread failed in an earlier call and left) must not hang the others.
// After one expiry the group is broken and every later barrier fails at once.
bool group_barrier_wait(kmcf_group *g)
{
    static const int timeout_s = getenv("KMCF_LOOPBACK_TIMEOUT_S") ? atoi(getenv("KMCF_LOOPBACK_TIMEOUT_S")) : 120;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const long gen = g->generation;
    if (++g->arrived == g->nranks) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return g->generation != gen || g->broken; }) || g->broken) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

#define group_barrier(g)                                                                                          \
    do {                                                                                                          \
        if (!group_barrier_wait(g)) {                                                                             \
            kmcf_set_error("loopback group: a rank did not arrive at a collective (it failed earlier, or timed out)"); \
            return KMCF_ERR_COMM;                                                                                 \
        }                                                                                                         \
    } while (0)

int loopback_allreduce(kmcf_comm *c, double *d_buf, int count)
{
    kmcf_group *g = c->group;
    KMCF_CHECK(count <= 8, KMCF_ERR_ARG, "loopback all-reduce: count %d > 8", count);
    KMCF_HIP(hipStreamSynchronize(c->stream));
    g->slot[c->rank] = d_buf;
    group_barrier(g);
    double acc[8] = {0}, tmp[8];
    for (int q = 0; q < g->nranks; ++q) {           // rank order, like one partial per rank in MPI_Allreduce
        KMCF_HIP(hipMemcpy(tmp, g->slot[q], (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i) acc[i] += tmp[i];
    }
    group_barrier(g);                                // everyone has read every slot
    KMCF_HIP(hipMemcpy(d_buf, acc, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    return KMCF_OK;
}

int loopback_halo(kmcf_matrix *m)
{
    kmcf_comm *c = m->comm;
    kmcf_group *g = c->group;
    KMCF_HIP(hipStreamSynchronize(c->stream));       // pack kernel done
    KMCF_HIP(hipStreamSynchronize(c->comm_stream));
    g->mat[c->rank] = m;
    group_barrier(g);
    int rc = KMCF_OK;
    for (int k = 1; k < m->number_of_neighbours && rc == KMCF_OK; ++k) {
        const int peer = m->neighbours[k];
        kmcf_matrix *pm = g->mat[peer];
        int kk = -1;
        for (int t = 1; t < pm->number_of_neighbours; ++t)
            if (pm->neighbours[t] == c->rank) kk = t;
        const size_t nr = m->cols_per_neighbour[k].size();
        if (kk < 0 || pm->rows_per_neighbour[kk].size() != nr) {
            kmcf_set_error("loopback halo: rank %d expects %zu values from rank %d, which sends %zu (matrix not structurally symmetric?)",
                           c->rank, nr, peer, kk < 0 ? (size_t)0 : pm->rows_per_neighbour[kk].size());
            rc = KMCF_ERR_COMM;
            break;
        }
        // device-to-device copies need not be host-synchronous: enqueue on this rank's stream and wait below
        if (nr && hipMemcpyAsync(m->d_p + m->n_loc + m->halo_offset[k], pm->d_send_buf + pm->send_offset[kk],
                                 nr * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
            kmcf_set_error("loopback halo: hipMemcpy failed");
            rc = KMCF_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == KMCF_OK) {
        kmcf_set_error("loopback halo: stream sync failed");
        rc = KMCF_ERR_HIP;
    }
    group_barrier(g);                                // peers may repack now
    return rc;
}

int loopback_allgatherv(kmcf_comm *c, void *d_buf, const int *counts, const int *displs, size_t es)
{
    kmcf_group *g = c->group;
    KMCF_HIP(hipStreamSynchronize(c->stream));
    g->slot[c->rank] = d_buf;
    group_barrier(g);
    for (int q = 0; q < g->nranks; ++q) {
        if (q == c->rank || counts[q] == 0) continue;
        KMCF_HIP(hipMemcpyAsync(static_cast<char *>(d_buf) + (size_t)displs[q] * es,
                                static_cast<char *>(g->slot[q]) + (size_t)displs[q] * es, (size_t)counts[q] * es,
                                hipMemcpyDeviceToDevice, c->stream));
    }
    KMCF_HIP(hipStreamSynchronize(c->stream));       // copies landed before any peer overwrites its slot
    group_barrier(g);
    return KMCF_OK;
}

}  // namespace

int kmcf_group_rendezvous(kmcf_comm *c)
{
    if (!c->group || !c->p2p_active || c->nranks <= 1 || c->in_solve) return KMCF_OK;
    group_barrier(c->group);
    return KMCF_OK;
}

// All P ranks of an in-process loopback group at once (out: array of nranks communicators); each is
// then driven by its own host thread.
extern "C" int kmcf_comm_create_loopback(kmcf_comm **out, int device, int nranks)
{
    KMCF_CHECK(out && nranks >= 1, KMCF_ERR_ARG, "kmcf_comm_create_loopback: bad argument");
    kmcf_group *g = new kmcf_group();
    g->nranks = nranks;
    g->slot.assign(nranks, nullptr);
    g->mat.assign(nranks, nullptr);
    g->refs = nranks;
    for (int r = 0; r < nranks; ++r) {
        int rc = kmcf_comm_create(&out[r], device, nranks, r);
        if (rc != KMCF_OK) return rc;
        out[r]->group = g;
        out[r]->connected = true;
    }
    // KMCF_TRANSPORT=p2p: the members drive the device-side peer-to-peer protocol (kmcf_p2p.hip) on each other's
    // windows -- same kernels, flags and sequence numbers as between processes, plain pointers instead of IPC
    // handles.  The ranks' kernels wait for each other ON the GPU, so their streams must map to different hardware
    // queues (GPU_MAX_HW_QUEUES >= 2 * nranks, set before HIP initialises; tests/conftest.py does).
    const char *tr = getenv("KMCF_TRANSPORT");
    if (tr && strcmp(tr, "p2p") == 0 && nranks > 1) {
        std::vector<char *> bases((size_t)nranks);
        for (int r = 0; r < nranks; ++r) {
            KMCF_TRY(kmcf_p2p_create(out[r]));
            bases[r] = kmcf_p2p_window(out[r]);
        }
        for (int r = 0; r < nranks; ++r) KMCF_TRY(kmcf_p2p_set_peers_direct(out[r], bases.data()));
    }
    return KMCF_OK;
}

extern "C" int kmcf_comm_select_transport(kmcf_comm *c, int use_p2p)
{
    KMCF_CHECK(c, KMCF_ERR_ARG, "kmcf_comm_select_transport: null comm");
    if (c->nranks == 1) return KMCF_OK;
    if (use_p2p) {
        KMCF_CHECK(c->p2p != nullptr, KMCF_ERR_STATE, "kmcf_comm_select_transport: the peer-to-peer transport was not set up");
        c->p2p_active = true;
        if (kmcf_p2p_check(c) != KMCF_OK) {          // it has failed before: stays off
            c->p2p_active = false;
            return KMCF_ERR_COMM;
        }
    } else {
        KMCF_CHECK(c->group || c->nccl_red, KMCF_ERR_STATE, "kmcf_comm_select_transport: no other transport is connected");
        c->p2p_active = false;
    }
    return KMCF_OK;
}

extern "C" const char *kmcf_comm_transport(const kmcf_comm *c)
{
    if (!c || c->nranks == 1) return "single";
    if (c->p2p_active) return c->group ? "p2p (in-process group)" : (c->nccl ? "p2p (bootstrapped over rccl)" : "p2p");
    return c->group ? "loopback" : "rccl";
}

extern "C" int kmcf_comm_rccl_ranks(const kmcf_comm *c)
{
    if (!c || !c->nccl_red || !g_rccl.CommCount) return 0;
    int n = 0;
    return g_rccl.CommCount(static_cast<ncclComm_t>(c->nccl_red), &n) == ncclSuccess ? n : -1;
}

// Sum `count` doubles in place over all ranks, on the compute stream, device resident.
int kmcf_comm_allreduce_sum(kmcf_comm *c, double *d_buf, int count)
{
    if (c->p2p_active && c->nranks > 1) {
        KMCF_TRY(kmcf_group_rendezvous(c));
        return kmcf_p2p_allreduce(c, d_buf, count);
    }
    if (c->group) return c->group->nranks > 1 ? loopback_allreduce(c, d_buf, count) : KMCF_OK;
    if (c->nranks == 1 && !c->force_collectives) return KMCF_OK;
    KMCF_CHECK(c->nccl_red, KMCF_ERR_COMM, "communicator not connected (call kmcf_comm_connect)");
    KMCF_NCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum,
                               static_cast<ncclComm_t>(c->nccl_red), c->stream));
    return KMCF_OK;
}

// Halo exchange of p on the comm stream: for every neighbour k>=1 send the packed
// rows_per_neighbour[k] entries and receive nnz_cols_per_neighbour[k] doubles
// directly into the halo slots of d_p (no unpack kernel: column ids were remapped
// to compact halo slots at matrix build).
int kmcf_comm_send_recv_halo(kmcf_matrix *m)
{
    kmcf_comm *c = m->comm;
    if (c->p2p_active && c->nranks > 1) return kmcf_p2p_halo_exchange(m);
    if (c->group) return c->group->nranks > 1 ? loopback_halo(m) : KMCF_OK;   // every rank takes part, neighbours or not
    if (m->number_of_neighbours <= 1) return KMCF_OK;
    KMCF_CHECK(c->nccl, KMCF_ERR_COMM, "communicator not connected (call kmcf_comm_connect)");
    ncclComm_t comm = static_cast<ncclComm_t>(c->nccl);
    KMCF_NCCL(g_rccl.GroupStart());
    for (int k = 1; k < m->number_of_neighbours; ++k) {
        int peer = m->neighbours[k];
        size_t ns = m->rows_per_neighbour[k].size();
        size_t nr = m->cols_per_neighbour[k].size();
        if (ns) KMCF_NCCL(g_rccl.Send(m->d_send_buf + m->send_offset[k], ns, ncclDouble, peer, comm, c->comm_stream));
        if (nr) KMCF_NCCL(g_rccl.Recv(m->d_p + m->n_loc + m->halo_offset[k], nr, ncclDouble, peer, comm, c->comm_stream));
    }
    KMCF_NCCL(g_rccl.GroupEnd());
    return KMCF_OK;
}

// In-place all-gather with uneven counts: rank q owns d_buf[displs[q] .. +counts[q]).
template <typename T>
static int allgatherv_impl(kmcf_comm *c, T *d_buf, const int *counts, const int *displs, ncclDataType_t dt)
{
    // gathers that fit the window's staging area go peer to peer; the rare big ones (the neighbour lists of the
    // replicated event step, once per run) take the transport underneath
    if (c->p2p_active && c->nranks > 1 &&
        kmcf_p2p_fits(c, ((size_t)displs[c->nranks - 1] + counts[c->nranks - 1]) * sizeof(T))) {
        KMCF_TRY(kmcf_group_rendezvous(c));
        return kmcf_p2p_allgatherv(c, d_buf, counts, displs, sizeof(T));
    }
    if (c->p2p_active && c->nranks > 1 && !c->group && !c->nccl_red) {
        kmcf_set_error("all-gather of %zu bytes exceeds the p2p staging area and no other transport is connected (KMCF_P2P_WINDOW_MB)",
                       ((size_t)displs[c->nranks - 1] + counts[c->nranks - 1]) * sizeof(T));
        return KMCF_ERR_NOMEM;
    }
    if (c->group) return c->group->nranks > 1 ? loopback_allgatherv(c, d_buf, counts, displs, sizeof(T)) : KMCF_OK;
    if (c->nranks == 1 && !c->force_collectives) return KMCF_OK;
    KMCF_CHECK(c->nccl_red, KMCF_ERR_COMM, "communicator not connected (call kmcf_comm_connect)");
    ncclComm_t comm = static_cast<ncclComm_t>(c->nccl_red);
    KMCF_NCCL(g_rccl.GroupStart());
    for (int q = 0; q < c->nranks; ++q)
        if (counts[q] > 0)
            KMCF_NCCL(g_rccl.Broadcast(d_buf + displs[q], d_buf + displs[q], (size_t)counts[q], dt, q, comm, c->stream));
    KMCF_NCCL(g_rccl.GroupEnd());
    return KMCF_OK;
}

int kmcf_comm_allgatherv_double(kmcf_comm *c, double *d_buf, const int *counts, const int *displs)
{
    return allgatherv_impl<double>(c, d_buf, counts, displs, ncclDouble);
}

int kmcf_comm_allgatherv_double_comm_stream(kmcf_comm *c, double *d_buf, const int *counts, const int *displs)
{
    if (c->p2p_active && c->nranks > 1 && kmcf_p2p_fits(c, ((size_t)displs[c->nranks - 1] + counts[c->nranks - 1]) * sizeof(double)))
        return kmcf_p2p_allgatherv(c, d_buf, counts, displs, sizeof(double), c->comm_stream);
    if (c->group || !c->nccl || c->nranks == 1) return KMCF_ERR_STATE;           // host-synchronous / nothing to overlap
    ncclComm_t comm = static_cast<ncclComm_t>(c->nccl);                           // the comm stream's communicator (halos)
    KMCF_NCCL(g_rccl.GroupStart());
    for (int q = 0; q < c->nranks; ++q)
        if (counts[q] > 0)
            KMCF_NCCL(g_rccl.Broadcast(d_buf + displs[q], d_buf + displs[q], (size_t)counts[q], ncclDouble, q, comm, c->comm_stream));
    KMCF_NCCL(g_rccl.GroupEnd());
    return KMCF_OK;
}

int kmcf_comm_allgatherv_int(kmcf_comm *c, int *d_buf, const int *counts, const int *displs)
{
    return allgatherv_impl<int>(c, d_buf, counts, displs, ncclInt32);
}
