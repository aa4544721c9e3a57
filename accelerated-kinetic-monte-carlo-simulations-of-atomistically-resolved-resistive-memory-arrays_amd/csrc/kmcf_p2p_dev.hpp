// Peer-to-peer transport: window layout, per-communicator / per-matrix state and the device-side primitives
// (release store, bounded wait) shared by kmcf_p2p.hip and the kernels that fold an exchange into their own work
// (kmcf_cg.hip: the single-reduction update kernel; kmcf_spmv.hip: the boundary-row pass).
#pragma once
#include "kmcf_internal.hpp"

constexpr int P2P_MAXR = 64;
// Every flag and every all-reduce slot is a 128-byte line of its own (P2P_FS 8-byte words): no line has two writers,
// and the window's owner never writes a line it polls.  (With several flags of different writers in one line, the
// owner's own store could leave the line in its L2 -- each XCD has its own -- and its polls of the NEIGHBOURING words,
// which a peer had meanwhile written to memory, kept hitting that copy: seen as intermittent time-outs of in-process
// groups.)
constexpr int P2P_FS = 16;
constexpr size_t P2P_OFF_RED_SLOT = 0;                                       // double [2][MAXR][P2P_FS]  (4 used)
constexpr size_t P2P_OFF_RED_FLAG = P2P_OFF_RED_SLOT + 2 * P2P_MAXR * 128;   // u64    [2][MAXR][P2P_FS]
constexpr size_t P2P_OFF_G_FLAG = P2P_OFF_RED_FLAG + 2 * P2P_MAXR * 128;     // u64    [MAXR][P2P_FS]   published gather sequence of rank q (written by q)
constexpr size_t P2P_OFF_G_ACK = P2P_OFF_G_FLAG + P2P_MAXR * 128;            // u64    [MAXR][P2P_FS]   gather sequence rank q has consumed from me
constexpr size_t P2P_OFF_BUMP = P2P_OFF_G_ACK + P2P_MAXR * 128;

typedef unsigned long long u64;

#ifndef KMCF_P2P_SLEEP
#define KMCF_P2P_SLEEP 8             // s_sleep argument between two polls of a flag (units of 64 clocks)
#endif

struct kmcf_p2p_dev;

struct kmcf_p2p {
    int nranks = 1, rank = 0;
    char *win = nullptr;
    size_t win_bytes = 0;
    bool fine_grained = false;
    std::vector<char *> peer;                // base of every rank's window in this address space
    std::vector<bool> ipc_opened;
    char **d_peer = nullptr;
    size_t stage_off = 0, stage_half = 0;    // two halves of gather staging
    size_t bump = 0;
    u64 seq_red = 0, seq_gather = 0;
    long long timeout_ticks = 0;             // wall_clock64 ticks (hipDeviceAttributeWallClockRate)
    int *d_err = nullptr;                    // device copy of the error word (polled by waiting kernels)
    int *h_err = nullptr;                    // pinned host copy (read by the host after a synchronisation)
    unsigned int *d_ctr = nullptr;           // last-block counters: [0] gather stage, [1] gather pull
};

// per-matrix state of the halo protocol
struct kmcf_p2p_halo {
    double **d_put_ptr = nullptr;            // remote address of every packed entry (buffer 0 of the receiver's landing zone)
    long long *d_put_stride = nullptr;       // per packed entry: doubles from buffer 0 to buffer 1 there (the receiver's halo size)
    u64 **d_put_flag = nullptr;              // remote flag per neighbour (k >= 1)
    u64 **d_ack_ptr = nullptr;               // per neighbour: where I acknowledge ITS puts (in its window)
    size_t land_off = 0, flag_off = 0;       // own landing zone (2 x n_halo doubles) and flags (nnb - 1) in my window
    size_t ack_off = 0;                      // acknowledgements of MY puts, written by the neighbours (nnb - 1) in my window
    u64 seq = 0;                             // latest sequence whose consumption (an SpMV) was enqueued
    u64 seq_put = 0;                         // latest sequence put (== seq, or seq + 1 while a kernel has put ahead for the next SpMV)
    unsigned int *d_ctr = nullptr;           // [0] put: last block raises the flags; [1] wait: last block acknowledges
    // "direct" protocol (kmcf_p2p_dev below): the rows that are sent, looked up by internal row
    int *d_put_row = nullptr;                // per internal row: -1, or its index among the sent rows
    int *d_putr_ptr = nullptr;               // per sent row: its entries in d_putr_addr / d_putr_stride (one per receiving neighbour)
    double **d_putr_addr = nullptr;
    long long *d_putr_stride = nullptr;
    // register-resident solve of a group (kmcf_cgr.hip): the halo travels as 16-byte {value, sequence} granules, written
    // by the lane that owns the row straight into the receiver's granule zone (two parities of n_halo granules behind
    // its landing buffers); every rank's sums go to a line per rank in every peer's reduction zone
    size_t ll_off = 0, red_off = 0;          // own granule zone / reduction zone (2 x P2P_MAXR lines) in my window
    u64 **d_putr_ll = nullptr;               // per entry of d_putr_addr: the remote granule (parity 0)
    long long *d_putr_ll_stride = nullptr;   // ... 8-byte words from parity 0 to parity 1 there (2 x the receiver's halo size)
    u64 **d_red_peer = nullptr;              // per rank: its reduction zone
};

// What a compute kernel needs to take part in the exchanges itself (by value in the kernel arguments).  "Direct"
// protocol of a group's single-reduction CG: no pack / put / wait / copy kernels and no second stream -- the kernel
// that produces the next SpMV input puts the rows its neighbours need, the boundary-row pass waits for its flags
// and reads the landing zone in place, the next kernel acknowledges.
struct kmcf_p2p_dev {
    char *const *peer;                       // window base of every rank
    int P, rank;
    long long timeout;
    int *d_err, *h_err;
    int n_nb;                                // neighbours (k >= 1)
    const u64 *flags;                        // my window: flag of neighbour k's puts at [k * P2P_FS] (acks alike)
    const u64 *acks;                         // my window: neighbour k's acknowledgement of MY puts
    u64 *const *ack_ptr;                     // neighbour k's window: my acknowledgement of ITS puts
    u64 *const *put_flag;                    // neighbour k's window: the flag of my puts
    const double *landing;                   // my window: two buffers of n_halo doubles
    int n_halo;
    const int *put_row, *putr_ptr;
    double *const *putr_addr;
    const long long *putr_stride;
    unsigned int *ctr;                       // last-block counter of the putting kernel
};

__device__ __forceinline__ double load_system(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void store_system(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void store_release_system(u64 *p, u64 v)
{
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the fence's write-back must have drained before the flag goes out
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait for *p >= v.  err codes: 1 all-reduce, 2 halo, 3 gather flag, 4 gather ack
__device__ __forceinline__ bool wait_ge(const u64 *p, u64 v, long long timeout, int *d_err, int *h_err, int code)
{
    const long long t0 = wall_clock64();
#ifdef KMCF_P2P_POLL_RMW
    while (__hip_atomic_fetch_add(const_cast<u64 *>(p), 0ull, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
#else
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
#endif
        if (__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;   // already failed
        if (wall_clock64() - t0 > timeout) {
#ifdef KMCF_P2P_DEBUG
            printf("p2p wait timeout: code %d want %llu have %llu at %p (block %d thread %d)\n", code, v,
                   (unsigned long long)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM), (const void *)p, (int)blockIdx.x, (int)threadIdx.x);
#endif
            __hip_atomic_store(d_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(h_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(KMCF_P2P_SLEEP);
    }
    return true;
}


// host side (kmcf_p2p.hip)
kmcf_p2p_dev kmcf_p2p_dev_of(const kmcf_matrix *m);
unsigned long long kmcf_p2p_next_red_seq(kmcf_comm *c);           // the all-reduce sequence counter of the communicator
unsigned long long kmcf_p2p_red_seq(const kmcf_comm *c);
void kmcf_p2p_set_red_seq(kmcf_comm *c, unsigned long long v);
unsigned long long *kmcf_p2p_halo_seq(kmcf_matrix *m, int which);  // 0: consumed, 1: put
