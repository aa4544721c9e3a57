// lab: what would ONE persistent kernel per CG solve cost on a rank's eighth of the 40 nm matrix?  Two phases per iteration
// (a 19 MB stream + a gathered vector, then an 18 MB vector update), separated by (a) kernel boundaries, as the library
// does today, (b) hand-rolled grid barriers inside one launch, the cross-block vector read and written around the L2
// (system-coherent loads / stores) so that the barrier needs no cache flush.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int BLOCK = 256;
// phase A ("SpMV"): stream `nA` doubles of matrix, gather p at pseudo-random nearby positions, write w
// phase B ("update"): read w, r, s, x; write r, s, x, p
template <bool SC1>
__device__ __forceinline__ double ldp(const double *p) { return SC1 ? __builtin_nontemporal_load(p) : *p; }

__device__ void phase_a(const double *__restrict__ A, size_t nA, const double *p, double *__restrict__ w, int n, int nblk, int blk, bool sc1)
{
    const size_t per = (nA + nblk - 1) / nblk;
    const size_t b0 = (size_t)blk * per, b1 = min(nA, b0 + per);
    double acc = 0.0;
    for (size_t i = b0 + threadIdx.x; i < b1; i += BLOCK) {
        const double a = A[i];
        const int j = (int)((i * 2654435761ull) % (size_t)n);
        const double pj = sc1 ? __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[j];
        acc += a * pj;
    }
    const int rows_per = (n + nblk - 1) / nblk;
    for (int r = blk * rows_per + threadIdx.x; r < min(n, (blk + 1) * rows_per); r += BLOCK) w[r] = acc;
}
__device__ void phase_b(const double *__restrict__ w, double *__restrict__ r, double *__restrict__ s, double *__restrict__ x, double *p, int n,
                        int nblk, int blk, bool sc1)
{
    const int rows_per = (n + nblk - 1) / nblk;
    for (int i = blk * rows_per + threadIdx.x; i < min(n, (blk + 1) * rows_per); i += BLOCK) {
        const double rv = r[i] - 0.5 * w[i], sv = s[i] * 0.9 + rv, xv = x[i] + 0.1 * sv;
        r[i] = rv; s[i] = sv; x[i] = xv;
        const double pv = rv + 0.3 * sv;
        if (sc1) __hip_atomic_store(p + i, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else p[i] = pv;
    }
}
__global__ __launch_bounds__(BLOCK) void ka(const double *A, size_t nA, const double *p, double *w, int n) { phase_a(A, nA, p, w, n, gridDim.x, blockIdx.x, false); }
__global__ __launch_bounds__(BLOCK) void kb(const double *w, double *r, double *s, double *x, double *p, int n) { phase_b(w, r, s, x, p, n, gridDim.x, blockIdx.x, false); }

__device__ __forceinline__ void grid_barrier(unsigned int *ctr, unsigned int target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// MODE 0: barriers with agent-scope release / acquire and plain vector accesses (the caches are flushed by the fences);
// MODE 1: the shared vector p through agent-scope atomics (around the non-coherent part of the caches), barrier relaxed
template <int MODE>
__global__ __launch_bounds__(BLOCK) void persistent(const double *A, size_t nA, double *p, double *w, double *r, double *s, double *x, int n, int iters,
                                                   unsigned int *ctr)
{
    unsigned int target = 0;
    for (int it = 0; it < iters; ++it) {
        phase_a(A, nA, p, w, n, gridDim.x, blockIdx.x, MODE == 1);
        target += gridDim.x;
        if (MODE == 1) {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        } else grid_barrier(ctr, target);
        phase_b(w, r, s, x, p, n, gridDim.x, blockIdx.x, MODE == 1);
        target += gridDim.x;
        if (MODE == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // (p's stores have left the wave before the others gather them)
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        } else grid_barrier(ctr, target);
    }
}
__global__ void barrier_only(int iters, unsigned int *ctr)
{
    unsigned int target = 0;
    for (int it = 0; it < iters; ++it) { target += gridDim.x; grid_barrier(ctr, target); }
}
int main()
{
    const int n = 225447; const size_t nA = (size_t)19e6 / 8;       // a rank's eighth: 19 MB of matrix stream
    double *A, *p, *w, *r, *s, *x; unsigned int *ctr;
    (void)hipMalloc(&A, nA * 8); (void)hipMalloc(&p, n * 8); (void)hipMalloc(&w, n * 8); (void)hipMalloc(&r, n * 8); (void)hipMalloc(&s, n * 8); (void)hipMalloc(&x, n * 8);
    (void)hipMalloc(&ctr, 4);
    (void)hipMemset(A, 0, nA * 8); (void)hipMemset(p, 0, n * 8); (void)hipMemset(w, 0, n * 8); (void)hipMemset(r, 0, n * 8); (void)hipMemset(s, 0, n * 8); (void)hipMemset(x, 0, n * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200;
    auto time = [&](auto f, const char *name) {
        f(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.2f us per iteration\n", name, ms * 1e3 / iters);
    };
    for (int grid : {512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, 96, "two kernels per iteration, grid %d", grid);
        time([&] { for (int it = 0; it < iters; ++it) { ka<<<grid, BLOCK>>>(A, nA, p, w, n); kb<<<grid, BLOCK>>>(w, r, s, x, p, n); } }, nm);
    }
    int dev = 0, ncu = 0, occ = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    for (int per_cu : {1, 2, 4}) {
        const int grid = ncu * per_cu;
        char nm[96];
        (void)hipMemset(ctr, 0, 4);
        snprintf(nm, 96, "grid barriers alone (2 per iteration), %d blocks", grid);
        time([&] { (void)hipMemsetAsync(ctr, 0, 4); barrier_only<<<grid, BLOCK>>>(2 * iters, ctr); }, nm);
        snprintf(nm, 96, "persistent, fenced barriers, %d blocks", grid);
        time([&] { (void)hipMemsetAsync(ctr, 0, 4); persistent<0><<<grid, BLOCK>>>(A, nA, p, w, r, s, x, n, iters, ctr); }, nm);
        snprintf(nm, 96, "persistent, p through agent-scope atomics, %d blocks", grid);
        time([&] { (void)hipMemsetAsync(ctr, 0, 4); persistent<1><<<grid, BLOCK>>>(A, nA, p, w, r, s, x, n, iters, ctr); }, nm);
    }
    (void)occ;
    return 0;
}
