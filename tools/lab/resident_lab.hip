// lab: what does ONE iteration of a REGISTER-RESIDENT CG cost on a rank's eighth of the 40 nm matrix?
// Model of csrc/kmcf_cgr.hip before it was written.  One block per 256-row tile, all blocks co-resident for the whole
// solve; a lane keeps its row's x, r, p, s, 1/diag and its slice of the entry stream in registers.  Per iteration only
// two things cross block boundaries, both around the (per-XCD, mutually incoherent) L2s with agent-scope accesses:
//   (1) z of the neighbouring tiles (window gather, after their per-tile sequence flags),
//   (2) the two dot products: every block publishes (value, sequence) in a line of its own, one leader per group of G1
//       blocks adds its group in a fixed order and publishes the group sum, every block adds the group sums in a fixed
//       order -- no read-modify-write on a shared word anywhere (256 same-address atomics cost ~10 us, gridbar_lab.hip).
// Output: us per iteration for a few variants.   hipcc -O3 --offload-arch=gfx950 resident_lab.hip -o resident_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
constexpr int BLOCK = 256;
constexpr int LINE = 16;                 // 8-byte words per 128-byte line
constexpr int NBR = 24;                  // neighbouring tiles a tile's window draws from
constexpr int WQ = 3;                    // outside window columns per lane (768 / 256)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u64 ld_u64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_f64(const double *p) { return __longlong_as_double((long long)ld_u64(reinterpret_cast<const u64 *>(p))); }
__device__ __forceinline__ void st_f64(double *p, double v) { st_u64(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v)); }

__device__ __forceinline__ bool wait_ge(const u64 *p, u64 v, long long t0, long long timeout, int *err)
{
    while (ld_u64(p) < v) {
        if (wall_clock64() - t0 > timeout) { *err = 1; return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// MODE bit 0: neighbour flags + window gather; bit 1: tree reduce; bit 2: z published
template <int MODE>
__global__ __launch_bounds__(BLOCK) void resident(int iters, int nblk, int g1, double *z /* [2][n] */, u64 *tflag /* [nblk][LINE] */,
                                                  double *slot /* [2][nblk][LINE] (value, seq) */, double *gslot /* [2][ngroups][LINE] */,
                                                  const int *nbr /* [nblk][NBR] */, const int *wcol /* [nblk][WQ][BLOCK] */, long long timeout, int *err,
                                                  double *out)
{
    __shared__ double xs[1024];
    __shared__ double wsum[4];
    __shared__ double bc[2];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = nblk * BLOCK, row = b * BLOCK + t;
    const int ngroups = (nblk + g1 - 1) / g1;
    double x = 0.0, r = 1.0 + 1e-3 * (row % 97), p = 0.0, s = 0.0, zv = r;
    int wc[WQ];
    for (int q = 0; q < WQ; ++q) wc[q] = wcol[((size_t)b * WQ + q) * BLOCK + t];
    const int my_nbr = t < NBR ? nbr[b * NBR + t] : 0;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        const int par = it & 1;
        // ---- (1) neighbours' z of iteration it - 1 (buffer par ^ 1 ... the one written last)
        double acc = zv;
        if (MODE & 1) {
            if (it > 1 && t < NBR) wait_ge(&tflag[(size_t)my_nbr * LINE], (u64)(it - 1), t0, timeout, err);
            __syncthreads();
            const double *zb = z + (size_t)(par ^ 1) * n;
            double g[WQ];
            for (int q = 0; q < WQ; ++q) g[q] = ld_f64(zb + wc[q]);
            xs[t] = zv;
            for (int q = 0; q < WQ; ++q) xs[BLOCK + q * BLOCK + t] = g[q];
            __syncthreads();
            for (int e = 0; e < 28; ++e) acc += 1e-3 * xs[(t * 7 + e * 37) & 1023];      // the row: 28 LDS reads
        }
        const double w = acc;
        // ---- (2) two dots, tree-reduced over the blocks
        double gamma = r * zv, delta = w * zv;
        if (MODE & 2) {
            gamma = wave_sum(gamma); delta = wave_sum(delta);
            if (lane == 0) { wsum[wv] = gamma; }
            __syncthreads();
            const double bg = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            __syncthreads();
            if (lane == 0) { wsum[wv] = delta; }
            __syncthreads();
            const double bd = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            double *sl = slot + ((size_t)par * nblk + b) * LINE;
            if (t == 0) {
                st_f64(sl, bg); st_f64(sl + 1, bd);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_u64(reinterpret_cast<u64 *>(sl + 2), (u64)it);
            }
            if (b % g1 == 0 && wv == 0) {                 // leader of a group: its blocks in order, one lane each (g1 <= 64)
                const int q = b + lane;
                double vg = 0.0, vd = 0.0;
                if (lane < g1 && q < nblk) {
                    const double *sq = slot + ((size_t)par * nblk + q) * LINE;
                    wait_ge(reinterpret_cast<const u64 *>(sq + 2), (u64)it, t0, timeout, err);
                    vg = ld_f64(sq); vd = ld_f64(sq + 1);
                }
                vg = wave_sum(vg); vd = wave_sum(vd);
                double *gs = gslot + ((size_t)par * ngroups + b / g1) * LINE;
                if (lane == 0) {
                    st_f64(gs, vg); st_f64(gs + 1, vd);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_u64(reinterpret_cast<u64 *>(gs + 2), (u64)it);
                }
            }
            if (wv == 1) {                                // every block: the group sums in order (ngroups <= 64)
                double vg = 0.0, vd = 0.0;
                if (lane < ngroups) {
                    const double *gq = gslot + ((size_t)par * ngroups + lane) * LINE;
                    wait_ge(reinterpret_cast<const u64 *>(gq + 2), (u64)it, t0, timeout, err);
                    vg = ld_f64(gq); vd = ld_f64(gq + 1);
                }
                vg = wave_sum(vg); vd = wave_sum(vd);
                if (lane == 0) { bc[0] = vg; bc[1] = vd; }
            }
            __syncthreads();
            gamma = bc[0]; delta = bc[1];
        }
        // ---- update in registers
        const double alpha = 1e-3 * gamma / (fabs(delta) + 1.0), beta = 0.5;
        p = zv + beta * p; s = w + beta * s; x += alpha * p; r -= alpha * s; zv = r * 0.999;
        if (MODE & 4) {
            st_f64(z + (size_t)par * n + row, zv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores are out before the barrier, the flag after it
            __syncthreads();
            if (t == 0) st_u64(&tflag[(size_t)b * LINE], (u64)it);
        }
    }
    out[row] = x + r + p + s;
}

template <int MODE>
static void run(const char *name, int nblk, int iters, int g1)
{
    const int n = nblk * BLOCK, ngroups = (nblk + g1 - 1) / g1;
    double *z, *slot, *gslot, *out; u64 *tflag; int *nbr, *wcol, *err;
    CK(hipMalloc(&z, 2 * (size_t)n * 8)); CK(hipMemset(z, 0, 2 * (size_t)n * 8));
    CK(hipMalloc(&slot, 2 * (size_t)nblk * LINE * 8)); CK(hipMemset(slot, 0, 2 * (size_t)nblk * LINE * 8));
    CK(hipMalloc(&gslot, 2 * (size_t)ngroups * LINE * 8)); CK(hipMemset(gslot, 0, 2 * (size_t)ngroups * LINE * 8));
    CK(hipMalloc(&tflag, (size_t)nblk * LINE * 8)); CK(hipMemset(tflag, 0, (size_t)nblk * LINE * 8));
    CK(hipMalloc(&out, (size_t)n * 8));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    std::vector<int> hn((size_t)nblk * NBR), hw((size_t)nblk * WQ * BLOCK);
    for (int b = 0; b < nblk; ++b) {
        for (int k = 0; k < NBR; ++k) hn[(size_t)b * NBR + k] = (b + (k - NBR / 2) * (k % 3 == 0 ? 1 : (k % 3 == 1 ? 7 : 29)) + 4 * nblk) % nblk;
        for (int q = 0; q < WQ; ++q)
            for (int t = 0; t < BLOCK; ++t) {
                const int src = hn[(size_t)b * NBR + (q * BLOCK + t) * NBR / (WQ * BLOCK)];
                hw[((size_t)b * WQ + q) * BLOCK + t] = src * BLOCK + (t * 5 + q * 11) % BLOCK;
            }
    }
    CK(hipMalloc(&nbr, hn.size() * 4)); CK(hipMemcpy(nbr, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&wcol, hw.size() * 4)); CK(hipMemcpy(wcol, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    int per_cu = 0, dev = 0; hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident<MODE>, BLOCK, 0));
    if ((long long)per_cu * prop.multiProcessorCount < nblk) { printf("%s: %d blocks do not fit (%d per CU x %d)\n", name, nblk, per_cu, prop.multiProcessorCount); return; }
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, dev));
    const long long timeout = (long long)rate * 1000 * 3;    // 3 s
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(tflag, 0, (size_t)nblk * LINE * 8)); CK(hipMemset(slot, 0, 2 * (size_t)nblk * LINE * 8)); CK(hipMemset(gslot, 0, 2 * (size_t)ngroups * LINE * 8));
        CK(hipEventRecord(e0));
        resident<MODE><<<nblk, BLOCK>>>(iters, nblk, g1, z, tflag, slot, gslot, nbr, wcol, timeout, err, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-44s blocks %4d (%d per CU allowed) g1 %2d: %.2f us per iteration%s\n", name, nblk, per_cu, g1, ms * 1e3 / iters, herr ? "  [TIMEOUT]" : "");
        if (herr) break;
    }
    hipFree(z); hipFree(slot); hipFree(gslot); hipFree(tflag); hipFree(out); hipFree(err); hipFree(nbr); hipFree(wcol);
}

int main(int argc, char **argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 881, iters = argc > 2 ? atoi(argv[2]) : 2000;
    run<0>("registers only (no exchange)", nblk, iters, 32);
    run<2>("tree reduce only", nblk, iters, 32);
    run<2>("tree reduce only", nblk, iters, 16);
    run<2>("tree reduce only", nblk, iters, 64);
    run<5>("z publish + neighbour flags + gather", nblk, iters, 32);
    run<7>("all: gather + tree reduce + publish", nblk, iters, 32);
    run<7>("all: gather + tree reduce + publish", nblk, iters, 16);
    run<7>("all", 256, iters, 16);
    run<7>("all", 512, iters, 32);
    run<7>("all", 1762, iters, 32);
    return 0;
}
