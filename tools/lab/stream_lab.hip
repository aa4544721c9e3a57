// lab: what a pure streaming read (and read+write) reaches on this part, as the roof for the SpMV numbers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dvec2 __attribute__((ext_vector_type(2)));
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const dvec2 *__restrict__ a, size_t n2, double *__restrict__ out)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        dvec2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { const dvec2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void triad_kernel(const dvec2 *__restrict__ a, const dvec2 *__restrict__ b, dvec2 *__restrict__ c, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) { const dvec2 x = a[i], y = b[i]; dvec2 r; r.x = x.x + 2.0 * y.x; r.y = x.y + 2.0 * y.y; c[i] = r; }
}
int main()
{
    const size_t bytes = (size_t)2 << 30, n2 = bytes / 16;
    dvec2 *a, *b, *c; double *out;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&c, bytes); (void)hipMalloc(&out, 8);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](auto f, const char *name, double gb) {
        f(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int r = 0; r < 10; ++r) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.1f us  %6.2f TB/s\n", name, ms * 100, gb * 10 / ms / 1e3);
    };
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "read x4 plain, grid %d", grid); time([&] { read_kernel<0, 4><<<grid, 256>>>(a, n2, out); }, nm, bytes / 1e9);
        snprintf(nm, 64, "read x4 nontemporal, grid %d", grid); time([&] { read_kernel<1, 4><<<grid, 256>>>(a, n2, out); }, nm, bytes / 1e9);
    }
    time([&] { read_kernel<1, 8><<<4096, 256>>>(a, n2, out); }, "read x8 nontemporal, grid 4096", bytes / 1e9);
    time([&] { triad_kernel<<<8192, 256>>>(a, b, c, n2); }, "triad (2 reads + 1 write), grid 8192", 3 * bytes / 1e9);
    // the size of the beyond-cache SpMV probe: 350 MB
    const size_t n350 = (size_t)350e6 / 16;
    time([&] { read_kernel<1, 4><<<4096, 256>>>(a, n350, out); }, "read 350 MB nontemporal, grid 4096", 0.35);
    time([&] { read_kernel<0, 4><<<4096, 256>>>(a, n350, out); }, "read 350 MB plain, grid 4096", 0.35);
    return 0;
}
