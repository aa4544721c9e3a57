// lab: is the device copy of hipMemcpy(device <- PAGEABLE host) complete when the call returns, as seen by a kernel that is
// launched right afterwards on a NON-BLOCKING stream?  (suspected behind the rare wrong pattern of in-process builds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
__global__ void poison(int *p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = -7; }
__global__ void check(const int *p, int n, int tag, int *bad) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (p[i] != tag + i) atomicAdd(bad, 1); }
std::atomic<long> nbad{0}, total{0};
void work(int tid, int iters, int mode)
{
    (void)hipSetDevice(0);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int *d_bad; (void)hipMalloc(&d_bad, 4);
    for (int it = 0; it < iters; ++it) {
        const int n = 200 + ((it * 7919 + tid * 104729) % 60000);
        int *d = nullptr; (void)hipMalloc(&d, n * sizeof(int));
        poison<<<64, 256, 0, st>>>(d, n);
        (void)hipMemsetAsync(d_bad, 0, 4, st);
        (void)hipStreamSynchronize(st);
        const int tag = tid * 1000000 + it * 7;
        std::vector<int> h(n);
        for (int i = 0; i < n; ++i) h[i] = tag + i;
        (void)hipMemcpy(d, h.data(), n * sizeof(int), hipMemcpyHostToDevice);          // synchronous, pageable source
        if (mode == 1) (void)hipStreamSynchronize(nullptr);
        check<<<64, 256, 0, st>>>(d, n, tag, d_bad);
        int b = 0;
        (void)hipMemcpyAsync(&b, d_bad, 4, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        if (b) { nbad += 1; if (nbad < 4) printf("thread %d iter %d: kernel saw %d of %d words not yet copied\n", tid, it, b, n); }
        total += 1;
        (void)hipFree(d);
    }
}
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 3000;
    for (int mode = 0; mode < 2; ++mode) {
        nbad = 0; total = 0;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t, iters, mode);
        for (auto &t : th) t.join();
        printf("%s: %d threads: %ld of %ld uploads were read incomplete by the next kernel on a non-blocking stream\n",
               mode ? "hipMemcpy + hipStreamSynchronize(null)" : "hipMemcpy alone", T, nbad.load(), total.load());
    }
    return 0;
}
