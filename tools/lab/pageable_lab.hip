// lab: do concurrent host threads corrupt each other's hipMemcpyAsync(device -> PAGEABLE host) on their own streams?
// (suspected behind a rare wrong column list in the in-process multi-rank pattern build)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
__global__ void fill(int *p, int n, int tag) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = tag * 1000003 + i; }
std::atomic<long> bad{0}, total{0};
void work(int tid, int iters, int mode)
{
    (void)hipSetDevice(0);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int it = 0; it < iters; ++it) {
        const int n = 500 + ((it * 7919 + tid * 104729) % 40000);
        int *d = nullptr; (void)hipMalloc(&d, n * sizeof(int));
        const int tag = tid * 100000 + it;
        fill<<<(n + 255) / 256, 256, 0, st>>>(d, n, tag);
        std::vector<int> h(n, -1);
        int *hp = h.data();
        int *pin = nullptr;
        if (mode == 1) { (void)hipHostMalloc(&pin, n * sizeof(int)); hp = pin; }
        (void)hipMemcpyAsync(hp, d, n * sizeof(int), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        long b = 0;
        for (int i = 0; i < n; ++i) b += hp[i] != tag * 1000003 + i;
        if (b) { bad += 1; if (bad < 5) printf("thread %d iter %d: %ld of %d words wrong (first words %d %d, expected %d)\n", tid, it, b, n, hp[0], hp[1], tag * 1000003); }
        total += 1;
        if (pin) (void)hipHostFree(pin);
        (void)hipFree(d);
    }
}
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 3000;
    for (int mode = 0; mode < 2; ++mode) {
        bad = 0; total = 0;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t, iters, mode);
        for (auto &t : th) t.join();
        printf("%s host buffers, %d threads: %ld of %ld copies corrupted\n", mode ? "pinned" : "pageable", T, bad.load(), total.load());
    }
    return 0;
}
