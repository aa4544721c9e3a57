// lab: wave-wide inclusive scan of doubles with DPP (row_shr 1,2,4,8 + row_bcast15/31) and wave_shr:1, against a serial scan
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp0(double v)          // lanes without a source (or outside ROWMASK) receive 0.0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double scan_incl(double v)
{
    v += dpp0<0x111, 0xf>(v);     // row_shr:1
    v += dpp0<0x112, 0xf>(v);     // row_shr:2
    v += dpp0<0x114, 0xf>(v);     // row_shr:4
    v += dpp0<0x118, 0xf>(v);     // row_shr:8
    v += dpp0<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
    v += dpp0<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3
    return v;
}
__global__ void k(double *out)
{
    const int l = threadIdx.x;
    const double v = 1.0 + (l * 7 % 13);
    const double inc = scan_incl(v);
    out[l] = inc;
    out[64 + l] = dpp0<0x138, 0xf>(inc);   // wave_shr:1 -> exclusive
    out[128 + l] = v;
}
int main()
{
    double *d, h[192];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double run = 0, bad = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) { const double ex = run; run += h[128 + l]; bad += (h[l] - run) * (h[l] - run); bad2 += (h[64 + l] - ex) * (h[64 + l] - ex); }
    printf("inclusive mismatch %g, exclusive mismatch %g (total %g)\n", bad, bad2, run);
    return 0;
}
