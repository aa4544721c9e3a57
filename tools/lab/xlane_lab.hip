// lab: cross-lane partners without the LDS crossbar (DPP + gfx950 permlane swaps), checked against __shfl_xor
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x16(double v)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (threadIdx.x & 16) != 0;
    return __hiloint2double((int)(odd ? b.x : b.y), (int)(odd ? a.x : a.y));
}
__device__ __forceinline__ double x32(double v)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const u2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = (threadIdx.x & 32) != 0;
    return __hiloint2double((int)(up ? b.x : b.y), (int)(up ? a.x : a.y));
}
__global__ void k(double *out)
{
    const int l = threadIdx.x;
    const double v = 1000.0 + l;
    out[0 * 64 + l] = dpp_f64<0xB1>(v) - __shfl_xor(v, 1, 64);
    out[1 * 64 + l] = dpp_f64<0x4E>(v) - __shfl_xor(v, 2, 64);
    out[2 * 64 + l] = dpp_f64<0x141>(v) - __shfl_xor(v, 7, 64);      // row_half_mirror
    out[3 * 64 + l] = dpp_f64<0x128>(v) - __shfl_xor(v, 8, 64);      // row_ror:8
    out[4 * 64 + l] = x16(v) - __shfl_xor(v, 16, 64);
    out[5 * 64 + l] = x32(v) - __shfl_xor(v, 32, 64);
}
int main()
{
    double *d, h[6 * 64];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 6; ++t) { double m = 0; for (int l = 0; l < 64; ++l) m += h[t * 64 + l] * h[t * 64 + l]; printf("pattern %d: mismatch %g\n", t, m); }
    return 0;
}
