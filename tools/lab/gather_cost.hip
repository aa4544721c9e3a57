// Diagnostic microbench: pieces of the stream SpMV on a synthetic banded random matrix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int U, int MODE>
__global__ __launch_bounds__(256) void k(int n_chunks, int cap, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ out, long nnz)
{
    __shared__ double prod[256*U];
    const int tid=threadIdx.x; const int xcd=blockIdx.x&7, bi=blockIdx.x>>3, nb8=gridDim.x>>3; const int Cx=(n_chunks+7)>>3;
    double acc=0;
    for(int g=bi; g<Cx; g+=nb8){ int c=xcd*Cx+g; if(c>=n_chunks) break; long base=(long)c*cap;
        double v[U]; int ci[U];
        #pragma unroll
        for(int u=0;u<U;++u){ long i=base+u*256+tid; bool in=i<nnz; v[u]= (MODE==2)?1.0:(in?val[i]:0.0); ci[u]= (MODE==3)?0:(in?col[i]:0);}        
        #pragma unroll
        for(int u=0;u<U;++u){ double xv = (MODE==1||MODE==3)? (double)ci[u] : x[ci[u]]; prod[u*256+tid]=v[u]*xv; }
        __syncthreads();
        // cheap reduction: each thread sums U entries
        double s=0; 
        #pragma unroll
        for(int u=0;u<U;++u) s+=prod[tid*U+u];
        acc+=s;
        __syncthreads();
    }
    out[blockIdx.x*256+tid]=acc;
}
int main(int argc,char**argv){
    int n=1597080; int deg=26; long nnz=(long)n*deg; int bw= argc>1?atoi(argv[1]):21000;
    std::vector<int> col(nnz); std::vector<double> val(nnz,1.0), x(n,1.0);
    srand(1);
    for(int r=0;r<n;++r){ int* c=&col[(long)r*deg]; for(int j=0;j<deg;++j){ long cc=r+ (long)(rand()%(2*bw+1))-bw; if(cc<0)cc=0; if(cc>=n)cc=n-1; c[j]=(int)cc;} std::sort(c,c+deg);}    
    int *dcol; double *dval,*dx,*dout; CK(hipMalloc(&dcol,nnz*4)); CK(hipMalloc(&dval,nnz*8)); CK(hipMalloc(&dx,n*8)); CK(hipMalloc(&dout,2048*256*8));
    CK(hipMemcpy(dcol,col.data(),nnz*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dval,val.data(),nnz*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dx,x.data(),n*8,hipMemcpyHostToDevice));
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run=[&](auto kern,const char*name,int U){ int cap=256*U; int nch=(int)((nnz+cap-1)/cap); for(int w=0;w<3;++w) kern<<<2048,256>>>(nch,cap,dcol,dval,dx,dout,nnz); hipEventRecord(e0); for(int i=0;i<20;++i) kern<<<2048,256>>>(nch,cap,dcol,dval,dx,dout,nnz); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("%-30s %8.2f us  %.0f GB/s(12B/nnz)\n",name,ms/20*1e3, nnz*12.0/(ms/20*1e-3)/1e9);};
    run(k<4,0>,"U4 full",4); run(k<4,1>,"U4 no gather",4); run(k<4,2>,"U4 no val (col+gather)",4); run(k<4,3>,"U4 val only",4);
    run(k<8,0>,"U8 full",8); run(k<8,1>,"U8 no gather",8); run(k<2,0>,"U2 full",2); run(k<2,1>,"U2 no gather",2);
    return 0; }
