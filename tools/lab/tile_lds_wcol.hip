// Lab v5: tile SpMV with an LDS x-window filled through a per-tile window->column map (coalesced loads of
// the map, run-wise contiguous gathers of x), 16-bit tile-local column ids, one workgroup per tile.
// Synthetic: n = 1 597 080 rows x 26 nnz, window = 9 runs of `per` columns around the tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
struct tile_meta { int nnz0; int row0; int w0; int pad; };
template<int U, int WMAX>
__global__ __launch_bounds__(256) void k(int n_tiles, const tile_meta* __restrict__ tm, const int* __restrict__ row_ptr,
    const int* __restrict__ wcol, const unsigned short* __restrict__ idx, const double* __restrict__ val,
    const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double xw[WMAX];
    __shared__ double prod[256*U];
    const int tid=threadIdx.x; const int xcd=blockIdx.x&7, bi=blockIdx.x>>3; const int Cx=(n_tiles+7)>>3;
    const int c=xcd*Cx+bi; if(bi>=Cx || c>=n_tiles) return;
    const tile_meta t=tm[c], t1=tm[c+1];
    const int base=t.nnz0, cnt=t1.nnz0-base, r0=t.row0, r1=t1.row0, w0=t.w0, W=t1.w0-w0;
    double v[U]; unsigned short ci[U];
    // window map first (its dependent gather is the longest chain), then the streams
    int wc[(WMAX+255)/256];
    #pragma unroll
    for(int q=0;q<(WMAX+255)/256;++q){ int w=q*256+tid; wc[q]= w<W ? __builtin_nontemporal_load(wcol+w0+w) : -1; }
    #pragma unroll
    for(int u=0;u<U;++u){ int i=u*256+tid; bool in=i<cnt; v[u]=in?__builtin_nontemporal_load(val+base+i):0.0; ci[u]=in?__builtin_nontemporal_load(idx+base+i):0; }
    int rr=r0+(tid>>2); int rp0=0,rp1=0; if(rr<r1){ rp0=row_ptr[rr]-base; rp1=row_ptr[rr+1]-base; }
    #pragma unroll
    for(int q=0;q<(WMAX+255)/256;++q){ int w=q*256+tid; if(wc[q]>=0) xw[w]=x[wc[q]]; }
    __syncthreads();
    #pragma unroll
    for(int u=0;u<U;++u){ int i=u*256+tid; if(i<cnt) prod[i]=v[u]*xw[ci[u]]; }
    __syncthreads();
    { const int l4=tid&3; if(rr<r1){ double s=0; for(int j=rp0+l4;j<rp1;j+=4) s+=prod[j]; s+=__shfl_xor(s,1,64); s+=__shfl_xor(s,2,64); if(l4==0) y[rr]=s; } }
}
int main(int argc,char**argv){
    const int n=1597080, deg=26; long nnz=(long)n*deg; int U=argc>1?atoi(argv[1]):8; int win=argc>2?atoi(argv[2]):450; long far=argc>3?atol(argv[3]):3000;
    int rows_per_tile=std::min(64,(256*U)/deg); int n_tiles=(n+rows_per_tile-1)/rows_per_tile;
    std::vector<int> rp(n+1); for(int i=0;i<=n;++i) rp[i]=i*deg;
    std::vector<tile_meta> tm(n_tiles+2); std::vector<unsigned short> idx(nnz); std::vector<double> val(nnz,1.0), x(n,1.0);
    srand(1); int per=win/9; int W=per*9; std::vector<int> wcol((size_t)(n_tiles+2)*W);
    for(int t=0;t<=n_tiles+1;++t){ int r0=std::min(t*rows_per_tile,n); tm[t].row0=r0; tm[t].nnz0=r0*deg; tm[t].w0=t*W;
        for(int r=0;r<9;++r){ long st=(long)r0 + (r-4)*far; if(st<0) st=0; if(st+per>n) st=n-per; for(int q=0;q<per;++q) wcol[(size_t)t*W+r*per+q]=(int)(st+q); } }
    for(long i=0;i<nnz;++i) idx[i]=rand()%W;
    int *drp,*dw; unsigned short* didx; double *dval,*dx,*dy; tile_meta* dtm;
    CK(hipMalloc(&drp,(n+1)*4)); CK(hipMalloc(&didx,nnz*2)); CK(hipMalloc(&dval,nnz*8)); CK(hipMalloc(&dx,n*8)); CK(hipMalloc(&dy,n*8)); CK(hipMalloc(&dtm,(n_tiles+2)*sizeof(tile_meta))); CK(hipMalloc(&dw,wcol.size()*4));
    CK(hipMemcpy(drp,rp.data(),(n+1)*4,hipMemcpyHostToDevice)); CK(hipMemcpy(didx,idx.data(),nnz*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dval,val.data(),nnz*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dx,x.data(),n*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dtm,tm.data(),(n_tiles+2)*sizeof(tile_meta),hipMemcpyHostToDevice)); CK(hipMemcpy(dw,wcol.data(),wcol.size()*4,hipMemcpyHostToDevice));
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run=[&](auto kern,const char*name){ int grid=((n_tiles+7)/8)*8; for(int w=0;w<3;++w) kern<<<grid,256>>>(n_tiles,dtm,drp,dw,didx,dval,dx,dy); CK(hipEventRecord(e0)); for(int i=0;i<20;++i) kern<<<grid,256>>>(n_tiles,dtm,drp,dw,didx,dval,dx,dy); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("%-6s U=%d win=%d far=%ld grid=%d rows/tile=%d  %8.2f us  alg(12nnz+20n) %.0f GB/s\n",name,U,W,far,grid,rows_per_tile,ms/20*1e3, (nnz*12.0+n*20.0)/(ms/20*1e-3)/1e9);};
    if(U==4) run(k<4,512>,"v5"); else if(U==8) run(k<8,1024>,"v5"); else run(k<2,512>,"v5");
    std::vector<double> yy(n); CK(hipMemcpy(yy.data(),dy,n*8,hipMemcpyDeviceToHost)); double s=0; for(int i=0;i<n;++i) s+=yy[i]; printf("checksum %.1f (expect %.1f)\n", s, (double)nnz);
    return 0; }
