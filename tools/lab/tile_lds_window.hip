// Lab v4: tile SpMV, LDS x-window, 16-bit local columns; tile meta prefetched one tile ahead (scalar),
// window filled run by run (block-uniform loop, no per-lane table lookups), one data round trip per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int MAXRUN=12;
struct tile_meta { int nnz0; int row0; int nrun; int pad; int run_start[MAXRUN]; int run_len[MAXRUN]; };   // 112 B
template<int U, int W, int PRE>
__global__ __launch_bounds__(256) void k(int n_tiles, const tile_meta* __restrict__ tm, const int* __restrict__ row_ptr, const unsigned short* __restrict__ idx, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double xw[W];
    __shared__ double prod[256*U];
    __shared__ tile_meta smeta[2];
    const int tid=threadIdx.x; const int xcd=blockIdx.x&7, bi=blockIdx.x>>3, nb8=gridDim.x>>3; const int Cx=(n_tiles+7)>>3;
    int g=bi; int c=xcd*Cx+g; bool have=(g<Cx && c<n_tiles);
    // meta of the first tile
    if(have && tid< (int)(sizeof(tile_meta)/4)+2) ((int*)&smeta[0])[tid] = ((const int*)&tm[c])[tid];   // also next tile's nnz0,row0
    int buf=0;
    __syncthreads();
    while(have){
        const tile_meta* t=&smeta[buf];
        const int base=t->nnz0, r0=t->row0; 
        const int* nxt=(const int*)(t+1); // (nnz0,row0 of next tile were copied behind the struct)  -- lab shortcut
        const int cnt = ((const int*)&smeta[buf])[sizeof(tile_meta)/4] - base; const int r1=((const int*)&smeta[buf])[sizeof(tile_meta)/4+1];
        (void)nxt;
        // issue every load of this tile at once
        double v[U]; unsigned short ci[U];
        #pragma unroll
        for(int u=0;u<U;++u){ int i=u*256+tid; bool in=i<cnt; v[u]=in?__builtin_nontemporal_load(val+base+i):0.0; ci[u]=in?__builtin_nontemporal_load(idx+base+i):0; }
        int rr=r0+(tid>>2); int rp0=0,rp1=0; if(rr<r1){ rp0=row_ptr[rr]-base; rp1=row_ptr[rr+1]-base; }
        int off=0; const int nrun=t->nrun;
        for(int r=0;r<nrun;++r){ const int st=t->run_start[r], ln=t->run_len[r]; if(tid<ln) xw[off+tid]=x[st+tid]; off+=ln; }
        // next tile's meta (independent of everything above)
        int gn=g+nb8, cn=xcd*Cx+gn; bool have_n=(gn<Cx && cn<n_tiles);
        if(have_n && tid<(int)(sizeof(tile_meta)/4)+2) ((int*)&smeta[buf^1])[tid] = ((const int*)&tm[cn])[tid];
        __syncthreads();
        #pragma unroll
        for(int u=0;u<U;++u){ int i=u*256+tid; if(i<cnt) prod[i]=v[u]*xw[ci[u]]; }
        __syncthreads();
        { const int l4=tid&3; if(rr<r1){ double s=0; for(int j=rp0+l4;j<rp1;j+=4) s+=prod[j]; s+=__shfl_xor(s,1,64); s+=__shfl_xor(s,2,64); if(l4==0) y[rr]=s; } }
        g=gn; c=cn; have=have_n; buf^=1;
        __syncthreads();
    }
}
int main(int argc,char**argv){
    const int n=1597080, deg=26; long nnz=(long)n*deg; int U=argc>1?atoi(argv[1]):4; int win=argc>2?atoi(argv[2]):450;
    int rows_per_tile=std::min(64,(256*U)/deg); int n_tiles=(n+rows_per_tile-1)/rows_per_tile;
    std::vector<int> rp(n+1); for(int i=0;i<=n;++i) rp[i]=i*deg;
    std::vector<tile_meta> tm(n_tiles+2); std::vector<unsigned short> idx(nnz); std::vector<double> val(nnz,1.0), x(n,1.0);
    srand(1); int per=win/9;
    for(int t=0;t<=n_tiles+1;++t){ int r0=std::min(t*rows_per_tile,n); tm[t].row0=r0; tm[t].nnz0=r0*deg; tm[t].nrun=9; for(int r=0;r<9;++r){ long st=(long)r0 + (r-4)*3000L; if(st<0) st=0; if(st+per>n) st=n-per; tm[t].run_start[r]=(int)st; tm[t].run_len[r]=per; } }
    for(long i=0;i<nnz;++i) idx[i]=rand()%(per*9);
    int *drp; unsigned short* didx; double *dval,*dx,*dy; tile_meta* dtm;
    CK(hipMalloc(&drp,(n+1)*4)); CK(hipMalloc(&didx,nnz*2)); CK(hipMalloc(&dval,nnz*8)); CK(hipMalloc(&dx,n*8)); CK(hipMalloc(&dy,n*8)); CK(hipMalloc(&dtm,(n_tiles+2)*sizeof(tile_meta)));
    CK(hipMemcpy(drp,rp.data(),(n+1)*4,hipMemcpyHostToDevice)); CK(hipMemcpy(didx,idx.data(),nnz*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dval,val.data(),nnz*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dx,x.data(),n*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dtm,tm.data(),(n_tiles+2)*sizeof(tile_meta),hipMemcpyHostToDevice));
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run=[&](auto kern,const char*name,int grid){ for(int w=0;w<3;++w) kern<<<grid,256>>>(n_tiles,dtm,drp,didx,dval,dx,dy); CK(hipEventRecord(e0)); for(int i=0;i<20;++i) kern<<<grid,256>>>(n_tiles,dtm,drp,didx,dval,dx,dy); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("%-6s U=%d win=%d grid=%d rows/tile=%d  %8.2f us  alg(12nnz+20n) %.0f GB/s\n",name,U,win,grid,rows_per_tile,ms/20*1e3, (nnz*12.0+n*20.0)/(ms/20*1e-3)/1e9);};
    for(int grid: {2048,3072,4096}){
      if(U==4) run(k<4,1024,0>,"v4",grid); else if(U==8) run(k<8,2048,0>,"v4",grid); else run(k<2,512,0>,"v4",grid);
    }
    std::vector<double> yy(n); CK(hipMemcpy(yy.data(),dy,n*8,hipMemcpyDeviceToHost)); double s=0; for(int i=0;i<n;++i) s+=yy[i]; printf("checksum %.1f (expect %.1f)\n", s, (double)nnz);
    return 0; }
