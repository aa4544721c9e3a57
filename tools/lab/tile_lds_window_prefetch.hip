// Lab: PIPELINED tile SpMV: next tile's values/indices/x-window/row_ptr prefetched into registers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
// tile meta: nnz0,row0 + window descriptor: per tile a flat list of W/256 source indices per thread is too heavy;
// instead runs: up to 16 runs; thread j of the window (j<wlen) finds its source by run prefix (computed on host into
// a per-tile small table run_off[17], run_start[16])
struct tile_meta { int nnz0; int row0; int wlen; int nrun; int run_start[16]; int run_off[16]; };
template<int U, int WPT /*window elems per thread*/>
__global__ __launch_bounds__(256) void k(int n_tiles, const tile_meta* __restrict__ tm, const int* __restrict__ row_ptr, const unsigned short* __restrict__ idx, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double xw[256*WPT];
    __shared__ double prod[256*U];
    const int tid=threadIdx.x; const int xcd=blockIdx.x&7, bi=blockIdx.x>>3, nb8=gridDim.x>>3; const int Cx=(n_tiles+7)>>3;
    double v[U]; unsigned short ci[U]; double wreg[WPT]; int rp0=0, rp1=0; int base=0,r0=0,r1=0,cnt=0,wlen=0;
    auto prefetch=[&](int c){
        const tile_meta* t=&tm[c]; base=t->nnz0; r0=t->row0; r1=tm[c+1].row0; cnt=tm[c+1].nnz0-base; wlen=t->wlen;
        #pragma unroll
        for(int u=0;u<U;++u){ int i=u*256+tid; bool in=i<cnt; v[u]=in?val[base+i]:0.0; ci[u]=in?idx[base+i]:0; }
        #pragma unroll
        for(int w=0;w<WPT;++w){ int j=w*256+tid; double xv=0; if(j<wlen){ int r=0; 
            #pragma unroll
            for(int q=1;q<16;++q) r += (q<t->nrun && j>=t->run_off[q]) ? 1:0;
            xv = x[t->run_start[r] + (j - t->run_off[r])]; } wreg[w]=xv; }
        int rr=r0+(tid>>2); if(rr<r1){ rp0=row_ptr[rr]-base; rp1=row_ptr[rr+1]-base; } else { rp0=rp1=0; }
    };
    int g=bi; int c=xcd*Cx+g; bool have = (g<Cx && c<n_tiles);
    if(have) prefetch(c);
    while(have){
        // commit window + products of the current tile
        #pragma unroll
        for(int w=0;w<WPT;++w) xw[w*256+tid]=wreg[w];
        __syncthreads();
        #pragma unroll
        for(int u=0;u<U;++u){ int i=u*256+tid; if(i<cnt) prod[i]=v[u]*xw[ci[u]]; }
        const int cr0=r0, cr1=r1, crp0=rp0, crp1=rp1;
        // prefetch the next tile while this one is reduced
        g+=nb8; c=xcd*Cx+g; have=(g<Cx && c<n_tiles);
        if(have) prefetch(c);
        __syncthreads();
        { const int l4=tid&3; int rr=cr0+(tid>>2); if(rr<cr1){ double s=0; for(int j=crp0+l4;j<crp1;j+=4) s+=prod[j]; s+=__shfl_xor(s,1,64); s+=__shfl_xor(s,2,64); if(l4==0) y[rr]=s; } }
        __syncthreads();
    }
}
int main(int argc,char**argv){
    const int n=1597080, deg=26; long nnz=(long)n*deg; int U=argc>1?atoi(argv[1]):4; int win=argc>2?atoi(argv[2]):480;
    int rows_per_tile=std::min(64,(256*U)/deg); int n_tiles=(n+rows_per_tile-1)/rows_per_tile;
    std::vector<int> rp(n+1); for(int i=0;i<=n;++i) rp[i]=i*deg;
    std::vector<tile_meta> tm(n_tiles+1); std::vector<unsigned short> idx(nnz); std::vector<double> val(nnz,1.0), x(n,1.0);
    srand(1);
    for(int t=0;t<=n_tiles;++t){ int r0=std::min(t*rows_per_tile,n); tm[t].row0=r0; tm[t].nnz0=r0*deg; tm[t].nrun=9; int per=win/9; tm[t].wlen=per*9; for(int r=0;r<9;++r){ long st=(long)r0 + (r-4)*3000L; if(st<0) st=0; if(st+per>n) st=n-per; tm[t].run_start[r]=(int)st; tm[t].run_off[r]=r*per; } }
    for(long i=0;i<nnz;++i) idx[i]=rand()%(win/9*9);
    int *drp; unsigned short* didx; double *dval,*dx,*dy; tile_meta* dtm;
    CK(hipMalloc(&drp,(n+1)*4)); CK(hipMalloc(&didx,nnz*2)); CK(hipMalloc(&dval,nnz*8)); CK(hipMalloc(&dx,n*8)); CK(hipMalloc(&dy,n*8)); CK(hipMalloc(&dtm,(n_tiles+1)*sizeof(tile_meta)));
    CK(hipMemcpy(drp,rp.data(),(n+1)*4,hipMemcpyHostToDevice)); CK(hipMemcpy(didx,idx.data(),nnz*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dval,val.data(),nnz*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dx,x.data(),n*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dtm,tm.data(),(n_tiles+1)*sizeof(tile_meta),hipMemcpyHostToDevice));
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run=[&](auto kern,const char*name,int grid){ for(int w=0;w<3;++w) kern<<<grid,256>>>(n_tiles,dtm,drp,didx,dval,dx,dy); CK(hipEventRecord(e0)); for(int i=0;i<20;++i) kern<<<grid,256>>>(n_tiles,dtm,drp,didx,dval,dx,dy); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("%-10s U=%d win=%d grid=%d rows/tile=%d  %8.2f us  alg(12nnz+20n) %.0f GB/s\n",name,U,win,grid,rows_per_tile,ms/20*1e3, (nnz*12.0+n*20.0)/(ms/20*1e-3)/1e9);};
    for(int grid: {1024,1536,2048}){
      if(U==4 && win<=512) run(k<4,2>,"pipe",grid); else if(U==4) run(k<4,4>,"pipe",grid); else if(U==8 && win<=1024) run(k<8,4>,"pipe",grid); else if(U==8) run(k<8,8>,"pipe",grid); else run(k<2,2>,"pipe",grid);
    }
    std::vector<double> yy(n); CK(hipMemcpy(yy.data(),dy,n*8,hipMemcpyDeviceToHost)); double s=0; for(int i=0;i<n;++i) s+=yy[i]; printf("checksum %.1f (expect %.1f)\n", s, (double)nnz);
    return 0; }
