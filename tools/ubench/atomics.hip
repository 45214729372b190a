// Microbenchmark: what does a scattered 64-bit atomic max cost on MI355X, vs stores / 32-bit / scopes / locality?
// hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics && ./atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int MODE> __global__ __launch_bounds__(256) void k(const u32* __restrict__ idx, u64 n, u64* f64, u32* f32, u64 tag){
  u64 i=(u64)blockIdx.x*256+threadIdx.x; if(i>=n) return; u32 c=idx[i]; u64 key=tag|(i<<16)|(c&0xfff);
  if(MODE==0) __hip_atomic_fetch_max(&f64[c],key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
  if(MODE==1) __hip_atomic_fetch_max(&f32[c],(u32)key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
  if(MODE==2) f64[c]=key;
  if(MODE==3) __hip_atomic_fetch_max(&f64[c],key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);
  if(MODE==4) { u64 o=__hip_atomic_fetch_max(&f64[c],key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); if(o==0x1234) f32[0]=1; }
  if(MODE==5) __hip_atomic_fetch_max(&f64[c],key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_SYSTEM);
  if(MODE==6) f32[c]=(u32)key;
  if(MODE==7) { u64 v=f64[c]; if(v==0x1234) f32[0]=1; }   // pure gather
  if(MODE==8) __hip_atomic_fetch_max(&f64[c],key,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WAVEFRONT);
}
template<int MODE> float run(const u32* d_idx,u64 n,u64* f64,u32* f32,int reps){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int r=0;r<3;r++) hipLaunchKernelGGL(k<MODE>,dim3((n+255)/256),dim3(256),0,0,d_idx,n,f64,f32,(u64)(r+1)<<44);
  CK(hipEventRecord(a,0));
  for(int r=0;r<reps;r++) hipLaunchKernelGGL(k<MODE>,dim3((n+255)/256),dim3(256),0,0,d_idx,n,f64,f32,(u64)(r+10)<<44);
  CK(hipEventRecord(b,0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  const u64 n=1000000; const u32 W=1760,H=1320; const u64 cells=(u64)W*H;
  u64* f64; u32* f32; u32* d_idx; CK(hipMalloc(&f64,cells*8)); CK(hipMalloc(&f32,cells*4)); CK(hipMalloc(&d_idx,n*4));
  CK(hipMemset(f64,0,cells*8)); CK(hipMemset(f32,0,cells*4));
  std::vector<u32> idx(n);
  const char* names[]={"random over frame","band: x~time (+-16 cols), y random  [row-major frame]","band, transposed frame [x][y]","all distinct sequential cells","sorted by cell (random set)"};
  for(int pat=0;pat<5;pat++){
    srand(1);
    for(u64 i=0;i<n;i++){
      u32 y=rand()%H; u32 xc=(u32)((double)i/n*(W-40))+ (rand()%32); 
      if(pat==0) idx[i]=((u64)rand()*RAND_MAX+rand())%cells;
      if(pat==1) idx[i]=y*W+xc;
      if(pat==2) idx[i]=xc*H+y;
      if(pat==3) idx[i]=i;
      if(pat==4) idx[i]=((u64)rand()*RAND_MAX+rand())%cells;
    }
    if(pat==4) std::sort(idx.begin(),idx.end());
    CK(hipMemcpy(d_idx,idx.data(),n*4,hipMemcpyHostToDevice));
    printf("pattern %d: %s\n",pat,names[pat]);
    printf("  atomic umax u64 agent   : %7.2f us\n",run<0>(d_idx,n,f64,f32,50));
    printf("  atomic umax u32 agent   : %7.2f us\n",run<1>(d_idx,n,f64,f32,50));
    printf("  plain store u64         : %7.2f us\n",run<2>(d_idx,n,f64,f32,50));
    printf("  plain store u32         : %7.2f us\n",run<6>(d_idx,n,f64,f32,50));
    printf("  atomic u64 workgroup    : %7.2f us\n",run<3>(d_idx,n,f64,f32,50));
    printf("  atomic u64 wavefront    : %7.2f us\n",run<8>(d_idx,n,f64,f32,50));
    printf("  atomic u64 agent+return : %7.2f us\n",run<4>(d_idx,n,f64,f32,50));
    printf("  atomic u64 system       : %7.2f us\n",run<5>(d_idx,n,f64,f32,50));
    printf("  gather u64 (load only)  : %7.2f us\n",run<7>(d_idx,n,f64,f32,50));
  }
  return 0;
}
