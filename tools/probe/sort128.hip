#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <random>
typedef unsigned __int128 u128;
int run(size_t n, unsigned bb, unsigned eb){
  std::mt19937_64 rng(1);
  std::vector<u128> k(n); std::vector<uint64_t> v(n);
  for(size_t i=0;i<n;i++){ k[i]=((u128)rng()<<64)| (rng() & ~0xFFFFFFFFull); v[i]=i; }
  u128 *dk,*dk2; uint64_t *dv,*dv2;
  hipMalloc(&dk,n*16); hipMalloc(&dk2,n*16); hipMalloc(&dv,n*8); hipMalloc(&dv2,n*8);
  hipMemcpy(dk,k.data(),n*16,hipMemcpyHostToDevice); hipMemcpy(dv,v.data(),n*8,hipMemcpyHostToDevice);
  size_t tb=0; hipError_t e=rocprim::radix_sort_pairs((void*)nullptr,tb,dk,dk2,dv,dv2,n,bb,eb,(hipStream_t)0);
  void* tmp; hipMalloc(&tmp,tb?tb:16);
  hipError_t e2=rocprim::radix_sort_pairs(tmp,tb,dk,dk2,dv,dv2,n,bb,eb,(hipStream_t)0);
  hipDeviceSynchronize();
  std::vector<u128> o(n); hipMemcpy(o.data(),dk2,n*16,hipMemcpyDeviceToHost);
  std::vector<u128> s=k; std::sort(s.begin(),s.end());
  bool ok = (s==o);
  printf("n=%zu bits[%u,%u) e=%d e2=%d tmp=%zu sorted_ok=%d first=%016lx%016lx exp=%016lx%016lx\n",n,bb,eb,(int)e,(int)e2,tb,(int)ok,(uint64_t)(o[0]>>64),(uint64_t)o[0],(uint64_t)(s[0]>>64),(uint64_t)s[0]);
  return 0;
}
int main(){ run(5000,0,128); run(5000,32,128); run(3000000,0,128); run(3000000,32,128); }
