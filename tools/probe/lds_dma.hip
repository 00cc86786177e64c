// tools/probe/lds_dma.hip -- semantics check of the gfx950 LDS-DMA load (global_load_lds_dwordx4) as the count kernel uses it:
// lane l of a wave fetches the 16 bytes at its own global address; they land at (LDS base of the wave) + 16 * l.  Exec-masked
// lanes write nothing.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_dma.hip -o /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const uint4* __restrict__ src, uint4* dst, int n_active, int rounds) {
    extern __shared__ uint4 buf[];
    const int tid = threadIdx.x, T = blockDim.x;
    for (int r = 0; r < rounds; ++r) buf[r * T + tid] = make_uint4(0xDEAD0000u + tid, r, 0, 0);
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const int c = r * T + tid;                       // chunk index; two chunks = one 32-byte record
        if (c < n_active)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)blockIdx.x * rounds * T + c), (lptr_t)(buf + r * T + (tid & ~63)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int r = 0; r < rounds; ++r) dst[(size_t)blockIdx.x * rounds * T + r * T + tid] = buf[r * T + ((tid * 37) % T)];   // read what OTHER waves fetched
}
int main() {
    const int T = 768, rounds = 2, blocks = 512, n_active = 1000;
    const size_t n = (size_t)blocks * rounds * T;
    std::vector<uint4> h(n), out(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_uint4((uint32_t)i, (uint32_t)(i * 2654435761u), (uint32_t)(i >> 3), ~(uint32_t)i);
    uint4 *d, *o;
    hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(T), rounds * T * 16, 0, d, o, n_active, rounds);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(out.data(), o, n * 16, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int b = 0; b < blocks; ++b)
        for (int r = 0; r < rounds; ++r)
            for (int t = 0; t < T; ++t) {
                const int srcl = (t * 37) % T, c = r * T + srcl;
                const uint4 got = out[(size_t)b * rounds * T + r * T + t];
                uint4 exp = c < n_active ? h[(size_t)b * rounds * T + c] : make_uint4(0xDEAD0000u + srcl, r, 0, 0);
                if (got.x != exp.x || got.y != exp.y || got.z != exp.z || got.w != exp.w) { if (bad < 5) printf("mismatch b=%d r=%d t=%d got %08x exp %08x\n", b, r, t, got.x, exp.x); ++bad; }
            }
    printf("lds dma: %zu mismatches of %zu\n", bad, n);
    return bad ? 1 : 0;
}
