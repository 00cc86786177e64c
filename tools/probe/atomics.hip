// Returning vs non-returning global atomics to random counters (the slot reservation pattern of the minimiser
// partition): what does one cost, and is it latency or throughput that binds?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int PER, int MODE, int STRIDE = 1>   // STRIDE: counters padded to one per STRIDE words
// MODE 0: non-returning, 1: returning serial chain (value feeds the next address), 2: returning independent,
// 3: returning, WORKGROUP scope (executes in the issuing XCD's L2, no cross-XCD coherence), 4: the same on a counter
//    range private to the XCD (block b runs on XCD b % 8)
__global__ void __launch_bounds__(256) probe(uint32_t* ctr, uint32_t nb, uint32_t* sink) {
#define CTR(i) ctr[(size_t)(i) * STRIDE]
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0, r[PER];
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < PER; ++i) r[i] = atomicAdd(&CTR(mix(t * PER + i) % nb), 1u);
#pragma unroll
        for (int i = 0; i < PER; ++i) acc += r[i];
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            uint32_t a = mix(t * PER + i + (MODE == 1 ? (acc & 1u) : 0u)) % nb;
            if (MODE == 0) atomicAdd(&CTR(a), 1u);
            else if (MODE == 3) acc += __hip_atomic_fetch_add(&CTR(a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 4) acc += __hip_atomic_fetch_add(&CTR((a % (nb / 8)) + (nb / 8) * (blockIdx.x & 7u)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += atomicAdd(&CTR(a), 1u);
        }
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
template <int PER, int MODE, int STRIDE = 1>
void run(const char* name, uint32_t* ctr, uint32_t nb, uint32_t* sink, uint64_t total) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned blocks = (unsigned)(total / PER / 256);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(ctr, 0, (size_t)nb * 4 * STRIDE);
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<PER, MODE, STRIDE>), dim3(blocks), dim3(256), 0, 0, ctr, nb, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) printf("%-42s %8.2f ms  %7.1f G atomics/s\n", name, ms, total / ms / 1e6);
    }
}
int main() {
    uint32_t nb = 2550000; uint64_t total = 671088640ull;
    uint32_t *ctr, *sink; hipMalloc(&ctr, (size_t)nb * 4 * 16); hipMalloc(&sink, 4);
    run<1, 0>("non-returning, 1/thread", ctr, nb, sink, total);
    run<8, 0>("non-returning, 8/thread", ctr, nb, sink, total);
    run<1, 1>("returning, 1/thread", ctr, nb, sink, total);
    run<8, 1>("returning, 8/thread dependent chain", ctr, nb, sink, total);
    run<8, 2>("returning, 8/thread independent", ctr, nb, sink, total);
    run<16, 2>("returning, 16/thread independent", ctr, nb, sink, total);
    run<1, 1, 4>("returning, 1/thread, counters 16 B apart", ctr, nb, sink, total);
    run<1, 1, 16>("returning, 1/thread, counters 64 B apart", ctr, nb, sink, total);
    run<1, 0, 16>("non-returning, counters 64 B apart", ctr, nb, sink, total);
    run<1, 3>("returning, workgroup scope", ctr, nb, sink, total);
    run<1, 4>("returning, workgroup scope, XCD-private", ctr, nb, sink, total);
    run<8, 4>("same, 8/thread chain", ctr, nb, sink, total);
    return 0;
}
