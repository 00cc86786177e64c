// How long do hipMalloc / hipFree of large blocks take (the first call of a context allocates ~150 GB of arena)?
// Measured (ROCm 7.0, MI355X): 0.2-0.4 ms as a rule, whatever the size up to 112 GB -- and now and then 1.5-6 s for either call, at
// sizes that were instant a moment before (the driver pays for an earlier free's unmapping at its own time).  So the seconds a
// first call sometimes takes after another context or torch released memory are the driver's, not a property of the block sizes;
// the arena exists so that steady-state calls never get there.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/malloc_time.hip -o /tmp/malloc_time && /tmp/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
int main() {
    hipFree(0);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t gb : {60ull, 64ull, 66ull, 68ull, 72ull, 80ull, 96ull, 112ull}) {
            void* p = nullptr;
            auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&p, gb << 30);
            auto t1 = std::chrono::steady_clock::now();
            if (e != hipSuccess) { printf("%zu GB: %s\n", gb, hipGetErrorString(e)); continue; }
            hipMemset(p, 0, 1 << 20); hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            hipFree(p);
            auto t3 = std::chrono::steady_clock::now();
            printf("rep %d  %3zu GB: hipMalloc %8.1f ms, hipFree %8.1f ms\n", rep, gb, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                   std::chrono::duration<double, std::milli>(t3 - t2).count());
        }
    return 0;
}
