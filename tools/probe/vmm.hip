// tools/probe/vmm.hip -- does the HIP virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap) work on this stack, and what does
// growing a mapping cost?  (The context's arena as ONE growing range instead of cached hipMalloc blocks.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/vmm.hip -o tools/probe/_bin/vmm && tools/probe/_bin/vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned long long* p, size_t n, unsigned long long v) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i * 512] = v + i; }
int main() {
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    const size_t VA = 240ull << 30, CH = 8ull << 30;
    void* base = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0));
    printf("reserve %zu GB: %.2f ms\n", VA >> 30, now() - t0);
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (int i = 0; i < 12; ++i) {
        hipMemGenericAllocationHandle_t h;
        double a = now();
        CK(hipMemCreate(&h, CH, &prop, 0));
        double b = now();
        CK(hipMemMap((char*)base + (size_t)i * CH, CH, 0, h, 0));
        double c = now();
        CK(hipMemSetAccess((char*)base + (size_t)i * CH, CH, &acc, 1));
        double d = now();
        hs.push_back(h);
        printf("chunk %d (8 GB): create %.2f map %.2f access %.2f ms\n", i, b - a, c - b, d - c);
    }
    // one kernel over the whole mapped range (contiguous across chunks)
    const size_t n = 12 * CH / 4096;
    double a = now();
    hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long*)base, n, 7ull);
    CK(hipDeviceSynchronize());
    printf("touch every 4 KB of 96 GB: %.2f ms\n", now() - a);
    a = now();
    for (int i = 11; i >= 6; --i) { CK(hipMemUnmap((char*)base + (size_t)i * CH, CH)); CK(hipMemRelease(hs[i])); }
    printf("unmap + release 48 GB: %.2f ms\n", now() - a);
    a = now();
    for (int i = 6; i < 12; ++i) { CK(hipMemCreate(&hs[i], CH, &prop, 0)); CK(hipMemMap((char*)base + (size_t)i * CH, CH, 0, hs[i], 0)); CK(hipMemSetAccess((char*)base + (size_t)i * CH, CH, &acc, 1)); }
    printf("map 48 GB again: %.2f ms\n", now() - a);
    void* p = nullptr;
    a = now(); CK(hipMalloc(&p, 48ull << 30)); printf("hipMalloc 48 GB next to it: %.2f ms\n", now() - a);
    a = now(); CK(hipFree(p)); printf("hipFree: %.2f ms\n", now() - a);
    a = now(); CK(hipMalloc(&p, 40ull << 30)); printf("hipMalloc 40 GB after the free: %.2f ms\n", now() - a);
    printf("ok\n");
    return 0;
}
