// Issue rate of the integer VALU instructions the hashes are built from (wave64, gfx950): which ones are full rate?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o tools/probe/_bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// MODE 0: xor/add chain   1: v_mul_lo_u32   2: v_mul_u32_u24   3: v_mad_u32_u24   4: rotate (v_alignbit)  5: v_mul_hi_u32
// 6: 64-bit shift  7: v_perm  8: v_bfe
template <int MODE>
__global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed * (threadIdx.x + 1 + j) + blockIdx.x;
    uint64_t q = ((uint64_t)a[0] << 32) | a[1];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) a[j] = (a[j] ^ seed) + 0x9E3779B1u;
            if (MODE == 1) a[j] = a[j] * 0x9E3779B1u;
            if (MODE == 2) a[j] = __umul24(a[j], 0x9E3779u) ^ seed;
            if (MODE == 3) a[j] = __umul24(a[j], 0x9E3779u) + seed;
            if (MODE == 4) a[j] = __builtin_rotateleft32(a[j], 13) ;
            if (MODE == 5) a[j] = __umulhi(a[j], 0x9E3779B1u);
            if (MODE == 7) a[j] = __builtin_amdgcn_perm(a[j], seed, 0x00010203u);
            if (MODE == 8) a[j] = __builtin_amdgcn_ubfe(a[j], 3, 27) + 1;
        }
        if (MODE == 6) { q = (q << 3) ^ (q >> 5); }
        if (MODE == 4) a[0] ^= i;
    }
    uint32_t s = (uint32_t)q;
#pragma unroll
    for (int j = 0; j < 8; ++j) s ^= a[j];
    if (s == 0x12345u) out[0] = s;
}
template <int MODE>
void run(const char* name, uint32_t* out, int per_iter) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096, blocks = 256 * 32;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, out, 12345u + rep, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double wave_instr = (double)blocks * 4 * iters * per_iter;           // wave-level instructions
        double per_simd_per_clk = wave_instr / (256.0 * 4) / (ms * 1e-3 * 2.4e9);
        if (rep) printf("%-28s %8.3f ms  %.3f wave-instr/clk/SIMD  (%.1f clk per instr)\n", name, ms, per_simd_per_clk, 1.0 / per_simd_per_clk);
    }
}
int main() {
    uint32_t* out; hipMalloc(&out, 64);
    run<0>("xor+add (2 ops)", out, 16);
    run<1>("v_mul_lo_u32", out, 8);
    run<2>("v_mul_u32_u24 + xor", out, 16);
    run<3>("v_mad_u32_u24", out, 8);
    run<4>("rotate", out, 8);
    run<5>("v_mul_hi_u32", out, 8);
    run<6>("64-bit shl/shr/xor", out, 3);
    run<7>("v_perm_b32", out, 8);
    run<8>("v_bfe_u32 + add", out, 16);
    return 0;
}
