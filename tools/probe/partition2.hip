// tools/probe/partition2.hip -- what would a two-level (coarse bin, LDS-staged) minimiser partition cost?  Measured, not costed.
//
// The one-pass partition of snk_msp.hip pays one returning device atomic and two scattered 16-byte stores per supermer record
// (0.69 G records of 32 B at the bench workload).  VERDICT r3 asks for the alternative to be measured on the device: records
// staged in LDS, sorted by a coarse bin, written as runs with ONE reservation per (tile, bin) and full sectors -- and a second
// level that brings the coarse bins to the granularity the count kernel consumes.  This program times the building blocks of
// every such design on synthetic 32-byte records with uniformly random bucket ids (the bucket hash of snk_common.h is uniform):
//
//   direct      today's emission on its own: coalesced read, atomicAdd(cursor[fine bucket]) returning, 2 x 16-byte stores to the
//               slot (NB = 2^21 fine buckets)                                    -> the floor the one-pass design sits on
//   direct2     the same with lane pairs storing the two halves of ONE record side by side (32 contiguous bytes per pair)
//   level<B,T>  one radix level: a workgroup stages a tile of T records in LDS, histogram over B bins (LDS atomics), one global
//               reservation per non-empty (tile, bin), the tile written out in bin order by consecutive lanes (16 B per lane,
//               runs of T/B records contiguous in HBM)                            -> level 1 (the emission of the scan kernel)
//               and level 2 (coarse bin -> super-bucket) are this kernel with different digits
//   index       level 3 for a count kernel that gathers: per super-bucket (32 fine buckets, ~10.5 k records) a u16 index list in
//               fine-bucket order (one workgroup per super-bucket, LDS counting sort over the record's bucket field)
//   gather      the count kernel's read: records of a super-bucket fetched by LDS-DMA in index-list order (512-record batches,
//               two lanes per record) against the same loop over contiguous records
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/partition2.hip -o tools/probe/_bin/partition2 && tools/probe/_bin/partition2 [records]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;

__host__ __device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
constexpr uint32_t NB_LOG = 21, NB = 1u << NB_LOG;            // fine buckets
__device__ __forceinline__ uint32_t rec_bucket(uint32_t w6) { return w6 >> 9; }     // (21 bits above the nine flag bits of word 6)

// synthetic records: bucket id in word 6, payload derived from the index (checked after every kernel by a sum)
__global__ void __launch_bounds__(256) gen_kernel(uint4* rec, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = mix((uint32_t)i), b = mix(a ^ 0x9E3779B9u);
    rec[2 * i] = make_uint4(a, b, a ^ b, (uint32_t)i);
    rec[2 * i + 1] = make_uint4(b + 1, a + 2, ((mix(b + 7) >> (32 - NB_LOG)) << 9) | 17u, (uint32_t)(i >> 3));
}
__global__ void __launch_bounds__(256) sum_kernel(const uint4* rec, const uint32_t* cursor, uint32_t nbins, uint64_t cap, unsigned long long* out) {
    // sum of word 3 (the index) over the records that are there: bin b holds cursor[b] records at b * cap
    unsigned long long s = 0, c = 0;
    for (uint32_t b = blockIdx.x; b < nbins; b += gridDim.x)
        for (uint32_t i = threadIdx.x; i < cursor[b]; i += 256) { s += rec[2 * ((uint64_t)b * cap + i)].w; ++c; }
    for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s); atomicAdd(&out[1], c); }
}

// ---- today's emission (without the scan)
template <bool PAIR>
__global__ void __launch_bounds__(256) direct_kernel(const uint4* __restrict__ in, uint64_t n, uint32_t* cursor, uint32_t cap, uint4* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    uint64_t at = 0;
    const bool live = i < n;
    if (live) {
        a = in[2 * i]; b = in[2 * i + 1];
        const uint32_t bk = rec_bucket(b.z);
        const uint32_t slot = atomicAdd(&cursor[bk], 1u);
        at = (uint64_t)bk * cap + (slot < cap ? slot : cap - 1);
    }
    if (!PAIR) { if (live) { out[2 * at] = a; out[2 * at + 1] = b; } }
    else {
        // lane pair (2i, 2i+1): first the even lane's record (even lane its low half, odd lane its high half), then the odd lane's
        const bool odd = threadIdx.x & 1;
        auto sw = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };   // quad_perm [1,0,3,2]
        const uint4 give = odd ? a : b;                       // what my partner stores for me
        const uint4 got = make_uint4(sw(give.x), sw(give.y), sw(give.z), sw(give.w));
        const uint32_t plo = sw((uint32_t)at), phi = sw((uint32_t)(at >> 32)), plive = sw(live ? 1u : 0u);
        const uint64_t pat = ((uint64_t)phi << 32) | plo;
        // store 1: the even lane's record
        if (!odd) { if (live) out[2 * at] = a; } else { if (plive) out[2 * pat + 1] = got; }
        // store 2: the odd lane's record
        if (!odd) { if (plive) out[2 * pat] = got; } else { if (live) out[2 * at + 1] = b; }
    }
}

// ---- one radix level through LDS
template <int BINS, int TILE, int THREADS>
__global__ void __launch_bounds__(THREADS) level_kernel(const uint4* __restrict__ in, const uint32_t* __restrict__ in_cursor, uint64_t in_cap, uint32_t tiles_per_bin,
                                                        uint32_t shift, uint32_t* cursor, uint64_t cap, uint32_t bin_mul, uint4* out, uint32_t* ovf) {
    // input: bin `ib` holds in_cursor[ib] records at ib * in_cap (level 1: ONE input bin holding everything); a workgroup takes tile
    // t of input bin ib; output bin = ib * bin_mul + digit, digit = (bucket >> shift) % BINS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* rec = reinterpret_cast<uint4*>(smem);                       // [TILE][2]
    uint16_t* order = reinterpret_cast<uint16_t*>(rec + 2 * TILE);     // [TILE] source record of sorted position
    uint16_t* dig = order + TILE;                                      // [TILE] its digit
    uint32_t* hist = reinterpret_cast<uint32_t*>(dig + TILE);          // [BINS]
    uint32_t* bstart = hist + BINS;                                    // [BINS] exclusive scan
    uint32_t* gbase = bstart + BINS;                                   // [BINS] reserved position in the output bin
    uint32_t* wsum = gbase + BINS;                                     // [THREADS/64]
    const int tid = threadIdx.x;
    const uint32_t ib = blockIdx.x / tiles_per_bin, t = blockIdx.x % tiles_per_bin;
    const uint64_t n_in = in_cursor[ib];
    const uint64_t r0 = (uint64_t)t * TILE;
    if (r0 >= n_in) return;
    const uint32_t nrec = (uint32_t)(n_in - r0 < (uint64_t)TILE ? n_in - r0 : (uint64_t)TILE);
    for (int b = tid; b < BINS; b += THREADS) hist[b] = 0;
    __syncthreads();
    constexpr int PER = TILE / THREADS;
    const uint4* src = in + 2 * (ib * in_cap + r0);
    // stage: coalesced 16-byte chunks
    uint4 v[2 * PER];
#pragma unroll
    for (int k = 0; k < 2 * PER; ++k) { const uint32_t c = k * THREADS + tid; v[k] = c < 2 * nrec ? src[c] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int k = 0; k < 2 * PER; ++k) rec[k * THREADS + tid] = v[k];
    __syncthreads();
    uint32_t myd[PER], myr[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t r = k * THREADS + tid;
        myd[k] = 0; myr[k] = 0;
        if (r < nrec) {
            const uint32_t w6 = reinterpret_cast<const uint32_t*>(rec)[8 * r + 6];
            myd[k] = (rec_bucket(w6) >> shift) & (BINS - 1);
            myr[k] = atomicAdd(&hist[myd[k]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the histogram (BINS <= 4 * THREADS handled by a per-thread chunk) + the global reservations
    {
        constexpr int CH = (BINS + THREADS - 1) / THREADS;
        uint32_t loc[CH], s = 0;
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int b = tid * CH + q; loc[q] = b < BINS ? hist[b] : 0; s += loc[q]; }
        uint32_t inc = s;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o); if ((tid & 63) >= o) inc += x; }
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        uint32_t wb = 0;
        for (int w = 0; w < (tid >> 6); ++w) wb += wsum[w];
        uint32_t ex = wb + inc - s;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int b = tid * CH + q;
            if (b < BINS) {
                bstart[b] = ex; ex += loc[q];
                uint32_t g = 0;
                if (loc[q]) g = atomicAdd(&cursor[ib * bin_mul + b], loc[q]);
                if (loc[q] && (uint64_t)g + loc[q] > cap) { atomicAdd(ovf, 1u); g = 0; }
                gbase[b] = g;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t r = k * THREADS + tid;
        if (r < nrec) { const uint32_t pos = bstart[myd[k]] + myr[k]; order[pos] = (uint16_t)r; dig[pos] = (uint16_t)myd[k]; }
    }
    __syncthreads();
    // write-out in bin order: consecutive lanes = consecutive 16-byte chunks of the sorted tile
#pragma unroll
    for (int k = 0; k < 2 * PER; ++k) {
        const uint32_t c = k * THREADS + tid, pos = c >> 1;
        if (pos < nrec) {
            const uint32_t d = dig[pos], s = order[pos];
            const uint64_t at = (uint64_t)(ib * bin_mul + d) * cap + gbase[d] + (pos - bstart[d]);
            out[2 * at + (c & 1)] = rec[2 * s + (c & 1)];
        }
    }
}

// ---- level 3: u16 index list of a super-bucket in fine-bucket order + the 32 fine offsets
template <int FINE, int THREADS>
__global__ void __launch_bounds__(THREADS) index_kernel(const uint4* __restrict__ rec, const uint32_t* __restrict__ cursor, uint64_t cap, uint16_t* idx, uint32_t* fine_off) {
    __shared__ uint32_t hist[FINE], start[FINE];
    const uint32_t sb = blockIdx.x, n = cursor[sb];
    const int tid = threadIdx.x;
    if (tid < FINE) hist[tid] = 0;
    __syncthreads();
    const uint32_t* w = reinterpret_cast<const uint32_t*>(rec + 2 * (uint64_t)sb * cap);
    constexpr int MAXPER = 24;                             // up to 24 * THREADS records per super-bucket
    uint16_t rk[MAXPER]; uint8_t f[MAXPER];
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
        const uint32_t r = k * THREADS + tid;
        if (r < n) { f[k] = (uint8_t)(rec_bucket(w[8 * r + 6]) & (FINE - 1)); rk[k] = (uint16_t)atomicAdd(&hist[f[k]], 1u); }
    }
    __syncthreads();
    if (tid == 0) { uint32_t a = 0; for (int q = 0; q < FINE; ++q) { start[q] = a; a += hist[q]; } }
    __syncthreads();
    if (tid < FINE) fine_off[(uint64_t)sb * (FINE + 1) + tid] = start[tid];
    if (tid == 0) fine_off[(uint64_t)sb * (FINE + 1) + FINE] = n;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
        const uint32_t r = k * THREADS + tid;
        if (r < n) idx[(uint64_t)sb * cap + start[f[k]] + rk[k]] = (uint16_t)r;
    }
}

// ---- the count kernel's read, by index list (GATHER) or contiguous: 512-record batches by LDS-DMA, the next batch in flight while
// this one is "consumed" (a few LDS reads per record -- the real kernel does ~250 wave instructions per k-mer here; what is
// measured is whether the fetch keeps up at all)
template <bool GATHER, int THREADS>
__global__ void __launch_bounds__(THREADS) gather_kernel(const uint4* __restrict__ rec, const uint32_t* __restrict__ cursor, uint64_t cap, const uint16_t* __restrict__ idx,
                                                         uint32_t n_sb, unsigned long long* out) {
    constexpr int BATCH = 512;
    __shared__ __attribute__((aligned(16))) uint4 buf[2][2 * BATCH];
    const int tid = threadIdx.x;
    unsigned long long acc = 0;
    for (uint32_t sb = blockIdx.x; sb < n_sb; sb += gridDim.x) {
        const uint32_t n = cursor[sb];
        const uint4* base = rec + 2 * (uint64_t)sb * cap;
        const uint16_t* ix = idx + (uint64_t)sb * cap;
        auto fetch = [&](uint32_t b0, int p) {
            for (int c = tid; c < 2 * BATCH; c += THREADS) {
                const uint32_t r = b0 + (c >> 1);
                if (r < n) {
                    const uint32_t s = GATHER ? ix[r] : r;
                    const uint32_t lds_at = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)(&buf[p][c & ~63]));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_at), "v"(base + 2 * (uint64_t)s + (c & 1)) : "memory");
                }
            }
        };
        int p = 0;
        fetch(0, 0);
        for (uint32_t b0 = 0; b0 < n; b0 += BATCH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (b0 + BATCH < n) fetch(b0 + BATCH, p ^ 1);
            for (int r = tid; r < BATCH; r += THREADS) if (b0 + r < n) acc += buf[p][2 * r].w + buf[p][2 * r + 1].x;
            p ^= 1;
            __syncthreads();
        }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) atomicAdd(out, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct timer {
    hipEvent_t a, b;
    timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    void start() { CHECK(hipEventRecord(a)); }
    float stop() { CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

static unsigned long long* d_sum;
static void check(const char* what, const uint4* rec, const uint32_t* cursor, uint32_t nbins, uint64_t cap, uint64_t n) {
    CHECK(hipMemset(d_sum, 0, 16));
    hipLaunchKernelGGL(sum_kernel, dim3(4096), dim3(256), 0, 0, rec, cursor, nbins, cap, d_sum);
    unsigned long long h[2];
    CHECK(hipMemcpy(h, d_sum, 16, hipMemcpyDeviceToHost));
    unsigned long long want = 0;
    // sum over i of (uint32_t)i
    const unsigned long long full = n >> 32, rem = n & 0xFFFFFFFFull;
    want = full * (0xFFFFFFFFull * 0x100000000ull / 2) + rem * (rem - 1) / 2;
    printf("    [%s] records %llu of %llu, index sum %s\n", what, h[1], (unsigned long long)n, h[0] == want ? "ok" : "WRONG");
}

template <int BINS, int TILE, int THREADS>
static float run_level(const char* name, const uint4* in, const uint32_t* in_cursor, uint32_t in_bins, uint64_t in_cap, uint64_t max_in, uint32_t shift,
                       uint32_t* cursor, uint64_t cap, uint4* out, uint32_t* d_ovf, uint64_t n, bool verify) {
    const size_t lds = (size_t)TILE * 32 + TILE * 4 + BINS * 12 + 64;
    CHECK(hipFuncSetAttribute((const void*)level_kernel<BINS, TILE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t tiles = (uint32_t)((max_in + TILE - 1) / TILE);
    timer t;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(cursor, 0, (size_t)in_bins * BINS * 4));
        CHECK(hipMemset(d_ovf, 0, 4));
        t.start();
        hipLaunchKernelGGL((level_kernel<BINS, TILE, THREADS>), dim3(in_bins * tiles), dim3(THREADS), lds, 0, in, in_cursor, in_cap, tiles, shift, cursor, cap, (uint32_t)BINS, out, d_ovf);
        const float ms = t.stop();
        if (ms < best) best = ms;
    }
    uint32_t ovf;
    CHECK(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost));
    printf("%-64s %7.2f ms  %6.2f TB/s (read + write of %.1f GB)  lds %zu B%s\n", name, best, 2.0 * n * 32 / best / 1e9, n * 32 / 1e9, lds, ovf ? "  OVERFLOW" : "");
    if (verify) check(name, out, cursor, in_bins * BINS, cap, n);
    return best;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? (uint64_t)atof(argv[1]) : 690000000ull;
    printf("records: %llu x 32 B = %.1f GB, %u fine buckets\n", (unsigned long long)n, n * 32 / 1e9, NB);
    uint4 *A, *B, *C;
    const uint64_t cap_fine = (uint64_t)((double)n / NB * 1.6 + 64);
    const uint64_t cap1 = (uint64_t)((double)n / 256 * 1.05 + 4096), cap2 = (uint64_t)((double)n / 65536 * 1.08 + 512);
    const size_t bufA = (size_t)n * 32 + 64, bufB = std::max<size_t>((size_t)NB * cap_fine, (size_t)256 * cap1) * 32 + 64, bufC = (size_t)65536 * cap2 * 32 + 64;
    CHECK(hipMalloc(&A, bufA)); CHECK(hipMalloc(&B, bufB)); CHECK(hipMalloc(&C, bufC));
    uint32_t *cur_fine, *cur1, *cur2, *one, *d_ovf;
    CHECK(hipMalloc(&cur_fine, (size_t)NB * 4)); CHECK(hipMalloc(&cur1, 1024 * 4)); CHECK(hipMalloc(&cur2, 65536 * 4 * 4)); CHECK(hipMalloc(&one, 4)); CHECK(hipMalloc(&d_ovf, 4));
    CHECK(hipMalloc(&d_sum, 16));
    hipLaunchKernelGGL(gen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, A, n);
    { uint32_t nn = (uint32_t)n; if (n >= (1ull << 32)) { printf("too many records\n"); return 2; } CHECK(hipMemcpy(one, &nn, 4, hipMemcpyHostToDevice)); }
    CHECK(hipDeviceSynchronize());
    timer t;
    // ---- today's emission
    for (int pair = 0; pair < 2; ++pair) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(cur_fine, 0, (size_t)NB * 4));
            t.start();
            if (pair) hipLaunchKernelGGL(direct_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, A, n, cur_fine, (uint32_t)cap_fine, B);
            else hipLaunchKernelGGL(direct_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, A, n, cur_fine, (uint32_t)cap_fine, B);
            const float ms = t.stop();
            if (ms < best) best = ms;
        }
        printf("%-64s %7.2f ms  %6.1f G records/s\n", pair ? "direct2: fine-bucket atomic + lane-paired 32-byte stores" : "direct: fine-bucket atomic + 2 x 16-byte scattered stores", best, n / best / 1e6);
        check("direct", B, cur_fine, NB, cap_fine, n);
    }
    // ---- level 1: everything -> 256 coarse bins (bucket bits 20..13)
    run_level<256, 1024, 256>("level1: 256 bins, tile 1024, 256 threads", A, one, 1, 0, n, 13, cur1, cap1, B, d_ovf, n, false);
    run_level<256, 2048, 256>("level1: 256 bins, tile 2048, 256 threads", A, one, 1, 0, n, 13, cur1, cap1, B, d_ovf, n, false);
    run_level<256, 2048, 512>("level1: 256 bins, tile 2048, 512 threads", A, one, 1, 0, n, 13, cur1, cap1, B, d_ovf, n, false);
    run_level<256, 4096, 512>("level1: 256 bins, tile 4096, 512 threads", A, one, 1, 0, n, 13, cur1, cap1, B, d_ovf, n, false);
    run_level<256, 4096, 1024>("level1: 256 bins, tile 4096, 1024 threads", A, one, 1, 0, n, 13, cur1, cap1, B, d_ovf, n, true);
    // ---- level 2: every coarse bin -> 256 super-buckets (bucket bits 12..5): B -> C
    std::vector<uint32_t> h1(256);
    CHECK(hipMemcpy(h1.data(), cur1, 1024, hipMemcpyDeviceToHost));
    uint64_t max1 = 0; for (uint32_t v : h1) max1 = v > max1 ? v : max1;
    run_level<256, 1024, 256>("level2: 256 x 256 super-buckets, tile 1024, 256 threads", B, cur1, 256, cap1, max1, 5, cur2, cap2, C, d_ovf, n, false);
    run_level<256, 2048, 512>("level2: 256 x 256 super-buckets, tile 2048, 512 threads", B, cur1, 256, cap1, max1, 5, cur2, cap2, C, d_ovf, n, false);
    run_level<256, 4096, 1024>("level2: 256 x 256 super-buckets, tile 4096, 1024 threads", B, cur1, 256, cap1, max1, 5, cur2, cap2, C, d_ovf, n, true);
    // ---- level 3: index lists
    uint16_t* idx; uint32_t* foff; unsigned long long* d_acc;
    CHECK(hipMalloc(&idx, (size_t)65536 * cap2 * 2)); CHECK(hipMalloc(&foff, (size_t)65536 * 33 * 4)); CHECK(hipMalloc(&d_acc, 8));
    if (cap2 > 24 * 512) printf("(super-buckets too large for the index kernel's registers: %llu)\n", (unsigned long long)cap2);
    else {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { t.start(); hipLaunchKernelGGL((index_kernel<32, 512>), dim3(65536), dim3(512), 0, 0, C, cur2, cap2, idx, foff); const float ms = t.stop(); if (ms < best) best = ms; }
        printf("%-64s %7.2f ms\n", "index: u16 list per super-bucket in fine order (reads word 6)", best);
        for (int g = 0; g < 2; ++g) {
            best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(d_acc, 0, 8));
                t.start();
                if (g) hipLaunchKernelGGL((gather_kernel<true, 768>), dim3(512 * 8), dim3(768), 0, 0, C, cur2, cap2, idx, 65536u, d_acc);
                else hipLaunchKernelGGL((gather_kernel<false, 768>), dim3(512 * 8), dim3(768), 0, 0, C, cur2, cap2, idx, 65536u, d_acc);
                const float ms = t.stop(); if (ms < best) best = ms;
            }
            unsigned long long acc; CHECK(hipMemcpy(&acc, d_acc, 8, hipMemcpyDeviceToHost));
            printf("%-64s %7.2f ms  %6.2f TB/s of records  (sum %llx)\n", g ? "gather: LDS-DMA by index list, 512-record batches" : "stream: LDS-DMA contiguous, 512-record batches", best, n * 32 / best / 1e9, acc);
        }
    }
    return 0;
}
