// snk_trim.hip -- K1 quality trim and K2 ASCII -> 2-bit packing.
//
// K1 replaces GoodLenTailFinder (lib/assembly/src/paths/long/BuildReadQGraph48.cc:65-89) ==
//    find_trim_len (lib/tada/src/cmd_msp.rs:129-146): scanning from the 3' end, the first run of K
//    consecutive quals >= min_qual ends the scan; good length = run start + K, else 0.
// K2 replaces base_to_bits (lib/tada/src/kmer/mod.rs:311-319) / the N->A rule of
//    10X/ParseBarcodedFastqs.cc:87-88.
//
// Both are HBM-streaming byte kernels.  K1 stops at the first run (a clean read is decided by its last K bytes).
#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_stages.h"

// the scan of one quality row (global or LDS memory); rows whose base is 4-byte aligned are read in words
__device__ __forceinline__ int snk_trim_row(const uint8_t* q, int len, bool row_aligned, uint32_t K, uint32_t min_qual) {
    uint32_t good = 0;
    int i = len;
    if (row_aligned) {
        while (i > 0 && (i & 3)) {
            --i;
            if (q[i] < min_qual) good = 0;
            else if (++good == K) return i + (int)K;
        }
        while (i >= 4) {
            uint32_t w = *reinterpret_cast<const uint32_t*>(q + i - 4);
            int found = -1;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                uint32_t v = (w >> (8 * b)) & 0xFFu;
                if (found < 0) {
                    if (v < min_qual) good = 0;
                    else if (++good == K) found = i - 4 + b + (int)K;
                }
            }
            if (found >= 0) return found;
            i -= 4;
        }
    }
    while (i > 0) {
        --i;
        if (q[i] < min_qual) good = 0;
        else if (++good == K) return i + (int)K;
    }
    return 0;
}

__global__ void __launch_bounds__(256) snk_trim_kernel(const uint8_t* __restrict__ quals, uint32_t qstride,
                                                       const uint16_t* __restrict__ lens, uint32_t read_len,
                                                       uint64_t n_reads, uint32_t K, uint32_t min_qual,
                                                       uint16_t* __restrict__ good_len) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t* q = quals + r * (uint64_t)qstride;
    int len = lens ? lens[r] : (int)read_len;
    if (len > (int)read_len) len = (int)read_len;      // a length beyond the row (bad caller data, mismatched fastb/qualp pair) must not walk off it
    const bool row_aligned = ((qstride & 3u) == 0) && ((((uintptr_t)quals) & 3u) == 0);
    good_len[r] = (uint16_t)snk_trim_row(q, len, row_aligned, K, min_qual);
}

// Tiled form for short rows (qstride % 4 == 0, 256 rows fit in LDS): the 256 rows of a workgroup are contiguous
// in memory, so they are streamed into LDS with 16-byte loads (every fetched byte of every cache line is used; a
// lane that walks its own row in global memory keeps 64 different lines in flight per load and stalls on the
// miss queue), then each lane scans its row out of LDS.
__global__ void __launch_bounds__(256) snk_trim_tile_kernel(const uint8_t* __restrict__ quals, uint32_t qstride,
                                                            const uint16_t* __restrict__ lens, uint32_t read_len,
                                                            uint64_t n_reads, uint32_t K, uint32_t min_qual,
                                                            uint16_t* __restrict__ good_len) {
    extern __shared__ uint4 tile4[];
    uint8_t* tile = reinterpret_cast<uint8_t*>(tile4);
    const uint64_t r0 = (uint64_t)blockIdx.x * 256;
    const uint64_t rows_here = n_reads - r0 < 256 ? n_reads - r0 : 256;
    const uint32_t bytes = (uint32_t)rows_here * qstride;
    const uint8_t* src = quals + r0 * qstride;             // 16-byte aligned: 256 * qstride is a multiple of 16
    // all loads of a thread are issued before the first LDS store (up to 10 x 16 bytes in flight per lane)
    constexpr int NV = 10;                                  // 256 rows x 160 bytes / (256 threads x 16 bytes)
    uint4 v[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const uint32_t o = (threadIdx.x + q * 256) * 16;
        v[q] = make_uint4(0, 0, 0, 0);
        if (o + 16 <= bytes) v[q] = *reinterpret_cast<const uint4*>(src + o);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const uint32_t o = (threadIdx.x + q * 256) * 16;
        if (o + 16 <= bytes) *reinterpret_cast<uint4*>(tile + o) = v[q];
        else if (o < bytes) for (uint32_t j = o; j < bytes; ++j) tile[j] = src[j];
    }
    __syncthreads();
    if (threadIdx.x >= rows_here) return;
    const uint64_t r = r0 + threadIdx.x;
    int len = lens ? lens[r] : (int)read_len;
    if (len > (int)read_len) len = (int)read_len;      // a length beyond the row (bad caller data, mismatched fastb/qualp pair) must not walk off it
    good_len[r] = (uint16_t)snk_trim_row(tile + threadIdx.x * qstride, len, true, K, min_qual);
}

extern "C" int snk_dev_trim(snk_ctx* ctx, const void* d_quals, uint32_t qstride, const void* d_lens, uint32_t read_len,
                            uint64_t n_reads, uint32_t K, uint32_t min_qual, void* d_good_len, void* stream) {
    if (ctx) snk_opts_enter(&ctx->opts);
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx || !d_quals || !d_good_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_trim: NULL argument");
    if (read_len > 65535 || qstride < read_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_trim: bad read_len/qstride");
    if (n_reads == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    uint64_t nb = (n_reads + 255) / 256;
    const bool tiled = (qstride & 3u) == 0 && qstride <= 160 && (((uintptr_t)d_quals) & 15u) == 0 && !snk_opt_u32("trim_rowwise", 0);
    if (tiled)
        hipLaunchKernelGGL(snk_trim_tile_kernel, dim3((unsigned)nb), dim3(256), 256 * qstride, st, (const uint8_t*)d_quals, qstride,
                           (const uint16_t*)d_lens, read_len, n_reads, K, min_qual, (uint16_t*)d_good_len);
    else
        hipLaunchKernelGGL(snk_trim_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const uint8_t*)d_quals, qstride,
                           (const uint16_t*)d_lens, read_len, n_reads, K, min_qual, (uint16_t*)d_good_len);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// one thread per output word (16 bases)
__global__ void __launch_bounds__(256) snk_pack_kernel(const uint8_t* __restrict__ ascii, uint32_t astride,
                                                       uint32_t read_len, uint64_t n_reads, uint32_t* __restrict__ rows,
                                                       uint32_t row_words) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = n_reads * row_words;
    if (t >= total) return;
    uint64_t r = t / row_words;
    uint32_t w = (uint32_t)(t - r * row_words);
    const uint8_t* a = ascii + r * (uint64_t)astride;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
        uint32_t i = w * 16 + j;
        uint32_t code = 0;
        if (i < read_len) {
            uint8_t c = a[i];
            code = (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T') ? 3u : 0u;
        }
        v = (v << 2) | code;
    }
    rows[t] = v;
}

extern "C" int snk_dev_pack_ascii(snk_ctx* ctx, const void* d_ascii, uint32_t astride, uint32_t read_len,
                                  uint64_t n_reads, void* d_rows, uint32_t row_words, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx || !d_ascii || !d_rows) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_pack_ascii: NULL argument");
    if (row_words * 16 < read_len || astride < read_len)
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_pack_ascii: stride too small");
    if (n_reads == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    uint64_t total = n_reads * row_words;
    uint64_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(snk_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const uint8_t*)d_ascii, astride, read_len,
                       n_reads, (uint32_t*)d_rows, row_words);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
