// snk_host.hip -- host-pointer convenience entry point, the .bv hand-off file (the graph-from-unitigs step lives in snk_hbv.hip).
//   snk_count_graph      : buildReadQGraph48 for host-resident reads (lib/assembly/src/paths/long/BuildReadQGraph48.h:24-34)
//   snk_write_bv/read_bv : lib/tada/src/debruijn.rs:895-929 <-> BuildReadQGraph48.cc:1640-1642
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <numeric>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"

// ---------------------------------------------------------------------------------------------------------------------
// Host seam.  The first version of snk_count_graph hipMalloc'ed every input, copied pageable memory synchronously, ordered
// the unitigs with std::sort on the host and repacked every key word by word: 0.9 Gk-mers/s at 5 M reads against 70+ on
// the device.  Now: device inputs and pinned staging buffers live in the context (grow-only, no hipMalloc in steady state);
// pageable host arrays go up in 64 MiB pieces through three pinned buffers filled by a few memcpy threads (pinned callers --
// snk_host_alloc_pinned -- are copied from in place); the quality rows, three quarters of the bytes and dead after the
// trim, never exist on the device in full: every piece is trimmed on the compute stream while the next is on the wire;
// the BVComp order (HBVFromEdges.cc:106-111) is one stable radix sort on the device (unitigs leave the join ordered by their
// first k-mer, which no two share), and the .bv file image is packed on the device.
namespace {

constexpr size_t STAGE_BYTES = 64ull << 20;
constexpr int STAGE_BUFS = 3;
constexpr int COPY_THREADS = 8;

struct host_io {
    int device = 0;
    hipStream_t copy = nullptr;
    void* pin[STAGE_BUFS] = {nullptr, nullptr, nullptr};
    hipEvent_t pin_free[STAGE_BUFS] = {nullptr, nullptr, nullptr};
    int next = 0;
    struct dbuf { void* p = nullptr; size_t cap = 0; };
    dbuf rows, ascii, lens, gl, bc, qchunk[2];
    hipEvent_t q_done[2] = {nullptr, nullptr}, q_up[2] = {nullptr, nullptr};
    ~host_io() {
        (void)hipSetDevice(device);
        for (auto& q : pin) if (q) (void)hipHostFree(q);
        for (auto& e : pin_free) if (e) (void)hipEventDestroy(e);
        for (auto& e : q_done) if (e) (void)hipEventDestroy(e);
        for (auto& e : q_up) if (e) (void)hipEventDestroy(e);
        for (dbuf* b : {&rows, &ascii, &lens, &gl, &bc, &qchunk[0], &qchunk[1]}) if (b->p) (void)hipFree(b->p);
        if (copy) (void)hipStreamDestroy(copy);
    }
};

int io_of(snk_ctx* ctx, host_io** out, char* err, size_t errcap) {
    if (!ctx->host_io) {
        // built in a local owner and published only when every resource exists: a failed hipHostMalloc (192 MiB of page-locked
        // memory) leaves the context without a half-made object (whose destructor frees what was made) and the next call retries
        std::unique_ptr<host_io> own(new host_io());
        host_io* io = own.get();
        io->device = ctx->device;
        SNK_HIP_TRY(hipStreamCreateWithFlags(&io->copy, hipStreamNonBlocking));
        for (int b = 0; b < STAGE_BUFS; ++b) {
            SNK_HIP_TRY(hipHostMalloc(&io->pin[b], STAGE_BYTES, hipHostMallocDefault));
            SNK_HIP_TRY(hipEventCreateWithFlags(&io->pin_free[b], hipEventDisableTiming));
        }
        for (int b = 0; b < 2; ++b) {
            SNK_HIP_TRY(hipEventCreateWithFlags(&io->q_done[b], hipEventDisableTiming));
            SNK_HIP_TRY(hipEventCreateWithFlags(&io->q_up[b], hipEventDisableTiming));
        }
        ctx->host_io = own.release();
        ctx->host_io_free = [](void* p) { delete static_cast<host_io*>(p); };
    }
    *out = static_cast<host_io*>(ctx->host_io);
    return SNK_OK;
}

int grow(host_io::dbuf* b, size_t bytes, char* err, size_t errcap) {
    bytes = std::max<size_t>(bytes, 256);
    if (b->cap >= bytes) return SNK_OK;
    if (b->p) { SNK_HIP_TRY(hipFree(b->p)); b->p = nullptr; b->cap = 0; }
    SNK_HIP_TRY(hipMalloc(&b->p, bytes + bytes / 8));
    b->cap = bytes + bytes / 8;
    return SNK_OK;
}

bool is_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

void parallel_copy(void* dst, const void* src, size_t n) {
    if (n < (4u << 20)) { memcpy(dst, src, n); return; }
    std::thread th[COPY_THREADS];
    const size_t per = (n / COPY_THREADS + 63) & ~(size_t)63;
    for (int t = 0; t < COPY_THREADS; ++t) {
        const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        th[t] = std::thread([=] { if (b > a) memcpy((char*)dst + a, (const char*)src + a, b - a); });
    }
    for (auto& t : th) t.join();
}

// host -> device on the copy stream; pageable sources go through the pinned ring
int upload(host_io* io, void* d_dst, const void* h_src, size_t bytes, char* err, size_t errcap) {
    if (!bytes) return SNK_OK;
    if (is_pinned(h_src)) { SNK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, io->copy)); return SNK_OK; }
    for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
        const size_t n = std::min(STAGE_BYTES, bytes - off);
        const int b = io->next;
        io->next = (io->next + 1) % STAGE_BUFS;
        SNK_HIP_TRY(hipEventSynchronize(io->pin_free[b]));          // its previous DMA has drained (a fresh event is complete)
        parallel_copy(io->pin[b], (const char*)h_src + off, n);
        SNK_HIP_TRY(hipMemcpyAsync((char*)d_dst + off, io->pin[b], n, hipMemcpyHostToDevice, io->copy));
        SNK_HIP_TRY(hipEventRecord(io->pin_free[b], io->copy));
    }
    return SNK_OK;
}

// ---- BVComp order + gathers on the device
__global__ void __launch_bounds__(256) bv_lenkey_kernel(const uint64_t* __restrict__ off, uint64_t U, uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    key[u] = ~(off[u + 1] - off[u]);            // ascending ~len = descending length; the sort is stable, ties keep first-k-mer order
    idx[u] = (uint32_t)u;
}
// sizes of the ordered unitigs: bases (plain copy) or .bv bytes (u32 length + ceil(len/4))
__global__ void __launch_bounds__(256) bv_sizes_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ order, uint64_t U, int image,
                                                       uint64_t* __restrict__ sz) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > U) return;
    uint64_t v = 0;
    if (r < U) { const uint64_t len = off[order[r] + 1] - off[order[r]]; v = image ? 4 + (len + 3) / 4 : len; }
    sz[r] = v;
}
// One thread per FOUR output bytes (a byte per lane and store made this kernel instruction-bound: 0.98 ms for the bench graph's 67 MB
// image, 350 GB/s): the owner of the group is found by binary search over the offsets; a group that lies inside one unitig's packed
// bases -- all but the few at a record's head or tail -- is 16 source bytes (four unaligned dword loads) and one dword store.
__device__ __forceinline__ uint8_t bv_byte(const uint64_t* __restrict__ off, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ order,
                                           const uint64_t* __restrict__ noff, uint64_t lo, uint64_t p, int image) {
    const uint64_t src = off[order[lo]], len = off[order[lo] + 1] - src, q = p - noff[lo];
    if (!image) return bases[src + q];
    if (q < 4) return (uint8_t)((uint32_t)len >> (8 * q));
    const uint64_t j = (q - 4) * 4;
    uint32_t v = 0;
    for (uint32_t t = 0; t < 4 && j + t < len; ++t) v |= (uint32_t)(bases[src + j + t] & 3u) << (2 * t);
    return (uint8_t)v;
}
__global__ void __launch_bounds__(256) bv_gather_kernel(const uint64_t* __restrict__ off, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ order,
                                                        const uint64_t* __restrict__ noff, uint64_t U, uint64_t total, int image, uint8_t* __restrict__ out) {
    const uint64_t p = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= total) return;
    uint64_t lo = 0, hi = U;                     // largest r with noff[r] <= p
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (noff[mid] <= p) lo = mid; else hi = mid; }
    struct __attribute__((packed)) u32_any { uint32_t v; };
    const uint64_t nend = noff[lo + 1];
    const uint64_t src = off[order[lo]], len = off[order[lo] + 1] - src, q = p - noff[lo];
    if (p + 4 <= total && p + 4 <= nend && (((uintptr_t)(out + p)) & 3u) == 0) {
        if (!image) { *reinterpret_cast<uint32_t*>(out + p) = reinterpret_cast<const u32_any*>(bases + src + q)->v; return; }
        const uint64_t j = (q - 4) * 4;
        if (q >= 4 && j + 16 <= len) {
            uint32_t v = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t x = reinterpret_cast<const u32_any*>(bases + src + j + 4 * t)->v & 0x03030303u;     // four bases, one per byte
                v |= ((x | (x >> 6) | (x >> 12) | (x >> 18)) & 0xFFu) << (8 * t);
            }
            *reinterpret_cast<uint32_t*>(out + p) = v;
            return;
        }
    }
    // a record's length word, its last bytes, a group across two records, the image's tail: byte by byte
    for (uint64_t pp = p; pp < p + 4 && pp < total; ++pp) {
        while (pp >= noff[lo + 1]) ++lo;
        out[pp] = bv_byte(off, bases, order, noff, lo, pp, image);
    }
}
__global__ void __launch_bounds__(256) key_words_kernel(const snk_kmer* __restrict__ keys, uint64_t n, uint4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t lo = reinterpret_cast<const uint64_t*>(keys)[2 * i], hi = reinterpret_cast<const uint64_t*>(keys)[2 * i + 1];
    out[i] = make_uint4((uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(lo >> 32), (uint32_t)lo);
}

template <typename T>
int arena(snk_ctx* ctx, size_t n, T** out, char* err, size_t errcap) {
    void* q = nullptr;
    int rc = snk_ctx_alloc(ctx, std::max<size_t>(n * sizeof(T), 16), &q, err, errcap);
    *out = (T*)q;
    return rc;
}

// unitigs resident on the device -> the reference's deterministic order (BVComp, HBVFromEdges.cc:106-111: length descending,
// then lexicographic), gathered on the device and downloaded: plain bases + offsets, or the bytes of the .bv hand-off file.
// by_first_kmer: the input is already ordered by its first K bases (how the join leaves it): one stable sort by length does;
// else (the union of several ranks' sets) the first k-mers are sorted first -- two unitigs never share theirs.
__global__ void __launch_bounds__(256) bv_headkey_kernel(const uint64_t* __restrict__ off, const uint8_t* __restrict__ bases, uint64_t U, uint32_t K,
                                                         snk_u128* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const uint8_t* b = bases + off[u];
    const uint64_t len = off[u + 1] - off[u];
    snk_u128 k = 0;
    for (uint32_t j = 0; j < K; ++j) k = (k << 2) | (snk_u128)(j < len ? (b[j] & 3u) : 0u);
    key[u] = k;
    idx[u] = (uint32_t)u;
}
__global__ void __launch_bounds__(256) bv_lenkey_of_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ idx1, uint64_t U, uint64_t* __restrict__ key) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < U) key[i] = ~(off[idx1[i] + 1] - off[idx1[i]]);
}
// device part: BVComp order (HBVFromEdges.cc:106-111), sizes, offsets and the gathered bytes -- plain bases, or the .bv payload
// (u32 length + ceil(len/4) bytes per unitig; header_bytes are left free in front of it for the file header)
int unitigs_bv_device(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t U, const uint64_t* d_off, const uint8_t* d_bases, bool by_first_kmer, bool want_image,
                      uint32_t header_bytes, uint8_t** d_out_p, uint64_t** d_noff_p, uint64_t* total_p, char* err, size_t errcap) {
    int rc;
    uint64_t *key, *key2, *sz, *noff;
    uint32_t *idx, *order;
    if ((rc = arena(ctx, U + 1, &key, err, errcap)) || (rc = arena(ctx, U + 1, &key2, err, errcap)) || (rc = arena(ctx, U + 1, &idx, err, errcap)) ||
        (rc = arena(ctx, U + 1, &order, err, errcap)) || (rc = arena(ctx, U + 2, &sz, err, errcap)) || (rc = arena(ctx, U + 2, &noff, err, errcap)))
        return rc;
    uint64_t total = 0;
    if (U) {
        size_t tb = 0, tb2 = 0, tb3 = 0;
        snk_u128 *hk = nullptr, *hk2 = nullptr;
        uint32_t* idx1 = nullptr;
        SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, key, key2, idx, order, (size_t)U, 0u, 64u, st));
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb2, sz, noff, (uint64_t)0, (size_t)(U + 1), rocprim::plus<uint64_t>(), st));
        if (!by_first_kmer) {
            if ((rc = arena(ctx, U + 1, &hk, err, errcap)) || (rc = arena(ctx, U + 1, &hk2, err, errcap)) || (rc = arena(ctx, U + 1, &idx1, err, errcap))) return rc;
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb3, hk, hk2, idx, idx1, (size_t)U, 0u, 128u, st));
        }
        uint8_t* tmp;
        if ((rc = arena(ctx, std::max(tb, std::max(tb2, tb3)), &tmp, err, errcap))) return rc;
        if (by_first_kmer) hipLaunchKernelGGL(bv_lenkey_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, d_off, U, key, idx);
        else {
            hipLaunchKernelGGL(bv_headkey_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, d_off, d_bases, U, K, hk, idx);
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb3, hk, hk2, idx, idx1, (size_t)U, 0u, 128u, st));
            hipLaunchKernelGGL(bv_lenkey_of_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, d_off, idx1, U, key);
            idx = idx1;
        }
        SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, key, key2, idx, order, (size_t)U, 0u, 64u, st));
        hipLaunchKernelGGL(bv_sizes_kernel, dim3((unsigned)((U + 256) / 256)), dim3(256), 0, st, d_off, order, U, want_image ? 1 : 0, sz);
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb2, sz, noff, (uint64_t)0, (size_t)(U + 1), rocprim::plus<uint64_t>(), st));
        SNK_HIP_TRY(hipMemcpyAsync(&total, noff + U, 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
    }
    uint8_t* d_out;
    if ((rc = arena(ctx, header_bytes + total + 16, &d_out, err, errcap))) return rc;
    if (total) hipLaunchKernelGGL(bv_gather_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, st, d_off, d_bases, order, noff, U, total,
                                  want_image ? 1 : 0, d_out + header_bytes);
    SNK_HIP_TRY(hipGetLastError());
    *d_out_p = d_out;
    *d_noff_p = noff;
    *total_p = total;
    return SNK_OK;
}

int unitigs_to_host(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t U, const uint64_t* d_off, const uint8_t* d_bases, bool by_first_kmer, bool want_image,
                    snk_result* out, char* err, size_t errcap) {
    int rc;
    out->n_unitigs = U;
    uint8_t* d_out = nullptr;
    uint64_t* noff = nullptr;
    uint64_t total = 0;
    if ((rc = unitigs_bv_device(ctx, st, K, U, d_off, d_bases, by_first_kmer, want_image, 0, &d_out, &noff, &total, err, errcap))) return rc;
    if (want_image) {
        out->bv_bytes = 16 + total;
        out->bv_image = (uint8_t*)malloc(out->bv_bytes);
        if (!out->bv_image) { snk_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "host allocation failed"); }
        memcpy(out->bv_image, "BINWRITE", 8);
        memcpy(out->bv_image + 8, &U, 8);
        if (total) SNK_HIP_TRY(hipMemcpyAsync(out->bv_image + 16, d_out, total, hipMemcpyDeviceToHost, st));
    } else {
        out->unitig_off = (uint64_t*)malloc((U + 1) * 8);
        out->unitig_bases = (uint8_t*)malloc(std::max<size_t>(total, 16));
        if (!out->unitig_off || !out->unitig_bases) { snk_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "host allocation failed"); }
        out->unitig_off[0] = 0;
        if (U) SNK_HIP_TRY(hipMemcpyAsync(out->unitig_off, noff, (U + 1) * 8, hipMemcpyDeviceToHost, st));
        if (total) SNK_HIP_TRY(hipMemcpyAsync(out->unitig_bases, d_out, total, hipMemcpyDeviceToHost, st));
    }
    return SNK_OK;
}

__global__ void bv_header_kernel(uint8_t* out, uint64_t U) {
    const char m[8] = {'B', 'I', 'N', 'W', 'R', 'I', 'T', 'E'};
    if (threadIdx.x < 8) out[threadIdx.x] = (uint8_t)m[threadIdx.x];
    else if (threadIdx.x < 16) out[threadIdx.x] = (uint8_t)(U >> (8 * (threadIdx.x - 8)));
}

int count_graph_impl(snk_ctx* ctx, const snk_reads* in, const snk_params* p, snk_result* out, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: NULL argument");
    if (!in->ascii && !in->rows) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: need ascii or rows");
    if (!in->quals && !in->good_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: need quals or good_len");
    if (in->read_len == 0 || in->read_len > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: read_len must be 1..256");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = ctx->stream;
    host_io* io = nullptr;
    int rc = io_of(ctx, &io, err, errcap);
    if (rc) return rc;
    const uint64_t n = in->n_reads;
    const uint32_t L = in->read_len, rw = (L + 15) / 16;
    hipEvent_t up_done;
    SNK_HIP_TRY(hipEventCreateWithFlags(&up_done, hipEventDisableTiming));
    struct ev_guard { hipEvent_t e; ~ev_guard() { (void)hipEventDestroy(e); } } eg{up_done};
    // ---- uploads (copy stream), the quality rows piece by piece with the trim behind each piece (compute stream)
    if ((rc = grow(&io->rows, n * rw * 4, err, errcap))) return rc;
    if (in->rows) { if ((rc = upload(io, io->rows.p, in->rows, n * rw * 4, err, errcap))) return rc; }
    else {
        if ((rc = grow(&io->ascii, n * L, err, errcap))) return rc;
        if ((rc = upload(io, io->ascii.p, in->ascii, n * L, err, errcap))) return rc;
    }
    if (in->lens) { if ((rc = grow(&io->lens, n * 2, err, errcap)) || (rc = upload(io, io->lens.p, in->lens, n * 2, err, errcap))) return rc; }
    if (in->bc) { if ((rc = grow(&io->bc, n * 4, err, errcap)) || (rc = upload(io, io->bc.p, in->bc, n * 4, err, errcap))) return rc; }
    if ((rc = grow(&io->gl, n * 2 + 16, err, errcap))) return rc;
    if (in->good_len) { if ((rc = upload(io, io->gl.p, in->good_len, n * 2, err, errcap))) return rc; }
    SNK_HIP_TRY(hipEventRecord(up_done, io->copy));
    if (!in->good_len) {
        SNK_HIP_TRY(hipStreamWaitEvent(st, up_done, 0));            // the trim reads the lengths
        const uint64_t rows_per = std::max<uint64_t>(1, (256ull << 20) / L) & ~255ull ? (std::max<uint64_t>(1, (256ull << 20) / L) & ~255ull) : 256;
        for (int b = 0; b < 2; ++b) if ((rc = grow(&io->qchunk[b], std::min<uint64_t>(rows_per, std::max<uint64_t>(n, 1)) * L, err, errcap))) return rc;
        int b = 0;
        bool used[2] = {false, false};
        for (uint64_t r0 = 0; r0 < n; r0 += rows_per, b ^= 1) {
            const uint64_t nr = std::min(rows_per, n - r0);
            if (used[b]) SNK_HIP_TRY(hipStreamWaitEvent(io->copy, io->q_done[b], 0));      // the trim of the piece before last has read this buffer
            if ((rc = upload(io, io->qchunk[b].p, in->quals + r0 * L, nr * L, err, errcap))) return rc;
            SNK_HIP_TRY(hipEventRecord(io->q_up[b], io->copy));
            SNK_HIP_TRY(hipStreamWaitEvent(st, io->q_up[b], 0));
            rc = snk_dev_trim(ctx, io->qchunk[b].p, L, in->lens ? (const char*)io->lens.p + r0 * 2 : nullptr, L, nr, p->K, p->min_qual,
                              (char*)io->gl.p + r0 * 2, st);
            if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
            SNK_HIP_TRY(hipEventRecord(io->q_done[b], st));
            used[b] = true;
        }
    } else SNK_HIP_TRY(hipStreamWaitEvent(st, up_done, 0));
    if (!in->rows) {
        rc = snk_dev_pack_ascii(ctx, io->ascii.p, L, L, n, io->rows.p, rw, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
    }
    snk_dev_reads dr;
    memset(&dr, 0, sizeof dr);
    dr.n_reads = n; dr.rows = io->rows.p; dr.row_words = rw; dr.read_len = L; dr.lens = in->lens ? io->lens.p : nullptr;
    dr.good_len = io->gl.p; dr.bc = in->bc ? io->bc.p : nullptr; dr.ign_bc_below = in->ign_bc_below;
    snk_dev_result r;
    snk_params pp = *p;
    const bool want_table = !(p->flags & SNK_F_NO_TABLE), want_image = (p->flags & SNK_F_BV_IMAGE) != 0;
    pp.flags &= ~(uint32_t)(SNK_F_NO_TABLE | SNK_F_BV_IMAGE);
    if (!want_table) pp.flags |= SNK_F_UNSORTED_TABLE;          // nobody will look at the table: do not sort it
    if ((rc = snk_dev_count_graph(ctx, &dr, &pp, &r, st, err, errcap))) return rc;
    out->n_instances = r.n_instances;
    out->n_kmers = r.n_kmers;
    out->spectrum_bins = r.spectrum_bins;
    memcpy(out->phase_ms, r.phase_ms, sizeof out->phase_ms);
    const uint64_t nk = want_table ? r.n_kmers : 0;
    if (want_table) {
        out->kmers = (uint32_t*)malloc(std::max<size_t>(nk * 16, 16));
        out->counts = (uint32_t*)malloc(std::max<size_t>(nk * 4, 16));
        out->ctx = (uint8_t*)malloc(std::max<size_t>(nk, 16));
    }
    out->spectrum = (uint64_t*)malloc(std::max<size_t>((size_t)r.spectrum_bins * 8, 16));
    if ((want_table && (!out->kmers || !out->counts || !out->ctx)) || !out->spectrum) { snk_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_count_graph: host allocation failed"); }
    if (nk) {
        uint4* kw;
        if ((rc = arena(ctx, nk, &kw, err, errcap))) return rc;
        hipLaunchKernelGGL(key_words_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, (const snk_kmer*)r.keys, nk, kw);
        SNK_HIP_TRY(hipMemcpyAsync(out->kmers, kw, nk * 16, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->counts, r.counts, nk * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->ctx, r.ctx, nk, hipMemcpyDeviceToHost, st));
    }
    if (r.spectrum_bins) SNK_HIP_TRY(hipMemcpyAsync(out->spectrum, r.spectrum, (size_t)r.spectrum_bins * 8, hipMemcpyDeviceToHost, st));
    if ((rc = unitigs_to_host(ctx, st, p->K, r.n_unitigs, (const uint64_t*)r.unitig_off, (const uint8_t*)r.unitig_bases, true, want_image, out, err, errcap))) return rc;
    SNK_HIP_TRY(snk_sync(st));
    return SNK_OK;
}

}  // namespace

// (library-internal: the gather of the sharded path ends here, snk_shard_step.hip)
int snk_unitigs_to_host(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t U, const uint64_t* d_off, const uint8_t* d_bases, bool by_first_kmer,
                        bool want_image, snk_result* out, char* err, size_t errcap) {
    int rc = unitigs_to_host(ctx, st, K, U, d_off, d_bases, by_first_kmer, want_image, out, err, errcap);
    if (rc) return rc;
    SNK_HIP_TRY(snk_sync(st));
    return SNK_OK;
}

// No C++ exception leaves an extern "C" entry point: a failed host allocation maps to SNK_E_NOMEM (the caller's exit code 99,
// system/RunTime.cc:195-221), anything else to SNK_E_INTERNAL.
#define SNK_GUARD(body)                                                                                       \
    try { body } catch (const std::bad_alloc&) { return snk_fail(SNK_E_NOMEM, err, errcap, "host allocation failed"); } \
    catch (const std::exception& ex) { return snk_fail(SNK_E_INTERNAL, err, errcap, "%s", ex.what()); }               \
    catch (...) { return snk_fail(SNK_E_INTERNAL, err, errcap, "unexpected exception"); }

extern "C" int snk_count_graph(snk_ctx* ctx, const snk_reads* in, const snk_params* p, snk_result* out, char* err, size_t errcap) {
    if (ctx) snk_opts_enter(&ctx->opts);
    // every error exit of the implementation leaves through here: uploads / kernels it queued are waited for (the caller may free
    // its input right after, and the next call reuses the context's staging buffers) and what it malloc'ed into *out is released
    int rc;
    if (out) memset(out, 0, sizeof *out);      // before any validation return: the error exit below frees what *out points at
    try { rc = count_graph_impl(ctx, in, p, out, err, errcap); }
    catch (const std::bad_alloc&) { rc = snk_fail(SNK_E_NOMEM, err, errcap, "host allocation failed"); }
    catch (const std::exception& ex) { rc = snk_fail(SNK_E_INTERNAL, err, errcap, "%s", ex.what()); }
    catch (...) { rc = snk_fail(SNK_E_INTERNAL, err, errcap, "unexpected exception"); }
    if (rc && ctx) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->host_io) (void)hipStreamSynchronize(static_cast<host_io*>(ctx->host_io)->copy);
        if (out) snk_free(out);
    }
    return rc;
}

extern "C" int snk_dev_bv_image(snk_ctx* ctx, uint32_t K, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases, int by_first_kmer,
                                const void** d_image, uint64_t* image_bytes, void* stream, char* err, size_t errcap) {
    if (!ctx || !d_image || !image_bytes || (n_unitigs && (!d_unitig_off || !d_unitig_bases))) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_bv_image: NULL argument");
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    SNK_GUARD(
        uint8_t* d_out = nullptr;
        uint64_t* noff = nullptr;
        uint64_t total = 0;
        int rc = unitigs_bv_device(ctx, st, K, n_unitigs, (const uint64_t*)d_unitig_off, (const uint8_t*)d_unitig_bases, by_first_kmer != 0, true, 16, &d_out, &noff, &total,
                                   err, errcap);
        if (rc) return rc;
        hipLaunchKernelGGL(bv_header_kernel, dim3(1), dim3(64), 0, st, d_out, n_unitigs);
        SNK_HIP_TRY(hipGetLastError());
        *d_image = d_out;
        *image_bytes = 16 + total;
        return SNK_OK;
    )
}

extern "C" int snk_host_alloc_pinned(size_t bytes, void** out, char* err, size_t errcap) {
    if (!out) return snk_fail(SNK_E_ARG, err, errcap, "snk_host_alloc_pinned: NULL argument");
    *out = nullptr;
    SNK_HIP_TRY(hipHostMalloc(out, std::max<size_t>(bytes, 16), hipHostMallocDefault));
    return SNK_OK;
}
extern "C" void snk_host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

extern "C" void snk_free(snk_result* r) {
    if (!r) return;
    free(r->kmers); free(r->counts); free(r->ctx); free(r->unitig_off); free(r->unitig_bases); free(r->spectrum); free(r->bv_image);
    memset(r, 0, sizeof *r);
}

static int write_bv_impl(const char* path, uint64_t n_unitigs, const uint64_t* off, const uint8_t* bases, char* err, size_t errcap) {
    FILE* f = fopen(path, "wb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_write_bv: cannot open %s", path);
    bool ok = fwrite("BINWRITE", 1, 8, f) == 8 && fwrite(&n_unitigs, 8, 1, f) == 1;
    std::vector<uint8_t> buf;
    for (uint64_t u = 0; ok && u < n_unitigs; ++u) {
        const uint64_t len64 = off[u + 1] - off[u];
        if (len64 > 0xFFFFFFFFull) { fclose(f); return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_write_bv: unitig longer than 2^32 bases"); }
        const uint32_t len = (uint32_t)len64;
        const uint8_t* b = bases + off[u];
        buf.assign((len + 3) / 4, 0);
        for (uint32_t j = 0; j < len; ++j) buf[j >> 2] |= (uint8_t)((b[j] & 3u) << (2 * (j & 3)));
        ok = fwrite(&len, 4, 1, f) == 1 && (buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size());
    }
    if (fclose(f) != 0) ok = false;
    return ok ? SNK_OK : snk_fail(SNK_E_IO, err, errcap, "snk_write_bv: short write to %s", path);
}
extern "C" int snk_write_bv(const char* path, uint64_t n_unitigs, const uint64_t* off, const uint8_t* bases, char* err, size_t errcap) {
    SNK_GUARD(return write_bv_impl(path, n_unitigs, off, bases, err, errcap);)
}

// The header is untrusted: the unitig count and every length are checked against the bytes that are really there before
// anything is sized by them (a corrupt .bv must come back as SNK_E_IO, not as std::length_error through the C ABI).
static int read_bv_impl(const char* path, uint64_t* n_unitigs, uint64_t** off_out, uint8_t** bases_out, char* err, size_t errcap) {
    FILE* f = fopen(path, "rb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: cannot open %s", path);
    struct closer { FILE* f; ~closer() { fclose(f); } } cl{f};
    if (fseek(f, 0, SEEK_END) != 0) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: cannot seek in %s", path);
    const long fsz = ftell(f);
    rewind(f);
    char magic[8];
    uint64_t n = 0;
    if (fsz < 16 || fread(magic, 1, 8, f) != 8 || memcmp(magic, "BINWRITE", 8) || fread(&n, 8, 1, f) != 1)
        return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: %s is not a BINWRITE file", path);
    if (n > ((uint64_t)fsz - 16) / 4) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: %s claims %llu unitigs in %ld bytes", path, (unsigned long long)n, fsz);
    std::vector<uint8_t> raw((size_t)fsz - 16);
    if (!raw.empty() && fread(raw.data(), 1, raw.size(), f) != raw.size()) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: truncated file");
    // pass 1: lengths and offsets
    uint64_t* off = (uint64_t*)malloc((n + 1) * 8);
    if (!off) return snk_fail(SNK_E_NOMEM, err, errcap, "snk_read_bv: host allocation failed");
    size_t p = 0;
    uint64_t tot = 0;
    for (uint64_t u = 0; u < n; ++u) {
        uint32_t len;
        if (p + 4 > raw.size()) { free(off); return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: truncated file"); }
        memcpy(&len, raw.data() + p, 4);
        p += 4;
        const size_t nb = ((size_t)len + 3) / 4;
        if (nb > raw.size() - p) { free(off); return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: truncated file"); }
        off[u] = tot;
        tot += len;
        p += nb;
    }
    off[n] = tot;
    uint8_t* bases = (uint8_t*)malloc(std::max<size_t>(tot, 16));
    if (!bases) { free(off); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_read_bv: host allocation failed"); }
    // pass 2: four bases per byte
    p = 0;
    for (uint64_t u = 0; u < n; ++u) {
        const uint64_t len = off[u + 1] - off[u];
        const uint8_t* src = raw.data() + p + 4;
        uint8_t* dst = bases + off[u];
        for (uint64_t j = 0; j + 4 <= len; j += 4) { const uint8_t b = src[j >> 2]; dst[j] = b & 3u; dst[j + 1] = (b >> 2) & 3u; dst[j + 2] = (b >> 4) & 3u; dst[j + 3] = b >> 6; }
        for (uint64_t j = len & ~3ull; j < len; ++j) dst[j] = (uint8_t)((src[j >> 2] >> (2 * (j & 3))) & 3u);
        p += 4 + (len + 3) / 4;
    }
    *n_unitigs = n;
    *off_out = off;
    *bases_out = bases;
    return SNK_OK;
}
extern "C" int snk_read_bv(const char* path, uint64_t* n_unitigs, uint64_t** off_out, uint8_t** bases_out, char* err, size_t errcap) {
    if (!path || !n_unitigs || !off_out || !bases_out) return snk_fail(SNK_E_ARG, err, errcap, "snk_read_bv: NULL argument");
    SNK_GUARD(return read_bv_impl(path, n_unitigs, off_out, bases_out, err, errcap);)
}
