// snk_dups.hip -- SURVEY f4: duplicate marking over the read paths, on the device.
//
// What it replaces: MarkDups, lib/assembly/src/10X/SecretOps.cc:413-593 (called right after the pathing, 10X/DF.cc:597-600;
// result written as a.dup).  The reference builds one record (first edge, offset on it, first five bases of the MATE, read id)
// per placed read, comparison-sorts the records, and walks the sorted array twice on one thread: groups of records that agree
// in the first three fields are duplicates of each other; the copy with the largest sum of base qualities (both mates; the
// earliest read on a tie) survives, every other member marks its PAIR.  Two statistics come with it: the share of duplicate
// reads whose group spans more than one barcode, and the pairs that are "artifactual" duplicates (base by base and quality by
// quality identical to an earlier member of a group with a tie).
//
// Here: the records never exist as such -- the three key fields are 74 bits (edge 32, offset 32 with the sign flipped, mate
// head 10), sorted by two stable LSD radix passes over (head) and (edge, offset) that carry the read id (ids start in order,
// so the id is the last key for free); group heads are found by comparing neighbours; one thread per group head then does
// exactly what the reference's loops do to its group (groups are short: a few reads), with the quality sums of the members
// of multi-read groups taken by one thread per member beforehand.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "snk_ctx.h"
#include "snk_stages.h"
#include "snk_common.h"


namespace {

template <typename T>
int dev(snk_ctx* ctx, size_t n, T** out, char* err, size_t errcap) {
    void* q = nullptr;
    int rc = snk_ctx_alloc(ctx, (n ? n : 1) * sizeof(T) + 16, &q, err, errcap);
    *out = (T*)q;
    return rc;
}

struct dup_in {
    const uint32_t* rows; uint32_t row_words, read_len;
    const uint16_t* lens;
    const uint8_t* quals; uint32_t qstride;
    const int32_t* bc;
    const int32_t* p_off; const uint32_t* p_n; const unsigned long long* p_start; const int32_t* p_edges;
    uint64_t n;
};

__device__ __forceinline__ uint32_t read_len_of(const dup_in& a, uint64_t r) { return a.lens ? min((uint32_t)a.lens[r], a.read_len) : a.read_len; }

// (e, offset, head of the mate): SecretOps.cc:430-441.  Reads without a path get the largest key and sort behind everything.
__global__ void __launch_bounds__(256) dup_key_kernel(dup_in a, unsigned long long* __restrict__ key, uint32_t* __restrict__ head, uint32_t* __restrict__ id,
                                                      unsigned long long* __restrict__ n_placed, uint32_t* __restrict__ range /* [3][256] max edge, max / min biased offset */) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool placed = false;
    uint32_t ke = 0, ko = 0;
    if (r < a.n) {
        id[r] = (uint32_t)r;
        unsigned long long k = ~0ull;
        uint32_t h = 0x3FFu;
        if (a.p_n[r]) {
            const uint32_t e = (uint32_t)a.p_edges[a.p_start[r]];
            const uint32_t off = (uint32_t)a.p_off[r] ^ 0x80000000u;          // signed order
            k = ((unsigned long long)e << 32) | off;
            ke = e; ko = off;
            h = a.rows[(r ^ 1ull) * a.row_words] >> 22;                        // five bases, MSB first = n*4 + base
            placed = true;
        }
        key[r] = k;
        head[r] = h;
    }
    // (one counter for every wave of the grid is 1.6 M atomics on ONE address, ~10 ns each: 16 of this kernel's 19 ms.  256 counters.)
    // (round 4: the 256 counters are 2 KB = 32 lines, and atomics on one 64-byte line are served one at a time whatever word they address:
    // 1.56 M waves x four atomics were still half of this kernel's 1.9 ms.  One addition per workgroup now, and the three range words
    // are looked at before they are touched -- they stop moving after the first few thousand reads.)
    __shared__ uint32_t wg_placed;
    if (threadIdx.x == 0) wg_placed = 0;
    __syncthreads();
    const unsigned long long m = __ballot(placed);
    const uint32_t slot = blockIdx.x & 255u;
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&wg_placed, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && wg_placed) atomicAdd(&n_placed[slot], (unsigned long long)wg_placed);
    // ranges of the placed reads' edges and offsets (they decide how many key bits the sort has to look at)
    uint32_t emax = placed ? ke : 0u, omax = placed ? ko : 0u, omin = placed ? ko : 0xFFFFFFFFu;
    for (int o = 32; o > 0; o >>= 1) {
        emax = max(emax, (uint32_t)__shfl_xor((int)emax, o)); omax = max(omax, (uint32_t)__shfl_xor((int)omax, o)); omin = min(omin, (uint32_t)__shfl_xor((int)omin, o));
    }
    if ((threadIdx.x & 63) == 0 && m) {
        // (a plain load may be stale -- then the atomic runs, as before; what it can never do is hide a value that still has to go in)
        if (emax > range[slot]) atomicMax(&range[slot], emax);
        if (omax > range[256 + slot]) atomicMax(&range[256 + slot], omax);
        if (omin < range[512 + slot]) atomicMin(&range[512 + slot], omin);
    }
}

__global__ void __launch_bounds__(256) dup_gather_key_kernel(const uint32_t* __restrict__ id, const unsigned long long* __restrict__ key, uint64_t n,
                                                             unsigned long long* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = key[id[i]];
}

// (edge, offset, head) in as few bits as the data need: edge << (ob + 10) | (offset - smallest offset) << 10 | head; unplaced reads
// get the one bit above -- ONE radix sort over those bits then orders the reads as the reference's records are ordered
__global__ void __launch_bounds__(256) dup_composite_kernel(unsigned long long* __restrict__ key, const uint32_t* __restrict__ head, uint64_t n, uint32_t omin,
                                                            uint32_t obits, uint32_t total_bits) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const unsigned long long k = key[r];
    key[r] = k == ~0ull ? (1ull << total_bits) : (((k >> 32) << (obits + 10u)) | ((unsigned long long)((uint32_t)k - omin) << 10) | head[r]);
}
__global__ void __launch_bounds__(256) dup_flag1_kernel(const unsigned long long* __restrict__ skey, uint64_t m, uint8_t* __restrict__ gstart, uint8_t* __restrict__ multi) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = skey[i];
    const bool same_prev = i > 0 && skey[i - 1] == k, same_next = i + 1 < m && skey[i + 1] == k;
    gstart[i] = same_prev ? 0 : 1;
    multi[i] = (same_prev || same_next) ? 1 : 0;
}

// group structure of the sorted array: gstart[i] = 1 iff position i opens a group; members of groups with more than one read
// are flagged (they need a quality sum)
__global__ void __launch_bounds__(256) dup_flag_kernel(const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sid,
                                                       const uint32_t* __restrict__ head, uint64_t m, uint8_t* __restrict__ gstart, uint8_t* __restrict__ multi) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = skey[i];
    const uint32_t h = head[sid[i]];
    const bool same_prev = i > 0 && skey[i - 1] == k && head[sid[i - 1]] == h;
    const bool same_next = i + 1 < m && skey[i + 1] == k && head[sid[i + 1]] == h;
    gstart[i] = same_prev ? 0 : 1;
    multi[i] = (same_prev || same_next) ? 1 : 0;
}

// sum of the qualities of a read and its mate (:480-499), members of multi-read groups only
__global__ void __launch_bounds__(256) dup_qsum_kernel(dup_in a, const uint32_t* __restrict__ sid, const uint8_t* __restrict__ multi, uint64_t m,
                                                       uint32_t* __restrict__ qsum) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !multi[i]) return;
    const uint64_t r = sid[i];
    uint32_t s = 0;
    for (int side = 0; side < 2; ++side) {
        const uint64_t x = side ? (r ^ 1ull) : r;
        const uint8_t* q = a.quals + x * a.qstride;
        const uint32_t L = read_len_of(a, x);
        for (uint32_t j = 0; j < L; ++j) s += q[j];
    }
    qsum[i] = s;
}

__device__ bool same_read(const dup_in& a, uint64_t x, uint64_t y) {
    const uint32_t L = read_len_of(a, x);
    if (L != read_len_of(a, y)) return false;
    const uint32_t full = L >> 4, rest = L & 15u;
    const uint32_t* rx = a.rows + x * a.row_words;
    const uint32_t* ry = a.rows + y * a.row_words;
    for (uint32_t w = 0; w < full; ++w) if (rx[w] != ry[w]) return false;
    if (rest && ((rx[full] ^ ry[full]) >> (32u - 2u * rest))) return false;
    const uint8_t* qx = a.quals + x * a.qstride;
    const uint8_t* qy = a.quals + y * a.qstride;
    for (uint32_t j = 0; j < L; ++j) if (qx[j] != qy[j]) return false;
    return true;
}

// one thread per group: the two walks of the reference (:449-474 and :505-556)
__global__ void __launch_bounds__(256) dup_group_kernel(dup_in a, const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sid,
                                                        const uint32_t* __restrict__ head, const uint8_t* __restrict__ gstart, const uint32_t* __restrict__ qsum,
                                                        uint64_t m, uint8_t* __restrict__ dup, uint8_t* __restrict__ art, unsigned long long* __restrict__ stat) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m || !gstart[j]) return;
    uint64_t k = j + 1;
    while (k < m && !gstart[k]) ++k;
    if (k - j < 2) return;
    // inter-barcode groups: b is the first member's barcode; while it is 0 it takes the next member's without a comparison
    int32_t b = a.bc ? a.bc[sid[j]] : 0;
    bool inter = false;
    for (uint64_t l = j + 1; l < k; ++l) {
        const int32_t c = a.bc ? a.bc[sid[l]] : 0;
        if (b == 0) b = c;
        else if (c != b) inter = true;
    }
    atomicAdd(&stat[0], (unsigned long long)(k - j - 1));
    if (inter) atomicAdd(&stat[1], (unsigned long long)(k - j - 1));
    // the survivor: largest quality sum, the earliest read among equals (ids ascend inside a group); a tie anywhere along the
    // walk (against the running maximum) sends the group to the artifact check
    uint64_t best = j;
    uint32_t q = qsum[j];
    bool tie = false;
    for (uint64_t l = j + 1; l < k; ++l) {
        const uint32_t ql = qsum[l];
        if (ql == q) tie = true;
        else if (ql > q) { q = ql; best = l; }
    }
    if (tie) {
        // Sort(qb) + runs (:530-547): a member is flagged when an identical read (bases and qualities) of a pair that does not
        // come later is in the group -- i.e. any earlier member, or the next one if that is its own mate
        for (uint64_t l = j; l < k; ++l) {
            const uint64_t x = sid[l];
            bool f = false;
            for (uint64_t l2 = j; l2 < l && !f; ++l2) f = same_read(a, x, sid[l2]);
            if (!f && l + 1 < k && (sid[l + 1] >> 1) == (x >> 1)) f = same_read(a, x, sid[l + 1]);
            if (f) art[x >> 1] = 1;
        }
    }
    for (uint64_t l = j; l < k; ++l)
        if (l != best) dup[sid[l] >> 1] = 1;
}

__global__ void __launch_bounds__(256) dup_count_kernel(const uint8_t* __restrict__ dup, const uint8_t* __restrict__ art, uint64_t np, unsigned long long* __restrict__ stat) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool d = i < np && dup[i], t = i < np && art[i];
    const unsigned long long md = __ballot(d), mt = __ballot(t);
    if ((threadIdx.x & 63) == 0) {
        if (md) atomicAdd(&stat[2], (unsigned long long)__popcll(md));
        if (mt) atomicAdd(&stat[3], (unsigned long long)__popcll(mt));
    }
}

}  // namespace

static int mark_dups_impl(snk_ctx* ctx, const snk_dev_reads* in, const snk_dev_paths* paths, snk_dev_dups* out, void* stream, char* err, size_t errcap);

extern "C" int snk_dev_mark_dups(snk_ctx* ctx, const snk_dev_reads* in, const snk_dev_paths* paths, snk_dev_dups* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !paths || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_mark_dups: NULL argument");
    // sort keys, group heads, quality sums (~35 B per read + the sort's own scratch) go back to the arena with the call; the
    // duplicate flags stay until the context's next snk_dev_count_graph / snk_shard_step
    const uint64_t mark = ctx->alloc_serial;
    const int rc = mark_dups_impl(ctx, in, paths, out, stream, err, errcap);
    (void)hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream);
    const void* keep[1] = {out->dup};
    snk_ctx_release_since(ctx, mark, keep, rc ? 0 : 1);
    if (rc) memset(out, 0, sizeof *out);
    return rc;
}

static int mark_dups_impl(snk_ctx* ctx, const snk_dev_reads* in, const snk_dev_paths* paths, snk_dev_dups* out, void* stream, char* err, size_t errcap) {
    const uint64_t n = in->n_reads;
    if (n != paths->n_reads) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_mark_dups: %llu reads, paths of %llu", (unsigned long long)n, (unsigned long long)paths->n_reads);
    if (n & 1ull) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_mark_dups: reads come in pairs (2q, 2q+1); got an odd number");
    if (n >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_mark_dups: more than 2^32 reads in one call");
    if (n && (!in->rows || !in->quals || in->read_len < 5 || in->row_words * 16 < in->read_len))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_mark_dups: need packed rows and quality rows of reads with at least five bases");
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    hipEvent_t e0, e1;
    SNK_HIP_TRY(hipEventCreate(&e0)); SNK_HIP_TRY(hipEventCreate(&e1));
    struct evg { hipEvent_t a, b; ~evg() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } g{e0, e1};
    SNK_HIP_TRY(hipEventRecord(e0, st));
    int rc;
    dup_in a;
    a.rows = (const uint32_t*)in->rows; a.row_words = in->row_words; a.read_len = in->read_len; a.lens = (const uint16_t*)in->lens;
    a.quals = (const uint8_t*)in->quals; a.qstride = in->qstride; a.bc = (const int32_t*)in->bc;
    a.p_off = (const int32_t*)paths->offset; a.p_n = (const uint32_t*)paths->n_edges; a.p_start = (const unsigned long long*)paths->start;
    a.p_edges = (const int32_t*)paths->edges; a.n = n;
    unsigned long long *key, *key2, *stat;
    uint32_t *head, *head2, *id, *id2, *qsum;
    uint8_t *gstart, *multi, *dup, *art;
    const uint64_t np = n / 2;
    if ((rc = dev(ctx, n, &key, err, errcap)) || (rc = dev(ctx, n, &key2, err, errcap)) || (rc = dev(ctx, n, &head, err, errcap)) || (rc = dev(ctx, n, &head2, err, errcap)) ||
        (rc = dev(ctx, n, &id, err, errcap)) || (rc = dev(ctx, n, &id2, err, errcap)) || (rc = dev(ctx, n, &qsum, err, errcap)) || (rc = dev(ctx, n, &gstart, err, errcap)) ||
        (rc = dev(ctx, n, &multi, err, errcap)) || (rc = dev(ctx, np, &dup, err, errcap)) || (rc = dev(ctx, np, &art, err, errcap)) || (rc = dev(ctx, 8 + 256, &stat, err, errcap)))
        return rc;
    SNK_HIP_TRY(hipMemsetAsync(stat, 0, (8 + 256) * 8, st));
    SNK_HIP_TRY(hipMemsetAsync(dup, 0, np ? np : 1, st));
    SNK_HIP_TRY(hipMemsetAsync(art, 0, np ? np : 1, st));
    unsigned long long h_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n_placed_total = 0;
    if (n) {
        const unsigned gn = (unsigned)((n + 255) / 256);
        uint32_t* range;
        if ((rc = dev(ctx, 768, &range, err, errcap))) return rc;
        SNK_HIP_TRY(hipMemsetAsync(range, 0, 512 * 4, st));
        SNK_HIP_TRY(hipMemsetAsync(range + 512, 0xFF, 256 * 4, st));
        hipLaunchKernelGGL(dup_key_kernel, dim3(gn), dim3(256), 0, st, a, key, head, id, stat + 8, range);
        unsigned long long h_placed[256];
        uint32_t h_range[768];
        SNK_HIP_TRY(hipMemcpyAsync(h_placed, stat + 8, sizeof h_placed, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(h_range, range, sizeof h_range, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        uint64_t m = 0;                                     // placed reads: the front of the sorted array
        uint32_t emax = 0, omax = 0, omin = 0xFFFFFFFFu;
        for (int q = 0; q < 256; ++q) { m += h_placed[q]; emax = std::max(emax, h_range[q]); omax = std::max(omax, h_range[256 + q]); omin = std::min(omin, h_range[512 + q]); }
        n_placed_total = m;
        auto bits_of = [](uint64_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b; };
        const uint32_t ebits = bits_of(emax), obits = m ? bits_of((uint64_t)omax - omin) : 0u, total_bits = ebits + obits + 10u;
        const unsigned long long* skey;
        const uint32_t* sid;
        const bool one_sort = total_bits <= 62 && !snk_opt_u32("dups_two_sorts", 0);
        if (one_sort) {
            // the reference's record order (edge, offset, mate head, read id) in ONE stable sort over total_bits + 1 bits (the bench graph: 42)
            hipLaunchKernelGGL(dup_composite_kernel, dim3(gn), dim3(256), 0, st, key, head, n, m ? omin : 0u, obits, total_bits);
            size_t tb = 0;
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, key, key2, id, id2, (size_t)n, 0u, total_bits + 1u, st));
            uint8_t* tmp;
            if ((rc = dev(ctx, tb, &tmp, err, errcap))) return rc;
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, key, key2, id, id2, (size_t)n, 0u, total_bits + 1u, st));
            skey = key2; sid = id2;
        } else {
            size_t tb1 = 0, tb2 = 0;
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb1, head, head2, id, id2, (size_t)n, 0u, 10u, st));
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb2, key, key2, id, id2, (size_t)n, 0u, 64u, st));
            size_t tb = tb1 > tb2 ? tb1 : tb2;
            uint8_t* tmp;
            if ((rc = dev(ctx, tb, &tmp, err, errcap))) return rc;
            size_t t = tb;
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, t, head, head2, id, id2, (size_t)n, 0u, 10u, st));          // id2 = ids by (head, id)
            hipLaunchKernelGGL(dup_gather_key_kernel, dim3(gn), dim3(256), 0, st, id2, key, n, key2);
            t = tb;
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, t, key2, key, id2, id, (size_t)n, 0u, 64u, st));            // key / id = (edge, offset, head, id) order
            skey = key; sid = id;
        }
        if (m) {
            const unsigned gm = (unsigned)((m + 255) / 256);
            if (one_sort) hipLaunchKernelGGL(dup_flag1_kernel, dim3(gm), dim3(256), 0, st, skey, m, gstart, multi);
            else hipLaunchKernelGGL(dup_flag_kernel, dim3(gm), dim3(256), 0, st, skey, sid, head, m, gstart, multi);
            hipLaunchKernelGGL(dup_qsum_kernel, dim3(gm), dim3(256), 0, st, a, sid, multi, m, qsum);
            hipLaunchKernelGGL(dup_group_kernel, dim3(gm), dim3(256), 0, st, a, skey, sid, head, gstart, qsum, m, dup, art, stat);
        }
        if (np) hipLaunchKernelGGL(dup_count_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, dup, art, np, stat);
        SNK_HIP_TRY(hipGetLastError());
        SNK_HIP_TRY(hipMemcpyAsync(h_stat, stat, 64, hipMemcpyDeviceToHost, st));
    }
    SNK_HIP_TRY(hipEventRecord(e1, st));
    SNK_HIP_TRY(snk_sync(st));
    out->n_pairs = np;
    out->dup = dup;
    out->n_dup_reads = h_stat[0];
    out->n_interdup_reads = h_stat[1];
    out->n_dup_pairs = h_stat[2];
    out->n_art_pairs = h_stat[3];
    out->n_placed = n_placed_total;
    out->interdup_rate = h_stat[0] ? (double)h_stat[1] / (double)h_stat[0] : 0.0;
    (void)hipEventElapsedTime(&out->ms, e0, e1);
    return SNK_OK;
}
