// snk_graph.hip -- K9..K11: adjacency prune, unitig ("edge") pull and packing, all on device.
//
// What it replaces:
//   KmerDict fill + recomputeAdjacencies  lib/assembly/src/kmers/ReadPather.h:346-385 (drop context bits whose
//                                          neighbour k-mer was not retained)
//   EdgeBuilder / buildEdges              lib/assembly/src/paths/long/BuildReadQGraph48.cc:327-541
//                                          (maximal unbranched walks, palindromes are 1-k-mer edges, canonical
//                                          orientation :457-464,481-485, smooth circles :348-397)
//   == tada build_sedges/build_edges       lib/tada/src/debruijn.rs:147-320,539-776.
//
// The reference walks each unitig sequentially under a spin-lock.  On the synthetic benchmark the
// whole genome is ONE unitig, so a walk has no parallelism at all; here the problem is recast as
// list ranking on the graph of reciprocal-unique links:
//   1. retained table sorted by key (rocPRIM radix sort on the 2K significant bits) -> index order ==
//      k-mer order, which also makes the run deterministic;
//   2. open-addressing index (fingerprint | position) in HBM for membership probes;
//   3. prune: every set context bit is probed once; sides left with exactly one bit remember the
//      neighbour's position and relative strand;
//   4. links: side s of node i is linked to the facing side of node j iff both sides have degree 1 and
//      neither k-mer is a palindrome (BuildReadQGraph48.cc:408-428,445-456);
//   5. pointer jumping (Wyllie) over the 2n directed states (node, exit side): distance to and identity
//      of both path ends for every node; states that never reach an end are on smooth circles: the
//      circle is cut at the left side of its minimum k-mer (canonicalizeCircle :375-397) and re-ranked;
//   6. orientation per path by the reference's rule (odd length: middle base & 2; even: lexicographic,
//      dna/CanonicalForm.h:35-48), prefix sums for offsets, one base per node scattered into place.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"
#include "snk_graph.h"

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int TB = 256;

__device__ __forceinline__ snk_kmer load_key(const snk_u128* keys, uint64_t i) {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(keys + i);
    snk_kmer k;
    k.lo = p[0];
    k.hi = p[1];
    return k;
}

// ------------------------------------------------------------------ index build / probe
__global__ void __launch_bounds__(TB) index_build_kernel(const snk_u128* __restrict__ keys, uint64_t n,
                                                         unsigned long long* __restrict__ tab, uint64_t mask) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    snk_kmer k = load_key(keys, i);
    uint32_t h1, h2;
    snk_kmer_hash2(k, &h1, &h2);
    uint64_t slot = (((uint64_t)h1 << 32) | h2) & mask;
    unsigned long long ent = ((unsigned long long)h1 << 32) | (unsigned long long)(i + 1);
    for (;;) {
        unsigned long long old = atomicCAS(&tab[slot], 0ull, ent);
        if (old == 0ull) break;
        slot = (slot + 1) & mask;
    }
}

__device__ __forceinline__ int64_t index_find(const snk_u128* __restrict__ keys, const unsigned long long* __restrict__ tab,
                                              uint64_t mask, snk_kmer k) {
    uint32_t h1, h2;
    snk_kmer_hash2(k, &h1, &h2);
    uint64_t slot = (((uint64_t)h1 << 32) | h2) & mask;
    for (;;) {
        unsigned long long e = tab[slot];
        if (e == 0ull) return -1;
        if ((uint32_t)(e >> 32) == h1) {
            uint64_t idx = (uint32_t)e - 1u;
            snk_kmer c = load_key(keys, idx);
            if (snk_kmer_eq(c, k)) return (int64_t)idx;
        }
        slot = (slot + 1) & mask;
    }
}

template <int K>
__device__ __forceinline__ int64_t find_any(const snk_u128* keys, const unsigned long long* tab, uint64_t mask, snk_kmer k,
                                            uint32_t* rev) {
    snk_kmer r = snk_kmer_rc<K>(k);
    bool isrev = snk_kmer_lt(r, k);   // KmerDict::findEntry canonicalises (ReadPather.h:241-245)
    *rev = isrev ? 1u : 0u;
    return index_find(keys, tab, mask, isrev ? r : k);
}

// ------------------------------------------------------------------ prune (ReadPather.h:346-385)
// side 0 = successors (low nibble), side 1 = predecessors (high nibble)
template <int K>
__global__ void __launch_bounds__(TB) prune_kernel(const snk_u128* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                   uint64_t n, const unsigned long long* __restrict__ tab, uint64_t mask,
                                                   uint32_t do_prune, uint8_t* __restrict__ ctx_out,
                                                   uint32_t* __restrict__ count_out, uint32_t* __restrict__ nbr) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    snk_kmer k = load_key(keys, i);
    uint64_t v = vals[i];
    uint32_t c = (uint32_t)(v & 0xFFu);
    count_out[i] = (uint32_t)(v >> 8);
    uint32_t keep = 0;
    uint32_t nb0 = NONE, nb1 = NONE;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        if (c & (1u << b)) {
            uint32_t rev;
            int64_t j = find_any<K>(keys, tab, mask, snk_kmer_succ<K>(k, b), &rev);
            if (j >= 0 || !do_prune) { keep |= 1u << b; nb0 = j >= 0 ? ((uint32_t)j << 1) | rev : NONE; }
        }
        if (c & (0x10u << b)) {
            uint32_t rev;
            int64_t j = find_any<K>(keys, tab, mask, snk_kmer_pred<K>(k, b), &rev);
            if (j >= 0 || !do_prune) { keep |= 0x10u << b; nb1 = j >= 0 ? ((uint32_t)j << 1) | rev : NONE; }
        }
    }
    ctx_out[i] = (uint8_t)keep;
    nbr[2 * i + 0] = __popc(keep & 0x0Fu) == 1 ? nb0 : NONE;
    nbr[2 * i + 1] = __popc(keep & 0xF0u) == 1 ? nb1 : NONE;
}

// ------------------------------------------------------------------ links (BuildReadQGraph48.cc:408-428,445-456)
template <int K>
__global__ void __launch_bounds__(TB) link_kernel(const snk_u128* __restrict__ keys, const uint8_t* __restrict__ ctx,
                                                  const uint32_t* __restrict__ nbr, uint64_t n, uint32_t* __restrict__ link) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= 2 * n) return;
    uint32_t nb = nbr[s];
    uint32_t out = NONE;
    if (nb != NONE) {
        uint64_t i = s >> 1;
        uint32_t side = (uint32_t)(s & 1);
        uint32_t j = nb >> 1, rev = nb & 1u;
        snk_kmer ki = load_key(keys, i), kj = load_key(keys, j);
        bool pal = snk_kmer_eq(ki, snk_kmer_rc<K>(ki)) || snk_kmer_eq(kj, snk_kmer_rc<K>(kj));
        uint32_t fs = side ^ 1u ^ rev;                 // side of j that faces i
        uint32_t cj = ctx[j];
        uint32_t deg = fs ? __popc(cj & 0xF0u) : __popc(cj & 0x0Fu);
        if (!pal && deg == 1) out = (j << 1) | fs;
    }
    link[s] = out;
}

// ------------------------------------------------------------------ list ranking over directed states
__global__ void __launch_bounds__(TB) rank_init_kernel(const uint32_t* __restrict__ link, uint64_t ns,
                                                       uint32_t* __restrict__ nxt, uint32_t* __restrict__ dist,
                                                       uint32_t* __restrict__ tail) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    uint32_t l = link[s];
    if (l == NONE) { nxt[s] = NONE; dist[s] = 0; tail[s] = (uint32_t)s; }
    else { nxt[s] = l ^ 1u; dist[s] = 1; tail[s] = l ^ 1u; }
}

__global__ void __launch_bounds__(TB) rank_round_kernel(const uint32_t* __restrict__ nxt_in, const uint32_t* __restrict__ dist_in,
                                                        const uint32_t* __restrict__ tail_in, uint64_t ns,
                                                        uint32_t* __restrict__ nxt_out, uint32_t* __restrict__ dist_out,
                                                        uint32_t* __restrict__ tail_out, uint32_t* __restrict__ changed) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    uint32_t n1 = nxt_in[s];
    if (n1 == NONE) { nxt_out[s] = NONE; dist_out[s] = dist_in[s]; tail_out[s] = tail_in[s]; return; }
    nxt_out[s] = nxt_in[n1];
    dist_out[s] = dist_in[s] + dist_in[n1];
    tail_out[s] = tail_in[n1];
    *changed = 1u;
}

// smooth circles: states that still have a successor after ceil(log2(ns))+1 rounds
__global__ void __launch_bounds__(TB) cyc_init_kernel(const uint32_t* __restrict__ nxt_final, const uint32_t* __restrict__ link,
                                                      uint64_t ns, uint32_t* __restrict__ jump, uint32_t* __restrict__ mn) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    if (nxt_final[s] == NONE) { jump[s] = NONE; mn[s] = NONE; }
    else { jump[s] = link[s] ^ 1u; mn[s] = (uint32_t)(s >> 1); }
}
__global__ void __launch_bounds__(TB) cyc_round_kernel(const uint32_t* __restrict__ jump_in, const uint32_t* __restrict__ mn_in,
                                                       uint64_t ns, uint32_t* __restrict__ jump_out, uint32_t* __restrict__ mn_out) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    uint32_t j = jump_in[s];
    if (j == NONE) { jump_out[s] = NONE; mn_out[s] = NONE; return; }
    uint32_t a = mn_in[s], b = mn_in[j];
    mn_out[s] = a < b ? a : b;
    jump_out[s] = jump_in[j];
}
// cut every circle at the left side of its minimum k-mer
__global__ void __launch_bounds__(TB) cyc_cut_kernel(const uint32_t* __restrict__ mn, uint64_t n, uint32_t* __restrict__ link,
                                                     uint32_t* __restrict__ n_cut) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint64_t s = 2 * i + 1;
    if (mn[s] == (uint32_t)i) {
        uint32_t partner = link[s];
        link[s] = NONE;
        if (partner != NONE) link[partner] = NONE;
        atomicAdd(n_cut, 1u);
    }
}

// ------------------------------------------------------------------ orientation, offsets, emission
template <int K>
__device__ __forceinline__ uint32_t oriented_base(snk_kmer k, bool rc, int idx) {
    // base idx of the k-mer read forward, or of its reverse complement
    return rc ? (snk_kmer_base<K>(k, K - 1 - idx) ^ 3u) : snk_kmer_base<K>(k, idx);
}

struct node_place {
    uint32_t pid;     // path id = smaller terminal state
    uint32_t other;   // the larger terminal state
    uint32_t n;       // nodes on the path
    uint32_t pos;     // position walking from terminal `pid`
    bool rc;          // traversed as reverse complement when walking from `pid`
};
__device__ __forceinline__ node_place place_of(const uint32_t* dist, const uint32_t* tail, uint64_t i) {
    uint32_t tR = tail[2 * i], tL = tail[2 * i + 1];
    uint32_t dR = dist[2 * i], dL = dist[2 * i + 1];
    node_place p;
    p.n = dR + dL + 1u;
    if (tL < tR) { p.pid = tL; p.other = tR; p.pos = dL; p.rc = false; }
    else { p.pid = tR; p.other = tL; p.pos = dR; p.rc = true; }
    return p;
}

// REV decision per path (getCanonicalForm, dna/CanonicalForm.h:35-48); written by exactly one node of the path
template <int K>
__global__ void __launch_bounds__(TB) orient_kernel(const snk_u128* __restrict__ keys, const uint32_t* __restrict__ dist,
                                                    const uint32_t* __restrict__ tail, uint64_t n, uint8_t* __restrict__ prev) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(dist, tail, i);
    uint64_t L = (uint64_t)K + p.n - 1;
    snk_kmer k = load_key(keys, i);
    if (L & 1) {
        uint64_t mid = L / 2;
        if (mid <= (uint64_t)(K - 1)) {
            if (p.pos == 0) prev[p.pid] = (oriented_base<K>(k, p.rc, (int)mid) & 2u) ? 1 : 0;
        } else if ((uint64_t)p.pos == mid - (K - 1)) {
            prev[p.pid] = (oriented_base<K>(k, p.rc, K - 1) & 2u) ? 1 : 0;
        }
    } else if (p.pos == 0) {
        snk_kmer first = p.rc ? snk_kmer_rc<K>(k) : k;
        uint32_t eB = p.other >> 1, xB = p.other & 1u;
        snk_kmer kb = load_key(keys, eB);
        snk_kmer rc_last = (xB == 0) ? snk_kmer_rc<K>(kb) : kb;   // rc of the last oriented k-mer
        prev[p.pid] = snk_kmer_lt(rc_last, first) ? 1 : 0;
    }
}

// head flags and unitig lengths
__global__ void __launch_bounds__(TB) head_kernel(const uint32_t* __restrict__ dist, const uint32_t* __restrict__ tail,
                                                  const uint8_t* __restrict__ prev, uint64_t n, uint32_t K,
                                                  uint32_t* __restrict__ hflag, uint64_t* __restrict__ hlen) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(dist, tail, i);
    uint32_t pos = prev[p.pid] ? p.n - 1u - p.pos : p.pos;
    bool head = pos == 0;
    hflag[i] = head ? 1u : 0u;
    hlen[i] = head ? (uint64_t)K + p.n - 1 : 0ull;
}
__global__ void __launch_bounds__(TB) head_place_kernel(const uint32_t* __restrict__ tail, const uint32_t* __restrict__ hflag,
                                                        const uint32_t* __restrict__ hidx, const uint64_t* __restrict__ hoff,
                                                        uint64_t n, uint64_t* __restrict__ poff, uint64_t* __restrict__ unitig_off) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    if (hflag[i]) {
        uint32_t tR = tail[2 * i], tL = tail[2 * i + 1];
        uint32_t pid = tL < tR ? tL : tR;
        poff[pid] = hoff[i];
        unitig_off[hidx[i]] = hoff[i];
    }
}
template <int K>
__global__ void __launch_bounds__(TB) emit_kernel(const snk_u128* __restrict__ keys, const uint32_t* __restrict__ dist,
                                                  const uint32_t* __restrict__ tail, const uint8_t* __restrict__ prev,
                                                  const uint64_t* __restrict__ poff, uint64_t n, uint8_t* __restrict__ bases) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(dist, tail, i);
    bool flip = prev[p.pid] != 0;
    uint32_t pos = flip ? p.n - 1u - p.pos : p.pos;
    bool rc = flip ? !p.rc : p.rc;
    snk_kmer k = load_key(keys, i);
    uint64_t off = poff[p.pid];
    if (pos == 0) {
        for (int b = 0; b < K; ++b) bases[off + b] = (uint8_t)oriented_base<K>(k, rc, b);
    } else {
        bases[off + (K - 1) + pos] = (uint8_t)oriented_base<K>(k, rc, K - 1);
    }
}

// k-mer spectrum of the retained table (WriteKmerSpectrum, BuildReadQGraph48.cc:199-216); LDS-privatised bins
constexpr int SPEC_LDS = 2048;
__global__ void __launch_bounds__(TB) spectrum_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                      unsigned long long* __restrict__ bins, uint32_t nbins) {
    __shared__ uint32_t h[SPEC_LDS];
    for (int j = threadIdx.x; j < SPEC_LDS; j += TB) h[j] = 0;
    __syncthreads();
    uint64_t stride = (uint64_t)gridDim.x * TB;
    for (uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x; i < n; i += stride) {
        uint32_t c = counts[i];
        if (c >= nbins) c = nbins - 1;
        if (c < (uint32_t)SPEC_LDS) atomicAdd(&h[c], 1u);
        else atomicAdd(&bins[c], 1ull);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < SPEC_LDS && j < (int)nbins; j += TB)
        if (h[j]) atomicAdd(&bins[j], (unsigned long long)h[j]);
}

inline unsigned nblk(uint64_t n) { return (unsigned)((n + TB - 1) / TB); }

}  // namespace

#define G_ALLOC(ptr, type, count)                                                       \
    do {                                                                                \
        void* _p = nullptr;                                                             \
        int _rc = snk_ctx_alloc(ctx, sizeof(type) * (size_t)(count), &_p, err, errcap); \
        if (_rc) return _rc;                                                            \
        ptr = (type*)_p;                                                                \
    } while (0)

__global__ void __launch_bounds__(256) sorted_check_kernel(const snk_u128* __restrict__ keys, uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n) return;
    if (!(keys[i] < keys[i + 1])) *bad = 1u;   // retained k-mers are distinct: strictly ascending
}

// sort (keys, vals) by key.  Only the top 2K bits are significant; rocPRIM's bit-range path is used for
// large inputs (onesweep) and the full 128-bit sort for small ones (its single-block/merge path mis-sorts
// 128-bit keys when begin_bit != 0 -- ROCm 7.2, see tools/probe/sort128.hip).  The result is verified.
int snk_graph_sort(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t n, snk_u128* keys_in, uint64_t* vals_in,
                   snk_u128* keys_out, uint64_t* vals_out, char* err, size_t errcap) {
    if (n == 0) return SNK_OK;
    uint32_t* bad = nullptr;
    {
        void* q;
        int rc = snk_ctx_alloc(ctx, 16, &q, err, errcap);
        if (rc) return rc;
        bad = (uint32_t*)q;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        unsigned begin_bit = (attempt == 0 && n >= (8u << 20)) ? 128u - 2u * K : 0u;
        size_t tmp_bytes = 0;
        SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n,
                                              begin_bit, 128u, st));
        void* tmp = nullptr;
        int rc = snk_ctx_alloc(ctx, tmp_bytes, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, begin_bit,
                                              128u, st));
        SNK_HIP_TRY(hipMemsetAsync(bad, 0, 4, st));
        hipLaunchKernelGGL(sorted_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys_out, n, bad);
        uint32_t h_bad = 0;
        SNK_HIP_TRY(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipStreamSynchronize(st));
        if (!h_bad) return SNK_OK;
        if (begin_bit == 0) break;
    }
    return snk_fail(SNK_E_INTERNAL, err, errcap, "retained k-mer table is not strictly ascending after the sort");
}

template <int K>
static int graph_impl(snk_ctx* ctx, hipStream_t st, const snk_u128* keys, const uint64_t* vals, uint64_t n,
                      uint32_t do_prune, bool want_unitigs, snk_graph_out* out, char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    if (n == 0) return SNK_OK;
    if (n >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 retained k-mers on one GPU (%llu)", (unsigned long long)n);
    // index
    uint64_t tg = 1024;
    while (tg < 2 * n) tg <<= 1;
    unsigned long long* tab;
    G_ALLOC(tab, unsigned long long, tg);
    SNK_HIP_TRY(hipMemsetAsync(tab, 0, tg * 8, st));
    hipLaunchKernelGGL(index_build_kernel, dim3(nblk(n)), dim3(TB), 0, st, keys, n, tab, tg - 1);
    // prune
    uint8_t* ctx_out; uint32_t* count_out; uint32_t* nbr;
    G_ALLOC(ctx_out, uint8_t, n);
    G_ALLOC(count_out, uint32_t, n);
    G_ALLOC(nbr, uint32_t, 2 * n);
    hipLaunchKernelGGL((prune_kernel<K>), dim3(nblk(n)), dim3(TB), 0, st, keys, vals, n, tab, tg - 1, do_prune, ctx_out,
                       count_out, nbr);
    SNK_HIP_TRY(hipGetLastError());
    out->ctx = ctx_out;
    out->counts = count_out;
    // spectrum
    constexpr uint32_t NBINS = 65536;
    unsigned long long* bins;
    G_ALLOC(bins, unsigned long long, NBINS);
    SNK_HIP_TRY(hipMemsetAsync(bins, 0, NBINS * 8, st));
    {
        unsigned g = nblk(n);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(spectrum_kernel, dim3(g), dim3(TB), 0, st, count_out, n, bins, NBINS);
    }
    out->spectrum = bins;
    out->spectrum_bins = NBINS;
    if (!want_unitigs) return SNK_OK;

    const uint64_t ns = 2 * n;
    uint32_t* link;
    G_ALLOC(link, uint32_t, ns);
    hipLaunchKernelGGL((link_kernel<K>), dim3(nblk(ns)), dim3(TB), 0, st, keys, ctx_out, nbr, n, link);
    SNK_HIP_TRY(hipGetLastError());

    uint32_t *nxt[2], *dst[2], *tl[2];
    for (int b = 0; b < 2; ++b) { G_ALLOC(nxt[b], uint32_t, ns); G_ALLOC(dst[b], uint32_t, ns); G_ALLOC(tl[b], uint32_t, ns); }
    uint32_t* flags;   // [0] changed, [1] circles cut
    G_ALLOC(flags, uint32_t, 4);
    uint32_t* h_flags = nullptr;
    SNK_HIP_TRY(hipHostMalloc((void**)&h_flags, 16, hipHostMallocDefault));
    int max_rounds = 2;
    while ((1ull << (max_rounds - 1)) < ns) ++max_rounds;
    int cur = 0;
    uint32_t rounds_total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        cur = 0;
        hipLaunchKernelGGL(rank_init_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, ns, nxt[0], dst[0], tl[0]);
        bool converged = false;
        for (int r = 0; r < max_rounds; ++r) {
            SNK_HIP_TRY(hipMemsetAsync(flags, 0, 4, st));
            hipLaunchKernelGGL(rank_round_kernel, dim3(nblk(ns)), dim3(TB), 0, st, nxt[cur], dst[cur], tl[cur], ns,
                               nxt[cur ^ 1], dst[cur ^ 1], tl[cur ^ 1], flags);
            cur ^= 1;
            ++rounds_total;
            SNK_HIP_TRY(hipMemcpyAsync(h_flags, flags, 4, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(hipStreamSynchronize(st));
            if (h_flags[0] == 0) { converged = true; break; }
        }
        if (converged) break;
        if (attempt == 1) { (void)hipHostFree(h_flags); return snk_fail(SNK_E_INTERNAL, err, errcap, "unitig ranking did not converge after the circle cut"); }
        // smooth circles: find each circle's minimum k-mer (index order == key order), cut there, rank again
        uint32_t *jump[2] = {dst[cur ^ 1], tl[cur ^ 1]};   // reuse the spare ranking buffers
        uint32_t* mn[2];
        G_ALLOC(mn[0], uint32_t, ns);
        G_ALLOC(mn[1], uint32_t, ns);
        hipLaunchKernelGGL(cyc_init_kernel, dim3(nblk(ns)), dim3(TB), 0, st, nxt[cur], link, ns, jump[0], mn[0]);
        int c2 = 0;
        for (int r = 0; r < max_rounds; ++r) {
            hipLaunchKernelGGL(cyc_round_kernel, dim3(nblk(ns)), dim3(TB), 0, st, jump[c2], mn[c2], ns, jump[c2 ^ 1], mn[c2 ^ 1]);
            c2 ^= 1;
        }
        SNK_HIP_TRY(hipMemsetAsync(flags + 1, 0, 4, st));
        hipLaunchKernelGGL(cyc_cut_kernel, dim3(nblk(n)), dim3(TB), 0, st, mn[c2], n, link, flags + 1);
        SNK_HIP_TRY(hipGetLastError());
    }
    SNK_HIP_TRY(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    out->n_circles = h_flags[1];
    out->rank_rounds = rounds_total;
    const uint32_t* dist = dst[cur];
    const uint32_t* tail = tl[cur];

    uint8_t* prev;
    G_ALLOC(prev, uint8_t, ns);
    SNK_HIP_TRY(hipMemsetAsync(prev, 0, ns, st));
    hipLaunchKernelGGL((orient_kernel<K>), dim3(nblk(n)), dim3(TB), 0, st, keys, dist, tail, n, prev);
    uint32_t *hflag, *hidx;
    uint64_t *hlen, *hoff;
    G_ALLOC(hflag, uint32_t, n + 1);
    G_ALLOC(hidx, uint32_t, n + 1);
    G_ALLOC(hlen, uint64_t, n + 1);
    G_ALLOC(hoff, uint64_t, n + 1);
    SNK_HIP_TRY(hipMemsetAsync(hflag + n, 0, 4, st));
    SNK_HIP_TRY(hipMemsetAsync(hlen + n, 0, 8, st));
    hipLaunchKernelGGL(head_kernel, dim3(nblk(n)), dim3(TB), 0, st, dist, tail, prev, n, (uint32_t)K, hflag, hlen);
    SNK_HIP_TRY(hipGetLastError());
    {
        size_t t1 = 0, t2 = 0;
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, t1, hflag, hidx, 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), st));
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, t2, hlen, hoff, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
        void* tmp;
        int rc = snk_ctx_alloc(ctx, t1 > t2 ? t1 : t2, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, t1, hflag, hidx, 0u, (size_t)(n + 1), rocprim::plus<uint32_t>(), st));
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, t2, hlen, hoff, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
    }
    uint64_t* h_tot = reinterpret_cast<uint64_t*>(h_flags);
    uint32_t h_nu = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_nu, hidx + n, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(h_tot, hoff + n, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    uint64_t n_unitigs = h_nu, total_bases = h_tot[0];
    (void)hipHostFree(h_flags);
    uint64_t *poff, *uoff;
    uint8_t* bases;
    G_ALLOC(poff, uint64_t, ns);
    G_ALLOC(uoff, uint64_t, n_unitigs + 1);
    G_ALLOC(bases, uint8_t, total_bases);
    hipLaunchKernelGGL(head_place_kernel, dim3(nblk(n)), dim3(TB), 0, st, tail, hflag, hidx, hoff, n, poff, uoff);
    SNK_HIP_TRY(hipMemcpyAsync(uoff + n_unitigs, hoff + n, 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((emit_kernel<K>), dim3(nblk(n)), dim3(TB), 0, st, keys, dist, tail, prev, poff, n, bases);
    SNK_HIP_TRY(hipGetLastError());
    out->n_unitigs = n_unitigs;
    out->total_bases = total_bases;
    out->unitig_off = uoff;
    out->unitig_bases = bases;
    return SNK_OK;
}

int snk_graph_build(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_u128* keys, const uint64_t* vals, uint64_t n,
                    uint32_t do_prune, bool want_unitigs, snk_graph_out* out, char* err, size_t errcap) {
    if (K == 48) return graph_impl<48>(ctx, st, keys, vals, n, do_prune, want_unitigs, out, err, errcap);
    if (K == 60) return graph_impl<60>(ctx, st, keys, vals, n, do_prune, want_unitigs, out, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}
