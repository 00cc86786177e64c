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
#include "snk_stages.h"

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
                                                     uint32_t* __restrict__ n_cut, uint8_t* __restrict__ circ_state) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint64_t s = 2 * i + 1;
    if (mn[s] == (uint32_t)i) {
        uint32_t partner = link[s];
        link[s] = NONE;
        if (partner != NONE) link[partner] = NONE;
        if (circ_state) { circ_state[s] = 1; if (partner != NONE) circ_state[partner] = 1; }   // both new terminals
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
// rk[state] = (distance to the end of the path, terminal state): one 8-byte record per state, so the ranking's
// second walk does one scattered store per state instead of two
__device__ __forceinline__ node_place place_of(const uint2* rk, uint64_t i) {
    const uint2 a = rk[2 * i], b = rk[2 * i + 1];
    uint32_t tR = a.y, tL = b.y;
    uint32_t dR = a.x, dL = b.x;
    node_place p;
    p.n = dR + dL + 1u;
    if (tL < tR) { p.pid = tL; p.other = tR; p.pos = dL; p.rc = false; }
    else { p.pid = tR; p.other = tL; p.pos = dR; p.rc = true; }
    return p;
}

// REV decision per path (getCanonicalForm, dna/CanonicalForm.h:35-48); written by exactly one node of the path
template <int K>
__global__ void __launch_bounds__(TB) orient_kernel(const snk_u128* __restrict__ keys, const uint2* __restrict__ rk, uint64_t n, uint8_t* __restrict__ prev) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(rk, i);
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
__global__ void __launch_bounds__(TB) head_kernel(const uint2* __restrict__ rk,
                                                  const uint8_t* __restrict__ prev, uint64_t n, uint32_t K,
                                                  uint32_t* __restrict__ hflag, uint64_t* __restrict__ hlen) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(rk, i);
    uint32_t pos = prev[p.pid] ? p.n - 1u - p.pos : p.pos;
    bool head = pos == 0;
    hflag[i] = head ? 1u : 0u;
    hlen[i] = head ? (uint64_t)K + p.n - 1 : 0ull;
}
__global__ void __launch_bounds__(TB) head_place_kernel(const uint2* __restrict__ rk, const uint32_t* __restrict__ hflag,
                                                        const uint32_t* __restrict__ hidx, const uint64_t* __restrict__ hoff,
                                                        uint64_t n, uint64_t* __restrict__ poff, uint64_t* __restrict__ unitig_off) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    if (hflag[i]) {
        uint32_t tR = rk[2 * i].y, tL = rk[2 * i + 1].y;
        uint32_t pid = tL < tR ? tL : tR;
        poff[pid] = hoff[i];
        unitig_off[hidx[i]] = hoff[i];
    }
}
template <int K>
__global__ void __launch_bounds__(TB) emit_kernel(const snk_u128* __restrict__ keys, const uint2* __restrict__ rk,
                                                  const uint8_t* __restrict__ prev,
                                                  const uint64_t* __restrict__ poff, uint64_t n, uint8_t* __restrict__ bases) {
    uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    node_place p = place_of(rk, i);
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
        if (c > 0xFFFFFFu) c = 0xFFFFFFu;            // the reference's KDef count saturates at 2^24-1 (kmers/ReadPather.h:128-129,145)
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
        SNK_HIP_TRY(snk_sync(st));
        if (!h_bad) return SNK_OK;
        if (begin_bit == 0) break;
    }
    return snk_fail(SNK_E_INTERNAL, err, errcap, "retained k-mer table is not strictly ascending after the sort");
}

// weighted start of the ranking (fragment join): the distance counts k-mers, not hops
__global__ void __launch_bounds__(TB) rank_init_w_kernel(const uint32_t* __restrict__ link, const uint32_t* __restrict__ w,
                                                         uint64_t ns, uint32_t* __restrict__ nxt, uint32_t* __restrict__ dist,
                                                         uint32_t* __restrict__ tail) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    uint32_t l = link[s];
    if (l == NONE) { nxt[s] = NONE; dist[s] = 0; tail[s] = (uint32_t)s; }
    else { nxt[s] = l ^ 1u; dist[s] = w[l >> 1]; tail[s] = l ^ 1u; }
}
__global__ void __launch_bounds__(TB) rank_zip_kernel(const uint32_t* __restrict__ dist, const uint32_t* __restrict__ tail, uint64_t ns,
                                                      uint2* __restrict__ rk) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s < ns) rk[s] = make_uint2(dist[s], tail[s]);
}

// List ranking over the 2n directed states (node, exit side) of a degree<=2 link graph: for every state the
// number of steps (or summed weights) to the end of its path and the terminal state.  Smooth circles are cut
// at the left side of their minimum node and ranked again.  link[] is modified by the cut.
static int rank_lists_wyllie(snk_ctx* ctx, hipStream_t st, uint32_t* link, uint64_t n, const uint32_t* weights, uint8_t* circ /* per state, nullable */,
                      const uint2** rk_out, uint32_t* n_circles, uint32_t* rounds,
                      char* err, size_t errcap) {
    const uint64_t ns = 2 * n;
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
        if (weights) hipLaunchKernelGGL(rank_init_w_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, weights, ns, nxt[0], dst[0], tl[0]);
        else hipLaunchKernelGGL(rank_init_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, ns, nxt[0], dst[0], tl[0]);
        bool converged = false;
        for (int r = 0; r < max_rounds; ++r) {
            SNK_HIP_TRY(hipMemsetAsync(flags, 0, 4, st));
            hipLaunchKernelGGL(rank_round_kernel, dim3(nblk(ns)), dim3(TB), 0, st, nxt[cur], dst[cur], tl[cur], ns,
                               nxt[cur ^ 1], dst[cur ^ 1], tl[cur ^ 1], flags);
            cur ^= 1;
            ++rounds_total;
            SNK_HIP_TRY(hipMemcpyAsync(h_flags, flags, 4, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
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
        hipLaunchKernelGGL(cyc_cut_kernel, dim3(nblk(n)), dim3(TB), 0, st, mn[c2], n, link, flags + 1, circ);
        SNK_HIP_TRY(hipGetLastError());
    }
    SNK_HIP_TRY(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    *n_circles = h_flags[1];
    *rounds = rounds_total;
    (void)hipHostFree(h_flags);
    uint2* rkz;
    G_ALLOC(rkz, uint2, ns);
    hipLaunchKernelGGL(rank_zip_kernel, dim3(nblk(ns)), dim3(TB), 0, st, dst[cur], tl[cur], ns, rkz);
    SNK_HIP_TRY(hipGetLastError());
    *rk_out = rkz;
    return SNK_OK;
}


// ---- work-efficient ranking: sparse ruling set.  Wyllie's pointer jumping touches every state log2(len) times
// (21-28 rounds of random 12-byte gathers on the benchmark, 60% of the whole step); here every state is touched
// twice: list heads and a hashed 1/32 sample of the states are "splitters", each walks to the next splitter, the
// short splitter list is ranked by pointer jumping, and a second walk hands the ranks to the states in between.
__device__ __forceinline__ bool sampled_state(uint32_t s, uint32_t split_mask) { return ((snk_mix32(s >> 1) >> 7) & split_mask) == 0; }

__global__ void __launch_bounds__(TB) spl_mark_kernel(const uint32_t* __restrict__ link, uint64_t ns, uint32_t split_mask,
                                                      uint8_t* __restrict__ spl, uint32_t* __restrict__ flag32) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    bool head = link[s ^ 1] == NONE;                 // nobody walks into s: it starts a list
    bool sp = head || sampled_state((uint32_t)s, split_mask);
    spl[s] = sp ? 1 : 0;
    flag32[s] = sp ? 1u : 0u;
}
__global__ void __launch_bounds__(TB) spl_collect_kernel(const uint8_t* __restrict__ spl, const uint32_t* __restrict__ sid, uint64_t ns,
                                                         uint32_t* __restrict__ spl_state) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns || !spl[s]) return;
    spl_state[sid[s]] = (uint32_t)s;
}
// one 8-byte record per state for the walks: link (32) | splitter id (31) | is-splitter (1) -- a walk step is then
// ONE random HBM transaction instead of two (link[] and spl[]): the walks are transaction bound (PMC: ~128 B
// fetched per step with separate arrays)
__global__ void __launch_bounds__(TB) spl_pack_kernel(const uint32_t* __restrict__ link, const uint8_t* __restrict__ spl,
                                                      const uint32_t* __restrict__ sid, uint64_t ns, unsigned long long* __restrict__ wrec) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= ns) return;
    unsigned long long r = link[s];
    if (spl[s]) r |= ((unsigned long long)sid[s] << 32) | (1ull << 63);
    wrec[s] = r;
}
__global__ void __launch_bounds__(TB) spl_walk1_kernel(const unsigned long long* __restrict__ wrec, const uint32_t* __restrict__ spl_state,
                                                       const uint32_t* __restrict__ w, uint64_t m, uint32_t* __restrict__ rnxt,
                                                       uint32_t* __restrict__ rdist, uint32_t* __restrict__ rtail) {
    uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= m) return;
    uint32_t cur = spl_state[k];
    unsigned long long rec = wrec[cur];
    uint32_t d = 0, nx = NONE;
    for (;;) {
        uint32_t l = (uint32_t)rec;
        if (l == NONE) { nx = NONE; break; }
        cur = l ^ 1u;
        rec = wrec[cur];
        d += w ? w[cur >> 1] : 1u;
        if (rec >> 63) { nx = (uint32_t)(rec >> 32) & 0x7FFFFFFFu; break; }
    }
    rnxt[k] = nx;
    rdist[k] = d;
    rtail[k] = cur;        // the terminal state when nx == NONE (overwritten by the jumping otherwise)
}
__global__ void __launch_bounds__(TB) spl_walk2_kernel(const unsigned long long* __restrict__ wrec, const uint32_t* __restrict__ spl_state,
                                                       const uint32_t* __restrict__ w, const uint32_t* __restrict__ rdist,
                                                       const uint32_t* __restrict__ rtail, uint64_t m, uint2* __restrict__ rk) {
    uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= m) return;
    uint32_t cur = spl_state[k];
    unsigned long long rec = wrec[cur];
    uint32_t d = rdist[k];
    const uint32_t t = rtail[k];
    for (;;) {
        rk[cur] = make_uint2(d, t);
        uint32_t l = (uint32_t)rec;
        if (l == NONE) break;
        cur = l ^ 1u;
        rec = wrec[cur];
        if (rec >> 63) break;
        d -= w ? w[cur >> 1] : 1u;
    }
}
__global__ void __launch_bounds__(TB) unranked_check_kernel(const uint2* __restrict__ rk, uint64_t ns, uint32_t* __restrict__ flag) {
    uint64_t s = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s < ns && rk[s].y == NONE) *flag = 1u;
}

// ---- circles, cut without the general algorithm.  The ruling-set ranking sees a circle in one of two ways: its splitters form a
// cycle in the splitter list (the jumping does not converge), or it holds no splitter at all and no walk reaches it.  Round 2
// answered both with Wyllie's pointer jumping over ALL states (ceil(log2 ns) rounds of random gathers: ~50 ms at 35 M states,
// a second at 280 M -- for ONE plasmid in the data set).  Here: the minimum sampled FRAGMENT of every splitter cycle by
// pointer jumping on the splitter list only (1/32 of the states; both states of a fragment are sampled together, so the two
// directed cycles of a circle agree on it), and the minimum fragment of a splitter-free circle (a few hundred states at most)
// by walking it from every unreached odd state; the circle is cut at the odd state of that fragment -- one cut per circle --
// and the caller ranks again.  Where a circle is cut does not matter in the join: jcircle_kernel rotates it to the
// reference's cut (canonicalizeCircle, BuildReadQGraph48.cc:375-397).
__global__ void __launch_bounds__(TB) spl_reach_kernel(const unsigned long long* __restrict__ wrec, const uint32_t* __restrict__ spl_state, uint64_t m,
                                                       uint8_t* __restrict__ reach) {
    const uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= m) return;
    uint32_t cur = spl_state[k];
    unsigned long long rec = wrec[cur];
    for (;;) {
        reach[cur] = 1;
        const uint32_t l = (uint32_t)rec;
        if (l == NONE) break;
        cur = l ^ 1u;
        rec = wrec[cur];
        if (rec >> 63) break;
    }
}
__global__ void __launch_bounds__(TB) scyc_init_kernel(const uint32_t* __restrict__ rn_final, const uint32_t* __restrict__ rn_orig,
                                                       const uint32_t* __restrict__ spl_state, uint64_t m, uint32_t* __restrict__ jump, uint32_t* __restrict__ mn) {
    const uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= m) return;
    const bool on = rn_final[k] != NONE;          // still has a successor after ceil(log2 m) + 1 doublings: on a cycle
    jump[k] = on ? rn_orig[k] : NONE;
    mn[k] = on ? spl_state[k] >> 1 : NONE;
}
__device__ __forceinline__ void cut_at(uint32_t* link, uint32_t s, uint8_t* circ, uint32_t* n_cut) {
    const uint32_t partner = link[s];
    link[s] = NONE;
    if (partner != NONE) link[partner] = NONE;
    circ[s] = 1;
    if (partner != NONE) circ[partner] = 1;       // both new terminals
    atomicAdd(n_cut, 1u);
}
__global__ void __launch_bounds__(TB) scyc_cut_kernel(const uint32_t* __restrict__ mn, const uint32_t* __restrict__ spl_state, uint64_t m,
                                                      uint32_t* __restrict__ link, uint32_t* __restrict__ n_cut, uint8_t* __restrict__ circ) {
    const uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (k >= m) return;
    const uint32_t s = spl_state[k];
    if (mn[k] != NONE && (s & 1u) && mn[k] == (s >> 1)) cut_at(link, s, circ, n_cut);
}
__global__ void __launch_bounds__(TB) free_circle_find_kernel(const uint8_t* __restrict__ reach, uint64_t ns, const uint32_t* __restrict__ link,
                                                              uint32_t* __restrict__ list, uint32_t cap, uint32_t* __restrict__ n_list, uint32_t* __restrict__ bad) {
    const uint64_t s0 = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (s0 >= ns || !(s0 & 1ull) || reach[s0]) return;
    // an unreached state lies on a circle without a splitter; every odd state of that circle walks it, the one of the minimum
    // fragment is where it will be cut.  Nothing is cut here: a cut also breaks the circle's OTHER direction, which other threads
    // are walking at this moment -- the states are listed and cut by the next kernel.
    const uint32_t s = (uint32_t)s0;
    uint32_t cur = s, mn = s >> 1;
    for (uint32_t steps = 0;; ++steps) {
        const uint32_t l = link[cur];
        if (l == NONE || steps > (1u << 22)) { atomicOr(bad, 1u); return; }      // not a circle / absurdly long: leave it to the general algorithm
        cur = l ^ 1u;
        if (cur == s) break;
        const uint32_t f = cur >> 1;
        mn = f < mn ? f : mn;
    }
    if (mn == (s >> 1)) { const uint32_t at = atomicAdd(n_list, 1u); if (at < cap) list[at] = s; else atomicOr(bad, 1u); }
}
__global__ void __launch_bounds__(TB) cut_list_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, uint32_t cap, uint32_t* __restrict__ link,
                                                      uint32_t* __restrict__ n_cut, uint8_t* __restrict__ circ) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    const uint32_t n = *n_list < cap ? *n_list : cap;
    if (i < n) cut_at(link, list[i], circ, n_cut);
}

// rn_orig: next splitter of every splitter after the first walk; rn_final: the same after the jumping (NONE unless on a cycle).
// *n_cut = circles cut, *bad = 1: something the fast path does not understand (the caller uses the general algorithm).
static int cut_circles_sparse(snk_ctx* ctx, hipStream_t st, uint32_t* link, uint64_t ns, uint64_t m, const uint32_t* spl_state,
                              const unsigned long long* wrec, const uint32_t* rn_orig, const uint32_t* rn_final, bool jump_converged, uint8_t* circ,
                              uint32_t* n_cut, uint32_t* bad, char* err, size_t errcap) {
    uint32_t* d_flags;
    G_ALLOC(d_flags, uint32_t, 4);
    SNK_HIP_TRY(hipMemsetAsync(d_flags, 0, 16, st));
    // (the reach marks come from the links as they are BEFORE any cut)
    uint8_t* reach;
    G_ALLOC(reach, uint8_t, ns + 1);
    SNK_HIP_TRY(hipMemsetAsync(reach, 0, ns + 1, st));
    if (m) hipLaunchKernelGGL(spl_reach_kernel, dim3(nblk(m)), dim3(TB), 0, st, wrec, spl_state, m, reach);
    if (!jump_converged && m) {
        uint32_t *jump[2], *mn[2];
        for (int b = 0; b < 2; ++b) { G_ALLOC(jump[b], uint32_t, m + 1); G_ALLOC(mn[b], uint32_t, m + 1); }
        hipLaunchKernelGGL(scyc_init_kernel, dim3(nblk(m)), dim3(TB), 0, st, rn_final, rn_orig, spl_state, m, jump[0], mn[0]);
        int max_rounds = 2;
        while ((1ull << (max_rounds - 1)) < m + 1) ++max_rounds;
        int c2 = 0;
        for (int r = 0; r < max_rounds; ++r) {
            hipLaunchKernelGGL(cyc_round_kernel, dim3(nblk(m)), dim3(TB), 0, st, jump[c2], mn[c2], m, jump[c2 ^ 1], mn[c2 ^ 1]);
            c2 ^= 1;
        }
        hipLaunchKernelGGL(scyc_cut_kernel, dim3(nblk(m)), dim3(TB), 0, st, mn[c2], spl_state, m, link, d_flags, circ);
    }
    {
        const uint32_t cap = 1u << 18;
        uint32_t* list;
        G_ALLOC(list, uint32_t, cap);
        hipLaunchKernelGGL(free_circle_find_kernel, dim3(nblk(ns)), dim3(TB), 0, st, reach, ns, (const uint32_t*)link, list, cap, d_flags + 2, d_flags + 1);
        hipLaunchKernelGGL(cut_list_kernel, dim3(cap / TB), dim3(TB), 0, st, (const uint32_t*)list, (const uint32_t*)(d_flags + 2), cap, link, d_flags, circ);
    }
    SNK_HIP_TRY(hipGetLastError());
    uint32_t h[2] = {0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h, d_flags, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    *n_cut = h[0];
    *bad = h[1];
    return SNK_OK;
}

static int rank_lists(snk_ctx* ctx, hipStream_t st, uint32_t* link, uint64_t n, const uint32_t* weights, uint8_t* circ /* per state, nullable */,
                      const uint2** rk_out, uint32_t* n_circles, uint32_t* rounds,
                      char* err, size_t errcap) {
    const uint64_t ns = 2 * n;
    if (ns < 4096 || snk_opt_u32("rank_wyllie", 0))
        return rank_lists_wyllie(ctx, st, link, n, weights, circ, rk_out, n_circles, rounds, err, errcap);
    uint32_t cut_total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
    uint8_t* spl;
    uint32_t *flag32, *sid;
    G_ALLOC(spl, uint8_t, ns + 1);
    G_ALLOC(flag32, uint32_t, ns + 1);
    G_ALLOC(sid, uint32_t, ns + 1);
    SNK_HIP_TRY(hipMemsetAsync(flag32 + ns, 0, 4, st));
    const uint32_t split_mask = (1u << snk_opt_u32("split_log2", 5)) - 1u;
    hipLaunchKernelGGL(spl_mark_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, ns, split_mask, spl, flag32);
    {
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, flag32, sid, 0u, (size_t)(ns + 1), rocprim::plus<uint32_t>(), st));
        void* tmp;
        int rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, flag32, sid, 0u, (size_t)(ns + 1), rocprim::plus<uint32_t>(), st));
    }
    uint32_t m32 = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&m32, sid + ns, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    const uint64_t m = m32;
    uint32_t* spl_state = flag32;      // flag32 is dead after the scan: reuse it for the compacted splitter list
    if (m) hipLaunchKernelGGL(spl_collect_kernel, dim3(nblk(ns)), dim3(TB), 0, st, spl, sid, ns, spl_state);
    uint32_t *rn[2], *rd[2], *rt[2];
    for (int b = 0; b < 2; ++b) { G_ALLOC(rn[b], uint32_t, m + 1); G_ALLOC(rd[b], uint32_t, m + 1); G_ALLOC(rt[b], uint32_t, m + 1); }
    unsigned long long* wrec;
    G_ALLOC(wrec, unsigned long long, ns);
    hipLaunchKernelGGL(spl_pack_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, spl, sid, ns, wrec);
    if (m) hipLaunchKernelGGL(spl_walk1_kernel, dim3(nblk(m)), dim3(TB), 0, st, wrec, spl_state, weights, m, rn[0], rd[0], rt[0]);
    SNK_HIP_TRY(hipGetLastError());
    uint32_t* rn_orig;
    G_ALLOC(rn_orig, uint32_t, m + 1);
    if (m) SNK_HIP_TRY(hipMemcpyAsync(rn_orig, rn[0], m * 4, hipMemcpyDeviceToDevice, st));
    // pointer jumping on the splitter list
    uint32_t* flags;
    G_ALLOC(flags, uint32_t, 4);
    int max_rounds = 2;
    while ((1ull << (max_rounds - 1)) < m + 1) ++max_rounds;
    int cur = 0;
    uint32_t r_done = 0;
    // Rounds are issued in batches without asking the device whether the last one still changed anything (a round over the
    // ~n/32 splitters takes 10-15 us, a read-back 25-30 us of idle device): the first batch covers lists of 2^12 splitters
    // (128 k fragments), and its verdict comes back together with the walk's "every state ranked" check.
    uint2* rk = nullptr;
    bool converged = false, unranked = false;
    G_ALLOC(rk, uint2, ns);
    const int batch0 = (int)snk_opt_u32("rank_round_batch0", 12);
    for (int r = 0; r < max_rounds && !converged;) {
        const int upto = r == 0 ? (batch0 < max_rounds ? batch0 : max_rounds) : max_rounds;
        SNK_HIP_TRY(hipMemsetAsync(flags, 0, 8, st));
        for (; r < upto; ++r) {
            if (m) hipLaunchKernelGGL(rank_round_kernel, dim3(nblk(m)), dim3(TB), 0, st, rn[cur], rd[cur], rt[cur], m, rn[cur ^ 1], rd[cur ^ 1], rt[cur ^ 1],
                                      r + 1 == upto ? flags : flags + 2);      // only the batch's last round reports
            cur ^= 1;
            ++r_done;
        }
        // optimistic: rank the states from what the rounds left (harmless if they had not converged: it is redone)
        SNK_HIP_TRY(hipMemsetAsync(rk, 0xFF, ns * 8, st));
        if (m) hipLaunchKernelGGL(spl_walk2_kernel, dim3(nblk(m)), dim3(TB), 0, st, wrec, spl_state, weights, rd[cur], rt[cur], m, rk);
        hipLaunchKernelGGL(unranked_check_kernel, dim3(nblk(ns)), dim3(TB), 0, st, rk, ns, flags + 1);
        uint32_t h2[2] = {0, 0};
        SNK_HIP_TRY(hipMemcpyAsync(h2, flags, 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        converged = h2[0] == 0;
        unranked = h2[1] != 0;          // states no walk reached: a circle without a splitter
    }
    if (converged && !unranked) {
        *rk_out = rk;
        *n_circles = cut_total;
        *rounds = r_done;
        return SNK_OK;
    }
    // circles.  In the join (circ given) they are cut here and the lists ranked once more; the k-mer level ranking of the global
    // graph stage needs the cut AT the minimum k-mer (it is the reference's cut there): the general algorithm does that.
    if (!circ || attempt == 1) break;
    uint32_t n_cut = 0, bad = 0;
    int rcc = cut_circles_sparse(ctx, st, link, ns, m, spl_state, wrec, rn_orig, rn[cur], converged, circ, &n_cut, &bad, err, errcap);
    if (rcc) return rcc;
    if (bad || n_cut == 0) break;
    cut_total += n_cut;
    }
    {
        uint32_t nc2 = 0;
        int rcw = rank_lists_wyllie(ctx, st, link, n, weights, circ, rk_out, &nc2, rounds, err, errcap);
        *n_circles = cut_total + nc2;
        return rcw;
    }
}

namespace {
// ---- partitioned ranking (sharded runs, owner-side join).  Every rank holds the job's whole link structure (all-gathered),
// but ranking it in full on every rank would cost each of them what the rank-0 funnel cost one.  The ruling-set scheme
// splits naturally: marking and packing are streaming passes (replicated, ~30 B per state); the two walks -- the random
// access part -- are done for a 1/world share of the splitters per rank; what leaves a rank is 16 B per splitter after
// walk 1 (all-gather) and 16 B per visited state after walk 2 (to the state's owner).  The splitter list itself (1/32 of
// the states) is jumped on every rank.  Lists that are circles are not handled here: the caller falls back to the
// replicated ranking (snk_join_rank) when it is told so -- the decision is the same on every rank because it is taken
// from replicated data.
__global__ void __launch_bounds__(TB) spl_walk1p_kernel(const unsigned long long* __restrict__ wrec, const uint32_t* __restrict__ spl_state,
                                                        const uint32_t* __restrict__ w, uint64_t k0, uint64_t cnt, uint4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= cnt) return;
    uint32_t cur = spl_state[k0 + i];
    unsigned long long rec = wrec[cur];
    uint32_t d = 0, nx = NONE, steps = 1;
    for (;;) {
        const uint32_t l = (uint32_t)rec;
        if (l == NONE) { nx = NONE; break; }
        cur = l ^ 1u;
        rec = wrec[cur];
        d += w[cur >> 1];
        if (rec >> 63) { nx = (uint32_t)(rec >> 32) & 0x7FFFFFFFu; break; }
        ++steps;
    }
    out[i] = make_uint4(nx, d, cur, steps);      // cur = the terminal state when nx == NONE
}
__global__ void __launch_bounds__(TB) prank_unzip_kernel(const uint4* __restrict__ all, uint64_t m, uint32_t* __restrict__ rn, uint32_t* __restrict__ rd,
                                                         uint32_t* __restrict__ rt, unsigned long long* __restrict__ total_steps) {
    const uint64_t k = (uint64_t)blockIdx.x * TB + threadIdx.x;
    unsigned long long st = 0;
    if (k < m) { const uint4 v = all[k]; rn[k] = v.x; rd[k] = v.y; rt[k] = v.z; st = v.w; }
    for (int o = 32; o > 0; o >>= 1) st += __shfl_xor(st, o);
    if ((threadIdx.x & 63) == 0 && st) atomicAdd(total_steps, st);
}
__global__ void __launch_bounds__(TB) prank_steps_kernel(const uint4* __restrict__ all, uint64_t k0, uint64_t cnt, uint64_t* __restrict__ steps) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i <= cnt) steps[i] = i < cnt ? all[k0 + i].w : 0ull;
}
// second walk of this rank's share: one 16-byte record (state, distance, terminal, 0) per visited state, at positions known
// from the step counts of the first walk
__global__ void __launch_bounds__(TB) spl_walk2p_kernel(const unsigned long long* __restrict__ wrec, const uint32_t* __restrict__ spl_state,
                                                        const uint32_t* __restrict__ w, const uint32_t* __restrict__ rdist, const uint32_t* __restrict__ rtail,
                                                        uint64_t k0, uint64_t cnt, const uint64_t* __restrict__ pos, uint4* __restrict__ rec_out) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= cnt) return;
    uint32_t cur = spl_state[k0 + i];
    unsigned long long rec = wrec[cur];
    uint32_t d = rdist[k0 + i];
    const uint32_t t = rtail[k0 + i];
    uint64_t p = pos[i];
    for (;;) {
        rec_out[p++] = make_uint4(cur, d, t, 0u);
        const uint32_t l = (uint32_t)rec;
        if (l == NONE) break;
        cur = l ^ 1u;
        rec = wrec[cur];
        if (rec >> 63) break;
        d -= w[cur >> 1];
    }
}
// records to the owners of their states: count / fill with the per-owner sums taken in LDS first, one reservation per owner and PR_TILES tiles
constexpr int PR_TILES = 16;
template <bool FILL>
__global__ void __launch_bounds__(256) prank_route_kernel(const uint4* __restrict__ rec, uint64_t n, const unsigned long long* __restrict__ frag_off, uint32_t world,
                                                          unsigned long long* __restrict__ cnt_or_cur, uint4* __restrict__ out) {
    extern __shared__ unsigned long long dynp[];          // [world] counts -> reserved bases, then [world] u32 local cursors
    uint32_t* lcur = reinterpret_cast<uint32_t*>(dynp + world);
    for (uint32_t r = threadIdx.x; r < world; r += 256) { dynp[r] = 0; lcur[r] = 0; }
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * 256 * PR_TILES + threadIdx.x;
    auto owner_of = [&](unsigned long long g) { uint32_t o = 0; while (o + 1 < world && g >= frag_off[o + 1]) ++o; return o; };
    for (int t = 0; t < PR_TILES; ++t) {
        const uint64_t i = i0 + (uint64_t)t * 256;
        if (i < n) atomicAdd(&dynp[owner_of(rec[i].x >> 1)], 1ull);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < world; r += 256) { const unsigned long long c = dynp[r]; if (c) dynp[r] = atomicAdd(&cnt_or_cur[r], c); }
    if (!FILL) return;
    __syncthreads();
    for (int t = 0; t < PR_TILES; ++t) {
        const uint64_t i = i0 + (uint64_t)t * 256;
        if (i < n) {
            const uint4 v = rec[i];
            const uint32_t owner = owner_of(v.x >> 1);
            out[dynp[owner] + atomicAdd(&lcur[owner], 1u)] = v;
        }
    }
}
__global__ void __launch_bounds__(TB) prank_apply_kernel(const uint4* __restrict__ rec, uint64_t n, unsigned long long state_base, uint64_t n_local_states,
                                                         uint2* __restrict__ rk, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint4 v = rec[i];
    const unsigned long long s = (unsigned long long)v.x - state_base;
    if (s < n_local_states) rk[s] = make_uint2(v.y, v.z);
    else *bad = 1u;
}
}  // namespace

template <int K>
static int graph_impl(snk_ctx* ctx, hipStream_t st, const snk_u128* keys, const uint64_t* vals, uint64_t n,
                      uint32_t do_prune, bool want_unitigs, snk_graph_out* out, char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    if (n == 0) {      // nothing retained: empty but well-formed outputs
        constexpr uint32_t NB0 = 65536;
        G_ALLOC(out->spectrum, unsigned long long, NB0);
        SNK_HIP_TRY(hipMemsetAsync(out->spectrum, 0, NB0 * 8, st));
        out->spectrum_bins = NB0;
        G_ALLOC(out->unitig_off, uint64_t, 2);
        SNK_HIP_TRY(hipMemsetAsync(out->unitig_off, 0, 16, st));
        G_ALLOC(out->unitig_bases, uint8_t, 16);
        G_ALLOC(out->ctx, uint8_t, 16);
        G_ALLOC(out->counts, uint32_t, 4);
        return SNK_OK;
    }
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
    {
        int rc = snk_spectrum(ctx, st, count_out, n, &out->spectrum, &out->spectrum_bins, err, errcap);
        if (rc) return rc;
    }
    if (!want_unitigs) return SNK_OK;

    const uint64_t ns = 2 * n;
    uint32_t* link;
    G_ALLOC(link, uint32_t, ns);
    hipLaunchKernelGGL((link_kernel<K>), dim3(nblk(ns)), dim3(TB), 0, st, keys, ctx_out, nbr, n, link);
    SNK_HIP_TRY(hipGetLastError());

    const uint2* rk = nullptr;
    {
        int rc = rank_lists(ctx, st, link, n, nullptr, nullptr, &rk, &out->n_circles, &out->rank_rounds, err, errcap);
        if (rc) return rc;
    }

    uint8_t* prev;
    G_ALLOC(prev, uint8_t, ns);
    SNK_HIP_TRY(hipMemsetAsync(prev, 0, ns, st));
    hipLaunchKernelGGL((orient_kernel<K>), dim3(nblk(n)), dim3(TB), 0, st, keys, rk, n, prev);
    uint32_t *hflag, *hidx;
    uint64_t *hlen, *hoff;
    G_ALLOC(hflag, uint32_t, n + 1);
    G_ALLOC(hidx, uint32_t, n + 1);
    G_ALLOC(hlen, uint64_t, n + 1);
    G_ALLOC(hoff, uint64_t, n + 1);
    SNK_HIP_TRY(hipMemsetAsync(hflag + n, 0, 4, st));
    SNK_HIP_TRY(hipMemsetAsync(hlen + n, 0, 8, st));
    hipLaunchKernelGGL(head_kernel, dim3(nblk(n)), dim3(TB), 0, st, rk, prev, n, (uint32_t)K, hflag, hlen);
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
    uint64_t h_tot = 0;
    uint32_t h_nu = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_nu, hidx + n, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_tot, hoff + n, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    uint64_t n_unitigs = h_nu, total_bases = h_tot;
    uint64_t *poff, *uoff;
    uint8_t* bases;
    G_ALLOC(poff, uint64_t, ns);
    G_ALLOC(uoff, uint64_t, n_unitigs + 1);
    G_ALLOC(bases, uint8_t, total_bases);
    hipLaunchKernelGGL(head_place_kernel, dim3(nblk(n)), dim3(TB), 0, st, rk, hflag, hidx, hoff, n, poff, uoff);
    SNK_HIP_TRY(hipMemcpyAsync(uoff + n_unitigs, hoff + n, 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((emit_kernel<K>), dim3(nblk(n)), dim3(TB), 0, st, keys, rk, prev, poff, n, bases);
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

// =====================================================================================================================
// Sharded graph stage and fragment join.  The per-rank work (prune, links, fragments) is the bucket-
// local stage of snk_local.hip; here live the pieces that answer membership queries of other ranks and the join:
//   * a side whose single neighbour lives in another chunk (same rank or not) ends its fragment and exports a "half
//     link" (my terminal state -> the neighbour's state).  Two half links that point at each other are a link (the
//     reciprocal-unique rule of BuildReadQGraph48.cc:408-428 checked half on each owner);
//   * the fragments (tada's per-shard sedges, lib/tada/src/debruijn.rs:296-320) are joined (tada's MAIN_ASM_SN
//     build_edges, debruijn.rs:733-776) by list ranking weighted by k-mers, then get the reference's orientation.
// =====================================================================================================================
namespace {

constexpr unsigned long long NONE64 = ~0ull;

__global__ void __launch_bounds__(TB) answer_kernel(const unsigned long long* __restrict__ q, uint64_t nq,
                                                    const snk_u128* __restrict__ keys, const unsigned long long* __restrict__ tab,
                                                    uint64_t mask, uint32_t* __restrict__ ans) {
    uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= nq) return;
    snk_kmer k;
    k.lo = q[3 * t];
    k.hi = q[3 * t + 1];
    int64_t j = index_find(keys, tab, mask, k);
    ans[t] = j >= 0 ? (uint32_t)j : NONE;
}

// answers come back in the order the queries were sent; qoff[r]..qoff[r+1] went to rank r
__global__ void __launch_bounds__(TB) apply_answers_kernel(const unsigned long long* __restrict__ q, const uint32_t* __restrict__ ans,
                                                           uint64_t nq, const unsigned long long* __restrict__ qoff, uint32_t world,
                                                           uint32_t do_prune, uint32_t* __restrict__ ctx_words,
                                                           uint32_t* __restrict__ rq_idx, uint16_t* __restrict__ rq_meta) {
    uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= nq) return;
    unsigned long long m = q[3 * t + 2];
    uint64_t i = (uint32_t)m;
    uint32_t bit = (uint32_t)(m >> 32) & 0xFFu, rev = (uint32_t)(m >> 40) & 1u;
    uint32_t a = ans[t];
    uint32_t rank = 0;
    while (rank + 1 < world && t >= qoff[rank + 1]) ++rank;
    uint32_t side = bit >> 2;
    if (a == NONE) {
        if (do_prune) atomicAnd(&ctx_words[i >> 2], ~((1u << bit) << (8 * (i & 3))));
    } else {
        rq_idx[2 * i + side] = a;
        rq_meta[2 * i + side] = (uint16_t)(rank | (rev << 15));
    }
}

// ---- join (rank 0)
__global__ void __launch_bounds__(TB) jhash_build_kernel(const unsigned long long* __restrict__ hl_self, uint64_t ne,
                                                         unsigned long long* __restrict__ hk, uint32_t* __restrict__ hv, uint64_t mask) {
    uint64_t e = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= ne) return;
    unsigned long long key = hl_self[e] + 1;
    uint64_t slot = snk_mix64(key) & mask;
    for (;;) {
        unsigned long long old = atomicCAS(&hk[slot], 0ull, key);
        if (old == 0ull) { hv[slot] = (uint32_t)e; break; }
        slot = (slot + 1) & mask;
    }
}
__global__ void __launch_bounds__(TB) jmatch_kernel(const unsigned long long* __restrict__ hl_self, const unsigned long long* __restrict__ hl_nb,
                                                    uint64_t ne, const unsigned long long* __restrict__ hk, const uint32_t* __restrict__ hv,
                                                    uint64_t mask, uint32_t* __restrict__ flink) {
    uint64_t e = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= ne) return;
    unsigned long long nb = hl_nb[e];
    uint32_t out = NONE;
    if (nb != NONE64) {
        unsigned long long key = nb + 1;
        uint64_t slot = snk_mix64(key) & mask;
        for (;;) {
            unsigned long long h = hk[slot];
            if (h == 0ull) break;
            if (h == key) {
                uint32_t e2 = __hip_atomic_load(&hv[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hl_nb[e2] == hl_self[e]) out = e2;      // the other owner points back at me: a link
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
    flink[e] = out;
}

// one-GPU runs: every fragment end registered itself under its terminal state (sfrag[state] = 2*fragment + end), so the
// partner of a half link is one load away -- no hash table over the 2F ends.  Entries of non-terminal states are never
// written; a half link only ever names a terminal state, and the back-check below rejects anything else.
__global__ void __launch_bounds__(TB) jmatch_direct_kernel(const unsigned long long* __restrict__ hl_self,
                                                           const unsigned long long* __restrict__ hl_nb, uint64_t ne,
                                                           const uint32_t* __restrict__ sfrag, uint64_t n_states,
                                                           uint32_t* __restrict__ flink) {
    uint64_t e = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= ne) return;
    const unsigned long long nb = hl_nb[e];
    uint32_t out = NONE;
    if (nb != NONE64 && nb < n_states) {
        const uint32_t e2 = sfrag[nb];
        if (e2 < ne && hl_self[e2] == nb && hl_nb[e2] == hl_self[e]) out = e2;
    }
    flink[e] = out;
}

// ---- sharded runs: fragment links decided on the owners.  An end whose half link names a state of rank q asks q (24 bytes:
// target state, my state, my global end id | my local end << 32); q looks the state up in its terminal-state table and
// answers with the global id of the end that sits there and points back, or NONE.
// A workgroup takes RT_TILES tiles of TB items and goes to the per-owner device counters ONCE (count the tiles, reserve, then fill them):
// these cursors are a handful of words on one 64-byte line, atomics on one line are served one at a time (~10 ns each), and with a
// reservation per 256 items the 136 k workgroups of a 17 M-fragment rank WERE these kernels' time (1.65-1.76 ms each on one rank;
// times the number of owners on eight).
constexpr int RT_TILES = 16;
template <bool FILL>
__global__ void __launch_bounds__(TB) jlink_query_kernel(const unsigned long long* __restrict__ hl_self, const unsigned long long* __restrict__ hl_nb,
                                                         uint64_t ne, const unsigned long long* __restrict__ node_off, uint32_t world,
                                                         unsigned long long my_end_base, unsigned long long* __restrict__ qcount_or_cursor,
                                                         unsigned long long* __restrict__ qbuf) {
    extern __shared__ unsigned long long dynl[];          // [world] counts -> reserved bases, then [world] u32 local cursors
    uint32_t* lcur = reinterpret_cast<uint32_t*>(dynl + world);
    for (uint32_t r = threadIdx.x; r < world; r += TB) { dynl[r] = 0; lcur[r] = 0; }
    __syncthreads();
    const uint64_t e0 = (uint64_t)blockIdx.x * TB * RT_TILES + threadIdx.x;
    auto owner_of = [&](unsigned long long nb) { uint32_t o = 0; const unsigned long long node = nb >> 1; while (o + 1 < world && node >= node_off[o + 1]) ++o; return o; };
    for (int t = 0; t < RT_TILES; ++t) {
        const uint64_t e = e0 + (uint64_t)t * TB;
        const unsigned long long nb = e < ne ? hl_nb[e] : NONE64;
        if (nb != NONE64) atomicAdd(&dynl[owner_of(nb)], 1ull);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < world; r += TB) {
        const unsigned long long c = dynl[r];
        if (c) dynl[r] = FILL ? atomicAdd(&qcount_or_cursor[r], c) : (atomicAdd(&qcount_or_cursor[r], c), 0ull);
    }
    if (!FILL) return;
    __syncthreads();
    for (int t = 0; t < RT_TILES; ++t) {
        const uint64_t e = e0 + (uint64_t)t * TB;
        const unsigned long long nb = e < ne ? hl_nb[e] : NONE64;
        if (nb != NONE64) {
            const uint32_t owner = owner_of(nb);
            const unsigned long long slot = dynl[owner] + atomicAdd(&lcur[owner], 1u);
            qbuf[3 * slot + 0] = nb;
            qbuf[3 * slot + 1] = hl_self[e];
            qbuf[3 * slot + 2] = (my_end_base + e) | ((unsigned long long)e << 32);
        }
    }
}
__global__ void __launch_bounds__(TB) jlink_answer_kernel(const unsigned long long* __restrict__ q, uint64_t nq,
                                                          const unsigned long long* __restrict__ hl_self, const unsigned long long* __restrict__ hl_nb,
                                                          uint64_t ne, const uint32_t* __restrict__ sfrag, unsigned long long my_state_base,
                                                          uint64_t n_local_states, unsigned long long my_end_base, uint32_t* __restrict__ ans) {
    const uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= nq) return;
    const unsigned long long target = q[3 * t], asker = q[3 * t + 1];
    uint32_t out = NONE;
    if (target >= my_state_base && target - my_state_base < n_local_states) {
        const uint32_t e2 = sfrag[target - my_state_base];
        if (e2 < ne && hl_self[e2] == target && hl_nb[e2] == asker) out = (uint32_t)(my_end_base + e2);
    }
    ans[t] = out;
}
__global__ void __launch_bounds__(TB) jlink_apply_kernel(const unsigned long long* __restrict__ q, const uint32_t* __restrict__ ans, uint64_t nq,
                                                         uint32_t* __restrict__ flink) {
    const uint64_t t = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= nq) return;
    flink[(uint32_t)(q[3 * t + 2] >> 32)] = ans[t];
}

struct frag_place { uint32_t pid, other; uint64_t N; uint64_t koff; bool rc; };
__device__ __forceinline__ frag_place frag_place_of(const uint2* rk, const uint32_t* nk, uint64_t f) {
    const uint2 a = rk[2 * f], b = rk[2 * f + 1];
    uint32_t tL = a.y, tR = b.y;
    uint32_t dL = a.x, dR = b.x;
    frag_place p;
    p.N = (uint64_t)dL + dR + nk[f];
    if (tL < tR) { p.pid = tL; p.other = tR; p.koff = dL; p.rc = false; }
    else { p.pid = tR; p.other = tL; p.koff = dR; p.rc = true; }
    return p;
}
// Placement of the fragments [f0, f0 + Fl) of a ranked fragment list: the unitig a fragment belongs to is named by the
// smaller of its two terminal states (pid), koff = k-mers in front of it when the unitig is read from that end, rc = it
// is read from its right end.  The fragment whose own left/right state IS pid (koff == 0) heads the unitig; N = k-mers of
// the whole unitig.  Everything the emission needs -- the ranking itself stays behind (sharded runs rank the job's whole
// fragment list and place only their own fragments).
constexpr unsigned long long PL_RC = 1ull << 63;
// rk covers the fragments from rk_f0 on (0: the whole list; sharded runs with the partitioned ranking: this rank's own);
// circ == NULL: no circle was cut
__global__ void __launch_bounds__(TB) jplace_kernel(const uint2* __restrict__ rk, uint64_t rk_f0, const uint32_t* __restrict__ nk, const uint8_t* __restrict__ circ,
                                                    uint64_t f0, uint64_t Fl, uint32_t* __restrict__ pl_pid, unsigned long long* __restrict__ pl_koff,
                                                    unsigned long long* __restrict__ pl_N, uint8_t* __restrict__ pl_circ) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= Fl) return;
    const uint2 a = rk[2 * (f0 + i - rk_f0)], b = rk[2 * (f0 + i - rk_f0) + 1];
    const uint32_t w = nk[f0 + i];
    frag_place p;
    p.N = (uint64_t)a.x + b.x + w;
    if (a.y < b.y) { p.pid = a.y; p.other = b.y; p.koff = a.x; p.rc = false; }
    else { p.pid = b.y; p.other = a.y; p.koff = b.x; p.rc = true; }
    pl_pid[i] = p.pid;
    pl_koff[i] = p.koff | (p.rc ? PL_RC : 0ull);
    pl_N[i] = p.N;
    pl_circ[i] = circ ? circ[p.pid] : 0;
}
// gfid: global id of every fragment (NULL: fragment f is f); a head is the fragment that owns its unitig's terminal state
__global__ void __launch_bounds__(TB) jhead_kernel(const uint32_t* __restrict__ pl_pid, const unsigned long long* __restrict__ pl_koff,
                                                   const unsigned long long* __restrict__ pl_N, const uint32_t* __restrict__ gfid, uint64_t F, uint32_t K,
                                                   uint32_t* __restrict__ hflag, uint64_t* __restrict__ hlen) {
    uint64_t f = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (f >= F) return;
    const uint32_t me = gfid ? gfid[f] : (uint32_t)f;
    bool head = (pl_koff[f] & ~PL_RC) == 0 && (pl_pid[f] >> 1) == me;
    hflag[f] = head ? 1u : 0u;
    hlen[f] = head ? pl_N[f] + K - 1 : 0ull;
}
// poff is indexed by (pid - pid_base): the terminal states of this rank's own fragments (every unitig is emitted by the
// rank that owns its head fragment)
__global__ void __launch_bounds__(TB) jhead_place_kernel(const uint32_t* __restrict__ pl_pid, const uint8_t* __restrict__ pl_circ,
                                                         const uint32_t* __restrict__ hflag,
                                                         const uint32_t* __restrict__ hidx, const uint64_t* __restrict__ hoff,
                                                         uint64_t F, uint32_t pid_base, uint64_t* __restrict__ poff,
                                                         uint64_t* __restrict__ uoff, uint8_t* __restrict__ ucirc,
                                                         const uint32_t* __restrict__ fgroup, uint32_t* __restrict__ ugroup) {
    uint64_t f = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (f >= F || !hflag[f]) return;
    poff[pl_pid[f] - pid_base] = hoff[f];
    uoff[hidx[f]] = hoff[f];
    ucirc[hidx[f]] = pl_circ[f];
    if (fgroup) ugroup[hidx[f]] = fgroup[f];
}
// Base arrays are bytes (one per base) at arbitrary byte offsets.  A lane that moves one byte per instruction makes the copy
// kernels instruction-bound (a wave instruction moves 64 bytes); these move 4 / 16 bytes per lane with dword accesses at byte
// alignment (the device runs in unaligned-access mode; the packed struct tells the compiler so).
struct __attribute__((packed)) u32_any { uint32_t v; };
__device__ __forceinline__ uint32_t ld_u32_any(const uint8_t* p) { return reinterpret_cast<const u32_any*>(p)->v; }
__device__ __forceinline__ void st_u32_any(uint8_t* p, uint32_t v) { reinterpret_cast<u32_any*>(p)->v = v; }
// 16 bytes dst[0..16) = src[0..16), or (rev) the reverse complement of the 16 bytes that END at src_end: dst[j] = src_end[-1-j] ^ 3
__device__ __forceinline__ void copy16(uint8_t* dst, const uint8_t* src, bool rev) {
    if (!rev) {
        const uint32_t a = ld_u32_any(src), b = ld_u32_any(src + 4), c = ld_u32_any(src + 8), d = ld_u32_any(src + 12);
        st_u32_any(dst, a); st_u32_any(dst + 4, b); st_u32_any(dst + 8, c); st_u32_any(dst + 12, d);
    } else {
        const uint32_t a = ld_u32_any(src - 4), b = ld_u32_any(src - 8), c = ld_u32_any(src - 12), d = ld_u32_any(src - 16);
        st_u32_any(dst, __builtin_bswap32(a) ^ 0x03030303u); st_u32_any(dst + 4, __builtin_bswap32(b) ^ 0x03030303u);
        st_u32_any(dst + 8, __builtin_bswap32(c) ^ 0x03030303u); st_u32_any(dst + 12, __builtin_bswap32(d) ^ 0x03030303u);
    }
}
constexpr uint32_t JCH = 4096;            // bases per work item of the unitig copy kernels: 256 lanes x 16
// number of JCH-base chunks of every unitig (work items of the copy kernels)
__global__ void __launch_bounds__(TB) jchunks_kernel(const uint64_t* __restrict__ boff, uint64_t F, uint32_t* __restrict__ nch) {
    uint64_t f = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (f >= F) return;
    nch[f] = (uint32_t)((boff[f + 1] - boff[f] + JCH - 1) / JCH);
}
__global__ void __launch_bounds__(TB) jchunk_owner_kernel(const uint32_t* __restrict__ choff, uint64_t F, uint32_t* __restrict__ owner) {
    uint64_t f = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (f >= F) return;
    if (choff[f + 1] > choff[f]) owner[choff[f]] = (uint32_t)f;
}
// provisional unitig sequences: every fragment copies the bases it alone stands for.
// Fragments are short (K-1 + ~17 bases): 8 lanes per fragment, 32 fragments per workgroup, no per-chunk owner tables.
__global__ void __launch_bounds__(256) jemit_kernel(uint64_t F, const uint64_t* __restrict__ boff, const uint8_t* __restrict__ fbases,
                                                    const uint32_t* __restrict__ pl_pid, const unsigned long long* __restrict__ pl_koff,
                                                    const uint32_t* __restrict__ nk, const uint64_t* __restrict__ poff, uint32_t pid_base,
                                                    uint32_t K, uint8_t* __restrict__ prov) {
    // (a grid-stride loop: eight lanes per fragment are more work items than one launch may have -- 2^32 -- from 2^29 fragments on, and such a
    // launch is cut short without an error: found at 750 M reads on one GPU, tools/r6_full_job.py)
    for (uint64_t f = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3); f < F; f += (uint64_t)gridDim.x * 32) {
    const uint32_t sub = threadIdx.x & 7u;
    const unsigned long long ko = pl_koff[f];
    const bool rc = (ko & PL_RC) != 0;                  // dst[p] = src[len - 1 - p] ^ 3
    // A fragment's first K-1 bases are the last K-1 of the fragment before it in the unitig: only the unitig's first fragment writes
    // them (round 4: every fragment copied all K-1 + ~15 of its bases -- 1.1 GB for 0.27 GB of unitigs).  In destination order the bases
    // that are left are [K-1, len): read forward from src + K-1, or backward from src + len - K (the same first `len` bytes of the code below).
    const uint32_t skip = (ko & ~PL_RC) == 0 ? 0u : K - 1;
    const uint32_t len = nk[f] + K - 1 - skip;
    const uint8_t* src = fbases + boff[f] + (rc ? 0u : skip);
    uint8_t* dst = prov + poff[pl_pid[f] - pid_base] + (ko & ~PL_RC) + skip;
    // bytes up to the first 4-byte boundary of dst, dwords, the last bytes
    uint32_t head = (4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u;
    if (head > len) head = len;
    if (sub < head) dst[sub] = rc ? (uint8_t)(src[len - 1 - sub] ^ 3u) : src[sub];
    const uint32_t ndw = (len - head) >> 2;
    for (uint32_t j = sub; j < ndw; j += 8) {
        const uint32_t o = head + 4 * j;
        const uint32_t v = rc ? (__builtin_bswap32(ld_u32_any(src + len - 4 - o)) ^ 0x03030303u) : ld_u32_any(src + o);
        *reinterpret_cast<uint32_t*>(dst + o) = v;
    }
    const uint32_t t0 = head + 4 * ndw;
    if (sub < len - t0) { const uint32_t q = t0 + sub; dst[q] = rc ? (uint8_t)(src[len - 1 - q] ^ 3u) : src[q]; }
    }
}
// canonical form of every unitig (dna/CanonicalForm.h:35-48) decided on the provisional sequence
__global__ void __launch_bounds__(TB) jform_kernel(const uint64_t* __restrict__ uoff, uint64_t U, const uint8_t* __restrict__ prov,
                                                   uint8_t* __restrict__ urev) {
    uint64_t u = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (u >= U) return;
    const uint8_t* b = prov + uoff[u];
    uint64_t len = uoff[u + 1] - uoff[u];
    uint8_t rev = 0;
    if (len & 1) rev = (b[len / 2] & 2) ? 1 : 0;
    else {
        for (uint64_t i = 0, j = len; i < j; ++i) {
            uint8_t f = b[i], r = (uint8_t)(b[--j] ^ 3);
            if (f < r) break;
            if (r < f) { rev = 1; break; }
        }
    }
    urev[u] = rev;
}
__global__ void __launch_bounds__(TB) jfinal_kernel(const uint64_t* __restrict__ uoff, const uint32_t* __restrict__ uowner_of_chunk,
                                                    const uint32_t* __restrict__ uchoff, const uint8_t* __restrict__ urev,
                                                    const uint8_t* __restrict__ prov, uint8_t* __restrict__ out) {
    const uint32_t item = blockIdx.x;
    const uint32_t u = uowner_of_chunk[item];
    const uint64_t len = uoff[u + 1] - uoff[u];
    const uint64_t p0 = (uint64_t)(item - uchoff[u]) * JCH + 16u * threadIdx.x;
    if (p0 >= len) return;
    const uint64_t o = uoff[u];
    const bool rev = urev[u] != 0;
    if (p0 + 16 <= len) copy16(out + o + p0, rev ? prov + o + len - p0 : prov + o + p0, rev);
    else for (uint64_t p = p0; p < len; ++p) out[o + p] = rev ? (uint8_t)(prov[o + len - 1 - p] ^ 3u) : prov[o + p];
}

__global__ void __launch_bounds__(TB) jcirc_list_kernel(const uint8_t* __restrict__ ucirc, uint64_t U, uint32_t* __restrict__ clist,
                                                        uint32_t* __restrict__ cnt) {
    uint64_t u = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (u < U && ucirc[u]) clist[atomicAdd(cnt, 1u)] = (uint32_t)u;
}
// One workgroup per circular unitig.  prov holds the closed sequence (N k-mers, N+K-1 bases, the last K-1 repeat the
// first K-1) cut at an arbitrary k-mer.  Reference form (canonicalizeCircle, BuildReadQGraph48.cc:375-397): the circle
// starts at its minimum canonical k-mer read forward.  Candidates: the k-mer at ring position j on strand 0, and its
// reverse complement, which sits at position (N-K-j) mod N of the reverse-complemented ring.
template <int K>
__global__ void __launch_bounds__(256) jcircle_kernel(const uint32_t* __restrict__ clist, const uint32_t* __restrict__ ccnt,
                                                      const uint64_t* __restrict__ uoff, uint8_t* __restrict__ prov, uint8_t* __restrict__ tmp) {
    __shared__ uint64_t bhi[256], blo[256];
    __shared__ uint32_t bpos[256];      // rotation << 1 | strand
    __shared__ uint32_t win;
    const uint32_t n_circ = *ccnt;      // the list's length stays on the device: the workgroups stride over it (no read-back for the grid)
    for (uint32_t c = blockIdx.x; c < n_circ; c += gridDim.x) {
    const uint32_t u = clist[c];
    const uint64_t o = uoff[u];
    const uint64_t len = uoff[u + 1] - o;
    const uint64_t N = len - (K - 1);
    uint8_t* ring = prov + o;
    const int tid = threadIdx.x;
    const uint64_t seg = (N + 255) / 256;
    const uint64_t a = (uint64_t)tid * seg, b = a + seg < N ? a + seg : N;
    snk_kmer best;
    best.hi = ~0ull; best.lo = ~0ull;
    uint32_t bp = 0xFFFFFFFFu;
    if (a < N) {
        snk_kmer f;
        f.hi = 0; f.lo = 0;
        for (int q = 0; q < K; ++q) f = snk_kmer_succ<K>(f, ring[a + q]);
        for (uint64_t j = a; j < b; ++j) {
            if (j > a) f = snk_kmer_succ<K>(f, ring[j + K - 1]);
            const snk_kmer r = snk_kmer_rc<K>(f);
            const uint32_t p0 = (uint32_t)j << 1;
            const uint32_t p1 = (uint32_t)((N - (K + j) % N) % N) << 1 | 1u;
            if (snk_kmer_lt(f, best) || (snk_kmer_eq(f, best) && p0 < bp)) { best = f; bp = p0; }
            if (snk_kmer_lt(r, best) || (snk_kmer_eq(r, best) && p1 < bp)) { best = r; bp = p1; }
        }
    }
    bhi[tid] = best.hi; blo[tid] = best.lo; bpos[tid] = bp;
    __syncthreads();
    if (tid == 0) {
        uint32_t w = 0;
        for (int t = 1; t < 256; ++t) {
            const bool lt = bhi[t] < bhi[w] || (bhi[t] == bhi[w] && (blo[t] < blo[w] || (blo[t] == blo[w] && bpos[t] < bpos[w])));
            if (lt) w = t;
        }
        win = bpos[w];
    }
    __syncthreads();
    const uint64_t i = win >> 1;
    const bool strand = win & 1u;
    uint8_t* t = tmp + o;
    for (uint64_t p = tid; p < len; p += 256) {
        if (!strand) t[p] = ring[(i + p) % N];
        else t[p] = (uint8_t)(ring[(2 * N - 1 - (i + p) % N) % N] ^ 3u);
    }
    __threadfence_block();
    __syncthreads();
    for (uint64_t p = tid; p < len; p += 256) ring[p] = t[p];
    __syncthreads();
    }
}

// deterministic output order: unitigs sorted by their first K bases (every k-mer belongs to exactly one unitig)
template <int K>
__global__ void __launch_bounds__(TB) jorder_key_kernel(const uint64_t* __restrict__ uoff, const uint8_t* __restrict__ bases, uint64_t U,
                                                        const uint32_t* __restrict__ ugroup, snk_u128* __restrict__ key,
                                                        uint32_t* __restrict__ idx) {
    uint64_t u = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (u >= U) return;
    const uint8_t* b = bases + uoff[u];
    snk_kmer f;
    f.hi = 0; f.lo = 0;
    for (int q = 0; q < K; ++q) f = snk_kmer_succ<K>(f, b[q]);
    snk_u128 k = ((snk_u128)f.hi << 64) | (snk_u128)f.lo;
    if (ugroup) k = ((snk_u128)ugroup[u] << 96) | (k >> 32);      // grouped runs (K=48: 96 key bits): group-major order
    key[u] = k;
    idx[u] = (uint32_t)u;
}
__global__ void __launch_bounds__(TB) jorder_len_kernel(const uint64_t* __restrict__ uoff, const uint32_t* __restrict__ idx, uint64_t U,
                                                        uint64_t* __restrict__ len) {
    uint64_t r = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (r > U) return;
    len[r] = r < U ? uoff[idx[r] + 1] - uoff[idx[r]] : 0ull;
}
__global__ void __launch_bounds__(256) jorder_copy_kernel(const uint32_t* __restrict__ owner, const uint32_t* __restrict__ choff,
                                                          const uint64_t* __restrict__ noff, const uint64_t* __restrict__ uoff,
                                                          const uint32_t* __restrict__ idx, const uint8_t* __restrict__ in,
                                                          const uint8_t* __restrict__ circ_in, uint8_t* __restrict__ out,
                                                          uint8_t* __restrict__ circ_out, const uint32_t* __restrict__ grp_in,
                                                          uint32_t* __restrict__ grp_out) {
    const uint32_t item = blockIdx.x;
    const uint32_t r = owner[item];
    const uint64_t len = noff[r + 1] - noff[r];
    const uint64_t p0 = (uint64_t)(item - choff[r]) * JCH + 16u * threadIdx.x;
    if (p0 == 0) { circ_out[r] = circ_in[idx[r]]; if (grp_in) grp_out[r] = grp_in[idx[r]]; }
    if (p0 >= len) return;
    const uint8_t* src = in + uoff[idx[r]] + p0;
    uint8_t* dst = out + noff[r] + p0;
    if (p0 + 16 <= len) copy16(dst, src, false);
    else for (uint64_t p = 0; p0 + p < len; ++p) dst[p] = src[p];
}

}  // namespace

// exclusive scan helper (u32 and u64), n+1 entries with a zero sentinel appended by the caller
template <typename T>
static int excl_scan(snk_ctx* ctx, hipStream_t st, const T* in, T* out, size_t count, char* err, size_t errcap) {
    size_t tb = 0;
    SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, in, out, (T)0, count, rocprim::plus<T>(), st));
    void* tmp;
    int rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap);
    if (rc) return rc;
    SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, in, out, (T)0, count, rocprim::plus<T>(), st));
    return SNK_OK;
}
// owner of every chunk from per-owner chunk counts: scatter the owner id at its first chunk, then max-scan
// total_ub != 0: an upper bound of the chunk count the caller knows without asking the device (sum of ceil(len / JCH) <= sum(len) / JCH
// + count): the tables are sized and the copy kernels launched for it -- no read-back; the owner of an item past the true total is the
// last owner, whose item then lies past its length, so the copy kernels' bound check drops it.
static int chunk_owners(snk_ctx* ctx, hipStream_t st, uint32_t* nch /*[count+1], last = 0*/, uint64_t count, uint32_t** choff_out,
                        uint32_t** owner_out, uint32_t* total_out, char* err, size_t errcap, uint64_t total_ub = 0) {
    uint32_t* choff;
    G_ALLOC(choff, uint32_t, count + 1);
    int rc = excl_scan<uint32_t>(ctx, st, nch, choff, count + 1, err, errcap);
    if (rc) return rc;
    uint32_t total = 0;
    if (total_ub) {
        if (total_ub > 0xFFFFFFF0ull) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "join: %llu copy chunks", (unsigned long long)total_ub);
        total = (uint32_t)total_ub;
    } else {
        SNK_HIP_TRY(hipMemcpyAsync(&total, choff + count, 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
    }
    uint32_t* owner;
    G_ALLOC(owner, uint32_t, (uint64_t)total + 1);
    SNK_HIP_TRY(hipMemsetAsync(owner, 0, ((uint64_t)total + 1) * 4, st));
    hipLaunchKernelGGL(jchunk_owner_kernel, dim3(nblk(count)), dim3(TB), 0, st, choff, count, owner);
    if (total) {
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::inclusive_scan((void*)nullptr, tb, owner, owner, (size_t)total, rocprim::maximum<uint32_t>(), st));
        void* tmp;
        if ((rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::inclusive_scan(tmp, tb, owner, owner, (size_t)total, rocprim::maximum<uint32_t>(), st));
    }
    *choff_out = choff;
    *owner_out = owner;
    *total_out = total;
    return SNK_OK;
}

int snk_dist_answer(snk_ctx* ctx, hipStream_t st, snk_dist_graph* g, const void* d_queries, uint64_t nq, void* d_ans, char* err,
                    size_t errcap) {
    if (nq) hipLaunchKernelGGL(answer_kernel, dim3(nblk(nq)), dim3(TB), 0, st, (const unsigned long long*)d_queries, nq, g->keys, g->index, g->index_mask, (uint32_t*)d_ans);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
int snk_dist_apply(snk_ctx* ctx, hipStream_t st, snk_dist_graph* g, const void* d_qbuf, const void* d_ans, uint64_t nq,
                   const unsigned long long* d_qoff, char* err, size_t errcap) {
    if (nq) hipLaunchKernelGGL(apply_answers_kernel, dim3(nblk(nq)), dim3(TB), 0, st, (const unsigned long long*)d_qbuf, (const uint32_t*)d_ans, nq, d_qoff, g->world, g->do_prune, reinterpret_cast<uint32_t*>(g->ctx), g->rq_idx, g->rq_meta);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// rank 0: fragments of every rank -> canonical unitigs
int snk_dist_join(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t F, const uint32_t* nk, const unsigned long long* hl_self,
                  const unsigned long long* hl_nb, const uint64_t* boff, const uint8_t* fbases, uint64_t total_fbases,
                  snk_join_out* out, char* err, size_t errcap, const uint32_t* fgroup, const uint32_t* sfrag, uint64_t n_states,
                  uint32_t* flink_given) {
    memset(out, 0, sizeof *out);
    if (F == 0) return snk_join_emit(ctx, st, K, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, out, err, errcap);
    if (F >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments at the join (%llu)", (unsigned long long)F);
    const uint64_t ne = 2 * F;
    uint32_t* flink = flink_given;          // sharded runs: the links were decided on the owners (jlink_* kernels)
    if (!flink) G_ALLOC(flink, uint32_t, ne);
    if (flink_given) {
    } else if (sfrag) {
        hipLaunchKernelGGL(jmatch_direct_kernel, dim3(nblk(ne)), dim3(TB), 0, st, hl_self, hl_nb, ne, sfrag, n_states, flink);
    } else {
        uint64_t tg = 1024;
        while (tg < 2 * ne) tg <<= 1;
        unsigned long long* hk;
        uint32_t* hv;
        G_ALLOC(hk, unsigned long long, tg);
        G_ALLOC(hv, uint32_t, tg);
        SNK_HIP_TRY(hipMemsetAsync(hk, 0, tg * 8, st));
        hipLaunchKernelGGL(jhash_build_kernel, dim3(nblk(ne)), dim3(TB), 0, st, hl_self, ne, hk, hv, tg - 1);
        hipLaunchKernelGGL(jmatch_kernel, dim3(nblk(ne)), dim3(TB), 0, st, hl_self, hl_nb, ne, hk, hv, tg - 1, flink);
    }
    SNK_HIP_TRY(hipGetLastError());
    uint8_t* circ;     // per fragment-end state: terminal created by cutting a circle
    G_ALLOC(circ, uint8_t, ne + 1);
    SNK_HIP_TRY(hipMemsetAsync(circ, 0, ne + 1, st));
    const uint2* rk;
    int rc = rank_lists(ctx, st, flink, F, nk, circ, &rk, &out->n_circles, &out->rank_rounds, err, errcap);
    if (rc) return rc;
    snk_placement pl;
    if ((rc = snk_join_place(ctx, st, rk, nk, circ, 0, F, &pl, err, errcap))) return rc;
    const uint32_t nc = out->n_circles, rr = out->rank_rounds;
    if ((rc = snk_join_emit(ctx, st, K, F, nk, nullptr, pl.pid, pl.koff, pl.N, pl.circ, 0, ne, boff, fbases, fgroup, out, err, errcap))) return rc;
    out->n_circles = nc;
    out->rank_rounds = rr;
    return SNK_OK;
}

// ranking of a whole fragment list (sharded runs: every rank holds the job's links and k-mer counts) -- rk[2F], circ[2F]
int snk_join_rank(snk_ctx* ctx, hipStream_t st, uint64_t F, const uint32_t* nk, uint32_t* flink, const uint2** rk_out, uint8_t** circ_out,
                  uint32_t* n_circles, uint32_t* rounds, char* err, size_t errcap, uint8_t* circ_given) {
    if (F >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments at the join (%llu)", (unsigned long long)F);
    uint8_t* circ = circ_given;          // given: the marks of circles an earlier (partitioned) attempt already cut in these links
    if (!circ) {
        G_ALLOC(circ, uint8_t, 2 * F + 1);
        SNK_HIP_TRY(hipMemsetAsync(circ, 0, 2 * F + 1, st));
    }
    *circ_out = circ;
    *n_circles = 0;
    *rounds = 0;
    if (F == 0) { *rk_out = nullptr; return SNK_OK; }
    return rank_lists(ctx, st, flink, F, nk, circ, rk_out, n_circles, rounds, err, errcap);
}

int snk_join_place(snk_ctx* ctx, hipStream_t st, const uint2* rk, const uint32_t* nk, const uint8_t* circ, uint64_t f0, uint64_t Fl,
                   snk_placement* pl, char* err, size_t errcap, uint64_t rk_f0) {
    memset(pl, 0, sizeof *pl);
    G_ALLOC(pl->pid, uint32_t, Fl + 1);
    G_ALLOC(pl->koff, unsigned long long, Fl + 1);
    G_ALLOC(pl->N, unsigned long long, Fl + 1);
    G_ALLOC(pl->circ, uint8_t, Fl + 1);
    if (Fl) hipLaunchKernelGGL(jplace_kernel, dim3(nblk(Fl)), dim3(TB), 0, st, rk, rk_f0, nk, circ, f0, Fl, pl->pid, pl->koff, pl->N, pl->circ);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// debugging aid (option join_dbg): bases outside 0..3 in a unitig buffer = bytes no fragment / copy wrote (the buffer is pre-filled with 0xEE)
__global__ void __launch_bounds__(256) jdbg_count_kernel(const uint8_t* __restrict__ b, uint64_t n, unsigned long long* __restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) c += b[i] > 3 ? 1ull : 0ull;
    if (c) atomicAdd(out, c);
}
static int jdbg_count(snk_ctx* ctx, hipStream_t st, const char* what, const uint8_t* b, uint64_t n, char* err, size_t errcap) {
    unsigned long long* d;
    G_ALLOC(d, unsigned long long, 1);
    SNK_HIP_TRY(hipMemsetAsync(d, 0, 8, st));
    hipLaunchKernelGGL(jdbg_count_kernel, dim3(4096), dim3(256), 0, st, b, n, d);
    unsigned long long h = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    fprintf(stderr, "[snk join dbg] %s: %llu of %llu bytes are not bases\n", what, h, (unsigned long long)n);
    return SNK_OK;
}

// Emission of the unitigs whose head fragment is among the F fragments given (one-GPU runs: all of them; sharded runs:
// the fragments routed to this rank, the owner of their unitig's head): heads -> offsets, every fragment copied into
// place, circles rotated to the reference's cut, canonical orientation, deterministic order.
int snk_join_emit(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t F, const uint32_t* nk, const uint32_t* gfid, const uint32_t* pl_pid,
                  const unsigned long long* pl_koff, const unsigned long long* pl_N, const uint8_t* pl_circ, uint32_t pid_base, uint64_t n_pid,
                  const uint64_t* boff, const uint8_t* fbases, const uint32_t* fgroup, snk_join_out* out, char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    if (F == 0) {
        G_ALLOC(out->unitig_off, uint64_t, 1);
        SNK_HIP_TRY(hipMemsetAsync(out->unitig_off, 0, 8, st));
        return SNK_OK;
    }
    int rc;
    uint32_t *hflag, *hidx;
    uint64_t *hlen, *hoff;
    G_ALLOC(hflag, uint32_t, F + 1);
    G_ALLOC(hidx, uint32_t, F + 1);
    G_ALLOC(hlen, uint64_t, F + 1);
    G_ALLOC(hoff, uint64_t, F + 1);
    SNK_HIP_TRY(hipMemsetAsync(hflag + F, 0, 4, st));
    SNK_HIP_TRY(hipMemsetAsync(hlen + F, 0, 8, st));
    hipLaunchKernelGGL(jhead_kernel, dim3(nblk(F)), dim3(TB), 0, st, pl_pid, pl_koff, pl_N, gfid, F, K, hflag, hlen);
    if ((rc = excl_scan<uint32_t>(ctx, st, hflag, hidx, F + 1, err, errcap))) return rc;
    if ((rc = excl_scan<uint64_t>(ctx, st, hlen, hoff, F + 1, err, errcap))) return rc;
    uint32_t h_nu = 0;
    uint64_t h_tot = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_nu, hidx + F, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_tot, hoff + F, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    const uint64_t U = h_nu;
    const uint64_t chunks_ub = U ? h_tot / JCH + U : 0;      // both copy passes: every unitig's ceil(len / JCH), in either order
    uint64_t *poff, *uoff;
    uint8_t *ucirc, *prov, *final_bases, *urev;
    G_ALLOC(poff, uint64_t, n_pid + 1);
    G_ALLOC(uoff, uint64_t, U + 1);
    G_ALLOC(ucirc, uint8_t, U + 1);
    G_ALLOC(urev, uint8_t, U + 1);
    uint32_t* ugroup = nullptr;
    if (fgroup) G_ALLOC(ugroup, uint32_t, U + 1);
    G_ALLOC(prov, uint8_t, h_tot + 1);
    G_ALLOC(final_bases, uint8_t, h_tot + 1);
    const bool jdbg = snk_opt_u32("join_dbg", 0) != 0;
    if (jdbg) { SNK_HIP_TRY(hipMemsetAsync(prov, 0xEE, h_tot + 1, st)); SNK_HIP_TRY(hipMemsetAsync(final_bases, 0xEE, h_tot + 1, st)); fprintf(stderr, "[snk join dbg] F %llu unitigs %llu bases %llu\n", (unsigned long long)F, (unsigned long long)U, (unsigned long long)h_tot); }
    hipLaunchKernelGGL(jhead_place_kernel, dim3(nblk(F)), dim3(TB), 0, st, pl_pid, pl_circ, hflag, hidx, hoff, F, pid_base, poff, uoff, ucirc, fgroup, ugroup);
    SNK_HIP_TRY(hipMemcpyAsync(uoff + U, hoff + F, 8, hipMemcpyDeviceToDevice, st));
    // copy every fragment into place
    hipLaunchKernelGGL(jemit_kernel, dim3((unsigned)std::min<uint64_t>((F + 31) / 32, 1ull << std::min(22u, snk_opt_u32("emit_grid_log2", 22)))), dim3(256), 0, st, F, boff, fbases, pl_pid, pl_koff, nk, poff, pid_base, K, prov);
    if (jdbg && (rc = jdbg_count(ctx, st, "provisional bases after the fragments' copies", prov, h_tot, err, errcap))) return rc;
    // circles that were cut at an arbitrary fragment boundary: rotate to the reference's cut (minimum k-mer, forward)
    {
        uint32_t *clist, *ccnt;
        G_ALLOC(clist, uint32_t, U + 1);
        G_ALLOC(ccnt, uint32_t, 4);
        SNK_HIP_TRY(hipMemsetAsync(ccnt, 0, 4, st));
        hipLaunchKernelGGL(jcirc_list_kernel, dim3(nblk(U)), dim3(TB), 0, st, ucirc, U, clist, ccnt);
        if (U) {
            const unsigned cg = (unsigned)(U < 1024 ? U : 1024);
            if (K == 48) hipLaunchKernelGGL((jcircle_kernel<48>), dim3(cg), dim3(256), 0, st, clist, ccnt, uoff, prov, final_bases);
            else hipLaunchKernelGGL((jcircle_kernel<60>), dim3(cg), dim3(256), 0, st, clist, ccnt, uoff, prov, final_bases);
        }
    }
    hipLaunchKernelGGL(jform_kernel, dim3(nblk(U)), dim3(TB), 0, st, uoff, U, prov, urev);
    uint32_t *unch, *uchoff, *uowner, utotal = 0;
    G_ALLOC(unch, uint32_t, U + 1);
    SNK_HIP_TRY(hipMemsetAsync(unch + U, 0, 4, st));
    hipLaunchKernelGGL(jchunks_kernel, dim3(nblk(U)), dim3(TB), 0, st, uoff, U, unch);
    if ((rc = chunk_owners(ctx, st, unch, U, &uchoff, &uowner, &utotal, err, errcap, chunks_ub))) return rc;
    if (utotal) hipLaunchKernelGGL(jfinal_kernel, dim3(utotal), dim3(256), 0, st, uoff, uowner, uchoff, urev, prov, final_bases);
    SNK_HIP_TRY(hipGetLastError());
    if (jdbg && (rc = jdbg_count(ctx, st, "oriented bases", final_bases, h_tot, err, errcap))) return rc;
    // deterministic order (fragment ids depend on the order in which workgroups reserved their output)
    uint64_t* noff = uoff;
    uint8_t* obases = final_bases;
    uint8_t* ocirc = ucirc;
    uint32_t* ogroup = ugroup;
    if (U > 1) {
        snk_u128 *ok_in, *ok_out;
        uint32_t *oi_in, *oi_out;
        uint64_t* olen;
        G_ALLOC(ok_in, snk_u128, U + 1);
        G_ALLOC(ok_out, snk_u128, U + 1);
        G_ALLOC(oi_in, uint32_t, U + 1);
        G_ALLOC(oi_out, uint32_t, U + 1);
        G_ALLOC(olen, uint64_t, U + 1);
        G_ALLOC(noff, uint64_t, U + 1);
        G_ALLOC(ocirc, uint8_t, U + 1);
        if (ugroup) G_ALLOC(ogroup, uint32_t, U + 1);
        obases = prov;         // the provisional buffer is dead: reuse it for the ordered copy
        if (K == 48) hipLaunchKernelGGL((jorder_key_kernel<48>), dim3(nblk(U)), dim3(TB), 0, st, uoff, final_bases, U, (const uint32_t*)ugroup, ok_in, oi_in);
        else hipLaunchKernelGGL((jorder_key_kernel<60>), dim3(nblk(U)), dim3(TB), 0, st, uoff, final_bases, U, (const uint32_t*)ugroup, ok_in, oi_in);
        {
            size_t tb = 0;
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, ok_in, ok_out, oi_in, oi_out, (size_t)U, 0u, 128u, st));
            void* tmp;
            if ((rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap))) return rc;
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, ok_in, ok_out, oi_in, oi_out, (size_t)U, 0u, 128u, st));
        }
        hipLaunchKernelGGL(jorder_len_kernel, dim3(nblk(U + 1)), dim3(TB), 0, st, uoff, oi_out, U, olen);
        if ((rc = excl_scan<uint64_t>(ctx, st, olen, noff, U + 1, err, errcap))) return rc;
        uint32_t *onch, *ochoff, *oowner, ototal = 0;
        G_ALLOC(onch, uint32_t, U + 1);
        SNK_HIP_TRY(hipMemsetAsync(onch + U, 0, 4, st));
        hipLaunchKernelGGL(jchunks_kernel, dim3(nblk(U)), dim3(TB), 0, st, noff, U, onch);
        if ((rc = chunk_owners(ctx, st, onch, U, &ochoff, &oowner, &ototal, err, errcap, chunks_ub))) return rc;
        if (jdbg) SNK_HIP_TRY(hipMemsetAsync(obases, 0xEE, h_tot + 1, st));
        if (ototal) hipLaunchKernelGGL(jorder_copy_kernel, dim3(ototal), dim3(256), 0, st, oowner, ochoff, noff, uoff, oi_out, final_bases, ucirc, obases, ocirc, (const uint32_t*)ugroup, ogroup);
        SNK_HIP_TRY(hipGetLastError());
        if (jdbg && (rc = jdbg_count(ctx, st, "ordered bases", obases, h_tot, err, errcap))) return rc;
    }
    // nothing is waited for here: the unitigs are stream-ordered results, the step's closing wait is the caller's
    out->n_unitigs = U;
    out->total_bases = h_tot;
    out->unitig_off = noff;
    out->unitig_bases = obases;
    out->unitig_circular = ocirc;
    out->unitig_group = ogroup;
    return SNK_OK;
}

// k-mer spectrum with as many bins as the largest (saturated) count needs -- WriteKmerSpectrum grows its vector to
// count+1 (BuildReadQGraph48.cc:199-216); at least 65536 bins
int snk_spectrum(snk_ctx* ctx, hipStream_t st, const uint32_t* counts, uint64_t n, unsigned long long** bins_out, uint32_t* nbins_out,
                 char* err, size_t errcap) {
    uint32_t h_max = 0;
    if (n) {
        uint32_t* d_max;
        G_ALLOC(d_max, uint32_t, 4);
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::reduce((void*)nullptr, tb, counts, d_max, 0u, (size_t)n, rocprim::maximum<uint32_t>(), st));
        void* tmp;
        int rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::reduce(tmp, tb, counts, d_max, 0u, (size_t)n, rocprim::maximum<uint32_t>(), st));
        SNK_HIP_TRY(hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
    }
    if (h_max > 0xFFFFFFu) h_max = 0xFFFFFFu;
    const uint32_t nbins = h_max + 1 > 65536u ? h_max + 1 : 65536u;
    unsigned long long* bins;
    G_ALLOC(bins, unsigned long long, nbins);
    SNK_HIP_TRY(hipMemsetAsync(bins, 0, (size_t)nbins * 8, st));
    if (n) {
        unsigned g = nblk(n);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(spectrum_kernel, dim3(g), dim3(TB), 0, st, counts, n, bins, nbins);
        SNK_HIP_TRY(hipGetLastError());
    }
    *bins_out = bins;
    *nbins_out = nbins;
    return SNK_OK;
}

// ---- sharded runs: link matching on the owners
int snk_dist_links_query(snk_ctx* ctx, hipStream_t st, bool fill, const snk_frag_out* fr, const unsigned long long* d_node_off, uint32_t world,
                         unsigned long long my_end_base, unsigned long long* d_count_or_cursor, void* d_qbuf, char* err, size_t errcap) {
    const uint64_t ne = 2 * fr->n_frags;
    if (ne == 0) return SNK_OK;
    if (ne >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments on one rank");
    const size_t lds = (size_t)world * 12 + 16;
    if (fill) hipLaunchKernelGGL((jlink_query_kernel<true>), dim3(nblk((ne + RT_TILES - 1) / RT_TILES)), dim3(TB), lds, st, fr->hl_self, fr->hl_nb, ne, d_node_off, world, my_end_base, d_count_or_cursor, (unsigned long long*)d_qbuf);
    else hipLaunchKernelGGL((jlink_query_kernel<false>), dim3(nblk((ne + RT_TILES - 1) / RT_TILES)), dim3(TB), lds, st, fr->hl_self, fr->hl_nb, ne, d_node_off, world, my_end_base, d_count_or_cursor, (unsigned long long*)nullptr);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
int snk_dist_links_answer(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_queries, uint64_t nq, unsigned long long my_state_base,
                          uint64_t n_local_states, unsigned long long my_end_base, void* d_ans, char* err, size_t errcap) {
    if (nq) hipLaunchKernelGGL(jlink_answer_kernel, dim3(nblk(nq)), dim3(TB), 0, st, (const unsigned long long*)d_queries, nq, fr->hl_self, fr->hl_nb,
                               2 * fr->n_frags, fr->sfrag, my_state_base, n_local_states, my_end_base, (uint32_t*)d_ans);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
int snk_dist_links_apply(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_qbuf, const void* d_ans, uint64_t nq, uint32_t** flink_out,
                         char* err, size_t errcap) {
    const uint64_t ne = 2 * fr->n_frags;
    uint32_t* flink;
    G_ALLOC(flink, uint32_t, ne + 2);
    SNK_HIP_TRY(hipMemsetAsync(flink, 0xFF, (ne + 2) * 4, st));
    if (nq) hipLaunchKernelGGL(jlink_apply_kernel, dim3(nblk(nq)), dim3(TB), 0, st, (const unsigned long long*)d_qbuf, (const uint32_t*)d_ans, nq, flink);
    SNK_HIP_TRY(hipGetLastError());
    *flink_out = flink;
    return SNK_OK;
}
// the same for queries / answers that sit in per-destination regions of `cap` items (one-pass routing, snk_shard_step.hip)
int snk_dist_links_apply_regions(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_qbuf, const void* d_ans, uint32_t world, uint64_t cap,
                                 const unsigned long long* counts, uint32_t** flink_out, char* err, size_t errcap) {
    const uint64_t ne = 2 * fr->n_frags;
    uint32_t* flink;
    G_ALLOC(flink, uint32_t, ne + 2);
    SNK_HIP_TRY(hipMemsetAsync(flink, 0xFF, (ne + 2) * 4, st));
    for (uint32_t p = 0; p < world; ++p)
        if (counts[p]) hipLaunchKernelGGL(jlink_apply_kernel, dim3(nblk(counts[p])), dim3(TB), 0, st, (const unsigned long long*)d_qbuf + 3 * cap * p,
                                          (const uint32_t*)d_ans + cap * p, (uint64_t)counts[p], flink);
    SNK_HIP_TRY(hipGetLastError());
    *flink_out = flink;
    return SNK_OK;
}

int snk_prank_begin(snk_ctx* ctx, hipStream_t st, uint64_t F, const uint32_t* nk, uint32_t* link, uint32_t rank, uint32_t world, snk_prank* P, char* err,
                    size_t errcap) {
    memset(P, 0, sizeof *P);
    const uint64_t ns = 2 * F;
    P->ns = ns; P->w = nk; P->link = link;
    if (F >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments at the join (%llu)", (unsigned long long)F);
    uint8_t* spl;
    uint32_t *flag32, *sid;
    G_ALLOC(spl, uint8_t, ns + 1);
    G_ALLOC(flag32, uint32_t, ns + 1);
    G_ALLOC(sid, uint32_t, ns + 1);
    SNK_HIP_TRY(hipMemsetAsync(flag32 + ns, 0, 4, st));
    const uint32_t split_mask = (1u << snk_opt_u32("split_log2", 5)) - 1u;
    if (ns) hipLaunchKernelGGL(spl_mark_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, ns, split_mask, spl, flag32);
    {
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, flag32, sid, 0u, (size_t)(ns + 1), rocprim::plus<uint32_t>(), st));
        void* tmp;
        int rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, flag32, sid, 0u, (size_t)(ns + 1), rocprim::plus<uint32_t>(), st));
    }
    uint32_t m32 = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&m32, sid + ns, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    const uint64_t m = m32;
    P->m = m;
    P->spl_state = flag32;             // flag32 is dead after the scan: reuse it for the compacted splitter list
    if (m) hipLaunchKernelGGL(spl_collect_kernel, dim3(nblk(ns)), dim3(TB), 0, st, spl, sid, ns, P->spl_state);
    G_ALLOC(P->wrec, unsigned long long, ns + 1);
    if (ns) hipLaunchKernelGGL(spl_pack_kernel, dim3(nblk(ns)), dim3(TB), 0, st, link, spl, sid, ns, P->wrec);
    P->k0 = m * rank / world;
    P->k1 = m * (rank + 1) / world;
    const uint64_t cnt = P->k1 - P->k0;
    G_ALLOC(P->w1_share, uint4, cnt + 1);
    if (cnt) hipLaunchKernelGGL(spl_walk1p_kernel, dim3(nblk(cnt)), dim3(TB), 0, st, P->wrec, P->spl_state, nk, P->k0, cnt, P->w1_share);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// w1_all: the first walk's results of all ranks in splitter order.  *circles = 1: some list is a circle (the caller ranks the
// replicated way); else rec_out / n_rec = this rank's share of (state, distance, terminal) records.
int snk_prank_walk(snk_ctx* ctx, hipStream_t st, snk_prank* P, const uint4* w1_all, uint32_t* circles, uint32_t* rounds, char* err, size_t errcap,
                   uint8_t* circ, uint32_t* n_cut_out) {
    const uint64_t m = P->m;
    *circles = 0;
    *rounds = 0;
    P->n_rec = 0;
    uint32_t *rn[2], *rd[2], *rt[2];
    for (int b = 0; b < 2; ++b) { G_ALLOC(rn[b], uint32_t, m + 1); G_ALLOC(rd[b], uint32_t, m + 1); G_ALLOC(rt[b], uint32_t, m + 1); }
    unsigned long long* tot;
    uint32_t* flags;
    G_ALLOC(tot, unsigned long long, 2);
    G_ALLOC(flags, uint32_t, 4);
    SNK_HIP_TRY(hipMemsetAsync(tot, 0, 8, st));
    if (m) hipLaunchKernelGGL(prank_unzip_kernel, dim3(nblk(m)), dim3(TB), 0, st, w1_all, m, rn[0], rd[0], rt[0], tot);
    uint32_t* rn_orig;
    G_ALLOC(rn_orig, uint32_t, m + 1);
    if (m) SNK_HIP_TRY(hipMemcpyAsync(rn_orig, rn[0], m * 4, hipMemcpyDeviceToDevice, st));
    // pointer jumping over the splitter list in batches of rounds: one read-back per batch (flag of the batch's last round +,
    // the first time, the total the first walk reached), not per round; rounds past convergence change nothing
    unsigned long long h_tot = 0;
    uint32_t h_flag = 0;
    int max_rounds = 2;
    while ((1ull << (max_rounds - 1)) < m + 1) ++max_rounds;
    int cur = 0;
    bool converged = false, first = true;
    const int BATCH_R = (int)snk_opt_u32("rank_round_batch", 8);
    for (int r = 0; r < max_rounds && !converged;) {
        int did = 0;
        for (; did < BATCH_R && r < max_rounds && m; ++did, ++r) {
            SNK_HIP_TRY(hipMemsetAsync(flags, 0, 4, st));
            hipLaunchKernelGGL(rank_round_kernel, dim3(nblk(m)), dim3(TB), 0, st, rn[cur], rd[cur], rt[cur], m, rn[cur ^ 1], rd[cur ^ 1], rt[cur ^ 1], flags);
            cur ^= 1;
            ++*rounds;
        }
        if (first) SNK_HIP_TRY(hipMemcpyAsync(&h_tot, tot, 8, hipMemcpyDeviceToHost, st));
        if (m) SNK_HIP_TRY(hipMemcpyAsync(&h_flag, flags, 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        first = false;
        if (m == 0 || h_flag == 0) converged = true;
        if (m == 0) break;
    }
    if (!converged || h_tot != P->ns) {
        // circles: splitters that form a cycle (the jumping did not converge) and / or states no walk reached (a circle without a
        // splitter).  Everything needed to cut them is replicated -- links, splitter list, the first walk of ALL ranks -- so every
        // rank makes the same cuts (cut_circles_sparse) and the caller starts the ranking again: *circles = 2.  1: fall back to the
        // replicated general algorithm (no circ array, or something the fast path does not understand).
        if (n_cut_out) *n_cut_out = 0;
        if (!circ) { *circles = 1; return SNK_OK; }
        uint32_t n_cut = 0, bad = 0;
        int rcc = cut_circles_sparse(ctx, st, P->link, P->ns, m, P->spl_state, P->wrec, rn_orig, rn[cur], converged, circ, &n_cut, &bad, err, errcap);
        if (rcc) return rcc;
        if (n_cut_out) *n_cut_out = n_cut;
        *circles = (bad || n_cut == 0) ? 1u : 2u;
        return SNK_OK;
    }
    const uint64_t cnt = P->k1 - P->k0;
    uint64_t *steps, *pos;
    G_ALLOC(steps, uint64_t, cnt + 1);
    G_ALLOC(pos, uint64_t, cnt + 1);
    hipLaunchKernelGGL(prank_steps_kernel, dim3(nblk(cnt + 1)), dim3(TB), 0, st, w1_all, P->k0, cnt, steps);
    int rc = excl_scan<uint64_t>(ctx, st, steps, pos, cnt + 1, err, errcap);
    if (rc) return rc;
    uint64_t n_rec = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&n_rec, pos + cnt, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    G_ALLOC(P->rec, uint4, n_rec + 1);
    if (cnt) hipLaunchKernelGGL(spl_walk2p_kernel, dim3(nblk(cnt)), dim3(TB), 0, st, P->wrec, P->spl_state, P->w, rd[cur], rt[cur], P->k0, cnt, pos, P->rec);
    SNK_HIP_TRY(hipGetLastError());
    P->n_rec = n_rec;
    return SNK_OK;
}

int snk_prank_route(snk_ctx* ctx, hipStream_t st, snk_prank* P, bool fill, const unsigned long long* d_frag_off, uint32_t world,
                    unsigned long long* d_cnt_or_cur, void* d_out, char* err, size_t errcap) {
    if (!P->n_rec) return SNK_OK;
    const unsigned nb = (unsigned)((P->n_rec + 256 * PR_TILES - 1) / (256 * PR_TILES));
    if (fill) hipLaunchKernelGGL((prank_route_kernel<true>), dim3(nb), dim3(256), world * 12ull + 16, st, P->rec, P->n_rec, d_frag_off, world, d_cnt_or_cur, (uint4*)d_out);
    else hipLaunchKernelGGL((prank_route_kernel<false>), dim3(nb), dim3(256), world * 12ull + 16, st, P->rec, P->n_rec, d_frag_off, world, d_cnt_or_cur, (uint4*)nullptr);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// the records that arrived -> rk of this rank's own states [state_base, state_base + n_local_states)
int snk_prank_apply(snk_ctx* ctx, hipStream_t st, const void* d_rec, uint64_t n, unsigned long long state_base, uint64_t n_local_states,
                    const uint2** rk_out, char* err, size_t errcap) {
    uint2* rk;
    uint32_t* flag;
    G_ALLOC(rk, uint2, n_local_states + 1);
    G_ALLOC(flag, uint32_t, 4);
    SNK_HIP_TRY(hipMemsetAsync(rk, 0xFF, (n_local_states + 1) * 8, st));
    SNK_HIP_TRY(hipMemsetAsync(flag, 0, 8, st));
    if (n) hipLaunchKernelGGL(prank_apply_kernel, dim3(nblk(n)), dim3(TB), 0, st, (const uint4*)d_rec, n, state_base, n_local_states, rk, flag);
    if (n_local_states) hipLaunchKernelGGL(unranked_check_kernel, dim3(nblk(n_local_states)), dim3(TB), 0, st, (const uint2*)rk, n_local_states, flag + 1);
    uint32_t h[2] = {0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h, flag, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if (h[0] || h[1]) return snk_fail(SNK_E_INTERNAL, err, errcap, "partitioned ranking: %s", h[0] ? "a record for a foreign state arrived" : "a local state was not ranked");
    *rk_out = rk;
    return SNK_OK;
}
