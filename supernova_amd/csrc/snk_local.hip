// snk_local.hip -- bucket-local graph stage: adjacency prune, links and unitig fragments inside the CU.
//
// What it replaces: the same reference code as snk_graph.hip (recomputeAdjacencies, kmers/ReadPather.h:346-385;
// EdgeBuilder, paths/long/BuildReadQGraph48.cc:327-541) -- structured like tada, which assembles every shard on
// its own ("sedges", lib/tada/src/debruijn.rs:147-320) and joins the pieces afterwards (build_edges, :539-776).
//
// Why: the retained k-mers of one minimiser bucket leave the count kernel as one contiguous chunk (<= 1216 k-mers,
// ~100 on average) and ~94 % of all de Bruijn adjacencies connect two k-mers of the same bucket (consecutive k-mers
// of a read share their minimiser).  The global formulation (snk_graph.hip: sort, HBM hash index, one random probe
// per context bit, list ranking over all 2n states) pays HBM-random-access prices for what is bucket-local work.
// Here a workgroup loads one chunk into LDS and does, without leaving the CU:
//   L1  membership of every neighbour through an LDS hash of the chunk; neighbours that are not in the chunk are
//       either provably absent (their minimiser bucket -- and sub-pass -- is this one) or "pending";
//   G   only k-mers with a pending bit (the bucket boundary, ~12 %) enter a global HBM index; pending bits are
//       resolved with one probe each;
//   L2  reciprocal-unique links inside the chunk, maximal local paths ("fragments") by short serial walks in LDS,
//       local smooth circles cut at their minimum k-mer; every fragment is written once (pid -> other orientation)
//       together with its two half links (the single remote neighbour of each end, if any).
// The fragments (one per ~17 k-mers) are then joined by the code that joins the per-rank fragments of the
// multi-GPU path (snk_dist_join): hash match of mutual half links, list ranking weighted by k-mers, copy into
// place, reference orientation, circles rotated to their minimum k-mer.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <type_traits>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"
#include "snk_graph.h"
#include "snk_stages.h"

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr unsigned long long NONE64 = ~0ull;
constexpr uint16_t NONE16 = 0xFFFFu;
constexpr int TB = 256;
inline unsigned nblk(uint64_t n) { return (unsigned)((n + TB - 1) / TB); }

struct chunk_src {
    const uint32_t* chunk_n;
    const uint32_t* chunk_base;
    const uint4* extra;
    const unsigned long long* region_off;
    uint32_t NB, n_extra, n_regions;
    const uint8_t* span;           // [NB] buckets merged into the chunk that starts here (bl_chunk_merge_kernel), or NULL
};
struct chunk_t {
    uint32_t bucket, n, lg, id;
    uint32_t span;                 // > 1: the chunk holds the survivors of `span` consecutive buckets of its count workgroup (bucket, bucket + G, ...)
    uint64_t base;
};
// The count kernel leaves its survivors in per-workgroup regions (region r = bucket % n_regions holds region_cursor[r] entries at
// r * region_cap).  The prune reads every chunk exactly once anyway: it takes the chunk from region space and writes the keys to their
// dense positions itself -- the separate region compaction (24 B read + 24 B written per k-mer, 2.9 ms at the bench) is gone, the
// (count << 8 | context) words never exist densely (they leave the prune as counts[] and ctx[]).  keys_r == NULL: the table is dense.
struct bl_regions {
    const snk_u128* keys_r;
    const uint64_t* vals_r;
    const uint64_t* desc_src;      // [nchunks] region-space position of every chunk
    snk_u128* keys_dense;
};
// sharded runs: this rank owns global buckets [bucket_base, bucket_base + NBl); NBl == 0 -> one GPU owns everything
struct bl_shard {
    uint32_t bucket_base, NBl, me;
    uint32_t G;                  // bucket stride of a count workgroup (= the table's regions): the buckets of a merged chunk lie G apart
    uint8_t* premote;            // [n] pending bits whose k-mer belongs to another rank, or NULL
};
// Small chunks cost the chunk kernels their fixed part (table clear, barriers, an under-filled wave): a count workgroup writes the survivors
// of its buckets b, b + G, b + 2G, ... behind each other into its region, so neighbouring unsplit buckets ARE one contiguous run of k-mers --
// they are handed to the graph stage as one chunk while the run fits the one-wave kernels (`cap` k-mers).  More neighbours are found inside
// a chunk, never fewer; a miss whose bucket is one of the chunk's is absent as before.  An empty bucket ends a run (it may be a split or
// hot bucket, whose survivors lie elsewhere).  One thread per region; error-rich reads: 5.6 M buckets of ~49 survivors.
// (the table's own chunk_n is read only: the merged sizes go to `merged`, which this graph stage's descriptors are made from -- a second
// prune over the same table, or any later reader of chunk_n, sees the table as the count kernel left it: ADVICE r5)
__global__ void __launch_bounds__(256) bl_chunk_merge_kernel(const uint32_t* __restrict__ chunk_n, const uint32_t* __restrict__ chunk_base, uint32_t NB, uint32_t G,
                                                             uint32_t cap, uint8_t* __restrict__ span, uint32_t* __restrict__ merged) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= G) return;
    uint32_t head = NONE, total = 0, m = 0, expect = 0;
    for (uint32_t b = r; b < NB; b += G) {
        const uint32_t n = chunk_n[b];
        span[b] = 1;
        merged[b] = n;
        if (n == 0) { head = NONE; continue; }
        const uint32_t base = chunk_base[b];
        if (head != NONE && base == expect && total + n <= cap && m < 255u) {
            total += n; ++m;
            merged[head] = total; merged[b] = 0; span[head] = (uint8_t)m;
        } else { head = b; total = n; m = 1; }
        expect = base + n;
    }
}
// one 16-byte record per chunk (dense position, k-mers, bucket, split_lg << 24 | split_id -- or 1 << 31 | merged buckets): every chunk
// kernel starts with ONE load instead of a chain of three dependent ones
__global__ void __launch_bounds__(256) bl_chunk_desc_kernel(chunk_src cs, uint32_t nchunks, uint4* __restrict__ desc, uint64_t region_cap, uint64_t* __restrict__ desc_src) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t bucket, n, off = 0, meta = 0, region;
    if (c < cs.NB) {
        bucket = c; n = cs.chunk_n[c];
        if (n) off = cs.chunk_base[c];
        region = bucket % cs.n_regions;
        if (cs.span && n && cs.span[c] > 1) meta = 0x80000000u | cs.span[c];
    } else {
        const uint4 e = cs.extra[c - cs.NB];
        bucket = e.x; off = e.y; n = e.z & 0xFFFu; region = e.z >> 12; meta = e.w;      // (a hot bucket's classes are counted by other workgroups than the bucket's own: snk_hot.hip)
    }
    const uint64_t base = cs.region_off[region] + off;
    desc[c] = make_uint4((uint32_t)base, n, bucket, meta);
    if (desc_src) desc_src[c] = (uint64_t)region * region_cap + off;      // where the chunk lies in region space (deferred compaction)
}
__device__ __forceinline__ chunk_t chunk_get(const uint4* __restrict__ desc, uint32_t c) {
    const uint4 d = desc[c];
    chunk_t k;
    k.base = d.x; k.n = d.y; k.bucket = d.z; k.lg = d.w >> 24; k.id = d.w & 0xFFFFFFu; k.span = 1;
    if (d.w >> 31) { k.lg = 0; k.id = 0; k.span = d.w & 0xFFFFu; }
    return k;
}

__device__ __forceinline__ snk_kmer load_key(const snk_u128* keys, uint64_t i) {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(keys + i);
    snk_kmer k;
    k.lo = p[0];
    k.hi = p[1];
    return k;
}
__device__ __forceinline__ uint32_t lhash(uint64_t hi, uint64_t lo) {
    uint32_t x = (uint32_t)(hi >> 32) * 0x9E3779B1u ^ (uint32_t)hi * 0x85EBCA77u ^ (uint32_t)(lo >> 32) * 0xC2B2AE3Du ^
                 (uint32_t)lo * 0x27D4EB2Fu;
    return snk_mix32(x);
}
template <int K>
__device__ __forceinline__ uint32_t oriented_base(snk_kmer k, bool rc, int idx) {
    return rc ? (snk_kmer_base<K>(k, K - 1 - idx) ^ 3u) : snk_kmer_base<K>(k, idx);
}

// ---------------------------------------------------------------------------------------------- L1: local prune
template <int K, int CAP, int T, bool BIG, bool GR, int MM>
__device__ __forceinline__ void bl_prune_chunk(const uint32_t c, const uint4* __restrict__ desc, uint32_t NB, bl_shard sh, const bl_regions& rg,
                                                     const snk_u128* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                     uint32_t do_prune, uint8_t* __restrict__ ctx_out,
                                                     uint32_t* __restrict__ count_out, uint8_t* __restrict__ pend_out,
                                                     uint32_t* __restrict__ nbr_out, uint32_t* __restrict__ nbnd,
                                                     uint32_t* __restrict__ biglist, uint32_t* __restrict__ nbig,
                                                     unsigned long long* __restrict__ gindex, uint64_t gmask) {
    constexpr int HT = CAP <= 256 ? 512 : 4096;
    // at K=48 without groups only the top 32 bits of the low key word are sequence (LDS per chunk = waves per CU)
    typedef typename std::conditional<(K <= 48 && !GR), uint32_t, uint64_t>::type klo_w;
    constexpr int KLS = (K <= 48 && !GR) ? 32 : 0;
    __shared__ uint64_t khi[CAP];
    __shared__ klo_w klo[CAP];
    __shared__ uint32_t ht[HT];
    __shared__ uint32_t bcnt;
    const int tid = threadIdx.x;
    const chunk_t ch = chunk_get(desc, c);
    if (ch.n == 0) return;
    if (!BIG && ch.n > (uint32_t)CAP) {
        if (tid == 0) biglist[atomicAdd(nbig, 1u)] = c;
        return;
    }
    const uint32_t n = ch.n;
    // where the chunk lies: dense, or still in its count region (then this kernel is what compacts it)
    uint64_t src = ch.base;
    if (rg.keys_r) { src = rg.desc_src[c]; keys = rg.keys_r; vals = rg.vals_r; }
    __syncthreads();      // the previous chunk of this workgroup is done with the LDS arrays
    for (int s = tid; s < HT; s += T) ht[s] = 0;
    if (tid == 0) bcnt = 0;
    constexpr int NPT = (CAP + T - 1) / T;             // nodes per thread
    uint64_t myv[NPT];
    snk_kmer kq[NPT];
#pragma unroll
    for (int q = 0; q < NPT; ++q) {          // all loads first, then the LDS stores
        const uint32_t i = tid + q * T;
        myv[q] = 0;
        if (i < n) {
            kq[q] = load_key(keys, src + i);
            myv[q] = vals[src + i];
        }
    }
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
        const uint32_t i = tid + q * T;
        if (i < n) {
            khi[i] = kq[q].hi;
            klo[i] = (klo_w)(kq[q].lo >> KLS);
            if (rg.keys_r) { uint64_t* d = reinterpret_cast<uint64_t*>(rg.keys_dense + ch.base + i); d[0] = kq[q].lo; d[1] = kq[q].hi; }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {
        uint32_t slot = lhash(khi[i], (uint64_t)klo[i] << KLS) & (HT - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&ht[slot], 0u, i + 1u);
            if (old == 0u) break;
            slot = (slot + 1) & (HT - 1);
        }
    }
    __syncthreads();
    const uint32_t split_mask = (1u << ch.lg) - 1u;
    uint32_t mybnd = 0;
    __shared__ uint16_t mlist[8 * T];        // misses of the current batch of T nodes: thread << 3 | bit
    __shared__ uint32_t resL[T];             // per thread: keep bits | pending bits << 8 decided for its misses
    __shared__ uint32_t mcnt;
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
        const uint32_t i0 = q * T;
        if (i0 >= n) break;
        const uint32_t i = i0 + tid;
        const bool act = i < n;
        snk_kmer k;
        k.hi = act ? khi[i] : 0ull;
        k.lo = act ? ((uint64_t)klo[i] << KLS) : 0ull;
        uint64_t tag = 0;                                    // grouped runs: the group id in the low 32 key bits
        if (GR) { tag = k.lo & 0xFFFFFFFFull; k.lo &= ~0xFFFFFFFFull; }
        const uint64_t v = myv[q];
        const uint32_t c0 = (uint32_t)(v & 0xFFu);
        uint32_t keep = 0, miss = 0, nb0 = NONE, nb1 = NONE;
        // the reverse complement of a neighbour follows from the k-mer's own in one shift: rc(succ(k, b)) =
        // pred(rc(k), 3 - b) and rc(pred(k, b)) = succ(rc(k), 3 - b) -- one 128-bit reversal per node, not per neighbour
        const snk_kmer kr = snk_kmer_rc<K>(k);
        for (uint32_t rem = c0; rem; rem &= rem - 1) {      // a k-mer has ~2 set bits: iterate over them, not over all 8
            const uint32_t bit = __ffs(rem) - 1;
            const snk_kmer y = bit < 4 ? snk_kmer_succ<K>(k, bit) : snk_kmer_pred<K>(k, bit - 4);
            const snk_kmer r = bit < 4 ? snk_kmer_pred<K>(kr, 3u - bit) : snk_kmer_succ<K>(kr, 7u - bit);
            const bool rev = snk_kmer_lt(r, y);
            snk_kmer cy = rev ? r : y;
            if (GR) cy.lo |= tag;                            // neighbours live in the same group
            int32_t j = -1;
            uint32_t slot = lhash(cy.hi, cy.lo) & (HT - 1);
            for (;;) {
                const uint32_t e = ht[slot];
                if (e == 0u) break;
                if (khi[e - 1] == cy.hi && ((uint64_t)klo[e - 1] << KLS) == cy.lo) { j = (int32_t)(e - 1); break; }
                slot = (slot + 1) & (HT - 1);
            }
            if (j >= 0) {
                keep |= 1u << bit;
                if (bit < 4) nb0 = ((uint32_t)j << 1) | (rev ? 1u : 0u); else nb1 = ((uint32_t)j << 1) | (rev ? 1u : 0u);
            } else miss |= 1u << bit;
        }
        // Neighbours that are not in this chunk: absent for certain iff this very sub-pass would have counted them, i.e.
        // their minimiser bucket is this one (and their split hash this sub-pass's).  A neighbour shares all but one of
        // its M-mers with k, so its minimum ordering key = min(minimum over the K-M shared ones, key of the new M-mer).
        // The misses of the batch (~0.15 per k-mer) are gathered into a dense list and classified by FOUR lanes each,
        // which split the shared M-mers between them -- instead of every lane of the wave idling through the scan
        // of the few lanes that have a miss.
        if (tid == 0) mcnt = 0;
        resL[tid] = 0;
        __syncthreads();
        uint32_t pos = snk_wave_alloc(&mcnt, (uint32_t)__popc(miss));
        if (miss) {
            for (uint32_t bit = 0; bit < 8; ++bit)
                if (miss & (1u << bit)) mlist[pos++] = (uint16_t)((tid << 3) | bit);
        }
        __syncthreads();
        const uint32_t nm = mcnt;
        constexpr int SH = K - MM;               // shared M-mers of a k-mer and its neighbour
        constexpr int PER = (SH + 3) / 4;
        for (uint32_t it0 = 0; it0 < nm; it0 += T / 4) {
            const uint32_t item = it0 + (tid >> 2), sub = tid & 3u;
            const bool on = item < nm;
            const uint32_t ent = on ? mlist[item] : 0u;
            const uint32_t th = ent >> 3, bit = ent & 7u;
            const uint32_t ni = i0 + th;
            snk_kmer kk;
            kk.hi = on ? khi[ni] : 0ull;
            kk.lo = on ? ((uint64_t)klo[ni] << KLS) : 0ull;
            uint64_t ktag = 0;
            if (GR) { ktag = kk.lo & 0xFFFFFFFFull; kk.lo &= ~0xFFFFFFFFull; }
            const int first = bit < 4 ? 1 : 0;   // successors share positions 1..K-M, predecessors 0..K-M-1
            if (do_prune & 2u) {                 // measurement aid (SNK_BL_NOCLASSIFY): every miss is pending, the index resolves them all -- same result
                if (on && sub == 0) atomicOr(&resL[th], (1u << bit) | (0x100u << bit));
                continue;
            }
            uint32_t mk = 0xFFFFFFFFu;
            for (int t = 0; t < PER; ++t) {
                const int sp = (int)sub * PER + t;
                if (sp < SH) {
                    const int pp = first + sp;
                    uint64_t w;
                    if (pp == 0) w = kk.hi;
                    else if (pp < 32) w = (kk.hi << (2 * pp)) | (kk.lo >> (64 - 2 * pp));
                    else if (pp == 32) w = kk.lo;
                    else w = kk.lo << (2 * pp - 64);
                    const uint32_t key = snk_mmer_key_top<MM>(w);
                    mk = key < mk ? key : mk;
                }
            }
            { const uint32_t o = __shfl_xor(mk, 1); mk = o < mk ? o : mk; }
            { const uint32_t o = __shfl_xor(mk, 2); mk = o < mk ? o : mk; }
            if (on && sub == 0) {
                const snk_kmer y = bit < 4 ? snk_kmer_succ<K>(kk, bit) : snk_kmer_pred<K>(kk, bit - 4);
                uint64_t wn;     // the one M-mer of the neighbour that k does not have (left-aligned)
                if (bit < 4) {
                    constexpr int pl = K - MM;
                    wn = pl < 32 ? ((y.hi << (2 * pl)) | (y.lo >> (64 - 2 * pl))) : (pl == 32 ? y.lo : (y.lo << (2 * pl - 64)));
                } else wn = y.hi;
                const uint32_t nkey = snk_mmer_key_top<MM>(wn);
                if (nkey < mk) mk = nkey;
                const uint32_t gb = snk_bucket_of_key(mk ^ (GR ? snk_group_mix((uint32_t)ktag) : 0u), NB);   // (global) bucket of the neighbour
                const uint32_t db = gb - (sh.bucket_base + ch.bucket);          // (wraps below the chunk's first bucket: then no multiple of G below span * G... checked)
                bool here = ch.span == 1 ? db == 0u : (gb >= sh.bucket_base + ch.bucket && db % sh.G == 0u && db / sh.G < ch.span);
                const bool remote = sh.NBl && gb / sh.NBl != sh.me;              // lives (if anywhere) on another rank
                if (here && ch.lg) {
                    const snk_kmer r = snk_kmer_rc<K>(y);
                    snk_kmer cn = snk_kmer_lt(r, y) ? r : y;
                    if (GR) cn.lo |= ktag;
                    uint32_t h1, h2;
                    snk_kmer_hash_count<(K > 48) || GR>(cn, &h1, &h2);       // the count kernel's split function
                    here = (h2 & split_mask) == ch.id;
                }
                if (here) { if (!(do_prune & 1u)) atomicOr(&resL[th], 1u << bit); }
                else atomicOr(&resL[th], (1u << bit) | (0x100u << bit) | (remote ? (0x10000u << bit) : 0u));
            }
        }
        __syncthreads();
        const uint32_t res = resL[tid];
        keep |= res & 0xFFu;
        const uint32_t pm = (res >> 8) & 0xFFu;
        __syncthreads();
        if (!act) continue;
        const uint64_t gi = ch.base + i;
        ctx_out[gi] = (uint8_t)keep;
        pend_out[gi] = (uint8_t)pm;
        if (sh.premote) sh.premote[gi] = (uint8_t)(res >> 16);
        count_out[gi] = (uint32_t)(v >> 8);
        // (chunk-local index << 1 | rev fits 12 bits -- a chunk holds at most 1280 k-mers --: one word per k-mer for both sides, 0xFFF = none;
        // bit 15: the k-mer is its own reverse complement -- the fragment kernel's sizing pass then needs no keys at all)
        nbr_out[gi] = (nb0 == NONE ? 0x0FFFu : nb0) | (snk_kmer_eq(k, kr) ? 0x8000u : 0u) | ((nb1 == NONE ? 0x0FFFu : nb1) << 16);
        if (pm) {
            ++mybnd;
            if (gindex) {
                // boundary k-mers enter the HBM index right here (round 2: a separate pass over all pend bytes and keys, 2.2 ms);
                // the index was sized from the previous call's boundary count -- the host falls back to the separate pass if it is too full
                snk_kmer kk;
                kk.hi = khi[i];
                kk.lo = (uint64_t)klo[i] << KLS;
                uint32_t h1, h2;
                snk_kmer_hash2(kk, &h1, &h2);
                uint64_t slot = (((uint64_t)h1 << 32) | h2) & gmask;
                const unsigned long long ent = ((unsigned long long)h1 << 32) | (unsigned long long)(gi + 1);
                // gindex[gmask + 1] = "give up" flag: the table was sized from the PREVIOUS call's boundary count; on other data (a first
                // call with many split chunks: most neighbours lie in another hash class) it can be several times too small, and 10^8
                // inserts probing 4096 slots each into a full table took 6.4 s (round 4, config.robust first_call_phases).  An insert
                // that finds no slot within 64 probes raises the flag, everybody stops inserting, the host runs the exact-size build pass.
                if (gindex[gmask + 1] == 0ull) {
                    bool placed = false;
                    for (uint32_t tries = 0; tries < 64; ++tries) {
                        const unsigned long long old = atomicCAS(&gindex[slot], 0ull, ent);
                        if (old == 0ull) { placed = true; break; }
                        slot = (slot + 1) & gmask;
                    }
                    if (!placed) gindex[gmask + 1] = 1ull;
                }
            }
        }
    }
    snk_wave_add(&bcnt, mybnd);
    __syncthreads();
    if (tid == 0) nbnd[c] = bcnt;
}
template <int K, int CAP, int T, bool BIG, bool GR, int MM>
__global__ void __launch_bounds__(T) bl_prune_kernel(const uint4* __restrict__ desc, uint32_t NB, bl_shard sh, bl_regions rg, const uint32_t* __restrict__ biglist_in,
                                                     uint32_t nchunks, uint32_t cpw,
                                                     const snk_u128* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                     uint32_t do_prune, uint8_t* __restrict__ ctx_out,
                                                     uint32_t* __restrict__ count_out, uint8_t* __restrict__ pend_out,
                                                     uint32_t* __restrict__ nbr_out, uint32_t* __restrict__ nbnd,
                                                     uint32_t* __restrict__ biglist, uint32_t* __restrict__ nbig, const uint32_t* __restrict__ n_dev,
                                                     unsigned long long* __restrict__ gindex, uint64_t gmask) {
    if (BIG && n_dev) {
        // the list's length is still on the device (no read-back between the two prune launches): a fixed grid strides over it
        const uint32_t nb = *n_dev;
        for (uint32_t w = blockIdx.x; w < nb; w += gridDim.x)
            bl_prune_chunk<K, CAP, T, BIG, GR, MM>(biglist_in[w], desc, NB, sh, rg, keys, vals, do_prune, ctx_out, count_out, pend_out, nbr_out, nbnd, biglist, nbig, gindex, gmask);
        return;
    }
    for (uint32_t r = 0; r < cpw; ++r) {
        const uint32_t w = blockIdx.x * cpw + r;
        if (w >= nchunks) return;
        bl_prune_chunk<K, CAP, T, BIG, GR, MM>(BIG ? biglist_in[w] : w, desc, NB, sh, rg, keys, vals, do_prune, ctx_out, count_out, pend_out, nbr_out,
                                       nbnd, biglist, nbig, gindex, gmask);
    }
}

// ---------------------------------------------------------------------------------------------- G: boundary index
__global__ void __launch_bounds__(TB) bl_index_build_kernel(const snk_u128* __restrict__ keys, const uint8_t* __restrict__ pend,
                                                            uint64_t n, unsigned long long* __restrict__ tab, uint64_t mask) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n || !pend[i]) return;
    const snk_kmer k = load_key(keys, i);
    uint32_t h1, h2;
    snk_kmer_hash2(k, &h1, &h2);
    uint64_t slot = (((uint64_t)h1 << 32) | h2) & mask;
    const unsigned long long ent = ((unsigned long long)h1 << 32) | (unsigned long long)(i + 1);
    for (;;) {
        const unsigned long long old = atomicCAS(&tab[slot], 0ull, ent);
        if (old == 0ull) break;
        slot = (slot + 1) & mask;
    }
}
// Only ~12 % of the k-mers have a pending bit: a workgroup scans 2048 pend bytes (8 per lane, one 8-byte load),
// gathers the boundary k-mers into an LDS list and resolves them with all lanes busy.
template <int K, bool GR>
__global__ void __launch_bounds__(TB) bl_resolve_kernel(const snk_u128* __restrict__ keys, const uint8_t* __restrict__ pend,
                                                        uint64_t n, const unsigned long long* __restrict__ tab, uint64_t mask,
                                                        uint32_t do_prune, uint8_t* __restrict__ ctx, uint32_t* __restrict__ rq,
                                                        const uint8_t* __restrict__ premote) {
    constexpr int SPAN = 8 * TB;
    __shared__ uint16_t list[SPAN];
    __shared__ uint32_t cnt;
    const uint64_t base = (uint64_t)blockIdx.x * SPAN;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    {
        const uint64_t i8 = base + 8ull * threadIdx.x;
        unsigned long long w = 0;
        if (i8 + 8 <= n) w = *reinterpret_cast<const unsigned long long*>(pend + i8);    // pend is allocated with 16 spare bytes
        else for (uint64_t q = i8; q < n; ++q) w |= (unsigned long long)pend[q] << (8 * (q - i8));
        if (w) {
            uint32_t m = 0;
            for (int q = 0; q < 8; ++q) if ((w >> (8 * q)) & 0xFFull) ++m;
            uint32_t pos = atomicAdd(&cnt, m);
            for (int q = 0; q < 8; ++q) if ((w >> (8 * q)) & 0xFFull) list[pos++] = (uint16_t)(8 * threadIdx.x + q);
        }
    }
    __syncthreads();
    const uint32_t m = cnt;
    for (uint32_t it = threadIdx.x; it < m; it += TB) {
        const uint64_t i = base + list[it];
        const uint32_t pm = pend[i] & (premote ? ~(uint32_t)premote[i] : 0xFFu);
        snk_kmer k = load_key(keys, i);
        uint64_t tag = 0;
        if (GR) { tag = k.lo & 0xFFFFFFFFull; k.lo &= ~0xFFFFFFFFull; }
        uint32_t c = ctx[i];
        const snk_kmer kr = snk_kmer_rc<K>(k);              // rc of a neighbour = one shift of rc(k), see bl_prune_chunk
        for (uint32_t rem = pm; rem; rem &= rem - 1) {
            const uint32_t bit = __ffs(rem) - 1;
            const snk_kmer y = bit < 4 ? snk_kmer_succ<K>(k, bit) : snk_kmer_pred<K>(k, bit - 4);
            const snk_kmer r = bit < 4 ? snk_kmer_pred<K>(kr, 3u - bit) : snk_kmer_succ<K>(kr, 7u - bit);
            const bool rev = snk_kmer_lt(r, y);
            snk_kmer cy = rev ? r : y;
            if (GR) cy.lo |= tag;
            uint32_t h1, h2;
            snk_kmer_hash2(cy, &h1, &h2);
            uint64_t slot = (((uint64_t)h1 << 32) | h2) & mask;
            int64_t j = -1;
            for (;;) {
                const unsigned long long e = tab[slot];
                if (e == 0ull) break;
                if ((uint32_t)(e >> 32) == h1) {
                    const uint64_t idx = (uint32_t)e - 1u;
                    if (snk_kmer_eq(load_key(keys, idx), cy)) { j = (int64_t)idx; break; }
                }
                slot = (slot + 1) & mask;
            }
            // rq[] is not initialised: every pending bit that survives gets its entry here (a side with two surviving
            // bits is a branch and its entry is never read)
            if (j >= 0) rq[2 * i + (bit >> 2)] = ((uint32_t)j << 1) | (rev ? 1u : 0u);
            else if (do_prune) c &= ~(1u << bit);
            else rq[2 * i + (bit >> 2)] = NONE;
        }
        ctx[i] = (uint8_t)c;
    }
}

// ---------------------------------------------------------------------------------------------- L2: fragments
// Paths inside a chunk are ranked by pointer jumping in LDS (<= log2(2n)+1 rounds over the 2n exit states), so every
// node knows its fragment (= smaller terminal state), its position and its orientation, and writes its own base;
// the K-base head k-mers are written by K lanes each.  COUNT pass: fragments per chunk (exact output sizing);
// EMIT pass: the same ranking, then the writes.
// Smooth circles that lie inside one chunk (tiny tandem repeats) are rare: the sizing pass counts only the open paths
// of a chunk (terminal states / 2, no ranking needed) and the circles take their fragment slots from a small pool behind
// the paths' slots.  cur keeps counting past the capacity; the host re-runs the emit pass with a larger pool then.
struct bl_extra_pool {
    unsigned long long* cur;       // [0] fragments, [1] bases
    uint64_t f0, b0;               // first pool fragment / base
    uint64_t fcap, bcap;
};
struct bl_dist_args {          // sharded runs: remote neighbours and the global node numbering
    const uint8_t* premote;
    const uint32_t* rq_idx;
    const uint16_t* rq_meta;
    const unsigned long long* node_off;
    unsigned long long my_node_off;
};
template <int K, int CAP, int T, bool BIG, bool EMIT, bool DIST, bool GR>
__global__ void __launch_bounds__(T) bl_frag_kernel(const uint4* __restrict__ desc, const uint32_t* __restrict__ biglist_in,
                                                    const snk_u128* __restrict__ keys, const uint8_t* __restrict__ ctx,
                                                    const uint8_t* __restrict__ pend, const uint32_t* __restrict__ nbr,
                                                    const uint32_t* __restrict__ rq, bl_dist_args da, uint32_t* __restrict__ nfrag,
                                                    const uint32_t* __restrict__ foff, const uint64_t* __restrict__ boff,
                                                    uint32_t* __restrict__ nk, unsigned long long* __restrict__ hl_self,
                                                    unsigned long long* __restrict__ hl_nb, uint64_t* __restrict__ bstart,
                                                    uint8_t* __restrict__ fbases, uint32_t* __restrict__ fgroup, uint32_t* __restrict__ sfrag, bl_extra_pool xp) {
    constexpr int SPT = (2 * CAP + T - 1) / T;        // states per thread
    // LDS per chunk decides how many chunk waves a CU holds (one wave each): at K=48 without groups only the top 32
    // bits of the low key word are sequence, and the neighbour list (dead once the links are built) shares its memory
    // with the fragment offsets (written after the ranking): 10.5 -> 8.3 KB, 15 -> 18 waves per CU
    typedef typename std::conditional<(K <= 48 && !GR), uint32_t, uint64_t>::type klo_w;
    constexpr int KLS = (K <= 48 && !GR) ? 32 : 0;
    __shared__ uint64_t khi[CAP];
    __shared__ klo_w klo[CAP];
    __shared__ uint16_t nbL[2 * CAP];                  // local neighbour << 1 | rev (chunk-local indices fit 16 bits)
    // ranking record of an exit state: next state | hops << FB | terminal << 2 FB (one LDS word per state)
    typedef typename std::conditional<(CAP <= 256), uint32_t, uint64_t>::type wrec_t;
    constexpr int FB = CAP <= 256 ? 10 : 16;
    constexpr uint32_t FM = (1u << FB) - 1u;           // field mask == "no next state"
    __shared__ uint16_t lnk[2 * CAP];
    __shared__ wrec_t wr[2 * CAP];
    // offset of a fragment's bases inside the chunk's output.  A chunk of n k-mers in F fragments has n + (K-1) F <= CAP K
    // bases: 16 bits do (per terminal state, in the dead neighbour list) unless CAP K > 65535 -- the 1280-node variant at
    // K=60, where 1216 one-k-mer fragments are 72960 bases: there the terminal state keeps the fragment's index and the
    // offsets are 32-bit words per fragment
    constexpr bool WIDE = (uint32_t)CAP * (uint32_t)K > 65535u;
    uint16_t* foffL = nbL;
    __shared__ uint32_t frel[WIDE ? CAP : 1];
    __shared__ uint16_t hnode[CAP];                    // heads: node << 1 | rc
    __shared__ uint8_t ctxL[CAP], pendL[CAP], palL[CAP];
    __shared__ uint32_t fcnt, bcnt, changed;
    const int tid = threadIdx.x;
    const uint32_t c = BIG ? biglist_in[blockIdx.x] : blockIdx.x;
    const chunk_t ch = chunk_get(desc, c);
    if (ch.n == 0) return;
    if (!BIG && ch.n > (uint32_t)CAP) return;
    const uint32_t n = ch.n;
    const uint32_t c_foff = EMIT ? foff[c] : 0u;       // issued now, needed after the ranking
    const uint64_t c_boff = EMIT ? boff[c] : 0ull;
    if (tid == 0) { fcnt = 0; bcnt = 0; changed = 0; }
    {   // every load of the chunk is issued before the first LDS store (a chunk wave starts with 5 loads per node)
        constexpr int NPT = (CAP + T - 1) / T;
        snk_kmer kq[NPT];
        uint32_t cq[NPT], pq[NPT];
        uint32_t nq[NPT];
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const uint32_t i = tid + q * T;
            if (i < n) {
                const uint64_t gi = ch.base + i;
                if (EMIT) kq[q] = load_key(keys, gi);          // (the sizing pass decides links from contexts, pending bits and the neighbour word alone)
                cq[q] = ctx[gi];
                pq[q] = pend[gi];
                nq[q] = nbr[gi];
            }
        }
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const uint32_t i = tid + q * T;
            if (i < n) {
                if (EMIT) {
                    const snk_kmer k = kq[q];
                    khi[i] = k.hi;
                    klo[i] = (klo_w)(k.lo >> KLS);
                }
                palL[i] = (uint8_t)((nq[q] >> 15) & 1u);          // (noted by the prune, which has the reverse complement at hand)
                ctxL[i] = (uint8_t)cq[q];
                pendL[i] = (uint8_t)pq[q];
                const uint32_t a0 = nq[q] & 0x0FFFu, a1 = (nq[q] >> 16) & 0x0FFFu;
                nbL[2 * i] = a0 == 0x0FFFu ? NONE16 : (uint16_t)a0;
                nbL[2 * i + 1] = a1 == 0x0FFFu ? NONE16 : (uint16_t)a1;
            }
        }
    }
    __syncthreads();
    // reciprocal-unique links inside the chunk (BuildReadQGraph48.cc:408-428): state = node << 1 | exit side
    uint32_t myterm = 0;
    for (uint32_t s = tid; s < 2 * n; s += T) {
        const uint32_t i = s >> 1, side = s & 1u;
        const uint32_t cc = ctxL[i];
        const uint32_t bits = side ? (cc >> 4) : (cc & 15u);
        uint16_t out = NONE16;
        if (__popc(bits) == 1) {
            const uint32_t b = __ffs(bits) - 1;
            if (!((pendL[i] >> (4 * side + b)) & 1u)) {
                const uint32_t nb = nbL[s];
                if (nb != NONE16) {
                    const uint32_t j = nb >> 1, rev = nb & 1u;
                    const uint32_t fs = side ^ 1u ^ rev;
                    const uint32_t cj = ctxL[j];
                    const uint32_t deg = fs ? __popc(cj & 0xF0u) : __popc(cj & 0x0Fu);
                    if (deg == 1 && !palL[i] && !palL[j]) out = (uint16_t)((j << 1) | fs);
                }
            }
        }
        lnk[s] = out;
        if (!EMIT) { if (out == NONE16) ++myterm; }
        else if (out == NONE16) wr[s] = (wrec_t)FM | ((wrec_t)s << (2 * FB));
        else wr[s] = (wrec_t)(out ^ 1u) | ((wrec_t)1 << FB) | ((wrec_t)(out ^ 1u) << (2 * FB));
    }
    if (!EMIT) {      // sizing pass: a chunk has (terminal states / 2) open paths
        snk_wave_add(&fcnt, myterm);
        __syncthreads();
        if (tid == 0) nfrag[c] = fcnt >> 1;
        return;
    }
    __syncthreads();
    // pointer jumping: nxt/dst/tl[s] = state reached / hops / terminal when leaving through exit state s
    int max_rounds = 2;
    while ((1u << (max_rounds - 1)) < 2 * n) ++max_rounds;
    for (int round = 0; round < max_rounds; ++round) {
        wrec_t rr[SPT];
        bool any = false;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            if ((uint32_t)(q * T) >= 2 * n) break;       // uniform: a typical chunk fills 4 of the 8 slots
            const uint32_t s = tid + q * T;
            if (s < 2 * n) {
                const wrec_t a0 = wr[s];
                const uint32_t n1 = (uint32_t)a0 & FM;
                rr[q] = a0;
                if (n1 != FM) {
                    const wrec_t a1 = wr[n1];
                    const uint32_t d = (((uint32_t)(a0 >> FB) & FM) + ((uint32_t)(a1 >> FB) & FM)) & FM;   // hops (garbage on circles)
                    rr[q] = (a1 & (wrec_t)FM) | ((wrec_t)d << FB) | (a1 & ((wrec_t)FM << (2 * FB)));
                    any = true;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            if ((uint32_t)(q * T) >= 2 * n) break;
            const uint32_t s = tid + q * T;
            if (s < 2 * n) wr[s] = rr[q];
        }
        if (any) changed = 1;
        __syncthreads();
        const bool go = changed != 0;
        __syncthreads();
        if (tid == 0) changed = 0;
        if (!go) break;
    }
    __syncthreads();
    auto w_next = [&](uint32_t st) -> uint32_t { return (uint32_t)wr[st] & FM; };
    auto w_dist = [&](uint32_t st) -> uint32_t { return (uint32_t)(wr[st] >> FB) & FM; };
    auto w_tail = [&](uint32_t st) -> uint32_t { return (uint32_t)(wr[st] >> (2 * FB)) & FM; };

    // half link of a fragment end: the single remote neighbour of that side (decided half on each owner)
    auto half_link = [&](uint32_t s) -> unsigned long long {
        const uint32_t i = s >> 1, side = s & 1u;
        const uint32_t cc = ctxL[i];
        const uint32_t bits = side ? (cc >> 4) : (cc & 15u);
        if (__popc(bits) != 1 || palL[i]) return NONE64;
        const uint32_t b = __ffs(bits) - 1;
        if (!((pendL[i] >> (4 * side + b)) & 1u)) return NONE64;
        snk_kmer k;
        k.hi = khi[i];
        k.lo = GR ? ((uint64_t)klo[i] & ~0xFFFFFFFFull) : ((uint64_t)klo[i] << KLS);
        const snk_kmer y = side ? snk_kmer_pred<K>(k, b) : snk_kmer_succ<K>(k, b);
        if (snk_kmer_eq(y, snk_kmer_rc<K>(y))) return NONE64;
        const uint64_t gi = ch.base + i;
        if (DIST && ((da.premote[gi] >> (4 * side + b)) & 1u)) {         // the neighbour lives on another rank
            const uint32_t a = da.rq_idx[2 * gi + side];
            if (a == NONE) return NONE64;
            const uint32_t meta = da.rq_meta[2 * gi + side];
            const uint32_t fs = side ^ 1u ^ (meta >> 15);
            return 2ull * (da.node_off[meta & 0x7FFFu] + a) + fs;
        }
        const uint32_t a = rq[2 * gi + side];
        if (a == NONE) return NONE64;
        const uint32_t rev = a & 1u, fs = side ^ 1u ^ rev;
        return 2ull * ((DIST ? da.my_node_off : 0ull) + (a >> 1)) + fs;
    };
    // fragment descriptor; returns the offset of its bases
    auto describe = [&](uint32_t pid, uint32_t other, uint32_t cnt, bool with_half, uint32_t head_node, bool head_rc) {
        const uint32_t lf = atomicAdd(&fcnt, 1u);
        const uint32_t rel = atomicAdd(&bcnt, cnt + (uint32_t)K - 1u);
        const uint64_t f = (uint64_t)c_foff + lf;
        nk[f] = cnt;
        hl_self[2 * f] = 2ull * ((DIST ? da.my_node_off : 0ull) + ch.base) + pid;
        hl_self[2 * f + 1] = 2ull * ((DIST ? da.my_node_off : 0ull) + ch.base) + other;
        hl_nb[2 * f] = with_half ? half_link(pid) : NONE64;
        hl_nb[2 * f + 1] = with_half ? half_link(other) : NONE64;
        bstart[f] = c_boff + rel;
        if (sfrag) { sfrag[2 * ch.base + pid] = (uint32_t)(2 * f); sfrag[2 * ch.base + other] = (uint32_t)(2 * f + 1); }
        if (GR) fgroup[f] = (uint32_t)klo[head_node];
        if (WIDE) { foffL[pid] = (uint16_t)lf; frel[lf] = rel; }
        else foffL[pid] = (uint16_t)rel;
        hnode[lf] = (uint16_t)((head_node << 1) | (head_rc ? 1u : 0u));
        return rel;
    };

    // nodes on open paths: position and orientation from the two ranks (walking from the smaller terminal)
    for (uint32_t i = tid; i < n; i += T) {
        if (w_next(2 * i) != FM) continue;                // on a circle
        const uint32_t tR = w_tail(2 * i), tL = w_tail(2 * i + 1), dR = w_dist(2 * i), dL = w_dist(2 * i + 1);
        const bool fwd = tL < tR;
        const uint32_t pos = fwd ? dL : dR;
        if (pos == 0) describe(fwd ? tL : tR, fwd ? tR : tL, dR + dL + 1u, true, i, !fwd);
    }
    // nodes no terminal is reachable from lie on smooth circles inside the chunk: cut at the left side of the minimum
    // k-mer (canonicalizeCircle, BuildReadQGraph48.cc:375-397); the circle's minimum node describes and writes it
    for (uint32_t i = tid; i < n; i += T) {
        if (w_next(2 * i) == FM) continue;
        uint32_t cur = i << 1, mn = i, cnt = 1;
        bool closed = false;
        while (cnt <= n) {
            const uint32_t l = lnk[cur];
            if (l == NONE16) break;
            const uint32_t j = l >> 1;
            if (j == i) { closed = true; break; }
            if (khi[j] < khi[mn] || (khi[j] == khi[mn] && klo[j] < klo[mn])) mn = j;
            cur = l ^ 1u;
            ++cnt;
        }
        if (closed && mn == i) {
            // a circle inside the chunk: its slot comes from the pool behind the paths' slots
            const unsigned long long xf = atomicAdd(&xp.cur[0], 1ull);
            const unsigned long long xb = atomicAdd(&xp.cur[1], (unsigned long long)(cnt + K - 1));
            if (xf < xp.fcap && xb + cnt + K - 1 <= xp.bcap) {
                uint32_t e = i << 1;                       // exit state of the last node: cnt-1 links from (i, side 0)
                for (uint32_t q = 1; q < cnt; ++q) e = lnk[e] ^ 1u;
                const uint64_t f = xp.f0 + xf, bo = xp.b0 + xb;
                const uint32_t pid = (i << 1) | 1u;
                nk[f] = cnt;
                hl_self[2 * f] = 2ull * ((DIST ? da.my_node_off : 0ull) + ch.base) + pid;
                hl_self[2 * f + 1] = 2ull * ((DIST ? da.my_node_off : 0ull) + ch.base) + e;
                hl_nb[2 * f] = NONE64;
                hl_nb[2 * f + 1] = NONE64;
                bstart[f] = bo;
                if (sfrag) { sfrag[2 * ch.base + pid] = (uint32_t)(2 * f); sfrag[2 * ch.base + e] = (uint32_t)(2 * f + 1); }
                if (GR) fgroup[f] = (uint32_t)klo[i];
                snk_kmer k;
                k.hi = khi[i];
                k.lo = (uint64_t)klo[i] << KLS;
                for (int b = 0; b < K; ++b) fbases[bo + b] = (uint8_t)oriented_base<K>(k, false, b);
                uint32_t curs = i << 1;
                for (uint32_t pos = 1; pos < cnt; ++pos) {
                    const uint32_t l = lnk[curs];
                    k.hi = khi[l >> 1];
                    k.lo = (uint64_t)klo[l >> 1] << KLS;
                    fbases[bo + (K - 1) + pos] = (uint8_t)oriented_base<K>(k, (l & 1u) == 0, K - 1);
                    curs = l ^ 1u;
                }
            }
        }
    }
    __syncthreads();
    // every path node writes its last base at its position
    for (uint32_t i = tid; i < n; i += T) {
        if (w_next(2 * i) != FM) continue;
        const uint32_t tR = w_tail(2 * i), tL = w_tail(2 * i + 1), dR = w_dist(2 * i), dL = w_dist(2 * i + 1);
        const bool fwd = tL < tR;
        const uint32_t pos = fwd ? dL : dR;
        if (pos == 0) continue;
        snk_kmer k;
        k.hi = khi[i];
        k.lo = (uint64_t)klo[i] << KLS;
        const uint32_t fo = foffL[fwd ? tL : tR];
        // (round 4, measured with these stores compiled out: 8.55 -> 8.23 ms -- the one-byte-per-node stores are NOT what the kernel
        // spends its time on, so packing the fragments' bases to 2 bits, DESIGN 8 item 6 of round 3, would not buy the 2 ms hoped for)
        fbases[c_boff + (WIDE ? frel[fo] : fo) + (K - 1) + pos] = (uint8_t)oriented_base<K>(k, !fwd, K - 1);
    }
    // head k-mers: K / 4 lanes per head, four bases (one dword store at byte alignment) per lane, four heads per wave
    // instruction (K lanes writing a byte each made this the kernel's largest block of memory instructions: one per head)
    struct __attribute__((packed)) u32_any { uint32_t v; };
    const uint32_t nheads = fcnt;
    constexpr int HPI = T / 16;                            // heads per iteration: sixteen lanes each
    static_assert(K % 4 == 0 && K / 4 <= 16, "a head is K / 4 dwords, one per lane of its group");
    for (uint32_t h0 = 0; h0 < nheads; h0 += HPI) {
        const uint32_t h = h0 + (tid >> 4);
        const int q = tid & 15;
        if (h < nheads && q < K / 4) {
            const uint32_t hn = hnode[h];
            const uint32_t i = hn >> 1;
            const bool rc = hn & 1u;
            uint32_t fo;
            if (WIDE) fo = frel[h];                               // head h belongs to fragment h of the chunk
            else { const uint32_t tR = w_tail(2 * i), tL = w_tail(2 * i + 1); fo = foffL[tL < tR ? tL : tR]; }
            snk_kmer k;
            k.hi = khi[i];
            k.lo = (uint64_t)klo[i] << KLS;
            const uint32_t v = oriented_base<K>(k, rc, 4 * q) | (oriented_base<K>(k, rc, 4 * q + 1) << 8) | (oriented_base<K>(k, rc, 4 * q + 2) << 16) |
                               (oriented_base<K>(k, rc, 4 * q + 3) << 24);
            reinterpret_cast<u32_any*>(fbases + c_boff + fo + 4 * q)->v = v;
        }
    }
}

__global__ void __launch_bounds__(TB) bl_chunk_bases_kernel(const uint4* __restrict__ desc, const uint32_t* __restrict__ nfrag, uint32_t nchunks,
                                                            uint32_t K, uint64_t* __restrict__ nbases) {
    const uint32_t c = blockIdx.x * TB + threadIdx.x;
    if (c > nchunks) return;
    if (c == nchunks) { nbases[c] = 0; return; }
    const uint32_t n = desc[c].y;
    nbases[c] = nfrag[c] ? (uint64_t)n + (uint64_t)(K - 1) * nfrag[c] : 0ull;
}

__global__ void __launch_bounds__(TB) bl_pack_vals_kernel(const uint32_t* __restrict__ counts, const uint8_t* __restrict__ ctx,
                                                          uint64_t n, uint64_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) vals[i] = ((uint64_t)counts[i] << 8) | ctx[i];
}
__global__ void __launch_bounds__(TB) bl_unpack_vals_kernel(const uint64_t* __restrict__ vals, uint64_t n,
                                                            uint32_t* __restrict__ counts, uint8_t* __restrict__ ctx) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) { counts[i] = (uint32_t)(vals[i] >> 8); ctx[i] = (uint8_t)(vals[i] & 0xFFu); }
}

}  // namespace

#define G_ALLOC(ptr, type, count)                                                       \
    do {                                                                                \
        void* _p = nullptr;                                                             \
        int _rc = snk_ctx_alloc(ctx, sizeof(type) * (size_t)(count), &_p, err, errcap); \
        if (_rc) return _rc;                                                            \
        ptr = (type*)_p;                                                                \
    } while (0)

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

constexpr int SCAP = 256, ST = 64;      // small chunks: one wave per chunk
constexpr int BCAP = SNK_GRAPH_CHUNK_MAX, BT = 256;    // big chunks (a count sub-pass that retains more is counted again in halves: snk_count.hip)

namespace {
// ---- sharded runs: membership queries for pending neighbours that belong to another rank
// query record: 3 x u64 = key lo, key hi, (node | bit << 32 | rev << 40)   (same as snk_graph.hip's sharded stage)
constexpr int QWORDS = 4;               // 8-byte words of pending flags per thread
constexpr int QSPAN = 8 * QWORDS * TB;      // k-mers per workgroup: ONE reservation per destination rank for all of them (the per-rank cursors are a few words on
                                            // one 64-byte line: atomics on a line are served one at a time; 131 k workgroups x 8 ranks were ~10 ms per pass)
template <int K, bool FILL>
__global__ void __launch_bounds__(TB) bl_query_kernel(const snk_u128* __restrict__ keys, const uint8_t* __restrict__ premote, uint64_t n,
                                                      uint32_t NB_total, uint32_t NBl, uint32_t world, uint32_t mlen,
                                                      unsigned long long* __restrict__ qcount_or_cursor,
                                                      unsigned long long* __restrict__ qbuf) {
    extern __shared__ unsigned long long dynq[];          // [world] counts, then reserved bases
    __shared__ uint16_t list[QSPAN];
    __shared__ uint32_t cnt;
    uint32_t* lcur = reinterpret_cast<uint32_t*>(dynq + world);   // [world] local cursors
    const uint64_t base = (uint64_t)blockIdx.x * QSPAN;
    if (threadIdx.x == 0) cnt = 0;
    for (uint32_t r = threadIdx.x; r < world; r += TB) { dynq[r] = 0; lcur[r] = 0; }
    __syncthreads();
    for (int t = 0; t < QWORDS; ++t) {
        const uint32_t w0 = 8u * ((uint32_t)t * TB + threadIdx.x);        // first item of this thread's word inside the span
        const uint64_t i8 = base + w0;
        unsigned long long w = 0;
        if (i8 + 8 <= n) w = *reinterpret_cast<const unsigned long long*>(premote + i8);
        else for (uint64_t q = i8; q < n; ++q) w |= (unsigned long long)premote[q] << (8 * (q - i8));
        if (w) {
            uint32_t m = 0;
            for (int q = 0; q < 8; ++q) if ((w >> (8 * q)) & 0xFFull) ++m;
            uint32_t pos = atomicAdd(&cnt, m);
            for (int q = 0; q < 8; ++q) if ((w >> (8 * q)) & 0xFFull) list[pos++] = (uint16_t)(w0 + q);
        }
    }
    __syncthreads();
    const uint32_t m = cnt;
    // pass 1: queries per destination rank
    for (uint32_t it = threadIdx.x; it < m; it += TB) {
        const uint64_t i = base + list[it];
        const snk_kmer k = load_key(keys, i);
        for (uint32_t rem = premote[i]; rem; rem &= rem - 1) {
            const uint32_t bit = __ffs(rem) - 1;
            const snk_kmer y = bit < 4 ? snk_kmer_succ<K>(k, bit) : snk_kmer_pred<K>(k, bit - 4);
            const uint32_t owner = (mlen == (uint32_t)SNK_M_LONG ? snk_bucket_of_kmer<K, SNK_M_LONG>(y, NB_total) : snk_bucket_of_kmer<K, SNK_M_OF(K)>(y, NB_total)) / NBl;
            atomicAdd(&dynq[owner], 1ull);
        }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < world; r += TB) {
        const unsigned long long c = dynq[r];
        if (c) dynq[r] = FILL ? atomicAdd(&qcount_or_cursor[r], c) : (atomicAdd(&qcount_or_cursor[r], c), 0ull);
    }
    if (!FILL) return;
    __syncthreads();
    // pass 2: every query takes a slot of its destination's segment
    for (uint32_t it = threadIdx.x; it < m; it += TB) {
        const uint64_t i = base + list[it];
        const snk_kmer k = load_key(keys, i);
        for (uint32_t rem = premote[i]; rem; rem &= rem - 1) {
            const uint32_t bit = __ffs(rem) - 1;
            const snk_kmer y = bit < 4 ? snk_kmer_succ<K>(k, bit) : snk_kmer_pred<K>(k, bit - 4);
            const uint32_t owner = (mlen == (uint32_t)SNK_M_LONG ? snk_bucket_of_kmer<K, SNK_M_LONG>(y, NB_total) : snk_bucket_of_kmer<K, SNK_M_OF(K)>(y, NB_total)) / NBl;
            const snk_kmer r = snk_kmer_rc<K>(y);
            const bool rev = snk_kmer_lt(r, y);
            const snk_kmer c = rev ? r : y;
            const unsigned long long slot = dynq[owner] + atomicAdd(&lcur[owner], 1u);
            qbuf[3 * slot + 0] = c.lo;
            qbuf[3 * slot + 1] = c.hi;
            qbuf[3 * slot + 2] = (unsigned long long)i | ((unsigned long long)bit << 32) | ((unsigned long long)(rev ? 1 : 0) << 40);
        }
    }
}
}  // namespace

// ---- stage 1: chunk records, local prune, boundary index, pending bits of this rank resolved
template <int K, bool GR>
static int bl_prune_impl(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, char* err, size_t errcap) {
    const snk_table* tab = B->tab;
    const uint64_t n = tab->n;
    chunk_src cs;
    cs.chunk_n = tab->chunk_n;
    cs.chunk_base = tab->chunk_base;
    cs.extra = tab->extra;
    cs.region_off = tab->region_off;
    cs.NB = tab->NB;
    cs.n_extra = tab->n_extra;
    cs.n_regions = tab->n_regions;
    cs.span = nullptr;
    // (SNK_CHUNK_MERGE = k-mers a merged chunk may hold, 0 = off: 1.5 % errors graph 43.0 -> 38.9 ms, 0.6 % 33.9 -> 32.9, groups 15.1 -> 13.6,
    // K=60 29.5 -> 28.2, the bench's reads 29.65 -> 29.4)
    const uint32_t merge_cap = std::min<uint32_t>(snk_opt_u32("chunk_merge", (uint32_t)SCAP), (uint32_t)SCAP);
    if (merge_cap && tab->chunk_n && tab->NB > tab->n_regions) {
        uint8_t* span;
        uint32_t* merged;
        G_ALLOC(span, uint8_t, (uint64_t)tab->NB + 1);
        G_ALLOC(merged, uint32_t, (uint64_t)tab->NB + 1);
        hipLaunchKernelGGL(bl_chunk_merge_kernel, dim3((tab->n_regions + 255) / 256), dim3(256), 0, st, tab->chunk_n, tab->chunk_base, tab->NB, tab->n_regions, merge_cap, span, merged);
        cs.span = span;
        cs.chunk_n = merged;
    }
    const uint32_t nchunks = tab->NB + tab->n_extra;
    B->nchunks = nchunks;
    G_ALLOC(B->desc, uint4, (uint64_t)nchunks + 1);
    uint64_t* desc_src = nullptr;
    if (tab->keys_r) G_ALLOC(desc_src, uint64_t, (uint64_t)nchunks + 1);
    hipLaunchKernelGGL(bl_chunk_desc_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, st, cs, nchunks, B->desc, tab->region_cap, desc_src);
    uint32_t *nbnd, *ctr;
    G_ALLOC(B->ctx, uint8_t, n + 16);
    G_ALLOC(B->pend, uint8_t, n + 16);
    G_ALLOC(B->counts, uint32_t, n + 4);
    G_ALLOC(B->nbr, uint32_t, n + 2);
    G_ALLOC(B->rq, uint32_t, 2 * n + 2);
    G_ALLOC(nbnd, uint32_t, (uint64_t)nchunks + 1);
    G_ALLOC(B->biglist, uint32_t, n / SCAP + 2);
    G_ALLOC(ctr, uint32_t, 16);
    B->premote = nullptr;
    if (B->world > 1 || B->force_dist) {
        G_ALLOC(B->premote, uint8_t, n + 16);
        SNK_HIP_TRY(hipMemsetAsync(B->premote, 0, n + 16, st));
    }
    SNK_HIP_TRY(hipMemsetAsync(nbnd, 0, ((uint64_t)nchunks + 1) * 4, st));
    SNK_HIP_TRY(hipMemsetAsync(ctr, 0, 64, st));
    bl_shard sh;
    sh.bucket_base = B->premote ? B->rank * B->NBl : 0u;
    sh.NBl = B->premote ? B->NBl : 0u;
    sh.me = B->rank;
    sh.G = tab->n_regions ? tab->n_regions : 1u;
    sh.premote = B->premote;
    const uint32_t NBh = B->premote ? B->NB_total : tab->NB;      // bucket count of the minimiser hash
    const uint32_t cpw = std::max(1u, snk_opt_u32("bl_cpw", 4));      // (chunks per workgroup: merged-away chunks are empty; 1 -> 4: graph 29.2 -> 28.8 ms, 1.5 % errors 38.7 -> 37.6)
    // boundary index, filled by the prune kernel itself: sized from what the previous call of this context found for a table of
    // this size (first call: n / 5); too small = too full -> the separate build pass below redoes it with the exact size
    uint64_t tg0 = 0;
    unsigned long long* index0 = nullptr;
    if (snk_opt_u32("bl_index_fused", 1)) {
        const uint64_t guess = (ctx->last_bnd_n == n && ctx->last_bnd) ? ctx->last_bnd + ctx->last_bnd / 8 : n / 5;
        tg0 = 1024;
        while (tg0 < 2 * guess) tg0 <<= 1;
        G_ALLOC(index0, unsigned long long, tg0 + 1);
        SNK_HIP_TRY(hipMemsetAsync(index0, 0, (tg0 + 1) * 8, st));
    }
    bl_regions rg;
    rg.keys_r = tab->keys_r; rg.vals_r = tab->vals_r; rg.desc_src = desc_src;
    rg.keys_dense = const_cast<snk_u128*>(tab->keys);
    const bool long_m = ctx->mlen == (uint32_t)SNK_M_LONG;
    hipLaunchKernelGGL((long_m ? bl_prune_kernel<K, SCAP, ST, false, GR, SNK_M_LONG> : bl_prune_kernel<K, SCAP, ST, false, GR, SNK_M_OF(K)>), dim3((nchunks + cpw - 1) / cpw), dim3(ST), 0, st, (const uint4*)B->desc, NBh, sh, rg,
                       (const uint32_t*)nullptr, nchunks, cpw, tab->keys, tab->vals, B->do_prune | (snk_opt_u32("bl_noclassify", 0) ? 2u : 0u), B->ctx, B->counts, B->pend, B->nbr, nbnd,
                       B->biglist, ctr, (const uint32_t*)nullptr, index0, tg0 - 1);
    SNK_HIP_TRY(hipGetLastError());
    // the chunks that did not fit the one-wave variant (split sub-passes near the table's capacity): their number stays on the
    // device until the read-back below -- the big variant runs a fixed grid that strides over the list
    uint32_t h_nbig = 0;
    SNK_HIP_TRY(hipMemcpyAsync(ctr + 1, ctr, 4, hipMemcpyDeviceToDevice, st));       // (ctr[0] is the small kernel's list cursor)
    hipLaunchKernelGGL((long_m ? bl_prune_kernel<K, BCAP, BT, true, GR, SNK_M_LONG> : bl_prune_kernel<K, BCAP, BT, true, GR, SNK_M_OF(K)>), dim3(2048), dim3(BT), 0, st, (const uint4*)B->desc, NBh, sh, rg,
                       (const uint32_t*)B->biglist, 0u, 1u, tab->keys, tab->vals, B->do_prune | (snk_opt_u32("bl_noclassify", 0) ? 2u : 0u), B->ctx, B->counts, B->pend, B->nbr,
                       nbnd, B->biglist, ctr + 2, (const uint32_t*)(ctr + 1), index0, tg0 - 1);
    SNK_HIP_TRY(hipGetLastError());
    if (tab->keys_r) {       // the region-partitioned copy is dead (stream order): later stages may reuse it
        snk_ctx_release_block(ctx, tab->keys_r);
        snk_ctx_release_block(ctx, tab->vals_r);
    }
    unsigned long long* d_sum;
    G_ALLOC(d_sum, unsigned long long, 2);
    {
        size_t tb = 0;
        auto in = rocprim::make_transform_iterator(nbnd, [] __device__(uint32_t v) { return (unsigned long long)v; });
        SNK_HIP_TRY(rocprim::reduce((void*)nullptr, tb, in, d_sum, 0ull, (size_t)nchunks, rocprim::plus<unsigned long long>(), st));
        void* tmp;
        int rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap);
        if (rc) return rc;
        SNK_HIP_TRY(rocprim::reduce(tmp, tb, in, d_sum, 0ull, (size_t)nchunks, rocprim::plus<unsigned long long>(), st));
    }
    unsigned long long h_bnd = 0, h_gaveup = 0;
    if (index0) SNK_HIP_TRY(hipMemcpyAsync(&h_gaveup, index0 + tg0, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_bnd, d_sum, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_nbig, ctr + 1, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    B->nbig = h_nbig;
    B->n_boundary = h_bnd;
    ctx->last_bnd = h_bnd; ctx->last_bnd_n = n;
    uint64_t tg = 1024;
    while (tg < 2 * h_bnd) tg <<= 1;
    const bool fused_ok = index0 && !h_gaveup && h_bnd * 4 <= tg0 * 3;          // load <= 0.75: every insert found a slot long before its probe limit
    if (fused_ok) { B->index = index0; tg = tg0; }
    else {
        if (index0) snk_ctx_release_block(ctx, index0);
        G_ALLOC(B->index, unsigned long long, tg);
        SNK_HIP_TRY(hipMemsetAsync(B->index, 0, tg * 8, st));
    }
    B->index_mask = tg - 1;
    if (h_bnd) {
        if (!fused_ok) hipLaunchKernelGGL(bl_index_build_kernel, dim3(nblk(n)), dim3(TB), 0, st, tab->keys, B->pend, n, B->index, tg - 1);
        hipLaunchKernelGGL((bl_resolve_kernel<K, GR>), dim3((unsigned)((n + 8 * TB - 1) / (8 * TB))), dim3(TB), 0, st, tab->keys, B->pend, n,
                           B->index, tg - 1, B->do_prune, B->ctx, B->rq, (const uint8_t*)B->premote);
    }
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// ---- stage 2: fragments of every chunk (exact sizing pass, then the writes)
template <int K, bool DIST, bool GR>
static int bl_fragments_impl(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, const bl_dist_args& da, snk_frag_out* out, char* err,
                             size_t errcap) {
    const snk_table* tab = B->tab;
    const uint32_t nchunks = B->nchunks, h_nbig = B->nbig;
    uint32_t *nfrag, *foff;
    uint64_t *nbases, *boff;
    G_ALLOC(nfrag, uint32_t, (uint64_t)nchunks + 1);
    G_ALLOC(foff, uint32_t, (uint64_t)nchunks + 1);
    G_ALLOC(nbases, uint64_t, (uint64_t)nchunks + 1);
    G_ALLOC(boff, uint64_t, (uint64_t)nchunks + 1);
    SNK_HIP_TRY(hipMemsetAsync(nfrag, 0, ((uint64_t)nchunks + 1) * 4, st));
    bl_extra_pool xp0;
    memset(&xp0, 0, sizeof xp0);
    // sizing pass: open paths per chunk (links only, no ranking)
    hipLaunchKernelGGL((bl_frag_kernel<K, SCAP, ST, false, false, DIST, GR>), dim3(nchunks), dim3(ST), 0, st, (const uint4*)B->desc,
                       (const uint32_t*)nullptr, tab->keys, B->ctx, B->pend, B->nbr, B->rq, da, nfrag, (const uint32_t*)nullptr,
                       (const uint64_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                       (uint64_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, xp0);
    if (h_nbig)
        hipLaunchKernelGGL((bl_frag_kernel<K, BCAP, BT, true, false, DIST, GR>), dim3(h_nbig), dim3(BT), 0, st, (const uint4*)B->desc,
                           (const uint32_t*)B->biglist, tab->keys, B->ctx, B->pend, B->nbr, B->rq, da, nfrag, (const uint32_t*)nullptr,
                           (const uint64_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                           (uint64_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, xp0);
    hipLaunchKernelGGL(bl_chunk_bases_kernel, dim3(nblk((uint64_t)nchunks + 1)), dim3(TB), 0, st, (const uint4*)B->desc, nfrag, nchunks,
                       (uint32_t)K, nbases);
    SNK_HIP_TRY(hipGetLastError());
    int rc;
    if ((rc = excl_scan<uint32_t>(ctx, st, nfrag, foff, (size_t)nchunks + 1, err, errcap))) return rc;
    if ((rc = excl_scan<uint64_t>(ctx, st, nbases, boff, (size_t)nchunks + 1, err, errcap))) return rc;
    uint32_t h_F = 0;
    uint64_t h_B = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_F, foff + nchunks, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_B, boff + nchunks, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    // the open paths' bases: every node once + K-1 per path; circle nodes are not in any path, their bases come from the pool
    unsigned long long* xcur;
    G_ALLOC(xcur, unsigned long long, 2);
    const uint32_t pool0 = snk_opt_u32("bl_pool", 4096);                 // SNK_BL_POOL=0: tests force the exact re-run
    uint64_t xf_cap = pool0 ? pool0 + h_F / 256 : 0, xb_cap = xf_cap * (K + 63);
    unsigned long long h_x[2] = {0, 0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        G_ALLOC(out->nk, uint32_t, (uint64_t)h_F + xf_cap + 1);
        G_ALLOC(out->hl_self, unsigned long long, 2ull * (h_F + xf_cap) + 2);
        G_ALLOC(out->hl_nb, unsigned long long, 2ull * (h_F + xf_cap) + 2);
        G_ALLOC(out->boff, uint64_t, (uint64_t)h_F + xf_cap + 2);
        G_ALLOC(out->bases, uint8_t, h_B + xb_cap + 16);
        out->fgroup = nullptr;
        if (GR) G_ALLOC(out->fgroup, uint32_t, (uint64_t)h_F + xf_cap + 1);
        G_ALLOC(out->sfrag, uint32_t, 2 * tab->n + 2);      // terminal state (local numbering) -> 2 * local fragment + end
        SNK_HIP_TRY(hipMemsetAsync(xcur, 0, 16, st));
        bl_extra_pool xp;
        xp.cur = xcur;
        xp.f0 = h_F;
        xp.b0 = h_B;
        xp.fcap = xf_cap;
        xp.bcap = xb_cap;
        hipLaunchKernelGGL((bl_frag_kernel<K, SCAP, ST, false, true, DIST, GR>), dim3(nchunks), dim3(ST), 0, st, (const uint4*)B->desc,
                           (const uint32_t*)nullptr, tab->keys, B->ctx, B->pend, B->nbr, B->rq, da, nfrag, foff, boff, out->nk, out->hl_self,
                           out->hl_nb, out->boff, out->bases, out->fgroup, out->sfrag, xp);
        if (h_nbig)
            hipLaunchKernelGGL((bl_frag_kernel<K, BCAP, BT, true, true, DIST, GR>), dim3(h_nbig), dim3(BT), 0, st, (const uint4*)B->desc,
                               (const uint32_t*)B->biglist, tab->keys, B->ctx, B->pend, B->nbr, B->rq, da, nfrag, foff, boff, out->nk,
                               out->hl_self, out->hl_nb, out->boff, out->bases, out->fgroup, out->sfrag, xp);
        SNK_HIP_TRY(hipGetLastError());
        SNK_HIP_TRY(hipMemcpyAsync(h_x, xcur, 16, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        if (h_x[0] <= xf_cap && h_x[1] <= xb_cap) break;
        if (attempt == 1) return snk_fail(SNK_E_INTERNAL, err, errcap, "bucket-local graph: circle pool overflow");
        xf_cap = h_x[0] + 64;          // exact requirement (the counters keep counting past the capacity)
        xb_cap = h_x[1] + 64;
    }
    out->n_frags = h_F + h_x[0];
    out->total_bases = h_B + h_x[1];
    out->n_local_circles = (uint32_t)h_x[0];
    return SNK_OK;
}

template <int K, bool GR>
static int local_graph_impl(snk_ctx* ctx, hipStream_t st, const snk_table* tab, uint32_t do_prune, bool want_unitigs,
                            bool sort_table, snk_graph_out* out, snk_u128** keys_final, float* ms, char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    const uint64_t n = tab->n;
    *keys_final = tab->keys;
    if (n == 0) {
        int rc0 = snk_spectrum(ctx, st, nullptr, 0, &out->spectrum, &out->spectrum_bins, err, errcap);
        if (rc0) return rc0;
        G_ALLOC(out->unitig_off, uint64_t, 2);
        SNK_HIP_TRY(hipMemsetAsync(out->unitig_off, 0, 16, st));
        G_ALLOC(out->unitig_bases, uint8_t, 16);
        G_ALLOC(out->ctx, uint8_t, 16);
        G_ALLOC(out->counts, uint32_t, 4);
        return SNK_OK;
    }
    if (n >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 retained k-mers on one GPU (%llu)", (unsigned long long)n);
#ifndef SNK_DBG_NO_CHUNK_CHECK      // (tuning builds that only time the count kernel with a larger table: tools/build_variant.sh)
    if (snk_count_slots(K) > (uint32_t)BCAP + 768u + 64u)
#else
    if (false)
#endif
        return snk_fail(SNK_E_INTERNAL, err, errcap, "bucket-local graph: chunk capacity %d is below the count table's limit", BCAP);
    snk_phase_timer tm(st);
    tm.mark();  // 0
    snk_bl_state B;
    memset(&B, 0, sizeof B);
    B.tab = tab;
    B.K = K;
    B.world = 1;
    B.do_prune = do_prune;
    int rc = bl_prune_impl<K, GR>(ctx, st, &B, err, errcap);
    if (rc) return rc;
    tm.mark();  // 1
    tm.mark();  // 2
    out->n_boundary = B.n_boundary;
    if (want_unitigs) {
        snk_frag_out fo;
        memset(&fo, 0, sizeof fo);
        bl_dist_args da;
        memset(&da, 0, sizeof da);
        if ((rc = bl_fragments_impl<K, false, GR>(ctx, st, &B, da, &fo, err, errcap))) return rc;
        tm.mark();  // 3
        snk_join_out jo;
        rc = snk_dist_join(ctx, st, K, fo.n_frags, fo.nk, fo.hl_self, fo.hl_nb, fo.boff, fo.bases, fo.total_bases, &jo, err, errcap, fo.fgroup, fo.sfrag, 2 * n);
        if (rc) return rc;
        tm.mark();  // 4
        out->n_unitigs = jo.n_unitigs;
        out->total_bases = jo.total_bases;
        out->unitig_off = jo.unitig_off;
        out->unitig_bases = jo.unitig_bases;
        out->unitig_group = jo.unitig_group;
        out->n_circles = jo.n_circles;
        out->rank_rounds = jo.rank_rounds;
        out->n_fragments = fo.n_frags;
    } else {
        G_ALLOC(out->unitig_off, uint64_t, 2);
        SNK_HIP_TRY(hipMemsetAsync(out->unitig_off, 0, 16, st));
        G_ALLOC(out->unitig_bases, uint8_t, 16);
        tm.mark();
        tm.mark();
    }
    out->ctx = B.ctx;
    out->counts = B.counts;
    if (sort_table) {
        uint64_t *v_in, *v_out;
        snk_u128* k_out;
        G_ALLOC(v_in, uint64_t, n + 1);
        G_ALLOC(v_out, uint64_t, n + 1);
        G_ALLOC(k_out, snk_u128, n + 1);
        hipLaunchKernelGGL(bl_pack_vals_kernel, dim3(nblk(n)), dim3(TB), 0, st, B.counts, B.ctx, n, v_in);
        rc = snk_graph_sort(ctx, st, K, n, tab->keys, v_in, k_out, v_out, err, errcap);
        if (rc) return rc;
        hipLaunchKernelGGL(bl_unpack_vals_kernel, dim3(nblk(n)), dim3(TB), 0, st, v_out, n, B.counts, B.ctx);
        *keys_final = k_out;
    }
    if ((rc = snk_spectrum(ctx, st, B.counts, n, &out->spectrum, &out->spectrum_bins, err, errcap))) return rc;
    tm.mark();  // 5
    SNK_HIP_TRY(hipGetLastError());
    if (ms) { ms[0] = tm.ms(0, 1); ms[1] = tm.ms(1, 2); ms[2] = tm.ms(2, 3); ms[3] = tm.ms(3, 4); ms[4] = tm.ms(4, 5); }
    return SNK_OK;
}

int snk_local_graph(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_table* tab, uint32_t do_prune, bool want_unitigs,
                    bool sort_table, bool grouped, snk_graph_out* out, snk_u128** keys_final, float* ms, char* err, size_t errcap) {
    if (tab->sorted) return snk_fail(SNK_E_INTERNAL, err, errcap, "bucket-local graph needs the table in chunk order");
    if (grouped) {
        if (K != 48) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "grouped graphs need K=48");
        return local_graph_impl<48, true>(ctx, st, tab, do_prune, want_unitigs, sort_table, out, keys_final, ms, err, errcap);
    }
    if (K == 48) return local_graph_impl<48, false>(ctx, st, tab, do_prune, want_unitigs, sort_table, out, keys_final, ms, err, errcap);
    if (K == 60) return local_graph_impl<60, false>(ctx, st, tab, do_prune, want_unitigs, sort_table, out, keys_final, ms, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}

// ---- sharded stages (snk_dist.hip drives them; the host runs the exchanges in between)
int snk_bl_dist_plan(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, unsigned long long* h_qcount, char* err, size_t errcap) {
    B->force_dist = 1;
    const uint64_t n = B->tab->n;
    G_ALLOC(B->qcount, unsigned long long, B->world + 4);      // (+ the words the step appends to the exchange of these counts)
    G_ALLOC(B->qcursor, unsigned long long, B->world + 1);
    G_ALLOC(B->rq_idx, uint32_t, 2 * n + 2);
    G_ALLOC(B->rq_meta, uint16_t, 2 * n + 2);
    SNK_HIP_TRY(hipMemsetAsync(B->qcount, 0, (B->world + 1) * 8ull, st));
    SNK_HIP_TRY(hipMemsetAsync(B->rq_idx, 0xFF, (2 * n + 2) * 4, st));
    if (h_qcount) for (uint32_t r = 0; r < B->world; ++r) h_qcount[r] = 0;
    if (n == 0) {
        G_ALLOC(B->ctx, uint8_t, 16);
        G_ALLOC(B->counts, uint32_t, 4);
        G_ALLOC(B->index, unsigned long long, 1024);
        SNK_HIP_TRY(hipMemsetAsync(B->index, 0, 1024 * 8, st));
        B->index_mask = 1023;
        return SNK_OK;
    }
    if (n >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 retained k-mers on one GPU (%llu)", (unsigned long long)n);
    int rc = B->K == 48 ? bl_prune_impl<48, false>(ctx, st, B, err, errcap) : bl_prune_impl<60, false>(ctx, st, B, err, errcap);
    if (rc) return rc;
    const size_t lds = (size_t)B->world * 12 + 16;
    const unsigned grid = (unsigned)((n + QSPAN - 1) / QSPAN);
    if (B->K == 48) hipLaunchKernelGGL((bl_query_kernel<48, false>), dim3(grid), dim3(TB), lds, st, B->tab->keys, (const uint8_t*)B->premote, n, B->NB_total, B->NBl, B->world, ctx->mlen, B->qcount, (unsigned long long*)nullptr);
    else hipLaunchKernelGGL((bl_query_kernel<60, false>), dim3(grid), dim3(TB), lds, st, B->tab->keys, (const uint8_t*)B->premote, n, B->NB_total, B->NBl, B->world, ctx->mlen, B->qcount, (unsigned long long*)nullptr);
    SNK_HIP_TRY(hipGetLastError());
    if (h_qcount) {
        SNK_HIP_TRY(hipMemcpyAsync(h_qcount, B->qcount, B->world * 8ull, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
    }
    return SNK_OK;
}
int snk_bl_dist_fill(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, const unsigned long long* d_qoff, void* d_qbuf, char* err, size_t errcap) {
    const uint64_t n = B->tab->n;
    if (n == 0) return SNK_OK;
    SNK_HIP_TRY(hipMemcpyAsync(B->qcursor, d_qoff, (B->world + 1) * 8ull, hipMemcpyDeviceToDevice, st));
    const size_t lds = (size_t)B->world * 12 + 16;
    const unsigned grid = (unsigned)((n + QSPAN - 1) / QSPAN);
    if (B->K == 48) hipLaunchKernelGGL((bl_query_kernel<48, true>), dim3(grid), dim3(TB), lds, st, B->tab->keys, (const uint8_t*)B->premote, n, B->NB_total, B->NBl, B->world, ctx->mlen, B->qcursor, (unsigned long long*)d_qbuf);
    else hipLaunchKernelGGL((bl_query_kernel<60, true>), dim3(grid), dim3(TB), lds, st, B->tab->keys, (const uint8_t*)B->premote, n, B->NB_total, B->NBl, B->world, ctx->mlen, B->qcursor, (unsigned long long*)d_qbuf);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
int snk_bl_dist_fragments(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, const unsigned long long* d_node_off, unsigned long long my_node_off,
                          snk_frag_out* out, char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    const uint64_t n = B->tab->n;
    {
        int rc0 = snk_spectrum(ctx, st, n ? B->counts : nullptr, n, &out->spectrum, &out->spectrum_bins, err, errcap);
        if (rc0) return rc0;
    }
    if (n == 0) {
        G_ALLOC(out->boff, uint64_t, 2);
        SNK_HIP_TRY(hipMemsetAsync(out->boff, 0, 16, st));
        G_ALLOC(out->bases, uint8_t, 16);
        G_ALLOC(out->nk, uint32_t, 4);
        G_ALLOC(out->hl_self, unsigned long long, 2);
        G_ALLOC(out->hl_nb, unsigned long long, 2);
        return SNK_OK;
    }
    bl_dist_args da;
    da.premote = B->premote;
    da.rq_idx = B->rq_idx;
    da.rq_meta = B->rq_meta;
    da.node_off = d_node_off;
    da.my_node_off = my_node_off;
    return B->K == 48 ? bl_fragments_impl<48, true, false>(ctx, st, B, da, out, err, errcap) : bl_fragments_impl<60, true, false>(ctx, st, B, da, out, err, errcap);
}
