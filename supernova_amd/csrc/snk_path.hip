// snk_path.hip -- f1: every read onto the unitig graph, on the device.
//
// What it replaces:
//   dictionary fill  k-mer -> (edge, offset)     lib/assembly/src/paths/long/BuildReadQGraph48.cc:1656-1664 (buildGraphFromMSP),
//                                                == KDef::set in buildEdges (:327-512)
//   Pather::path                                 :705-748    seed by dictionary look-up, extend by exact match, gaps in between
//   HBVPather::algorithmTwo                      :1217-1336  seeds on hanging edges, captured gaps, short last seed, connectivity
//   pathPartsToReadPath                          :1393-1428
//   ExtendReadPath::attemptLeft/RightExtension   paths/long/ExtendReadPath.cc:108-358 (+ scoreLeft/RightOverlap :15-106)
//   pathReads                                    :1441-1469  (the reference: one thread per 500 k reads, a hash-set look-up per k-mer)
//
// Layout.  The dictionary is an open-addressing table in HBM over all k-mers of all unitigs, two slots per k-mer: the 128-bit
// canonical key and a u64 (unitig, strand of the canonical form inside the unitig, offset) -- a look-up is exact with ONE
// dependent load on a miss and two on a hit.  One wave per read, and the whole thing is latency bound (a read is a chain of
// dependent random accesses), so the chain is kept short: the first k-mer is looked up by one lane (99 % of the reads of a
// deep data set start on the graph); only after a miss do the 64 lanes look up 64 consecutive positions at once (a read with
// an error has K missing k-mers in a row -- one round instead of 48 dependent look-ups); the exact-match extension compares
// 128 bases per step with two ballots; per-unitig records (offset, length, both HBV edges, hanging-seed flags) come with one
// 32-byte load.  The bookkeeping that follows (a handful of parts per read) is the reference's sequential logic, run by lane 0
// over the parts in LDS; paths leave through a wave-level reservation and are put in read order by a gather.
#include <string.h>

#include <vector>

#include <rocprim/rocprim.hpp>
#include "snk_stages.h"

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"

namespace {

constexpr int PCAP = 112;      // path parts per read (a 150-base read has at most 103 k-mers)
constexpr int PMAX = 64;       // edges per read path
// The parts of sixteen reads sit in LDS.  At full capacity that is 29 KB per workgroup and four workgroups per CU; a read of a
// deep data set has one to three parts, so the kernel runs with room for PCAP1 parts / PMAX1 edges (8 KB, occupancy limited
// by registers) and the few reads that need more are listed and redone by the full-capacity variant.
constexpr int PCAP1 = 20;
constexpr int PMAX1 = 16;
constexpr int GPARTS = 4;      // parts per read that travel to the finishing kernel
constexpr int PMT = 8;         // edges per read there

struct path_graph {            // device arrays
    const uint64_t* uoff;      // [U+1] unitig offsets into ubases (device order)
    const uint8_t* ubases;     // 1 byte per base
    const uint32_t* upack;     // the same bases, 2 bits each, 16 per word, first base in the top bits (one word of slack at either end): what the
                               //     seed check and the exact-match extension read (a read's whole window is 40 bytes instead of 150 byte loads)
    const uint4* uinfo;        // [U] x 2: {offset lo, offset hi, bases, flags}, {fwd edge, rev edge, 0, 0}; flags bit 0 / 1: a seed on the
                               //     forward / reverse-complement copy is dropped (hanging edge rule, BuildReadQGraph48.cc:1240-1247)
    const int32_t *fwd, *rev;  // [U] HBV edge of the unitig read forward / reverse-complemented
    const int32_t *vleft, *vright;         // [E]
    const int32_t *e_unitig;               // [E] device index of the unitig the edge is a copy of
    const uint8_t* e_rc;                   // [E]
    const int32_t *to_off, *to_v, *to_e;   // [N+1], [E], [E]: in-edges of a vertex, AddEdge order (graph/DigraphTemplate.h:2572-2582)
    const int32_t *from_off, *from_v, *from_e;
    const unsigned long long* dslot;   // dictionary, 8 bytes per slot: fingerprint (30 bits) | strand (1) | position (33 bits: first base of
                               //   the k-mer in the concatenated unitigs); all ones: empty.  strand: the unitig holds the reverse complement
                               //   of the canonical form.  A probe is one 8-byte access, a claim ONE 64-bit compare-and-swap; a fingerprint
                               //   match is VERIFIED against the unitig's bases, so look-ups stay exact (2^-30 false candidates per probe of
                               //   an occupied slot: a few reads per 10^8 take the verified path).  24 bytes per unitig k-mer.
    const uint32_t* ublk;      // unitig that holds base 256 b of the concatenation (position -> unitig: this entry, then a step or two along uoff)
    uint64_t dcap;             // slots: 3 per k-mer, not a power of two
    unsigned long long fp_mask; // 30 ones; SNK_PATH_FP_MASK (tests) narrows the fingerprint so that false matches happen and the verified path runs
    // Minimiser index (round 4; ment != NULL: the look-ups go through it and dslot is not built).  A k-mer lies on the graph iff its
    // minimiser -- the 16-mer with the smallest ordering key among its K-15, the leftmost on ties -- occurs in a unitig at the matching
    // place and the K bases around it agree.  The index lists the places a unitig k-mer's window picks (its leftmost AND its rightmost
    // minimum: a read may run against the unitig's strand), ~2 / (K-14) of the unitig bases: 8 bytes each, sorted by key behind a
    // directory over the key's top bits -- 0.14 GB for the bench graph's 0.27 G k-mers where the k-mer dictionary takes 6.4 GB, small
    // enough to live in the 256 MB Infinity Cache together with the packed unitigs the candidates are verified against.
    const unsigned long long* ment;    // [n_ment] key low 30 bits << 34 | the unitig's 16-mer is the reverse complement of the canonical one << 33 | position
    const uint32_t* mdir;      // [(1 << mdir_bits) + 1] first entry of every key prefix
    uint32_t mdir_bits;
    uint32_t idx_dbg;          // SNK_PATH_IDX_DBG (timing aid, results invalid): 1 = window minimum only, 2 = + directory and entries, 3 = + base comparison (no unitig-bounds check)
};

template <int K>
__device__ __forceinline__ snk_kmer kmer_at(const uint32_t* words, uint32_t nwords, uint32_t pos) {
    const uint32_t wi = pos >> 4, sh = 2u * (pos & 15u);
    uint32_t W[5];
#pragma unroll
    for (uint32_t q = 0; q < 5; ++q) W[q] = wi + q < nwords ? words[wi + q] : 0u;
    const uint64_t A = ((uint64_t)W[0] << 32) | W[1], B = ((uint64_t)W[2] << 32) | W[3], C = (uint64_t)W[4] << 32;
    snk_kmer f;
    f.hi = sh ? ((A << sh) | (B >> (64 - sh))) : A;
    const uint64_t lo = sh ? ((B << sh) | (C >> (64 - sh))) : B;
    f.lo = lo & ~((1ull << (128 - 2 * K)) - 1ull);
    return f;
}
// the same from a long packed array (no bound: the caller's array has slack behind its last base)
template <int K>
__device__ __forceinline__ snk_kmer kmer_at64(const uint32_t* words, uint64_t pos) {
    const uint64_t wi = pos >> 4;
    const uint32_t sh = 2u * ((uint32_t)pos & 15u);
    const uint64_t A = ((uint64_t)words[wi] << 32) | words[wi + 1], B = ((uint64_t)words[wi + 2] << 32) | words[wi + 3], C = (uint64_t)words[wi + 4] << 32;
    snk_kmer f;
    f.hi = sh ? ((A << sh) | (B >> (64 - sh))) : A;
    const uint64_t lo = sh ? ((B << sh) | (C >> (64 - sh))) : B;
    f.lo = lo & ~((1ull << (128 - 2 * K)) - 1ull);
    return f;
}
__device__ __forceinline__ uint64_t dict_slot(snk_kmer c, uint64_t cap) {
    uint32_t h1, h2;
    snk_kmer_hash2(c, &h1, &h2);
    return __umul64hi(((uint64_t)h1 << 32) | h2, cap);          // multiply-shift range reduction
}

// 64-bit fingerprint of a canonical k-mer, independent of the slot hash; never all ones (that is the empty slot)
__device__ __forceinline__ unsigned long long dict_fp(snk_kmer c) {
    unsigned long long x = c.hi * 0x9E3779B97F4A7C15ull ^ (c.lo + 0xD1B54A32D192ED03ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x += c.lo * 0x94D049BB133111EBull; x ^= x >> 31;
    return x == ~0ull ? 0x7FFFFFFFFFFFFFFFull : x;
}
// ---- dictionary build.  A thread takes DB_RUN consecutive base positions of the concatenated unitigs: the first k-mer is read
// base by base, the following ones roll (one byte each).  A slot IS one 64-bit word (fingerprint | strand | position): it is
// claimed and filled by ONE compare-and-swap on all ones; no lock array, no second store, nothing to clear but the slots.
// Round 2 had 32-byte slots holding the whole key (80 bytes per unitig k-mer with the lock word), early round 3 16-byte slots
// (fingerprint word + value word, 40 bytes per k-mer, 30 ms: a CAS and a store to a random sector each); 24 bytes and 17 ms now
// (bound by the device's random atomics).
constexpr int DB_RUN = 16;
// 16 bases of the byte array -> one word (the same bit order as the packed read rows)
__global__ void __launch_bounds__(256) upack_kernel(const uint8_t* __restrict__ ubases, uint64_t total, uint32_t* __restrict__ out, uint64_t n_words) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    uint32_t v = 0;
    const uint64_t b0 = w * 16;
    if (b0 + 16 <= total && ((uintptr_t)(ubases + b0) & 15u) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(ubases + b0);
        const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t y = x[j] & 0x03030303u;               // bytes b0..b3 = bases 4j..4j+3 (little endian: base 4j in the low byte)
            const uint32_t four = ((y & 3u) << 6) | (((y >> 8) & 3u) << 4) | (((y >> 16) & 3u) << 2) | ((y >> 24) & 3u);
            v |= four << (24 - 8 * j);
        }
    } else {
        for (int j = 0; j < 16; ++j) { const uint64_t b = b0 + j; v = (v << 2) | (b < total ? (uint32_t)ubases[b] & 3u : 0u); }
    }
    out[w] = v;
}
// unitig of every 256th base of the concatenation (largest u with uoff[u] <= 256 b)
__global__ void __launch_bounds__(256) ublk_kernel(const uint64_t* __restrict__ uoff, uint64_t U, uint64_t n_blk, uint32_t* __restrict__ ublk) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blk) return;
    const uint64_t p = b << 8;
    uint64_t lo = 0, hi = U;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (uoff[mid] <= p) lo = mid; else hi = mid; }
    ublk[b] = (uint32_t)lo;
}
constexpr uint64_t UPAD = 512;      // bases of slack in front of (and behind) the packed unitigs: a window may start 270 bases before a hit or run 270 past it
// 16 bases starting at base p of a packed array (two words, one funnel shift)
__device__ __forceinline__ uint32_t packed16(const uint32_t* P, uint64_t p) {
    const uint64_t wi = p >> 4;
    const uint32_t sh = 2u * ((uint32_t)p & 15u);
    const uint32_t w0 = P[wi], w1 = P[wi + 1];
    return (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> 32);
}

template <int K>
__global__ void __launch_bounds__(256) dict_build_kernel(const uint64_t* __restrict__ uoff, const uint8_t* __restrict__ ubases, uint64_t U, uint64_t total,
                                                         unsigned long long* __restrict__ dslot, uint64_t dcap, unsigned long long fp_mask) {
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * DB_RUN;
    if (p0 >= total) return;
    uint64_t lo = 0, hi = U;                     // largest u with uoff[u] <= p0
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (uoff[mid] <= p0) lo = mid; else hi = mid; }
    uint64_t ue = uoff[lo + 1];
    snk_kmer f;
    f.hi = 0; f.lo = 0;
    bool have = false;                            // f = the k-mer that starts at p - 1
    for (uint64_t p = p0; p < p0 + DB_RUN && p < total; ++p) {
        while (p >= ue) { ++lo; ue = uoff[lo + 1]; have = false; }
        if (p + K > ue) { have = false; continue; }                           // the last K-1 positions of a unitig start no k-mer
        if (have) f = snk_kmer_succ<K>(f, ubases[p + K - 1] & 3u);
        else {
            f.hi = 0; f.lo = 0;
            for (int q = 0; q < K; ++q) f = snk_kmer_succ<K>(f, ubases[p + q] & 3u);
            have = true;
        }
        const snk_kmer r = snk_kmer_rc<K>(f);
        const bool rev = snk_kmer_lt(r, f);
        const snk_kmer c = rev ? r : f;
        const unsigned long long val = ((dict_fp(c) >> 34 & fp_mask) << 34) | ((unsigned long long)(rev ? 1u : 0u) << 33) | p;
        uint64_t s = dict_slot(c, dcap);
        for (;;) {
            if (atomicCAS(dslot + s, ~0ull, val) == ~0ull) break;
            if (++s == dcap) s = 0;
        }
    }
}
// ---- minimiser index build.  A thread takes MM_RUN consecutive windows (k-mer start positions) of the concatenated unitigs: the
// ordering keys of the MM_RUN + W 16-mers they cover are computed once (registers, static indices), every window's leftmost and
// rightmost minimum falls out of W compares.  A place is listed when the window's choice moves (L and R only move right along a
// unitig), so every chosen place is listed once per role.  COUNT pass: places per workgroup; FILL pass: the same scan writes them
// at the workgroup's offset -- no atomics, no guessed capacity.
constexpr int MM_RUN = 16;
__device__ __forceinline__ uint32_t mm_key(uint32_t x, uint32_t* orient) {
    const uint32_t rx = snk_rev2_32(~x);
    *orient = rx < x ? 1u : 0u;
    return snk_minimizer_key(x, rx);
}
template <int K, bool FILL>
__global__ void __launch_bounds__(256) mm_scan_kernel(const uint64_t* __restrict__ uoff, const uint32_t* __restrict__ upack, uint64_t U, uint64_t total,
                                                      uint32_t* __restrict__ wg_count, const uint64_t* __restrict__ wg_off, uint32_t* __restrict__ okey,
                                                      unsigned long long* __restrict__ oval) {
    constexpr int W = K - 15;                      // 16-mers per k-mer
    constexpr int NK = MM_RUN + W;                 // keys of windows t0 - 1 .. t0 + MM_RUN - 1
    constexpr int NWD = (NK + 15 + 15) / 16 + 1;   // aligned words that hold their bases
    __shared__ uint32_t lcur;                      // places of this workgroup so far
    if (threadIdx.x == 0) lcur = 0;
    __syncthreads();
    const uint64_t t0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * MM_RUN;
    const uint64_t out0 = FILL ? wg_off[blockIdx.x] : 0ull;
    uint32_t mine = 0;
    auto emit = [&](uint32_t bk, unsigned long long orib, int at, uint64_t tbase) {
        if (FILL) {
            const uint32_t s = atomicAdd(&lcur, 1u);     // (the order inside a workgroup does not matter: the places are sorted by key afterwards)
            okey[out0 + s] = bk;
            oval[out0 + s] = ((unsigned long long)(bk & 0x3FFFFFFFu) << 34) | (((orib >> at) & 1ull) << 33) | (tbase + (uint64_t)at);
        } else ++mine;
    };
    if (t0 < total) {
        // bases from t0 - 1 on, aligned to a word boundary (the slack in front of the packed unitigs covers t0 = 0)
        const uint64_t b0 = UPAD + t0 - 1;
        const uint32_t* wp = upack + (b0 >> 4);
        const uint32_t sh = 2u * ((uint32_t)b0 & 15u);
        uint32_t A[NWD];
#pragma unroll
        for (int q = 0; q < NWD; ++q) A[q] = (uint32_t)(((((uint64_t)wp[q] << 32) | wp[q + 1]) << sh) >> 32);
        uint32_t key[NK], ori = 0;                 // key j: the 16-mer at t0 - 1 + j
        unsigned long long orib = 0;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const uint32_t x = (uint32_t)(((((uint64_t)A[j >> 4] << 32) | A[(j >> 4) + 1]) << (2 * (j & 15))) >> 32);
            key[j] = mm_key(x, &ori);
            orib |= (unsigned long long)ori << j;
        }
        uint64_t lo = 0, hi = U;                   // unitig of t0 (largest u with uoff[u] <= t0)
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (uoff[mid] <= t0) lo = mid; else hi = mid; }
        uint64_t us = uoff[lo], ue = uoff[lo + 1];
        int pl = -1, pr = -1;                      // the choices of the window before (relative to t0 - 1), -1: none in this unitig
        {   // window t0 - 1, if it is a window of the same unitig
            if (t0 > us) {
                uint32_t bk = key[0]; int bl = 0, br = 0;
#pragma unroll
                for (int j = 1; j < W; ++j) { if (key[j] < bk) { bk = key[j]; bl = j; br = j; } else if (key[j] == bk) br = j; }
                if (t0 - 1 + K <= ue) { pl = bl; pr = br; }
            }
        }
#pragma unroll
        for (int w = 0; w < MM_RUN; ++w) {
            const uint64_t t = t0 + w;
            if (t >= ue && t < total) { ++lo; us = ue; ue = uoff[lo + 1]; pl = -1; pr = -1; while (t >= ue) { ++lo; us = ue; ue = uoff[lo + 1]; } }
            if (t < total && t + K <= ue) {
                uint32_t bk = key[w + 1]; int bl = w + 1, br = w + 1;
#pragma unroll
                for (int j = 1; j < W; ++j) { const uint32_t kj = key[w + 1 + j]; if (kj < bk) { bk = kj; bl = w + 1 + j; br = bl; } else if (kj == bk) br = w + 1 + j; }
                if (bl != pl) emit(bk, orib, bl, t0 - 1);
                if (br != bl && br != pr) emit(bk, orib, br, t0 - 1);
                pl = bl; pr = br;
            } else { pl = -1; pr = -1; }
        }
    }
    if (!FILL) {
        snk_wave_add(&lcur, mine);
        __syncthreads();
        if (threadIdx.x == 0) wg_count[blockIdx.x] = lcur;
    }
}
__global__ void __launch_bounds__(256) mm_widen_kernel(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n) out[i] = i < n ? in[i] : 0ull;
}
__global__ void __launch_bounds__(256) mm_hist_kernel(const uint32_t* __restrict__ key, uint64_t n, uint32_t shift, uint32_t* __restrict__ hist) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&hist[key[i] >> shift], 1u);
}

// the ordering keys of a read's 16-mers, by the lanes of its group (keys[i]: the 16-mer at base i)
__device__ __forceinline__ void mm_read_keys(const uint32_t* row, uint32_t n, uint32_t cap, int sub, int gs, uint32_t* keys) {
    const uint32_t nk = n >= 16u ? (n - 15u < cap ? n - 15u : cap) : 0u;
    for (uint32_t i = (uint32_t)sub; i < nk; i += (uint32_t)gs) { uint32_t o; keys[i] = mm_key(packed16(row, i), &o); }
}
// dict_find through the minimiser index: the k-mer at base `pos` of the read (keys: mm_read_keys of that read).  Always exact.
template <int K>
__device__ __forceinline__ bool mm_find(const path_graph& G, const uint32_t* row, const uint32_t* keys, uint32_t pos, uint32_t* hrc, uint64_t* hpos) {
    constexpr uint32_t W = K - 15;
    uint32_t best = keys[pos], bq = 0;
    for (uint32_t j = 1; j < W; ++j) { const uint32_t k = keys[pos + j]; if (k < best) { best = k; bq = j; } }
    if (G.idx_dbg == 1) return best == 0x12345u && bq == 99u;
    const uint32_t x = packed16(row, pos + bq), rx = snk_rev2_32(~x);
    const uint32_t orient_r = rx < x ? 1u : 0u;
    const uint32_t b = best >> (32u - G.mdir_bits);
    const uint32_t lo = G.mdir[b], hi = G.mdir[b + 1];
    const unsigned long long want = (unsigned long long)(best & 0x3FFFFFFFu);
    const snk_kmer f = kmer_at<K>(row, 20, pos);
    const snk_kmer fr = snk_kmer_rc<K>(f);
    for (uint32_t c = lo; c < hi; ++c) {
        const unsigned long long e = G.ment[c];
        if ((e >> 34) != want) continue;
        if (G.idx_dbg == 2) { if (e == 0x123456789ull) *hrc = 1; continue; }
        const uint64_t P = e & 0x1FFFFFFFFull;
        const uint32_t orient_u = (uint32_t)(e >> 33) & 1u;
        // same strand: the k-mer starts bq bases before the place; opposite strand: its reverse complement starts K - 16 - bq before it
        for (uint32_t rc = 0; rc < 2; ++rc) {
            if (x != rx && (orient_u ^ orient_r) != rc) continue;         // (a 16-mer that is its own reverse complement fits both ways)
            const uint32_t back = rc ? (uint32_t)K - 16u - bq : bq;
            if (P < back) continue;
            const uint64_t t = P - back;
            const snk_kmer g = kmer_at64<K>(G.upack, UPAD + t);      // (the packed array has UPAD bases of slack at either end)
            if (!snk_kmer_eq(g, rc ? fr : f)) continue;
            if (G.idx_dbg == 3) { *hrc = rc; *hpos = t; return true; }
            uint32_t u = G.ublk[t >> 8];                                  // the K bases lie inside ONE unitig?
            while (G.uoff[u + 1] <= t) ++u;
            if (t + K > G.uoff[u + 1]) continue;
            *hrc = rc;
            *hpos = t;
            return true;
        }
    }
    return false;
}
// per-unitig record for the pather
__global__ void __launch_bounds__(256) uinfo_kernel(const uint64_t* __restrict__ uoff, uint64_t U, const int32_t* __restrict__ fwd, const int32_t* __restrict__ rev,
                                                    const uint8_t* __restrict__ e_drop, uint4* __restrict__ uinfo) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const uint64_t o = uoff[u];
    const uint32_t len = (uint32_t)(uoff[u + 1] - o);
    const int32_t f = fwd[u], r = rev[u];
    uinfo[2 * u] = make_uint4((uint32_t)o, (uint32_t)(o >> 32), len, (uint32_t)e_drop[f] | ((uint32_t)e_drop[r] << 1));
    uinfo[2 * u + 1] = make_uint4((uint32_t)f, (uint32_t)r, 0u, 0u);
}
// ---- the sequential part of a read's path, run by one lane (parts in LDS)
struct ppart { uint32_t unitig; uint32_t off_rc; uint32_t len; uint32_t elen; };      // elen == 0: a gap (only len counts); off_rc = offset | rc << 31
__device__ __forceinline__ bool p_gap(const ppart& p) { return p.elen == 0; }
__device__ __forceinline__ uint32_t p_off(const ppart& p) { return p.off_rc & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t p_rc(const ppart& p) { return p.off_rc >> 31; }
__device__ __forceinline__ bool p_same_edge(const ppart& a, const ppart& b) { return a.unitig == b.unitig && p_rc(a) == p_rc(b); }
__device__ __forceinline__ ppart make_gap(uint32_t len) { ppart g; g.unitig = 0; g.off_rc = 0; g.len = len; g.elen = 0; return g; }

__device__ __forceinline__ uint32_t u_len(const path_graph& G, uint32_t u) { return G.uinfo[2 * (uint64_t)u].z; }
__device__ __forceinline__ uint32_t u_base(const path_graph& G, uint32_t u, uint32_t rc, uint32_t i) {
    const uint4 q = G.uinfo[2 * (uint64_t)u];
    const uint8_t* b = G.ubases + (((uint64_t)q.y << 32) | q.x);
    return rc ? (uint32_t)(b[q.z - 1 - i] ^ 3u) & 3u : (uint32_t)b[i] & 3u;
}
__device__ __forceinline__ uint32_t e_len(const path_graph& G, int32_t e) { return u_len(G, (uint32_t)G.e_unitig[e]); }
__device__ __forceinline__ int to_size(const path_graph& G, int32_t v) { return G.to_off[v + 1] - G.to_off[v]; }
__device__ __forceinline__ int from_size(const path_graph& G, int32_t v) { return G.from_off[v + 1] - G.from_off[v]; }
__device__ __forceinline__ int32_t part_edge(const path_graph& G, const ppart& p) { const uint4 q = G.uinfo[2 * (uint64_t)p.unitig + 1]; return (int32_t)(p_rc(p) ? q.y : q.x); }

// PathPart::isConformingCapturedGap :659-666 (unsigned arithmetic as written)
__device__ bool conforming_gap(const ppart* p, uint32_t max_jitter) {
    const ppart &prev = p[-1], &next = p[1];
    uint32_t graph_dist = p_off(next) - (p_off(prev) + prev.len);
    if (!p_same_edge(prev, next)) graph_dist += prev.elen;
    const int32_t d = (int32_t)(p->len - graph_dist);
    return (uint32_t)(d < 0 ? -d : d) <= max_jitter;
}
// Pather::isJoinable :808-814: the last K-1 bases of both edges, each in its part's orientation (as written in the reference)
template <int K>
__device__ bool joinable(const path_graph& G, const ppart& a, const ppart& b) {
    if (a.unitig == b.unitig) return true;
    const uint32_t la = u_len(G, a.unitig), lb = u_len(G, b.unitig);
    for (uint32_t q = 0; q + 1 < (uint32_t)K; ++q)
        if (u_base(G, a.unitig, p_rc(a), la - (K - 1) + q) != u_base(G, b.unitig, p_rc(b), lb - (K - 1) + q)) return false;
    return true;
}
__device__ __forceinline__ uint32_t read_base(const uint32_t* row, uint32_t p) { return (row[p >> 4] >> (30 - 2 * (p & 15))) & 3u; }
// scoreLeftOverlap / scoreRightOverlap, ExtendReadPath.cc:15-106: decay 0.2, Q2 counted as Q20, 10 per read base left over
template <int K>
__device__ uint32_t score_overlap(const path_graph& G, const uint32_t* row, const uint8_t* __restrict__ quals /* the read's row in HBM */, uint32_t n, uint32_t start, int32_t e,
                                  bool left) {
    const uint32_t esz = e_len(G, e), u = (uint32_t)G.e_unitig[e], erc = G.e_rc[e];
    uint32_t qsum = 0, penalty = 0, steps = 0;
    for (;; ++steps) {
        if (steps >= start) break;
        uint32_t rp, ep;
        if (!left) { rp = n - start + steps; ep = K - 1 + steps; if (ep >= esz) break; }
        else { rp = start - 1 - steps; if (steps + K > esz) break; ep = esz - K - steps; }
        if (read_base(row, rp) != u_base(G, u, erc, ep)) {
            const uint32_t q = quals[rp] == 2 ? 20u : (uint32_t)quals[rp];
            penalty += q;
            qsum += penalty;
        } else if (penalty > 0) penalty = (uint32_t)((double)penalty - 0.2 * (double)penalty);      // penalty -= (pDecay*penalty)
    }
    return qsum + 10u * (start - steps);
}
// attemptLeftwardExtension :123-239 / attemptRightwardExtension :242-358
template <int K>
__device__ bool extend_once(const path_graph& G, int32_t* path, int* np, int pmax, int32_t* offset, const uint32_t* row, const uint8_t* quals, uint32_t n, bool left) {
    if (!*np) return false;
    uint64_t last_gap;
    if (left) {
        if (*offset >= 0) return false;
        last_gap = (uint64_t)(-(int64_t)*offset);
    } else {
        int32_t g = (int32_t)n + *offset;
        for (int i = 0; i < *np; ++i) g -= (int32_t)(e_len(G, path[i]) - K + 1);
        g -= K - 1;
        if (g < 10) return false;
        last_gap = (uint64_t)g;
    }
    if (last_gap < 10) return false;
    const int32_t v = left ? G.vleft[path[0]] : G.vright[path[*np - 1]];
    const int32_t* off = left ? G.to_off : G.from_off;
    const int32_t* ee = (left ? G.to_e : G.from_e) + off[v];
    const int32_t* vd = (left ? G.to_v : G.from_v) + off[v];
    const int ne = off[v + 1] - off[v];
    int nlong = 0, nshort = 0;
    bool short_same = true;
    int32_t short_dest = -1;
    for (int i = 0; i < ne; ++i) {
        const bool hanging = left ? (to_size(G, vd[i]) == 0 && from_size(G, vd[i]) == 1) : (from_size(G, vd[i]) == 0 && to_size(G, vd[i]) == 1);
        const bool lng = (uint64_t)e_len(G, ee[i]) - (K - 1) >= last_gap;
        nlong += lng ? 1 : 0;
        if (!lng && !hanging) { if (nshort && vd[i] != short_dest) short_same = false; short_dest = vd[i]; ++nshort; }
    }
    if (ne != 1 && nshort > 0) {
        if (nlong > 0 || !short_same) return false;
        if ((left ? to_size(G, short_dest) : from_size(G, short_dest)) != 1) return false;
    }
    int32_t least_edge = -1;
    uint32_t least = 0xFFFFFFFFu;
    for (int i = 0; i < ne; ++i) {
        const bool hanging = left ? (to_size(G, vd[i]) == 0 && from_size(G, vd[i]) == 1) : (from_size(G, vd[i]) == 0 && to_size(G, vd[i]) == 1);
        if (!hanging || ne == 1) {
            const uint32_t sc = score_overlap<K>(G, row, quals, n, (uint32_t)last_gap, ee[i], left);
            if (sc < least) { least_edge = ee[i]; least = sc; }
        }
    }
    if (least_edge == -1 || (uint64_t)least > last_gap * 10) return false;
    if (*np >= pmax) { *offset = 0x7FFFFFFF; return false; }      // marks the path as too long (reported, never truncated silently)
    if (left) {
        for (int i = *np; i > 0; --i) path[i] = path[i - 1];
        path[0] = least_edge;
        *offset += (int32_t)(e_len(G, least_edge) - K + 1);
    } else path[*np] = least_edge;
    ++*np;
    return true;
}

// algorithmTwo after Pather::path: m parts in LDS -> path[np], offset
template <int K>
__device__ void finish_path(const path_graph& G, ppart* parts, int m, const uint32_t* row, const uint8_t* quals, uint32_t n, int32_t* path, int pmax, int* np_out,
                            int32_t* off_out) {
    // seeds on hanging edges become gaps, adjacent gaps merge (:1235-1262) -- in place: the write index never passes the read index
    int m2 = 0;
    for (int i = 0; i < m; ++i) {
        ppart p = parts[i];
        if (!p_gap(p) && ((G.uinfo[2 * (uint64_t)p.unitig].w >> p_rc(p)) & 1u)) p = make_gap(p.len);      // precomputed per edge (host, drop_flags)
        if (p_gap(p) && m2 && p_gap(parts[m2 - 1])) parts[m2 - 1].len += p.len;
        else parts[m2++] = p;
    }
    m = m2;
    // a captured gap that does not fit the graph (:1268-1292)
    if (m >= 3) {
        uint32_t seeds = p_gap(parts[0]) ? 0u : 1u;
        for (int p = 1; p < m - 1; ++p) {
            if (!p_gap(parts[p])) { ++seeds; continue; }
            if (!conforming_gap(&parts[p], 3) || !joinable<K>(G, parts[p - 1], parts[p + 1])) {
                if (seeds > 1) {
                    ppart t = make_gap(parts[p - 1].len);
                    for (int q = p; q < m; ++q) t.len += parts[q].len;
                    parts[p - 1] = t;
                    m = p;
                } else {
                    for (int q = p + 1; q < m; ++q) parts[p].len += parts[q].len;
                    m = p + 1;
                }
                break;
            }
        }
    }
    // a last seed of <= 5 k-mers at the very start of an edge is not trusted (:1298-1312)
    if (p_gap(parts[m - 1]) && m > 1) {
        const ppart l2 = parts[m - 2];
        if (p_off(l2) == 0 && l2.len <= 5) { ppart last = parts[m - 1]; last.len += l2.len; m -= 2; parts[m++] = last; }
    } else if (!p_gap(parts[m - 1])) {
        if (p_off(parts[m - 1]) == 0 && parts[m - 1].len <= 5) parts[m - 1] = make_gap(parts[m - 1].len);
    }
    // pathPartsToReadPath :1393-1428
    int np = 0;
    int32_t offset = 0;
    int last = -1;
    bool too_long = false;
    for (int i = 0; i < m; ++i) {
        if (p_gap(parts[i])) continue;
        if (last >= 0 && p_same_edge(parts[last], parts[i])) continue;
        if (np < pmax) path[np++] = part_edge(G, parts[i]); else too_long = true;
        last = i;
    }
    if (np) offset = !p_gap(parts[0]) ? (int32_t)p_off(parts[0]) : (int32_t)p_off(parts[1]) - (int32_t)parts[0].len;
    // adjacent edges must share a vertex (:1316-1323)
    for (int i = 0; i + 1 < np; ++i) if (G.vright[path[i]] != G.vleft[path[i + 1]]) { np = i + 1; break; }
    // ExtendReadPath::attemptLeftRightExtension :108-119
    while (extend_once<K>(G, path, &np, pmax, &offset, row, quals, n, true)) {}
    while (extend_once<K>(G, path, &np, pmax, &offset, row, quals, n, false)) {}
    *np_out = (too_long || offset == 0x7FFFFFFF) ? -1 : np;
    *off_out = offset;
}

struct path_args {
    path_graph G;
    const uint32_t* rows; uint32_t row_words; uint32_t read_len;
    const uint8_t* quals; uint32_t qstride;
    const uint16_t* lens;
    uint64_t n_reads;
    int32_t* out_off;                 // [n] offset of the read on its first edge (ReadPath::mOffset)
    uint32_t* out_n;                  // [n + 1] edges per read
    int32_t* out_e0;                  // [n] first edge of the path (-1: none)
    unsigned long long* out_start;    // [n] position of the read's FURTHER edges in the scratch list
    int32_t* scratch;                 // second and later edges in completion order
    unsigned long long* cursor;       // [0] edges reserved, [1] error flags, [2] (unitig, barcode) keys reserved, [3] reads to redo
    uint64_t scratch_cap;
    // per-unitig barcode lists (the rest of SURVEY f4; tada's edge -> barcode sets, lib/tada/src/cmd_main_asm.rs:91-151,
    // debruijn.rs:115-131): a barcoded read contributes its barcode to every unitig one of its k-mers lies on -- exactly the
    // unitigs of its non-gap path parts, since Pather::path looks every k-mer up that no exact-match run covers
    const uint32_t* slow;             // MODE 1: the reads the fast pass left (gm == 0xFE), ascending
    uint64_t n_slow;
    uint32_t* redo;                   // reads that did not fit the small capacities (first pass: written through cursor[3]; second: read)
    uint64_t redo_cap, n_redo;
    uint32_t force_redo;              // SNK_PATH_REDO_ALL=1 (tests): the first pass hands every read to the second
    // split first pass: the group kernel stops after Pather::path and leaves the parts (at most GPARTS per read; more: redo list)
    // in HBM; a one-thread-per-read kernel does the sequential rest (algorithmTwo, extension) at full lane occupancy
    ppart* gparts;                    // [n][GPARTS]
    uint8_t* gm;                      // [n] parts of the read, 0xFF: handed to the redo list
    const int32_t* bc;                // raw barcode ids, or NULL: no lists
    unsigned long long* ub_first;     // [n] unitig << 32 | barcode of the read's first such unitig (~0: none)
    unsigned long long* ub_more;      // further ones, through cursor[2]
    uint64_t ub_cap;
};

// one dictionary look-up: canonical form, probe, the strand the read is on relative to the unitig.  A slot whose fingerprint
// matches is a candidate: the pather's group checks the seed's K bases against the unitig with all its lanes (the lines the
// exact-match extension reads next); should that ever fail -- 2^-64 per probe -- the read is redone with verify = true, where a
// match is checked here, base by base, and a false one just continues the probe sequence.  Look-ups are exact either way.
template <int K>
__device__ __forceinline__ bool dict_find(const path_graph& G, const uint32_t* row, uint32_t pos, uint32_t* hu, uint32_t* ho, uint32_t* hrc, bool verify,
                                          uint64_t* hpos = nullptr /* given: the hit's position in the concatenation instead of (unitig, offset) */) {
    const snk_kmer f = kmer_at<K>(row, 20, pos);
    const snk_kmer rk = snk_kmer_rc<K>(f);
    const bool rrev = snk_kmer_lt(rk, f);
    const snk_kmer c = rrev ? rk : f;
    const unsigned long long fp = dict_fp(c) >> 34 & G.fp_mask;
    uint64_t s = dict_slot(c, G.dcap);
    for (;;) {
        const unsigned long long k = G.dslot[s];
        if (k == ~0ull) return false;
        if ((k >> 34) == fp) {
            const uint64_t pos = k & 0x1FFFFFFFFull;
            const uint32_t urev = (uint32_t)(k >> 33) & 1u;
            bool ok = true;
            if (verify) {
                const uint8_t* ub = G.ubases + pos;
                snk_kmer g;
                g.hi = 0; g.lo = 0;
                for (int q = 0; q < K; ++q) g = snk_kmer_succ<K>(g, ub[q] & 3u);
                const snk_kmer cu = urev ? snk_kmer_rc<K>(g) : g;         // canonical form of the unitig's k-mer
                ok = snk_kmer_eq(cu, c);
            }
            if (ok) {
                *hrc = urev ^ (rrev ? 1u : 0u);           // read k-mer vs the unitig's k-mer: CF<K>::isRC, dna/CanonicalForm.h:85-91
                if (hpos) { *hpos = pos; return true; }
                // position -> unitig and offset (a k-mer never crosses a unitig boundary)
                uint32_t u = G.ublk[pos >> 8];
                while (G.uoff[u + 1] <= pos) ++u;
                *hu = u;
                *ho = (uint32_t)(pos - G.uoff[u]);
                return true;
            }
        }
        if (++s == G.dcap) s = 0;
    }
}

// Sixteen lanes per read, four reads per wave: the kernel is bound by instruction issue (a read is ~1000 wave-level
// instructions whatever the number of active lanes), so four reads share every instruction.  A group's control flow is
// uniform inside the group; the groups of a wave diverge like threads do.
#ifndef SNK_PATH_OCC
#define SNK_PATH_OCC 8           // workgroups per CU the register allocation aims at (5 until round 6: with four / eight lanes per read the passes want reads in flight, not registers: 57.6 -> 54.3 ms)
#endif
// GS lanes per read (16 or 8): the kernel is latency bound -- a clean read is a chain of ~6 dependent HBM round trips whatever the
// number of lanes that wait for them -- so eight lanes per read put twice as many reads in flight per wave (round 3: 134 -> see DESIGN);
// the full-capacity second pass keeps sixteen (its parts take 29 KB of LDS per sixteen reads).
// MODE 0 "fast": every read, but only as far as a clean read goes -- first k-mer found, exact-match run to the end of the read;
//        anything else (a miss, a run that ends early) marks the read (gm = 0xFE) and moves on.  Four reads share a wave and the
//        wave takes the time of its slowest: with one read in four carrying an error, 70 % of the waves paid an error read's
//        ~2000 instructions for all four (profiles/r02_pmc_path_instmix.csv); now the clean three quarters cost ~500 each wave
//        and the rest are pathed together, where every group of a wave has the same kind of work.
// MODE 1 "slow": the reads of a list (a.slow), the whole algorithm, small capacities.   MODE 2: the full-capacity pass over the
// redo list.   MODE 3: every read, the whole algorithm (SNK_PATH_TWO_PASS=0 / the fused variant).
// IDX: the look-ups go through the minimiser index (its own instantiation: the key array in LDS and the second look-up path cost the
// dictionary variant 9 % when they were a run-time switch)
template <int K, int PC, int PM, int MODE, int GS, bool IDX>
__global__ void __launch_bounds__(256, MODE == 2 ? 4 : SNK_PATH_OCC) path_kernel(path_args a) {
    constexpr bool SECOND = MODE == 2;
    constexpr int NG = 256 / GS;                      // reads per workgroup
    constexpr uint32_t GM = (1u << GS) - 1u;
    static_assert(GS == 16 || GS == 8 || GS == 4, "lanes per read");
    __shared__ uint32_t rowL[NG][20];
    // ordering keys of the read's 16-mers (minimiser index): the fast pass looks the first k-mer up and nothing else
    constexpr uint32_t KEYCAP = !IDX ? 1u : (MODE == 0 ? 64u : 20u * 16u - 15u);
    __shared__ uint32_t keysL[IDX ? NG : 1][KEYCAP];
    __shared__ ppart partsL[NG][PC];
    __shared__ int32_t pathL[NG][PM];
    __shared__ int32_t resL[NG][4];
    const int lane = threadIdx.x & 63, sub = lane & (GS - 1), gsh = lane & (64 - GS);       // gsh: first lane of my group
    const int gw = threadIdx.x / GS;                                           // group inside the workgroup
    const path_graph& G = a.G;
    uint32_t* row = rowL[gw];
    ppart* parts = partsL[gw];
    const uint64_t ng = (uint64_t)gridDim.x * NG;
    const uint64_t n_items = SECOND ? a.n_redo : (MODE == 1 ? a.n_slow : a.n_reads);
    for (uint64_t r0 = (uint64_t)blockIdx.x * NG; r0 < n_items; r0 += ng) {
        const bool live = r0 + gw < n_items;
        const uint64_t r = SECOND ? (live ? (uint64_t)a.redo[r0 + gw] : 0ull) : (MODE == 1 ? (live ? (uint64_t)a.slow[r0 + gw] : 0ull) : r0 + gw);
        uint32_t n = 0;
        if (live) {
            n = a.lens ? a.lens[r] : a.read_len;
            if (n > a.read_len) n = a.read_len;
            for (uint32_t w = (uint32_t)sub; w < 20u; w += GS) row[w] = w < a.row_words ? a.rows[r * a.row_words + w] : 0u;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (IDX) {
            if (live) mm_read_keys(row, n, KEYCAP, sub, GS, keysL[IDX ? gw : 0]);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
        // ---- Pather::path :705-748
        int m = 0;
        bool overflow = false, deferred = false;
        uint32_t resume_i = 0;
        if (MODE == 1 && live) {              // continue where the fast pass stopped
            const ppart res = a.gparts[r * GPARTS + 1];
            resume_i = res.unitig;
            m = (int)res.off_rc;
            if (m && sub == 0) parts[0] = a.gparts[r * GPARTS];
        }
        if (live && n < (uint32_t)K) { if (sub == 0) parts[0] = make_gap(n); m = 1; }
        else if (live) {
            const uint32_t end = n - K + 1;
            uint32_t i = resume_i, gap = 0;
            // after a miss the next look-ups go 16 positions at a time.  (The slow pass starts that way: where the fast pass stopped,
            // the k-mer that follows a mismatch ends on the mismatching base; 16 look-ups at once are the same 16 sequential ones.)
            bool wide = MODE == 1;
            bool exact = false;               // a fingerprint match failed the base check once: this read verifies inside dict_find
            while (i < end) {
                const uint32_t pos = i + sub;
                bool hit = false;
                uint32_t hu = 0, ho = 0, hrc = 0;
                uint64_t hpos = 0;
                if (pos < end && (wide || sub == 0)) hit = IDX ? mm_find<K>(G, row, keysL[IDX ? gw : 0], pos, &hrc, &hpos) : dict_find<K>(G, row, pos, &hu, &ho, &hrc, exact, &hpos);
                const uint32_t hm = (uint32_t)(__ballot(hit) >> gsh) & GM;
                if (!hm) {
                    if (MODE == 0) { deferred = true; resume_i = i; break; }          // the first k-mer is not on the graph: the slow pass takes the read
                    const uint32_t step = !wide ? 1u : (end - i < (uint32_t)GS ? end - i : (uint32_t)GS);
                    gap += step; i += step; wide = true;
                    continue;
                }
                const int first = __ffs((int)hm) - 1;
                const uint32_t rc = __shfl(hrc, gsh + first);
                const uint64_t hp = ((uint64_t)__shfl((uint32_t)(hpos >> 32), gsh + first) << 32) | __shfl((uint32_t)hpos, gsh + first);
                // The read from the seed on against the unitig (or its reverse complement), 2-bit words on both sides, the whole
                // window in ONE step: CPL bases per lane, 256 per group.  L = bases that agree from the seed's first base on; fewer
                // than K is a false fingerprint match (the seed check), the rest is matchLen's exact-match extension (:549-558).
                // The window is addressed by the hit's POSITION in the concatenated unitigs, so its loads go out before the unitig
                // is known (position -> block table -> a step or two along uoff -> uinfo: three dependent loads that now run beside
                // the window's); what lies behind the unitig's end is cut off afterwards.
                constexpr uint32_t CPL = 256 / GS;
                const uint32_t rs = i + (uint32_t)first;
                const uint32_t Rmax = n - rs;
                uint32_t w0[CPL / 16], w1[CPL / 16], wsh[CPL / 16];
#pragma unroll
                for (uint32_t c = 0; c < CPL / 16; ++c) {
                    const uint32_t j = CPL * (uint32_t)sub + 16u * c;
                    w0[c] = w1[c] = 0; wsh[c] = 0;
                    if (j < Rmax) {
                        const uint64_t q = !rc ? UPAD + hp + j : UPAD + hp + (uint64_t)(K - 1) - j - 15u;      // rc: the 16 bases that END at the mirrored position
                        w0[c] = G.upack[q >> 4]; w1[c] = G.upack[(q >> 4) + 1]; wsh[c] = 2u * ((uint32_t)q & 15u);
                    }
                }
                uint32_t u = G.ublk[hp >> 8];
                while (G.uoff[u + 1] <= hp) ++u;
                const uint4 ui = G.uinfo[2 * (uint64_t)u];
                const uint32_t sz = ui.z;
                const uint32_t o0 = (uint32_t)(hp - (((uint64_t)ui.y << 32) | ui.x));
                const uint32_t off = !rc ? o0 : sz - o0 - (uint32_t)K;           // :726-729
                const uint32_t Lmax = Rmax < (sz - off) ? Rmax : (sz - off);
                uint32_t mt = 0;
                bool stop = false;
#pragma unroll
                for (uint32_t c = 0; c < CPL / 16; ++c) {
                    const uint32_t j = CPL * (uint32_t)sub + 16u * c;
                    if (!stop) {
                        if (j >= Lmax) stop = true;
                        else {
                            const uint32_t vl = Lmax - j < 16u ? Lmax - j : 16u;
                            const uint32_t xr = packed16(row, rs + j);
                            uint32_t xu = (uint32_t)(((((uint64_t)w0[c] << 32) | w1[c]) << wsh[c]) >> 32);
                            if (rc) xu = snk_rev2_32(~xu);
                            const uint32_t d = xr ^ xu;
                            const uint32_t mm = d ? (uint32_t)__clz((int)d) >> 1 : 16u;
                            if (mm >= vl) { mt += vl; stop = vl < 16u; }
                            else { mt += mm; stop = true; }
                        }
                    }
                }
                const uint32_t smask = (uint32_t)(__ballot(stop) >> gsh) & GM;
                uint32_t L = 256u;
                if (smask) { const int fl = __ffs((int)smask) - 1; L = CPL * (uint32_t)fl + __shfl(mt, gsh + fl); }
                if (L < (uint32_t)K) { exact = true; continue; }       // (never with verified look-ups) redo this round with them
                wide = false;
                gap += (uint32_t)first;
                i += (uint32_t)first;
                const uint32_t len = L - (uint32_t)K + 1u;
                if (sub == 0) {
                    if (gap) { if (m < PC) parts[m] = make_gap(gap); }
                    const int at = m + (gap ? 1 : 0);
                    if (at < PC) { ppart p; p.unitig = u; p.off_rc = off | (rc << 31); p.len = len; p.elen = sz - K + 1; parts[at] = p; }
                }
                m += gap ? 2 : 1;
                if (m > PC) { overflow = true; break; }
                gap = 0;
                i += len;
                if (MODE == 0 && i < end) { deferred = true; resume_i = i; break; }  // the run ended inside the read
            }
            if (gap && !overflow) { if (m < PC) { if (sub == 0) parts[m] = make_gap(gap); ++m; } else overflow = true; }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (MODE == 0 && deferred) {
            // what was found so far travels with the read: the first part (if the first k-mer was on the graph) and where to go on
            if (sub == 0) {
                a.gm[r] = 0xFE;
                if (m) a.gparts[r * GPARTS] = parts[0];
                ppart res; res.unitig = resume_i; res.off_rc = (uint32_t)m; res.len = 0; res.elen = 0;
                a.gparts[r * GPARTS + 1] = res;
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // ---- the rest of algorithmTwo + the extension: sequential, the group's first lane
        if (live && sub == 0 && a.bc) {
            // (unitig, barcode) keys of this read -- from the parts as Pather::path left them (finish_path rewrites them)
            unsigned long long first = ~0ull;
            const int32_t b = a.bc[r];
            if (b > 0 && !overflow) {
                uint32_t fu = 0xFFFFFFFFu, lastu = 0xFFFFFFFFu;
                for (int p = 0; p < m; ++p) {
                    if (p_gap(parts[p])) continue;
                    const uint32_t u = parts[p].unitig;
                    if (fu == 0xFFFFFFFFu) { fu = u; first = ((unsigned long long)u << 32) | (uint32_t)b; }
                    else if (u != fu && u != lastu) {
                        const unsigned long long at = atomicAdd(&a.cursor[2], 1ull);
                        if (at < a.ub_cap) a.ub_more[at] = ((unsigned long long)u << 32) | (uint32_t)b;
                    }
                    lastu = u;
                }
            }
            a.ub_first[r] = first;
        }
        if (!SECOND && a.gparts) {
            // split first pass: hand the parts over (or the read to the redo list) and go on to the next reads
            if (live && sub == 0) {
                if (a.force_redo || overflow || m > GPARTS) {
                    const unsigned long long at = atomicAdd(&a.cursor[3], 1ull);
                    if (at < a.redo_cap) a.redo[at] = (uint32_t)r;
                    a.gm[r] = 0xFF;
                    a.out_off[r] = 0; a.out_n[r] = 0; a.out_e0[r] = -1; a.out_start[r] = 0;
                } else {
                    for (int p = 0; p < m; ++p) a.gparts[r * GPARTS + p] = parts[p];
                    a.gm[r] = (uint8_t)m;
                }
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (live && sub == 0) {
            int np = 0;
            int32_t off = 0;
            if (!SECOND && a.force_redo) overflow = true;
            if (!overflow) finish_path<K>(G, parts, m, row, a.quals + r * a.qstride, n, pathL[gw], PM, &np, &off);
            if (np < 0) { overflow = true; np = 0; off = 0; }
            if (!SECOND && overflow) {          // not an error yet: the full-capacity pass takes this read
                const unsigned long long at = atomicAdd(&a.cursor[3], 1ull);
                if (at < a.redo_cap) a.redo[at] = (uint32_t)r;
                overflow = false;
            }
            // a path's first edge travels with the read; only the rest (one read in a thousand has more than one edge on a deep
            // data set) takes a reservation on the shared cursor -- 1e8 atomics on one address were a second of serialisation
            unsigned long long st = 0;
            if (np > 1) st = atomicAdd(&a.cursor[0], (unsigned long long)(np - 1));
            if (overflow || (np > 1 && st + (unsigned long long)(np - 1) > a.scratch_cap)) { atomicOr(&a.cursor[1], overflow ? 1ull : 2ull); np = 0; }
            a.out_off[r] = off;
            a.out_n[r] = (uint32_t)np;
            a.out_e0[r] = np ? pathL[gw][0] : -1;
            a.out_start[r] = st;
            resL[gw][0] = np;
            resL[gw][1] = (int32_t)(st & 0xFFFFFFFFu);
            resL[gw][2] = (int32_t)(st >> 32);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (live) {
            const int np = resL[gw][0];
            const unsigned long long st = ((unsigned long long)(uint32_t)resL[gw][2] << 32) | (uint32_t)resL[gw][1];
            for (int q = 1 + sub; q < np; q += GS) a.scratch[st + q - 1] = pathL[gw][q];
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void __launch_bounds__(256) slow_flag_kernel(const uint8_t* __restrict__ gm, uint64_t n, uint64_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n) flag[i] = (i < n && gm[i] == 0xFE) ? 1ull : 0ull;
}
__global__ void __launch_bounds__(256) slow_fill_kernel(const uint8_t* __restrict__ gm, const uint64_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ list) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && gm[i] == 0xFE) list[pos[i]] = (uint32_t)i;
}
// the sequential rest of a read's pathing (algorithmTwo after Pather::path, the extension), one thread per read
template <int K>
__global__ void __launch_bounds__(256) path_finish_kernel(path_args a) {
    const path_graph& G = a.G;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < a.n_reads; r += stride) {
        const int m = a.gm[r];
        if (m == 0xFF) continue;
        uint32_t n = a.lens ? a.lens[r] : a.read_len;
        if (n > a.read_len) n = a.read_len;
        uint32_t row[20];
#pragma unroll
        for (uint32_t q = 0; q < 20; ++q) row[q] = q < a.row_words ? a.rows[r * a.row_words + q] : 0u;
        ppart parts[GPARTS];
#pragma unroll
        for (int p = 0; p < GPARTS; ++p) if (p < m) parts[p] = a.gparts[r * GPARTS + p];
        int32_t path[PMT];
        int np = 0;
        int32_t off = 0;
        finish_path<K>(G, parts, m, row, a.quals + r * a.qstride, n, path, PMT, &np, &off);
        if (np < 0) {                               // more edges than fit here: the full-capacity pass takes the read
            const unsigned long long at = atomicAdd(&a.cursor[3], 1ull);
            if (at < a.redo_cap) a.redo[at] = (uint32_t)r;
            np = 0; off = 0;
        }
        unsigned long long st = 0;
        if (np > 1) st = atomicAdd(&a.cursor[0], (unsigned long long)(np - 1));
        if (np > 1 && st + (unsigned long long)(np - 1) > a.scratch_cap) { atomicOr(&a.cursor[1], 2ull); np = 0; }
        a.out_off[r] = off;
        a.out_n[r] = (uint32_t)np;
        a.out_e0[r] = np ? path[0] : -1;
        a.out_start[r] = st;
        for (int q = 1; q < np; ++q) a.scratch[st + q - 1] = path[q];
    }
}

__global__ void __launch_bounds__(256) path_gather_kernel(const uint32_t* __restrict__ n, const int32_t* __restrict__ e0, const unsigned long long* __restrict__ start,
                                                          const uint64_t* __restrict__ pos, const int32_t* __restrict__ scratch, uint64_t n_reads,
                                                          int32_t* __restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t c = n[r];
    if (c) out[pos[r]] = e0[r];
    for (uint32_t q = 1; q < c; ++q) out[pos[r] + q] = scratch[start[r] + q - 1];
}
__global__ void __launch_bounds__(256) widen_kernel(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n) out[i] = i < n ? in[i] : 0ull;
}

template <typename T>
int dev(snk_ctx* ctx, size_t n, T** out, char* err, size_t errcap) {
    void* q = nullptr;
    int rc = snk_ctx_alloc(ctx, (n ? n : 1) * sizeof(T) + 16, &q, err, errcap);
    *out = (T*)q;
    return rc;
}
template <typename T>
int up(snk_ctx* ctx, hipStream_t st, const std::vector<T>& h, const T** out, char* err, size_t errcap) {
    T* d;
    int rc = dev(ctx, h.size(), &d, err, errcap);
    if (rc) return rc;
    if (!h.empty()) SNK_HIP_TRY(hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
    *out = d;
    return SNK_OK;
}
int scan64(snk_ctx* ctx, hipStream_t st, const uint64_t* in, uint64_t* out, size_t count, char* err, size_t errcap) {
    size_t tb = 0;
    SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, in, out, (uint64_t)0, count, rocprim::plus<uint64_t>(), st));
    void* tmp;
    int rc = snk_ctx_alloc(ctx, tb + 16, &tmp, err, errcap);
    if (rc) return rc;
    SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, in, out, (uint64_t)0, count, rocprim::plus<uint64_t>(), st));
    return SNK_OK;
}

// adjacency in AddEdge order: a vertex's edges ascending by the vertex at their other end, equal ones in edge-id order
void adjacency(int32_t N, int32_t E, const int32_t* key_v, const int32_t* other_v, std::vector<int32_t>& off, std::vector<int32_t>& vv, std::vector<int32_t>& ee) {
    off.assign((size_t)N + 2, 0);
    for (int32_t e = 0; e < E; ++e) off[key_v[e] + 1]++;
    for (int32_t v = 0; v < N; ++v) off[v + 1] += off[v];
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    std::vector<std::pair<int32_t, int32_t>> tmp((size_t)E);
    for (int32_t e = 0; e < E; ++e) tmp[cur[key_v[e]]++] = {other_v[e], e};
    for (int32_t v = 0; v < N; ++v) std::sort(tmp.begin() + off[v], tmp.begin() + off[v + 1]);
    vv.resize((size_t)E);
    ee.resize((size_t)E);
    for (int32_t i = 0; i < E; ++i) { vv[i] = tmp[i].first; ee[i] = tmp[i].second; }
}

// sorted (unitig << 32 | barcode) keys -> the first of every run of equal keys (the all-ones filler excluded)
__global__ void __launch_bounds__(256) ubc_flag_kernel(const unsigned long long* __restrict__ k, uint64_t n, uint64_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (k[i] != ~0ull && (i == 0 || k[i] != k[i - 1])) ? 1ull : 0ull;
    else if (i == n) flag[i] = 0;
}
__global__ void __launch_bounds__(256) ubc_scatter_kernel(const unsigned long long* __restrict__ k, const uint64_t* __restrict__ flag, const uint64_t* __restrict__ pos,
                                                          uint64_t n, uint32_t* __restrict__ bcs, uint64_t* __restrict__ per_unitig) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n && flag[i];
    unsigned long long key = 0;
    if (on) { key = k[i]; bcs[pos[i]] = (uint32_t)key; }
    // the keys are sorted: a wave's keys belong to one unitig or a few -- one addition per unitig and wave (lane by lane, 64 additions
    // to ONE counter were served one at a time)
    const uint32_t u = (uint32_t)(key >> 32);
    unsigned long long act = __ballot(on);
    const uint32_t lane = threadIdx.x & 63u;
    while (act) {
        const int lead = __ffsll((long long)act) - 1;
        const uint32_t u0 = (uint32_t)__shfl((int)u, lead);
        const unsigned long long same = __ballot(on && u == u0);
        if ((int)lane == lead) atomicAdd((unsigned long long*)&per_unitig[u0], (unsigned long long)__popcll(same));
        act &= ~same;
    }
}

// ---- second, independent derivation of the (unitig, barcode) keys (SNK_PATH_UNITIG_BCS_EXHAUSTIVE; a cross-check, ~50x the
// work): EVERY k-mer of every barcoded read is looked up (verified look-ups), which is literally what tada's per-shard code does
// before MAIN_ASM_SN unions the lists (barcodes_for_sedge, lib/tada/src/debruijn.rs:115-131).  One thread per read; a key is
// emitted whenever the unitig differs from the read's previous hit.
template <int K>
__global__ void __launch_bounds__(256) ubc_exhaustive_kernel(path_graph G, const uint32_t* __restrict__ rows, uint32_t row_words, uint32_t read_len,
                                                             const uint16_t* __restrict__ lens, const int32_t* __restrict__ bc, uint64_t n,
                                                             unsigned long long* __restrict__ keys, uint64_t cap, unsigned long long* __restrict__ cursor) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int32_t b = bc[r];
    if (b <= 0) return;
    uint32_t row[20];
    for (uint32_t q = 0; q < 20; ++q) row[q] = q < row_words ? rows[r * row_words + q] : 0u;
    uint32_t len = lens ? lens[r] : read_len;
    if (len > read_len) len = read_len;
    if (len < (uint32_t)K) return;
    uint32_t last = 0xFFFFFFFFu;
    for (uint32_t pos = 0; pos + K <= len; ++pos) {
        uint32_t hu, ho, hrc;
        if (!dict_find<K>(G, row, pos, &hu, &ho, &hrc, true)) continue;
        if (hu == last) continue;
        last = hu;
        const unsigned long long at = atomicAdd(cursor, 1ull);
        if (at < cap) keys[at] = ((unsigned long long)hu << 32) | (uint32_t)b;
    }
}
// the reference stops adding barcodes to an edge once its list holds 20 000 entries (cmd_main_asm.rs:115): lengths under the cut
__global__ void __launch_bounds__(256) ubc_cut_len_kernel(const uint64_t* __restrict__ off, uint64_t U, uint64_t cut, uint64_t* __restrict__ len_out, uint32_t* __restrict__ any) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u > U) return;
    if (u == U) { len_out[u] = 0; return; }
    const uint64_t l = off[u + 1] - off[u];
    len_out[u] = l < cut ? l : cut;
    if (l > cut) atomicOr(any, 1u);
}
__global__ void __launch_bounds__(64) ubc_cut_copy_kernel(const uint64_t* __restrict__ off, const uint64_t* __restrict__ noff, uint64_t U, const uint32_t* __restrict__ in,
                                                          uint32_t* __restrict__ out) {
    const uint64_t u = blockIdx.x;
    if (u >= U) return;
    const uint64_t n = noff[u + 1] - noff[u];
    for (uint64_t i = threadIdx.x; i < n; i += 64) out[noff[u] + i] = in[off[u] + i];
}

template <int K>
int path_impl(snk_ctx* ctx, hipStream_t st, const snk_dev_reads* in, uint64_t U, const uint64_t* d_uoff, const uint8_t* d_ubases, const snk_hbv* h,
              uint32_t flags, snk_dev_paths* out, char* err, size_t errcap) {
    int rc;
    const uint64_t n = in->n_reads;
    hipEvent_t e0, e1, e2, e3, e2b;
    SNK_HIP_TRY(hipEventCreate(&e0)); SNK_HIP_TRY(hipEventCreate(&e1)); SNK_HIP_TRY(hipEventCreate(&e2)); SNK_HIP_TRY(hipEventCreate(&e3)); SNK_HIP_TRY(hipEventCreate(&e2b));
    struct evg { hipEvent_t a, b, c, d, e; ~evg() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); (void)hipEventDestroy(d); (void)hipEventDestroy(e); } } g{e0, e1, e2, e3, e2b};
    SNK_HIP_TRY(hipEventRecord(e0, st));
    // ---- graph tables (host: O(U + E)), in the device's unitig numbering
    const int32_t N = h->n_vertices, E = h->n_edges;
    std::vector<int32_t> fwd(U), rev(U), eu((size_t)E), off_to, v_to, e_to, off_from, v_from, e_from;
    std::vector<uint8_t> erc((size_t)E);
    for (uint64_t r = 0; r < U; ++r) {
        const uint64_t d = h->bvcomp_order ? (uint64_t)h->bvcomp_order[r] : r;
        fwd[d] = h->fwd_xlat[r];
        rev[d] = h->rev_xlat[r];
    }
    for (int32_t e = 0; e < E; ++e) {
        const uint64_t r = (uint64_t)h->src_unitig[e];
        eu[e] = (int32_t)(h->bvcomp_order ? h->bvcomp_order[r] : (int32_t)r);
        erc[e] = h->is_rc[e];
    }
    adjacency(N, E, h->v_right, h->v_left, off_to, v_to, e_to);
    adjacency(N, E, h->v_left, h->v_right, off_from, v_from, e_from);
    std::vector<int32_t> vl(h->v_left, h->v_left + E), vr(h->v_right, h->v_right + E);
    path_args a;
    memset(&a, 0, sizeof a);
    path_graph& G = a.G;
    G.uoff = d_uoff; G.ubases = d_ubases;
    if ((rc = up(ctx, st, fwd, &G.fwd, err, errcap)) || (rc = up(ctx, st, rev, &G.rev, err, errcap)) || (rc = up(ctx, st, vl, &G.vleft, err, errcap)) ||
        (rc = up(ctx, st, vr, &G.vright, err, errcap)) || (rc = up(ctx, st, eu, &G.e_unitig, err, errcap)) || (rc = up(ctx, st, erc, &G.e_rc, err, errcap)) ||
        (rc = up(ctx, st, off_to, &G.to_off, err, errcap)) || (rc = up(ctx, st, v_to, &G.to_v, err, errcap)) || (rc = up(ctx, st, e_to, &G.to_e, err, errcap)) ||
        (rc = up(ctx, st, off_from, &G.from_off, err, errcap)) || (rc = up(ctx, st, v_from, &G.from_v, err, errcap)) || (rc = up(ctx, st, e_from, &G.from_e, err, errcap)))
        return rc;
    // ---- per-unitig records + dictionary
    uint64_t h_off_last = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_off_last, d_uoff + U, 8, hipMemcpyDeviceToHost, st));
    // seeds on a short hanging edge are dropped (BuildReadQGraph48.cc:1240-1247): a property of the edge, decided here once
    std::vector<uint8_t> drop((size_t)E + 1, 0);
    SNK_HIP_TRY(snk_sync(st));
    {
        std::vector<uint64_t> h_uoff(U + 1);
        if (U) SNK_HIP_TRY(hipMemcpy(h_uoff.data(), d_uoff, (U + 1) * 8, hipMemcpyDeviceToHost));
        for (int32_t e = 0; e < E; ++e) {
            const int32_t vl_ = vl[e], vr_ = vr[e];
            const uint64_t kmers = h_uoff[(size_t)eu[e] + 1] - h_uoff[(size_t)eu[e]] - (K - 1);
            drop[e] = (off_to[vl_ + 1] - off_to[vl_] == 0 && off_to[vr_ + 1] - off_to[vr_] > 1 && off_from[vr_ + 1] - off_from[vr_] > 0 && kmers <= 100) ? 1 : 0;
        }
    }
    const uint8_t* d_drop;
    if ((rc = up(ctx, st, drop, &d_drop, err, errcap))) return rc;
    uint4* uinfo;
    if ((rc = dev(ctx, 2 * U + 2, &uinfo, err, errcap))) return rc;
    if (U) hipLaunchKernelGGL(uinfo_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, d_uoff, U, G.fwd, G.rev, d_drop, uinfo);
    G.uinfo = uinfo;
    const uint64_t total_bases = h_off_last;
    {
        // the unitigs once more, 2 bits per base (a quarter of the byte array), UPAD bases of slack at either end
        const uint64_t n_words = (total_bases + 15) / 16 + 1;
        uint32_t* upack;
        if ((rc = dev(ctx, n_words + 2 * (UPAD / 16) + 2, &upack, err, errcap))) return rc;
        SNK_HIP_TRY(hipMemsetAsync(upack, 0, (UPAD / 16) * 4, st));
        SNK_HIP_TRY(hipMemsetAsync(upack + UPAD / 16 + n_words, 0, (UPAD / 16 + 2) * 4, st));
        hipLaunchKernelGGL(upack_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_ubases, total_bases, upack + UPAD / 16, n_words);
        G.upack = upack;
    }
    const uint64_t nk = total_bases >= U * (uint64_t)(K - 1) ? total_bases - U * (uint64_t)(K - 1) : 0;
    if (total_bases >= (1ull << 33) - 1) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_path_reads: more than 2^33 unitig bases (the dictionary keeps 33-bit positions)");
    // Look-ups: the k-mer dictionary (a slot per unitig k-mer: 24 bytes each, one probe or two per look-up) or the minimiser index
    // (8 bytes per ~17 unitig k-mers, every look-up verified against the packed unitigs: measured 2.6 x slower at bench size --
    // 182 against 71 ms per 100 M reads, the chain of dependent loads per look-up is three long instead of one or two -- but 0.14 GB
    // where the dictionary takes 6.4 GB).  The dictionary while it fits, the index when it does not (a human-size graph's dictionary
    // is ~85 GB next to the reads: rounds 1-3 refused such a graph); SNK_PATH_INDEX=1 / 0 forces one or the other.  The exhaustive
    // cross-check of the unitig barcode lists reads the dictionary.
    const uint64_t spk10 = snk_opt_u32("path_slots_x10", 30);         // slots per unitig k-mer x 10 (measured: 2.5 -> 67.2 ms pathing, 3 -> 63.5, 4 -> 61.8)
    uint64_t cap = ((nk * spk10 / 10 + 1024) + 63) & ~63ull;              // load 1/3: the chain of dependent probes is what a read waits for
    bool dict_fits = true;
    uint64_t free_b = 0;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            uint64_t cached = 0;
            for (auto& b : ctx->blocks) if (!b.used) cached += b.bytes;     // the arena's idle blocks can be handed back to the driver
            for (auto& f : ctx->va_free) cached += f.bytes;                 // (growing arena: mapped and free)
            free_b = (uint64_t)fr + cached;
            dict_fits = cap * 8ull <= free_b;
        }
        const uint64_t max_mb = snk_opt_u32("path_dict_max_kb", 0);     // (tests: a dictionary above this size "does not fit")
        if (max_mb && cap * 8ull > (max_mb << 10)) dict_fits = false;
    }
    const bool use_index = snk_opt_is_set("path_index") ? snk_opt_u32("path_index", 0) == 1u : !dict_fits;
    const bool need_kdict = !use_index || (flags & SNK_PATH_UNITIG_BCS_EXHAUSTIVE);
    unsigned long long* dslot = nullptr;
    unsigned long long fp_mask = 0x3FFFFFFFull;
    if (need_kdict) {
    if (!dict_fits && !snk_opt_u32("path_dict_max_kb", 0))
        return snk_fail(SNK_E_NOMEM, err, errcap, "snk_dev_path_reads: the k-mer dictionary of this graph (%llu unitig k-mers, 3 slots of 8 B each) needs %.1f GB, "
                        "%.1f GB of HBM are free; let the minimiser index take the look-ups (option path_index unset or 1)", (unsigned long long)nk, cap * 8ull / 1e9, free_b / 1e9);
    if (snk_opt_is_set("path_fp_mask")) { const unsigned long long m = snk_opt_u64("path_fp_mask", 0) & 0x3FFFFFFFull; if (m) fp_mask = m; }
    if ((rc = dev(ctx, cap, &dslot, err, errcap))) return rc;
    SNK_HIP_TRY(hipMemsetAsync(dslot, 0xFF, cap * 8, st));
    if (total_bases) hipLaunchKernelGGL((dict_build_kernel<K>), dim3((unsigned)(((total_bases + DB_RUN - 1) / DB_RUN + 255) / 256)), dim3(256), 0, st, d_uoff, d_ubases, U, total_bases, dslot, cap, fp_mask);
    }
    uint64_t n_ment = 0;
    if (use_index) {
        // places per workgroup, their offsets, the places, sorted by key, the directory over the key's top bits
        const uint64_t n_wg = total_bases ? ((total_bases + MM_RUN - 1) / MM_RUN + 255) / 256 : 0;
        uint32_t *wgc, *okey, *okey2, *mdir, *mhist;
        uint64_t *wg64, *wgo;
        unsigned long long *oval, *oval2;
        if ((rc = dev(ctx, n_wg + 2, &wgc, err, errcap)) || (rc = dev(ctx, n_wg + 2, &wg64, err, errcap)) || (rc = dev(ctx, n_wg + 2, &wgo, err, errcap))) return rc;
        if (n_wg) {
            hipLaunchKernelGGL((mm_scan_kernel<K, false>), dim3((unsigned)n_wg), dim3(256), 0, st, d_uoff, G.upack, U, total_bases, wgc, (const uint64_t*)nullptr, (uint32_t*)nullptr,
                               (unsigned long long*)nullptr);
            hipLaunchKernelGGL(mm_widen_kernel, dim3((unsigned)((n_wg + 256) / 256)), dim3(256), 0, st, wgc, n_wg, wg64);
            if ((rc = scan64(ctx, st, wg64, wgo, n_wg + 1, err, errcap))) return rc;
            SNK_HIP_TRY(hipMemcpyAsync(&n_ment, wgo + n_wg, 8, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
        }
        if (n_ment >= (1ull << 32) - 2) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_path_reads: more than 2^32 minimiser places");
        if ((rc = dev(ctx, n_ment + 1, &okey, err, errcap)) || (rc = dev(ctx, n_ment + 1, &okey2, err, errcap)) || (rc = dev(ctx, n_ment + 1, &oval, err, errcap)) ||
            (rc = dev(ctx, n_ment + 1, &oval2, err, errcap)))
            return rc;
        uint32_t bits = 8;
        while (bits < 26 && (2ull << bits) < n_ment) ++bits;             // ~2-4 places per directory entry
        if ((rc = dev(ctx, (1ull << bits) + 2, &mdir, err, errcap)) || (rc = dev(ctx, (1ull << bits) + 2, &mhist, err, errcap))) return rc;
        SNK_HIP_TRY(hipMemsetAsync(mdir, 0, ((1ull << bits) + 2) * 4, st));
        SNK_HIP_TRY(hipMemsetAsync(mhist, 0, ((1ull << bits) + 2) * 4, st));
        if (n_ment) {
            hipLaunchKernelGGL((mm_scan_kernel<K, true>), dim3((unsigned)n_wg), dim3(256), 0, st, d_uoff, G.upack, U, total_bases, wgc, (const uint64_t*)wgo, okey, oval);
            size_t tb = 0;
            SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, okey, okey2, oval, oval2, (size_t)n_ment, 0u, 32u, st));
            void* tmp;
            if ((rc = snk_ctx_alloc(ctx, tb + 64, &tmp, err, errcap))) return rc;
            SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, okey, okey2, oval, oval2, (size_t)n_ment, 0u, 32u, st));
            hipLaunchKernelGGL(mm_hist_kernel, dim3((unsigned)((n_ment + 255) / 256)), dim3(256), 0, st, okey2, n_ment, 32u - bits, mhist);
            size_t tb2 = 0;
            SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb2, mhist, mdir, 0u, ((size_t)1 << bits) + 1, rocprim::plus<uint32_t>(), st));
            void* tmp2;
            if ((rc = snk_ctx_alloc(ctx, tb2 + 64, &tmp2, err, errcap))) return rc;
            SNK_HIP_TRY(rocprim::exclusive_scan(tmp2, tb2, mhist, mdir, 0u, ((size_t)1 << bits) + 1, rocprim::plus<uint32_t>(), st));
        }
        G.ment = oval2; G.mdir = mdir; G.mdir_bits = bits; G.idx_dbg = snk_opt_u32("path_idx_dbg", 0);
        if (!need_kdict) cap = n_ment;
        out->lookup_index = 1;
    }
    {
        const uint64_t n_blk = (total_bases >> 8) + 2;
        uint32_t* ublk;
        if ((rc = dev(ctx, n_blk, &ublk, err, errcap))) return rc;
        if (U) hipLaunchKernelGGL(ublk_kernel, dim3((unsigned)((n_blk + 255) / 256)), dim3(256), 0, st, d_uoff, U, n_blk, ublk);
        else SNK_HIP_TRY(hipMemsetAsync(ublk, 0, n_blk * 4, st));
        G.ublk = ublk;
    }
    SNK_HIP_TRY(hipGetLastError());
    SNK_HIP_TRY(snk_sync(st));            // drop[] has been copied
    G.dslot = dslot; G.dcap = cap; G.fp_mask = fp_mask;
    SNK_HIP_TRY(hipEventRecord(e1, st));
    // ---- the reads
    a.rows = (const uint32_t*)in->rows; a.row_words = in->row_words; a.read_len = in->read_len;
    a.quals = (const uint8_t*)in->quals; a.qstride = in->qstride; a.lens = (const uint16_t*)in->lens; a.n_reads = n;
    uint32_t* out_n;
    unsigned long long *out_start, *cursor;
    int32_t *out_off, *out_e0, *scratch = nullptr, *edges = nullptr;
    uint64_t *n64, *pos;
    if ((rc = dev(ctx, n + 1, &out_n, err, errcap)) || (rc = dev(ctx, n + 1, &out_start, err, errcap)) || (rc = dev(ctx, n + 1, &out_off, err, errcap)) || (rc = dev(ctx, n + 1, &out_e0, err, errcap)) ||
        (rc = dev(ctx, 4, &cursor, err, errcap)) || (rc = dev(ctx, n + 2, &n64, err, errcap)) || (rc = dev(ctx, n + 2, &pos, err, errcap)))
        return rc;
    uint64_t scap = n / 4 + 65536;                    // second and later edges only; a wrong guess costs one re-run
    const bool want_bcs = (flags & SNK_PATH_UNITIG_BCS) && in->bc;
    uint64_t ubcap = want_bcs ? n / 4 + 65536 : 0;    // (unitig, barcode) keys beyond a read's first
    unsigned long long* ubk = nullptr;                // [n + ubcap]: the reads' first keys, then the further ones
    unsigned long long h_cur[4] = {0, 0, 0, 0};
    ppart* gparts = nullptr;
    uint8_t* gm = nullptr;
    if (!snk_opt_u32("path_fused", 0)) {          // default: split first pass (SNK_PATH_FUSED=1: the group kernel does it all)
        if ((rc = dev(ctx, n * GPARTS + 1, &gparts, err, errcap)) || (rc = dev(ctx, n + 1, &gm, err, errcap))) return rc;
    }
    uint32_t* redo = nullptr;
    uint64_t *slow_flag = nullptr, *slow_pos = nullptr;
    uint64_t rcap = n / 64 + 65536;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (!scratch && (rc = dev(ctx, scap, &scratch, err, errcap))) return rc;
        if (want_bcs && !ubk && (rc = dev(ctx, n + ubcap, &ubk, err, errcap))) return rc;
        SNK_HIP_TRY(hipMemsetAsync(cursor, 0, 32, st));
        a.out_off = out_off; a.out_n = out_n; a.out_e0 = out_e0; a.out_start = out_start; a.scratch = scratch; a.cursor = cursor; a.scratch_cap = scap;
        a.bc = want_bcs ? (const int32_t*)in->bc : nullptr; a.ub_first = ubk; a.ub_more = ubk ? ubk + n : nullptr; a.ub_cap = ubcap;
        if (!redo && (rc = dev(ctx, rcap, &redo, err, errcap))) return rc;
        a.redo = redo; a.redo_cap = rcap; a.n_redo = 0; a.force_redo = snk_opt_u32("path_redo_all", 0);
        a.gparts = gparts; a.gm = gm;
        if (n) {
            // sixteen lanes per read (eight put twice the reads in flight but the kernel is issue bound: 145.6 ms against 134.5)
            uint64_t grid = (n + 15) / 16;
            const uint64_t gmax = (uint64_t)ctx->n_cu * 64;
            if (grid > gmax) grid = gmax;
            if (gparts && snk_opt_u32("path_two_pass", 1)) {
#ifndef SNK_PATH_FAST_GS
#define SNK_PATH_FAST_GS 4          // lanes per read of the fast pass (the template's 8 of round 3 -> 4: sixteen reads share a wave's instructions; 62.2 -> 57.7 ms with the slow pass at 8)
#endif
                if (snk_opt_u32("path_fast_gs", 8) == 8) {
                    uint64_t g0 = (n + 256 / SNK_PATH_FAST_GS - 1) / (256 / SNK_PATH_FAST_GS);
                    if (g0 > gmax) g0 = gmax;
                    if (use_index) hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 0, 8, true>), dim3((unsigned)g0), dim3(256), 0, st, a);
#ifndef SNK_PATH_FAST_PC
#define SNK_PATH_FAST_PC 4           // parts per read the fast pass has room for (it writes one: 4 KB of LDS per workgroup instead of 20)
#endif
                    else hipLaunchKernelGGL((path_kernel<K, SNK_PATH_FAST_PC, PMAX1, 0, SNK_PATH_FAST_GS, false>), dim3((unsigned)g0), dim3(256), 0, st, a);
                } else if (use_index) hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 0, 16, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 0, 16, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
                // the reads it left, in read order
                if (!slow_flag && ((rc = dev(ctx, n + 2, &slow_flag, err, errcap)) || (rc = dev(ctx, n + 2, &slow_pos, err, errcap)))) return rc;
                hipLaunchKernelGGL(slow_flag_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, gm, n, slow_flag);
                if ((rc = scan64(ctx, st, slow_flag, slow_pos, n + 1, err, errcap))) return rc;
                uint64_t n_slow = 0;
                SNK_HIP_TRY(hipMemcpyAsync(&n_slow, slow_pos + n, 8, hipMemcpyDeviceToHost, st));
                SNK_HIP_TRY(snk_sync(st));
                if (n_slow) {
                    uint32_t* slow;
                    if ((rc = dev(ctx, n_slow + 1, &slow, err, errcap))) return rc;
                    hipLaunchKernelGGL(slow_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gm, slow_pos, n, slow);
                    a.slow = slow; a.n_slow = n_slow;
#ifndef SNK_PATH_SLOW_GS
#define SNK_PATH_SLOW_GS 8           // lanes per read of the slow pass: the pass is bound by its instruction stream (profiles/r06_path_wide_rounds.log), eight reads share a wave's
#endif                               //   instructions instead of four: 63.3 -> 60.3 ms per 100 M reads (four alternating runs each, same box)
                    uint64_t g1 = (n_slow + 256 / SNK_PATH_SLOW_GS - 1) / (256 / SNK_PATH_SLOW_GS);
                    if (g1 > gmax) g1 = gmax;
                    if (use_index) hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 1, 16, true>), dim3((unsigned)g1), dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 1, SNK_PATH_SLOW_GS, false>), dim3((unsigned)g1), dim3(256), 0, st, a);
                }
                out->n_slow = n_slow;
            } else if (use_index) hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 3, 16, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((path_kernel<K, PCAP1, PMAX1, 3, 16, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
            if (gparts) {
                uint64_t g2 = (n + 255) / 256;
                if (g2 > gmax) g2 = gmax;
                hipLaunchKernelGGL((path_finish_kernel<K>), dim3((unsigned)g2), dim3(256), 0, st, a);
            }
        }
        SNK_HIP_TRY(hipGetLastError());
        SNK_HIP_TRY(hipMemcpyAsync(h_cur, cursor, 32, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        if (h_cur[3] > rcap) {                       // more reads to redo than the list holds: once more with a longer list
            if (attempt == 2) return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_dev_path_reads: redo list overflow");
            snk_ctx_release_block(ctx, redo); redo = nullptr; rcap = h_cur[3] + 1024;
            continue;
        }
        if (h_cur[3]) {                              // the reads with many parts / edges, at full capacity
            a.n_redo = h_cur[3];
            uint64_t grid = (a.n_redo + 15) / 16;
            const uint64_t gmax = (uint64_t)ctx->n_cu * 64;
            if (grid > gmax) grid = gmax;
            if (use_index) hipLaunchKernelGGL((path_kernel<K, PCAP, PMAX, 2, 16, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((path_kernel<K, PCAP, PMAX, 2, 16, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
            SNK_HIP_TRY(hipGetLastError());
            SNK_HIP_TRY(hipMemcpyAsync(h_cur, cursor, 32, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
        }
        if (h_cur[1] & 1ull) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_path_reads: a read has more than %d path parts or %d edges", PCAP, PMAX);
        const bool e_over = (h_cur[1] & 2ull) != 0, b_over = want_bcs && h_cur[2] > ubcap;
        if (!e_over && !b_over) break;
        if (attempt == 2) return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_dev_path_reads: path scratch overflow");
        if (e_over) { snk_ctx_release_block(ctx, scratch); scratch = nullptr; scap = h_cur[0] + 1024; }
        if (b_over) { snk_ctx_release_block(ctx, ubk); ubk = nullptr; ubcap = h_cur[2] + 1024; }
    }
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, out_n, n, n64);
    if ((rc = scan64(ctx, st, n64, pos, n + 1, err, errcap))) return rc;
    uint64_t total = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&total, pos + n, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if ((rc = dev(ctx, total + 1, &edges, err, errcap))) return rc;
    if (n) hipLaunchKernelGGL(path_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_n, out_e0, out_start, pos, scratch, n, edges);
    SNK_HIP_TRY(hipGetLastError());
    SNK_HIP_TRY(hipEventRecord(e2, st));
    SNK_HIP_TRY(snk_sync(st));
    snk_ctx_release_block(ctx, scratch);
    out->n_reads = n;
    out->n_edges_total = total;
    out->offset = out_off;
    out->n_edges = out_n;
    out->start = pos;
    out->edges = edges;
    out->dict_slots = cap;
    (void)hipEventElapsedTime(&out->dict_ms, e0, e1);
    (void)hipEventElapsedTime(&out->path_ms, e1, e2);
    if (want_bcs) {
        // sort the keys, keep the first of every run, count per unitig: sorted distinct barcodes per unitig
        uint64_t nkeys = n + h_cur[2];
        if (flags & SNK_PATH_UNITIG_BCS_EXHAUSTIVE) {
            const uint64_t xcap = n * 16 + 1024;
            unsigned long long* xk;
            if ((rc = dev(ctx, xcap, &xk, err, errcap))) return rc;
            SNK_HIP_TRY(hipMemsetAsync(cursor, 0, 8, st));
            if (n) hipLaunchKernelGGL((ubc_exhaustive_kernel<K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, a.rows, a.row_words, a.read_len, a.lens,
                                      (const int32_t*)in->bc, n, xk, xcap, cursor);
            unsigned long long hx = 0;
            SNK_HIP_TRY(hipMemcpyAsync(&hx, cursor, 8, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
            if (hx > xcap) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_path_reads: the exhaustive barcode derivation found more than 16 unitigs per read on average");
            ubk = xk;
            nkeys = hx;
        }
        // Everything this stage allocates is allocated HERE, in front of its timed interval (seven arena allocations, one
        // of which could land on a hipMalloc, and a temp-size query sat inside it -- 76 ms on the driver's box against 7 ms here): the
        // barcode array gets its upper bound (one entry per key) instead of waiting for the distinct count.
        unsigned long long* ks;
        uint64_t *flag, *upos, *per_u, *uoff_out;
        uint32_t* bcs = nullptr;
        // the key is unitig << 32 | barcode: only the bits that can differ are sorted (the unitig field's all-ones pattern is above every
        // unitig id, so the unused first-key slots -- all ones -- still sort behind everything)
        unsigned ubits = 1;
        while (ubits < 32 && (1ull << ubits) < U + 2) ++ubits;
        const unsigned end_bit = 32 + ubits;
        size_t tb = 0;
        uint8_t* tmp = nullptr;
        if ((rc = dev(ctx, nkeys + 1, &ks, err, errcap)) || (rc = dev(ctx, nkeys + 2, &flag, err, errcap)) || (rc = dev(ctx, nkeys + 2, &upos, err, errcap)) ||
            (rc = dev(ctx, U + 2, &per_u, err, errcap)) || (rc = dev(ctx, U + 2, &uoff_out, err, errcap)) || (rc = dev(ctx, nkeys + 1, &bcs, err, errcap)))
            return rc;
        if (nkeys) {
            SNK_HIP_TRY(rocprim::radix_sort_keys((void*)nullptr, tb, ubk, ks, (size_t)nkeys, 0u, end_bit, st));
            if ((rc = dev(ctx, tb, &tmp, err, errcap))) return rc;
        }
        SNK_HIP_TRY(hipEventRecord(e2b, st));
        uint64_t n_unique = 0;
        if (nkeys) {
            SNK_HIP_TRY(rocprim::radix_sort_keys(tmp, tb, ubk, ks, (size_t)nkeys, 0u, end_bit, st));
            hipLaunchKernelGGL(ubc_flag_kernel, dim3((unsigned)((nkeys + 256) / 256)), dim3(256), 0, st, ks, nkeys, flag);
            if ((rc = scan64(ctx, st, flag, upos, nkeys + 1, err, errcap))) return rc;
            SNK_HIP_TRY(hipMemcpyAsync(&n_unique, upos + nkeys, 8, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
        }
        SNK_HIP_TRY(hipMemsetAsync(per_u, 0, (U + 2) * 8, st));
        if (nkeys) hipLaunchKernelGGL(ubc_scatter_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, ks, flag, upos, nkeys, bcs, per_u);
        if ((rc = scan64(ctx, st, per_u, uoff_out, U + 1, err, errcap))) return rc;
        SNK_HIP_TRY(hipGetLastError());
        SNK_HIP_TRY(snk_sync(st));
        if (!(flags & SNK_PATH_UNITIG_BCS_NOCUT) && U) {
            // the reference's cut (cmd_main_asm.rs:115 stops extending an edge's list at 20 000 entries, in the order its shards
            // are visited and with duplicates counted between compactions -- not reproducible without its shard layout); here:
            // a unitig keeps its 20 000 SMALLEST barcode ids.  Lists under the cut -- all but high-copy repeats -- are unaffected.
            const uint64_t cut = snk_opt_u32("unitig_bc_cut", 20000);
            uint64_t *clen, *noff2;
            uint32_t* any;
            if ((rc = dev(ctx, U + 2, &clen, err, errcap)) || (rc = dev(ctx, U + 2, &noff2, err, errcap)) || (rc = dev(ctx, 4, &any, err, errcap))) return rc;
            SNK_HIP_TRY(hipMemsetAsync(any, 0, 4, st));
            hipLaunchKernelGGL(ubc_cut_len_kernel, dim3((unsigned)((U + 256) / 256)), dim3(256), 0, st, uoff_out, U, cut, clen, any);
            uint32_t h_any = 0;
            SNK_HIP_TRY(hipMemcpyAsync(&h_any, any, 4, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
            if (h_any) {
                if ((rc = scan64(ctx, st, clen, noff2, U + 1, err, errcap))) return rc;
                uint64_t n_cut = 0;
                SNK_HIP_TRY(hipMemcpyAsync(&n_cut, noff2 + U, 8, hipMemcpyDeviceToHost, st));
                SNK_HIP_TRY(snk_sync(st));
                uint32_t* bcs2;
                if ((rc = dev(ctx, n_cut + 1, &bcs2, err, errcap))) return rc;
                hipLaunchKernelGGL(ubc_cut_copy_kernel, dim3((unsigned)U), dim3(64), 0, st, uoff_out, noff2, U, bcs, bcs2);
                SNK_HIP_TRY(hipGetLastError());
                uoff_out = noff2; bcs = bcs2; n_unique = n_cut;
            }
        }
        out->unitig_bc_off = uoff_out;
        out->unitig_bcs = bcs;
        out->n_unitig_bcs = n_unique;
        SNK_HIP_TRY(hipEventRecord(e3, st));
        SNK_HIP_TRY(snk_sync(st));
        (void)hipEventElapsedTime(&out->bcs_ms, e2b, e3);
    }
    return SNK_OK;
}

}  // namespace

extern "C" int snk_dev_path_reads(snk_ctx* ctx, uint32_t K, const snk_dev_reads* in, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases,
                                  const snk_hbv* h, snk_dev_paths* out, void* stream, char* err, size_t errcap) {
    return snk_dev_path_reads2(ctx, K, in, n_unitigs, d_unitig_off, d_unitig_bases, h, 0u, out, stream, err, errcap);
}

extern "C" int snk_dev_path_reads2(snk_ctx* ctx, uint32_t K, const snk_dev_reads* in, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases,
                                   const snk_hbv* h, uint32_t flags, snk_dev_paths* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !h || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_path_reads: NULL argument");
    if (K != 48 && K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
    if (in->n_reads && (!in->rows || !in->quals || in->read_len == 0 || in->read_len > 256 || in->row_words * 16 < in->read_len || in->row_words > 16))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_path_reads: need packed rows and quality rows, reads of at most 256 bases");
    if (n_unitigs && (!d_unitig_off || !d_unitig_bases)) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_path_reads: NULL unitig arrays");
    if (n_unitigs >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_path_reads: too many unitigs");
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    // the call's scratch (graph tables, dictionary, parts, sort buffers: ~0.3 KB per read + 40 B per unitig k-mer) goes back to the
    // arena when it returns; only the paths (and barcode lists) stay, until the context's next snk_dev_count_graph / snk_shard_step
    const uint64_t mark = ctx->alloc_serial;
    int rc;
    try {
        if (K == 48) rc = path_impl<48>(ctx, st, in, n_unitigs, (const uint64_t*)d_unitig_off, (const uint8_t*)d_unitig_bases, h, flags, out, err, errcap);
        else rc = path_impl<60>(ctx, st, in, n_unitigs, (const uint64_t*)d_unitig_off, (const uint8_t*)d_unitig_bases, h, flags, out, err, errcap);
    } catch (const std::bad_alloc&) { rc = snk_fail(SNK_E_NOMEM, err, errcap, "snk_dev_path_reads: host allocation failed"); }
    (void)hipStreamSynchronize(st);          // also on the error paths: nothing of this call is still running when its scratch is handed back
    const void* keep[6] = {out->offset, out->n_edges, out->start, out->edges, out->unitig_bc_off, out->unitig_bcs};
    snk_ctx_release_since(ctx, mark, keep, rc ? 0 : 6);
    if (rc) memset(out, 0, sizeof *out);
    return rc;
}
