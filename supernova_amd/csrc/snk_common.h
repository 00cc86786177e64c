// snk_common.h -- host+device primitives shared by every translation unit of libsnk.
//
// Bit-level conventions (reference lib/assembly/src/kmers/KMer.h:153-160,344-350
// and lib/tada/src/kmer/mod.rs:527-537):
//   base code A=0 C=1 G=2 T=3, complement = b ^ 3
//   packed read row : u32 words, base i in word i>>4 at bits [31-2*(i&15) .. 30-2*(i&15)]  (MSB first)
//   k-mer           : 128-bit value (hi,lo), base i at bits 127-2i..126-2i, low 128-2K bits zero.
//                     For K=48 the top 96 bits equal KMer<48>'s three u32 words; word-lexicographic
//                     order == lexicographic order on bases (KMer.h:305-311).
//   context byte    : pred one-hot <<4 | succ one-hot  (KMerContext.h:27-28,36-37,55-56)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SNK_HD __host__ __device__ __forceinline__
#else
#define SNK_HD static inline
#endif

// ---------------------------------------------------------------- hashing (counter based RNG + table hash)
SNK_HD uint64_t snk_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
SNK_HD uint64_t snk_rng(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return snk_mix64(seed ^ snk_mix64(stream * 0xD1B54A32D192ED03ull + ctr));
}
SNK_HD uint32_t snk_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

#if defined(__HIPCC__)
// inclusive prefix sum over the 64 lanes of a wave without LDS traffic (six DPP adds: inside the rows of 16, then across them).
// Lanes that are switched off contribute nothing and get nothing; the callers run it with the whole wave active.
__device__ __forceinline__ uint32_t snk_wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
// Every lane of a wave asks for v slots (0 allowed) behind an LDS counter: exclusive position of the lane.  One LDS atomic per
// wave -- the compiler's own treatment of a divergent-valued atomicAdd is a scalar loop over the active lanes.
__device__ __forceinline__ uint32_t snk_wave_alloc(uint32_t* counter, uint32_t v) {
    const uint32_t incl = snk_wave_scan_incl(v);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(counter, total);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
    return base + incl - v;
}
__device__ __forceinline__ void snk_wave_add(uint32_t* counter, uint32_t v) {
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)snk_wave_scan_incl(v), 63);
    if ((threadIdx.x & 63) == 63 && total) atomicAdd(counter, total);
}
#endif

// ---------------------------------------------------------------- 128-bit k-mer value
struct snk_kmer {
    uint64_t hi, lo;
};
SNK_HD bool snk_kmer_lt(snk_kmer a, snk_kmer b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
SNK_HD bool snk_kmer_eq(snk_kmer a, snk_kmer b) { return a.hi == b.hi && a.lo == b.lo; }

// reverse the 32 two-bit groups of a 64-bit word
SNK_HD uint64_t snk_rev2(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
#else
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
#endif
}
SNK_HD uint32_t snk_rev2_32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
#else
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
// reverse complement of a K-base value (K even or odd, K<=64); SURVEY App. A.3
template <int K>
SNK_HD snk_kmer snk_kmer_rc(snk_kmer k) {
    uint64_t a = snk_rev2(~k.lo), b = snk_rev2(~k.hi);  // full 64-group reversal: (a,b)
    constexpr int S = 128 - 2 * K;                      // drop the reversed padding
    snk_kmer r;
    if constexpr (S == 0) { r.hi = a; r.lo = b; }
    else if constexpr (S < 64) { r.hi = (a << S) | (b >> (64 - S)); r.lo = b << S; }
    else if constexpr (S == 64) { r.hi = b; r.lo = 0; }
    else { r.hi = b << (S - 64); r.lo = 0; }
    return r;
}
// append base b on the right (drop the leftmost): KMer::toSuccessor, KMer.h:203-216
template <int K>
SNK_HD snk_kmer snk_kmer_succ(snk_kmer k, uint32_t b) {
    snk_kmer r;
    if constexpr (K <= 32) { r.hi = (k.hi << 2) | ((uint64_t)b << (64 - 2 * K)); r.lo = 0; }
    else { r.hi = (k.hi << 2) | (k.lo >> 62); r.lo = (k.lo << 2) | ((uint64_t)b << (128 - 2 * K)); }
    return r;
}
// prepend base b on the left (drop the rightmost): KMer::toPredecessor, KMer.h:189-201
template <int K>
SNK_HD snk_kmer snk_kmer_pred(snk_kmer k, uint32_t b) {
    snk_kmer r;
    r.lo = (k.lo >> 2) | (k.hi << 62);
    r.hi = (k.hi >> 2) | ((uint64_t)b << 62);
    if constexpr (K < 32) { r.lo = 0; r.hi &= ~((1ull << (64 - 2 * K)) - 1); }
    else if constexpr (K == 32) { r.lo = 0; }
    else if constexpr (K < 64) { r.lo &= ~((1ull << (128 - 2 * K)) - 1); }
    return r;
}
template <int K>
SNK_HD uint32_t snk_kmer_base(snk_kmer k, int i) {
    return i < 32 ? (uint32_t)(k.hi >> (62 - 2 * i)) & 3u : (uint32_t)(k.lo >> (62 - 2 * (i - 32))) & 3u;
}
// reverse-complement a context byte = reverse its 8 bits (KMerContext.cc:19 gRCVals)
SNK_HD uint32_t snk_ctx_rc(uint32_t c) {
    c = ((c >> 4) & 0x0F) | ((c & 0x0F) << 4);
    c = ((c >> 2) & 0x33) | ((c & 0x33) << 2);
    c = ((c >> 1) & 0x55) | ((c & 0x55) << 1);
    return c;
}
// two 32-bit hashes of a k-mer value: each word is multiplied by its own odd constant, the products are combined (xor for
// one hash, add for the other) and finalised.  12 integer multiplies instead of the 24 of a murmur3 round per word --
// this hash runs once per k-mer instance in the count kernel.
SNK_HD uint32_t snk_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
SNK_HD void snk_kmer_hash2(snk_kmer k, uint32_t* h1out, uint32_t* h2out) {
    const uint32_t w0 = (uint32_t)(k.hi >> 32), w1 = (uint32_t)k.hi, w2 = (uint32_t)(k.lo >> 32), w3 = (uint32_t)k.lo;
    const uint32_t a = (w0 * 0xcc9e2d51u) ^ snk_rotl32(w1 * 0x1b873593u, 11) ^ (w2 * 0x85ebca77u) ^ snk_rotl32(w3 * 0xc2b2ae3du, 19);
    const uint32_t b = (w0 * 0x9e3779b1u) + snk_rotl32(w1 * 0x27d4eb2fu, 7) + (w2 * 0x165667b1u) + snk_rotl32(w3 * 0xd3a2646du, 17);
    *h1out = snk_mix32(a ^ 0x9747b28cu);
    *h2out = snk_mix32(b + 0x3c6ef372u);
}
// The count kernel's hash (once per inserted k-mer, so it is kept short: one multiply per key word, one avalanche round for
// the slot word h1, one more multiply for the tag / split word h2).  W3 = false: the low key word is known to be zero
// (K <= 48, ungrouped) and is left out -- the value is the same as with W3 = true on such a key.  The bucket-local graph
// stage derives the hash-split id of a neighbour from h2 (snk_local.hip), so both sides use this function.
template <bool W3>
SNK_HD void snk_kmer_hash_count(snk_kmer k, uint32_t* h1out, uint32_t* h2out) {
    const uint32_t w0 = (uint32_t)(k.hi >> 32), w1 = (uint32_t)k.hi, w2 = (uint32_t)(k.lo >> 32), w3 = (uint32_t)k.lo;
    uint32_t a = (w0 * 0xcc9e2d51u) ^ snk_rotl32(w1 * 0x1b873593u, 13) ^ (w2 * 0x85ebca77u);
    if (W3) a ^= snk_rotl32(w3 * 0xc2b2ae3du, 19);
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15;
    uint32_t b = a * 0x846ca68bu;
    b ^= b >> 16;
    *h1out = a;
    *h2out = b;
}

// ---------------------------------------------------------------- minimiser order and bucket (shared by K3/K4 and
// the sharded graph stage, which must find the bucket -- hence the owner rank -- of an arbitrary k-mer)
// ordering key of an M=16-mer given its code and its reverse complement's code: hash of the canonical one
SNK_HD uint32_t snk_minimizer_key(uint32_t code, uint32_t rcode) { return snk_mix32(code < rcode ? code : rcode); }
// bucket = re-mixed ordering key (see snk_msp.hip: never the raw key, never anything but the key)
SNK_HD uint32_t snk_bucket_of_key(uint32_t key, uint32_t NB) {
    uint32_t h = snk_mix32(key ^ 0x5bd1e995u) * 0x9E3779B1u;
    h ^= h >> 15;
    return (uint32_t)(((uint64_t)h * NB) >> 32);
}
// grouped runs (per-barcode local graphs, BASELINE config 5): a k-mer belongs to (group, k-mer); the group id rides in
// the 32 low bits of the 128-bit key (free at K=48) and is folded into the bucket choice
SNK_HD uint32_t snk_group_mix(uint32_t group) { return snk_mix32(group * 0x9E3779B1u + 0x7F4A7C15u); }
// bucket of a k-mer = bucket of the minimum ordering key over its K-15 16-mers (strand symmetric)
// M-mers longer than 16 bases: codes are 64-bit, the ordering key stays a 32-bit hash of the canonical one
SNK_HD uint64_t snk_rev2_64(uint64_t x) { return ((uint64_t)snk_rev2_32((uint32_t)x) << 32) | snk_rev2_32((uint32_t)(x >> 32)); }
SNK_HD uint32_t snk_minimizer_key64(uint64_t code, uint64_t rcode) {
    const uint64_t c = code < rcode ? code : rcode;
    return snk_mix32((uint32_t)c ^ ((uint32_t)(c >> 32) * 0x9E3779B1u + 0x7F4A7C15u));
}
// ordering key of the M-mer in the top 2M bits of v (the bases from some position of a k-mer on, left-aligned)
template <int M>
SNK_HD uint32_t snk_mmer_key_top(uint64_t v) {
    if constexpr (M <= 16) {
        const uint32_t x = (uint32_t)(v >> (64 - 2 * M));                  // the M bases as a number
        const uint32_t rx = snk_rev2_32(~(x << (32 - 2 * M))) & (M == 16 ? 0xFFFFFFFFu : ((1u << (2 * (M & 15))) - 1u));   // ... and their reverse complement
        return snk_minimizer_key(x, rx);
    } else {
        static_assert(M <= 32, "the M-mer is held in one 64-bit word");
        const uint64_t top = v & ~((1ull << (64 - 2 * M)) - 1ull);
        const uint64_t x = top >> (64 - 2 * M);
        const uint64_t rx = snk_rev2_64(~top) & ((1ull << (2 * M)) - 1ull);
        return snk_minimizer_key64(x, rx);
    }
}
template <int K, int M>
SNK_HD uint32_t snk_bucket_of_kmer(snk_kmer k, uint32_t NB) {
    uint32_t best = 0xFFFFFFFFu;
    for (int p = 0; p + M <= K; ++p) {
        uint64_t v;
        if (p == 0) v = k.hi;
        else if (p < 32) v = (k.hi << (2 * p)) | (k.lo >> (64 - 2 * p));
        else if (p == 32) v = k.lo;
        else v = k.lo << (2 * p - 64);
        uint32_t key = snk_mmer_key_top<M>(v);
        best = key < best ? key : best;
    }
    return snk_bucket_of_key(best, NB);
}

// base i of a packed row
SNK_HD uint32_t snk_row_base(const uint32_t* row, int i) { return (row[i >> 4] >> (30 - 2 * (i & 15))) & 3u; }
