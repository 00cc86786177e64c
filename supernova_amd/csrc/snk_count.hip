// snk_count.hip -- K5..K8: supermers of one minimiser bucket -> canonical k-mers -> counts, barcode
// rule, OR of contexts -> filtered table.  One workgroup per bucket, the whole reduce in LDS.
//
// What it replaces:
//   Kmerizer::map           lib/assembly/src/paths/long/BuildReadQGraph48.cc:155-172 (roll, canonicalise, context)
//   MapReduceEngine reduce  lib/assembly/src/MapReduceEngine.h:574-584 (std::sort + run detection)
//   summarizeEntries        BuildReadQGraph48.cc:92-105   (sum of counts, OR of contexts)
//   areIgnoredBarcodes / areEnoughBarcodes  :108-137       (>= minBC distinct barcodes > 0, or any bc == -1)
//   Kmerizer::reduce        :174-181                       (keep iff count >= minFreq && bc test)
//   == process_kmer_shard_opt, lib/tada/src/utils.rs:322-408 (sort + group_by per shard).
//
// The reference materialises one 24-byte record per k-mer instance and comparison-sorts them.  Here
// the instances never leave the CU: a bucket's supermers (32 B per ~17 k-mers) are streamed from HBM
// once, every k-mer is rolled in registers and inserted into an open-addressing hash table in LDS
// (key 96/120 bit, count, barcode state, context byte) with LDS atomics.  A bucket whose distinct
// k-mers do not fit is re-run split by a second hash (2, 4, ... sub-passes) -- correctness never
// depends on the bucket sizing.  Barcode state machine (sufficient for minBC <= 2): 0 = none yet,
// id = exactly one barcode id seen, MULTI = two different ids, IGN = a bc == -1 read contributed.
#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"

namespace {

constexpr uint32_t BC_MULTI = 0xFFFFFFFEu;
constexpr uint32_t BC_IGN = 0xFFFFFFFFu;
constexpr uint32_t CNT_LOCK = 0x80000000u;
constexpr int MAX_SPLIT_LOG2 = 16;

template <int K> struct lo_t { typedef uint32_t type; };       // K<=48: only the top 32 bits of lo are used
template <> struct lo_t<60> { typedef uint64_t type; };

template <int K> __device__ __forceinline__ typename lo_t<K>::type lo_pack(uint64_t lo);
template <> __device__ __forceinline__ uint32_t lo_pack<48>(uint64_t lo) { return (uint32_t)(lo >> 32); }
template <> __device__ __forceinline__ uint64_t lo_pack<60>(uint64_t lo) { return lo; }
template <int K> __device__ __forceinline__ uint64_t lo_unpack(typename lo_t<K>::type v);
template <> __device__ __forceinline__ uint64_t lo_unpack<48>(uint32_t v) { return (uint64_t)v << 32; }
template <> __device__ __forceinline__ uint64_t lo_unpack<60>(uint64_t v) { return v; }

template <int K, int THREADS, int SLOTS>
__global__ void __launch_bounds__(THREADS) snk_count_kernel(snk_count_args a) {
    typedef typename lo_t<K>::type lo_type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* khi = reinterpret_cast<uint64_t*>(smem_raw);                         // [SLOTS]
    lo_type* klo = reinterpret_cast<lo_type*>(khi + SLOTS);                         // [SLOTS]
    uint32_t* cnt = reinterpret_cast<uint32_t*>(klo + SLOTS);                       // [SLOTS] 0 empty, LOCK while being claimed
    uint32_t* bcs = cnt + SLOTS;                                                    // [SLOTS]
    uint32_t* ctxw = bcs + SLOTS;                                                   // [SLOTS/4] context bytes
    uint32_t* ctl = ctxw + SLOTS / 4;                                               // control words
    // ctl[0] stack pointer, ctl[1] occupied slots, ctl[2] overflow flag, ctl[3] current split log2, ctl[4] current split id
    // ctl[8..8+2*MAX) split stack
    volatile uint32_t* vctl = ctl;
    const int tid = threadIdx.x;
    const uint32_t bucket = blockIdx.x;
    constexpr uint32_t LIMIT = SLOTS - THREADS - 64;   // claims allowed before a sub-pass is declared overflowing

    if (tid == 0) { ctl[0] = 1; ctl[8] = 0; ctl[9] = 0; }
    uint32_t splits_done = 0;
    for (;;) {
        __syncthreads();
        if (vctl[0] == 0) break;
        __syncthreads();
        if (tid == 0) {
            uint32_t sp = ctl[0] - 1;
            ctl[0] = sp;
            ctl[3] = ctl[8 + 2 * sp];
            ctl[4] = ctl[9 + 2 * sp];
            ctl[1] = 0;
            ctl[2] = 0;
        }
        for (int s = tid; s < SLOTS; s += THREADS) { cnt[s] = 0; bcs[s] = 0; }
        for (int s = tid; s < SLOTS / 4; s += THREADS) ctxw[s] = 0;
        __syncthreads();
        const uint32_t split_lg = vctl[3], split_id = vctl[4];
        const uint32_t split_mask = (1u << split_lg) - 1u;

        for (uint32_t seg = 0; seg < a.nseg; ++seg) {
            const uint64_t beg = a.seg_off[(uint64_t)seg * (a.NB + 1) + bucket];
            const uint64_t end = a.seg_off[(uint64_t)seg * (a.NB + 1) + bucket + 1];
            for (uint64_t base = beg; base < end; base += THREADS) {
                const uint64_t idx = base + tid;
                uint32_t nkm = 0;
                snk_kmer f = {0, 0}, rc = {0, 0};
                uint64_t t_hi = 0, t_lo = 0;
                uint32_t pred = 0, havepred = 0, hasR = 0;
                int32_t bc = 0;
                if (idx < end && vctl[2] == 0) {
                    uint4 r0 = a.records[idx * 2], r1 = a.records[idx * 2 + 1];
                    uint64_t X0 = ((uint64_t)r0.x << 32) | r0.y, X1 = ((uint64_t)r0.z << 32) | r0.w;
                    uint64_t X2 = ((uint64_t)r1.x << 32) | r1.y, X3 = (uint64_t)(r1.z & 0xFFFFF000u) << 32;
                    uint32_t meta = r1.z & 0xFFFu;
                    nkm = meta & 0x7Fu;
                    uint32_t hasL = (meta >> 7) & 1u;
                    hasR = (meta >> 8) & 1u;
                    bc = (int32_t)r1.w;
                    havepred = hasL;
                    pred = (uint32_t)(X0 >> 62);
                    const uint32_t s = 2u * hasL;              // first k-mer starts at base hasL of the record
                    // first k-mer = 2K bits at bit offset s
                    uint64_t A0 = s ? ((X0 << s) | (X1 >> (64 - s))) : X0;
                    uint64_t A1 = s ? ((X1 << s) | (X2 >> (64 - s))) : X1;
                    f.hi = A0;
                    f.lo = A1 & ~((1ull << (128 - 2 * K)) - 1ull);
                    rc = snk_kmer_rc<K>(f);
                    // tail = bases following the first k-mer, MSB aligned: bit offset s + 2K  (in [64,128))
                    const uint32_t off = s + 2u * K - 64u;     // offset inside (X1,X2,X3)
                    t_hi = (X1 << off) | (X2 >> (64 - off));
                    t_lo = (X2 << off) | (X3 >> (64 - off));
                }
                uint32_t maxn = nkm;
                for (int o = 32; o > 0; o >>= 1) { uint32_t v = __shfl_xor(maxn, o); maxn = v > maxn ? v : maxn; }
                for (uint32_t j = 0; j < maxn; ++j) {
                    if (j < nkm) {
                        const uint32_t nb = (uint32_t)(t_hi >> 62);
                        const uint32_t havesucc = (j + 1 < nkm) ? 1u : hasR;
                        uint32_t ctx = (havepred ? (0x10u << pred) : 0u) | (havesucc ? (1u << nb) : 0u);
                        const bool rev = snk_kmer_lt(rc, f);      // isRev(): store the reverse complement (:164)
                        const snk_kmer c = rev ? rc : f;
                        if (rev) ctx = snk_ctx_rc(ctx);
                        uint32_t h1, h2;
                        snk_kmer_hash2(c, &h1, &h2);
                        if ((h2 & split_mask) == split_id) {
                            uint32_t slot = (uint32_t)(((uint64_t)h1 * SLOTS) >> 32);
                            const lo_type clo = lo_pack<K>(c.lo);
                            for (;;) {
                                uint32_t cv = __hip_atomic_load(&cnt[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (cv == 0) {
                                    if (vctl[2]) { nkm = 0; break; }   // sub-pass already declared overflowing: stop claiming
                                    uint32_t old = atomicCAS(&cnt[slot], 0u, CNT_LOCK);
                                    if (old == 0) {
                                        khi[slot] = c.hi;
                                        klo[slot] = clo;
                                        bcs[slot] = bc > 0 ? (uint32_t)bc : (bc == -1 ? BC_IGN : 0u);
                                        if (ctx) atomicOr(&ctxw[slot >> 2], ctx << (8 * (slot & 3)));
                                        __hip_atomic_store(&cnt[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        uint32_t occ = atomicAdd(&ctl[1], 1u);
                                        if (occ >= LIMIT) vctl[2] = 1;
                                        break;
                                    }
                                    continue;   // lost the race: look at the same slot again
                                }
                                if (cv == CNT_LOCK) continue;
                                if (khi[slot] == c.hi && klo[slot] == clo) {
                                    atomicAdd(&cnt[slot], 1u);
                                    if (ctx) atomicOr(&ctxw[slot >> 2], ctx << (8 * (slot & 3)));
                                    if (bc == -1) atomicMax(&bcs[slot], BC_IGN);
                                    else if (bc > 0) {
                                        uint32_t ob = atomicCAS(&bcs[slot], 0u, (uint32_t)bc);
                                        if (ob != 0 && ob != (uint32_t)bc && ob < BC_MULTI) atomicMax(&bcs[slot], BC_MULTI);
                                    }
                                    break;
                                }
                                slot = slot + 1 == SLOTS ? 0 : slot + 1;
                            }
                        }
                        // roll to the next k-mer of the supermer
                        pred = (uint32_t)(f.hi >> 62);
                        havepred = 1;
                        f = snk_kmer_succ<K>(f, nb);
                        rc = snk_kmer_pred<K>(rc, nb ^ 3u);
                        t_hi = (t_hi << 2) | (t_lo >> 62);
                        t_lo <<= 2;
                    }
                }
            }
        }
        __syncthreads();
        if (vctl[2]) {   // too many distinct k-mers for one table: split this sub-pass in two by one more hash bit
            if (tid == 0) {
                if (split_lg >= MAX_SPLIT_LOG2) { atomicExch(&a.status[1], 1u); }
                else {
                    uint32_t sp = vctl[0];
                    ctl[8 + 2 * sp] = split_lg + 1; ctl[9 + 2 * sp] = split_id;
                    ctl[10 + 2 * sp] = split_lg + 1; ctl[11 + 2 * sp] = split_id | (1u << split_lg);
                    ctl[0] = sp + 2;
                }
            }
            ++splits_done;
            continue;
        }
        // K8: filter + compact the surviving entries to the global table
        for (int s0 = 0; s0 < SLOTS; s0 += THREADS) {
            const int s = s0 + tid;
            const uint32_t c = cnt[s];
            bool ok = c >= a.min_freq && c != 0;
            if (ok && a.bc_mode) {
                const uint32_t b = bcs[s];
                ok = a.bc_mode == 1 ? (b != 0) : (b >= BC_MULTI);
            }
            const unsigned long long m = __ballot(ok);
            if (m) {
                const int lane = tid & 63;
                unsigned long long basepos = 0;
                if (lane == 0) basepos = atomicAdd(a.out_cursor, (unsigned long long)__popcll(m));
                basepos = __shfl(basepos, 0);
                if (ok) {
                    const uint64_t pos = basepos + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < a.out_cap) {
                        const uint32_t cx = (ctxw[s >> 2] >> (8 * (s & 3))) & 0xFFu;
                        a.out_keys[pos] = ((snk_u128)khi[s] << 64) | (snk_u128)lo_unpack<K>(klo[s]);
                        a.out_vals[pos] = ((uint64_t)c << 8) | cx;
                    } else {
                        a.status[0] = 1;
                    }
                }
            }
        }
        if (tid == 0) atomicMax(&a.status[3], vctl[1]);
    }
    if (tid == 0 && splits_done) atomicAdd(&a.status[2], 1u);
}

template <int K> struct cfg;
template <> struct cfg<48> { static constexpr int THREADS = 512; static constexpr int SLOTS = 3072; };
template <> struct cfg<60> { static constexpr int THREADS = 512; static constexpr int SLOTS = 2560; };

template <int K>
size_t lds_bytes() {
    return (size_t)cfg<K>::SLOTS * (8 + sizeof(typename lo_t<K>::type) + 4 + 4 + 1) + 4 * (8 + 2 * (MAX_SPLIT_LOG2 + 3) * 2);
}

template <int K>
int launch(hipStream_t st, const snk_count_args& a, char* err, size_t errcap) {
    auto kern = snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS>;
    size_t lds = lds_bytes<K>();
    SNK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(a.NB), dim3(cfg<K>::THREADS), lds, st, a);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

}  // namespace

uint32_t snk_count_slots(uint32_t K) { return K == 60 ? cfg<60>::SLOTS : cfg<48>::SLOTS; }

int snk_launch_count(uint32_t K, hipStream_t st, const snk_count_args& a, char* err, size_t errcap) {
    if (a.NB == 0) return SNK_OK;
    if (K == 48) return launch<48>(st, a, err, errcap);
    if (K == 60) return launch<60>(st, a, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}
