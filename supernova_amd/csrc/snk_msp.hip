// snk_msp.hip -- K3/K4: minimiser scan of every trimmed read, supermer emission into minimiser buckets.
//
// What it replaces:
//   msp::simple_scan            lib/tada/src/msp/mod.rs:60-134   (sliding-window minimiser, slices)
//   Bsp::new / Exts::from_slice lib/tada/src/kmer/mod.rs:367-378, kmer/exts.rs:71-84 (supermer record + flanks)
//   shardio write path          lib/tada/external/rust-shardio/src/shard.rs:184-211 (group by shard)
//   and, for path B, the hash->(pass,bin) map of MapReduceEngine.h:315-326.
// Shard assignment is internal to the reference (App. A.10): counts and unitigs do not depend on it.
// This build uses M-mers (SNK_M_OF(K), snk_kernels.h) ordered by a 32-bit hash of the canonical M-mer (strand symmetric, so a
// k-mer and its reverse complement always land in the same bucket -- the invariant of
// check_consistent_shard, lib/tada/src/kmer/mod.rs:1102-1150).
//
// Layout: one thread per read, 256 reads per workgroup.  The packed rows of the workgroup are staged
// in LDS with one coalesced sweep; every thread then owns one LDS column ([word][thread], conflict
// free).  The window minimum is computed without data-dependent control flow by the block
// decomposition (suffix minima of window-sized blocks in an LDS column, running prefix minimum in a
// register), so a wave never serialises on "rescan on expiry".  Supermer starts are appended to a
// short per-thread LDS list and emitted in a second, short loop.
//
// ONE pass: every bucket owns `cap` record slots (cap = expected supermers per bucket + 5 sigma of the occupancy model in snk_stages.hip; the expectation is
// exact bookkeeping: k-mer instances are known after the trim, a random-order minimiser starts a supermer every
// (W+1)/2 k-mers); a supermer takes slot atomicAdd(cursor) of its bucket, the few that do not fit go to an overflow
// list that is grouped by bucket afterwards and read by the count kernel as a second segment.  Correctness never
// depends on the estimate.  The sharded path runs the same pass and then compacts the used slots into its exact,
// destination-contiguous send buffer (snk_stages.hip).
#include <type_traits>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"

namespace {

constexpr int BD = 256;
constexpr int LCAP = 16;

template <int M>
__device__ __forceinline__ uint32_t mmer_key(const uint32_t* rowL, int tid, uint32_t row_words, int p) {
    uint32_t wi = (uint32_t)p >> 4;
    uint32_t w0 = rowL[wi * BD + tid];
    uint32_t w1 = rowL[(wi + 1) * BD + tid];                 // (row `row_words` of the staged rows is all zero)
    uint32_t s = 2u * ((uint32_t)p & 15u);
    if constexpr (M > 16) {
        const uint32_t wj = wi + 2 < row_words ? wi + 2 : row_words;
        const uint32_t w2 = rowL[wj * BD + tid];
        const uint64_t v = ((((uint64_t)w0 << 32) | w1) << s) | (s ? (uint64_t)(w2 >> (32u - s)) : 0ull);      // 32 bases from p on
        return snk_mmer_key_top<M>(v);
    } else {
    uint32_t x = (uint32_t)(((((uint64_t)w0 << 32) | w1) << s) >> 32);   // 16 bases starting at p, MSB first
    uint32_t rx = snk_rev2_32(~x);                             // reverse complement of those 16 bases
    uint32_t code, rcode;
    if (M == 16) { code = x; rcode = rx; }
    else { code = x >> (32 - 2 * M); rcode = rx & ((1u << (2 * M)) - 1u); }
    return snk_minimizer_key(code, rcode);
    }
}

// Bucket of the supermer whose minimiser sits at position p.  Two rules make this safe:
//  * it must NOT be taken from the ordering key directly: minimisers are window minima of that key, so
//    their keys crowd near zero (a Beta(1,w) law) and most supermers would fall into the lowest few
//    percent of the buckets -- the key is re-mixed first;
//  * it must be a function of the ordering key ONLY (not of the M-mer or its position): when two
//    different M-mers of a window tie on the key, the two strands may pick different ones, and a k-mer
//    and its reverse complement must still meet in one bucket.
template <int M>
__device__ __forceinline__ uint32_t mmer_bucket(const uint32_t* rowL, int tid, uint32_t row_words, int p, uint32_t NB, uint32_t gmix) {
    return snk_bucket_of_key(mmer_key<M>(rowL, tid, row_words, p) ^ gmix, NB);
}

// extract 32 bits starting at base `a + 16*j` of the row column
__device__ __forceinline__ uint32_t row_window(const uint32_t* rowL, int tid, uint32_t row_words, uint32_t a, uint32_t j) {
    uint32_t wi = (a >> 4) + j;
    uint32_t w0 = wi < row_words ? rowL[wi * BD + tid] : 0u;
    uint32_t w1 = wi + 1 < row_words ? rowL[(wi + 1) * BD + tid] : 0u;
    uint32_t s = 2u * (a & 15u);
    return s ? ((w0 << s) | (w1 >> (32u - s))) : w0;
}


// ---- fused quality trim (K1 inside the partition kernel) ----------------------------------------------------------
// The rule is snk_trim.hip's (GoodLenTailFinder, BuildReadQGraph48.cc:65-89): from the 3' end, the first run of K
// quals >= min_qual ends the scan, good length = run start + K.  A read whose last K quals are all good -- nearly all --
// is decided by K/4 + 1 words of its row.  The row of any other lane is read by its wave, a lane per word (one coalesced
// load), the good-qual bits of the four byte positions are collected by ballots, and the highest run of K good bases falls
// out of a few 64-bit shift-ands in scalar registers.  No workgroup barrier, no LDS.
// The separate trim kernel streamed 15 GB in 3 ms with nothing else to do; here the loads ride under the slot atomics.
__device__ __forceinline__ uint32_t qual_ge4(uint32_t x, uint32_t y4) {       // bit 7 of every byte: x.byte >= y.byte (unsigned)
    const uint32_t H = 0x80808080u;
    const uint32_t t = (x | H) - (y4 & ~H);
    return ((x & ~y4) | (~(x ^ y4) & t)) & H;
}
// Good length from the four byte planes of a quality row (bit j of P[b]: base 4j + b has a good qual), all wave-uniform:
// a run of K = 4q good bases from base 4 j0 + b0 is q good words from j0 in the planes b >= b0 and from j0 + 1 in the others.
template <int K>
__device__ __forceinline__ int trim_from_planes(uint64_t P0, uint64_t P1, uint64_t P2, uint64_t P3, int len) {
    constexpr int q = K / 4;
    static_assert(q >= 8 && q <= 16, "run ladder below");
    uint64_t R[4] = {P0, P1, P2, P3};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = (len - b + 3) >> 2;                 // words whose byte b lies inside the read
        uint64_t r = n <= 0 ? 0ull : (n >= 64 ? R[b] : (R[b] & ((1ull << n) - 1ull)));
        r &= r >> 1; r &= r >> 2; r &= r >> 4;             // runs of 8 words
        r &= r >> (q - 8);                                 // runs of q
        R[b] = r;
    }
    int best = -1;
#pragma unroll
    for (int b0 = 0; b0 < 4; ++b0) {
        uint64_t c = ~0ull;
#pragma unroll
        for (int b = 0; b < 4; ++b) c &= (b >= b0) ? R[b] : (R[b] >> 1);
        if (c) { const int p = 4 * (63 - __clzll((long long)c)) + b0; best = p > best ? p : best; }
    }
    return best < 0 ? 0 : best + K;
}

// One call per thread at the top of the partition kernel (inlined; what keeps it from disturbing the kernel's register allocation is
// described at the call).
template <int K>
__device__ __forceinline__ int fused_trim(const snk_msp_args& a, int tid0, uint64_t r0) {
    int g_trim = 0;
        const uint64_t rq = r0 + tid0;
        int qlen = 0;
        if (rq < a.n_reads) {
            qlen = a.lens ? (int)a.lens[rq] : (int)a.read_len;
            if (qlen > (int)a.read_len) qlen = (int)a.read_len;
        }
        const uint32_t mq4 = (a.min_qual > 255u ? 255u : a.min_qual) * 0x01010101u;
        // step 1: the last K quals of every row of the wave, 16 bytes per lane: four lanes cover a row's window (64 contiguous
        // bytes), a wave instruction covers 16 rows.  (A lane reading its own row's 13 words cost the kernel 5.5 ms: it pays per
        // memory request.)  A window without a low qual decides the read.
        const int lane = tid0 & 63;
        const uint8_t* wave_rows = a.quals + (r0 + (uint64_t)(tid0 & ~63)) * a.qstride;
        const int rw_all = (int)(a.qstride >> 2);
        uint32_t mybad = 0;
        {
            struct __attribute__((packed, aligned(4))) q16 { uint32_t w[4]; };
            q16 xv[4];
            int st[4], lo_[4], ql_[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int R = 16 * it + (lane >> 2);
                ql_[it] = __builtin_amdgcn_ds_bpermute(4 * R, qlen);      // (explicit addresses off tid0: nothing here is shared with the code behind the trim)
                lo_[it] = ql_[it] - K;
                int start = (lo_[it] >> 2) + 4 * (lane & 3);
                if (start + 4 > rw_all) start = rw_all - 4;
                st[it] = start;
                xv[it].w[0] = xv[it].w[1] = xv[it].w[2] = xv[it].w[3] = 0xFFFFFFFFu;
                if (lo_[it] >= 0) xv[it] = *reinterpret_cast<const q16*>(wave_rows + (uint64_t)R * a.qstride + 4 * start);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                uint32_t bad = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j4 = 4 * (st[it] + k);
                    int lb = lo_[it] - j4, hb = ql_[it] - j4;                 // bytes [lb, hb) of this word lie inside the window
                    lb = lb < 0 ? 0 : (lb > 4 ? 4 : lb);
                    hb = hb < 0 ? 0 : (hb > 4 ? 4 : hb);
                    const uint32_t below_h = hb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hb)) - 1u);
                    const uint32_t below_l = lb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * lb)) - 1u);
                    bad |= ~qual_ge4(xv[it].w[k], mq4) & 0x80808080u & below_h & ~below_l;
                }
                if (lo_[it] < 0) bad = 0;
                bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
                bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(16 * (lane & 15), (int)bad);
                if ((lane >> 4) == it) mybad = v;
            }
        }
        bool scan = false;
        if (qlen >= K && a.min_qual <= 255u) { g_trim = qlen; scan = mybad != 0u; }
#ifndef SNK_FT_NOSLOW
        // step 2: the rows that need the scan, one after the other, by the whole wave (four rows' loads in flight)
        unsigned long long todo = __ballot(scan);
        const uint32_t rw = (a.qstride >> 2) < 40u ? (a.qstride >> 2) : 40u;     // words of a row that can matter (read_len <= 160)
        while (todo) {
            int own[4];
            uint32_t x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                own[i] = todo ? __ffsll((long long)todo) - 1 : -1;
                if (own[i] >= 0) todo &= todo - 1ull;
                x[i] = 0u;
                if (own[i] >= 0 && (uint32_t)lane < rw) x[i] = reinterpret_cast<const uint32_t*>(wave_rows + (uint64_t)own[i] * a.qstride)[lane];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (own[i] < 0) break;
                const uint32_t gq = ((uint32_t)lane < rw) ? qual_ge4(x[i], mq4) : 0u;
                const uint64_t P0 = __ballot((gq >> 7) & 1u), P1 = __ballot((gq >> 15) & 1u), P2 = __ballot((gq >> 23) & 1u), P3 = __ballot((gq >> 31) & 1u);
                const int len_o = __builtin_amdgcn_readlane(qlen, own[i]);
                const int g_o = trim_from_planes<K>(P0, P1, P2, P3, len_o);
                if (lane == own[i]) g_trim = g_o;
            }
        }
#endif
        if (rq < a.n_reads) a.good_out[rq] = (uint16_t)g_trim;
    return g_trim;
}

// Occupancy is what overlaps one wave's emit loop (slot atomics, scattered stores: device throughput limits) with other
// waves' scans: 26.8 KB of LDS admit five workgroups per CU, and at K=48 the kernel fits 96 VGPRs with two spills
// (36.0 -> 33.0 ms at 1e8 reads against the 105 registers / four waves per SIMD the compiler picks on its own).  Keeping
// the minimisers' keys in a second LDS list to save their re-derivation in the emit loop costs the fifth workgroup and
// was dropped again; six waves per SIMD (80 VGPRs) spill 28 registers.  K=60 (45 keys in registers) stays at four.
template <int K, int M, bool TRIM, bool DENSE, bool RANGED = false>
#ifndef SNK_MSP_OCC48
#define SNK_MSP_OCC48 5
#endif
__global__ void __launch_bounds__(BD, K == 48 ? SNK_MSP_OCC48 : 4) snk_msp_kernel(snk_msp_args a) {
    constexpr int W = K - M + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t row_words = a.row_words;
    const uint32_t NB = a.NB;
    const uint64_t n_reads = a.n_reads;
    uint32_t* rowL = smem;                              // [row_words + 1][BD]; the last row is zero: words behind a read's row are read there
    // Only the POSITION of every suffix minimum is kept in LDS (1 byte per entry); the keys live in registers.
    uint16_t* lst = reinterpret_cast<uint16_t*>(rowL + (size_t)(row_words + 1) * BD);   // [LCAP][BD] minimiser position << 8 | first k-mer
    uint8_t* sfxp = reinterpret_cast<uint8_t*>(lst + (size_t)LCAP * BD);  // [W][BD] position of the suffix minimum
    uint32_t* hotL = reinterpret_cast<uint32_t*>(sfxp + (size_t)W * BD);  // [SNK_MSP_HOT_TAB] buckets that stopped reserving slots (bucket + 1)
    static_assert(SNK_MSP_HOT_TAB == BD && (W * BD) % 4 == 0, "one table entry per thread");
    const int tid0 = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * BD;
    // fused trim: finished before the rows are staged (its registers are free again when the staging loads go out; live
    // across them they cost spills whose reloads serialise those loads)
    static_assert(K % 4 == 0 && K >= 32 && K < 96, "the window below is K/4 + 1 aligned words");
    // (the length goes through good_out and is read back below where the kernel without the trim reads good_len: everything
    // of the trim is dead before the staging, and the body behind it compiles as it does without -- carried in a register it
    // is spilled, and reloaded from scratch in every turn of the emit loop: +5 ms)
    if (TRIM) (void)fused_trim<K>(a, tid0, r0);
    // (a fresh value behind the trim: the register allocator would otherwise carry the thread id through it in scratch and reload
    // it before every staging load below, one load in flight at a time)
    int tid = tid0;
    if (TRIM) asm volatile("" : "+v"(tid) :: "memory");
    // coalesced stage of the workgroup's rows
    {
        uint64_t nrows = n_reads - r0 < (uint64_t)BD ? n_reads - r0 : (uint64_t)BD;
        uint32_t total = (uint32_t)nrows * row_words;
        const uint32_t* src = a.rows + r0 * row_words;
        // row_words (<= 16) words per thread, all loads in flight before the first LDS store
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t idx = tid + q * BD;
            v[q] = ((uint32_t)q < row_words && idx < total) ? src[idx] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t idx = tid + q * BD;
            if ((uint32_t)q < row_words && idx < total) {
                const uint32_t t = idx / row_words, w = idx - t * row_words;
                rowL[w * BD + t] = v[q];
            }
        }
    }
    rowL[row_words * BD + tid] = 0u;
    // (an agent-scope load: the table is written by other workgroups of this launch, a plain load may be served from a stale line for good)
    if (!DENSE) hotL[tid] = a.hot_tab ? __hip_atomic_load(&a.hot_tab[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    const uint64_t r = r0 + tid;
    int g;
    g = (r < n_reads) ? (int)a.good_len[r] : 0;        // (fused trim: good_len == good_out, written above by this thread)
    if (g > (int)a.read_len) g = (int)a.read_len;     // a caller-supplied good length never reaches past the packed row
    if (g < K + 1) g = 0;                               // reads with fewer than 2 k-mers are skipped (:160)
    const int nk = g ? g - K + 1 : 0;
    const int npos = g ? g - M + 1 : 0;
    if (TRIM) {
        // the sizing figures of snk_msp_plan_kernel, per wave, spread over SNK_MSP_PLAN_SLOTS counters (same-address atomics queue)
        unsigned long long inst = (unsigned long long)nk, live = g ? 1ull : 0ull;
        for (int off = 32; off > 0; off >>= 1) { inst += __shfl_xor(inst, off); live += __shfl_xor(live, off); }
        if ((tid & 63) == 0 && live) {
            const uint32_t slot = (blockIdx.x * (BD / 64) + (tid >> 6)) % SNK_MSP_PLAN_SLOTS;
            atomicAdd(&a.plan[2 * slot], inst);
            atomicAdd(&a.plan[2 * slot + 1], live);
        }
    }
    int32_t mybc = 0;
    // word 7 of a record is the barcode STATE the count kernel starts from: 0 none, id > 0, 0xFFFFFFFF (-1) a read under the
    // ignore rule; other non-positive ids count as none (areEnoughBarcodes only looks at ids > 0, BuildReadQGraph48.cc:117-137)
    if (g) { mybc = (a.bc && (int64_t)(a.read_index_base + r) >= a.ign_bc_below) ? a.bc[r] : -1; if (mybc < -1) mybc = 0; }
    uint32_t gmix = 0;                                   // grouped runs: word 7 of the record carries the group id
    if (a.group && g) { const uint32_t grp = a.group[r]; gmix = snk_group_mix(grp); mybc = (int32_t)grp; }

    int curpos = -1;              // minimiser position of the open supermer
    int cnt = 0;                  // closed+open supermer starts in the list

    // emit list entries [0, upto) ; entry e covers k-mers [start_e, start_{e+1}-1], the last one ends at last_end.
    // Measured on the 100 M-read workload (tools/msp_probe.py, SNK_MSP_DBG): scan + record stores 33 ms; with the
    // 0.68 G slot reservations 38 ms (random global atomics run at ~27 G/s, tools/probe/atomics.hip, and overlap with
    // other waves' scans).  What made this kernel take 65 ms for a while was the OVERFLOW path: with a capacity of
    // mean + 4 sqrt(mean), 1.7 % of the supermers overflowed (bucket occupancy is far from Poisson, see the sizing in
    // snk_pipeline.hip) and each took a returning atomic on the ONE overflow cursor (tools/msp_probe2.py).
    uint32_t dense_base = 0;          // DENSE: position of this thread's first record of the final flush
    auto flush = [&](int upto, int last_end, bool last) {
        if (a.dbg == 3) { if (upto == 0x7FFF && last_end == 0x7FFF) a.cursor[0] = 1; return; }      // probe: the scan alone
        int maxn = upto;
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(maxn, off); maxn = o > maxn ? o : maxn; }
        for (int e = 0; e < maxn; ++e) {
            uint32_t w[8];
            uint64_t at = 0;
            if (e < upto) {
                uint32_t ent = lst[e * BD + tid];
                uint32_t s = ent & 0xFFu;
                uint32_t en = (e + 1 < upto) ? (((uint32_t)lst[(e + 1) * BD + tid] & 0xFFu) - 1u) : (uint32_t)last_end;
                uint32_t bucket = mmer_bucket<M>(rowL, tid, row_words, (int)(ent >> 8), NB, gmix);
                {
                    bool ok = true;
                    uint32_t slot = 0;
                    if (RANGED && (bucket < a.b_lo || bucket >= a.b_hi)) ok = false;      // another pass's bucket
                    else if (DENSE) {
                        // no slot reservation: records go out densely in read order (a workgroup's block was reserved with ONE atomic,
                        // a thread's records follow each other inside it); which bucket a record belongs to is written next to it and
                        // the count kernel finds a bucket's records through a sorted index list (snk_stages.hip)
                        uint64_t pos;
                        if (last) pos = (uint64_t)dense_base + (uint32_t)e;
                        else {
                            // a list drained in mid-read (more than LCAP supermers in one read): one reservation per wave and turn
                            const unsigned long long m = __ballot(1);
                            const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
                            unsigned long long o = 0;
                            if (lane == leader) o = atomicAdd(a.dense_cursor, (unsigned long long)__popcll(m));
                            pos = __shfl(o, leader) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        }
                        if (pos < a.dense_cap) { at = pos; a.dense_bkt[pos] = bucket; }
                        else ok = false;                                   // the host sees the cursor beyond the capacity and re-runs
                    } else {
                    if (hotL[bucket % SNK_MSP_HOT_TAB] == bucket + 1u) slot = 0xFFFFFFFFu;
                    else {
                        slot = a.dbg == 2 ? ((ent * 2654435761u) % (a.cap ? a.cap : 1u)) : atomicAdd(&a.cursor[bucket], 1u);
                        // (noted again every 1024 reservations: two hot buckets may share an entry)
                        if (slot >= a.hot_thr && ((slot - a.hot_thr) & 1023u) == 0u && a.hot_tab)
                            __hip_atomic_store(&a.hot_tab[bucket % SNK_MSP_HOT_TAB], bucket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (slot < a.cap) at = (uint64_t)(RANGED ? bucket - a.b_lo : bucket) * a.cap + slot;
                    else {
                        // overflow list: ONE reservation per wave (same-address atomics are served one at a time)
                        const unsigned long long m = __ballot(1);
                        const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
                        // ... on one of SNK_OVF_SUBLISTS cursors, by wave: with a repeat-rich genome a few per cent of all supermers
                        // overflow, nearly every wave has one in every turn, and ten million reservations on ONE address were 108 ms
                        const uint32_t sub = (blockIdx.x * (BD / 64) + ((uint32_t)tid >> 6)) & (SNK_OVF_SUBLISTS - 1u);
                        uint32_t o = 0;
                        if (lane == leader) o = atomicAdd(&a.ovf_cursor[sub * SNK_OVF_CUR_STRIDE], (uint32_t)__popcll(m));
                        o = __shfl(o, leader) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (o < a.ovf_cap) { const uint32_t g = sub * a.ovf_cap + o; at = a.ovf_base + g; a.ovf_bucket[g] = bucket; }
                        else ok = false;                               // the host sees a cursor beyond its sub-list and re-runs
                    }
                    }
                    if (ok && a.dbg != 1) {
                        uint32_t n_kmers = en - s + 1u;
                        uint32_t hasL = s > 0 ? 1u : 0u;
                        uint32_t hasR = (en + (uint32_t)K < (uint32_t)g) ? 1u : 0u;
                        uint32_t a0 = s - hasL;
                        uint32_t bits = 2u * (n_kmers + (uint32_t)K - 1u + hasL + hasR);
                        // the record's seven base words: eight consecutive row words (behind the row: the zero row), one funnel shift
                        // and one mask each -- no word-index tests (the first version compiled into seven exec-masked blocks)
                        uint32_t rw[8];
                        const uint32_t wi0 = a0 >> 4, fs = 2u * (a0 & 15u);
#pragma unroll
                        for (uint32_t q = 0; q < 8; ++q) rw[q] = rowL[min(wi0 + q, row_words) * BD + tid];
#pragma unroll
                        for (uint32_t j = 0; j < 7; ++j) {
                            const uint32_t x = (uint32_t)(((((uint64_t)rw[j] << 32) | rw[j + 1]) << fs) >> 32);
                            int rbits = (int)bits - 32 * (int)j;
                            rbits = rbits < 0 ? 0 : (rbits > 32 ? 32 : rbits);
                            w[j] = x & (uint32_t)(0xFFFFFFFF00000000ull >> rbits);     // the top rbits bits
                        }
                        w[6] |= n_kmers | (hasL << 7) | (hasR << 8);
                        w[7] = (uint32_t)mybc;
                        uint4* dst = a.records + at * 2;
#ifdef SNK_MSP_NT
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store((v4u){w[0], w[1], w[2], w[3]}, reinterpret_cast<v4u*>(dst));
                        __builtin_nontemporal_store((v4u){w[4], w[5], w[6], w[7]}, reinterpret_cast<v4u*>(dst) + 1);
#else
                        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
#endif
                    }
                }
            }
        }
    };

    const int nblocks = (nk + W - 1) / W;
    int maxblocks = nblocks;
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(maxblocks, off); maxblocks = o > maxblocks ? o : maxblocks; }
    // Every ordering key is computed ONCE, by one forward roll over the read: x = the 16 bases at the current
    // position, rx = their reverse complement, both updated with the next base (the row word is consumed two bits
    // at a time; all lanes are at the same position, so the refill is a uniform branch).  kk[t] (registers, static
    // indices -- both passes are fully unrolled) holds the raw keys of the next block after a forward pass and that
    // block's suffix-minimum keys after its suffix pass.
    static_assert(M >= 8 && M <= 31, "the rolling window below is one 32-bit word, or one 64-bit word for M > 16");
    typedef typename std::conditional<(M > 16), uint64_t, uint32_t>::type mm_t;
    constexpr mm_t MMASK = (M == 16 || M == 32) ? ~(mm_t)0 : (((mm_t)1 << (2 * (M & 31))) - (mm_t)1);
    uint32_t kk[W];
    mm_t x = 0, rx = 0;
    uint32_t cw = 0;
    int rp = 0;                                  // position of x
    auto roll = [&]() {                          // advance x/rx to position rp+1
        const int nb = rp + M;                   // index of the incoming base
        if ((nb & 15) == 0) { const uint32_t wi = (uint32_t)nb >> 4; cw = wi < row_words ? rowL[wi * BD + tid] : 0u; }
        const uint32_t base = cw >> 30;
        cw <<= 2;
        x = ((x << 2) | (mm_t)base) & MMASK;
        rx = (rx >> 2) | ((mm_t)(base ^ 3u) << (2 * M - 2));
        ++rp;
    };
    auto key_now = [&]() -> uint32_t {
        if constexpr (M > 16) return snk_minimizer_key64(x, rx);
        else return snk_minimizer_key(x, rx);
    };
    if (maxblocks > 0) {
        const uint32_t w0 = rowL[tid];
        if constexpr (M > 16) {
            const uint32_t w1 = rowL[BD + tid];                  // (a one-word row: the zero row)
            const uint64_t v0 = ((uint64_t)w0 << 32) | w1;
            x = v0 >> (64 - 2 * M);                              // the first M bases as a number,
            rx = snk_rev2_64(~(v0 & ~((1ull << (64 - 2 * M)) - 1ull))) & MMASK;      // their reverse complement,
            cw = w1 << (2 * (M - 16));                           // and the bases that follow, next one on top
        } else {
            x = w0 >> (32 - 2 * (M & 31));           // the first M bases as a number,
            rx = snk_rev2_32(~w0) & MMASK;           // their reverse complement,
            cw = M == 16 ? (row_words > 1 ? rowL[BD + tid] : 0u) : (w0 << (2 * (M & 15)));      // and the bases that follow, next one on top
        }
#pragma unroll
        for (int t = 0; t < W; ++t) {            // raw keys of block 0
            kk[t] = (t < npos) ? key_now() : 0xFFFFFFFFu;
            roll();
        }
    }
    for (int b = 0; b < maxblocks; ++b) {
        // suffix minima of block b (positions b*W .. b*W+W-1), right to left; "<=" keeps the leftmost on ties
        uint32_t run = 0xFFFFFFFFu;
        int runp = 0;
#pragma unroll
        for (int t = W - 1; t >= 0; --t) {
            const uint32_t key = kk[t];
            if ((b * W + t) < npos && key <= run) { run = key; runp = b * W + t; }
            kk[t] = run;
            sfxp[t * BD + tid] = (uint8_t)(run == 0xFFFFFFFFu ? 0xFF : runp);     // 0xFF: no valid position in this suffix
        }
        // k-mers of block b: window = suffix of block b from t  U  prefix of block b+1 of length t.  Fast path: no
        // list drain inside the unrolled loop; a lane whose list is full only notes it, and the block is redone below.
        const int curpos0 = curpos, cnt0 = cnt;
        bool lost = false;
        {
            uint32_t pfx = 0xFFFFFFFFu;
            int pfxp = 0;
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const int i = b * W + t;
                const int sp = (int)sfxp[t * BD + tid];
                const uint32_t sv = kk[t];
                const int candp = (sv <= pfx) ? sp : pfxp;    // the suffix part lies left of the prefix part
                const bool isnew = (i < nk) && (candp != curpos);
                if (isnew) {
                    curpos = candp;
                    if (cnt < LCAP) { lst[cnt * BD + tid] = (uint16_t)(((uint32_t)candp << 8) | (uint32_t)i); ++cnt; }
                    else lost = true;
                }
                const int p2 = (b + 1) * W + t;          // == rp: the roll is exactly one block ahead
                const bool v2 = p2 < npos;
                const uint32_t k2 = v2 ? key_now() : 0xFFFFFFFFu;
                kk[t] = k2;
                if (v2 && k2 < pfx) { pfx = k2; pfxp = p2; }
                roll();
            }
        }
        if (__any(lost)) {
            // some read of the wave has more than LCAP supermers (low-complexity sequence): redo this block with the
            // list drained whenever it is full.  Keys come from the staged rows again (kk already holds block b+1).
            curpos = curpos0;
            cnt = cnt0;
            uint32_t pfx = 0xFFFFFFFFu;
            int pfxp = 0;
            uint32_t sv = 0xFFFFFFFFu;
            int svp = -1;
            for (int t = 0; t < W; ++t) {
                const int i = b * W + t;
                const int sp = (int)sfxp[t * BD + tid];
                if (sp != svp) { svp = sp; sv = sp == 0xFF ? 0xFFFFFFFFu : mmer_key<M>(rowL, tid, row_words, sp); }
                const int candp = (sv <= pfx) ? sp : pfxp;
                const bool isnew = (i < nk) && (candp != curpos);
                if (__any(isnew && cnt == LCAP)) {     // every lane of the wave drains its list; the open supermer stays
                    const int upto = cnt > 0 ? cnt - 1 : 0;
                    const int last_end = cnt > 0 ? (int)(lst[(cnt - 1) * BD + tid] & 0xFFu) - 1 : 0;
                    flush(upto, last_end, false);
                    if (cnt > 0) { lst[tid] = lst[(cnt - 1) * BD + tid]; cnt = 1; }
                }
                if (isnew) {
                    curpos = candp;
                    lst[cnt * BD + tid] = (uint16_t)(((uint32_t)candp << 8) | (uint32_t)i);
                    ++cnt;
                }
                const int p2 = (b + 1) * W + t;
                if (p2 < npos) {
                    const uint32_t k2 = mmer_key<M>(rowL, tid, row_words, p2);
                    if (k2 < pfx) { pfx = k2; pfxp = p2; }
                }
            }
        }
    }
    if (DENSE) {
        // the workgroup's block of the dense record array: exclusive scan of the threads' supermer counts, one atomic per workgroup
        uint32_t* wtot = reinterpret_cast<uint32_t*>(sfxp);        // (the suffix-minimum positions are dead: every wave is past its scan at the barrier)
        __syncthreads();
        const uint32_t incl = snk_wave_scan_incl((uint32_t)cnt);
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < BD / 64; ++w) { const uint32_t v = wtot[w]; wtot[w] = tot; tot += v; }
            const unsigned long long b = tot ? atomicAdd(a.dense_cursor, (unsigned long long)tot) : 0ull;
            wtot[BD / 64] = (uint32_t)b; wtot[BD / 64 + 1] = (uint32_t)(b >> 32);
        }
        __syncthreads();
        const uint64_t wgb = ((uint64_t)wtot[BD / 64 + 1] << 32) | wtot[BD / 64];
        // (positions are 32-bit inside the kernel: the host never launches the dense path on more than 2^32 records)
        dense_base = (uint32_t)wgb + wtot[tid >> 6] + incl - (uint32_t)cnt;
    }
    flush(cnt, nk - 1, true);

}

// k-mer instances and reads that contribute any (exact sizing of the single pass)
__global__ void __launch_bounds__(256) snk_msp_plan_kernel(const uint16_t* __restrict__ good_len, uint64_t n_reads, uint32_t K,
                                                           unsigned long long* __restrict__ out /* [0] instances, [1] live reads */) {
    unsigned long long inst = 0, live = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    // eight lengths per 16-byte load (the array is allocated with slack; the tail is handled one by one)
    const uint64_t n8 = (((uintptr_t)good_len & 15u) == 0) ? n_reads / 8 : 0;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n8; q += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(good_len)[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t g0 = w[j] & 0xFFFFu, g1 = w[j] >> 16;
            if (g0 >= K + 1) { inst += g0 - K + 1; ++live; }
            if (g1 >= K + 1) { inst += g1 - K + 1; ++live; }
        }
    }
    for (uint64_t r = n8 * 8 + (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += stride) {
        const uint32_t g = good_len[r];
        if (g >= K + 1) { inst += g - K + 1; ++live; }
    }
    for (int off = 32; off > 0; off >>= 1) { inst += __shfl_xor(inst, off); live += __shfl_xor(live, off); }
    if ((threadIdx.x & 63) == 0 && live) { atomicAdd(&out[0], inst); atomicAdd(&out[1], live); }
}

}  // namespace

size_t snk_msp_lds_bytes(uint32_t K, uint32_t M, uint32_t row_words) {
    return (size_t)(row_words + 1) * BD * 4 + (size_t)LCAP * BD * 2 + (size_t)(K - M + 1) * BD + SNK_MSP_HOT_TAB * 4 + 64;
}

template <int K, int M, bool TRIM, bool DENSE>
static int launch_msp_ktd(hipStream_t st, const snk_msp_args& a, char* err, size_t errcap) {
    size_t lds = snk_msp_lds_bytes(K, M, a.row_words);
    unsigned nb = (unsigned)((a.n_reads + BD - 1) / BD);
    SNK_HIP_TRY(hipFuncSetAttribute((const void*)snk_msp_kernel<K, M, TRIM, DENSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((snk_msp_kernel<K, M, TRIM, DENSE>), dim3(nb), dim3(BD), lds, st, a);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
template <int K, int M, bool TRIM>
static int launch_msp_kt(hipStream_t st, const snk_msp_args& a, char* err, size_t errcap) {
    if (a.b_hi) {
        if (a.dense_bkt) return snk_fail(SNK_E_ARG, err, errcap, "partition: bucket-range passes take the slot layout, not the dense one");
        size_t lds = snk_msp_lds_bytes(K, M, a.row_words);
        unsigned nb = (unsigned)((a.n_reads + BD - 1) / BD);
        SNK_HIP_TRY(hipFuncSetAttribute((const void*)snk_msp_kernel<K, M, TRIM, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((snk_msp_kernel<K, M, TRIM, false, true>), dim3(nb), dim3(BD), lds, st, a);
        SNK_HIP_TRY(hipGetLastError());
        return SNK_OK;
    }
    if (a.dense_bkt) {
        if (!a.dense_cursor || a.dense_cap >= (1ull << 32)) return snk_fail(SNK_E_ARG, err, errcap, "dense partition: needs its cursor and fewer than 2^32 record positions");
        return launch_msp_ktd<K, M, TRIM, true>(st, a, err, errcap);
    }
    return launch_msp_ktd<K, M, TRIM, false>(st, a, err, errcap);
}
template <int K, int M>
static int launch_msp_k(hipStream_t st, const snk_msp_args& a, char* err, size_t errcap) {
    if (a.quals) {
        if (!a.good_out || !a.plan || (a.qstride & 3u) || (((uintptr_t)a.quals) & 3u) || a.read_len > 160)
            return snk_fail(SNK_E_ARG, err, errcap, "fused trim: needs good_out, plan, 4-byte aligned quality rows and read_len <= 160");
        return launch_msp_kt<K, M, true>(st, a, err, errcap);
    }
    return launch_msp_kt<K, M, false>(st, a, err, errcap);
}

int snk_launch_msp(uint32_t K, uint32_t mlen, hipStream_t st, const snk_msp_args& a, char* err, size_t errcap) {
    if (a.n_reads == 0) return SNK_OK;
    if (a.row_words > 16) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "reads longer than 256 bases are not supported (row_words=%u)", a.row_words);
    if (mlen != SNK_M_LONG && mlen != (uint32_t)SNK_M_OF(K)) return snk_fail(SNK_E_INTERNAL, err, errcap, "partition: minimiser length %u", mlen);
    if (K == 48) return mlen == SNK_M_LONG ? launch_msp_k<48, SNK_M_LONG>(st, a, err, errcap) : launch_msp_k<48, SNK_M_OF(48)>(st, a, err, errcap);
    if (K == 60) return mlen == SNK_M_LONG ? launch_msp_k<60, SNK_M_LONG>(st, a, err, errcap) : launch_msp_k<60, SNK_M_OF(60)>(st, a, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}

int snk_launch_msp_plan(hipStream_t st, const uint16_t* good_len, uint64_t n_reads, uint32_t K, unsigned long long* out2,
                        char* err, size_t errcap) {
    SNK_HIP_TRY(hipMemsetAsync(out2, 0, 16, st));
    if (n_reads) {
        unsigned g = (unsigned)((n_reads / 8 + 255) / 256 + 1);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(snk_msp_plan_kernel, dim3(g), dim3(256), 0, st, good_len, n_reads, K, out2);
    }
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
