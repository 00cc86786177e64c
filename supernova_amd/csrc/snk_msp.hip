// snk_msp.hip -- K3/K4: minimiser scan of every trimmed read, supermer emission into minimiser buckets.
//
// What it replaces:
//   msp::simple_scan            lib/tada/src/msp/mod.rs:60-134   (sliding-window minimiser, slices)
//   Bsp::new / Exts::from_slice lib/tada/src/kmer/mod.rs:367-378, kmer/exts.rs:71-84 (supermer record + flanks)
//   shardio write path          lib/tada/external/rust-shardio/src/shard.rs:184-211 (group by shard)
//   and, for path B, the hash->(pass,bin) map of MapReduceEngine.h:315-326.
// Shard assignment is internal to the reference (App. A.10): counts and unitigs do not depend on it.
// This build uses M=16-mers ordered by a 32-bit hash of the canonical M-mer (strand symmetric, so a
// k-mer and its reverse complement always land in the same bucket -- the invariant of
// check_consistent_shard, lib/tada/src/kmer/mod.rs:1102-1150).
//
// Layout: one thread per read, 256 reads per workgroup.  The packed rows of the workgroup are staged
// in LDS with one coalesced sweep; every thread then owns one LDS column ([word][thread], conflict
// free).  The window minimum is computed without data-dependent control flow by the block
// decomposition (suffix minima of window-sized blocks in an LDS column, running prefix minimum in a
// register), so a wave never serialises on "rescan on expiry".  Supermer starts are appended to a
// short per-thread LDS list and emitted in a second, short loop (histogram pass: one atomic per
// supermer; scatter pass: slot reservation + one 32-byte record).
//
// Supermer record (32 B, two 16-byte stores), words MSB-first like a read row:
//   bits [0, 2*n_ext)      the supermer bases including one flanking base on each side when the read
//                          has one inside its good length (n_ext <= 2K-M+2)
//   word 6 bits 0..11      n_kmers (7 bits) | hasL << 7 | hasR << 8
//   word 7                 barcode (int32; -1 = ignore-rule read, BuildReadQGraph48.cc:158-159)
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
    uint32_t w1 = (wi + 1 < row_words) ? rowL[(wi + 1) * BD + tid] : 0u;
    uint32_t s = 2u * ((uint32_t)p & 15u);
    uint32_t x = s ? ((w0 << s) | (w1 >> (32u - s))) : w0;   // 16 bases starting at p, MSB first
    uint32_t rx = snk_rev2_32(~x);                             // reverse complement of those 16 bases
    uint32_t code, rcode;
    if (M == 16) { code = x; rcode = rx; }
    else { code = x >> (32 - 2 * M); rcode = rx & ((1u << (2 * M)) - 1u); }
    return snk_minimizer_key(code, rcode);
}

// Bucket of the supermer whose minimiser sits at position p.  Two rules make this safe:
//  * it must NOT be taken from the ordering key directly: minimisers are window minima of that key, so
//    their keys crowd near zero (a Beta(1,w) law) and most supermers would fall into the lowest few
//    percent of the buckets -- the key is re-mixed first;
//  * it must be a function of the ordering key ONLY (not of the M-mer or its position): when two
//    different M-mers of a window tie on the key, the two strands may pick different ones, and a k-mer
//    and its reverse complement must still meet in one bucket.
template <int M>
__device__ __forceinline__ uint32_t mmer_bucket(const uint32_t* rowL, int tid, uint32_t row_words, int p, uint32_t NB) {
    return snk_bucket_of_key(mmer_key<M>(rowL, tid, row_words, p), NB);
}

// extract 32 bits starting at base `a + 16*j` of the row column
__device__ __forceinline__ uint32_t row_window(const uint32_t* rowL, int tid, uint32_t row_words, uint32_t a, uint32_t j) {
    uint32_t wi = (a >> 4) + j;
    uint32_t w0 = wi < row_words ? rowL[wi * BD + tid] : 0u;
    uint32_t w1 = wi + 1 < row_words ? rowL[(wi + 1) * BD + tid] : 0u;
    uint32_t s = 2u * (a & 15u);
    return s ? ((w0 << s) | (w1 >> (32u - s))) : w0;
}

template <int K, int M, bool SCATTER>
__global__ void __launch_bounds__(BD) snk_msp_kernel(const uint32_t* __restrict__ rows, uint32_t row_words,
                                                     const uint16_t* __restrict__ good_len,
                                                     const int32_t* __restrict__ bc, int64_t ign_bc_below,
                                                     uint64_t read_index_base, uint64_t n_reads, uint32_t NB,
                                                     uint32_t* __restrict__ hist_or_cursor,
                                                     uint4* __restrict__ records,
                                                     unsigned long long* __restrict__ n_inst_out,
                                                     uint16_t* __restrict__ slist, uint8_t* __restrict__ scount) {
    constexpr int W = K - M + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* rowL = smem;                              // [row_words][BD]
    // Only the POSITION of every suffix minimum is kept in LDS (1 byte per entry); its key is recomputed when the
    // position changes (a few times per block).  Dropping the 4-byte key column cuts the LDS footprint from 59 KB
    // to 26 KB per workgroup -> 6 instead of 2 workgroups per CU; the kernel is latency bound (PMC: 22 % VALU
    // utilisation at 8 waves/CU), so occupancy is what pays.
    uint16_t* lst = reinterpret_cast<uint16_t*>(rowL + (size_t)row_words * BD);   // [LCAP][BD] minimiser position << 8 | first k-mer
    uint8_t* sfxp = reinterpret_cast<uint8_t*>(lst + (size_t)LCAP * BD);  // [W][BD] position of the suffix minimum
    const int tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * BD;
    // coalesced stage of the workgroup's rows
    {
        uint64_t nrows = n_reads - r0 < (uint64_t)BD ? n_reads - r0 : (uint64_t)BD;
        uint32_t total = (uint32_t)nrows * row_words;
        const uint32_t* src = rows + r0 * row_words;
        for (uint32_t idx = tid; idx < total; idx += BD) {
            uint32_t t = idx / row_words, w = idx - t * row_words;
            rowL[w * BD + t] = src[idx];
        }
    }
    __syncthreads();
    const uint64_t r = r0 + tid;
    int g = (r < n_reads) ? (int)good_len[r] : 0;
    if (g < K + 1) g = 0;                               // reads with fewer than 2 k-mers are skipped (:160)
    const int nk = g ? g - K + 1 : 0;
    const int npos = g ? g - M + 1 : 0;
    int32_t mybc = 0;
    if (SCATTER && g) mybc = (bc && (int64_t)(read_index_base + r) >= ign_bc_below) ? bc[r] : -1;

    int curpos = -1;              // minimiser position of the open supermer
    int cnt = 0;                  // closed+open supermer starts in the list

    // emit list entries [0, upto) ; entry e covers k-mers [start_e, start_{e+1}-1], the last one ends at last_end
    auto flush = [&](int upto, int last_end) {
        int maxn = upto;
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(maxn, off); maxn = o > maxn ? o : maxn; }
        for (int e = 0; e < maxn; ++e) {
            if (e < upto) {
                uint32_t ent = lst[e * BD + tid];
                uint32_t s = ent & 0xFFu;
                uint32_t en = (e + 1 < upto) ? (((uint32_t)lst[(e + 1) * BD + tid] & 0xFFu) - 1u) : (uint32_t)last_end;
                uint32_t bucket = mmer_bucket<M>(rowL, tid, row_words, (int)(ent >> 8), NB);
                if (!SCATTER) {
                    atomicAdd(&hist_or_cursor[bucket], 1u);
                } else {
                    uint32_t slot = atomicAdd(&hist_or_cursor[bucket], 1u);
                    uint32_t n_kmers = en - s + 1u;
                    uint32_t hasL = s > 0 ? 1u : 0u;
                    uint32_t hasR = (en + (uint32_t)K < (uint32_t)g) ? 1u : 0u;
                    uint32_t a = s - hasL;
                    uint32_t bits = 2u * (n_kmers + (uint32_t)K - 1u + hasL + hasR);
                    uint32_t w[8];
#pragma unroll
                    for (uint32_t j = 0; j < 7; ++j) {
                        uint32_t x = row_window(rowL, tid, row_words, a, j);
                        uint32_t lo = 32u * j;
                        if (bits <= lo) x = 0u;
                        else if (bits - lo < 32u) x &= ~(0xFFFFFFFFu >> (bits - lo));
                        w[j] = x;
                    }
                    w[6] |= n_kmers | (hasL << 7) | (hasR << 8);
                    w[7] = (uint32_t)mybc;
                    uint4* dst = records + (size_t)slot * 2;
                    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
                }
            }
        }
    };

    // The histogram pass saves every read's supermer list (minimiser position, first k-mer; <= LCAP entries,
    // [entry][read] so that a wave's accesses coalesce); the scatter pass replays it instead of scanning again.
    // A read whose list overflowed is marked 0xFF and its wave falls back to the scan.
    bool overflowed = false;
    bool replay = false;
    if (SCATTER && scount) {
        uint32_t sc = (r < n_reads) ? scount[r] : 0u;
        replay = !__any(sc == 0xFFu);
        if (replay) {
            cnt = (int)sc;
            for (int e = 0; e < cnt; ++e) lst[e * BD + tid] = slist[(uint64_t)e * n_reads + r];
        }
    }
    const int nblocks = replay ? 0 : (nk + W - 1) / W;
    int maxblocks = nblocks;
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(maxblocks, off); maxblocks = o > maxblocks ? o : maxblocks; }
    for (int b = 0; b < maxblocks; ++b) {
        // suffix minima of block b (positions b*W .. b*W+W-1), right to left; "<=" keeps the leftmost on ties
        uint32_t run = 0xFFFFFFFFu;
        int runp = 0;
        for (int t = W - 1; t >= 0; --t) {
            int p = b * W + t;
            if (p < npos) {
                uint32_t key = mmer_key<M>(rowL, tid, row_words, p);
                if (key <= run) { run = key; runp = p; }
            }
            sfxp[t * BD + tid] = (uint8_t)(run == 0xFFFFFFFFu ? 0xFF : runp);     // 0xFF: no valid position in this suffix
        }
        // k-mers of block b: window = suffix of block b from t  U  prefix of block b+1 of length t
        uint32_t pfx = 0xFFFFFFFFu;
        int pfxp = 0;
        uint32_t sv = 0xFFFFFFFFu;
        int svp = -1;
        for (int t = 0; t < W; ++t) {
            int i = b * W + t;
            const int sp = (int)sfxp[t * BD + tid];
            if (sp != svp) { svp = sp; sv = sp == 0xFF ? 0xFFFFFFFFu : mmer_key<M>(rowL, tid, row_words, sp); }
            int candp = (sv <= pfx) ? sp : pfxp;    // the suffix part lies left of the prefix part
            bool isnew = (i < nk) && (candp != curpos);
            if (__any(isnew && cnt == LCAP)) {     // some lane's list is full: every lane of the wave drains its list
                // the open supermer (last entry) stays; flush() has wave-wide shuffles, so it is called uniformly
                const int upto = cnt > 0 ? cnt - 1 : 0;
                const int last_end = cnt > 0 ? (int)(lst[(cnt - 1) * BD + tid] & 0xFFu) - 1 : 0;
                flush(upto, last_end);
                overflowed = true;
                if (cnt > 0) { lst[tid] = lst[(cnt - 1) * BD + tid]; cnt = 1; }
            }
            if (isnew) {
                curpos = candp;
                lst[cnt * BD + tid] = (uint16_t)(((uint32_t)candp << 8) | (uint32_t)i);   // minimiser position | first k-mer
                ++cnt;
            }
            int p2 = (b + 1) * W + t;
            if (p2 < npos) {
                uint32_t k2 = mmer_key<M>(rowL, tid, row_words, p2);
                if (k2 < pfx) { pfx = k2; pfxp = p2; }
            }
        }
    }
    if (!SCATTER && scount && r < n_reads) {
        scount[r] = overflowed ? (uint8_t)0xFF : (uint8_t)cnt;
        if (!overflowed) for (int e = 0; e < cnt; ++e) slist[(uint64_t)e * n_reads + r] = lst[e * BD + tid];
    }
    flush(cnt, nk - 1);

    if (!SCATTER) {
        unsigned long long v = (unsigned long long)nk;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0 && v) atomicAdd(n_inst_out, v);
    }
}

}  // namespace

size_t snk_msp_lds_bytes(uint32_t K, uint32_t M, uint32_t row_words) {
    return (size_t)row_words * BD * 4 + (size_t)LCAP * BD * 2 + (size_t)(K - M + 1) * BD + 64;
}

template <int K, int M>
static int launch_msp(bool scatter, hipStream_t st, const uint32_t* rows, uint32_t row_words, const uint16_t* good_len,
                      const int32_t* bc, int64_t ign_bc_below, uint64_t read_index_base, uint64_t n_reads, uint32_t NB,
                      uint32_t* hist_or_cursor, void* records, unsigned long long* n_inst, uint16_t* slist, uint8_t* scount,
                      char* err, size_t errcap) {
    size_t lds = snk_msp_lds_bytes(K, M, row_words);
    unsigned nb = (unsigned)((n_reads + BD - 1) / BD);
    if (scatter) {
        SNK_HIP_TRY(hipFuncSetAttribute((const void*)snk_msp_kernel<K, M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((snk_msp_kernel<K, M, true>), dim3(nb), dim3(BD), lds, st, rows, row_words, good_len, bc,
                           ign_bc_below, read_index_base, n_reads, NB, hist_or_cursor, (uint4*)records, n_inst, slist, scount);
    } else {
        SNK_HIP_TRY(hipFuncSetAttribute((const void*)snk_msp_kernel<K, M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((snk_msp_kernel<K, M, false>), dim3(nb), dim3(BD), lds, st, rows, row_words, good_len, bc,
                           ign_bc_below, read_index_base, n_reads, NB, hist_or_cursor, (uint4*)records, n_inst, slist, scount);
    }
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

int snk_launch_msp(uint32_t K, bool scatter, hipStream_t st, const uint32_t* rows, uint32_t row_words,
                   const uint16_t* good_len, const int32_t* bc, int64_t ign_bc_below, uint64_t read_index_base,
                   uint64_t n_reads, uint32_t NB, uint32_t* hist_or_cursor, void* records, unsigned long long* n_inst,
                   uint16_t* slist, uint8_t* scount, char* err, size_t errcap) {
    if (n_reads == 0) return SNK_OK;
    if (row_words > 16) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "reads longer than 256 bases are not supported (row_words=%u)", row_words);
    if (K == 48)
        return launch_msp<48, SNK_M>(scatter, st, rows, row_words, good_len, bc, ign_bc_below, read_index_base, n_reads, NB, hist_or_cursor, records, n_inst, slist, scount, err, errcap);
    if (K == 60)
        return launch_msp<60, SNK_M>(scatter, st, rows, row_words, good_len, bc, ign_bc_below, read_index_base, n_reads, NB, hist_or_cursor, records, n_inst, slist, scount, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}
