// snk_hot.hip -- hot minimiser buckets: a bucket that holds far more supermers than its capacity is re-partitioned by K-MER HASH.
//
// A minimiser bucket is counted by ONE workgroup in an LDS table of ~1200 distinct k-mers; a bucket with more is counted in hash-split
// sub-passes, each of which reads ALL records of the bucket.  That is fine for the odd bucket that is twice too large -- and quadratic
// for a minimiser site that thousands of loci share: an interspersed repeat family (10^4 copies of a 300-bp element at 1-3 %
// divergence) puts ~4 x 10^5 supermers with ~2 x 10^5 distinct k-mers behind ONE minimiser, i.e. hundreds of sub-passes over
// hundreds of thousands of records, serial in one workgroup (round 4, bench.py config.robust: 48 s on a step that takes 0.1 s).
// Real genomes are made of such families; the reference's sort-based reduce does not care (MapReduceEngine.h:574-584 sorts k-mer
// records, a k-mer's multiplicity changes nothing), so neither may this.
//
// What is done about it: the records of a hot bucket are expanded into single-k-mer records (the same 32-byte layout with n_kmers = 1:
// the k-mer, its two flank bases, the barcode word), each of which goes to the VIRTUAL bucket (hot bucket, class), class = the low
// bits of the count kernel's own split hash of the canonical k-mer -- all instances of a k-mer meet in one class, the classes of a
// bucket are counted by different workgroups in parallel, and a class holds about what a normal bucket holds.  The count kernel runs
// a second launch over the virtual buckets; a pass over virtual bucket (b, lg, id) starts where a hash-split sub-pass (lg, id) of
// bucket b would (further splits add hash bits above lg) and reports its chunk under the real bucket, so the bucket-local graph stage
// sees nothing new.  Exact sizing (count pass, scan, scatter pass): nothing is guessed, nothing can overflow.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"
#include "snk_stages.h"

namespace {

constexpr int HT = 256;               // threads per workgroup of the expansion kernels = records staged per workgroup

struct seg_tab {                      // a bucket's records: segment s holds [beg[s * stride + b], end[s * stride + b]) of the record array
    const uint64_t* beg;
    const uint64_t* end;
    uint32_t stride, nseg;
};
struct hot_tab {
    const uint32_t* bucket;           // [n_hot]
    const uint32_t* lg;               // [n_hot] classes = 1 << lg
    const uint64_t* rbase;            // [n_hot + 1] records of the hot buckets before this one
    const uint32_t* vbase;            // [n_hot + 1] virtual buckets before this one
    uint32_t n_hot;
};

// ---- which buckets are hot
__global__ void __launch_bounds__(256) hot_scan_kernel(seg_tab sg, uint32_t NB, uint32_t thresh, uint32_t hot_cap, uint32_t* __restrict__ hot_b,
                                                       uint32_t* __restrict__ hot_r, unsigned long long* __restrict__ ctr) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= NB) return;
    uint64_t r = 0;
    for (uint32_t q = 0; q < sg.nseg; ++q) r += sg.end[(uint64_t)q * sg.stride + b] - sg.beg[(uint64_t)q * sg.stride + b];
    if (r >= thresh) {
        const unsigned long long i = atomicAdd(&ctr[0], 1ull);
        if (i < hot_cap) { hot_b[i] = b; hot_r[i] = (uint32_t)(r > 0xFFFFFFFFull ? 0xFFFFFFFFull : r); }
    }
}
// one thread: class counts and the two running sums (a few thousand hot buckets at most)
__global__ void hot_plan_kernel(const uint32_t* __restrict__ hot_r, uint32_t hot_cap, uint32_t inst_per_class, uint32_t* __restrict__ lg, uint64_t* __restrict__ rbase,
                                uint32_t* __restrict__ vbase, unsigned long long* __restrict__ ctr) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long n = ctr[0];
    if (n > hot_cap) n = hot_cap;
    unsigned long long racc = 0, vacc = 0;
    for (uint32_t i = 0; i < (uint32_t)n; ++i) {
        // a record holds up to K - M + 1 k-mers, ~16 on average: classes of ~inst_per_class instances
        const unsigned long long inst = (unsigned long long)hot_r[i] * 16ull;
        uint32_t l = 1;
        while (l < 12 && (inst >> l) > inst_per_class) ++l;
        lg[i] = l;
        rbase[i] = racc; racc += hot_r[i];
        vbase[i] = (uint32_t)vacc; vacc += 1ull << l;
    }
    rbase[n] = racc;
    vbase[n] = (uint32_t)vacc;
    ctr[1] = racc;
    ctr[2] = vacc;
}
// the hot buckets leave the main launch; every virtual bucket learns who it is
__global__ void __launch_bounds__(256) hot_meta_kernel(hot_tab h, uint2* __restrict__ vmeta) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    const uint32_t NBv = h.vbase[h.n_hot];
    if (v >= NBv) return;
    uint32_t lo = 0, hi = h.n_hot;                    // largest i with vbase[i] <= v
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (h.vbase[mid] <= v) lo = mid; else hi = mid; }
    const uint32_t cls = v - h.vbase[lo];
    vmeta[v] = make_uint2(h.bucket[lo], (h.lg[lo] << 24) | cls);
}
// the hot buckets leave the main launch (empty in every segment); their bounds stay in saved[(2 s) * n_hot + i] / [(2 s + 1) * n_hot + i]
__global__ void __launch_bounds__(256) hot_mask_kernel(hot_tab h, const uint64_t* __restrict__ sbeg, uint64_t* __restrict__ send, uint32_t stride, uint32_t nseg,
                                                       uint64_t* __restrict__ saved) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h.n_hot) return;
    const uint32_t b = h.bucket[i];
    for (uint32_t q = 0; q < nseg; ++q) {
        const uint64_t x = sbeg[(uint64_t)q * stride + b], y = send[(uint64_t)q * stride + b];
        saved[(2ull * q) * h.n_hot + i] = x;
        saved[(2ull * q + 1) * h.n_hot + i] = y;
        send[(uint64_t)q * stride + b] = x;
    }
}

// ---- expansion: one thread per hot record, its k-mers one after the other.  SCATTER = false: count the instances per virtual bucket;
// true: write the single-k-mer records behind the virtual buckets' cursors.
// The instances of a workgroup are counted in LDS first (a workgroup's 256 records are -- but for a boundary -- records of ONE hot bucket,
// so an LDS counter per class of that bucket does): ONE global atomic per class and workgroup instead of one per instance.  The buckets
// this is for are the ones where instances repeat: a homopolymer's bucket holds a handful of distinct k-mers, and one global counter
// took every instance of the bucket, one at a time (same-address atomics queue, ~10 ns each).
constexpr int HOT_LG_MAX = 12;
template <int K, bool GROUPED, bool SCATTER>
__global__ void __launch_bounds__(HT) hot_expand_kernel(hot_tab h, const uint4* __restrict__ records, const uint64_t* __restrict__ saved, uint32_t nseg,
                                                        uint32_t* __restrict__ vcount, const uint64_t* __restrict__ voff, uint4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t rec[(HT + 1) * 9];        // staged records, nine words each (the ninth is zero: reads behind word 7), one zero record in front
    __shared__ uint32_t hist[1 << HOT_LG_MAX];                                  // instances of this workgroup per class of its first bucket; then the running slot inside the class
    __shared__ uint32_t cbase[SCATTER ? (1 << HOT_LG_MAX) : 1];                 // SCATTER: the workgroup's first slot in every class
    __shared__ uint32_t i0_s;
    const int tid = threadIdx.x;
    const uint64_t t = (uint64_t)blockIdx.x * HT + tid;
    const uint64_t total = h.rbase[h.n_hot];
    const bool live = t < total;
    uint32_t i = 0;
    uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
    if (live) {
        uint32_t lo = 0, hi = h.n_hot;                // largest i with rbase[i] <= t
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (h.rbase[mid] <= t) lo = mid; else hi = mid; }
        i = lo;
        uint64_t r = t - h.rbase[i];
        // the bucket's records: its segments one after the other (saved: the bounds before the bucket was masked out).  An index is an
        // offset from `records` that may wrap (the sharded step keeps a rank's own records outside its receive buffer): address arithmetic
        // on integers
        uint64_t at = 0;
        for (uint32_t q = 0; q < nseg; ++q) {
            const uint64_t x = saved[(2ull * q) * h.n_hot + i], n = saved[(2ull * q + 1) * h.n_hot + i] - x;
            if (r < n) { at = x + r; break; }
            r -= n;
        }
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<uintptr_t>(records) + at * 32ull);
        ra = src[0];
        rb = src[1];
    }
    uint32_t* my = rec + (tid + 1) * 9;
    my[0] = ra.x; my[1] = ra.y; my[2] = ra.z; my[3] = ra.w; my[4] = rb.x; my[5] = rb.y; my[6] = rb.z; my[7] = rb.w; my[8] = 0u;
    if (tid < 9) rec[tid] = 0u;
    if (tid == 0) i0_s = i;                            // thread 0 is live whenever the workgroup has a record
    __syncthreads();
    const uint32_t i0 = i0_s;
    const uint32_t ncls0 = 1u << h.lg[i0], vb00 = h.vbase[i0];
    for (uint32_t c = tid; c < ncls0; c += HT) hist[c] = 0u;
    __syncthreads();
    const uint32_t m6 = rb.z, w7 = rb.w;
    const uint32_t n_i = live ? (m6 & 0x7Fu) : 0u, hasL = (m6 >> 7) & 1u, hasR = (m6 >> 8) & 1u;
    const uint32_t lg = h.lg[live ? i : i0], vb0 = h.vbase[live ? i : i0];
    const bool mine = i == i0;                         // counted in LDS
    auto funnel = [](uint32_t hi_, uint32_t lo_, uint32_t s) { return (uint32_t)(((((uint64_t)hi_ << 32) | lo_) << s) >> 32); };
    // class of k-mer j of this thread's record (the extraction of snk_count.hip's insert phase: words wi .. wi+4 of the record)
    auto class_of = [&](uint32_t j) -> uint32_t {
        const uint32_t o = hasL + j;
        const uint32_t wi = o >> 4, sh = (2u * o) & 31u;
        const uint32_t* wp = my + wi;
        const uint32_t W0 = wp[0], W1 = wp[1], W2 = wp[2], W3 = wp[3];
        const uint32_t F0 = funnel(W0, W1, sh), F1 = funnel(W1, W2, sh), F2 = funnel(W2, W3, sh);
        uint32_t F3 = 0;
        if (K == 60) { const uint32_t W4 = wp[4]; F3 = funnel(W3, W4, sh) & 0xFFFFFF00u; }
        snk_kmer f;
        f.hi = ((uint64_t)F0 << 32) | F1;
        f.lo = ((uint64_t)F2 << 32) | F3;
        if (K == 48) f.lo &= 0xFFFFFFFF00000000ull;
        const snk_kmer r = snk_kmer_rc<K>(f);
        snk_kmer c = snk_kmer_lt(r, f) ? r : f;
        if (GROUPED) c.lo |= (uint64_t)w7;
        uint32_t h1, h2;
        snk_kmer_hash_count<(K > 48) || GROUPED>(c, &h1, &h2);
        return h2 & ((1u << lg) - 1u);
    };
    for (uint32_t j = 0; j < n_i; ++j) {
        const uint32_t c = class_of(j);
        if (mine) atomicAdd(&hist[c], 1u);
        else if (!SCATTER) atomicAdd(&vcount[vb0 + c], 1u);
    }
    __syncthreads();
    for (uint32_t c = tid; c < ncls0; c += HT) {
        const uint32_t n = hist[c];
        if (n) {
            const uint32_t first = atomicAdd(&vcount[vb00 + c], n);
            if (SCATTER) { cbase[c] = first; hist[c] = 0u; }
        }
    }
    if (!SCATTER) return;
    __syncthreads();
    for (uint32_t j = 0; j < n_i; ++j) {
        const uint32_t c = class_of(j);
        const uint32_t slot = mine ? cbase[c] + atomicAdd(&hist[c], 1u) : atomicAdd(&vcount[vb0 + c], 1u);
        const uint64_t dst = voff[vb0 + c] + slot;
        // the single-k-mer record: bases o - hasL' .. o + K - 1 + hasR' of the supermer's base stream
        const uint32_t o = hasL + j;
        const uint32_t nhasL = o > 0 ? 1u : 0u, nhasR = (j + 1u < n_i) ? 1u : hasR;
        const uint32_t a0 = o - nhasL, bits = 2u * ((uint32_t)K + nhasL + nhasR);
        const uint32_t wj = a0 >> 4, fs = (2u * a0) & 31u;
        const uint32_t* wq = my + wj;
        uint32_t nw[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const uint32_t x = funnel(wq[q], (wj + q + 1 <= 8) ? wq[q + 1] : 0u, fs);
            int rbits = (int)bits - 32 * q;
            rbits = rbits < 0 ? 0 : (rbits > 32 ? 32 : rbits);
            nw[q] = x & (uint32_t)(0xFFFFFFFF00000000ull >> rbits);
        }
        // (word 6 of the source record carries flag bits in its low nine bits: they are never inside the K + 2 bases taken here -- a
        // supermer's bases end above them -- but the funnel shift may drag them into a word that the mask above cuts to nothing)
        out[2 * dst] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
        out[2 * dst + 1] = make_uint4(nw[4], 0u, 1u | (nhasL << 7) | (nhasR << 8), w7);
    }
}

__global__ void __launch_bounds__(256) hot_seg_kernel(const uint64_t* __restrict__ voff, uint32_t NBv, uint64_t* __restrict__ seg) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= NBv) return;
    seg[v] = voff[v];
    seg[(uint64_t)NBv + v] = voff[v + 1];
}

template <typename T>
int alloc(snk_ctx* ctx, size_t n, T** out, char* err, size_t errcap) {
    void* q = nullptr;
    int rc = snk_ctx_alloc(ctx, (n ? n : 1) * sizeof(T) + 64, &q, err, errcap);
    *out = (T*)q;
    return rc;
}

template <int K, bool GROUPED>
int expand(hipStream_t st, const hot_tab& h, uint64_t total_records, const uint4* records, const uint64_t* saved, uint32_t nseg, uint32_t* vcount,
           const uint64_t* voff, uint4* out, bool scatter, char* err, size_t errcap) {
    const unsigned grid = (unsigned)((total_records + HT - 1) / HT);
    if (!scatter) hipLaunchKernelGGL((hot_expand_kernel<K, GROUPED, false>), dim3(grid), dim3(HT), 0, st, h, records, saved, nseg, vcount, voff, out);
    else hipLaunchKernelGGL((hot_expand_kernel<K, GROUPED, true>), dim3(grid), dim3(HT), 0, st, h, records, saved, nseg, vcount, voff, out);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

}  // namespace

// Two steps: the PLAN needs the buckets' sizes only (which buckets are hot, their classes, the virtual buckets; the hot buckets leave
// the main launch's segment table), the EXPANSION reads their records.  On one rank they follow each other; a rank of the N-GPU job
// plans from the exchanged histograms and expands when the records have arrived (snk_shard_step.hip).
struct hot_plan_state {
    hot_tab h;
    uint64_t* saved;
    uint32_t nseg, K;
    bool grouped;
    uint64_t n_rec;
    uint32_t *vcount;
    uint64_t *voff, *vseg;
};
static int hot_expand_run(snk_ctx* ctx, hipStream_t st, const void* records, snk_hot* hot, char* err, size_t errcap) {
    hot_plan_state* P = static_cast<hot_plan_state*>(hot->plan);
    const uint32_t NBv = hot->NBv;
    int rc;
    auto run = [&](bool scatter, uint4* out) -> int {
        if (P->grouped) return expand<48, true>(st, P->h, P->n_rec, (const uint4*)records, P->saved, P->nseg, P->vcount, P->voff, out, scatter, err, errcap);
        if (P->K == 48) return expand<48, false>(st, P->h, P->n_rec, (const uint4*)records, P->saved, P->nseg, P->vcount, P->voff, out, scatter, err, errcap);
        return expand<60, false>(st, P->h, P->n_rec, (const uint4*)records, P->saved, P->nseg, P->vcount, P->voff, out, scatter, err, errcap);
    };
    SNK_HIP_TRY(hipMemsetAsync(P->vcount, 0, ((size_t)NBv + 1) * 4, st));
    if ((rc = run(false, nullptr))) return rc;
    {
        auto in = rocprim::make_transform_iterator(P->vcount, [] __device__(uint32_t v) { return (uint64_t)v; });
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, in, P->voff, (uint64_t)0, (size_t)NBv + 1, rocprim::plus<uint64_t>(), st));
        void* tmp;
        if ((rc = snk_ctx_alloc(ctx, tb + 64, &tmp, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, in, P->voff, (uint64_t)0, (size_t)NBv + 1, rocprim::plus<uint64_t>(), st));
    }
    uint64_t n_inst = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&n_inst, P->voff + NBv, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    uint4* vrec;
    if ((rc = alloc(ctx, 2 * (size_t)n_inst + 2, &vrec, err, errcap))) return rc;
    SNK_HIP_TRY(hipMemsetAsync(P->vcount, 0, ((size_t)NBv + 1) * 4, st));
    if ((rc = run(true, vrec))) return rc;
    hipLaunchKernelGGL(hot_seg_kernel, dim3((NBv + 255) / 256), dim3(256), 0, st, P->voff, NBv, P->vseg);
    SNK_HIP_TRY(hipGetLastError());
    hot->n_instances = n_inst;
    hot->records = vrec;
    hot->seg = P->vseg;
    return SNK_OK;
}
int snk_stage_hot_expand(snk_ctx* ctx, hipStream_t st, const void* records, snk_hot* hot, char* err, size_t errcap) {
    if (!hot || hot->NBv == 0 || hot->records) return SNK_OK;
    int rc = hot_expand_run(ctx, st, records, hot, err, errcap);
    delete static_cast<hot_plan_state*>(hot->plan);
    hot->plan = nullptr;
    return rc;
}
void snk_stage_hot_drop(snk_hot* hot) {
    if (hot && hot->plan) { delete static_cast<hot_plan_state*>(hot->plan); hot->plan = nullptr; }
}

int snk_stage_hot_plan(snk_ctx* ctx, hipStream_t st, uint32_t K, bool grouped, uint64_t* seg_beg, uint64_t* seg_end, uint32_t stride, uint32_t nseg, uint32_t NB,
                       uint32_t cap, snk_hot* hot, char* err, size_t errcap) {
    memset(hot, 0, sizeof *hot);
    if (NB == 0 || nseg == 0) return SNK_OK;
    // hot = more records than eight times the slots of a bucket (and never fewer than SNK_HOT_MIN: a normal hash-split pass or two
    // over a few thousand records is cheaper than the expansion)
    const uint32_t floor_ = snk_opt_u32("hot_min", 8192);
    uint64_t thr = (uint64_t)cap * snk_opt_u32("hot_factor", 8);
    if (thr < floor_) thr = floor_;
    if (thr > 0xFFFFFFFFull || snk_opt_u32("hot", 1) == 0) return SNK_OK;
    const uint32_t hot_cap = 1u << 16;
    int rc;
    uint32_t *hot_b, *hot_r, *lg, *vbase;
    uint64_t* rbase;
    unsigned long long* ctr;
    if ((rc = alloc(ctx, hot_cap, &hot_b, err, errcap)) || (rc = alloc(ctx, hot_cap, &hot_r, err, errcap)) || (rc = alloc(ctx, hot_cap + 1, &lg, err, errcap)) ||
        (rc = alloc(ctx, hot_cap + 1, &vbase, err, errcap)) || (rc = alloc(ctx, hot_cap + 1, &rbase, err, errcap)) || (rc = alloc(ctx, 8, &ctr, err, errcap)))
        return rc;
    SNK_HIP_TRY(hipMemsetAsync(ctr, 0, 64, st));
    const seg_tab sg{seg_beg, seg_end, stride, nseg};
    hipLaunchKernelGGL(hot_scan_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, sg, NB, (uint32_t)thr, hot_cap, hot_b, hot_r, ctr);
    hipLaunchKernelGGL(hot_plan_kernel, dim3(1), dim3(64), 0, st, hot_r, hot_cap, snk_opt_u32("hot_class_inst", 6000), lg, rbase, vbase, ctr);
    unsigned long long h_ctr[3] = {0, 0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h_ctr, ctr, 24, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if (h_ctr[0] == 0) return SNK_OK;
    if (h_ctr[0] > hot_cap) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than %u hot minimiser buckets", hot_cap);
    const uint32_t n_hot = (uint32_t)h_ctr[0], NBv = (uint32_t)h_ctr[2];
    hot_plan_state* P = new hot_plan_state();
    P->h = hot_tab{hot_b, lg, rbase, vbase, n_hot};
    P->nseg = nseg; P->K = K; P->grouped = grouped; P->n_rec = h_ctr[1];
    hot->plan = P;
    // the main segment table shows the hot buckets empty; their bounds are kept for the expansion
    uint2* vmeta;
    if ((rc = alloc(ctx, 2ull * nseg * n_hot, &P->saved, err, errcap)) || (rc = alloc(ctx, NBv, &vmeta, err, errcap)) || (rc = alloc(ctx, (size_t)NBv + 1, &P->vcount, err, errcap)) ||
        (rc = alloc(ctx, (size_t)NBv + 2, &P->voff, err, errcap)) || (rc = alloc(ctx, 2ull * NBv, &P->vseg, err, errcap))) {
        snk_stage_hot_drop(hot);
        return rc;
    }
    hipLaunchKernelGGL(hot_mask_kernel, dim3((n_hot + 255) / 256), dim3(256), 0, st, P->h, seg_beg, seg_end, stride, nseg, P->saved);
    hipLaunchKernelGGL(hot_meta_kernel, dim3((NBv + 255) / 256), dim3(256), 0, st, P->h, vmeta);
    SNK_HIP_TRY(hipGetLastError());
    hot->n_hot = n_hot;
    hot->NBv = NBv;
    hot->n_records = P->n_rec;
    hot->vmeta = vmeta;
    return SNK_OK;
}

// one rank, records resident: plan and expand
int snk_stage_hot(snk_ctx* ctx, hipStream_t st, uint32_t K, bool grouped, snk_partition* part, snk_hot* hot, char* err, size_t errcap) {
    memset(hot, 0, sizeof *hot);
    if (part->gidx || part->NB == 0 || part->n_overflow == 0) return SNK_OK;        // a bucket far above its capacity has records on the overflow list
    const uint32_t NB = part->NB;
    int rc = snk_stage_hot_plan(ctx, st, K, grouped, part->seg, part->seg + NB, 2 * NB, 2, NB, part->cap, hot, err, errcap);
    if (rc || hot->NBv == 0) return rc;
    return snk_stage_hot_expand(ctx, st, part->records, hot, err, errcap);
}
