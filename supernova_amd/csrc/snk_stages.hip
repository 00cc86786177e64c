// snk_stages.hip -- reusable pipeline stages shared by the single-GPU and the sharded paths.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdlib.h>
#include <math.h>

#include <cmath>
#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_graph.h"
#include "snk_kernels.h"
#include "snk_stages.h"


// K5-K8 + gather + sort: supermer records of NB buckets (nseg segments) -> dense retained table sorted by key.
// status: device u32[16] scratch words.
int snk_stage_count_table(snk_ctx* ctx, hipStream_t st, uint32_t K, const void* records, const uint64_t* seg_beg,
                          const uint64_t* seg_end, uint32_t seg_stride, uint32_t nseg, uint32_t NB, uint32_t min_freq, uint32_t bc_mode, uint32_t grouped, uint64_t n_inst_hint,
                          uint32_t* status, bool want_sort, snk_table* out, char* err, size_t errcap, const snk_count_ranges* ranges, snk_count_pilot* pilot,
                          const uint32_t* gidx, bool defer_compact, const snk_hot* hot) {
    if (hot && hot->NBv == 0) hot = nullptr;
    defer_compact = defer_compact && !want_sort && snk_opt_u32("defer_compact", 1) != 0;
    int rc;
    snk_phase_timer tm(st), kt(st);
    tm.mark();
    // ---- K5-K8 count + filter into a region-partitioned table, then gather the regions densely.
    // Every retained k-mer has >= min_freq instances; deep coverage retains far fewer (56x: ~1/38 of them).
    uint32_t n_regions = 1;
    if ((rc = snk_count_regions(K, grouped, nseg, NB, bc_mode, &n_regions, err, errcap))) return rc;
    uint64_t est = n_inst_hint / (min_freq > 1 ? 12 : 1) + 4096;     // first call only; a wrong guess costs one re-run
    // A job in bucket-range passes is short of memory, and instances / 12 are 2 bytes per instance it may not have (deep coverage retains 1 in 38):
    // it starts from 1 in 32 and lets its first range say what the data retain (`range0_probe` below: one repeated range when that is more)
    const bool tight = ranges && ranges->n > 1 && ranges->replay && min_freq > 1 && ctx->plan_mem && (double)est * 24.0 > 0.10 * (double)ctx->plan_mem;
    if (tight) est = n_inst_hint / 32 + 4096;
    if (ctx->last_n_kmers && ctx->last_n_instances == n_inst_hint) est = ctx->last_n_kmers + ctx->last_n_kmers / (tight ? 8 : 2) + 4096;
    bool range0_probe = tight && !(ctx->last_n_kmers && ctx->last_n_instances == n_inst_hint);
    uint64_t region_cap = est / n_regions + 64;
    // (a sparse, clustered output -- per-barcode groups at min_freq 4: 155 survivors per region on average, several hundred in some -- overflowed
    // the average-based capacity in EVERY call and the count kernel ran twice, 127 instead of 64 ms: the fullest region of the last call counts too)
    if (ctx->last_n_kmers && ctx->last_n_instances == n_inst_hint && ctx->last_region_n == n_regions && ctx->last_region_max + ctx->last_region_max / 4 + 64 > region_cap)
        region_cap = ctx->last_region_max + ctx->last_region_max / 4 + 64;
    snk_u128 *keys_r = nullptr, *keys_a = nullptr, *keys_b = nullptr;
    uint64_t *vals_r = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    unsigned long long *rcur = nullptr, *roff = nullptr;
    uint64_t n_kmers = 0;
    uint32_t h_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_regions + 1) * 8ull, &q, err, errcap))) return rc; rcur = (unsigned long long*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_regions + 1) * 8ull, &q, err, errcap))) return rc; roff = (unsigned long long*)q;
    }
    std::vector<unsigned long long> h_rcur(n_regions);
    uint32_t *chunk_n = nullptr, *chunk_base = nullptr;
    uint4* extra = nullptr;
    uint32_t extra_cap = 0;
    if (!want_sort) {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc; chunk_n = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc; chunk_base = (uint32_t*)q;
        // sub-passes of split buckets: a few per cent of the buckets split once; remember what the last call needed
        extra_cap = NB / 8 > (1u << 16) ? NB / 8 : (1u << 16);
        if (ctx->last_extra > extra_cap) extra_cap = ctx->last_extra + ctx->last_extra / 4;
        if (hot) extra_cap += hot->NBv + hot->NBv / 2;          // every virtual bucket reports its chunk through this list
        if (ranges && ranges->hots) extra_cap += NB / 16 + (1u << 16);      // (bucket-range passes: the hot buckets are only known pass by pass; too small = one repeated run)
    }
    bool pilot_regrown = false;
    for (int attempt = 0; attempt < 4; ++attempt) {
        void* q;
        if (!want_sort) {
            if ((rc = snk_ctx_alloc(ctx, (size_t)extra_cap * 16 + 16, &q, err, errcap))) return rc; extra = (uint4*)q;
            SNK_HIP_TRY(hipMemsetAsync(chunk_n, 0, (NB + 1) * 4ull, st));
        }
        if ((rc = snk_ctx_alloc(ctx, region_cap * n_regions * 16, &q, err, errcap))) return rc; keys_r = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, region_cap * n_regions * 8, &q, err, errcap))) return rc; vals_r = (uint64_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(rcur, 0, (n_regions + 1) * 8ull, st));
        SNK_HIP_TRY(hipMemsetAsync(status, 0, 32, st));
        snk_count_args ca;
        ca.records = (const uint4*)records;
        ca.seg_beg = seg_beg;
        ca.seg_end = seg_end;
        ca.seg_stride = seg_stride;
        ca.nseg = nseg;
        ca.gidx = gidx;
        ca.vmeta = nullptr;
        ca.NB = NB;
        ca.min_freq = min_freq;
        ca.bc_mode = bc_mode;
        ca.grouped = grouped;
        ca.tight = ctx->count_tight;
        // (per-barcode groups, min_freq >= 3: three reads of one barcode over one k-mer are rare, two -- the mates of a pair -- are not; with
        // min_freq 2 a quarter of the instances pass the filter and the larger buckets cost more than they save: 428 against 270 ms.
        // SNK_COUNT_SCREEN: 0 never, 1 at min_freq >= 3, 2 at min_freq >= 2 as well)
        { const uint32_t sc = grouped ? snk_opt_u32("count_screen", 1) : 0u; ca.screen = (sc && min_freq >= (sc >= 2 ? 2u : 3u)) ? std::min(min_freq, 3u) : 0u; }
        if (!grouped && K == 48 && ctx->count_screen && ctx->count_tight && bc_mode <= 2u) ca.screen = std::min(std::min(min_freq, 3u), ctx->count_screen);
        ca.bucket0 = 0;
        ca.out_keys = keys_r;
        ca.out_vals = vals_r;
        ca.region_cap = region_cap;
        ca.n_regions = n_regions;
        ca.region_cursor = rcur;
        ca.status = status;
        ca.chunk_n = chunk_n;
        ca.chunk_base = chunk_base;
        ca.extra = extra;
        ca.extra_cap = extra_cap;
        ca.dbg = snk_opt_u32("count_dbg", 0);
        ca.prof = nullptr;
#ifdef SNK_COUNT_PROF
        {
            void* pq;
            if ((rc = snk_ctx_alloc(ctx, 64, &pq, err, errcap))) return rc;
            ca.prof = (unsigned long long*)pq;
            SNK_HIP_TRY(hipMemsetAsync(ca.prof, 0, 64, st));
        }
#endif
        kt.n = 0;
        kt.mark();
        bool pilot_regrow = false;
        // the pilot: 1/64 of the buckets, then a look at how full their tables ran
        auto run_pilot = [&](uint32_t b0, uint32_t NBp, bool* retarget, bool* regrow) -> int {
            snk_count_args cp = ca;
            cp.bucket0 = b0;
            cp.NB = b0 + NBp;
            int r2;
            if ((r2 = snk_launch_count(K, st, cp, err, errcap))) return r2;
            uint32_t h_p[8];
            SNK_HIP_TRY(hipMemcpyAsync(h_p, status, 32, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
            pilot->per_bucket = (double)h_p[5] * 16.0 / NBp;
            {
                // the pilot also says how many k-mers survive: a first call whose guess (instances / 12) is too small for this data
                // (min_freq 1 or 2, shallow coverage) learns it here instead of from a full count run whose regions overflowed
                std::vector<unsigned long long> h_rc(n_regions);
                SNK_HIP_TRY(hipMemcpy(h_rc.data(), rcur, n_regions * 8ull, hipMemcpyDeviceToHost));
                unsigned long long sum = 0;
                for (uint32_t r = 0; r < n_regions; ++r) sum += h_rc[r];
                const uint64_t need = (uint64_t)((double)sum * ((double)(NB - b0) / NBp) * 1.25 / n_regions) + 64;
                if (need > region_cap && b0 == 0 && !pilot_regrown) { region_cap = need; *regrow = true; return SNK_OK; }
                // ... and should the caller partition again (retarget), its next run sizes its regions from what survived here, not from the
                // blanket instances / 12: at 1.5 % errors that is 20 GB of regions mapped for 7 GB of survivors -- on a first call the arena
                // grows by what is asked for, at the driver's ~30 ms per GB
                if (b0 == 0 && sum && snk_opt_u32("pilot_est", 1)) { ctx->last_n_kmers = (uint64_t)((double)sum * ((double)NB / NBp)); ctx->last_n_instances = n_inst_hint; }
            }
            if (pilot->agree && (r2 = pilot->agree(pilot->user, &pilot->per_bucket))) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: the pilot's exchange failed (%d)", r2);
            *retarget = pilot->per_bucket > 0.85 * snk_count_limit(K, grouped, ctx->count_tight) && !h_p[1];
            return SNK_OK;
        };
        if (ranges && ranges->n && (attempt == 0 || ranges->replay)) {
            // the records of bucket range r may still be on their way: the caller's hook makes the stream wait for
            // them, the ranges before it are being counted meanwhile
            for (uint32_t r = 0; r < ranges->n; ++r) {
                if (ranges->ready && (rc = ranges->ready(ranges->user, r))) { if (ranges->replay) return rc; return snk_fail(SNK_E_ARG, err, errcap, "count: the range hook failed for range %u (%d)", r, rc); }
                snk_count_args cr = ca;
                cr.bucket0 = ranges->bounds[r];
                cr.NB = ranges->bounds[r + 1];
                if (r == 0 && attempt == 0 && pilot && NB >= 16384 && n_inst_hint) {
                    const uint32_t NBp = NB / 64 < cr.NB - cr.bucket0 ? NB / 64 : cr.NB - cr.bucket0;
                    bool retarget = false, regrow = false;
                    if ((rc = run_pilot(cr.bucket0, NBp, &retarget, &regrow))) return rc;
                    if (retarget) return SNK_RETARGET;
                    if (regrow) { pilot_regrow = true; break; }
                    cr.bucket0 += NBp;
                }
                if (cr.bucket0 < cr.NB && (rc = snk_launch_count(K, st, cr, err, errcap))) return rc;
                if (r == 0 && range0_probe) {
                    // every region has taken its share of range 0 (a workgroup's buckets are spread over all ranges): what the fullest holds
                    // now, times the ranges, is what it will hold -- 15 % on top; larger regions and range 0 once more if that is more than the guess
                    range0_probe = false;
                    std::vector<unsigned long long> h_rc(n_regions);
                    SNK_HIP_TRY(hipMemcpyAsync(h_rc.data(), rcur, n_regions * 8ull, hipMemcpyDeviceToHost, st));
                    SNK_HIP_TRY(snk_sync(st));
                    unsigned long long mx0 = 0;
                    for (uint32_t q2 = 0; q2 < n_regions; ++q2) mx0 = std::max(mx0, h_rc[q2]);
                    const double share = (double)(ranges->bounds[1] - ranges->bounds[0]) / (double)NB;
                    const uint64_t need = (uint64_t)((double)mx0 / share * 1.15) + 64;
                    if (need > region_cap) { region_cap = need; pilot_regrow = true; break; }
                }
            }
        } else if (attempt == 0 && pilot && NB >= 16384 && n_inst_hint) {
            const uint32_t NBp = NB / 64;
            bool retarget = false, regrow = false;
            if ((rc = run_pilot(0, NBp, &retarget, &regrow))) return rc;
            if (retarget) return SNK_RETARGET;
            if (regrow) pilot_regrow = true;
            else {
                snk_count_args cr = ca;
                cr.bucket0 = NBp;
                if ((rc = snk_launch_count(K, st, cr, err, errcap))) return rc;
            }
        } else if ((rc = snk_launch_count(K, st, ca, err, errcap))) return rc;
        kt.mark();
        if (ranges && ranges->hots && !pilot_regrow) {
            for (const snk_hot& h : *ranges->hots) {
                if (!h.NBv || !h.records) continue;
                snk_count_args cv = ca;
                cv.records = (const uint4*)h.records; cv.seg_beg = h.seg; cv.seg_end = h.seg + h.NBv; cv.seg_stride = h.NBv; cv.nseg = 1; cv.gidx = nullptr;
                cv.vmeta = h.vmeta; cv.NB = h.NBv; cv.bucket0 = 0;
                if ((rc = snk_launch_count(K, st, cv, err, errcap))) return rc;
            }
        }
        if (hot && !pilot_regrow) {
            // the hot buckets' hash classes: the same kernel over the virtual buckets, appending to the same regions and chunk list
            if (!hot->records) {        // planned, not expanded yet (sharded step): everything the expansion reads has to be there first
                snk_hot* hm = const_cast<snk_hot*>(hot);
                if (hm->before_expand && (rc = hm->before_expand(hm->user))) return snk_fail(SNK_E_ARG, err, errcap, "count: the hot buckets' hook failed (%d)", rc);
                if ((rc = snk_stage_hot_expand(ctx, st, hm->src_records, hm, err, errcap))) return rc;
            }
            snk_count_args cv = ca;
            cv.records = (const uint4*)hot->records;
            cv.seg_beg = hot->seg;
            cv.seg_end = hot->seg + hot->NBv;
            cv.seg_stride = hot->NBv;
            cv.nseg = 1;
            cv.gidx = nullptr;
            cv.vmeta = hot->vmeta;
            cv.NB = hot->NBv;
            cv.bucket0 = 0;
            if ((rc = snk_launch_count(K, st, cv, err, errcap))) return rc;
        }
        if (pilot_regrow) {           // larger regions, the pilot once more (nothing else has run)
            snk_ctx_release_block(ctx, keys_r);
            snk_ctx_release_block(ctx, vals_r);
            if (extra) snk_ctx_release_block(ctx, extra);
            pilot_regrown = true;
            --attempt;
            continue;
        }
        SNK_HIP_TRY(hipMemcpyAsync(h_rcur.data(), rcur, n_regions * 8ull, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(h_status, status, 32, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
#ifdef SNK_COUNT_PROF
        {
            unsigned long long hp[8];
            (void)hipMemcpy(hp, ca.prof, 64, hipMemcpyDeviceToHost);
            unsigned long long tot = 0;
            for (int q = 0; q < 8; ++q) tot += hp[q];
            fprintf(stderr, "[snk prof] count kernel, cycles of thread 0 per phase (%% of total): control %.1f, clear %.1f, stage %.1f, dedupe+scan %.1f, map %.1f, insert %.1f, emit-count %.1f, emit-write %.1f\n",
                    100.0 * hp[0] / tot, 100.0 * hp[1] / tot, 100.0 * hp[2] / tot, 100.0 * hp[3] / tot, 100.0 * hp[4] / tot, 100.0 * hp[5] / tot, 100.0 * hp[6] / tot, 100.0 * hp[7] / tot);
        }
#endif
        if (ca.dbg >= 2) {
            unsigned long long d[3];
            (void)hipMemcpy(d, status + 4, 24, hipMemcpyDeviceToHost);
            fprintf(stderr, "[snk dbg] lane probe iterations %llu, wave-level iterations %llu, max lane iterations in one probe %llu\n", d[0], d[1], d[2]);
        }
        if (h_status[1]) return snk_fail(SNK_E_INTERNAL, err, errcap, h_status[1] == 2 ? "count: a table probe did not terminate" : "count: bucket split depth exceeded");
        unsigned long long mx = 0;
        n_kmers = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { n_kmers += h_rcur[r]; if (h_rcur[r] > mx) mx = h_rcur[r]; }
        const bool extra_ovf = !want_sort && h_status[4] > extra_cap;
        if (!h_status[0] && mx <= region_cap && !extra_ovf) break;
        if (attempt == 3) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: region overflow (%llu > %llu)", mx, (unsigned long long)region_cap);
        snk_ctx_release_block(ctx, keys_r);
        snk_ctx_release_block(ctx, vals_r);
        if (mx > region_cap) region_cap = mx + 64;     // exact requirement is known now (cursors keep counting past the cap)
        if (extra_ovf) extra_cap = h_status[4] + 64;
    }
    if (ranges && ranges->finished) ranges->finished(ranges->user);
    {
        // exclusive offsets of the regions (host: one region per count workgroup, ~16 k) and the dense gather
        std::vector<unsigned long long>& h_off = ctx->h_region_off;      // lives in the context: the upload below is not waited for
        h_off.resize(n_regions + 1);
        unsigned long long acc = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { h_off[r] = acc; acc += h_rcur[r]; }
        h_off[n_regions] = acc;
        SNK_HIP_TRY(hipMemcpyAsync(roff, h_off.data(), (n_regions + 1) * 8ull, hipMemcpyHostToDevice, st));
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 16, &q, err, errcap))) return rc; keys_a = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 8, &q, err, errcap))) return rc; vals_a = (uint64_t*)q;
        out->keys_r = nullptr; out->vals_r = nullptr; out->region_cap = region_cap;
        if (defer_compact) {
            // the bucket-local prune compacts while it reads (snk_local.hip); the dense value words are never made
            snk_ctx_release_block(ctx, vals_a);
            vals_a = nullptr;
            out->keys_r = keys_r; out->vals_r = vals_r;
        } else {
            if ((rc = snk_launch_compact_regions(st, keys_r, vals_r, region_cap, n_regions, rcur, roff, keys_a, vals_a, err, errcap))) return rc;
            snk_ctx_release_block(ctx, keys_r);      // the region-partitioned copy is dead: later stages may reuse it
            snk_ctx_release_block(ctx, vals_r);
        }
    }
    out->buckets_split = h_status[2];
    out->max_slots_used = h_status[3];
    out->distinct = (uint64_t)h_status[5] * 16;
    out->n = n_kmers;
    ctx->last_n_kmers = n_kmers;
    ctx->last_n_instances = n_inst_hint;
    { unsigned long long mxr = 0; for (uint32_t r = 0; r < n_regions; ++r) if (h_rcur[r] > mxr) mxr = h_rcur[r]; ctx->last_region_max = mxr; ctx->last_region_n = n_regions; }
    tm.mark();
    out->sorted = want_sort;
    out->NB = NB;
    out->n_regions = n_regions;
    out->n_extra = want_sort ? 0u : h_status[4];
    if (!want_sort) ctx->last_extra = h_status[4];
    out->chunk_n = chunk_n;
    out->chunk_base = chunk_base;
    out->extra = extra;
    out->region_off = roff;
    if (want_sort) {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 16, &q, err, errcap))) return rc; keys_b = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 8, &q, err, errcap))) return rc; vals_b = (uint64_t*)q;
        if ((rc = snk_graph_sort(ctx, st, K, n_kmers, keys_a, vals_a, keys_b, vals_b, err, errcap))) return rc;
    } else {
        keys_b = keys_a;
        vals_b = vals_a;
    }
    tm.mark();
    out->keys = keys_b;
    out->vals = vals_b;
    out->count_ms = tm.ms(0, 1);
    out->sort_ms = tm.ms(1, 2);
    out->count_kernel_ms = kt.ms(0, 1);
    return SNK_OK;
}

// ===================================================================================================================
// Minimiser partition in one pass (snk_msp.hip, MODE SINGLE): bucket b owns records [b*cap, (b+1)*cap), what does not
// fit goes to an overflow list that is grouped by bucket here (segment 1 of the count kernel's input).
namespace {
// segment tables of the single-pass partition: layout seg[0..NB) = begin of segment 0, [NB..2NB) = end of segment 0,
// [2NB..3NB) = begin of segment 1 (overflow), [3NB..4NB) = end of segment 1
__global__ void __launch_bounds__(256) seg0_kernel(const uint32_t* __restrict__ cursor, uint32_t NB, uint32_t cap,
                                                   uint64_t* __restrict__ seg, unsigned long long* __restrict__ total) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0;
    if (b < NB) {
        const uint32_t c = cursor[b];
        v = c < cap ? c : cap;                 // (beyond the capacity the cursor is a lower bound: hot buckets stop counting, snk_msp.hip; the host adds the overflow lists)
        seg[b] = (uint64_t)b * cap;
        seg[NB + b] = (uint64_t)b * cap + (c < cap ? c : cap);
        seg[2ull * NB + b] = 0;
        seg[3ull * NB + b] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    // (64 counters: the grid's 32 k waves on ONE address were 0.3 of this kernel's 0.4 ms)
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&total[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u], v);
}
// cursor[b] = supermers of bucket b, the overflowed ones included -- exact again once the overflow list is grouped by bucket
__global__ void __launch_bounds__(256) cursor_exact_kernel(const uint64_t* __restrict__ seg, uint32_t NB, uint32_t* __restrict__ cursor) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b < NB) cursor[b] = (uint32_t)((seg[(uint64_t)NB + b] - seg[b]) + (seg[3ull * NB + b] - seg[2ull * NB + b]));
}
// slot reservations stop for a bucket that was handed slot hot_thr (SNK_MSP_HOT_FACTOR x capacity, 0 = never)
static uint32_t msp_hot_thr(uint32_t cap) {
    const uint64_t f = snk_opt_u32("msp_hot_factor", 32);
    const uint64_t t = f * cap;
    const uint64_t lo = snk_opt_u32("msp_hot_min", 4096);
    return f == 0 ? 0xFFFFFFFFu : (uint32_t)(t < lo ? lo : (t > 0x7FFFFFFFull ? 0x7FFFFFFFull : t));
}
// (instances, contributing reads) of one slab added to the streamed job's counters
__global__ void plan_add_kernel(const unsigned long long* __restrict__ two, unsigned long long* __restrict__ plan) {
    if (threadIdx.x < 2) atomicAdd(&plan[threadIdx.x], two[threadIdx.x]);
}
__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
}
// the used prefixes of the overflow sub-lists as ONE list: position j -> slot idx[j] and its bucket key[j]
struct ovf_pre { uint32_t pre[SNK_OVF_SUBLISTS + 1]; };
__global__ void __launch_bounds__(256) ovf_index_kernel(ovf_pre P, uint32_t sub_cap, const uint32_t* __restrict__ ovf_bucket, uint32_t n, uint32_t* __restrict__ idx,
                                                        uint32_t* __restrict__ key) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    uint32_t lo = 0, hi = SNK_OVF_SUBLISTS;                 // largest s with pre[s] <= j
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.pre[mid] <= j) lo = mid; else hi = mid; }
    const uint32_t g = lo * sub_cap + (j - P.pre[lo]);
    idx[j] = g;
    key[j] = ovf_bucket[g];
}
__global__ void __launch_bounds__(256) ovf_gather_kernel(const uint4* rec, uint64_t src_base, uint64_t dst_base,
                                                         const uint32_t* __restrict__ idx, const uint32_t* __restrict__ key, uint32_t n,
                                                         uint32_t NB, uint4* out, uint64_t* __restrict__ seg) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = (src_base + idx[i]) * 2, d = (dst_base + i) * 2;
    out[d] = rec[s];
    out[d + 1] = rec[s + 1];
    const uint32_t b = key[i];
    if (i == 0 || key[i - 1] != b) seg[2ull * NB + b] = dst_base + i;
    if (i + 1 == n || key[i + 1] != b) seg[3ull * NB + b] = dst_base + i + 1;
}


// sharded runs: copy the used slots (+ overflow records) of every bucket to exact offsets of a compact send buffer
__global__ void __launch_bounds__(256) compact_buckets_kernel(const uint4* __restrict__ rec, const uint64_t* __restrict__ seg, uint32_t NB,
                                                              const uint32_t* __restrict__ offsets, uint4* __restrict__ out, uint32_t skip_lo, uint32_t skip_hi) {
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= NB || (b >= skip_lo && b < skip_hi)) return;
    const uint32_t lane = threadIdx.x & 63;
    uint64_t dst = (uint64_t)offsets[b] * 2;
    for (int sgm = 0; sgm < 2; ++sgm) {
        const uint64_t beg = seg[(2ull * sgm) * NB + b] * 2, end = seg[(2ull * sgm + 1) * NB + b] * 2;
        for (uint64_t q = beg + lane; q < end; q += 64) out[dst + (q - beg)] = rec[q];
        dst += end - beg;
    }
}
}  // namespace

static int snk_msp_segments(snk_ctx* ctx, hipStream_t st, uint32_t NB, uint32_t cap, const uint32_t* cursor, uint4* records, uint64_t ovf_base,
                     uint64_t sub_cap, const uint32_t* ovf_bucket, const uint32_t* h_sub /* [SNK_OVF_SUBLISTS] used slots of every sub-list */, uint64_t* seg, char* err,
                     size_t errcap) {
    (void)cap; (void)cursor;
    ovf_pre P;
    uint64_t acc = 0;
    for (uint32_t q = 0; q < SNK_OVF_SUBLISTS; ++q) { P.pre[q] = (uint32_t)acc; acc += h_sub[q]; }
    P.pre[SNK_OVF_SUBLISTS] = (uint32_t)acc;
    const uint32_t n_ovf = (uint32_t)acc;
    // the index / key / sort temporaries below are dead behind the gather: handed back here (everything runs on `st`, the arena's reuse is in
    // stream order), so that bucket-range passes -- up to 64 of them, each replayed run repeating them -- do not pile them up next to the slot
    // array of a job that was too large for the device in the first place
    const uint64_t mark = ctx->alloc_serial;
    struct give_back { snk_ctx* c; uint64_t m; ~give_back() { snk_ctx_release_since(c, m, nullptr, 0); } } _gb{ctx, mark};
    if (n_ovf) {
        uint32_t *idx_in, *idx_out, *key_in, *key_out;
        void* q;
        int rc;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; idx_in = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; idx_out = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; key_in = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; key_out = (uint32_t*)q;
        hipLaunchKernelGGL(ovf_index_kernel, dim3((n_ovf + 255) / 256), dim3(256), 0, st, P, (uint32_t)sub_cap, ovf_bucket, n_ovf, idx_in, key_in);
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, key_in, key_out, idx_in, idx_out, (size_t)n_ovf, 0u, 32u, st));
        if ((rc = snk_ctx_alloc(ctx, tb, &q, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::radix_sort_pairs(q, tb, key_in, key_out, idx_in, idx_out, (size_t)n_ovf, 0u, 32u, st));
        // the grouped copy goes behind all the sub-lists
        hipLaunchKernelGGL(ovf_gather_kernel, dim3((n_ovf + 255) / 256), dim3(256), 0, st, records, ovf_base, ovf_base + sub_cap * SNK_OVF_SUBLISTS, idx_out,
                           key_out, n_ovf, NB, records, seg);
    }
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}


int snk_stage_partition_plan(snk_ctx* ctx, hipStream_t st, uint32_t K, const uint16_t* good_len, uint64_t n_reads,
                             unsigned long long h_plan[2], char* err, size_t errcap, unsigned long long** d_plan_out) {
    unsigned long long* counters;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc;
    counters = (unsigned long long*)q;
    if ((rc = snk_launch_msp_plan(st, good_len, n_reads, K, counters, err, errcap))) return rc;
    if (d_plan_out) { *d_plan_out = counters; return SNK_OK; }      // the caller reads the counters with a later read-back
    SNK_HIP_TRY(hipMemcpyAsync(h_plan, counters, 16, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    return SNK_OK;
}

bool snk_fused_trim_ok(const snk_dev_reads* in) {
    return in->quals && !in->good_len && (in->qstride & 3u) == 0 && (((uintptr_t)in->quals) & 3u) == 0 && in->read_len <= 160 && in->qstride >= in->read_len &&
           (!in->lens || (((uintptr_t)in->lens) & 1u) == 0) && snk_opt_u32("trim_fused", 1) != 0;
}

// expected supermers of a pass and the record slots a bucket gets
static void partition_capacity(snk_ctx* ctx, uint32_t K, uint32_t NB, unsigned long long n_inst, unsigned long long n_live, bool grouped, double* est_super_out,
                               uint64_t* cap_out, uint32_t passes = 1, uint64_t* ideal_out = nullptr) {
    const uint32_t Wm = K - ctx->mlen + 1;
    // a random-order minimiser starts a new supermer every (W+1)/2 k-mers, and every contributing read starts one
    const double est_super = (double)n_inst * 2.0 / (Wm + 1) + (double)n_live;
    const double mean = est_super / NB;
    // Bucket occupancy is NOT Poisson in the supermers: a minimiser site of the genome contributes one supermer per read that
    // covers it (c ~ 38 at 56x), so a bucket of `mean` supermers holds mean / c sites and its count has sigma = sqrt(mean c).
    // capacity = mean + 5 sigma.  At 4000-instance buckets on one GPU (mean 340, 9 sites) that is 2.7 x mean -- the 2.5 x mean + 32
    // this code used before; the sharded path partitions the same reads into world x as many buckets (mean 42 at eight ranks:
    // ONE site), where 2.5 x mean lost a tenth of the supermers to the overflow list, overran it, and ran the pass twice
    // (78 instead of 34 ms).  c is a property of the data set: 48 covers 70x; per-barcode groups see a site once or twice.
    // Overflowing supermers are correct (segment 1), only slower; the slots that stay empty are never touched.
    const double site_records = (double)snk_opt_u32("msp_site_records", grouped ? 3u : 48u);
    // ... and how many sigma is a matter of memory: at 5 the slots are 2.9 x the records (100 M reads: 66 GB of slots for 22 GB of records, nothing
    // overflows); at 3 sigma 0.03 % of the supermers take the overflow list, at 1.2 sigma 2.2 % do, for +1 / +1.5 ms of the 104 ms step
    // (tools/r6_cap_sigma.sh, profiles/r06_cap_sigma.log).  So: 5 sigma while the slots stay below 30 % of the device, 3 above, 1.5 if those are
    // still more than 30 % -- only where a bucket holds several sites (the Gaussian regime; a rank of the 8-GPU job has ONE site per bucket and
    // keeps its 5 sigma, the 45 % budget below and the larger overflow list).  Option msp_sigmas_x10 pins it.
    // (memory = what this context can count on, ctx->plan_mem: a caller that holds 60 GB of reads leaves less than one that holds 15.)  A job in
    // bucket-range passes is short of memory by definition: 1.5 sigma (what saves a pass saves a scan of every read).
    double sig = 5.0;
    const double sigma = std::sqrt(mean * site_records);
    if (snk_opt_is_set("msp_sigmas_x10")) sig = 0.1 * snk_opt_u32("msp_sigmas_x10", 50);
    // (1.5 where that saves a pass -- a pass fewer at 600 and 800 M reads: 73.4 -> 83.3 and 64.5 -> 72.3 Gk-mers/s, 1.5-2 % of the supermers take the overflow lists --
    // and as much slack as the same number of passes holds otherwise: snk_partition_passes_needed leaves its choice in ctx->pass_sigma)
    else if (passes > 1) sig = mean >= 4.0 * site_records ? (ctx->pass_sigma > 0.0 ? ctx->pass_sigma : 1.5) : 5.0;
    else if (ctx->plan_mem && mean >= 4.0 * site_records) {
        const double lim = 0.30 * (double)ctx->plan_mem;
        if ((mean + 5.0 * sigma + 16.0) * NB * 32.0 > lim) sig = 3.0;
        if (sig < 5.0 && (mean + 3.0 * sigma + 16.0) * NB * 32.0 > lim) sig = 1.5;
        // ... and of what the arena already holds.  Memory a process has not mapped yet is not free: the driver clears what another
        // process used before it hands it out, ~33 ms per GB (a second snk_mspedges run on 100 M reads spent 2.06 s mapping its 62.5 GB of
        // 5-sigma slots and 0.5 s computing: profiles/r06_oneshot.log) -- 35 GB of slack cost a second, the overflow list it saves 1.5 ms.
        // So the slack is taken only when the arena has it (a host that wants the last 1.5 ms maps ahead: snk_ctx_reserve, as bench.py does).
        if (sig > 1.5 && snk_opt_u32("lean_cold", 1)) {
            const double other = 2.6 * (double)n_inst;          // regions, table, graph stage: ~27 GB per 10 G instances at 56x
            if ((double)ctx->plan_mapped < (mean + sig * sigma + 16.0) * NB * 32.0 + other) sig = 1.5;
        }
    }
    uint64_t cap64 = (uint64_t)(mean + sig * sigma + 16.0);
    cap64 = cap64 * snk_opt_u32("msp_cap_pct", 100) / 100;
    if (cap64 < 2) cap64 = 2;
    cap64 = (cap64 + 1) & ~1ull;
    if (ideal_out) *ideal_out = cap64;
    if (passes <= 1) {
        // not more than half of what the context can count on for the slots: beyond that the capacity shrinks and the overflow list takes the rest
        const uint64_t tot = ctx->plan_mem;
        if (tot) {
            const uint64_t budget = (uint64_t)((double)tot * 0.50);
            if (cap64 * NB * 32ull > budget) {
                uint64_t c2 = budget / (NB * 32ull);
                const uint64_t floor_ = (uint64_t)(mean * 1.25 + 8.0);
                if (c2 < floor_) c2 = floor_;
                if (c2 < cap64) cap64 = c2 & ~1ull;
            }
        }
    }
    *est_super_out = est_super;
    *cap_out = cap64;
}

// ---- dense partition: no slots, no slot reservations.  The 0.69 G returning atomics of the one-pass partition are what its kernel
// waits for (27 G/s device-wide whatever their flavour: tools/probe/atomics.hip; the emission alone takes the kernel's whole time:
// tools/probe/partition2.hip).  Here the records leave the scan kernel in read order -- a workgroup reserves its block with one atomic --
// next to a u32 bucket id each; what is sorted is (bucket, position): 8 bytes instead of 32 per record, by a radix sort over the
// ceil(log2 NB) key bits; the count kernel fetches a bucket's records through the sorted positions (LDS-DMA takes per-lane addresses).
namespace {
__global__ void __launch_bounds__(256) seg_from_sorted_kernel(const uint32_t* __restrict__ key, uint32_t n, uint32_t NB, uint64_t* __restrict__ seg) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[i];
    if (k >= NB) return;
    if (i == 0 || key[i - 1] != k) seg[k] = i;
    if (i + 1 == n || key[i + 1] != k) seg[(uint64_t)NB + k] = (uint64_t)i + 1;
}
int partition_dense(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_dev_reads* in, const uint16_t* good_len, uint32_t NB, double est_super, bool grouped,
                    uint32_t* status, snk_partition* out, char* err, size_t errcap, const unsigned long long* d_plan, unsigned long long* h_plan,
                    const snk_fused_trim* ft) {
    int rc;
    const uint64_t n_reads = in->n_reads;
    uint64_t dcap = (uint64_t)(est_super * 1.12) + (1u << 20);
    if (ctx->last_ovf_nb == 0xD0000000u && ctx->last_ovf_reads == n_reads && ctx->last_dense) dcap = ctx->last_dense + ctx->last_dense / 64 + 65536;   // what the last call on these reads needed
    uint64_t* seg = nullptr;
    unsigned long long* d_cur = nullptr;
    unsigned long long* d_fplan = nullptr;
    std::vector<unsigned long long> h_fplan;
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, 4ull * NB * 8 + 64, &q, err, errcap))) return rc; seg = (uint64_t*)q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; d_cur = (unsigned long long*)q;
        if (ft) {
            if (!h_plan) return snk_fail(SNK_E_ARG, err, errcap, "fused trim: the caller takes the instance count from h_plan");
            if ((rc = snk_ctx_alloc(ctx, 2ull * SNK_MSP_PLAN_SLOTS * 8, &q, err, errcap))) return rc; d_fplan = (unsigned long long*)q;
            h_fplan.resize(2 * SNK_MSP_PLAN_SLOTS);
        }
    }
    snk_phase_timer kt(st);
    void* records = nullptr;
    uint32_t* bkt = nullptr;
    unsigned long long h_n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (dcap >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "dense partition: more than 2^32 supermer records");
        void* q;
        if ((rc = snk_ctx_alloc(ctx, dcap * 32 + 64, &records, err, errcap))) return rc;
        if ((rc = snk_ctx_alloc(ctx, dcap * 4 + 64, &q, err, errcap))) return rc; bkt = (uint32_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(d_cur, 0, 8, st));
        snk_msp_args ma;
        memset(&ma, 0, sizeof ma);
        ma.rows = (const uint32_t*)in->rows; ma.row_words = in->row_words; ma.read_len = in->read_len; ma.good_len = good_len; ma.bc = (const int32_t*)in->bc;
        ma.ign_bc_below = in->ign_bc_below; ma.read_index_base = in->read_index_base; ma.n_reads = n_reads; ma.NB = NB;
        ma.group = grouped ? (const uint32_t*)in->group : nullptr;
        ma.records = (uint4*)records;
        ma.dense_bkt = bkt; ma.dense_cursor = d_cur; ma.dense_cap = dcap;
        ma.dbg = snk_opt_u32("msp_dbg", 0);
        if (ft) {
            ma.good_len = ft->good_out; ma.quals = (const uint8_t*)ft->quals; ma.qstride = ft->qstride; ma.min_qual = ft->min_qual;
            ma.lens = (const uint16_t*)ft->lens; ma.good_out = ft->good_out; ma.plan = d_fplan;
            SNK_HIP_TRY(hipMemsetAsync(d_fplan, 0, 2ull * SNK_MSP_PLAN_SLOTS * 8, st));
        }
        kt.n = 0;
        kt.mark();  // 0
        if ((rc = snk_launch_msp(K, ctx->mlen, st, ma, err, errcap))) return rc;
        kt.mark();  // 1
        SNK_HIP_TRY(hipMemcpyAsync(&h_n, d_cur, 8, hipMemcpyDeviceToHost, st));
        if (d_plan && h_plan && !ft) SNK_HIP_TRY(hipMemcpyAsync(h_plan, d_plan, 16, hipMemcpyDeviceToHost, st));
        if (ft) SNK_HIP_TRY(hipMemcpyAsync(h_fplan.data(), d_fplan, h_fplan.size() * 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        if (ft) {
            h_plan[0] = h_plan[1] = 0;
            for (int q2 = 0; q2 < SNK_MSP_PLAN_SLOTS; ++q2) { h_plan[0] += h_fplan[2 * q2]; h_plan[1] += h_fplan[2 * q2 + 1]; }
        }
        ctx->last_ovf_nb = 0xD0000000u; ctx->last_ovf_reads = n_reads; ctx->last_dense = h_n;
        if (h_n <= dcap) break;
        if (attempt == 1) return snk_fail(SNK_E_INTERNAL, err, errcap, "dense partition: record array too small (%llu > %llu)", h_n, (unsigned long long)dcap);
        snk_ctx_release_block(ctx, records);
        snk_ctx_release_block(ctx, bkt);
        dcap = h_n + 65536;
    }
    // ---- (bucket, position) sorted by bucket; the bounds of every bucket in the sorted list
    uint32_t *bkt2 = nullptr, *gidx = nullptr;
    const size_t n = (size_t)h_n;
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, n * 4 + 64, &q, err, errcap))) return rc; bkt2 = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, n * 4 + 64, &q, err, errcap))) return rc; gidx = (uint32_t*)q;
    }
    SNK_HIP_TRY(hipMemsetAsync(seg, 0, 4ull * NB * 8, st));
    if (n) {
        unsigned bits = 1;
        while (bits < 32 && (1ull << bits) < NB) ++bits;
        rocprim::counting_iterator<uint32_t> pos(0);
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, bkt, bkt2, pos, gidx, n, 0u, bits, st));
        void* tmp;
        if ((rc = snk_ctx_alloc(ctx, tb + 64, &tmp, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, bkt, bkt2, pos, gidx, n, 0u, bits, st));
        hipLaunchKernelGGL(seg_from_sorted_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bkt2, (uint32_t)n, NB, seg);
        SNK_HIP_TRY(hipGetLastError());
        snk_ctx_release_block(ctx, tmp);
    }
    kt.mark();  // 2
    snk_ctx_release_block(ctx, bkt);
    snk_ctx_release_block(ctx, bkt2);
    (void)status;
    out->NB = NB;
    out->cap = 0;
    out->nseg = 1;
    out->n_overflow = 0;
    out->n_supermers = h_n;
    out->records = records;
    out->cursor = nullptr;
    out->seg = seg;
    out->gidx = gidx;
    out->kernel_ms = kt.ms(0, 1);
    out->sort_ms = kt.ms(1, 2);
    return SNK_OK;
}
}  // namespace

// ---- measurement aid (SNK_OVERLAP_PROBE, snk_pipeline.hip): the last partition launch once more, into scratch memory of its own, on
// ANOTHER stream -- next to whatever the caller's stream runs (the count kernel).  Says what the hardware makes of an atomics-bound and a
// VALU-bound kernel that are resident at the same time.  Results of the call are not touched.
#ifdef SNK_PROBES      // measurement aids (tools/overlap_probe*.py; tuning builds: tools/build_variant.sh probes -DSNK_PROBES): not in the product build
static snk_msp_args snk_probe_last_msp;
static uint32_t snk_probe_last_msp_K = 0;
static uint64_t snk_probe_last_msp_ovf_cap = 0;
// lean emitter probe: what the slot reservations + record stores cost without LDS and with few registers (a kernel that can sit NEXT TO
// the count kernel's two 79-KB workgroups per CU): thread i reserves a slot of pseudo-random bucket hash(i) and stores a 32-byte record
__global__ void __launch_bounds__(64) probe_lean_emit_kernel(uint32_t* __restrict__ cursor, uint4* __restrict__ records, uint32_t NB, uint32_t cap, uint64_t n, uint32_t per_thread) {
    const uint64_t t0 = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * per_thread;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint64_t i = t0 + k;
        if (i >= n) return;
        const uint32_t h = snk_mix32((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
        const uint32_t b = (uint32_t)(((uint64_t)h * NB) >> 32);
        const uint32_t slot = atomicAdd(&cursor[b], 1u);
        if (slot < cap) {
            uint4* dst = records + ((uint64_t)b * cap + slot) * 2;
            dst[0] = make_uint4(h, h + 1, h + 2, h + 3);
            dst[1] = make_uint4(h + 4, h + 5, h + 6, (uint32_t)i);
        }
    }
}

// ... and the same as a PERSISTENT grid of a few waves per SIMD (a grid of millions of tiny workgroups takes every free wave slot and
// starves the count kernel's 768-thread workgroups: the two then run one after the other): four records in flight per lane
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(24))) probe_lean_emit_persistent(uint32_t* __restrict__ cursor, uint4* __restrict__ records, uint32_t NB, uint32_t cap, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 64 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint32_t h[4], b[4], slot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + k * stride;
            h[k] = snk_mix32((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
            b[k] = (uint32_t)(((uint64_t)h[k] * NB) >> 32);
            slot[k] = i < n ? atomicAdd(&cursor[b[k]], 1u) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (slot[k] < cap) {
                uint4* dst = records + ((uint64_t)b[k] * cap + slot[k]) * 2;
                dst[0] = make_uint4(h[k], h[k] + 1, h[k] + 2, h[k] + 3);
                dst[1] = make_uint4(h[k] + 4, h[k] + 5, h[k] + 6, h[k] + 7);
            }
        }
    }
}

int snk_probe_relaunch_msp(snk_ctx* ctx, hipStream_t s2, uint32_t dbg, char* err, size_t errcap) {
    if (!snk_probe_last_msp_K) return SNK_OK;
    snk_msp_args ma = snk_probe_last_msp;
    if (dbg >= 8) {       // 8: lean emitter, one record per thread; 9: sixteen per thread (fewer, longer-lived waves); 10 / 11: a quarter of the records
        const uint32_t NB = ma.NB;
        void* q;
        int rc2;
        if ((rc2 = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc2; uint32_t* cur = (uint32_t*)q;
        if ((rc2 = snk_ctx_alloc(ctx, (size_t)NB * ma.cap * 32 + 64, &q, err, errcap))) return rc2; uint4* recs = (uint4*)q;
        SNK_HIP_TRY(hipMemsetAsync(cur, 0, (NB + 1) * 4ull, s2));
        uint64_t n = (uint64_t)(0.98 * (double)NB * ma.cap * 0.55);       // ~ the supermers of the call (slots are ~1.8x the mean)
        if (dbg >= 10) n /= 4;
        if (dbg >= 16) {          // 16 + w: persistent grid of w workgroups (waves) per CU, all the records
            n = (uint64_t)(0.98 * (double)NB * ma.cap * 0.55);
            const uint32_t wpc = dbg - 16;
            hipLaunchKernelGGL(probe_lean_emit_persistent, dim3((unsigned)ctx->n_cu * wpc), dim3(64), 0, s2, cur, recs, NB, ma.cap, n);
            SNK_HIP_TRY(hipGetLastError());
            fprintf(stderr, "[snk overlap probe] persistent lean emitter: %llu records, %u waves per CU\n", (unsigned long long)n, wpc);
            return SNK_OK;
        }
        const uint32_t per = (dbg & 1u) ? 16u : 1u;
        const uint64_t threads = (n + per - 1) / per;
        hipLaunchKernelGGL(probe_lean_emit_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, s2, cur, recs, NB, ma.cap, n, per);
        SNK_HIP_TRY(hipGetLastError());
        fprintf(stderr, "[snk overlap probe] lean emitter: %llu records into %u buckets of %u slots, %u per thread\n", (unsigned long long)n, NB, ma.cap, per);
        return SNK_OK;
    }
    const uint32_t NB = ma.NB;
    int rc;
    void* q;
    if ((rc = snk_ctx_alloc(ctx, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, &q, err, errcap))) return rc; ma.cursor = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, ((size_t)NB * ma.cap + 2 * snk_probe_last_msp_ovf_cap) * 32 + 64, &q, err, errcap))) return rc; ma.records = (uint4*)q;
    if ((rc = snk_ctx_alloc(ctx, snk_probe_last_msp_ovf_cap * 4 + 64, &q, err, errcap))) return rc; ma.ovf_bucket = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4 + 64, &q, err, errcap))) return rc; ma.ovf_cursor = (uint32_t*)q;
    ma.hot_tab = ma.cursor + NB + 1;
    if (ma.plan) { if ((rc = snk_ctx_alloc(ctx, 2ull * SNK_MSP_PLAN_SLOTS * 8, &q, err, errcap))) return rc; ma.plan = (unsigned long long*)q; SNK_HIP_TRY(hipMemsetAsync(ma.plan, 0, 2ull * SNK_MSP_PLAN_SLOTS * 8, s2)); }
    ma.dbg = dbg;
    SNK_HIP_TRY(hipMemsetAsync(ma.cursor, 0, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, s2));
    SNK_HIP_TRY(hipMemsetAsync(ma.ovf_cursor, 0, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4, s2));
    return snk_launch_msp(snk_probe_last_msp_K, ctx->mlen, s2, ma, err, errcap);
}
#endif

int snk_stage_partition(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_dev_reads* in, const uint16_t* good_len, uint32_t NB,
                        unsigned long long n_inst, unsigned long long n_live, bool grouped, uint32_t* status, snk_partition* out,
                        char* err, size_t errcap, const unsigned long long* d_plan, unsigned long long* h_plan, const snk_fused_trim* ft, bool allow_dense) {
    memset(out, 0, sizeof *out);
    const uint64_t n_reads = in->n_reads;
    double est_super = 0;
    uint64_t cap64 = 0;
    partition_capacity(ctx, K, NB, n_inst, n_live, grouped, &est_super, &cap64);
    if (allow_dense && est_super * 1.25 + 2e6 < 4.0e9 && snk_opt_u32("msp_dense", 0))
        return partition_dense(ctx, st, K, in, good_len, NB, est_super, grouped, status, out, err, errcap, d_plan, h_plan, ft);
    if (cap64 * NB >= (1ull << 40) || cap64 >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "bucket capacity out of range");
    const uint32_t cap = (uint32_t)cap64;
    uint64_t ovf_cap = (uint64_t)(est_super / 16) + 65536;
    // (the same job geometry as the last call: what that one needed, so that a steady stream of calls never runs the pass twice)
    if (ctx->last_ovf_nb == NB && ctx->last_ovf_reads == n_reads && (uint64_t)ctx->last_ovf * 5 / 4 + 65536 > ovf_cap) ovf_cap = (uint64_t)ctx->last_ovf * 5 / 4 + 65536;
    int rc;
    uint32_t* cursor = nullptr;
    uint64_t* seg = nullptr;           // [2 segments][beg NB | end NB]
    unsigned long long* d_total = nullptr;
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, &q, err, errcap))) return rc; cursor = (uint32_t*)q;     // the hot table behind the cursors
        if ((rc = snk_ctx_alloc(ctx, 4ull * NB * 8 + 64, &q, err, errcap))) return rc; seg = (uint64_t*)q;
        if ((rc = snk_ctx_alloc(ctx, 64 * 8, &q, err, errcap))) return rc; d_total = (unsigned long long*)q;
    }
    // fused trim: the kernel's own sizing counters (the exact instance count comes back with the pass's read-back)
    unsigned long long* d_fplan = nullptr;
    std::vector<unsigned long long> h_fplan;
    if (ft) {
        void* q;
        if (!h_plan) return snk_fail(SNK_E_ARG, err, errcap, "fused trim: the caller takes the instance count from h_plan");
        if ((rc = snk_ctx_alloc(ctx, 2ull * SNK_MSP_PLAN_SLOTS * 8, &q, err, errcap))) return rc; d_fplan = (unsigned long long*)q;
        h_fplan.resize(2 * SNK_MSP_PLAN_SLOTS);
    }
    snk_phase_timer kt(st);
    void* records = nullptr;
    uint32_t* ovf_bucket = nullptr;
    uint32_t* ovf_cur = nullptr;        // [SNK_OVF_SUBLISTS] the sub-lists' cursors
    { void* q; if ((rc = snk_ctx_alloc(ctx, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4 + 64, &q, err, errcap))) return rc; ovf_cur = (uint32_t*)q; }
    uint32_t h_novf = 0;
    uint32_t h_sub[SNK_OVF_SUBLISTS];
    uint32_t h_cur[SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE];      // the cursors as they lie on the device, one per 128 bytes
    unsigned long long h_total = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        ovf_cap = (ovf_cap + SNK_OVF_SUBLISTS - 1) / SNK_OVF_SUBLISTS * SNK_OVF_SUBLISTS;
        if (ovf_cap >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "supermer overflow list too large");
        const uint64_t sub_cap = ovf_cap / SNK_OVF_SUBLISTS;
        void* q;
        if ((rc = snk_ctx_alloc(ctx, ((size_t)NB * cap + 2 * ovf_cap) * 32 + 64, &records, err, errcap))) return rc;
        if ((rc = snk_ctx_alloc(ctx, ovf_cap * 4 + 64, &q, err, errcap))) return rc; ovf_bucket = (uint32_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(cursor, 0, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, st));
        SNK_HIP_TRY(hipMemsetAsync(ovf_cur, 0, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4, st));
        snk_msp_args ma;
        memset(&ma, 0, sizeof ma);
        ma.rows = (const uint32_t*)in->rows; ma.row_words = in->row_words; ma.read_len = in->read_len; ma.good_len = good_len; ma.bc = (const int32_t*)in->bc;
        ma.ign_bc_below = in->ign_bc_below; ma.read_index_base = in->read_index_base; ma.n_reads = n_reads; ma.NB = NB;
        ma.group = grouped ? (const uint32_t*)in->group : nullptr;
        ma.cursor = cursor; ma.records = (uint4*)records; ma.cap = cap; ma.ovf_cap = (uint32_t)sub_cap;
        ma.ovf_base = (uint64_t)NB * cap; ma.ovf_bucket = ovf_bucket; ma.ovf_cursor = ovf_cur;
        ma.hot_tab = cursor + NB + 1; ma.hot_thr = msp_hot_thr(cap);
        ma.dbg = snk_opt_u32("msp_dbg", 0);
        if (ft) {
            ma.good_len = ft->good_out; ma.quals = (const uint8_t*)ft->quals; ma.qstride = ft->qstride; ma.min_qual = ft->min_qual;
            ma.lens = (const uint16_t*)ft->lens; ma.good_out = ft->good_out; ma.plan = d_fplan;
            SNK_HIP_TRY(hipMemsetAsync(d_fplan, 0, 2ull * SNK_MSP_PLAN_SLOTS * 8, st));
        }
        kt.n = 0;
        kt.mark();  // 0
        if ((rc = snk_launch_msp(K, ctx->mlen, st, ma, err, errcap))) return rc;
        kt.mark();  // 1
#ifdef SNK_PROBES
        snk_probe_last_msp = ma; snk_probe_last_msp_K = K; snk_probe_last_msp_ovf_cap = ovf_cap;
#endif
        // segment 0 (the fixed-capacity slots) and the supermer total need the cursors only: one read-back for everything the
        // host wants to know about this pass (overflow count, supermers, and the caller's trim statistics if asked for)
        SNK_HIP_TRY(hipMemsetAsync(d_total, 0, 64 * 8, st));
        hipLaunchKernelGGL(seg0_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, cursor, NB, cap, seg, d_total);
        SNK_HIP_TRY(hipMemcpyAsync(h_cur, ovf_cur, sizeof h_cur, hipMemcpyDeviceToHost, st));
        unsigned long long h_tot64[64];
        SNK_HIP_TRY(hipMemcpyAsync(h_tot64, d_total, sizeof h_tot64, hipMemcpyDeviceToHost, st));
        if (d_plan && h_plan && !ft) SNK_HIP_TRY(hipMemcpyAsync(h_plan, d_plan, 16, hipMemcpyDeviceToHost, st));
        if (ft) SNK_HIP_TRY(hipMemcpyAsync(h_fplan.data(), d_fplan, h_fplan.size() * 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        for (uint32_t q2 = 0; q2 < SNK_OVF_SUBLISTS; ++q2) h_sub[q2] = h_cur[q2 * SNK_OVF_CUR_STRIDE];
        h_total = 0;
        for (int q = 0; q < 64; ++q) h_total += h_tot64[q];
        if (ft) {
            h_plan[0] = h_plan[1] = 0;
            for (int q = 0; q < SNK_MSP_PLAN_SLOTS; ++q) { h_plan[0] += h_fplan[2 * q]; h_plan[1] += h_fplan[2 * q + 1]; }
        }
        // every sub-list within its slots?  (the wanted total sizes the next call's list, the largest sub-list a re-run's)
        uint64_t want = 0, mx = 0;
        for (uint32_t q2 = 0; q2 < SNK_OVF_SUBLISTS; ++q2) { want += h_sub[q2]; if (h_sub[q2] > mx) mx = h_sub[q2]; }
        h_novf = (uint32_t)(want > 0xFFFFFFFFull ? 0xFFFFFFFFull : want);
        h_total += want;
        ctx->last_ovf = (uint32_t)(mx * SNK_OVF_SUBLISTS > 0xFFFFFFFFull ? 0xFFFFFFFFull : mx * SNK_OVF_SUBLISTS); ctx->last_ovf_nb = NB; ctx->last_ovf_reads = n_reads;
        if (mx <= sub_cap) break;
        if (attempt == 2) return snk_fail(SNK_E_INTERNAL, err, errcap, "supermer overflow list too small (%llu > %llu)", (unsigned long long)mx, (unsigned long long)sub_cap);
        snk_ctx_release_block(ctx, records);
        snk_ctx_release_block(ctx, ovf_bucket);
        ovf_cap = (mx + mx / 8 + 1024) * SNK_OVF_SUBLISTS;
    }
    // segment 1: the overflow records grouped by bucket
    if ((rc = snk_msp_segments(ctx, st, NB, cap, cursor, (uint4*)records, (uint64_t)NB * cap, ovf_cap / SNK_OVF_SUBLISTS, ovf_bucket, h_sub, seg, err, errcap))) return rc;
    if (h_novf) hipLaunchKernelGGL(cursor_exact_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, seg, NB, cursor);
    out->NB = NB;
    out->cap = cap;
    out->nseg = h_novf ? 2u : 1u;
    out->n_overflow = h_novf;
    out->n_supermers = h_total;
    out->records = records;
    out->cursor = cursor;
    out->seg = seg;
    out->kernel_ms = kt.ms(0, 1);
    return SNK_OK;
}

// ===================================================================================================================
// Bucket-range passes (snk_stages.h): the same kernel, RANGED; every pass reuses the slot array and the overflow area.
namespace {
__global__ void __launch_bounds__(256) vmeta_shift_kernel(uint2* __restrict__ vm, uint32_t n, uint32_t add) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) vm[i].x += add;
}
__global__ void __launch_bounds__(256) seg0_range_kernel(const uint32_t* __restrict__ cursor, uint32_t b_lo, uint32_t b_hi, uint32_t NB, uint32_t cap,
                                                         uint64_t* __restrict__ seg, unsigned long long* __restrict__ total) {
    const uint32_t b = b_lo + blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0;
    if (b < b_hi) {
        const uint32_t c = cursor[b];
        v = c < cap ? c : cap;
        seg[b] = (uint64_t)(b - b_lo) * cap;
        seg[NB + b] = (uint64_t)(b - b_lo) * cap + v;
        seg[2ull * NB + b] = 0;
        seg[3ull * NB + b] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&total[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u], v);
}
}  // namespace

uint32_t snk_partition_passes_needed(snk_ctx* ctx, uint32_t K, uint32_t NB, unsigned long long n_inst, unsigned long long n_live, bool grouped) {
    const uint32_t forced = snk_opt_u32("partition_passes", 0);
    ctx->pass_sigma = 0.0;
    if (forced) return forced > 64 ? 64u : forced;
    double est = 0;
    uint64_t cap = 0, ideal = 0;
    partition_capacity(ctx, K, NB, n_inst, n_live, grouped, &est, &cap, 1, &ideal);
    const uint64_t tot = ctx->plan_mem;
    if (!tot || ideal * NB * 32ull <= (uint64_t)((double)tot * 0.50)) return 1;       // (the one-pass partition's own limit)
    // A pass scans every read again (13 ms per 100 M reads): as few as fit.  A pass's slots (at the capacity a job in passes gets) take 45 %
    // of what the context can count on -- next to them the count regions (24 bytes per retained k-mer) and the overflow lists have to fit;
    // the graph stage that follows needs less than both (800 M reads: 112 + 58 + 15 GB of 248 while counting, 156 GB at the end)
    const uint64_t per_pass = (uint64_t)((double)tot * 0.45);
    auto passes_at = [&](double sg) {
        ctx->pass_sigma = sg;
        partition_capacity(ctx, K, NB, n_inst, n_live, grouped, &est, &cap, 2, &ideal);
        const uint64_t q = (ideal * NB * 32ull + per_pass - 1) / per_pass;
        return q < 2 ? (uint64_t)2 : q;
    };
    const uint64_t p = passes_at(1.5);
    if (passes_at(5.0) != p && passes_at(3.0) != p) ctx->pass_sigma = 1.5;      // (the last one tried that still gives p passes stays in ctx->pass_sigma)
    return (uint32_t)(p > 64 ? 64 : p);
}

int snk_partition_passes_open(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_dev_reads* in, const uint16_t* good_len, const snk_fused_trim* ft, uint32_t NB,
                              uint32_t passes, unsigned long long n_inst, unsigned long long n_live, bool grouped, snk_partition_passes* S, char* err, size_t errcap) {
    if (passes < 1 || passes > 64 || passes > NB) return snk_fail(SNK_E_ARG, err, errcap, "partition passes: 1..64 (and at most one per bucket)");
    S->ctx = ctx; S->st = st; S->K = K; S->NB = NB; S->P = passes; S->grouped = grouped; S->in = *in; S->good_len = good_len; S->fused = ft != nullptr;
    if (ft) S->ft = *ft;
    S->hots = nullptr; S->n_hot = 0;
    S->err = err; S->errcap = errcap; S->n_supermers = 0; S->n_overflow = 0; S->kernel_ms = 0.f; S->runs = 0; S->h_plan[0] = n_inst; S->h_plan[1] = n_live;
    double est_super = 0;
    uint64_t cap64 = 0;
    partition_capacity(ctx, K, NB, n_inst, n_live, grouped, &est_super, &cap64, passes);
    if (cap64 >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "bucket capacity out of range");
    S->cap = (uint32_t)cap64;
    uint32_t widest = 0;
    for (uint32_t r = 0; r <= passes; ++r) S->bounds[r] = (uint32_t)((uint64_t)NB * r / passes);
    for (uint32_t r = 0; r < passes; ++r) widest = std::max(widest, S->bounds[r + 1] - S->bounds[r]);
    S->slots_per_pass = (uint64_t)widest * S->cap;
    // (nothing can be looked at and run again with a larger list: a generous one, as for a streamed job)
    S->ovf_cap = ((uint64_t)(est_super / passes / 6) + (1u << 20) + SNK_OVF_SUBLISTS - 1) / SNK_OVF_SUBLISTS * SNK_OVF_SUBLISTS;
    if (S->ovf_cap >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "supermer overflow list too large");
    int rc;
    void* q;
    if ((rc = snk_ctx_alloc(ctx, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, &q, err, errcap))) return rc; S->cursor = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, 4ull * NB * 8 + 64, &q, err, errcap))) return rc; S->seg = (uint64_t*)q;
    if ((rc = snk_ctx_alloc(ctx, 64 * 8, &q, err, errcap))) return rc; S->d_total = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, 2ull * SNK_MSP_PLAN_SLOTS * 8, &q, err, errcap))) return rc; S->d_fplan = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, (S->slots_per_pass + 2 * S->ovf_cap) * 32 + 64, &S->records, err, errcap))) return rc;
    if ((rc = snk_ctx_alloc(ctx, S->ovf_cap * 4 + 64, &q, err, errcap))) return rc; S->ovf_bucket = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4 + 64, &q, err, errcap))) return rc; S->ovf_cur = (uint32_t*)q;
    return SNK_OK;
}

int snk_partition_passes_run(void* user, uint32_t r) {
    snk_partition_passes* S = static_cast<snk_partition_passes*>(user);
    snk_ctx* ctx = S->ctx;
    hipStream_t st = S->st;
    char* err = S->err;
    size_t errcap = S->errcap;
    if (r >= S->P) return snk_fail(SNK_E_INTERNAL, err, errcap, "partition passes: range %u of %u", r, S->P);
    const uint32_t NB = S->NB, b_lo = S->bounds[r], b_hi = S->bounds[r + 1];
    if (r == 0) {       // (a run of all passes starts: also a repeated one)
        SNK_HIP_TRY(hipMemsetAsync(S->cursor, 0, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, st));
        S->n_supermers = 0; S->n_overflow = 0; S->kernel_ms = 0.f; S->n_hot = 0;
        if (S->hots) S->hots->clear();
        ++S->runs;
    }
    SNK_HIP_TRY(hipMemsetAsync(S->ovf_cur, 0, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4, st));
    SNK_HIP_TRY(hipMemsetAsync(S->cursor + NB + 1, 0, SNK_MSP_HOT_TAB * 4ull, st));        // (the table of buckets that stopped reserving slots is a pass's own)
    const bool trim_now = S->fused && r == 0 && S->runs == 1;          // the first pass derives the good lengths, the others read them
    snk_msp_args ma;
    memset(&ma, 0, sizeof ma);
    const snk_dev_reads* in = &S->in;
    ma.rows = (const uint32_t*)in->rows; ma.row_words = in->row_words; ma.read_len = in->read_len; ma.good_len = S->fused ? S->ft.good_out : S->good_len; ma.bc = (const int32_t*)in->bc;
    ma.ign_bc_below = in->ign_bc_below; ma.read_index_base = in->read_index_base; ma.n_reads = in->n_reads; ma.NB = NB;
    ma.group = S->grouped ? (const uint32_t*)in->group : nullptr;
    ma.cursor = S->cursor; ma.records = (uint4*)S->records; ma.cap = S->cap; ma.ovf_cap = (uint32_t)(S->ovf_cap / SNK_OVF_SUBLISTS);
    ma.ovf_base = S->slots_per_pass; ma.ovf_bucket = S->ovf_bucket; ma.ovf_cursor = S->ovf_cur;
    ma.hot_tab = S->cursor + NB + 1; ma.hot_thr = msp_hot_thr(S->cap);
    ma.b_lo = b_lo; ma.b_hi = b_hi;
    if (trim_now) {
        ma.quals = (const uint8_t*)S->ft.quals; ma.qstride = S->ft.qstride; ma.min_qual = S->ft.min_qual;
        ma.lens = (const uint16_t*)S->ft.lens; ma.good_out = S->ft.good_out; ma.plan = S->d_fplan;
        SNK_HIP_TRY(hipMemsetAsync(S->d_fplan, 0, 2ull * SNK_MSP_PLAN_SLOTS * 8, st));
    }
    snk_phase_timer kt(st);
    kt.mark();
    int rc;
    if ((rc = snk_launch_msp(S->K, ctx->mlen, st, ma, err, errcap))) return rc;
    kt.mark();
    SNK_HIP_TRY(hipMemsetAsync(S->d_total, 0, 64 * 8, st));
    hipLaunchKernelGGL(seg0_range_kernel, dim3((b_hi - b_lo + 255) / 256), dim3(256), 0, st, S->cursor, b_lo, b_hi, NB, S->cap, S->seg, S->d_total);
    uint32_t h_cur[SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE], h_sub[SNK_OVF_SUBLISTS];
    unsigned long long h_tot64[64];
    std::vector<unsigned long long> h_fplan(2 * SNK_MSP_PLAN_SLOTS);
    SNK_HIP_TRY(hipMemcpyAsync(h_cur, S->ovf_cur, sizeof h_cur, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(h_tot64, S->d_total, sizeof h_tot64, hipMemcpyDeviceToHost, st));
    if (trim_now) SNK_HIP_TRY(hipMemcpyAsync(h_fplan.data(), S->d_fplan, h_fplan.size() * 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if (trim_now) {
        S->h_plan[0] = S->h_plan[1] = 0;
        for (int q = 0; q < SNK_MSP_PLAN_SLOTS; ++q) { S->h_plan[0] += h_fplan[2 * q]; S->h_plan[1] += h_fplan[2 * q + 1]; }
    }
    uint64_t want = 0, mx = 0, tot = 0;
    for (uint32_t q = 0; q < SNK_OVF_SUBLISTS; ++q) { h_sub[q] = h_cur[q * SNK_OVF_CUR_STRIDE]; want += h_sub[q]; if (h_sub[q] > mx) mx = h_sub[q]; }
    for (int q = 0; q < 64; ++q) tot += h_tot64[q];
    if (mx > S->ovf_cap / SNK_OVF_SUBLISTS)
        return snk_fail(SNK_E_NOMEM, err, errcap, "partition pass %u of %u: %llu supermers beyond their buckets' capacity, the overflow list holds %llu (a few minimisers carry a "
                        "large share of the data): more passes (SNK_PARTITION_PASSES) give the list more room", r, S->P, (unsigned long long)want, (unsigned long long)S->ovf_cap);
    S->n_supermers += tot + want;
    S->n_overflow += want;
    S->kernel_ms += kt.ms(0, 1);
    if ((rc = snk_msp_segments(ctx, st, NB, S->cap, S->cursor, (uint4*)S->records, S->slots_per_pass, S->ovf_cap / SNK_OVF_SUBLISTS, S->ovf_bucket, h_sub, S->seg, err, errcap))) return rc;
    // hot minimiser buckets of this range (snk_hot.hip): planned from the range's part of the segment table and expanded NOW, while their
    // records are in the slot array; the count stage counts the virtual buckets behind its ranged launches
    if (S->hots && want) {
        snk_hot hot;
        if ((rc = snk_stage_hot_plan(ctx, st, S->K, S->grouped, S->seg + b_lo, S->seg + NB + b_lo, 2 * NB, 2, b_hi - b_lo, S->cap, &hot, err, errcap))) return rc;
        if (hot.NBv) {
            hipLaunchKernelGGL(vmeta_shift_kernel, dim3((hot.NBv + 255) / 256), dim3(256), 0, st, const_cast<uint2*>(hot.vmeta), hot.NBv, b_lo);      // (the plan numbered the range's buckets from 0)
            if ((rc = snk_stage_hot_expand(ctx, st, S->records, &hot, err, errcap))) return rc;
            S->n_hot += hot.n_hot;
            S->hots->push_back(hot);
        } else snk_stage_hot_drop(&hot);
    }
    return SNK_OK;
}

int snk_stage_partition_compact(snk_ctx* ctx, hipStream_t st, const snk_partition* part, const uint32_t* d_offsets, void* d_out, char* err,
                                size_t errcap) {
    if (part->NB == 0) return SNK_OK;
    hipLaunchKernelGGL(compact_buckets_kernel, dim3((part->NB + 3) / 4), dim3(256), 0, st, (const uint4*)part->records, part->seg, part->NB,
                       d_offsets, (uint4*)d_out, 0u, 0u);
    SNK_HIP_TRY(hipGetLastError());
    SNK_HIP_TRY(snk_sync(st));
    snk_ctx_release_block(ctx, part->records);      // the slot layout is dead once the send buffer is filled
    return SNK_OK;
}

// the same for the buckets OUTSIDE [skip_lo, skip_hi): the rank's own buckets stay where they are and are counted in place
// (snk_shard_step.hip); nothing is waited for, the slot layout stays alive
int snk_stage_partition_compact_remote(snk_ctx* ctx, hipStream_t st, const snk_partition* part, const uint32_t* d_offsets, void* d_out,
                                       uint32_t skip_lo, uint32_t skip_hi, char* err, size_t errcap) {
    if (part->NB == 0) return SNK_OK;
    hipLaunchKernelGGL(compact_buckets_kernel, dim3((part->NB + 3) / 4), dim3(256), 0, st, (const uint4*)part->records, part->seg, part->NB,
                       d_offsets, (uint4*)d_out, skip_lo, skip_hi);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}


// ===================================================================================================================
// The one-pass partition as a JOB that takes its reads slab by slab (snk_dev_stream_*: the reads of a job need not be resident;
// a slab is partitioned while the next one is being decoded / uploaded -- what the reference does when it streams the FASTQ
// chunks through its partitioner, lib/tada/src/cmd_msp.rs:55-69, and re-scans in passes when memory is short,
// MapReduceEngine.h:452-468).  Sized from the caller's upper bound of the job's reads; every slab's launch appends to the same
// bucket slots (the cursors persist), close() builds the segment tables.  Nothing can be run twice here -- the slabs are gone --
// so the overflow list is generous and exceeding it is an error, not a retry.
int snk_partition_open(snk_ctx* ctx, hipStream_t st, uint32_t K, uint32_t NB, unsigned long long n_inst_ub, unsigned long long n_live_ub, bool grouped,
                       uint32_t* status, snk_partition_job* J, char* err, size_t errcap) {
    memset(J, 0, sizeof *J);
    double est_super = 0;
    uint64_t cap64 = 0;
    partition_capacity(ctx, K, NB, n_inst_ub, n_live_ub, grouped, &est_super, &cap64);
    if (cap64 * NB >= (1ull << 40) || cap64 >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "bucket capacity out of range");
    J->K = K; J->NB = NB; J->cap = (uint32_t)cap64; J->grouped = grouped; J->status = status;
    J->ovf_cap = ((uint64_t)(est_super / 6) + (1u << 20) + SNK_OVF_SUBLISTS - 1) / SNK_OVF_SUBLISTS * SNK_OVF_SUBLISTS;
    if (J->ovf_cap >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "supermer overflow list too large");
    int rc;
    void* q;
    if ((rc = snk_ctx_alloc(ctx, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, &q, err, errcap))) return rc; J->cursor = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, 4ull * NB * 8 + 64, &q, err, errcap))) return rc; J->seg = (uint64_t*)q;
    if ((rc = snk_ctx_alloc(ctx, 64 * 8, &q, err, errcap))) return rc; J->d_total = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, 2ull * SNK_MSP_PLAN_SLOTS * 8, &q, err, errcap))) return rc; J->d_plan = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, ((size_t)NB * J->cap + 2 * J->ovf_cap) * 32 + 64, &J->records, err, errcap))) return rc;
    if ((rc = snk_ctx_alloc(ctx, J->ovf_cap * 4 + 64, &q, err, errcap))) return rc; J->ovf_bucket = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4 + 64, &q, err, errcap))) return rc; J->ovf_cur = (uint32_t*)q;
    SNK_HIP_TRY(hipMemsetAsync(J->cursor, 0, (NB + 1 + SNK_MSP_HOT_TAB) * 4ull, st));
    SNK_HIP_TRY(hipMemsetAsync(J->ovf_cur, 0, SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE * 4, st));
    SNK_HIP_TRY(hipMemsetAsync(J->d_plan, 0, 2ull * SNK_MSP_PLAN_SLOTS * 8, st));
    return SNK_OK;
}

// one slab: good_len = its good lengths (u16 per read, already computed), or ft = the fused trim's inputs (then the lengths are written
// to ft->good_out by the kernel).  Nothing is waited for.
int snk_partition_add(snk_ctx* ctx, hipStream_t st, snk_partition_job* J, const snk_dev_reads* in, const uint16_t* good_len, const snk_fused_trim* ft, char* err,
                      size_t errcap) {
    (void)ctx;
    if (in->n_reads == 0) return SNK_OK;
    snk_msp_args ma;
    memset(&ma, 0, sizeof ma);
    ma.rows = (const uint32_t*)in->rows; ma.row_words = in->row_words; ma.read_len = in->read_len; ma.good_len = good_len; ma.bc = (const int32_t*)in->bc;
    ma.ign_bc_below = in->ign_bc_below; ma.read_index_base = in->read_index_base; ma.n_reads = in->n_reads; ma.NB = J->NB;
    ma.group = J->grouped ? (const uint32_t*)in->group : nullptr;
    ma.cursor = J->cursor; ma.records = (uint4*)J->records; ma.cap = J->cap; ma.ovf_cap = (uint32_t)(J->ovf_cap / SNK_OVF_SUBLISTS);
    ma.ovf_base = (uint64_t)J->NB * J->cap; ma.ovf_bucket = J->ovf_bucket; ma.ovf_cursor = J->ovf_cur;
    ma.hot_tab = J->cursor + J->NB + 1; ma.hot_thr = msp_hot_thr(J->cap);
    if (ft) {
        ma.good_len = ft->good_out; ma.quals = (const uint8_t*)ft->quals; ma.qstride = ft->qstride; ma.min_qual = ft->min_qual;
        ma.lens = (const uint16_t*)ft->lens; ma.good_out = ft->good_out; ma.plan = J->d_plan;
    }
    int rc = snk_launch_msp(J->K, ctx->mlen, st, ma, err, errcap);
    if (rc) return rc;
    if (!ft) {
        // the sizing figures the fused kernel adds up itself: instances and contributing reads of this slab
        unsigned long long* two;
        void* q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; two = (unsigned long long*)q;
        if ((rc = snk_launch_msp_plan(st, good_len, in->n_reads, J->K, two, err, errcap))) return rc;
        hipLaunchKernelGGL(plan_add_kernel, dim3(1), dim3(64), 0, st, two, J->d_plan);
        SNK_HIP_TRY(hipGetLastError());
    }
    J->n_reads += in->n_reads;
    ++J->n_slabs;
    return SNK_OK;
}

int snk_partition_close(snk_ctx* ctx, hipStream_t st, snk_partition_job* J, snk_partition* out, unsigned long long h_plan[2], char* err, size_t errcap) {
    memset(out, 0, sizeof *out);
    const uint32_t NB = J->NB;
    SNK_HIP_TRY(hipMemsetAsync(J->d_total, 0, 64 * 8, st));
    hipLaunchKernelGGL(seg0_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, J->cursor, NB, J->cap, J->seg, J->d_total);
    uint32_t h_novf = 0;
    uint32_t h_sub[SNK_OVF_SUBLISTS];
    uint32_t h_cur[SNK_OVF_SUBLISTS * SNK_OVF_CUR_STRIDE];      // the cursors as they lie on the device, one per 128 bytes
    unsigned long long h_tot64[64];
    std::vector<unsigned long long> h_fplan(2 * SNK_MSP_PLAN_SLOTS);
    SNK_HIP_TRY(hipMemcpyAsync(h_cur, J->ovf_cur, sizeof h_cur, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(h_tot64, J->d_total, sizeof h_tot64, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(h_fplan.data(), J->d_plan, h_fplan.size() * 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    for (uint32_t q = 0; q < SNK_OVF_SUBLISTS; ++q) h_sub[q] = h_cur[q * SNK_OVF_CUR_STRIDE];
    unsigned long long h_total = 0;
    for (int q = 0; q < 64; ++q) h_total += h_tot64[q];
    h_plan[0] = h_plan[1] = 0;
    for (int q = 0; q < SNK_MSP_PLAN_SLOTS; ++q) { h_plan[0] += h_fplan[2 * q]; h_plan[1] += h_fplan[2 * q + 1]; }
    uint64_t want = 0, mx = 0;
    for (uint32_t q = 0; q < SNK_OVF_SUBLISTS; ++q) { want += h_sub[q]; if (h_sub[q] > mx) mx = h_sub[q]; }
    h_novf = (uint32_t)(want > 0xFFFFFFFFull ? 0xFFFFFFFFull : want);
    h_total += want;
    if (mx > J->ovf_cap / SNK_OVF_SUBLISTS)
        return snk_fail(SNK_E_NOMEM, err, errcap, "streamed partition: %u supermers beyond their buckets' capacity, the overflow list holds %llu (the job's read total was "
                        "underestimated, or a few minimisers carry a large share of the data): run it resident or with a larger total", h_novf, (unsigned long long)J->ovf_cap);
    int rc;
    if ((rc = snk_msp_segments(ctx, st, NB, J->cap, J->cursor, (uint4*)J->records, (uint64_t)NB * J->cap, J->ovf_cap / SNK_OVF_SUBLISTS, J->ovf_bucket, h_sub, J->seg, err, errcap))) return rc;
    if (h_novf) hipLaunchKernelGGL(cursor_exact_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, J->seg, NB, J->cursor);
    out->NB = NB;
    out->cap = J->cap;
    out->nseg = h_novf ? 2u : 1u;
    out->n_overflow = h_novf;
    out->n_supermers = h_total;
    out->records = J->records;
    out->cursor = J->cursor;
    out->seg = J->seg;
    out->kernel_ms = 0.f;
    return SNK_OK;
}
