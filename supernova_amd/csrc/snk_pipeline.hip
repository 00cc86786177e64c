// snk_pipeline.hip -- orchestration of the device-resident count+graph path behind the C ABI.
//
// snk_dev_count_graph is the MI355X replacement of the body of buildReadQGraph48
// (lib/assembly/src/paths/long/BuildReadQGraph48.cc:1688-1774, pPaths == nullptr branch):
//   createDict (:218-325)  ->  trim, MSP partition, LDS count/filter, sort, index, prune
//   buildEdges (:514-541)  ->  links, list ranking, canonical unitigs
// and of tada's MSP -> SHARD_ASM -> MAIN_ASM_SN chain (lib/tada/src/cmd_msp.rs:38-80,
// cmd_shard_asm.rs:37-94, cmd_main_asm.rs:25-89).
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdlib.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_graph.h"
#include "snk_kernels.h"
#include "snk_stages.h"

namespace {

__global__ void widen_offsets_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// A fingerprint of the reads of a resident call: 4096 sixteen-byte samples of the packed rows (and of the quality rows or good lengths)
// at evenly spread places, mixed with their place and added up -- order independent, one small workgroup.  What it is for: the sizing
// history of a context (distinct k-mers per instance -> bucket size, retained share -> chunk and region sizes) is a property of the DATA
// SET; a context that meets other data of the same size used to start from the old data's figures -- 0.6 % errors after the bench's
// 0.2 %: first call 371 ms instead of 150 (1.2 M of 2 M buckets hash-split), 28x coverage after 56x: 644 ms (the count regions overflow
// and the kernel runs again).  With another fingerprint the call forgets the history and looks at its first buckets instead.
__global__ void __launch_bounds__(256) input_fp_kernel(const uint4* __restrict__ rows16, uint64_t n16, const uint4* __restrict__ aux16, uint64_t m16,
                                                       unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int s = 0; s < 16; ++s) {
        const uint64_t q = (uint64_t)(threadIdx.x * 16 + s);
        if (n16) {
            const uint64_t at = (n16 / 4096) * q + (q * 0x9E37u) % (n16 / 4096 + 1);
            if (at < n16) { const uint4 v = rows16[at]; acc += snk_mix64(((unsigned long long)v.x << 32 | v.y) ^ snk_mix64(((unsigned long long)v.z << 32 | v.w) + at)); }
        }
        if (m16) {
            const uint64_t at = (m16 / 4096) * q + (q * 0x79B9u) % (m16 / 4096 + 1);
            if (at < m16) { const uint4 v = aux16[at]; acc += snk_mix64(((unsigned long long)v.x << 32 | v.y) ^ snk_mix64(((unsigned long long)v.z << 32 | v.w) + at + 0x51ED27ull)); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

typedef snk_phase_timer phase_timer;


}  // namespace

// everything behind the count stage: bucket-local (or global) graph, result fields, phase times.  tm: marks 0..3 are the caller's
// (start, trim, plan, partition).
static int graph_tail(snk_ctx* ctx, hipStream_t st, const snk_params* p, uint32_t K, bool grouped, bool local_graph, uint64_t n_reads, unsigned long long h_ninst,
                      snk_table& tab, const snk_partition& part, snk_dev_result* out, phase_timer& tm, char* err, size_t errcap) {
    int rc;
    void* records = part.records;
    if (h_ninst) { ctx->claim_ratio = (ctx->count_screen && !grouped && ctx->screen_ratio > 0.0) ? ctx->screen_ratio : (double)tab.distinct / (double)h_ninst; ctx->claim_ratio_reads = n_reads; ctx->claim_ratio_k = K * 2 + (grouped ? 1u : 0u) + 256u * ctx->mlen;
                   ctx->retain_ratio = (double)tab.n / (double)h_ninst; }
    snk_ctx_release_block(ctx, records);       // the fixed-capacity supermer slots: the graph stage may reuse the memory
    const uint64_t n_kmers = tab.n;
    out->buckets_split = tab.buckets_split;
    out->max_slots_used = tab.max_slots_used;
    out->n_kmers = n_kmers;
    out->keys = tab.keys;
    tm.mark();  // 4 (count+gather) and 5 (sort) are reported from the stage's own events
    tm.mark();

    // ---- prune + unitigs
    snk_graph_out go;
    if (local_graph) {
        snk_u128* keys_final = nullptr;
        rc = snk_local_graph(ctx, st, K, &tab, p->min_freq > 1 ? 1u : 0u, !(p->flags & SNK_F_NO_GRAPH),
                             !(p->flags & SNK_F_UNSORTED_TABLE), grouped, &go, &keys_final, out->graph_ms, err, errcap);
        if (rc) return rc;
        out->keys = keys_final;
        out->n_boundary = go.n_boundary;
        out->n_fragments = go.n_fragments;
    } else {
        rc = snk_graph_build(ctx, st, K, tab.keys, tab.vals, n_kmers, p->min_freq > 1 ? 1u : 0u, !(p->flags & SNK_F_NO_GRAPH), &go,
                             err, errcap);
        if (rc) return rc;
    }
    tm.mark();  // 6
    SNK_HIP_TRY(snk_sync(st));
    out->counts = go.counts;
    out->ctx = go.ctx;
    out->spectrum = go.spectrum;
    out->spectrum_bins = go.spectrum_bins;
    out->n_unitigs = go.n_unitigs;
    out->unitig_total_bases = go.total_bases;
    out->unitig_off = go.unitig_off;
    out->unitig_bases = go.unitig_bases;
    out->unitig_group = go.unitig_group;
    out->n_circles = go.n_circles;
    out->rank_rounds = go.rank_rounds;
    out->phase_ms[0] = tm.ms(0, 1);
    out->phase_ms[1] = tm.ms(1, 2);
    out->phase_ms[2] = tm.ms(2, 3);
    out->phase_ms[3] = tab.count_ms;
    out->phase_ms[4] = tab.sort_ms;
    out->phase_ms[5] = tm.ms(5, 6);
    out->phase_ms[7] = tm.ms(0, 6);
    out->kernel_ms[0] = 0.f;
    out->kernel_ms[1] = part.kernel_ms;
    out->kernel_ms[2] = tab.count_kernel_ms;
    out->scratch_bytes = ctx->peak_alloc;
    return SNK_OK;
}

extern "C" int snk_dev_count_graph(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, snk_dev_result* out,
                                   void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 8) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule tells up to eight distinct barcodes apart (min_bc <= 8)", p->min_bc);
    snk_set_mlen(ctx, p);
    if (in->n_reads && (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: bad rows/read_len (read_len <= 256)");
    if (in->n_reads && !in->good_len && !in->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: need quals or good_len");
    const bool grouped = (p->flags & SNK_F_GROUPED) != 0;
    if (grouped) {
        if (p->K != 48) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED: the group id rides in the 32 key bits that are free at K=48 only");
        if (!in->group) return snk_fail(SNK_E_ARG, err, errcap, "SNK_F_GROUPED: snk_dev_reads.group is NULL");
        if (p->min_bc > 0 && in->bc) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED: per-group graphs use the frequency rule only (min_bc = 0)");
        if ((p->flags & SNK_F_GLOBAL_GRAPH) || snk_opt_u32("global_graph", 0)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED needs the bucket-local graph stage");
    }
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    ctx->arena_legacy = false;
    snk_ctx_release_scratch(ctx);
    memset(out, 0, sizeof *out);
    const uint32_t K = p->K;
    const uint64_t n_reads = in->n_reads;
    out->n_reads = n_reads;
    phase_timer tm(st);
    tm.mark();  // 0

    // ---- K1 trim: inside the partition kernel when the quality rows allow it (its loads ride under the slot reservations),
    // else its own streaming kernel
    const uint16_t* good_len = (const uint16_t*)in->good_len;
    const bool fused = n_reads && snk_fused_trim_ok(in);
    snk_fused_trim ft;
    if (!good_len && n_reads) {
        void* gl = nullptr;
        int rc = snk_ctx_alloc(ctx, n_reads * 2 + 2, &gl, err, errcap);
        if (rc) return rc;
        if (fused) { ft.quals = in->quals; ft.qstride = in->qstride; ft.lens = in->lens; ft.min_qual = p->min_qual; ft.good_out = (uint16_t*)gl; }
        else {
            rc = snk_dev_trim(ctx, in->quals, in->qstride, in->lens, in->read_len, n_reads, K, p->min_qual, gl, st);
            if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        }
        good_len = (const uint16_t*)gl;
    }
    out->good_len = good_len;
    tm.mark();  // 1

    // ---- K3/K4 minimiser partition in ONE pass: exact k-mer instance count -> expected supermers -> fixed bucket capacity
    // (fused trim: the count is not known yet; every base of every read is the bound, a few per cent above what the trim leaves)
    uint32_t* status = nullptr;
    {
        void* q;
        int rc;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; status = (uint32_t*)q;
    }
    SNK_HIP_TRY(hipMemsetAsync(status, 0, 64, st));
    unsigned long long h_plan[2] = {0, 0};
    int rc = SNK_OK;
    if (fused) {
        const unsigned long long kpr = in->read_len >= K ? in->read_len - K + 1 : 0;
        h_plan[0] = n_reads * kpr;
        h_plan[1] = n_reads;
    } else
        rc = snk_stage_partition_plan(ctx, st, K, good_len, n_reads, h_plan, err, errcap);
    if (rc) return rc;
    unsigned long long h_ninst = h_plan[0];
    out->n_instances = h_ninst;
    // instances per bucket: sized so that the DISTINCT k-mers of a bucket fit the LDS table (1216 claims).  At 56x coverage and
    // 0.2 % errors 5000 instances hold ~800 distinct k-mers; per-barcode groups see every locus once or twice, so nearly every
    // instance is distinct there.  Error-rich or shallow data have more distinct k-mers per instance: with the default size
    // nearly every bucket would overflow its table and be counted in two to four hash-split sub-passes (0.6 % errors: count
    // 46 -> 133 ms).  The ratio is a property of the data set: the previous call's is used if there is one, else the count stage
    // looks at its first 1/64 of the buckets and asks for a second partition when they overflow as a rule.
    // ---- is this the data set the context's sizing history was made on?  (one 8-byte read-back: ~40 us)
    bool same_data = true;
    if (n_reads && p->n_buckets == 0 && snk_opt_u32("input_fp", 1)) {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc;
        unsigned long long* d_fp = (unsigned long long*)q;
        SNK_HIP_TRY(hipMemsetAsync(d_fp, 0, 8, st));
        const uint64_t n16 = (((uintptr_t)in->rows & 15u) == 0) ? n_reads * (uint64_t)in->row_words / 4 : 0;
        const void* aux = in->quals ? in->quals : in->good_len;
        const uint64_t m16 = (aux && ((uintptr_t)aux & 15u) == 0) ? (in->quals ? n_reads * (uint64_t)in->qstride : n_reads * 2ull) / 16 : 0;
        hipLaunchKernelGGL(input_fp_kernel, dim3(1), dim3(256), 0, st, (const uint4*)in->rows, n16, (const uint4*)aux, m16, d_fp);
        unsigned long long h_fp = 0;
        SNK_HIP_TRY(hipMemcpyAsync(&h_fp, d_fp, 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        h_fp ^= snk_mix64(n_reads * 0x9E3779B97F4A7C15ull + in->read_len);
        same_data = ctx->have_input_fp && ctx->last_input_fp == h_fp;
        ctx->last_input_fp = h_fp; ctx->have_input_fp = true;
        if (!same_data) { ctx->last_n_kmers = 0; ctx->last_bnd = 0; ctx->last_region_max = 0; }       // (the count regions' and the boundary index's sizes were that data's)
    }
    // The count kernel has two ways to keep its probe loops supplied with free slots (snk_count.hip): a margin of one round of every wave
    // (1216 of 2048 slots usable, nothing to pay per round) or booked slots (15/16 usable, one LDS atomic round trip per wave and round:
    // 45.1 instead of 43.0 ms on the bench model).  Data whose tables run full -- sequencing errors, per-barcode groups -- are counted the
    // second way: fewer, fuller buckets (1.5 % errors: 8.4 M -> 5.6 M buckets, 218 -> 188 ms; 0.6 %: 149 -> 136; groups: 183 -> 177).
    // SNK_COUNT_TIGHT = 0 never, = n always with n usable slots.
    ctx->count_tight = 0;
    const uint32_t plain_target = K == 48 ? 5000u : 3500u;
    auto tight_for = [&](double ratio) -> uint32_t {
        const uint32_t tries = snk_opt_u32("tight_tries", 48) << 16;
        if (snk_opt_is_set("count_tight")) {
            const uint32_t v = snk_opt_u32("count_tight", 0);
            return v ? (std::min(std::max(v, 256u), snk_count_slots(K) - 64u) | tries) : 0u;
        }
        const bool full = grouped || (ratio > 0.0 && 0.65 * (double)snk_count_limit(K, 0u, 0u) / ratio < (double)plain_target);
        return full ? ((snk_count_slots(K) - snk_count_slots(K) / 16u) | tries) : 0u;
    };
    // (per-barcode groups: nearly every instance is a distinct entry, the bucket IS the table: three quarters of its capacity on average)
    // (... unless the bit filter in front of the table is on -- min_freq >= 2: then the table only sees the (group, k-mer) pairs that can be retained,
    // one in ten, and a bucket is as large as one batch of 512 records and ten instances per lane allow: beyond 6000 buckets start to fall out
    // of the filter -- 92.9 ms at 4800, 92.2 at 5600, 94.6 at 6400, `profiles/r05_count_screen_groups.log`)
    const bool group_screen = grouped && snk_opt_u32("count_screen", 1) != 0 && p->min_freq >= (snk_opt_u32("count_screen", 1) >= 2 ? 2u : 3u);
    auto default_target_now = [&]() -> uint32_t { return grouped ? ((group_screen && ctx->count_tight) ? 5200u : (uint32_t)(0.74 * snk_count_limit(K, 1u, ctx->count_tight))) : plain_target; };
    const bool target_forced = snk_opt_is_set("target_inst");
    // ... and the RETAINED k-mers of a bucket are one chunk of the bucket-local graph stage, whose one-wave kernels hold 256 of them
    // (larger chunks take the slower big-chunk variants): at half the coverage twice as many k-mers survive per instance, every other
    // chunk was over the line and the graph stage took 81 instead of ~58 ms.  From the previous call's retained share: chunks of ~180 (28x coverage, with merged chunks behind it: 153.2 ms at 120, 149.5 at 150, 147.7 at 180, 147.5 at 210).
    const double retain = (same_data && ctx->retain_ratio > 0.0 && ctx->claim_ratio_reads == n_reads && ctx->claim_ratio_k == K * 2 + (grouped ? 1u : 0u) + 256u * ctx->mlen) ? ctx->retain_ratio : 0.0;
    auto target_for = [&](double ratio) -> uint32_t {
        const uint32_t default_target = default_target_now();
        if (target_forced) return snk_opt_u32("target_inst", default_target);
        if (ctx->count_screen && !grouped) return snk_opt_u32("screen_target", 4000);
        if (retain > 0.0 && (!grouped || group_screen)) {       // (groups behind the bit filter: buckets of 5200 instances, unless that many would retain more than a graph chunk holds)
            const double t = (double)snk_opt_u32("chunk_kmers", 180) / retain;
            if (t < (double)default_target) {
                uint32_t tt = t < 600.0 ? 600u : (uint32_t)t;
                if (ratio > 0.0) {           // the tighter of the two limits
                    const double lim = (double)snk_count_limit(K, 0u, ctx->count_tight);
                    if (0.65 * lim / ratio < (double)default_target) {
                        const double t2 = 0.01 * snk_opt_u32("bucket_fill_pct", 50) * lim / ratio;
                        if (t2 < (double)tt) tt = t2 < 600.0 ? 600u : (uint32_t)t2;
                    }
                }
                return tt;
            }
        }
        if (!(ratio > 0.0)) return default_target;
        // measured: the default size is right while the tables run up to ~65 % full on average (the bench model: 800 of 1216); data
        // that would fill them further do best at ~50 % (0.6 % errors: 239 ms with the default size, 186 at 80 %, 154 at 50 %)
        const double lim = (double)snk_count_limit(K, grouped ? 1u : 0u, ctx->count_tight);
        if (0.65 * lim / ratio >= (double)default_target) return default_target;
        const double t = 0.01 * snk_opt_u32("bucket_fill_pct", 50) * lim / ratio;
        return t >= (double)default_target ? default_target : (t < 600.0 ? 600u : (uint32_t)t);
    };
    const bool have_hint = same_data && ctx->claim_ratio > 0.0 && ctx->claim_ratio_reads == n_reads && ctx->claim_ratio_k == K * 2 + (grouped ? 1u : 0u) + 256u * ctx->mlen;
    double ratio = have_hint ? ctx->claim_ratio : 0.0;
    const bool adaptive = p->n_buckets == 0 && !target_forced && !grouped && snk_opt_u32("adaptive_buckets", 1) != 0;      // (the per-barcode default is tuned at ratio ~1)
    uint32_t NB = 0;
    snk_partition part;
    snk_table tab;
    void* records = nullptr;
    const bool local_graph = !(p->flags & SNK_F_GLOBAL_GRAPH) && !snk_opt_u32("global_graph", 0);
    const uint64_t mark = ctx->alloc_serial;
    const unsigned long long ub_inst = h_plan[0], ub_live = h_plan[1];
    for (int pass = 0; pass < 2; ++pass) {
        NB = p->n_buckets;
        ctx->count_tight = tight_for(adaptive ? ratio : (have_hint ? ctx->claim_ratio : 0.0));
        // ungrouped reads whose tables run very full (1.5 % errors: 0.41 distinct k-mers per instance, most of them seen once or twice): the bit
        // filter of the per-barcode groups in front of a 1024-slot table -- the table then sees what can be retained and the bucket is as large
        // as one batch of records and ten instances per lane allow.  SNK_COUNT_SCREEN_NG: 0 never, 2 always, default: ratio above 0.3
        // (0.6 % errors, ratio 0.21, lose with it: a fifth of their instances are singletons, the first pass costs more than it saves)
        {
            const double r_now = adaptive ? ratio : (have_hint ? ctx->claim_ratio : 0.0);
            const uint32_t ng = snk_opt_u32("count_screen_ng", 1);
            ctx->count_screen = (!grouped && K == 48 && p->min_freq >= 3 && (!in->bc || p->min_bc <= 2) && ng && (ng >= 2 || r_now > 0.01 * snk_opt_u32("screen_ratio_pct", 30))) ? 3u : 0u;
            if (snk_opt_is_set("count_tight") && snk_opt_u32("count_tight", 1) == 0u) ctx->count_screen = 0;      // (the filter comes with booked slots)
            if (ctx->count_screen && !ctx->count_tight) ctx->count_tight = (snk_count_slots(K) - snk_count_slots(K) / 16u) | (snk_opt_u32("tight_tries", 48) << 16);
            if (ctx->count_screen && r_now > 0.0) ctx->screen_ratio = r_now;      // (what the screened call reports is the table's view: the decision keeps the ratio it was made on)
            ctx->last_count_limit = snk_count_limit(K, grouped ? 1u : 0u, ctx->count_tight);
            if ((ctx->count_screen || (group_screen && ctx->count_tight)) && K == 48) ctx->last_count_limit = std::min(ctx->last_count_limit, snk_count_screen_limit());
        }
        if (NB == 0) {
            const uint32_t target = target_for(adaptive ? ratio : 0.0);
            uint64_t nb = (ub_inst + target - 1) / target;
            const uint64_t nb_max = 1ull << 25;      // (2^23 until round 6: at 800 M reads that is 9700 instances per bucket, a third of the buckets split; 1.2 B reads as per-barcode graphs want 23.5 M)
            if (nb < 1) nb = 1;
            if (nb > nb_max) nb = nb_max;
            NB = (uint32_t)nb;
        }
        {
            // a caller's bucket count is honoured down to ~1 M k-mer instances per bucket: a bucket is counted by ONE
            // workgroup that re-reads all its records in every hash-split sub-pass, so 20 M instances in one bucket would be
            // thousands of passes over a million records (finite, but minutes)
            const uint64_t nb_floor = (ub_inst >> 20) + 1;
            if (NB < nb_floor) NB = (uint32_t)nb_floor;
        }
        out->n_buckets = NB;
        if (pass == 0) tm.mark();  // 2
        h_plan[0] = ub_inst; h_plan[1] = ub_live;
        // ---- a job whose slots would not fit: bucket-range passes over one slot array (snk_stages.h); the count stage's range hook
        //      partitions range r right before range r is counted
        const uint32_t n_passes = (NB >= 2 && !snk_opt_u32("msp_dense", 0)) ? std::min<uint32_t>(snk_partition_passes_needed(ctx, K, NB, ub_inst, ub_live, grouped), NB) : 1u;
        ctx->last_partition_passes = n_passes;
        if (n_passes > 1) {
            snk_partition_passes PS;
            if ((rc = snk_partition_passes_open(ctx, st, K, in, good_len, fused ? &ft : nullptr, NB, n_passes, ub_inst, ub_live, grouped, &PS, err, errcap))) return rc;
            if (pass == 0) tm.mark();  // 3 (the partition's time is inside the count stage's here)
            std::vector<snk_hot> pass_hots;
            PS.hots = &pass_hots;
            snk_count_ranges rgs{n_passes, PS.bounds, snk_partition_passes_run, &PS, true, &pass_hots};
            rgs.finished = [](void* u) {
                snk_partition_passes* S = static_cast<snk_partition_passes*>(u);
                snk_ctx_release_block(S->ctx, S->records); S->records = nullptr;
                snk_ctx_release_block(S->ctx, S->ovf_bucket); S->ovf_bucket = nullptr;
            };
            // (the first range's first buckets are the pilot here too: data whose tables run full -- error-rich reads -- are partitioned again into
            // smaller buckets and get the count kernel they want, as in the one-pass path below)
            snk_count_pilot pilot_p{0.0, nullptr, nullptr};
            const bool want_pilot_p = adaptive && pass == 0 && !have_hint;
            rc = snk_stage_count_table(ctx, st, K, PS.records, PS.seg, PS.seg + NB, 2 * NB, 2u, NB, p->min_freq, (in->bc && !grouped) ? p->min_bc : 0u, grouped ? 1u : 0u,
                                       ub_inst, status, !local_graph, &tab, err, errcap, &rgs, want_pilot_p ? &pilot_p : nullptr, nullptr, local_graph, nullptr);
            if (rc == SNK_RETARGET) {
                const unsigned long long inst_now = PS.h_plan[0] ? PS.h_plan[0] : ub_inst;
                snk_ctx_release_since(ctx, mark, nullptr, 0);
                SNK_HIP_TRY(hipMemsetAsync(status, 0, 64, st));
                ratio = inst_now ? pilot_p.per_bucket * (double)NB / (double)inst_now : 0.0;
                out->repartitioned = 1;
                continue;
            }
            if (rc) return rc;
            h_ninst = PS.h_plan[0];
            memset(&part, 0, sizeof part);
            part.NB = NB; part.cap = PS.cap; part.nseg = 2; part.n_overflow = (uint32_t)std::min<uint64_t>(PS.n_overflow, 0xFFFFFFFFull); part.n_supermers = PS.n_supermers;
            part.records = PS.records; part.cursor = PS.cursor; part.seg = PS.seg; part.kernel_ms = PS.kernel_ms;
            out->n_instances = h_ninst;
            out->n_supermers = part.n_supermers;
            out->n_overflow = part.n_overflow;
            out->n_hot_buckets = PS.n_hot;
            break;
        }
        rc = snk_stage_partition(ctx, st, K, in, good_len, NB, h_plan[0], h_plan[1], grouped, status, &part, err, errcap, nullptr, fused ? h_plan : nullptr,
                                 fused ? &ft : nullptr, true);
        if (rc) return rc;
        h_ninst = h_plan[0];               // (fused trim: now the exact count)
        out->n_instances = h_ninst;
        records = part.records;
        out->n_supermers = part.n_supermers;
        out->n_overflow = part.n_overflow;
        if (pass == 0) tm.mark();  // 3

        // ---- hot minimiser buckets (repeat families, homopolymer runs): re-partitioned by k-mer hash (snk_hot.hip)
        snk_hot hot;
        if ((rc = snk_stage_hot(ctx, st, K, grouped, &part, &hot, err, errcap))) return rc;
        out->n_hot_buckets = hot.n_hot;
        // ---- K5-K8 count + filter + gather (+ sort for the global graph stage)
        snk_count_pilot pilot{0.0, nullptr, nullptr};
        const bool want_pilot = adaptive && pass == 0 && !have_hint;
#ifdef SNK_PROBES
        // measurement aid: SNK_OVERLAP_PROBE = 1: the partition kernel once more (into scratch) on a second stream NEXT TO the count
        // kernel; 2: the same launch alone (waited for before the count starts); +4: the second stream has high priority;
        // SNK_OVERLAP_PROBE_DBG = the relaunched kernel's dbg mode (1 no record stores, 2 no slot atomics, 3 scan only)
        const uint32_t oprobe = snk_opt_u32("overlap_probe", 0);
        hipStream_t s2 = nullptr, s2hi = nullptr;      // (a measurement aid of tuning builds: created per probed call, destroyed below)
        hipStream_t sp2 = nullptr;
        hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
        if (oprobe) {
            { SNK_HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); SNK_HIP_TRY(hipStreamCreateWithPriority(&s2hi, hipStreamNonBlocking, hi)); }
            sp2 = (oprobe & 4u) ? s2hi : s2;
            for (auto& e : pe) SNK_HIP_TRY(hipEventCreate(&e));
            SNK_HIP_TRY(hipEventRecord(pe[0], st));
            SNK_HIP_TRY(hipStreamWaitEvent(sp2, pe[0], 0));
            SNK_HIP_TRY(hipEventRecord(pe[1], sp2));
            if ((rc = snk_probe_relaunch_msp(ctx, sp2, snk_opt_u32("overlap_probe_dbg", 0), err, errcap))) return rc;
            SNK_HIP_TRY(hipEventRecord(pe[2], sp2));
            if ((oprobe & 3u) == 2u) SNK_HIP_TRY(hipStreamSynchronize(sp2));
        }
#endif
        rc = snk_stage_count_table(ctx, st, K, records, part.seg, part.seg + NB, 2 * NB, part.nseg, NB, p->min_freq, (in->bc && !grouped) ? p->min_bc : 0u, grouped ? 1u : 0u,
                                   h_ninst, status, !local_graph, &tab, err, errcap, nullptr, want_pilot ? &pilot : nullptr, part.gidx, local_graph, &hot);
#ifdef SNK_PROBES
        if (oprobe) {
            SNK_HIP_TRY(hipStreamSynchronize(sp2));
            float pm = 0.f;
            (void)hipEventElapsedTime(&pm, pe[1], pe[2]);
            fprintf(stderr, "[snk overlap probe] mode %u: partition kernel (first launch, alone) %.2f ms; relaunched %s %.2f ms; count kernel %.2f ms; count stage %.2f ms\n", oprobe,
                    part.kernel_ms, (oprobe & 3u) == 2u ? "alone" : "next to the count kernel", pm, tab.count_kernel_ms, tab.count_ms);
            for (auto& e : pe) (void)hipEventDestroy(e);
            (void)hipStreamDestroy(s2); (void)hipStreamDestroy(s2hi);
        }
#endif
        if (rc == SNK_RETARGET) {
            // everything since the partition goes back to the arena; the good lengths and the status words stay
            snk_ctx_release_since(ctx, mark, nullptr, 0);
            SNK_HIP_TRY(hipMemsetAsync(status, 0, 64, st));
            ratio = h_ninst ? pilot.per_bucket * (double)NB / (double)h_ninst : 0.0;
            out->repartitioned = 1;
            continue;
        }
        if (rc) return rc;
        break;
    }
    return graph_tail(ctx, st, p, K, grouped, local_graph, n_reads, h_ninst, tab, part, out, tm, err, errcap);
}

// ---------------------------------------------------------------------------------------------------------------------
// Streamed input (VERDICT r3 missing #2): the reads of a job arrive slab by slab -- as the FASTH decoder delivers them -- and are
// partitioned as they arrive; a slab's buffers can be reused as soon as the stream has passed its launch, so the job's reads are never
// resident as a whole and the ingest of slab i + 1 overlaps the partition of slab i.  The reference streams its FASTQ chunks into the
// partitioner the same way (lib/tada/src/cmd_msp.rs:55-69).  Same results as snk_dev_count_graph on the concatenation, bit for bit.
namespace {
struct stream_job {
    snk_params p;
    uint32_t K = 0, read_len = 0, row_words = 0, NB = 0;
    bool grouped = false, has_bc = false, open = false;
    uint64_t total_ub = 0;
    snk_partition_job J;
    uint32_t* status = nullptr;
    uint16_t* good_len = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;       // around every slab's launches: the partition time of the job
    double t_begin = 0;
};
void stream_job_free(void* q) {
    stream_job* j = static_cast<stream_job*>(q);
    if (!j) return;
    for (auto& e : j->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete j;
}
void stream_job_invalidate(void* q) {
    if (q) static_cast<stream_job*>(q)->open = false;
}
}  // namespace

extern "C" int snk_dev_stream_begin(snk_ctx* ctx, const snk_params* p, uint32_t read_len, uint64_t total_reads_ub, int has_bc, void* stream, char* err, size_t errcap) {
    if (!ctx || !p) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_begin: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 8) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule tells up to eight distinct barcodes apart (min_bc <= 8)", p->min_bc);
    snk_set_mlen(ctx, p);
    ctx->count_tight = 0;          // (a streamed job cannot partition again: the default kernel and its bucket rule)
    ctx->count_screen = 0;
    if (read_len == 0 || read_len > 256 || total_reads_ub == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_begin: read_len 1..256 and an upper bound of the job's reads are needed");
    if ((p->flags & SNK_F_GROUPED)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_stream_begin: per-group graphs take their reads resident (snk_dev_count_graph)");
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    ctx->arena_legacy = false;
    snk_ctx_release_scratch(ctx);
    if (ctx->stream_job) { stream_job_free(ctx->stream_job); ctx->stream_job = nullptr; }
    stream_job* j = new stream_job();
    ctx->stream_job = j;
    ctx->stream_job_free = stream_job_free;
    ctx->stream_job_invalidate = stream_job_invalidate;
    j->p = *p; j->K = p->K; j->read_len = read_len; j->row_words = (read_len + 15) / 16; j->has_bc = has_bc != 0; j->total_ub = total_reads_ub;
    const uint32_t K = p->K;
    const unsigned long long kpr = read_len >= K ? read_len - K + 1 : 0;
    const unsigned long long ub_inst = total_reads_ub * kpr;
    // bucket count: as snk_dev_count_graph sizes it; the distinct-k-mers-per-instance ratio of the previous call on this context (same
    // read total) is used if there is one -- a streamed job cannot look at its first buckets and partition again, its slabs are gone:
    // error-rich data without that history are counted in hash-split sub-passes (slower, same result)
    uint32_t NB = p->n_buckets;
    if (NB == 0) {
        uint32_t target = K == 48 ? 5000u : 3500u;
        if (snk_opt_is_set("target_inst")) target = snk_opt_u32("target_inst", target);
        else if (ctx->claim_ratio > 0.0 && ctx->claim_ratio_reads == total_reads_ub && ctx->claim_ratio_k == K * 2 + 256u * ctx->mlen) {
            const double lim = (double)snk_count_limit(K, 0u, ctx->count_tight);
            if (0.65 * lim / ctx->claim_ratio < (double)target) {
                const double t = 0.01 * snk_opt_u32("bucket_fill_pct", 50) * lim / ctx->claim_ratio;
                target = t < 600.0 ? 600u : (uint32_t)t;
            }
        }
        uint64_t nb = (ub_inst + target - 1) / target;
        if (nb < 1) nb = 1;
        if (nb > (1ull << 23)) nb = 1ull << 23;
        NB = (uint32_t)nb;
    }
    { const uint64_t nb_floor = (ub_inst >> 20) + 1; if (NB < nb_floor) NB = (uint32_t)nb_floor; }
    j->NB = NB;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; j->status = (uint32_t*)q;
    SNK_HIP_TRY(hipMemsetAsync(j->status, 0, 64, st));
    if ((rc = snk_ctx_alloc(ctx, total_reads_ub * 2 + 64, &q, err, errcap))) return rc; j->good_len = (uint16_t*)q;
    if ((rc = snk_partition_open(ctx, st, K, NB, ub_inst, total_reads_ub, false, j->status, &j->J, err, errcap))) return rc;
    j->open = true;
    return SNK_OK;
}

extern "C" int snk_dev_stream_append(snk_ctx* ctx, const snk_dev_reads* slab, void* stream, char* err, size_t errcap) {
    if (!ctx || !slab) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: NULL argument");
    stream_job* j = static_cast<stream_job*>(ctx->stream_job);
    if (!j || !j->open) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: no open job (snk_dev_stream_begin)");
    snk_set_mlen(ctx, &j->p);
    if (slab->n_reads == 0) return SNK_OK;
    if (!slab->rows || slab->read_len != j->read_len || slab->row_words * 16 < slab->read_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: bad rows / read_len (the job's is %u)", j->read_len);
    if (!slab->good_len && !slab->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: need quals or good_len");
    if ((slab->bc != nullptr) != j->has_bc) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: every slab carries barcodes, or none does");
    if (j->J.n_reads + slab->n_reads > j->total_ub)
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_append: more reads than the job's upper bound (%llu + %llu > %llu)", (unsigned long long)j->J.n_reads,
                        (unsigned long long)slab->n_reads, (unsigned long long)j->total_ub);
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    snk_dev_reads r = *slab;
    if (r.read_index_base == 0) r.read_index_base = j->J.n_reads;           // reads are numbered in arrival order unless the caller numbers them
    uint16_t* gl = j->good_len + j->J.n_reads;
    hipEvent_t e0, e1;
    SNK_HIP_TRY(hipEventCreate(&e0));
    SNK_HIP_TRY(hipEventCreate(&e1));
    j->ev.emplace_back(e0, e1);
    SNK_HIP_TRY(hipEventRecord(e0, st));
    int rc;
    if (slab->good_len) {
        SNK_HIP_TRY(hipMemcpyAsync(gl, slab->good_len, slab->n_reads * 2, hipMemcpyDeviceToDevice, st));
        rc = snk_partition_add(ctx, st, &j->J, &r, gl, nullptr, err, errcap);
    } else if (snk_fused_trim_ok(&r)) {
        snk_fused_trim ft;
        ft.quals = r.quals; ft.qstride = r.qstride; ft.lens = r.lens; ft.min_qual = j->p.min_qual; ft.good_out = gl;
        rc = snk_partition_add(ctx, st, &j->J, &r, gl, &ft, err, errcap);
    } else {
        rc = snk_dev_trim(ctx, r.quals, r.qstride, r.lens, r.read_len, r.n_reads, j->K, j->p.min_qual, gl, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        rc = snk_partition_add(ctx, st, &j->J, &r, gl, nullptr, err, errcap);
    }
    if (rc) return rc;
    SNK_HIP_TRY(hipEventRecord(e1, st));
    return SNK_OK;
}

extern "C" int snk_dev_stream_finish(snk_ctx* ctx, snk_dev_result* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_finish: NULL argument");
    stream_job* j = static_cast<stream_job*>(ctx->stream_job);
    if (!j || !j->open) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_stream_finish: no open job (snk_dev_stream_begin)");
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    j->open = false;
    memset(out, 0, sizeof *out);
    const snk_params* p = &j->p;
    snk_set_mlen(ctx, p);
    const uint32_t K = j->K;
    phase_timer tm(st);
    tm.mark(); tm.mark(); tm.mark();   // 0, 1, 2 (no trim / plan phase of their own)
    snk_partition part;
    unsigned long long h_plan[2] = {0, 0};
    int rc = snk_partition_close(ctx, st, &j->J, &part, h_plan, err, errcap);
    if (rc) return rc;
    tm.mark();  // 3
    float part_ms = 0.f;
    for (auto& e : j->ev) { float t = 0.f; if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) part_ms += t; }
    for (auto& e : j->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    j->ev.clear();
    part.kernel_ms = part_ms;
    const unsigned long long h_ninst = h_plan[0];
    out->n_reads = j->J.n_reads;
    out->good_len = j->good_len;
    out->n_instances = h_ninst;
    out->n_buckets = j->NB;
    out->n_supermers = part.n_supermers;
    out->n_overflow = part.n_overflow;
    const bool local_graph = !(p->flags & SNK_F_GLOBAL_GRAPH) && !snk_opt_u32("global_graph", 0);
    snk_hot hot;
    if ((rc = snk_stage_hot(ctx, st, K, false, &part, &hot, err, errcap))) return rc;
    out->n_hot_buckets = hot.n_hot;
    snk_table tab;
    rc = snk_stage_count_table(ctx, st, K, part.records, part.seg, part.seg + j->NB, 2 * j->NB, part.nseg, j->NB, p->min_freq, j->has_bc ? p->min_bc : 0u, 0u, h_ninst, j->status,
                               !local_graph, &tab, err, errcap, nullptr, nullptr, nullptr, local_graph, &hot);
    if (rc) return rc;
    // (the ratio is keyed by the job's read bound: the next job of the same size starts from it)
    rc = graph_tail(ctx, st, p, K, false, local_graph, j->total_ub, h_ninst, tab, part, out, tm, err, errcap);
    if (rc) return rc;
    out->phase_ms[2] = part_ms;
    return SNK_OK;
}

extern "C" int snk_dev_download(snk_ctx* ctx, const void* d_src, void* h_dst, size_t bytes, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_download: NULL ctx");
    if (bytes == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    SNK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    return SNK_OK;
}
