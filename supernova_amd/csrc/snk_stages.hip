// snk_stages.hip -- reusable pipeline stages shared by the single-GPU and the sharded paths.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <stdlib.h>

#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_graph.h"
#include "snk_kernels.h"
#include "snk_stages.h"

uint32_t snk_env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}
#define env_u32 snk_env_u32

// K5-K8 + gather + sort: supermer records of NB buckets (nseg segments) -> dense retained table sorted by key.
// status: device u32[16] scratch words.
int snk_stage_count_table(snk_ctx* ctx, hipStream_t st, uint32_t K, const void* records, const uint64_t* seg_beg,
                          const uint64_t* seg_end, uint32_t seg_stride, uint32_t nseg, uint32_t NB, uint32_t min_freq, uint32_t bc_mode, uint32_t grouped, uint64_t n_inst_hint,
                          uint32_t* status, bool want_sort, snk_table* out, char* err, size_t errcap) {
    int rc;
    snk_phase_timer tm(st), kt(st);
    tm.mark();
    // ---- K5-K8 count + filter into a region-partitioned table, then gather the regions densely.
    // Every retained k-mer has >= min_freq instances; deep coverage retains far fewer (56x: ~1/38 of them).
    uint32_t n_regions = NB < 4096 ? NB : 4096;
    uint64_t est = n_inst_hint / (min_freq > 1 ? 8 : 1) + 4096;
    if (ctx->last_n_kmers && ctx->last_n_instances == n_inst_hint) est = ctx->last_n_kmers + ctx->last_n_kmers / 2 + 4096;
    uint64_t region_cap = est / n_regions + 64;
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
        extra_cap = 1u << 16;
    }
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
        ca.NB = NB;
        ca.min_freq = min_freq;
        ca.bc_mode = bc_mode;
        ca.grouped = grouped;
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
        ca.dbg = env_u32("SNK_COUNT_DBG", 0);
        kt.n = 0;
        kt.mark();
        if ((rc = snk_launch_count(K, st, ca, err, errcap))) return rc;
        kt.mark();
        SNK_HIP_TRY(hipMemcpyAsync(h_rcur.data(), rcur, n_regions * 8ull, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(h_status, status, 32, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipStreamSynchronize(st));
        if (ca.dbg >= 2) {
            unsigned long long d[3];
            (void)hipMemcpy(d, status + 4, 24, hipMemcpyDeviceToHost);
            fprintf(stderr, "[snk dbg] lane probe iterations %llu, wave-level iterations %llu, max lane iterations in one probe %llu\n", d[0], d[1], d[2]);
        }
        if (h_status[1]) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: bucket split depth exceeded");
        unsigned long long mx = 0;
        n_kmers = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { n_kmers += h_rcur[r]; if (h_rcur[r] > mx) mx = h_rcur[r]; }
        const bool extra_ovf = !want_sort && h_status[4] > extra_cap;
        if (!h_status[0] && mx <= region_cap && !extra_ovf) break;
        if (attempt == 3) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: region overflow (%llu > %llu)", mx, (unsigned long long)region_cap);
        if (mx > region_cap) region_cap = mx + 64;     // exact requirement is known now (cursors keep counting past the cap)
        if (extra_ovf) extra_cap = h_status[4] + 64;
    }
    {
        // exclusive offsets of the regions (host: n_regions <= 4096) and the dense gather
        std::vector<unsigned long long> h_off(n_regions + 1);
        unsigned long long acc = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { h_off[r] = acc; acc += h_rcur[r]; }
        h_off[n_regions] = acc;
        SNK_HIP_TRY(hipMemcpyAsync(roff, h_off.data(), (n_regions + 1) * 8ull, hipMemcpyHostToDevice, st));
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 16, &q, err, errcap))) return rc; keys_a = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 8, &q, err, errcap))) return rc; vals_a = (uint64_t*)q;
        if ((rc = snk_launch_compact_regions(st, keys_r, vals_r, region_cap, n_regions, rcur, roff, keys_a, vals_a, err, errcap))) return rc;
        SNK_HIP_TRY(hipStreamSynchronize(st));   // h_off is a stack vector: the upload must finish before it goes away
    }
    out->buckets_split = h_status[2];
    out->max_slots_used = h_status[3];
    out->n = n_kmers;
    ctx->last_n_kmers = n_kmers;
    ctx->last_n_instances = n_inst_hint;
    tm.mark();
    out->sorted = want_sort;
    out->NB = NB;
    out->n_regions = n_regions;
    out->n_extra = want_sort ? 0u : h_status[4];
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
