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

#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_graph.h"
#include "snk_kernels.h"

namespace {

__global__ void widen_offsets_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

struct phase_timer {
    hipStream_t st;
    hipEvent_t ev[16];
    int n = 0;
    bool ok = true;
    explicit phase_timer(hipStream_t s) : st(s) {
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) ok = false;
    }
    ~phase_timer() { for (auto& e : ev) (void)hipEventDestroy(e); }
    void mark() { if (ok && n < 16) (void)hipEventRecord(ev[n++], st); }
    float ms(int a, int b) {
        float t = 0;
        if (!ok || a >= n || b >= n) return 0;
        (void)hipEventSynchronize(ev[b]);
        (void)hipEventElapsedTime(&t, ev[a], ev[b]);
        return t;
    }
};

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

}  // namespace

extern "C" int snk_dev_count_graph(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, snk_dev_result* out,
                                   void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 2) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule supports 0, 1, 2", p->min_bc);
    if (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256)
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: bad rows/read_len (read_len <= 256)");
    if (!in->good_len && !in->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: need quals or good_len");
    SNK_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    snk_ctx_release_scratch(ctx);
    memset(out, 0, sizeof *out);
    const uint32_t K = p->K;
    const uint64_t n_reads = in->n_reads;
    out->n_reads = n_reads;
    phase_timer tm(st);
    tm.mark();  // 0

    // ---- K1 trim
    const uint16_t* good_len = (const uint16_t*)in->good_len;
    if (!good_len) {
        void* gl = nullptr;
        int rc = snk_ctx_alloc(ctx, n_reads * 2 + 2, &gl, err, errcap);
        if (rc) return rc;
        rc = snk_dev_trim(ctx, in->quals, in->qstride, in->lens, in->read_len, n_reads, K, p->min_qual, gl, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        good_len = (const uint16_t*)gl;
    }
    out->good_len = good_len;
    tm.mark();  // 1

    // ---- K3 histogram pass
    uint32_t NB = p->n_buckets;
    if (NB == 0) {
        uint64_t inst_ub = n_reads * (uint64_t)(in->read_len >= K ? in->read_len - K + 1 : 0);
        uint32_t target = env_u32("SNK_TARGET_INST", K == 48 ? 4500u : 4000u);
        uint64_t nb = (inst_ub + target - 1) / target;
        if (nb < 1) nb = 1;
        if (nb > (1u << 22)) nb = 1u << 22;
        NB = (uint32_t)nb;
    }
    out->n_buckets = NB;
    uint32_t *hist = nullptr, *cursor = nullptr;
    uint64_t* seg_off = nullptr;
    unsigned long long* counters = nullptr;   // [0] instances, [1] output cursor
    uint32_t* status = nullptr;
    {
        void* q;
        int rc;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc; hist = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc; cursor = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 8ull, &q, err, errcap))) return rc; seg_off = (uint64_t*)q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; counters = (unsigned long long*)q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; status = (uint32_t*)q;
    }
    SNK_HIP_TRY(hipMemsetAsync(hist, 0, (NB + 1) * 4ull, st));
    SNK_HIP_TRY(hipMemsetAsync(counters, 0, 64, st));
    SNK_HIP_TRY(hipMemsetAsync(status, 0, 64, st));
    phase_timer kt(st);   // single-launch timings (events on the launch stream right around the kernel)
    kt.mark();  // 0
    int rc = snk_launch_msp(K, false, st, (const uint32_t*)in->rows, in->row_words, good_len, (const int32_t*)in->bc,
                            in->ign_bc_below, in->read_index_base, n_reads, NB, hist, nullptr, counters, err, errcap);
    if (rc) return rc;
    kt.mark();  // 1
    {
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, hist, cursor, 0u, (size_t)(NB + 1), rocprim::plus<uint32_t>(), st));
        void* tmp;
        if ((rc = snk_ctx_alloc(ctx, tb, &tmp, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, hist, cursor, 0u, (size_t)(NB + 1), rocprim::plus<uint32_t>(), st));
    }
    hipLaunchKernelGGL(widen_offsets_kernel, dim3((NB + 1 + 255) / 256), dim3(256), 0, st, cursor, seg_off, NB + 1);
    uint32_t h_nsuper = 0;
    unsigned long long h_ninst = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&h_nsuper, cursor + NB, 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipMemcpyAsync(&h_ninst, counters, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    out->n_supermers = h_nsuper;
    out->n_instances = h_ninst;
    tm.mark();  // 2

    // ---- K4 scatter pass
    void* records = nullptr;
    if ((rc = snk_ctx_alloc(ctx, (size_t)h_nsuper * 32 + 32, &records, err, errcap))) return rc;
    kt.mark();  // 2
    rc = snk_launch_msp(K, true, st, (const uint32_t*)in->rows, in->row_words, good_len, (const int32_t*)in->bc,
                        in->ign_bc_below, in->read_index_base, n_reads, NB, cursor, records, nullptr, err, errcap);
    if (rc) return rc;
    kt.mark();  // 3
    tm.mark();  // 3

    // ---- K5-K8 count + filter into a region-partitioned table, then gather the regions densely.
    // Every retained k-mer has >= min_freq instances; deep coverage retains far fewer (56x: ~1/38 of them).
    uint32_t n_regions = NB < 4096 ? NB : 4096;
    uint64_t est = h_ninst / (p->min_freq > 1 ? 8 : 1) + 4096;
    if (ctx->last_n_kmers && ctx->last_n_instances == h_ninst) est = ctx->last_n_kmers + ctx->last_n_kmers / 2 + 4096;
    uint64_t region_cap = est / n_regions + 64;
    snk_u128 *keys_r = nullptr, *keys_a = nullptr, *keys_b = nullptr;
    uint64_t *vals_r = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    unsigned long long *rcur = nullptr, *roff = nullptr;
    uint64_t n_kmers = 0;
    uint32_t h_status[4] = {0, 0, 0, 0};
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_regions + 1) * 8ull, &q, err, errcap))) return rc; rcur = (unsigned long long*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_regions + 1) * 8ull, &q, err, errcap))) return rc; roff = (unsigned long long*)q;
    }
    std::vector<unsigned long long> h_rcur(n_regions);
    for (int attempt = 0; attempt < 3; ++attempt) {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, region_cap * n_regions * 16, &q, err, errcap))) return rc; keys_r = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, region_cap * n_regions * 8, &q, err, errcap))) return rc; vals_r = (uint64_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(rcur, 0, (n_regions + 1) * 8ull, st));
        SNK_HIP_TRY(hipMemsetAsync(status, 0, 16, st));
        snk_count_args ca;
        ca.records = (const uint4*)records;
        ca.seg_off = seg_off;
        ca.nseg = 1;
        ca.NB = NB;
        ca.min_freq = p->min_freq;
        ca.bc_mode = in->bc ? p->min_bc : 0;    // no barcode vector -> bc_test is always true (:176-178)
        ca.out_keys = keys_r;
        ca.out_vals = vals_r;
        ca.region_cap = region_cap;
        ca.n_regions = n_regions;
        ca.region_cursor = rcur;
        ca.status = status;
        ca.dbg = env_u32("SNK_COUNT_DBG", 0);
        if (kt.n > 4) kt.n = 4;
        kt.mark();  // 4
        if ((rc = snk_launch_count(K, st, ca, err, errcap))) return rc;
        kt.mark();  // 5
        SNK_HIP_TRY(hipMemcpyAsync(h_rcur.data(), rcur, n_regions * 8ull, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(h_status, status, 16, hipMemcpyDeviceToHost, st));
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
        if (!h_status[0] && mx <= region_cap) break;
        if (attempt == 2) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: region overflow (%llu > %llu)", mx, (unsigned long long)region_cap);
        region_cap = mx + 64;     // exact requirement is known now (cursors keep counting past the cap)
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
    out->n_kmers = n_kmers;
    ctx->last_n_kmers = n_kmers;
    ctx->last_n_instances = h_ninst;
    tm.mark();  // 4

    // ---- sort by key
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 16, &q, err, errcap))) return rc; keys_b = (snk_u128*)q;
        if ((rc = snk_ctx_alloc(ctx, (n_kmers + 1) * 8, &q, err, errcap))) return rc; vals_b = (uint64_t*)q;
        if ((rc = snk_graph_sort(ctx, st, K, n_kmers, keys_a, vals_a, keys_b, vals_b, err, errcap))) return rc;
    }
    out->keys = keys_b;
    tm.mark();  // 5

    // ---- prune + unitigs
    snk_graph_out go;
    rc = snk_graph_build(ctx, st, K, keys_b, vals_b, n_kmers, p->min_freq > 1 ? 1u : 0u, !(p->flags & SNK_F_NO_GRAPH), &go,
                         err, errcap);
    if (rc) return rc;
    tm.mark();  // 6
    SNK_HIP_TRY(hipStreamSynchronize(st));
    out->counts = go.counts;
    out->ctx = go.ctx;
    out->spectrum = go.spectrum;
    out->spectrum_bins = go.spectrum_bins;
    out->n_unitigs = go.n_unitigs;
    out->unitig_total_bases = go.total_bases;
    out->unitig_off = go.unitig_off;
    out->unitig_bases = go.unitig_bases;
    out->n_circles = go.n_circles;
    out->rank_rounds = go.rank_rounds;
    out->phase_ms[0] = tm.ms(0, 1);
    out->phase_ms[1] = tm.ms(1, 2);
    out->phase_ms[2] = tm.ms(2, 3);
    out->phase_ms[3] = tm.ms(3, 4);
    out->phase_ms[4] = tm.ms(4, 5);
    out->phase_ms[5] = tm.ms(5, 6);
    out->phase_ms[7] = tm.ms(0, 6);
    out->kernel_ms[0] = kt.ms(0, 1);
    out->kernel_ms[1] = kt.ms(2, 3);
    out->kernel_ms[2] = kt.ms(4, 5);
    out->scratch_bytes = ctx->total_alloc;
    return SNK_OK;
}

extern "C" int snk_dev_download(snk_ctx* ctx, const void* d_src, void* h_dst, size_t bytes, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_download: NULL ctx");
    if (bytes == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    SNK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    return SNK_OK;
}
