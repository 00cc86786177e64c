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
#include "snk_stages.h"

namespace {

__global__ void widen_offsets_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

typedef snk_phase_timer phase_timer;

#define env_u32 snk_env_u32

}  // namespace

extern "C" int snk_dev_count_graph(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, snk_dev_result* out,
                                   void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 2) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule supports 0, 1, 2", p->min_bc);
    if (in->n_reads && (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: bad rows/read_len (read_len <= 256)");
    if (in->n_reads && !in->good_len && !in->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: need quals or good_len");
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
    if (!good_len && n_reads) {
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
        uint32_t target = env_u32("SNK_TARGET_INST", K == 48 ? 4000u : 3500u);
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
    uint16_t* slist = nullptr;
    uint8_t* scount = nullptr;
    {
        void* q;
        int rc2;
        if ((rc2 = snk_ctx_alloc(ctx, (size_t)SNK_MSP_LCAP * n_reads * 2 + 64, &q, err, errcap))) return rc2; slist = (uint16_t*)q;
        if ((rc2 = snk_ctx_alloc(ctx, n_reads + 64, &q, err, errcap))) return rc2; scount = (uint8_t*)q;
    }
    phase_timer kt(st);   // single-launch timings (events on the launch stream right around the kernel)
    kt.mark();  // 0
    int rc = snk_launch_msp(K, false, st, (const uint32_t*)in->rows, in->row_words, good_len, (const int32_t*)in->bc,
                            in->ign_bc_below, in->read_index_base, n_reads, NB, hist, nullptr, counters, slist, scount, err, errcap);
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
                        in->ign_bc_below, in->read_index_base, n_reads, NB, cursor, records, nullptr, slist, scount, err, errcap);
    if (rc) return rc;
    kt.mark();  // 3
    tm.mark();  // 3

    // ---- K5-K8 count + filter + gather (+ sort for the global graph stage)
    const bool local_graph = !(p->flags & SNK_F_GLOBAL_GRAPH) && !env_u32("SNK_GLOBAL_GRAPH", 0);
    snk_table tab;
    rc = snk_stage_count_table(ctx, st, K, records, seg_off, 1, NB, p->min_freq, in->bc ? p->min_bc : 0u, h_ninst, status,
                               !local_graph, &tab, err, errcap);
    if (rc) return rc;
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
                             !(p->flags & SNK_F_UNSORTED_TABLE), &go, &keys_final, out->graph_ms, err, errcap);
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
    out->phase_ms[3] = tab.count_ms;
    out->phase_ms[4] = tab.sort_ms;
    out->phase_ms[5] = tm.ms(5, 6);
    out->phase_ms[7] = tm.ms(0, 6);
    out->kernel_ms[0] = kt.ms(0, 1);
    out->kernel_ms[1] = kt.ms(2, 3);
    out->kernel_ms[2] = tab.count_kernel_ms;
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
