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

// segment tables of the single-pass partition: layout seg[0..NB) = begin of segment 0, [NB..2NB) = end of segment 0,
// [2NB..3NB) = begin of segment 1 (overflow), [3NB..4NB) = end of segment 1
__global__ void __launch_bounds__(256) seg0_kernel(const uint32_t* __restrict__ cursor, uint32_t NB, uint32_t cap,
                                                   uint64_t* __restrict__ seg, unsigned long long* __restrict__ total) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0;
    if (b < NB) {
        const uint32_t c = cursor[b];
        v = c;
        seg[b] = (uint64_t)b * cap;
        seg[NB + b] = (uint64_t)b * cap + (c < cap ? c : cap);
        seg[2ull * NB + b] = 0;
        seg[3ull * NB + b] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}
__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
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

#define env_u32 snk_env_u32

int snk_msp_segments(snk_ctx* ctx, hipStream_t st, uint32_t NB, uint32_t cap, const uint32_t* cursor, uint4* records, uint64_t ovf_base,
                     uint64_t ovf_cap, const uint32_t* ovf_bucket, uint32_t n_ovf, uint64_t* seg, unsigned long long* d_total, char* err,
                     size_t errcap) {
    SNK_HIP_TRY(hipMemsetAsync(d_total, 0, 8, st));
    hipLaunchKernelGGL(seg0_kernel, dim3((NB + 255) / 256), dim3(256), 0, st, cursor, NB, cap, seg, d_total);
    if (n_ovf) {
        uint32_t *idx_in, *idx_out, *key_out;
        void* q;
        int rc;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; idx_in = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; idx_out = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, (size_t)n_ovf * 4 + 16, &q, err, errcap))) return rc; key_out = (uint32_t*)q;
        hipLaunchKernelGGL(iota_kernel, dim3((n_ovf + 255) / 256), dim3(256), 0, st, idx_in, n_ovf);
        size_t tb = 0;
        SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb, ovf_bucket, key_out, idx_in, idx_out, (size_t)n_ovf, 0u, 32u, st));
        if ((rc = snk_ctx_alloc(ctx, tb, &q, err, errcap))) return rc;
        SNK_HIP_TRY(rocprim::radix_sort_pairs(q, tb, ovf_bucket, key_out, idx_in, idx_out, (size_t)n_ovf, 0u, 32u, st));
        hipLaunchKernelGGL(ovf_gather_kernel, dim3((n_ovf + 255) / 256), dim3(256), 0, st, records, ovf_base, ovf_base + ovf_cap, idx_out,
                           key_out, n_ovf, NB, records, seg);
    }
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

}  // namespace

extern "C" int snk_dev_count_graph(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, snk_dev_result* out,
                                   void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 2) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule supports 0, 1, 2", p->min_bc);
    if (in->n_reads && (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: bad rows/read_len (read_len <= 256)");
    if (in->n_reads && !in->good_len && !in->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_count_graph: need quals or good_len");
    const bool grouped = (p->flags & SNK_F_GROUPED) != 0;
    if (grouped) {
        if (p->K != 48) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED: the group id rides in the 32 key bits that are free at K=48 only");
        if (!in->group) return snk_fail(SNK_E_ARG, err, errcap, "SNK_F_GROUPED: snk_dev_reads.group is NULL");
        if (p->min_bc > 0 && in->bc) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED: per-group graphs use the frequency rule only (min_bc = 0)");
        if ((p->flags & SNK_F_GLOBAL_GRAPH) || snk_env_u32("SNK_GLOBAL_GRAPH", 0)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "SNK_F_GROUPED needs the bucket-local graph stage");
    }
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

    // ---- K3/K4 minimiser partition in ONE pass: exact k-mer instance count -> expected supermers -> fixed bucket capacity
    unsigned long long* counters = nullptr;   // [0] instances, [1] live reads
    uint32_t* status = nullptr;
    {
        void* q;
        int rc;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; counters = (unsigned long long*)q;
        if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; status = (uint32_t*)q;
    }
    SNK_HIP_TRY(hipMemsetAsync(status, 0, 64, st));
    int rc = snk_launch_msp_plan(st, good_len, n_reads, K, counters, err, errcap);
    if (rc) return rc;
    unsigned long long h_plan[2] = {0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h_plan, counters, 16, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long h_ninst = h_plan[0];
    out->n_instances = h_ninst;
    uint32_t NB = p->n_buckets;
    if (NB == 0) {
        // instances per bucket: sized so that the DISTINCT k-mers of a bucket fit the LDS table (1216 claims).  At 56x
        // coverage 4000 instances hold ~500 distinct k-mers; per-barcode groups see every locus once or twice, so
        // nearly every instance is distinct there
        uint32_t target = env_u32("SNK_TARGET_INST", grouped ? 900u : (K == 48 ? 4000u : 3500u));
        uint64_t nb = (h_ninst + target - 1) / target;
        const uint64_t nb_max = grouped ? (1ull << 24) : (1ull << 22);
        if (nb < 1) nb = 1;
        if (nb > nb_max) nb = nb_max;
        NB = (uint32_t)nb;
    }
    out->n_buckets = NB;
    const uint32_t Wm = K - SNK_M + 1;
    // a random-order minimiser starts a new supermer every (W+1)/2 k-mers, and every contributing read starts one
    const double est_super = (double)h_ninst * 2.0 / (Wm + 1) + (double)h_plan[1];
    const double mean = est_super / NB;
    // Bucket occupancy is NOT Poisson in the supermers: a minimiser site of the genome contributes one supermer per
    // read that covers it (~38 at 56x), so a 4000-instance bucket holds only ~7 sites and its supermer count has a
    // relative sigma of ~37 %.  With 1.25 x mean + 4 sqrt(mean) 1.7 % of the supermers overflowed and their
    // reservations on the single overflow cursor cost 30 ms (tools/msp_probe2.py).  2.5 x mean is > 5 sigma of the site
    // count at 56x and generous below; the slots that stay empty are never touched.
    uint64_t cap64 = (uint64_t)(mean * 2.5 + 64.0);
    cap64 = cap64 * env_u32("SNK_MSP_CAP_PCT", 100) / 100;
    if (cap64 < 2) cap64 = 2;
    cap64 = (cap64 + 1) & ~1ull;
    if (cap64 * NB >= (1ull << 40) || cap64 >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "bucket capacity out of range");
    const uint32_t cap = (uint32_t)cap64;
    uint64_t ovf_cap = (uint64_t)(est_super / 16) + 65536;
    tm.mark();  // 2
    uint32_t* cursor = nullptr;
    uint64_t* seg = nullptr;           // [2 segments][beg NB | end NB]
    {
        void* q;
        if ((rc = snk_ctx_alloc(ctx, (NB + 1) * 4ull, &q, err, errcap))) return rc; cursor = (uint32_t*)q;
        if ((rc = snk_ctx_alloc(ctx, 4ull * NB * 8 + 64, &q, err, errcap))) return rc; seg = (uint64_t*)q;
    }
    phase_timer kt(st);   // single-launch timings (events on the launch stream right around the kernel)
    void* records = nullptr;
    uint32_t* ovf_bucket = nullptr;
    uint32_t h_novf = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (ovf_cap >= (1ull << 31)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "supermer overflow list too large");
        void* q;
        if ((rc = snk_ctx_alloc(ctx, ((size_t)NB * cap + 2 * ovf_cap) * 32 + 64, &records, err, errcap))) return rc;
        if ((rc = snk_ctx_alloc(ctx, ovf_cap * 4 + 64, &q, err, errcap))) return rc; ovf_bucket = (uint32_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(cursor, 0, (NB + 1) * 4ull, st));
        SNK_HIP_TRY(hipMemsetAsync(status + 8, 0, 4, st));
        snk_msp_args ma;
        memset(&ma, 0, sizeof ma);
        ma.rows = (const uint32_t*)in->rows; ma.row_words = in->row_words; ma.good_len = good_len; ma.bc = (const int32_t*)in->bc;
        ma.ign_bc_below = in->ign_bc_below; ma.read_index_base = in->read_index_base; ma.n_reads = n_reads; ma.NB = NB;
        ma.group = grouped ? (const uint32_t*)in->group : nullptr;
        ma.hist_or_cursor = cursor; ma.records = (uint4*)records; ma.cap = cap; ma.ovf_cap = (uint32_t)ovf_cap;
        ma.ovf_base = (uint64_t)NB * cap; ma.ovf_bucket = ovf_bucket; ma.ovf_cursor = status + 8;
        ma.dbg = env_u32("SNK_MSP_DBG", 0);
        kt.n = 0;
        kt.mark();  // 0
        if ((rc = snk_launch_msp_args(K, SNK_MSP_MODE_SINGLE, st, ma, err, errcap))) return rc;
        kt.mark();  // 1
        SNK_HIP_TRY(hipMemcpyAsync(&h_novf, status + 8, 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipStreamSynchronize(st));
        if (h_novf <= ovf_cap) break;
        if (attempt == 2) return snk_fail(SNK_E_INTERNAL, err, errcap, "supermer overflow list too small (%u > %llu)", h_novf, (unsigned long long)ovf_cap);
        ovf_cap = (uint64_t)h_novf + 65536;
    }
    uint32_t nseg = 1;
    {
        // segment 0: the fixed-capacity slots; segment 1: the overflow records grouped by bucket
        unsigned long long* d_total = counters + 4;
        int rc2 = snk_msp_segments(ctx, st, NB, cap, cursor, (uint4*)records, (uint64_t)NB * cap, ovf_cap, ovf_bucket, h_novf, seg, d_total,
                                   err, errcap);
        if (rc2) return rc2;
        unsigned long long h_total = 0;
        SNK_HIP_TRY(hipMemcpyAsync(&h_total, d_total, 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipStreamSynchronize(st));
        out->n_supermers = h_total;
        if (h_novf) nseg = 2;
    }
    out->n_overflow = h_novf;
    tm.mark();  // 3

    // ---- K5-K8 count + filter + gather (+ sort for the global graph stage)
    const bool local_graph = !(p->flags & SNK_F_GLOBAL_GRAPH) && !env_u32("SNK_GLOBAL_GRAPH", 0);
    snk_table tab;
    rc = snk_stage_count_table(ctx, st, K, records, seg, seg + NB, 2 * NB, nseg, NB, p->min_freq, (in->bc && !grouped) ? p->min_bc : 0u, grouped ? 1u : 0u, h_ninst, status,
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
    SNK_HIP_TRY(hipStreamSynchronize(st));
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
    out->kernel_ms[1] = kt.ms(0, 1);
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
