// snk_dist.hip -- C ABI of the minimiser-sharded (multi-GPU) path: stage entry points between which the host
// runs the exchanges (torch.distributed over RCCL/xGMI).  SURVEY.md 8(e); the reference's counterpart is the
// shardio exchange + per-shard processing + global join of tada
// (lib/tada/external/rust-shardio/src/shard.rs:184-211,488-493; cmd_shard_asm.rs:37-94; cmd_main_asm.rs:25-89)
// and the thread swizzle of MapReduceEngine.h:362-385.
//
//   rank r:  snk_shard_hist  -> [all-to-all of bucket counts] -> snk_shard_scatter -> [all-to-all of records]
//            -> snk_shard_count -> snk_shard_prune_plan/fill -> [all-to-all of queries] -> snk_shard_prune_answer
//            -> [all-to-all of answers] -> snk_shard_prune_apply -> snk_shard_fragments -> [gather to rank 0]
//   rank 0:  snk_shard_join
#include <string.h>

#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_graph.h"
#include "snk_kernels.h"
#include "snk_stages.h"
#include "snk_shard.h"

static snk_shard_state* state_of(snk_ctx* ctx) { return snk_shard_state_of(ctx); }
void snk_shard_state_free(void* p) { delete static_cast<snk_shard_state*>(p); }
void snk_shard_state_invalidate_job(void* p) { if (p) static_cast<snk_shard_state*>(p)->job_open = false; }

int snk_shard_begin(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, uint32_t rank, uint32_t world, uint32_t NB_total,
                    uint64_t* n_instances, hipStream_t st, char* err, size_t errcap) {
    if (!ctx || !in || !p) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 8) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule tells up to eight distinct barcodes apart (min_bc <= 8)", p->min_bc);
    if (world == 0 || rank >= world || NB_total == 0 || NB_total % world) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: NB_total must be a positive multiple of world");
    if (world > 0x7FFF) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "world > 32767");
    if (in->n_reads && (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256 || (!in->quals && !in->good_len)))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: bad reads");
    SNK_HIP_TRY(snk_enter(ctx));
    snk_ctx_release_scratch(ctx);
    snk_shard_state* S = state_of(ctx);
    S->reads = *in;
    S->params = *p;
    S->rank = rank; S->world = world; S->NB_total = NB_total; S->NBl = NB_total / world;
    S->circ_all = nullptr; S->join_circles = 0;
    const uint16_t* good_len = (const uint16_t*)in->good_len;
    int rc;
    const bool fused = in->n_reads && snk_fused_trim_ok(in);      // the trim inside the partition kernel (snk_msp.hip)
    snk_fused_trim ft;
    if (!good_len) {
        void* gl;
        if ((rc = snk_ctx_alloc(ctx, in->n_reads * 2 + 2, &gl, err, errcap))) return rc;
        if (fused) { ft.quals = in->quals; ft.qstride = in->qstride; ft.lens = in->lens; ft.min_qual = p->min_qual; ft.good_out = (uint16_t*)gl; }
        else if (in->n_reads) {
            rc = snk_dev_trim(ctx, in->quals, in->qstride, in->lens, in->read_len, in->n_reads, p->K, p->min_qual, gl, st);
            if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        }
        good_len = (const uint16_t*)gl;
    }
    S->good_len = good_len;
    void* q;
    if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; S->status = (uint32_t*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->status, 0, 64, st));
    // one partition pass into fixed-capacity bucket slots (as on one GPU); the histogram is its cursor array
    // the slots are sized from what the untrimmed reads could hold at most (a few per cent above what the trim leaves: the
    // capacity is 2.5 x the mean anyway); the exact instance count comes back with the partition's own read-back
    unsigned long long h_plan[2] = {0, 0};
    unsigned long long* d_plan = nullptr;
    if (!fused && (rc = snk_stage_partition_plan(ctx, st, p->K, good_len, in->n_reads, h_plan, err, errcap, &d_plan))) return rc;
    const unsigned long long kpr = in->read_len >= p->K ? in->read_len - p->K + 1 : 0;
    if ((rc = snk_stage_partition(ctx, st, p->K, in, good_len, NB_total, in->n_reads * kpr, in->n_reads, false, S->status, &S->part, err, errcap, d_plan, h_plan,
                                  fused ? &ft : nullptr))) return rc;
    if (n_instances) *n_instances = h_plan[0];
    return SNK_OK;
}

// ---- streamed step: the partition runs slab by slab while the next slab is decoded / uploaded (lib/tada/src/cmd_msp.rs:55-69 streams its
// FASTQ chunks into the partitioner the same way); everything behind the partition is the resident step's
int snk_shard_job_open(snk_ctx* ctx, const snk_params* p, uint32_t rank, uint32_t world, uint32_t NB_total, uint32_t read_len, uint64_t reads_ub,
                       uint64_t total_reads, int has_bc, hipStream_t st, char* err, size_t errcap) {
    if (!ctx || !p) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_begin: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 8) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule tells up to eight distinct barcodes apart (min_bc <= 8)", p->min_bc);
    if (world == 0 || rank >= world || NB_total == 0 || NB_total % world) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_begin: NB_total must be a positive multiple of world");
    if (read_len == 0 || read_len > 256 || reads_ub == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_begin: read_len 1..256 and an upper bound of this rank's reads are needed");
    SNK_HIP_TRY(snk_enter(ctx));
    snk_ctx_release_scratch(ctx);
    snk_shard_state* S = state_of(ctx);
    memset(&S->reads, 0, sizeof S->reads);
    S->params = *p;
    S->rank = rank; S->world = world; S->NB_total = NB_total; S->NBl = NB_total / world;
    S->circ_all = nullptr; S->join_circles = 0;
    S->job_open = false; S->job_read_len = read_len; S->job_has_bc = has_bc; S->job_reads_ub = reads_ub; S->job_total_reads = total_reads;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; S->status = (uint32_t*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->status, 0, 64, st));
    if ((rc = snk_ctx_alloc(ctx, reads_ub * 2 + 64, &q, err, errcap))) return rc; S->job_good_len = (uint16_t*)q;
    const unsigned long long kpr = read_len >= p->K ? read_len - p->K + 1 : 0;
    if ((rc = snk_partition_open(ctx, st, p->K, NB_total, reads_ub * kpr, reads_ub, false, S->status, &S->job, err, errcap))) return rc;
    S->job_open = true;
    return SNK_OK;
}
int snk_shard_job_add(snk_ctx* ctx, const snk_dev_reads* slab, hipStream_t st, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !slab) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: NULL argument / no open step");
    snk_shard_state* S = state_of(ctx);
    if (!S->job_open) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: no open step (snk_shard_stream_begin)");
    if (slab->n_reads == 0) return SNK_OK;
    if (!slab->rows || slab->read_len != S->job_read_len || slab->row_words * 16 < slab->read_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: bad rows / read_len (the step's is %u)", S->job_read_len);
    if (!slab->good_len && !slab->quals) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: need quals or good_len");
    if ((slab->bc != nullptr) != (S->job_has_bc != 0)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: every slab carries barcodes, or none does");
    if (S->job.n_reads + slab->n_reads > S->job_reads_ub)
        return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: more reads than the rank's upper bound (%llu + %llu > %llu)", (unsigned long long)S->job.n_reads,
                        (unsigned long long)slab->n_reads, (unsigned long long)S->job_reads_ub);
    SNK_HIP_TRY(snk_enter(ctx));
    snk_dev_reads r = *slab;
    uint16_t* gl = S->job_good_len + S->job.n_reads;
    int rc;
    if (slab->good_len) {
        SNK_HIP_TRY(hipMemcpyAsync(gl, slab->good_len, slab->n_reads * 2, hipMemcpyDeviceToDevice, st));
        rc = snk_partition_add(ctx, st, &S->job, &r, gl, nullptr, err, errcap);
    } else if (snk_fused_trim_ok(&r)) {
        snk_fused_trim ft;
        ft.quals = r.quals; ft.qstride = r.qstride; ft.lens = r.lens; ft.min_qual = S->params.min_qual; ft.good_out = gl;
        rc = snk_partition_add(ctx, st, &S->job, &r, gl, &ft, err, errcap);
    } else {
        rc = snk_dev_trim(ctx, r.quals, r.qstride, r.lens, r.read_len, r.n_reads, S->params.K, S->params.min_qual, gl, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        rc = snk_partition_add(ctx, st, &S->job, &r, gl, nullptr, err, errcap);
    }
    return rc;
}
int snk_shard_job_adopt(snk_ctx* ctx, uint64_t* n_instances, hipStream_t st, char* err, size_t errcap) {
    snk_shard_state* S = state_of(ctx);
    if (!S->job_open) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_finish: no open step");
    S->job_open = false;
    S->good_len = S->job_good_len;
    S->reads.n_reads = S->job.n_reads;
    S->reads.read_len = S->job_read_len;
    unsigned long long h_plan[2] = {0, 0};
    int rc = snk_partition_close(ctx, st, &S->job, &S->part, h_plan, err, errcap);
    if (rc) return rc;
    if (n_instances) *n_instances = h_plan[0];
    return SNK_OK;
}

extern "C" int snk_shard_hist(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, uint32_t rank, uint32_t world,
                              uint32_t NB_total, void* d_hist, uint64_t* n_instances, void* stream, char* err, size_t errcap) {
    if (!ctx || !d_hist) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: NULL argument");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    int rc = snk_shard_begin(ctx, in, p, rank, world, NB_total, n_instances, st, err, errcap);
    if (rc) return rc;
    // the "scatter" stage compacts the slots into the caller's exact, destination-contiguous send buffer
    SNK_HIP_TRY(hipMemcpyAsync(d_hist, state_of(ctx)->part.cursor, (size_t)NB_total * 4, hipMemcpyDeviceToDevice, st));
    SNK_HIP_TRY(snk_sync(st));
    return SNK_OK;
}

extern "C" int snk_shard_scatter(snk_ctx* ctx, const void* d_offsets, void* d_records, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_offsets || !d_records) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_scatter: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    return snk_stage_partition_compact(ctx, st, &S->part, (const uint32_t*)d_offsets, d_records, err, errcap);
}

extern "C" int snk_shard_count(snk_ctx* ctx, const void* d_records, const void* d_seg_off, uint64_t n_inst_hint, int has_bc,
                               uint64_t* n_kmers, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_seg_off) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_count: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    int rc = snk_stage_count_table(ctx, st, S->params.K, d_records, (const uint64_t*)d_seg_off, (const uint64_t*)d_seg_off + 1, S->NBl + 1, S->world, S->NBl, S->params.min_freq,
                                   has_bc ? S->params.min_bc : 0u, 0u, n_inst_hint, S->status, false, &S->tab, err, errcap);
    if (rc) return rc;
    if (n_kmers) *n_kmers = S->tab.n;
    return SNK_OK;
}

extern "C" int snk_shard_count_ranged(snk_ctx* ctx, const void* d_records, const void* d_seg_off, uint64_t n_inst_hint, int has_bc,
                                      uint32_t n_ranges, const uint32_t* bounds, int (*ready)(void*, uint32_t), void* user,
                                      uint64_t* n_kmers, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_seg_off || (n_ranges && !bounds)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_count_ranged: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    if (n_ranges && (bounds[0] != 0 || bounds[n_ranges] != S->NBl)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_count_ranged: the ranges must cover the local buckets");
    for (uint32_t r = 0; r < n_ranges; ++r) if (bounds[r] > bounds[r + 1]) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_count_ranged: descending range bounds");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    snk_count_ranges rg{n_ranges, bounds, ready, user};
    int rc = snk_stage_count_table(ctx, st, S->params.K, d_records, (const uint64_t*)d_seg_off, (const uint64_t*)d_seg_off + 1, S->NBl + 1, S->world, S->NBl, S->params.min_freq,
                                   has_bc ? S->params.min_bc : 0u, 0u, n_inst_hint, S->status, false, &S->tab, err, errcap, n_ranges ? &rg : nullptr);
    if (rc) return rc;
    if (n_kmers) *n_kmers = S->tab.n;
    return SNK_OK;
}

extern "C" int snk_shard_prune_plan(snk_ctx* ctx, uint64_t* h_qcount, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prune_plan: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    // bucket-local prune (snk_local.hip); only neighbours owned by another rank become queries
    snk_bl_state& B = S->bl;
    memset(&B, 0, sizeof B);
    B.tab = &S->tab;
    B.K = S->params.K; B.rank = S->rank; B.world = S->world; B.NB_total = S->NB_total; B.NBl = S->NBl;
    B.do_prune = S->params.min_freq > 1 ? 1u : 0u;
    std::vector<unsigned long long> h(S->world);
    int rc = snk_bl_dist_plan(ctx, st, &B, h_qcount ? h.data() : nullptr, err, errcap);      // NULL: the counts stay on the device (B.qcount)
    if (rc) return rc;
    if (h_qcount) for (uint32_t r = 0; r < S->world; ++r) h_qcount[r] = h[r];
    // the answering / applying kernels of the global sharded stage work on the same arrays
    snk_dist_graph& g = S->g;
    memset(&g, 0, sizeof g);
    g.K = B.K; g.rank = B.rank; g.world = B.world; g.NB_total = B.NB_total; g.NBl = B.NBl; g.do_prune = B.do_prune;
    g.n = S->tab.n; g.keys = S->tab.keys; g.vals = S->tab.vals;
    g.index = B.index; g.index_mask = B.index_mask;
    g.ctx = B.ctx; g.counts = B.counts; g.rq_idx = B.rq_idx; g.rq_meta = B.rq_meta;
    return SNK_OK;
}
extern "C" int snk_shard_prune_fill(snk_ctx* ctx, const void* d_qoff, void* d_qbuf, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_qoff) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prune_fill: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    return snk_bl_dist_fill(ctx, stream ? (hipStream_t)stream : ctx->stream, &S->bl, (const unsigned long long*)d_qoff, d_qbuf, err, errcap);
}
extern "C" int snk_shard_prune_answer(snk_ctx* ctx, const void* d_queries, uint64_t nq, void* d_ans, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prune_answer: no session");
    snk_shard_state* S = state_of(ctx);
    return snk_dist_answer(ctx, stream ? (hipStream_t)stream : ctx->stream, &S->g, d_queries, nq, d_ans, err, errcap);
}
extern "C" int snk_shard_prune_apply(snk_ctx* ctx, const void* d_qbuf, const void* d_ans, uint64_t nq, const void* d_qoff, void* stream,
                                     char* err, size_t errcap) {
    if (!ctx || !ctx->shard) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prune_apply: no session");
    snk_shard_state* S = state_of(ctx);
    return snk_dist_apply(ctx, stream ? (hipStream_t)stream : ctx->stream, &S->g, d_qbuf, d_ans, nq, (const unsigned long long*)d_qoff, err, errcap);
}

extern "C" int snk_shard_fragments(snk_ctx* ctx, const void* d_node_off, uint64_t my_node_off, snk_shard_frags* out, void* stream,
                                   char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !out || !d_node_off) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_fragments: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    snk_frag_out fo;
    int rc = snk_bl_dist_fragments(ctx, st, &S->bl, (const unsigned long long*)d_node_off, my_node_off, &fo, err, errcap);
    if (rc) return rc;
    S->frags = fo;
    S->d_node_off = (const unsigned long long*)d_node_off;
    S->my_node_off = my_node_off;
    memset(out, 0, sizeof *out);
    out->n_kmers = S->tab.n;
    out->keys = S->tab.keys;
    out->counts = S->bl.counts;
    out->ctx = S->bl.ctx;
    out->spectrum = fo.spectrum;
    out->spectrum_bins = fo.spectrum_bins;
    out->n_frags = fo.n_frags;
    out->total_bases = fo.total_bases;
    out->nk = fo.nk;
    out->hl_self = fo.hl_self;
    out->hl_nb = fo.hl_nb;
    out->boff = fo.boff;
    out->bases = fo.bases;
    out->n_circles = fo.n_circles;
    out->rank_rounds = fo.rank_rounds;
    out->buckets_split = S->tab.buckets_split;
    out->max_slots_used = S->tab.max_slots_used;
    out->count_ms = S->tab.count_ms;
    out->sort_ms = S->tab.sort_ms;
    out->count_kernel_ms = S->tab.count_kernel_ms;
    return SNK_OK;
}

// ---- fragment links decided on the owners (instead of a hash match over every end on rank 0)
extern "C" int snk_shard_links_plan(snk_ctx* ctx, uint64_t my_frag_off, uint64_t* h_qcount, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !h_qcount) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_links_plan: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    S->my_end_base = 2ull * my_frag_off;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc; S->lq_count = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc; S->lq_cursor = (unsigned long long*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->lq_count, 0, (S->world + 1) * 8ull, st));
    if ((rc = snk_dist_links_query(ctx, st, false, &S->frags, S->d_node_off, S->world, S->my_end_base, S->lq_count, nullptr, err, errcap))) return rc;
    std::vector<unsigned long long> h(S->world);
    SNK_HIP_TRY(hipMemcpyAsync(h.data(), S->lq_count, S->world * 8ull, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    for (uint32_t r = 0; r < S->world; ++r) h_qcount[r] = h[r];
    return SNK_OK;
}
extern "C" int snk_shard_links_fill(snk_ctx* ctx, const void* d_qoff, void* d_qbuf, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_qoff) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_links_fill: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    SNK_HIP_TRY(hipMemcpyAsync(S->lq_cursor, d_qoff, (S->world + 1) * 8ull, hipMemcpyDeviceToDevice, st));
    return snk_dist_links_query(ctx, st, true, &S->frags, S->d_node_off, S->world, S->my_end_base, S->lq_cursor, d_qbuf, err, errcap);
}
extern "C" int snk_shard_links_answer(snk_ctx* ctx, const void* d_queries, uint64_t nq, void* d_ans, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_links_answer: no session");
    snk_shard_state* S = state_of(ctx);
    return snk_dist_links_answer(ctx, stream ? (hipStream_t)stream : ctx->stream, &S->frags, d_queries, nq, 2ull * S->my_node_off, 2ull * S->tab.n,
                                 S->my_end_base, d_ans, err, errcap);
}
extern "C" int snk_shard_links_apply(snk_ctx* ctx, const void* d_qbuf, const void* d_ans, uint64_t nq, const void** d_flink, void* stream,
                                     char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_flink) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_links_apply: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    uint32_t* fl = nullptr;
    int rc = snk_dist_links_apply(ctx, stream ? (hipStream_t)stream : ctx->stream, &S->frags, d_qbuf, d_ans, nq, &fl, err, errcap);
    if (rc) return rc;
    *d_flink = fl;
    return SNK_OK;
}

extern "C" int snk_shard_join(snk_ctx* ctx, uint32_t K, uint64_t n_frags, const void* d_nk, const void* d_hl_self, const void* d_hl_nb,
                              const void* d_boff, const void* d_bases, uint64_t total_bases, snk_shard_unitigs* out, void* stream,
                              char* err, size_t errcap) {
    return snk_shard_join_linked(ctx, K, n_frags, d_nk, d_hl_self, d_hl_nb, nullptr, d_boff, d_bases, total_bases, out, stream, err, errcap);
}

extern "C" int snk_shard_join_linked(snk_ctx* ctx, uint32_t K, uint64_t n_frags, const void* d_nk, const void* d_hl_self, const void* d_hl_nb,
                                     void* d_flink, const void* d_boff, const void* d_bases, uint64_t total_bases, snk_shard_unitigs* out,
                                     void* stream, char* err, size_t errcap) {
    if (!ctx || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_join: NULL argument");
    if (!d_flink && (!d_hl_self || !d_hl_nb)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_join: need half links or links");
    if (K != 48 && K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    snk_join_out jo;
    int rc = snk_dist_join(ctx, st, K, n_frags, (const uint32_t*)d_nk, (const unsigned long long*)d_hl_self,
                           (const unsigned long long*)d_hl_nb, (const uint64_t*)d_boff, (const uint8_t*)d_bases, total_bases, &jo, err, errcap, nullptr, nullptr, 0,
                           (uint32_t*)d_flink);
    if (rc) return rc;
    out->n_unitigs = jo.n_unitigs;
    out->total_bases = jo.total_bases;
    out->unitig_off = jo.unitig_off;
    out->unitig_bases = jo.unitig_bases;
    out->unitig_circular = jo.unitig_circular;
    out->n_circles = jo.n_circles;
    out->rank_rounds = jo.rank_rounds;
    return SNK_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Owner-side join (no rank-0 funnel).  tada's MAIN_ASM_SN is one process (lib/tada/src/cmd_main_asm.rs:25-89,184-193) and so
// was the first version of this path: every fragment travelled to rank 0, which ranked and wrote all unitigs -- serial in
// the size of the job.  Now the only thing every rank sees of the whole job is the LINK structure (8 + 4 bytes per fragment,
// all-gathered): it ranks the fragment lists, places its own fragments (unitig = the smaller terminal state, offset,
// strand) and sends each fragment's bases to the rank that owns the unitig's head fragment, which writes the unitig.
//   snk_shard_place       links + k-mer counts of all ranks -> placement of my fragments, fragments / bases per owner
//   snk_shard_route_fill  32-byte headers + bases grouped by owner (the two send buffers of one all-to-all each)
//   snk_shard_emit        headers + bases received -> this rank's unitigs (canonical, circles rotated, ordered)
namespace {
constexpr unsigned long long RT_RC = 1ull, RT_CIRC = 2ull, RT_HEAD = 4ull;
__device__ __forceinline__ uint32_t owner_of_frag(const unsigned long long* __restrict__ frag_off, uint32_t world, unsigned long long g) {
    uint32_t r = 0;
    while (r + 1 < world && g >= frag_off[r + 1]) ++r;
    return r;
}
// FILL = false: count fragments and base bytes per owner; true: write header and bases at reserved positions.
// header (4 x u64): pid | gfid << 32;  nk | flags << 32;  koff (heads: N);  offset of the bases inside the owner's segment.
// One fragment per thread for the bookkeeping: the workgroup adds up its fragments and bytes per owner in LDS and goes to
// the `world` device counters once per owner (2 M fragments on the same few addresses, one device atomic each, took 100 ms);
// then the bases are copied by 8 lanes per fragment.
constexpr int ROUTE_TILES = 16;      // tiles of 256 fragments per workgroup: ONE reservation per owner for all of them (see jlink_query_kernel, snk_graph.hip)
template <bool FILL>
__global__ void __launch_bounds__(256) route_kernel(uint64_t Fl, unsigned long long f0, const uint32_t* __restrict__ nk, const uint32_t* __restrict__ pl_pid,
                                                    const unsigned long long* __restrict__ pl_koff, const unsigned long long* __restrict__ pl_N,
                                                    const uint8_t* __restrict__ pl_circ, const unsigned long long* __restrict__ frag_off, uint32_t world,
                                                    uint32_t K, const uint64_t* __restrict__ boff, const uint8_t* __restrict__ bases,
                                                    unsigned long long* __restrict__ cnt_or_cur /* [2][world] */, unsigned long long* __restrict__ hdr,
                                                    uint8_t* __restrict__ bout, const unsigned long long* __restrict__ base_seg /* [world] byte offset of every owner's segment */) {
    extern __shared__ unsigned long long dynr[];          // [world] fragments -> reserved first header, [world] bytes -> reserved first byte, [2][world] running cursors
    __shared__ unsigned long long c_src[256], c_dst[256];
    __shared__ uint32_t c_len[256];
    unsigned long long* lfr = dynr;
    unsigned long long* lby = dynr + world;
    unsigned long long* cfr = dynr + 2 * world;
    unsigned long long* cby = dynr + 3 * world;
    for (uint32_t r = threadIdx.x; r < 4 * world; r += 256) dynr[r] = 0;
    __syncthreads();
    const uint64_t fbase = (uint64_t)blockIdx.x * 256 * ROUTE_TILES + threadIdx.x;
    for (int t = 0; t < ROUTE_TILES; ++t) {
        const uint64_t f = fbase + (uint64_t)t * 256;
        if (f < Fl) {
            const uint32_t owner = owner_of_frag(frag_off, world, pl_pid[f] >> 1);
            atomicAdd(&lfr[owner], 1ull);
            atomicAdd(&lby[owner], (unsigned long long)((uint64_t)nk[f] + K - 1));
        }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < world; r += 256) {
        const unsigned long long cf = lfr[r], cb = lby[r];
        if (cf) {
            const unsigned long long a = atomicAdd(&cnt_or_cur[r], cf), b = atomicAdd(&cnt_or_cur[world + r], cb);
            lfr[r] = a; lby[r] = b;
        }
    }
    if (!FILL) return;
    for (int t = 0; t < ROUTE_TILES; ++t) {
        __syncthreads();       // the reservations are in LDS / the copy of the tile before is through with c_src, c_dst, c_len
        const uint64_t f = fbase + (uint64_t)t * 256;
        c_len[threadIdx.x] = 0;
        if (f < Fl) {
            const uint32_t pid = pl_pid[f];
            const uint32_t owner = owner_of_frag(frag_off, world, pid >> 1);
            const uint64_t len = (uint64_t)nk[f] + K - 1;
            const unsigned long long slot = lfr[owner] + atomicAdd(&cfr[owner], 1ull), bo = lby[owner] + atomicAdd(&cby[owner], (unsigned long long)len);
            const unsigned long long ko = pl_koff[f];
            const unsigned long long g = f0 + f;
            const bool head = (ko & ~(1ull << 63)) == 0 && (pid >> 1) == (uint32_t)g;
            const unsigned long long flags = ((ko >> 63) ? RT_RC : 0ull) | (pl_circ[f] ? RT_CIRC : 0ull) | (head ? RT_HEAD : 0ull);
            hdr[4 * slot + 0] = (unsigned long long)pid | (g << 32);
            hdr[4 * slot + 1] = (unsigned long long)nk[f] | (flags << 32);
            hdr[4 * slot + 2] = head ? pl_N[f] : (ko & ~(1ull << 63));
            hdr[4 * slot + 3] = bo - base_seg[owner];
            c_src[threadIdx.x] = boff[f];
            c_dst[threadIdx.x] = bo;
            c_len[threadIdx.x] = (uint32_t)len;
        }
        __syncthreads();
        const uint32_t sub = threadIdx.x & 7u;
        for (uint32_t i = threadIdx.x >> 3; i < 256; i += 32) {
            const uint32_t n = c_len[i];
            const uint8_t* src = bases + c_src[i];
            uint8_t* dst = bout + c_dst[i];
            // bytes up to dst's first 4-byte boundary, dwords (read at byte alignment), the last bytes -- a byte per lane and instruction
            // made the copy instruction-bound (snk_graph.hip, jemit_kernel)
            struct __attribute__((packed)) u32_any { uint32_t v; };
            uint32_t head = (4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u;
            if (head > n) head = n;
            if (sub < head) dst[sub] = src[sub];
            const uint32_t ndw = (n - head) >> 2;
            for (uint32_t j = sub; j < ndw; j += 8) *reinterpret_cast<uint32_t*>(dst + head + 4 * j) = reinterpret_cast<const u32_any*>(src + head + 4 * j)->v;
            const uint32_t t0 = head + 4 * ndw;
            if (sub < n - t0) dst[t0 + sub] = src[t0 + sub];
        }
    }
}
// received headers -> the arrays snk_join_emit wants; hdr_seg / base_seg: first header / first base byte of every source rank
__global__ void __launch_bounds__(256) unroute_kernel(const unsigned long long* __restrict__ hdr, uint64_t F, const unsigned long long* __restrict__ hdr_seg,
                                                      const unsigned long long* __restrict__ base_seg, uint32_t world, uint32_t* __restrict__ nk,
                                                      uint32_t* __restrict__ gfid, uint32_t* __restrict__ pid, unsigned long long* __restrict__ koff,
                                                      unsigned long long* __restrict__ N, uint8_t* __restrict__ circ, uint64_t* __restrict__ boff) {
    const uint64_t f = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    uint32_t src = 0;
    while (src + 1 < world && f >= hdr_seg[src + 1]) ++src;
    const unsigned long long w0 = hdr[4 * f], w1 = hdr[4 * f + 1], w2 = hdr[4 * f + 2], w3 = hdr[4 * f + 3];
    const unsigned long long flags = w1 >> 32;
    pid[f] = (uint32_t)w0;
    gfid[f] = (uint32_t)(w0 >> 32);
    nk[f] = (uint32_t)w1;
    const bool head = flags & RT_HEAD;
    koff[f] = (head ? 0ull : w2) | ((flags & RT_RC) ? (1ull << 63) : 0ull);
    N[f] = head ? w2 : 0ull;
    circ[f] = (flags & RT_CIRC) ? 1 : 0;
    boff[f] = base_seg[src] + w3;
}
}  // namespace

// placement of this rank's fragments from a ranking (whole list: rk_f0 = 0; own states only: rk_f0 = my_frag_off) and the
// fragments / base bytes owed to every owner
static int place_and_count(snk_ctx* ctx, snk_shard_state* S, hipStream_t st, uint32_t K, const uint2* rk, uint64_t rk_f0, const uint8_t* circ,
                           const uint32_t* nk_all, const void* d_frag_off, uint64_t* h_frags_to, uint64_t* h_bases_to, char* err, size_t errcap) {
    const uint64_t Fl = S->frags.n_frags;
    int rc;
    if ((rc = snk_join_place(ctx, st, rk, nk_all, circ, S->my_frag_off, Fl, &S->pl, err, errcap, rk_f0))) return rc;
    void* q;
    if (!h_frags_to) {          // the caller routes in one pass into per-owner regions (snk_shard_step.hip): nothing to count
        if ((rc = snk_ctx_alloc(ctx, 2ull * (S->world + 1) * 8, &q, err, errcap))) return rc; S->rt_cursor = (unsigned long long*)q;
        return SNK_OK;
    }
    if ((rc = snk_ctx_alloc(ctx, 2ull * (S->world + 1) * 8, &q, err, errcap))) return rc; S->rt_count = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, 2ull * (S->world + 1) * 8, &q, err, errcap))) return rc; S->rt_cursor = (unsigned long long*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->rt_count, 0, 2ull * (S->world + 1) * 8, st));
    if (Fl) hipLaunchKernelGGL((route_kernel<false>), dim3((unsigned)((Fl + 256 * ROUTE_TILES - 1) / (256 * ROUTE_TILES))), dim3(256), 4ull * S->world * 8, st, Fl, (unsigned long long)S->my_frag_off, S->frags.nk, S->pl.pid, S->pl.koff,
                               S->pl.N, S->pl.circ, (const unsigned long long*)d_frag_off, S->world, K, S->frags.boff, S->frags.bases, S->rt_count, nullptr, nullptr, nullptr);
    SNK_HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> h(2 * S->world);
    SNK_HIP_TRY(hipMemcpyAsync(h.data(), S->rt_count, 2ull * S->world * 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    for (uint32_t r = 0; r < S->world; ++r) { h_frags_to[r] = h[r]; h_bases_to[r] = h[S->world + r]; }
    return SNK_OK;
}

extern "C" int snk_shard_place(snk_ctx* ctx, uint32_t K, uint64_t n_frags_total, const void* d_nk_all, void* d_flink_all, const void* d_frag_off,
                               uint64_t my_frag_off, uint64_t* h_frags_to /* [world] */, uint64_t* h_bases_to /* [world] */, void* stream, char* err,
                               size_t errcap) {
    if (!ctx || !ctx->shard || !d_frag_off || (!h_frags_to) != (!h_bases_to)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_place: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    if (2 * n_frags_total >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments in the job");
    S->my_frag_off = my_frag_off;
    S->n_frags_total = n_frags_total;
    const uint2* rk = nullptr;
    uint8_t* circ = nullptr;
    uint32_t nc = 0;
    int rc = snk_join_rank(ctx, st, n_frags_total, (const uint32_t*)d_nk_all, (uint32_t*)d_flink_all, &rk, &circ, &nc, &S->join_rounds, err, errcap, S->circ_all);
    if (rc) return rc;
    S->join_circles += nc;
    return place_and_count(ctx, S, st, K, rk, 0, circ, (const uint32_t*)d_nk_all, d_frag_off, h_frags_to, h_bases_to, err, errcap);
}

// ---- the ranking partitioned over the ranks (snk_graph.hip: snk_prank_*): begin -> [all-gather of 16 B per splitter] -> walk ->
// [all-to-all of 16 B per state] -> snk_shard_place_ranked.  *circles = 1 after the walk: a list is a circle, use snk_shard_place.
extern "C" int snk_shard_prank_begin(snk_ctx* ctx, uint64_t n_frags_total, const void* d_nk_all, void* d_flink_all, uint64_t my_frag_off,
                                     uint64_t* n_splitters, const void** d_w1_share, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !n_splitters || !d_w1_share) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prank_begin: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    if (2 * n_frags_total >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments in the job");
    S->my_frag_off = my_frag_off;
    S->n_frags_total = n_frags_total;
    S->nk_all = (const uint32_t*)d_nk_all;
    if (!S->circ_all) {          // first ranking attempt of this step (a second one follows a circle cut and keeps the marks)
        void* q;
        int rc0 = snk_ctx_alloc(ctx, 2 * n_frags_total + 16, &q, err, errcap);
        if (rc0) return rc0;
        S->circ_all = (uint8_t*)q;
        SNK_HIP_TRY(hipMemsetAsync(S->circ_all, 0, 2 * n_frags_total + 16, st));
    }
    int rc = snk_prank_begin(ctx, st, n_frags_total, (const uint32_t*)d_nk_all, (uint32_t*)d_flink_all, S->rank, S->world, &S->pr, err, errcap);
    if (rc) return rc;
    *n_splitters = S->pr.m;       // (the share is consumed by a collective on the same stream: nothing to wait for here)
    *d_w1_share = S->pr.w1_share;
    return SNK_OK;
}
extern "C" int snk_shard_prank_walk(snk_ctx* ctx, const void* d_w1_all, const void* d_frag_off, uint64_t* h_recs_to /* [world] */, uint32_t* circles,
                                    void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !circles) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prank_walk: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    uint32_t n_cut = 0;
    int rc = snk_prank_walk(ctx, st, &S->pr, (const uint4*)d_w1_all, circles, &S->join_rounds, err, errcap, S->circ_all, &n_cut);
    if (rc) return rc;
    S->join_circles += n_cut;
    if (h_recs_to) for (uint32_t r = 0; r < S->world; ++r) h_recs_to[r] = 0;
    if (*circles) return SNK_OK;       // 2: circles were cut in the replicated links, call snk_shard_prank_begin again; 1: use snk_shard_place
    if (!h_recs_to) return SNK_OK;       // one-pass routing into per-owner regions: S->pr.n_rec records, no count pass
    void* q;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc;
    unsigned long long* cnt = (unsigned long long*)q;
    SNK_HIP_TRY(hipMemsetAsync(cnt, 0, (S->world + 1) * 8ull, st));
    if ((rc = snk_prank_route(ctx, st, &S->pr, false, (const unsigned long long*)d_frag_off, S->world, cnt, nullptr, err, errcap))) return rc;
    std::vector<unsigned long long> h(S->world);
    SNK_HIP_TRY(hipMemcpyAsync(h.data(), cnt, S->world * 8ull, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    for (uint32_t r = 0; r < S->world; ++r) h_recs_to[r] = h[r];
    return SNK_OK;
}
extern "C" int snk_shard_prank_route(snk_ctx* ctx, const void* d_frag_off, const void* d_rec_off /* u64[world]: first record of every owner */, void* d_out,
                                     void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_rec_off) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prank_route: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc;
    SNK_HIP_TRY(hipMemcpyAsync(q, d_rec_off, S->world * 8ull, hipMemcpyDeviceToDevice, st));
    S->pr_cursor = (unsigned long long*)q;
    return snk_prank_route(ctx, st, &S->pr, true, (const unsigned long long*)d_frag_off, S->world, (unsigned long long*)q, d_out, err, errcap);
}
extern "C" int snk_shard_place_ranked(snk_ctx* ctx, uint32_t K, const void* d_recs, uint64_t n_recs, const void* d_frag_off, uint64_t* h_frags_to,
                                      uint64_t* h_bases_to, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_frag_off || (!h_frags_to) != (!h_bases_to)) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_place_ranked: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint2* rk = nullptr;
    int rc = snk_prank_apply(ctx, st, d_recs, n_recs, 2ull * S->my_frag_off, 2ull * S->frags.n_frags, &rk, err, errcap);
    if (rc) return rc;
    return place_and_count(ctx, S, st, K, rk, S->my_frag_off, S->join_circles ? S->circ_all : nullptr, S->nk_all, d_frag_off, h_frags_to, h_bases_to, err, errcap);
}

extern "C" int snk_shard_route_fill(snk_ctx* ctx, uint32_t K, const void* d_frag_off, const void* d_hdr_off /* u64[world]: first header of every owner */,
                                    const void* d_base_off /* u64[world]: first base byte of every owner */, void* d_hdr, void* d_bases, void* stream,
                                    char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_frag_off || !d_hdr_off || !d_base_off) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_route_fill: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint64_t Fl = S->frags.n_frags;
    SNK_HIP_TRY(hipMemcpyAsync(S->rt_cursor, d_hdr_off, S->world * 8ull, hipMemcpyDeviceToDevice, st));
    SNK_HIP_TRY(hipMemcpyAsync(S->rt_cursor + S->world, d_base_off, S->world * 8ull, hipMemcpyDeviceToDevice, st));
    if (Fl) hipLaunchKernelGGL((route_kernel<true>), dim3((unsigned)((Fl + 256 * ROUTE_TILES - 1) / (256 * ROUTE_TILES))), dim3(256), 4ull * S->world * 8, st, Fl, (unsigned long long)S->my_frag_off, S->frags.nk, S->pl.pid, S->pl.koff,
                               S->pl.N, S->pl.circ, (const unsigned long long*)d_frag_off, S->world, K, S->frags.boff, S->frags.bases, S->rt_cursor,
                               (unsigned long long*)d_hdr, (uint8_t*)d_bases, (const unsigned long long*)d_base_off);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

extern "C" int snk_shard_emit(snk_ctx* ctx, uint32_t K, uint64_t n_recv, const void* d_hdr, const void* d_hdr_seg /* u64[world+1] */,
                              const void* d_base_seg /* u64[world+1] */, const void* d_bases, snk_shard_unitigs* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_emit: NULL argument / no session");
    if (K != 48 && K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    uint32_t *nk, *gfid, *pid;
    unsigned long long *koff, *N;
    uint8_t* circ;
    uint64_t* boff;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1) * 4, &q, err, errcap))) return rc; nk = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1) * 4, &q, err, errcap))) return rc; gfid = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1) * 4, &q, err, errcap))) return rc; pid = (uint32_t*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1) * 8, &q, err, errcap))) return rc; koff = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1) * 8, &q, err, errcap))) return rc; N = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 1), &q, err, errcap))) return rc; circ = (uint8_t*)q;
    if ((rc = snk_ctx_alloc(ctx, (n_recv + 2) * 8, &q, err, errcap))) return rc; boff = (uint64_t*)q;
    if (n_recv) hipLaunchKernelGGL(unroute_kernel, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, st, (const unsigned long long*)d_hdr, n_recv,
                                   (const unsigned long long*)d_hdr_seg, (const unsigned long long*)d_base_seg, S->world, nk, gfid, pid, koff, N, circ, boff);
    SNK_HIP_TRY(hipGetLastError());
    snk_join_out jo;
    // every pid received here is a terminal state of one of this rank's own fragments
    rc = snk_join_emit(ctx, st, K, n_recv, nk, gfid, pid, koff, N, circ, (uint32_t)(2 * S->my_frag_off), 2 * S->frags.n_frags, boff, (const uint8_t*)d_bases,
                       nullptr, &jo, err, errcap);
    if (rc) return rc;
    out->n_unitigs = jo.n_unitigs;
    out->total_bases = jo.total_bases;
    out->unitig_off = jo.unitig_off;
    out->unitig_bases = jo.unitig_bases;
    out->unitig_circular = jo.unitig_circular;
    out->n_circles = S->join_circles;
    out->rank_rounds = S->join_rounds;
    return SNK_OK;
}

// ---- fragment bases on the wire: 2 bits per base (the gather to rank 0 is the only place where bases cross xGMI)
namespace {
__global__ void __launch_bounds__(256) pack2_kernel(const uint8_t* __restrict__ in, uint64_t n, uint8_t* __restrict__ out) {
    // one thread per 16 bases -> one 32-bit word (base j of a byte at bits 2*(j%4), as in the .bv packing)
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t b0 = t * 16;
    if (b0 >= n) return;
    uint32_t w = 0;
    if (b0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + b0);
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= ((q[k] >> (8 * j)) & 3u) << (2 * (4 * k + j));
    } else {
        for (uint64_t j = 0; b0 + j < n; ++j) w |= (uint32_t)(in[b0 + j] & 3u) << (2 * j);
    }
    reinterpret_cast<uint32_t*>(out)[t] = w;
}
__global__ void __launch_bounds__(256) unpack2_kernel(const uint8_t* __restrict__ in, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t b0 = t * 16;
    if (b0 >= n) return;
    const uint32_t w = reinterpret_cast<const uint32_t*>(in)[t];
    if (b0 + 16 <= n && ((uintptr_t)(out + b0) & 15u) == 0) {
        uint32_t q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) q[k] |= ((w >> (2 * (4 * k + j))) & 3u) << (8 * j);
        }
        *reinterpret_cast<uint4*>(out + b0) = make_uint4(q[0], q[1], q[2], q[3]);
    } else {
        for (uint64_t j = 0; j < 16 && b0 + j < n; ++j) out[b0 + j] = (uint8_t)((w >> (2 * j)) & 3u);
    }
}
}  // namespace

extern "C" uint64_t snk_pack2_bytes(uint64_t n_bases) { return ((n_bases + 15) / 16) * 4; }

extern "C" int snk_dev_pack2(snk_ctx* ctx, const void* d_bases, uint64_t n_bases, void* d_packed, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx || (n_bases && (!d_bases || !d_packed))) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_pack2: NULL argument");
    if (((uintptr_t)d_bases & 15u) || ((uintptr_t)d_packed & 3u)) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_pack2: bases must be 16-byte, packed 4-byte aligned");
    if (n_bases == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint64_t nt = (n_bases + 15) / 16;
    hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d_bases, n_bases, (uint8_t*)d_packed);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

extern "C" int snk_dev_unpack2(snk_ctx* ctx, const void* d_packed, uint64_t n_bases, void* d_bases, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx || (n_bases && (!d_bases || !d_packed))) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_unpack2: NULL argument");
    if ((uintptr_t)d_packed & 3u) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_unpack2: packed input must be 4-byte aligned");
    if (n_bases == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint64_t nt = (n_bases + 15) / 16;
    hipLaunchKernelGGL(unpack2_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d_packed, n_bases, (uint8_t*)d_bases);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
