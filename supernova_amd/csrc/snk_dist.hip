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

struct snk_shard_state {
    snk_dev_reads reads;
    snk_params params;
    uint32_t rank = 0, world = 1, NB_total = 0, NBl = 0;
    const uint16_t* good_len = nullptr;
    uint32_t* status = nullptr;
    snk_partition part{};
    snk_table tab{};
    snk_dist_graph g{};
    snk_bl_state bl{};
    snk_frag_out frags{};              // this rank's fragments (valid from snk_shard_fragments on)
    const unsigned long long* d_node_off = nullptr;
    unsigned long long my_node_off = 0, my_end_base = 0;
    unsigned long long *lq_count = nullptr, *lq_cursor = nullptr;
    snk_phase_timer* tm = nullptr;
};

static snk_shard_state* state_of(snk_ctx* ctx) {
    if (!ctx->shard) ctx->shard = new snk_shard_state();
    return static_cast<snk_shard_state*>(ctx->shard);
}
void snk_shard_state_free(void* p) { delete static_cast<snk_shard_state*>(p); }

extern "C" int snk_shard_hist(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, uint32_t rank, uint32_t world,
                              uint32_t NB_total, void* d_hist, uint64_t* n_instances, void* stream, char* err, size_t errcap) {
    if (!ctx || !in || !p || !d_hist) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: NULL argument");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if (p->min_bc > 2) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: the device barcode rule supports 0, 1, 2", p->min_bc);
    if (world == 0 || rank >= world || NB_total == 0 || NB_total % world) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: NB_total must be a positive multiple of world");
    if (world > 0x7FFF) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "world > 32767");
    if (!in->rows || in->row_words * 16 < in->read_len || in->read_len > 256 || (!in->quals && !in->good_len))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_hist: bad reads");
    SNK_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    snk_ctx_release_scratch(ctx);
    snk_shard_state* S = state_of(ctx);
    S->reads = *in;
    S->params = *p;
    S->rank = rank; S->world = world; S->NB_total = NB_total; S->NBl = NB_total / world;
    const uint16_t* good_len = (const uint16_t*)in->good_len;
    int rc;
    if (!good_len) {
        void* gl;
        if ((rc = snk_ctx_alloc(ctx, in->n_reads * 2 + 2, &gl, err, errcap))) return rc;
        rc = snk_dev_trim(ctx, in->quals, in->qstride, in->lens, in->read_len, in->n_reads, p->K, p->min_qual, gl, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
        good_len = (const uint16_t*)gl;
    }
    S->good_len = good_len;
    void* q;
    if ((rc = snk_ctx_alloc(ctx, 64, &q, err, errcap))) return rc; S->status = (uint32_t*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->status, 0, 64, st));
    // one partition pass into fixed-capacity bucket slots (as on one GPU); the histogram is its cursor array, the
    // "scatter" stage compacts the slots into the caller's exact, destination-contiguous send buffer
    unsigned long long h_plan[2] = {0, 0};
    if ((rc = snk_stage_partition_plan(ctx, st, p->K, good_len, in->n_reads, h_plan, err, errcap))) return rc;
    if ((rc = snk_stage_partition(ctx, st, p->K, in, good_len, NB_total, h_plan[0], h_plan[1], false, S->status, &S->part, err, errcap))) return rc;
    SNK_HIP_TRY(hipMemcpyAsync(d_hist, S->part.cursor, (size_t)NB_total * 4, hipMemcpyDeviceToDevice, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    if (n_instances) *n_instances = h_plan[0];
    return SNK_OK;
}

extern "C" int snk_shard_scatter(snk_ctx* ctx, const void* d_offsets, void* d_records, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_offsets || !d_records) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_scatter: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    return snk_stage_partition_compact(ctx, st, &S->part, (const uint32_t*)d_offsets, d_records, err, errcap);
}

extern "C" int snk_shard_count(snk_ctx* ctx, const void* d_records, const void* d_seg_off, uint64_t n_inst_hint, int has_bc,
                               uint64_t* n_kmers, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_seg_off) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_count: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
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
    snk_count_ranges rg{n_ranges, bounds, ready, user};
    int rc = snk_stage_count_table(ctx, st, S->params.K, d_records, (const uint64_t*)d_seg_off, (const uint64_t*)d_seg_off + 1, S->NBl + 1, S->world, S->NBl, S->params.min_freq,
                                   has_bc ? S->params.min_bc : 0u, 0u, n_inst_hint, S->status, false, &S->tab, err, errcap, n_ranges ? &rg : nullptr);
    if (rc) return rc;
    if (n_kmers) *n_kmers = S->tab.n;
    return SNK_OK;
}

extern "C" int snk_shard_prune_plan(snk_ctx* ctx, uint64_t* h_qcount, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !h_qcount) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_prune_plan: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // bucket-local prune (snk_local.hip); only neighbours owned by another rank become queries
    snk_bl_state& B = S->bl;
    memset(&B, 0, sizeof B);
    B.tab = &S->tab;
    B.K = S->params.K; B.rank = S->rank; B.world = S->world; B.NB_total = S->NB_total; B.NBl = S->NBl;
    B.do_prune = S->params.min_freq > 1 ? 1u : 0u;
    std::vector<unsigned long long> h(S->world);
    int rc = snk_bl_dist_plan(ctx, st, &B, h.data(), err, errcap);
    if (rc) return rc;
    for (uint32_t r = 0; r < S->world; ++r) h_qcount[r] = h[r];
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
    S->my_end_base = 2ull * my_frag_off;
    void* q;
    int rc;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc; S->lq_count = (unsigned long long*)q;
    if ((rc = snk_ctx_alloc(ctx, (S->world + 1) * 8ull, &q, err, errcap))) return rc; S->lq_cursor = (unsigned long long*)q;
    SNK_HIP_TRY(hipMemsetAsync(S->lq_count, 0, (S->world + 1) * 8ull, st));
    if ((rc = snk_dist_links_query(ctx, st, false, &S->frags, S->d_node_off, S->world, S->my_end_base, S->lq_count, nullptr, err, errcap))) return rc;
    std::vector<unsigned long long> h(S->world);
    SNK_HIP_TRY(hipMemcpyAsync(h.data(), S->lq_count, S->world * 8ull, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t r = 0; r < S->world; ++r) h_qcount[r] = h[r];
    return SNK_OK;
}
extern "C" int snk_shard_links_fill(snk_ctx* ctx, const void* d_qoff, void* d_qbuf, void* stream, char* err, size_t errcap) {
    if (!ctx || !ctx->shard || !d_qoff) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_links_fill: NULL argument / no session");
    snk_shard_state* S = state_of(ctx);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
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
    const uint64_t nt = (n_bases + 15) / 16;
    hipLaunchKernelGGL(unpack2_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d_packed, n_bases, (uint8_t*)d_bases);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
