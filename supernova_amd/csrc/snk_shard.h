// snk_shard.h -- per-context session state of the minimiser-sharded path (snk_dist.hip, snk_shard_step.hip).
#pragma once
#include "snk_ctx.h"
#include "snk_graph.h"
#include "snk_kernels.h"
#include "snk_stages.h"
#include "snk_shard_phases.h"

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
    // owner-side join: placement of this rank's fragments, destination rank of each, route cursors
    snk_placement pl{};
    uint32_t* dest = nullptr;
    unsigned long long *rt_count = nullptr, *rt_cursor = nullptr;      // [2][world]: fragments, base bytes
    unsigned long long my_frag_off = 0;
    uint64_t n_frags_total = 0;
    uint32_t join_circles = 0, join_rounds = 0;
    snk_prank pr{};
    uint8_t* circ_all = nullptr;               // [2 * fragments of the job] terminals made by cutting circles (replicated; reset per step)
    unsigned long long* pr_cursor = nullptr;   // [world] cursors of the rank-record routing (end positions after the fill)
    const uint32_t* nk_all = nullptr;
    snk_phase_timer* tm = nullptr;
    // a streamed step (snk_shard_stream_*): the rank's reads arrive slab by slab and are partitioned as they come
    snk_partition_job job{};
    bool job_open = false;
    uint32_t job_read_len = 0;
    int job_has_bc = 0;
    uint64_t job_reads_ub = 0, job_total_reads = 0;
    uint16_t* job_good_len = nullptr;
};


inline snk_shard_state* snk_shard_state_of(snk_ctx* ctx) {
    if (!ctx->shard) ctx->shard = new snk_shard_state();
    return static_cast<snk_shard_state*>(ctx->shard);
}
// trim + one-pass minimiser partition over all NB_total buckets of the job (what snk_shard_hist does before it copies the histogram out)
// the streamed variant of snk_shard_begin: open (sizes the job's slots), add a slab, adopt (closes the job: S->part as snk_shard_begin leaves it)
int snk_shard_job_open(snk_ctx* ctx, const snk_params* p, uint32_t rank, uint32_t world, uint32_t NB_total, uint32_t read_len, uint64_t reads_ub,
                       uint64_t total_reads, int has_bc, hipStream_t st, char* err, size_t errcap);
int snk_shard_job_add(snk_ctx* ctx, const snk_dev_reads* slab, hipStream_t st, char* err, size_t errcap);
int snk_shard_job_adopt(snk_ctx* ctx, uint64_t* n_instances, hipStream_t st, char* err, size_t errcap);
int snk_shard_begin(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, uint32_t rank, uint32_t world, uint32_t NB_total,
                    uint64_t* n_instances, hipStream_t st, char* err, size_t errcap);
