// snk_graph.h -- internal interface of the graph stage (snk_graph.hip).
#pragma once
#include "snk_ctx.h"
#include "snk_kernels.h"

struct snk_graph_out {
    uint8_t* ctx;                 // [n] pruned context bytes (sorted k-mer order)
    uint32_t* counts;             // [n]
    unsigned long long* spectrum; // [spectrum_bins] retained k-mers per count (last bin = overflow)
    uint32_t spectrum_bins;
    uint64_t n_unitigs;
    uint64_t total_bases;
    uint64_t* unitig_off;         // [n_unitigs+1] offsets into unitig_bases
    uint8_t* unitig_bases;        // base codes, canonical orientation, ordered by head k-mer
    uint32_t* unitig_group;       // grouped runs: group of every unitig, else NULL
    uint32_t n_circles;
    uint32_t rank_rounds;
    uint64_t n_boundary;          // bucket-local path: k-mers with a neighbour outside their chunk
    uint64_t n_fragments;         // bucket-local path: local unitig fragments handed to the join
};

int snk_graph_sort(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t n, snk_u128* keys_in, uint64_t* vals_in,
                   snk_u128* keys_out, uint64_t* vals_out, char* err, size_t errcap);
int snk_graph_build(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_u128* keys, const uint64_t* vals, uint64_t n,
                    uint32_t do_prune, bool want_unitigs, snk_graph_out* out, char* err, size_t errcap);

// ---- sharded graph stage (snk_graph.hip, second half)
struct snk_dist_graph {
    uint32_t K, rank, world, NB_total, NBl, do_prune;
    uint64_t n;
    const snk_u128* keys;
    const uint64_t* vals;
    unsigned long long* index;
    uint64_t index_mask;
    uint8_t* ctx;
    uint32_t* counts;
    uint32_t* rq_idx;              // answers of the remote membership queries
    uint16_t* rq_meta;
};
struct snk_frag_out {
    uint64_t n_frags, total_bases;
    uint32_t* nk;
    unsigned long long* hl_self;
    unsigned long long* hl_nb;
    uint64_t* boff;
    uint8_t* bases;
    unsigned long long* spectrum;
    uint32_t spectrum_bins, n_circles, rank_rounds;
    uint32_t* fgroup;          // grouped runs: group of every fragment, else NULL
    uint32_t* sfrag;           // one-GPU runs: [2n] terminal state -> 2*fragment + end, else NULL
    uint32_t n_local_circles;  // smooth circles inside one chunk (their fragments sit behind the paths')
};
struct snk_join_out {
    uint64_t n_unitigs, total_bases;
    uint64_t* unitig_off;
    uint8_t* unitig_bases;
    uint8_t* unitig_circular;   // 1: a circle that spanned fragments (already rotated to the reference's cut)
    uint32_t* unitig_group;     // grouped runs (fgroup given): group of every unitig; unitigs ordered by (group, first K bases)
    uint32_t n_circles, rank_rounds;
};
// placement of fragments inside their unitigs (snk_graph.hip: jplace_kernel), device arrays
struct snk_placement {
    uint32_t* pid;               // terminal state that names the unitig
    unsigned long long* koff;    // k-mers in front of the fragment | 1 << 63: read from its right end
    unsigned long long* N;       // k-mers of the whole unitig
    uint8_t* circ;               // the unitig is a circle that was cut at an arbitrary fragment
};
int snk_join_rank(snk_ctx* ctx, hipStream_t st, uint64_t F, const uint32_t* nk, uint32_t* flink, const uint2** rk_out, uint8_t** circ_out,
                  uint32_t* n_circles, uint32_t* rounds, char* err, size_t errcap, uint8_t* circ_given = nullptr);
int snk_join_place(snk_ctx* ctx, hipStream_t st, const uint2* rk, const uint32_t* nk, const uint8_t* circ, uint64_t f0, uint64_t Fl,
                   snk_placement* pl, char* err, size_t errcap, uint64_t rk_f0 = 0);
// partitioned ranking of the job's fragment lists (sharded runs): replicated streaming setup, walks for a 1/world share
struct snk_prank {
    uint64_t ns, m, k0, k1, n_rec;
    const uint32_t* w;
    uint32_t* link;
    unsigned long long* wrec;
    uint32_t* spl_state;
    uint4* w1_share;           // [k1-k0] first walk of this rank's splitters: next splitter, distance, tail, states
    uint4* rec;                // [n_rec] second walk: state, distance, terminal, 0
};
int snk_prank_begin(snk_ctx* ctx, hipStream_t st, uint64_t F, const uint32_t* nk, uint32_t* link, uint32_t rank, uint32_t world, snk_prank* P, char* err,
                    size_t errcap);
int snk_prank_walk(snk_ctx* ctx, hipStream_t st, snk_prank* P, const uint4* w1_all, uint32_t* circles, uint32_t* rounds, char* err, size_t errcap,
                   uint8_t* circ = nullptr /* [ns]: given = circles are cut here (*circles = 2: begin again) */, uint32_t* n_cut = nullptr);
int snk_prank_route(snk_ctx* ctx, hipStream_t st, snk_prank* P, bool fill, const unsigned long long* d_frag_off, uint32_t world,
                    unsigned long long* d_cnt_or_cur, void* d_out, char* err, size_t errcap);
int snk_prank_apply(snk_ctx* ctx, hipStream_t st, const void* d_rec, uint64_t n, unsigned long long state_base, uint64_t n_local_states,
                    const uint2** rk_out, char* err, size_t errcap);
int snk_join_emit(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t F, const uint32_t* nk, const uint32_t* gfid, const uint32_t* pl_pid,
                  const unsigned long long* pl_koff, const unsigned long long* pl_N, const uint8_t* pl_circ, uint32_t pid_base, uint64_t n_pid,
                  const uint64_t* boff, const uint8_t* fbases, const uint32_t* fgroup, snk_join_out* out, char* err, size_t errcap);
int snk_dist_answer(snk_ctx* ctx, hipStream_t st, snk_dist_graph* g, const void* d_queries, uint64_t nq, void* d_ans, char* err,
                    size_t errcap);
int snk_dist_apply(snk_ctx* ctx, hipStream_t st, snk_dist_graph* g, const void* d_qbuf, const void* d_ans, uint64_t nq,
                   const unsigned long long* d_qoff, char* err, size_t errcap);
int snk_dist_join(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t F, const uint32_t* nk, const unsigned long long* hl_self,
                  const unsigned long long* hl_nb, const uint64_t* boff, const uint8_t* fbases, uint64_t total_fbases,
                  snk_join_out* out, char* err, size_t errcap, const uint32_t* fgroup = nullptr, const uint32_t* sfrag = nullptr,
                  uint64_t n_states = 0, uint32_t* flink_given = nullptr);
int snk_dist_links_query(snk_ctx* ctx, hipStream_t st, bool fill, const snk_frag_out* fr, const unsigned long long* d_node_off, uint32_t world,
                         unsigned long long my_end_base, unsigned long long* d_count_or_cursor, void* d_qbuf, char* err, size_t errcap);
int snk_dist_links_answer(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_queries, uint64_t nq, unsigned long long my_state_base,
                          uint64_t n_local_states, unsigned long long my_end_base, void* d_ans, char* err, size_t errcap);
int snk_dist_links_apply(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_qbuf, const void* d_ans, uint64_t nq, uint32_t** flink_out,
                         char* err, size_t errcap);

int snk_dist_links_apply_regions(snk_ctx* ctx, hipStream_t st, const snk_frag_out* fr, const void* d_qbuf, const void* d_ans, uint32_t world, uint64_t cap,
                                 const unsigned long long* counts, uint32_t** flink_out, char* err, size_t errcap);

// ---- bucket-local graph stage (snk_local.hip): table in chunk order -> pruned contexts + canonical unitigs
struct snk_table;
struct snk_bl_state {          // device arrays of the bucket-local stage (indexed by table position / chunk)
    const snk_table* tab;
    uint32_t K, rank, world, NB_total, NBl, do_prune, force_dist;
    uint4* desc;
    uint32_t nchunks, nbig;
    uint32_t* biglist;
    uint8_t *ctx, *pend, *premote;
    uint32_t *counts, *nbr, *rq;
    uint32_t* rq_idx;          // sharded: answers of the remote membership queries
    uint16_t* rq_meta;
    unsigned long long* index; // boundary k-mers only
    uint64_t index_mask, n_boundary;
    unsigned long long *qcount, *qcursor;
};
int snk_bl_dist_plan(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, unsigned long long* h_qcount, char* err, size_t errcap);
int snk_bl_dist_fill(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, const unsigned long long* d_qoff, void* d_qbuf, char* err, size_t errcap);
int snk_bl_dist_fragments(snk_ctx* ctx, hipStream_t st, snk_bl_state* B, const unsigned long long* d_node_off, unsigned long long my_node_off,
                          snk_frag_out* out, char* err, size_t errcap);
int snk_local_graph(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_table* tab, uint32_t do_prune, bool want_unitigs,
                    bool sort_table, bool grouped, snk_graph_out* out, snk_u128** keys_final, float* ms /* [5] or NULL */, char* err,
                    size_t errcap);
int snk_spectrum(snk_ctx* ctx, hipStream_t st, const uint32_t* counts, uint64_t n, unsigned long long** bins_out, uint32_t* nbins_out,
                 char* err, size_t errcap);
