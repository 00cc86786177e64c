// snk_shard_phases.h -- the phases of the minimiser-sharded step (snk_shard_step.hip drives them; snk_dist.hip / snk_graph.hip implement
// them).  Library-internal since round 4: until round 2 a Python loop called them one by one through the C ABI; nothing outside
// libsnk does any more, so they are no longer exported (hidden visibility) and no longer part of include/snk.h.
#pragma once
#include "../../include/snk.h"

#pragma GCC visibility push(hidden)
#ifdef __cplusplus
extern "C" {
#endif
/* ---- minimiser-sharded multi-GPU path ------------------------------------------------
 * One process per GPU.  The k-mer space is cut into NB_total minimiser buckets; rank r owns buckets
 * [r*NB_total/world, (r+1)*NB_total/world).  The host runs the exchanges between the stages (RCCL
 * all-to-all over xGMI); the reference's counterpart is the shardio file exchange + per-shard assembly +
 * global join of tada (rust-shardio/src/shard.rs:184-211,488-493; cmd_shard_asm.rs:37-94;
 * cmd_main_asm.rs:25-89) and the in-memory swizzle of MapReduceEngine.h:362-385.  Device pointers; pointers
 * returned by a stage stay valid until the next snk_shard_hist on that context. */
/* trim + supermer histogram over all NB_total buckets (d_hist: u32[NB_total]) */
int snk_shard_hist(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, uint32_t rank, uint32_t world,
                   uint32_t NB_total, void* d_hist, uint64_t* n_instances, void* stream, char* err, size_t errcap);
/* d_offsets: u32[NB_total+1] exclusive scan of d_hist; d_records: 32 bytes per supermer, bucket-major */
int snk_shard_scatter(snk_ctx* ctx, const void* d_offsets, void* d_records, void* stream, char* err, size_t errcap);
/* count the records received for my buckets: d_seg_off u64[world*(NB_total/world+1)] absolute record offsets */
int snk_shard_count(snk_ctx* ctx, const void* d_records, const void* d_seg_off, uint64_t n_inst_hint, int has_bc,
                    uint64_t* n_kmers, void* stream, char* err, size_t errcap);
/* the same, counting the local buckets in n_ranges ranges [bounds[r], bounds[r+1]) (bounds[0] = 0, bounds[n] = local
 * buckets): ready(user, r) is called right before range r is launched, so the caller can make the stream wait for the
 * records of that range while the ranges before it are being counted (exchange overlapped with counting). */
int snk_shard_count_ranged(snk_ctx* ctx, const void* d_records, const void* d_seg_off, uint64_t n_inst_hint, int has_bc,
                           uint32_t n_ranges, const uint32_t* bounds, int (*ready)(void* user, uint32_t r), void* user,
                           uint64_t* n_kmers, void* stream, char* err, size_t errcap);
/* adjacency prune with remote membership queries (24 bytes each, answers 4 bytes each) */
int snk_shard_prune_plan(snk_ctx* ctx, uint64_t* h_qcount /* [world] */, void* stream, char* err, size_t errcap);
int snk_shard_prune_fill(snk_ctx* ctx, const void* d_qoff /* u64[world+1] */, void* d_qbuf, void* stream, char* err, size_t errcap);
int snk_shard_prune_answer(snk_ctx* ctx, const void* d_queries, uint64_t nq, void* d_ans, void* stream, char* err, size_t errcap);
int snk_shard_prune_apply(snk_ctx* ctx, const void* d_qbuf, const void* d_ans, uint64_t nq, const void* d_qoff, void* stream,
                          char* err, size_t errcap);
typedef struct snk_shard_frags {
    uint64_t n_kmers;            /* this rank's share of the retained table */
    const void* keys;            /* as snk_dev_result.keys, in bucket order (not sorted) */
    const void* counts;
    const void* ctx;
    const void* spectrum;
    uint32_t spectrum_bins;
    uint32_t n_circles;
    uint64_t n_frags;            /* local unitig fragments (tada's sedges) */
    uint64_t total_bases;
    const void* nk;              /* u32[n_frags] k-mers per fragment */
    const void* hl_self;         /* u64[2*n_frags] global state id of each fragment end */
    const void* hl_nb;           /* u64[2*n_frags] global state id the end wants to link to, or ~0 */
    const void* boff;            /* u64[n_frags] offset of every fragment's bases (nk + K - 1 of them) in `bases` */
    const void* bases;           /* u8 base codes */
    uint32_t rank_rounds, buckets_split, max_slots_used, reserved;
    float count_ms, sort_ms, count_kernel_ms, reserved_f;
} snk_shard_frags;
/* d_node_off: u64[world+1] exclusive scan of the ranks' n_kmers (global node numbering) */
int snk_shard_fragments(snk_ctx* ctx, const void* d_node_off, uint64_t my_node_off, snk_shard_frags* out, void* stream,
                        char* err, size_t errcap);
typedef struct snk_shard_unitigs {
    uint64_t n_unitigs, total_bases;
    const void* unitig_off;      /* u64[n_unitigs+1] */
    const void* unitig_bases;    /* u8 base codes, canonical orientation */
    const void* unitig_circular; /* u8[n_unitigs]: 1 = a circle that spanned fragments (already rotated to the reference's cut) */
    uint32_t n_circles, rank_rounds;
} snk_shard_unitigs;
/* Fragment links decided on the owners (optional; without it rank 0 matches the half links itself): every fragment
 * end whose half link names a state of rank q asks q (24 bytes), q answers with the global id of the fragment end that
 * sits on that state and points back, or ~0 (4 bytes).  my_frag_off = fragments of the ranks in front of this one.
 * After snk_shard_links_apply, *d_flink is u32[2*n_frags]: global fragment-end id linked to each local end, or ~0. */
int snk_shard_links_plan(snk_ctx* ctx, uint64_t my_frag_off, uint64_t* h_qcount /* [world] */, void* stream, char* err, size_t errcap);
int snk_shard_links_fill(snk_ctx* ctx, const void* d_qoff /* u64[world+1] */, void* d_qbuf, void* stream, char* err, size_t errcap);
int snk_shard_links_answer(snk_ctx* ctx, const void* d_queries, uint64_t nq, void* d_ans, void* stream, char* err, size_t errcap);
int snk_shard_links_apply(snk_ctx* ctx, const void* d_qbuf, const void* d_ans, uint64_t nq, const void** d_flink, void* stream,
                          char* err, size_t errcap);
/* rank 0: join the gathered fragments of every rank (tada MAIN_ASM_SN build_edges) */
int snk_shard_join(snk_ctx* ctx, uint32_t K, uint64_t n_frags, const void* d_nk, const void* d_hl_self, const void* d_hl_nb,
                   const void* d_boff, const void* d_bases, uint64_t total_bases, snk_shard_unitigs* out, void* stream,
                   char* err, size_t errcap);
/* the same with the links already decided (d_flink: u32[2*n_frags], modified; d_hl_self/d_hl_nb may then be NULL) */
int snk_shard_join_linked(snk_ctx* ctx, uint32_t K, uint64_t n_frags, const void* d_nk, const void* d_hl_self, const void* d_hl_nb,
                          void* d_flink, const void* d_boff, const void* d_bases, uint64_t total_bases, snk_shard_unitigs* out,
                          void* stream, char* err, size_t errcap);
/* Owner-side join (replaces the rank-0 funnel of tada's MAIN_ASM_SN, lib/tada/src/cmd_main_asm.rs:25-89,184-193; its
 * build_edges is lib/tada/src/debruijn.rs:733-776): every rank all-gathers the LINKS (u32[2F] global end ids from
 * snk_shard_links_apply) and k-mer counts (u32[F]) of all fragments, ranks the lists, places its own fragments and sends each
 * to the rank that owns its unitig's head fragment, which writes the unitig.  d_frag_off: u64[world+1] fragments in front of
 * every rank.  snk_shard_place -> fragments / base bytes this rank owes every owner; snk_shard_route_fill writes the 32-byte
 * headers and the bases grouped by owner at the given offsets (two all-to-alls follow); snk_shard_emit turns what arrived
 * (d_hdr_seg / d_base_seg: first header / first base byte of every source rank, u64[world+1]) into this rank's unitigs. */
int snk_shard_place(snk_ctx* ctx, uint32_t K, uint64_t n_frags_total, const void* d_nk_all, void* d_flink_all, const void* d_frag_off,
                    uint64_t my_frag_off, uint64_t* h_frags_to /* [world] */, uint64_t* h_bases_to /* [world] */, void* stream, char* err,
                    size_t errcap);
/* The ranking itself partitioned over the ranks (a rank walks 1/world of the splitters of the ruling-set scheme; only the
 * streaming set-up and the short splitter list are replicated): snk_shard_prank_begin -> all-gather of *d_w1_share (16 B per
 * splitter, shares [m*r/world, m*(r+1)/world)) -> snk_shard_prank_walk (*circles = 1: some list is a circle, rank the
 * replicated way with snk_shard_place; else records owed to every owner) -> snk_shard_prank_route -> all-to-all (16 B per
 * state) -> snk_shard_place_ranked (same outputs as snk_shard_place). */
int snk_shard_prank_begin(snk_ctx* ctx, uint64_t n_frags_total, const void* d_nk_all, void* d_flink_all, uint64_t my_frag_off,
                          uint64_t* n_splitters, const void** d_w1_share, void* stream, char* err, size_t errcap);
int snk_shard_prank_walk(snk_ctx* ctx, const void* d_w1_all, const void* d_frag_off, uint64_t* h_recs_to /* [world] */, uint32_t* circles,
                         void* stream, char* err, size_t errcap);
int snk_shard_prank_route(snk_ctx* ctx, const void* d_frag_off, const void* d_rec_off, void* d_out, void* stream, char* err, size_t errcap);
int snk_shard_place_ranked(snk_ctx* ctx, uint32_t K, const void* d_recs, uint64_t n_recs, const void* d_frag_off, uint64_t* h_frags_to,
                           uint64_t* h_bases_to, void* stream, char* err, size_t errcap);
int snk_shard_route_fill(snk_ctx* ctx, uint32_t K, const void* d_frag_off, const void* d_hdr_off, const void* d_base_off, void* d_hdr, void* d_bases,
                         void* stream, char* err, size_t errcap);
int snk_shard_emit(snk_ctx* ctx, uint32_t K, uint64_t n_recv, const void* d_hdr, const void* d_hdr_seg, const void* d_base_seg, const void* d_bases,
                   snk_shard_unitigs* out, void* stream, char* err, size_t errcap);
#ifdef __cplusplus
}
#endif
#pragma GCC visibility pop
