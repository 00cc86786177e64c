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
    uint32_t n_circles;
    uint32_t rank_rounds;
};

int snk_graph_sort(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t n, snk_u128* keys_in, uint64_t* vals_in,
                   snk_u128* keys_out, uint64_t* vals_out, char* err, size_t errcap);
int snk_graph_build(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_u128* keys, const uint64_t* vals, uint64_t n,
                    uint32_t do_prune, bool want_unitigs, snk_graph_out* out, char* err, size_t errcap);
