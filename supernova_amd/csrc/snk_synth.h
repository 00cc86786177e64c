// snk_synth.h -- counter-based synthetic linked-read generator, host+device.
//
// Not a reference component: the reference ships no generator for this path (its Rust sim_tests
// use unseeded RNGs, lib/tada/src/sim_tests.rs:73-109).  Every draw is a pure function of
// (seed, stream, counter) so that the host (tests, oracle input) and the device (bench at 1e8 reads)
// produce bit-identical reads without storing a genome.
#pragma once
#include "../../include/snk.h"
#include "snk_common.h"

enum { SNK_ST_GENOME = 1, SNK_ST_PAIR = 2, SNK_ST_PAIR2 = 3, SNK_ST_MOL = 4, SNK_ST_ERR = 5, SNK_ST_ERRPOS = 6, SNK_ST_TAIL = 7 };

SNK_HD uint32_t snk_genome_base(uint64_t seed, uint64_t p) {
    uint64_t w = snk_rng(seed, SNK_ST_GENOME, p >> 5);
    return (uint32_t)(w >> (2 * (p & 31))) & 3u;
}

struct snk_read_plan {
    uint64_t start;   // genome position of the read's leftmost base on the forward strand
    uint32_t rc;      // 1: read = reverse complement of genome[start, start+len)
    int32_t bc;
    uint32_t n_err;
    uint32_t err_pos[4];
    uint32_t err_sub[4];
    uint32_t tail;    // number of trailing Q2 bases
};

SNK_HD snk_read_plan snk_synth_plan(const snk_synth_params& sp, uint64_t r) {
    snk_read_plan pl;
    const uint64_t G = sp.genome_len;
    const uint32_t L = sp.read_len;
    uint32_t mol_len = sp.mol_len < G ? sp.mol_len : (uint32_t)G;
    uint64_t q = r >> 1;
    uint32_t mate = (uint32_t)(r & 1);
    uint64_t bci = q / sp.pairs_per_bc;
    uint64_t hq = snk_rng(sp.seed, SNK_ST_PAIR, q);
    pl.bc = ((hq & 0xFFFFF) * 1000000ull >> 20) < sp.unbarcoded_ppm ? 0 : (int32_t)(bci + 1);
    uint32_t mol = (uint32_t)((hq >> 20) % sp.mols_per_bc);
    uint64_t hm = snk_rng(sp.seed, SNK_ST_MOL, bci * sp.mols_per_bc + mol);
    uint64_t mol_start = (hm >> 1) % (G - mol_len + 1);
    uint32_t strand = (uint32_t)(hm & 1);
    uint32_t ins = sp.insert_min + (uint32_t)((hq >> 40) % sp.insert_span);
    if (ins > mol_len) ins = mol_len;
    if (ins < L) ins = L;
    uint64_t h2 = snk_rng(sp.seed, SNK_ST_PAIR2, q);
    uint64_t off = h2 % (mol_len - ins + 1);
    uint64_t frag = mol_start + off;
    if ((mate ^ strand) == 0) { pl.start = frag; pl.rc = 0; }
    else { pl.start = frag + ins - L; pl.rc = 1; }
    uint64_t he = snk_rng(sp.seed, SNK_ST_ERR, r);
    uint32_t u = (uint32_t)he;
    uint32_t ne = 0;
    if (sp.sub_ppm) { while (ne < 4 && u > sp.err_cdf[ne]) ++ne; }
    pl.n_err = ne;
    for (uint32_t j = 0; j < 4; ++j) {
        uint64_t hp = snk_rng(sp.seed, SNK_ST_ERRPOS, r * 4 + j);
        pl.err_pos[j] = (uint32_t)(hp % L);
        pl.err_sub[j] = 1 + (uint32_t)((hp >> 32) % 3);
    }
    uint64_t ht = snk_rng(sp.seed, SNK_ST_TAIL, r);
    pl.tail = (((ht & 0xFFFFF) * 1000000ull >> 20) < sp.lowq_tail_ppm) ? (uint32_t)((ht >> 32) % (sp.tail_max + 1)) : 0;
    return pl;
}

// base i (read orientation) before substitutions
SNK_HD uint32_t snk_synth_clean_base(const snk_synth_params& sp, const snk_read_plan& pl, uint32_t i) {
    if (!pl.rc) return snk_genome_base(sp.seed, pl.start + i);
    return snk_genome_base(sp.seed, pl.start + (sp.read_len - 1 - i)) ^ 3u;
}

// generate one read: rows/quals may be null
SNK_HD void snk_synth_read(const snk_synth_params& sp, uint64_t r, uint32_t* row, uint32_t row_words, uint8_t* qual,
                           int32_t* bc) {
    snk_read_plan pl = snk_synth_plan(sp, r);
    const uint32_t L = sp.read_len;
    if (bc) *bc = pl.bc;
    if (row) {
        for (uint32_t w = 0; w < row_words; ++w) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 16; ++j) {
                uint32_t i = w * 16 + j;
                uint32_t b = 0;
                if (i < L) {
                    b = snk_synth_clean_base(sp, pl, i);
                    for (uint32_t e = 0; e < pl.n_err; ++e)
                        if (pl.err_pos[e] == i) b = (b + pl.err_sub[e]) & 3u;
                }
                v = (v << 2) | b;
            }
            row[w] = v;
        }
    }
    if (qual) {
        for (uint32_t i = 0; i < L; ++i) {
            uint32_t qv = 30;
            for (uint32_t e = 0; e < pl.n_err; ++e)
                if (pl.err_pos[e] == i) qv = 12;
            if (i + pl.tail >= L) qv = 2;
            qual[i] = (uint8_t)qv;
        }
    }
}
