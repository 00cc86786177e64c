// snk_synth.h -- counter-based synthetic linked-read generator, host+device.
//
// Not a reference component: the reference ships no generator for this path (its Rust sim_tests
// use unseeded RNGs, lib/tada/src/sim_tests.rs:73-109).  Every draw is a pure function of
// (seed, stream, counter) so that the host (tests, oracle input) and the device (bench at 1e8 reads)
// produce bit-identical reads without storing a genome.
#pragma once
#include "../../include/snk.h"
#include "snk_common.h"

enum { SNK_ST_GENOME = 1, SNK_ST_PAIR = 2, SNK_ST_PAIR2 = 3, SNK_ST_MOL = 4, SNK_ST_ERR = 5, SNK_ST_ERRPOS = 6, SNK_ST_TAIL = 7,
       SNK_ST_BLOCK = 8, SNK_ST_FAMILY = 9, SNK_ST_FAMMUT = 10, SNK_ST_STR = 11, SNK_ST_SEGDUP = 12 };

SNK_HD uint32_t snk_genome_raw(uint64_t seed, uint64_t p) {
    uint64_t w = snk_rng(seed, SNK_ST_GENOME, p >> 5);
    return (uint32_t)(w >> (2 * (p & 31))) & 3u;
}
// Repeat-rich genome (snk_synth_params.repeat_mode: bit 0 families, bit 1 segmental duplications, bit 2 tandem repeats, bit 3 poly-A; 15 = all): what real linked-read data have and an i.i.d. genome has not -- minimiser
// sites shared by thousands of loci, k-mers far above the coverage depth, branching graphs.  Everything is a pure function of the
// position, so host and device agree without a stored genome:
//   * segmental duplications: the odd 64-kb superblocks S with hash(S) % 4 == 0 carry, at offset 8192, an exact copy of the 5000
//     bases at the same offset of an EVEN superblock (even ones never redirect, so the copy is a copy of what is really there);
//   * per 4-kb block B (after the redirect), from hash(B):
//       60 %: bases [256, 556) are an element of one of four families: the family's consensus with 1, 2 or 3 % substitutions;
//        3 %: bases [1024, 1024 + 40..200) are a tandem repeat of a 1..6-base unit;
//        2 %: bases [2048, 2048 + 20..80) are a poly-A run;
//     everything else is the i.i.d. background.
SNK_HD uint32_t snk_genome_base(const snk_synth_params& sp, uint64_t p) {
    const uint64_t seed = sp.seed;
    if (!sp.repeat_mode) return snk_genome_raw(seed, p);
    // bit 4: a CROWDED minimiser space -- every fifth base is A, so a 16-mer takes one of 4^12 + 4 x 4^13 = 285 M values instead of 4.3 G
    // while every 48-mer stays unique (76 free bits): a 268 Mb genome then has ~1 site per canonical 16-mer value, as a human genome has
    // (3.1 G sites over 2.1 G values) -- the minimiser-sharing regime of configs 3-5 at a size one GPU holds at 56x
    if ((sp.repeat_mode & 16u) && p % 5 == 0) return 0u;
    const uint64_t S = p >> 16;
    const uint32_t so = (uint32_t)(p & 0xFFFF);
    if ((sp.repeat_mode & 2u) && (S & 1) && so >= 8192 && so < 8192 + 5000) {
        const uint64_t hs = snk_rng(seed, SNK_ST_SEGDUP, S);
        const uint64_t n_even = ((sp.genome_len >> 16) + 1) >> 1;        // even superblocks that lie inside the genome
        if ((hs & 3) == 0 && n_even) p = (((hs >> 8) % n_even) << 17) + so;
    }
    const uint64_t B = p >> 12;
    const uint32_t o = (uint32_t)(p & 4095);
    const uint64_t hb = snk_rng(seed, SNK_ST_BLOCK, B);
    if (o >= 256 && o < 556) {
        if ((sp.repeat_mode & 1u) && (hb & 0xFF) < 154) {
            const uint32_t fam = (uint32_t)(hb >> 8) & 3u, div_pct = 1u + (uint32_t)((hb >> 10) % 3);
            const uint32_t i = o - 256;
            uint32_t c = snk_genome_raw(snk_mix64(seed ^ (0xFA11ull + fam)), i);
            const uint64_t hm = snk_rng(seed, SNK_ST_FAMMUT, B * 512 + i);
            if ((uint32_t)(hm % 100) < div_pct) c = (c + 1u + (uint32_t)((hm >> 32) % 3)) & 3u;
            return c;
        }
    } else if (o >= 1024 && o < 1024 + 200) {
        if ((sp.repeat_mode & 4u) && ((hb >> 16) & 0xFF) < 8) {
            const uint32_t len = 40u + (uint32_t)((hb >> 24) & 0xFF) % 161u, unit = 1u + (uint32_t)((hb >> 32) & 0xFF) % 6u;
            if (o - 1024 < len) {
                const uint64_t hu = snk_rng(seed, SNK_ST_STR, B);
                return (uint32_t)(hu >> (2 * ((o - 1024) % unit))) & 3u;
            }
        }
    } else if (o >= 2048 && o < 2048 + 80) {
        if ((sp.repeat_mode & 8u) && ((hb >> 40) & 0xFF) < 5) {
            const uint32_t len = 20u + (uint32_t)((hb >> 48) & 0xFF) % 61u;
            if (o - 2048 < len) return 0u;
        }
    }
    return snk_genome_raw(seed, p);
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
    if (!pl.rc) return snk_genome_base(sp, pl.start + i);
    return snk_genome_base(sp, pl.start + (sp.read_len - 1 - i)) ^ 3u;
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
