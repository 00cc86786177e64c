/* snk_oracle.c -- CPU restatement (plain C99) of the reference count + de Bruijn unitig graph path.
 * TEST INFRASTRUCTURE ONLY: see snk_oracle.h.  Each function cites the reference code it restates
 * (paths relative to /root/reference).  Semantics are those of "path B" (lib/assembly, C++); the
 * documented differences to path A (lib/tada, Rust) are in SURVEY.md App. A.9.
 *
 * Pinning: tests/test_oracle_golden.py compares every output of this file with golden vectors
 * dumped from the reference binary itself (tests/golden/, made by tests/golden/make_golden.py via
 * oracle/_ref/snref_driver) and with the reference's known-answer tests
 * (lib/tada/src/cmd_msp.rs:329-350 test_qv_trim_read; lib/tada/src/msp/mod.rs:202-220 test_slice).
 */
#include "snk_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t hi, lo; } kmer_t;

static inline int kmer_lt(kmer_t a, kmer_t b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
static inline int kmer_eq(kmer_t a, kmer_t b) { return a.hi == b.hi && a.lo == b.lo; }

/* base i (0 = leftmost) lives at bits 127-2i..126-2i of (hi,lo): lib/assembly/src/kmers/KMer.h:153-160 */
static inline uint32_t kmer_base(kmer_t k, uint32_t i) {
    return i < 32 ? (uint32_t)(k.hi >> (62 - 2 * i)) & 3u : (uint32_t)(k.lo >> (62 - 2 * (i - 32))) & 3u;
}
static inline kmer_t kmer_set(kmer_t k, uint32_t i, uint32_t b) {
    if (i < 32) { k.hi &= ~(3ull << (62 - 2 * i)); k.hi |= (uint64_t)b << (62 - 2 * i); }
    else { k.lo &= ~(3ull << (62 - 2 * (i - 32))); k.lo |= (uint64_t)b << (62 - 2 * (i - 32)); }
    return k;
}
/* KMer::toSuccessor, kmers/KMer.h:203-216 */
static inline kmer_t kmer_succ(kmer_t k, uint32_t K, uint32_t b) {
    kmer_t r;
    r.hi = (k.hi << 2) | (k.lo >> 62);
    r.lo = k.lo << 2;
    return kmer_set(r, K - 1, b);
}
/* KMer::toPredecessor, kmers/KMer.h:189-201 */
static inline kmer_t kmer_pred(kmer_t k, uint32_t K, uint32_t b) {
    kmer_t r;
    k = kmer_set(k, K - 1, 0); /* drop the last base first so nothing leaks into the padding */
    r.lo = (k.lo >> 2) | (k.hi << 62);
    r.hi = k.hi >> 2;
    return kmer_set(r, 0, b);
}
/* KMer::rc, kmers/KMer.h:218-244 (written base by base here) */
static kmer_t kmer_rc(kmer_t k, uint32_t K) {
    kmer_t r = {0, 0};
    for (uint32_t i = 0; i < K; ++i) r = kmer_set(r, K - 1 - i, kmer_base(k, i) ^ 3u);
    return r;
}
static kmer_t kmer_from(const uint8_t* b, uint32_t K) {
    kmer_t r = {0, 0};
    for (uint32_t i = 0; i < K; ++i) r = kmer_set(r, i, b[i] & 3u);
    return r;
}
/* KMerContext::rc = bit reversal of the byte, kmers/KMerContext.cc:19 */
static inline uint8_t ctx_rc(uint8_t c) {
    c = (uint8_t)(((c >> 4) & 0x0F) | ((c & 0x0F) << 4));
    c = (uint8_t)(((c >> 2) & 0x33) | ((c & 0x33) << 2));
    c = (uint8_t)(((c >> 1) & 0x55) | ((c & 0x55) << 1));
    return c;
}
static const uint8_t SIDE_COUNT[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4}; /* KMerContext.cc gSideCounts */
static const uint8_t BITS2VAL[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   /* KMerContext.cc gBits2Val */
#define CTX_PRED(c) ((uint8_t)((c) >> 4))
#define CTX_SUCC(c) ((uint8_t)((c)&0x0F))

/* ------------------------------------------------------------------------------------------------
 * a1  quality trim.  GoodLenTailFinder, paths/long/BuildReadQGraph48.cc:72-82
 *                  == find_trim_len, lib/tada/src/cmd_msp.rs:129-146 */
uint32_t sno_good_len(const uint8_t* q, uint32_t len, uint32_t K, uint32_t min_qual) {
    uint32_t good = 0;
    for (uint32_t i = len; i-- > 0;) {
        if (q[i] < min_qual) good = 0;
        else if (++good == K) return i + K;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a3  minimiser substring partition as tada does it.  simple_scan, lib/tada/src/msp/mod.rs:60-134
 *     (compute_pvals :17-46).  seq = base codes.  Returns number of slices (or -1 if cap too small). */
int sno_msp_scan(uint32_t k, uint32_t p, const uint8_t* seq, uint32_t len, const uint32_t* perm, sno_slice* out,
                 int cap) {
    if (len < k || p > 16 || p > k) return -1;
    uint32_t np = len - p + 1;
    uint32_t* pv = (uint32_t*)malloc(sizeof(uint32_t) * np);
    uint32_t mask = p == 16 ? 0xFFFFFFFFu : ((1u << (2 * p)) - 1);
    for (uint32_t i = 0; i < np; ++i) {
        uint32_t f = 0, r = 0;
        for (uint32_t j = 0; j < p; ++j) {
            f = (f << 2) | (seq[i + j] & 3u);
            r = (r << 2) | ((3u - seq[i + p - 1 - j]) & 3u); /* p-mer of the reverse complement at the same place */
        }
        f &= mask; r &= mask;
        uint32_t a = perm ? perm[f] : f, b = perm ? perm[r] : r;
        pv[i] = a < b ? a : b;
    }
    int ns = 0;
    uint32_t m = len;
    uint32_t nk = m - k + 1;
    uint32_t* starts = (uint32_t*)malloc(sizeof(uint32_t) * (nk + 1));
    uint32_t* mins = (uint32_t*)malloc(sizeof(uint32_t) * (nk + 1));
    uint32_t nm = 0;
    /* find_min(start, stop): leftmost strict minimum over [start, stop] */
    uint32_t min_pos = 0;
    for (uint32_t pos = 1; pos <= k - p; ++pos) if (pv[pos] < pv[min_pos]) min_pos = pos;
    starts[nm] = 0; mins[nm++] = min_pos;
    for (uint32_t i = 0; i < nk; ++i) {
        if (i > min_pos) {
            min_pos = i;
            for (uint32_t pos = i + 1; pos <= i + k - p; ++pos) if (pv[pos] < pv[min_pos]) min_pos = pos;
            starts[nm] = i; mins[nm++] = min_pos;
        } else {
            uint32_t j = i + k - p;
            uint32_t test = pv[min_pos] <= pv[j] ? min_pos : j; /* pmin keeps the old one on ties */
            if (test != min_pos) { min_pos = test; starts[nm] = i; mins[nm++] = min_pos; }
        }
    }
    for (uint32_t s = 0; s < nm; ++s) {
        if (ns >= cap) { ns = -1; break; }
        out[ns].value = pv[mins[s]];
        out[ns].min_pos = mins[s];
        out[ns].start = starts[s];
        out[ns].len = (s + 1 < nm) ? starts[s + 1] + k - 1 - starts[s] : m - starts[s];
        ++ns;
    }
    free(pv); free(starts); free(mins);
    return ns;
}

/* ------------------------------------------------------------------------------------------------
 * a6-a9  k-merise, canonicalise, sort, reduce, filter.
 *   map    : Kmerizer::map      BuildReadQGraph48.cc:155-172  (initial/final contexts :163,170; len<K+1 skipped :160)
 *   sort   : MapReduceEngine    MapReduceEngine.h:574-584 (std::sort + run detection)
 *   reduce : summarizeEntries   BuildReadQGraph48.cc:92-105 ; areIgnoredBarcodes :108-114 ; areEnoughBarcodes :117-137
 *            Kmerizer::reduce   :174-181
 */
typedef struct { kmer_t k; int32_t bc; uint8_t ctx; } inst_t;

static void radix_sort_inst(inst_t* a, inst_t* tmp, uint64_t n) {
    /* LSD radix, 16-bit digits over (hi,lo); passes whose digit is constant are skipped */
    uint64_t* cnt = (uint64_t*)malloc(sizeof(uint64_t) * 65536);
    inst_t *src = a, *dst = tmp;
    for (int pass = 0; pass < 8; ++pass) {
        int shift = (pass & 3) * 16;
        int use_lo = pass < 4;
        memset(cnt, 0, sizeof(uint64_t) * 65536);
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t w = use_lo ? src[i].k.lo : src[i].k.hi;
            cnt[(w >> shift) & 0xFFFF]++;
        }
        int trivial = 0;
        for (int d = 0; d < 65536; ++d) if (cnt[d] == n) { trivial = 1; break; }
        if (trivial) continue;
        uint64_t sum = 0;
        for (int d = 0; d < 65536; ++d) { uint64_t c = cnt[d]; cnt[d] = sum; sum += c; }
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t w = use_lo ? src[i].k.lo : src[i].k.hi;
            dst[cnt[(w >> shift) & 0xFFFF]++] = src[i];
        }
        inst_t* t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(inst_t) * n);
    free(cnt);
}

static void key_words(kmer_t k, uint32_t* w) {
    w[0] = (uint32_t)(k.hi >> 32); w[1] = (uint32_t)k.hi; w[2] = (uint32_t)(k.lo >> 32); w[3] = (uint32_t)k.lo;
}
static kmer_t words_key(const uint32_t* w) {
    kmer_t k; k.hi = ((uint64_t)w[0] << 32) | w[1]; k.lo = ((uint64_t)w[2] << 32) | w[3]; return k;
}

static int64_t table_find(const sno_table* t, kmer_t k) {
    uint64_t lo = 0, hi = t->n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        kmer_t m = words_key(t->key + 4 * mid);
        if (kmer_lt(m, k)) lo = mid + 1; else hi = mid;
    }
    if (lo < t->n && kmer_eq(words_key(t->key + 4 * lo), k)) return (int64_t)lo;
    return -1;
}
/* KmerDict::findEntry canonicalises first, kmers/ReadPather.h:241-245 */
static int64_t table_find_any(const sno_table* t, kmer_t k, uint32_t K, int* was_rev) {
    kmer_t r = kmer_rc(k, K);
    int rev = kmer_lt(r, k); /* CanonicalForm::REV  <=> rc < fwd, dna/CanonicalForm.h:58-67 */
    if (was_rev) *was_rev = rev;
    return table_find(t, rev ? r : k);
}

int sno_count(const uint8_t* bases, uint32_t stride, const uint32_t* good_len, const int32_t* bcp, uint64_t n_reads,
              uint32_t K, uint32_t min_freq, uint32_t min_bc, int64_t ign_bc_below, sno_table* out,
              uint64_t* n_instances) {
    if (K < 2 || K > 64 || (K & 1)) return -1;
    memset(out, 0, sizeof *out);
    uint64_t ninst = 0;
    for (uint64_t r = 0; r < n_reads; ++r) if (good_len[r] >= K + 1) ninst += good_len[r] - K + 1;
    if (n_instances) *n_instances = ninst;
    inst_t* v = (inst_t*)malloc(sizeof(inst_t) * (ninst ? ninst : 1));
    inst_t* tmp = (inst_t*)malloc(sizeof(inst_t) * (ninst ? ninst : 1));
    if (!v || !tmp) { free(v); free(tmp); return -2; }
    uint64_t w = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        uint32_t len = good_len[r];
        if (len < K + 1) continue; /* :160 */
        int32_t bc = -1;
        if ((int64_t)r >= ign_bc_below && bcp) bc = bcp[r]; /* :158-159 */
        const uint8_t* b = bases + r * (uint64_t)stride;
        kmer_t f = kmer_from(b, K);
        kmer_t rc = kmer_rc(f, K);
        for (uint32_t i = 0; i + K <= len; ++i) {
            if (i) { f = kmer_succ(f, K, b[i + K - 1] & 3u); rc = kmer_pred(rc, K, (b[i + K - 1] & 3u) ^ 3u); }
            uint8_t c = 0;
            if (i > 0) c |= (uint8_t)(1u << (b[i - 1] & 3u)) << 4;       /* predecessor one-hot, high nibble */
            if (i + K < len) c |= (uint8_t)(1u << (b[i + K] & 3u));       /* successor one-hot, low nibble   */
            inst_t e;
            if (kmer_lt(rc, f)) { e.k = rc; e.ctx = ctx_rc(c); } else { e.k = f; e.ctx = c; } /* isRev() -> rc :164 */
            e.bc = bc;
            v[w++] = e;
        }
    }
    radix_sort_inst(v, tmp, ninst);
    free(tmp);
    /* count the groups that survive, then fill */
    uint64_t cap = 1024, nk = 0;
    uint32_t* key = (uint32_t*)malloc(sizeof(uint32_t) * 4 * cap);
    uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * cap);
    uint8_t* ctx = (uint8_t*)malloc(cap);
    int32_t* seen = (int32_t*)malloc(sizeof(int32_t) * (min_bc ? min_bc : 1));
    for (uint64_t i = 0; i < ninst;) {
        uint64_t j = i;
        uint8_t c = 0;
        int ignored = 0;
        uint32_t nseen = 0;
        while (j < ninst && kmer_eq(v[j].k, v[i].k)) {
            c |= v[j].ctx;
            int32_t bc = v[j].bc;
            if (bc == -1) ignored = 1;
            else if (bc > 0 && nseen < min_bc) { /* "don't count unset (-1) or BC==0" :128 */
                uint32_t s = 0;
                while (s < nseen && seen[s] != bc) ++s;
                if (s == nseen) seen[nseen++] = bc;
            }
            ++j;
        }
        uint64_t count = j - i;
        int bc_test = 1;
        if (bcp) bc_test = ignored || nseen >= min_bc; /* :176-178 */
        if (count >= min_freq && bc_test) {
            if (nk == cap) {
                cap *= 2;
                key = (uint32_t*)realloc(key, sizeof(uint32_t) * 4 * cap);
                cnt = (uint32_t*)realloc(cnt, sizeof(uint32_t) * cap);
                ctx = (uint8_t*)realloc(ctx, cap);
            }
            key_words(v[i].k, key + 4 * nk);
            cnt[nk] = count > 0xFFFFFFull ? 0xFFFFFFu : (uint32_t)count; /* KDef::setCount saturates at MAX_OFFSET = 2^24-1, kmers/ReadPather.h:127-131,145 */
            ctx[nk] = c;
            ++nk;
        }
        i = j;
    }
    free(seen);
    free(v);
    out->n = nk; out->key = key; out->count = cnt; out->ctx_raw = ctx;
    out->ctx = (uint8_t*)malloc(nk ? nk : 1);
    /* a11 adjacency prune: KmerDict::recomputeAdjacencies / AdjProc, kmers/ReadPather.h:346-385
     * The reference runs it only when minFreq > 1 (BuildReadQGraph48.cc:320-321); with minFreq <= 1
     * bits pointing at k-mers dropped by the barcode rule stay set -- mirrored here. */
    for (uint64_t i = 0; i < nk; ++i) {
        uint8_t c = ctx[i];
        if (min_freq > 1) {
            kmer_t k = words_key(key + 4 * i);
            for (uint32_t b = 0; b < 4; ++b) {
                if (c & (1u << b)) { if (table_find_any(out, kmer_succ(k, K, b), K, NULL) < 0) c &= (uint8_t)~(1u << b); }
                if (c & (0x10u << b)) { if (table_find_any(out, kmer_pred(k, K, b), K, NULL) < 0) c &= (uint8_t)~(0x10u << b); }
            }
        }
        out->ctx[i] = c;
    }
    return 0;
}

void sno_table_free(sno_table* t) { free(t->key); free(t->count); free(t->ctx_raw); free(t->ctx); memset(t, 0, sizeof *t); }

/* ------------------------------------------------------------------------------------------------
 * a12  unitig pull.  EdgeBuilder, BuildReadQGraph48.cc:327-512; buildEdges :514-541.
 */
typedef struct { uint8_t* b; uint64_t n, cap; } seq_t;
static void seq_push(seq_t* s, uint8_t x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 256; s->b = (uint8_t*)realloc(s->b, s->cap); }
    s->b[s->n++] = x;
}
typedef struct { int64_t* v; uint64_t n, cap; } idx_t;
static void idx_push(idx_t* s, int64_t x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 256; s->v = (int64_t*)realloc(s->v, sizeof(int64_t) * s->cap); }
    s->v[s->n++] = x;
}
/* getCanonicalForm of a base sequence, dna/CanonicalForm.h:35-48: 0 FWD, 1 REV, 2 PALINDROME */
static int seq_form(const uint8_t* b, uint64_t len) {
    if (len & 1) return (b[len / 2] & 2) ? 1 : 0;
    for (uint64_t i = 0, j = len; i < j;) {
        uint8_t f = b[i], r = (uint8_t)(b[--j] ^ 3);
        if (f < r) return 0;
        if (r < f) return 1;
        ++i;
    }
    return 2;
}
static void seq_revcomp(uint8_t* b, uint64_t len) {
    for (uint64_t i = 0, j = len - 1; i < j; ++i, --j) { uint8_t t = b[i] ^ 3; b[i] = b[j] ^ 3; b[j] = t; }
    if (len & 1) b[len / 2] ^= 3;
}
static int kmer_is_pal(kmer_t k, uint32_t K) { return kmer_eq(k, kmer_rc(k, K)); } /* even K only */

typedef struct {
    const sno_table* t; uint32_t K;
    int64_t* edge_of;          /* per retained k-mer: unitig id or -1 (KDef::isNull) */
    seq_t seq; idx_t ents;     /* mEdgeSeq / mEdgeEntries */
    /* output edges */
    uint8_t* bases; uint64_t nb, capb; uint64_t* off; uint64_t ne, cape;
    int err;
} eb_t;

/* EdgeBuilder::lookup :466-476 -- context returned in the orientation of `k` */
static int64_t eb_lookup(eb_t* eb, kmer_t k, uint8_t* ctx) {
    int rev;
    int64_t i = table_find_any(eb->t, k, eb->K, &rev);
    if (i < 0) { eb->err = 1; *ctx = 0; return -1; } /* ForceAssert(result) */
    *ctx = rev ? ctx_rc(eb->t->ctx[i]) : eb->t->ctx[i];
    return i;
}
/* EdgeBuilder::addEdge :478-506 */
static void eb_add_edge(eb_t* eb) {
    if (seq_form(eb->seq.b, eb->seq.n) == 1) {
        seq_revcomp(eb->seq.b, eb->seq.n);
        for (uint64_t i = 0, j = eb->ents.n - 1; i < j; ++i, --j) { int64_t t = eb->ents.v[i]; eb->ents.v[i] = eb->ents.v[j]; eb->ents.v[j] = t; }
    }
    if (eb->ne + 1 >= eb->cape) { eb->cape = eb->cape ? eb->cape * 2 : 1024; eb->off = (uint64_t*)realloc(eb->off, sizeof(uint64_t) * (eb->cape + 1)); }
    while (eb->nb + eb->seq.n > eb->capb) { eb->capb = eb->capb ? eb->capb * 2 : 65536; eb->bases = (uint8_t*)realloc(eb->bases, eb->capb); }
    memcpy(eb->bases + eb->nb, eb->seq.b, eb->seq.n);
    eb->off[eb->ne] = eb->nb;
    eb->nb += eb->seq.n;
    for (uint64_t i = 0; i < eb->ents.n; ++i) {
        if (eb->edge_of[eb->ents.v[i]] != -1) eb->err = 2; /* "Having trouble with preoccupied kmers." */
        eb->edge_of[eb->ents.v[i]] = (int64_t)eb->ne;
    }
    eb->ne++;
    eb->off[eb->ne] = eb->nb;
    eb->seq.n = 0; eb->ents.n = 0;
}
/* EdgeBuilder::extend :445-464 */
static void eb_extend(eb_t* eb, kmer_t k, uint8_t ctx) {
    kmer_t next = k;
    while (SIDE_COUNT[CTX_SUCC(ctx)] == 1) {
        uint8_t s = BITS2VAL[CTX_SUCC(ctx)];
        next = kmer_succ(next, eb->K, s);
        if (kmer_is_pal(next, eb->K)) break;
        int64_t e = eb_lookup(eb, next, &ctx);
        if (e < 0) break;
        if (SIDE_COUNT[CTX_PRED(ctx)] != 1) break;
        seq_push(&eb->seq, s);
        idx_push(&eb->ents, e);
    }
    int form = seq_form(eb->seq.b, eb->seq.n);
    if (form == 2 && eb->seq.n != eb->K) eb->err = 3; /* ForceAssertEq(mEdgeSeq.size(),K) */
    if (form == 1) { eb->seq.n = 0; eb->ents.n = 0; }
    else eb_add_edge(eb);
}
static void eb_seq_assign_kmer(eb_t* eb, kmer_t k) {
    eb->seq.n = 0;
    for (uint32_t i = 0; i < eb->K; ++i) seq_push(&eb->seq, (uint8_t)kmer_base(k, i));
}
/* upstreamExtensionPossible :408-417 / downstreamExtensionPossible :419-428 */
static int eb_up_possible(eb_t* eb, kmer_t k, uint8_t ctx) {
    if (SIDE_COUNT[CTX_PRED(ctx)] != 1) return 0;
    kmer_t p = kmer_pred(k, eb->K, BITS2VAL[CTX_PRED(ctx)]);
    if (kmer_is_pal(p, eb->K)) return 0;
    uint8_t c2;
    if (eb_lookup(eb, p, &c2) < 0) return 0;
    return SIDE_COUNT[CTX_SUCC(c2)] == 1;
}
static int eb_down_possible(eb_t* eb, kmer_t k, uint8_t ctx) {
    if (SIDE_COUNT[CTX_SUCC(ctx)] != 1) return 0;
    kmer_t s = kmer_succ(k, eb->K, BITS2VAL[CTX_SUCC(ctx)]);
    if (kmer_is_pal(s, eb->K)) return 0;
    uint8_t c2;
    if (eb_lookup(eb, s, &c2) < 0) return 0;
    return SIDE_COUNT[CTX_PRED(c2)] == 1;
}
/* EdgeBuilder::buildEdge :335-345 */
static void eb_build_edge(eb_t* eb, int64_t i) {
    kmer_t k = words_key(eb->t->key + 4 * i);
    uint8_t ctx = eb->t->ctx[i];
    if (kmer_is_pal(k, eb->K)) { eb_seq_assign_kmer(eb, k); idx_push(&eb->ents, i); eb_add_edge(eb); return; }
    int up = eb_up_possible(eb, k, ctx);
    if (up) {
        if (eb_down_possible(eb, k, ctx)) return;
        /* extendUpstream :435-438 */
        kmer_t r = kmer_rc(k, eb->K);
        eb_seq_assign_kmer(eb, r);
        idx_push(&eb->ents, i);
        eb_extend(eb, r, ctx_rc(ctx));
    } else if (eb_down_possible(eb, k, ctx)) {
        eb_seq_assign_kmer(eb, k);
        idx_push(&eb->ents, i);
        eb_extend(eb, k, ctx);
    } else {
        eb_seq_assign_kmer(eb, k); idx_push(&eb->ents, i); eb_add_edge(eb);
    }
}
/* EdgeBuilder::simpleCircle :348-372 + canonicalizeCircle :375-397 */
static void eb_simple_circle(eb_t* eb, int64_t first) {
    uint32_t K = eb->K;
    kmer_t k = words_key(eb->t->key + 4 * first);
    uint8_t ctx = eb->t->ctx[first];
    eb_seq_assign_kmer(eb, k);
    idx_push(&eb->ents, first);
    for (;;) {
        if (SIDE_COUNT[CTX_PRED(ctx)] != 1 || SIDE_COUNT[CTX_SUCC(ctx)] != 1) { eb->err = 4; return; }
        uint8_t s = BITS2VAL[CTX_SUCC(ctx)];
        k = kmer_succ(k, K, s);
        int64_t e = eb_lookup(eb, k, &ctx);
        if (e == first) break;
        if (e < 0 || eb->edge_of[e] != -1) { eb->err = 5; return; } /* "Failed to close circle." */
        seq_push(&eb->seq, s);
        idx_push(&eb->ents, e);
    }
    /* canonicalizeCircle: rotate so that the minimum canonical k-mer comes first, in FWD form */
    uint64_t idx = 0;
    for (uint64_t i = 1; i < eb->ents.n; ++i) if (eb->ents.v[i] < eb->ents.v[idx]) idx = i; /* table is sorted: min index == min k-mer */
    if (seq_form(eb->seq.b + idx, K) == 1) {
        seq_revcomp(eb->seq.b, eb->seq.n);
        for (uint64_t i = 0, j = eb->ents.n - 1; i < j; ++i, --j) { int64_t t = eb->ents.v[i]; eb->ents.v[i] = eb->ents.v[j]; eb->ents.v[j] = t; }
        idx = eb->seq.n - idx - K;
    }
    if (idx) {
        uint64_t n = eb->seq.n;
        uint8_t* bv = (uint8_t*)malloc(n);
        uint64_t w = 0;
        for (uint64_t i = idx; i < n; ++i) bv[w++] = eb->seq.b[i];
        for (uint64_t i = K - 1; i < K + idx - 1; ++i) bv[w++] = eb->seq.b[i];
        memcpy(eb->seq.b, bv, n);
        free(bv);
        uint64_t m = eb->ents.n;
        int64_t* ev = (int64_t*)malloc(sizeof(int64_t) * m);
        for (uint64_t i = 0; i < m; ++i) ev[i] = eb->ents.v[(i + idx) % m];
        memcpy(eb->ents.v, ev, sizeof(int64_t) * m);
        free(ev);
    }
    eb_add_edge(eb);
}

/* BVComp, paths/long/HBVFromEdges.cc:106-111: length descending, then lexicographic */
typedef struct { const uint8_t* b; uint64_t len; } sref_t;
static int sref_cmp(const void* pa, const void* pb) {
    const sref_t* a = (const sref_t*)pa; const sref_t* b = (const sref_t*)pb;
    if (a->len != b->len) return a->len > b->len ? -1 : 1;
    int c = memcmp(a->b, b->b, a->len);
    return c;
}

int sno_unitigs_build(const sno_table* t, uint32_t K, sno_unitigs* out) {
    memset(out, 0, sizeof *out);
    eb_t eb;
    memset(&eb, 0, sizeof eb);
    eb.t = t; eb.K = K;
    eb.edge_of = (int64_t*)malloc(sizeof(int64_t) * (t->n ? t->n : 1));
    for (uint64_t i = 0; i < t->n; ++i) eb.edge_of[i] = -1;
    for (uint64_t i = 0; i < t->n; ++i) if (eb.edge_of[i] == -1) eb_build_edge(&eb, (int64_t)i);   /* :519-523 */
    for (uint64_t i = 0; i < t->n; ++i) if (eb.edge_of[i] == -1) eb_simple_circle(&eb, (int64_t)i); /* :534-537 */
    int err = eb.err;
    /* deterministic order */
    sref_t* refs = (sref_t*)malloc(sizeof(sref_t) * (eb.ne ? eb.ne : 1));
    for (uint64_t e = 0; e < eb.ne; ++e) { refs[e].b = eb.bases + eb.off[e]; refs[e].len = eb.off[e + 1] - eb.off[e]; }
    qsort(refs, eb.ne, sizeof(sref_t), sref_cmp);
    out->n = eb.ne;
    out->off = (uint64_t*)malloc(sizeof(uint64_t) * (eb.ne + 1));
    out->bases = (uint8_t*)malloc(eb.nb ? eb.nb : 1);
    uint64_t w = 0;
    for (uint64_t e = 0; e < eb.ne; ++e) { out->off[e] = w; memcpy(out->bases + w, refs[e].b, refs[e].len); w += refs[e].len; }
    out->off[eb.ne] = w;
    free(refs); free(eb.bases); free(eb.off); free(eb.edge_of); free(eb.seq.b); free(eb.ents.v);
    return err ? -10 - err : 0;
}
void sno_unitigs_free(sno_unitigs* u) { free(u->off); free(u->bases); memset(u, 0, sizeof *u); }

/* ------------------------------------------------------------------------------------------------
 * a13  .bv hand-off file.  Writer: TempGraph::write_to_sn_format lib/tada/src/debruijn.rs:895-929;
 *      reader: BinaryReader::readFile(vec<basevector>) BuildReadQGraph48.cc:1640-1642.
 *      "BINWRITE", u64 count, per entry u32 length + ceil(len/4) bytes, base j at bits 2*(j%4). */
int sno_write_bv(const char* path, const sno_unitigs* u) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fwrite("BINWRITE", 1, 8, f);
    uint64_t n = u->n;
    fwrite(&n, 8, 1, f);
    for (uint64_t e = 0; e < n; ++e) {
        uint32_t len = (uint32_t)(u->off[e + 1] - u->off[e]);
        fwrite(&len, 4, 1, f);
        const uint8_t* b = u->bases + u->off[e];
        for (uint32_t j = 0; j < len; j += 4) {
            uint8_t v = 0;
            for (uint32_t q = 0; q < 4 && j + q < len; ++q) v |= (uint8_t)((b[j + q] & 3u) << (2 * q));
            fputc(v, f);
        }
    }
    fclose(f);
    return 0;
}
int sno_read_bv(const char* path, sno_unitigs* out) {
    memset(out, 0, sizeof *out);
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    char magic[8];
    uint64_t n;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BINWRITE", 8) || fread(&n, 8, 1, f) != 1) { fclose(f); return -2; }
    out->n = n;
    out->off = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
    uint64_t cap = 1024, w = 0;
    out->bases = (uint8_t*)malloc(cap);
    for (uint64_t e = 0; e < n; ++e) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) { fclose(f); return -3; }
        while (w + len > cap) { cap *= 2; out->bases = (uint8_t*)realloc(out->bases, cap); }
        out->off[e] = w;
        for (uint32_t j = 0; j < len; j += 4) {
            int v = fgetc(f);
            if (v < 0) { fclose(f); return -3; }
            for (uint32_t q = 0; q < 4 && j + q < len; ++q) out->bases[w++] = (uint8_t)((v >> (2 * q)) & 3);
        }
    }
    out->off[n] = w;
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a14  graph from unitigs.  buildHBVFromEdges, paths/long/HBVFromEdges.cc:244-296:
 *   VertexDictBuilder::map/reduce :136-168  (4 edge ends per unitig, 2 if the unitig is a palindrome;
 *       a vertex = one distinct (K-1)-mer; its incident list is sorted by EEComp :113-122)
 *   edge order = BVComp :276-277 (the input here is already in that order)
 *   HBVBuilder::processQueue :198-229   (FIFO flood fill assigning vertex ids and HBV edge ids)
 */
typedef struct { uint32_t edge; uint8_t rc, distal; } eend_t;
typedef struct { const sno_unitigs* u; uint32_t kl; } eectx_t;
static const eectx_t* g_ee;
static inline uint8_t ee_base(const eectx_t* c, eend_t e, uint32_t j) {
    const uint8_t* b = c->u->bases + c->u->off[e.edge];
    uint64_t len = c->u->off[e.edge + 1] - c->u->off[e.edge];
    /* position j of the (K-1)-mer at the proximal (distal=0) or distal end of the edge read fwd or rc */
    uint64_t p = e.distal ? len - c->kl + j : j;
    return e.rc ? (uint8_t)(b[len - 1 - p] ^ 3) : b[p];
}
static int ee_seq_cmp(eend_t a, eend_t b) {
    for (uint32_t j = 0; j < g_ee->kl; ++j) {
        uint8_t x = ee_base(g_ee, a, j), y = ee_base(g_ee, b, j);
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
static int ee_cmp(const void* pa, const void* pb) { /* group by sequence, then EEComp */
    eend_t a = *(const eend_t*)pa, b = *(const eend_t*)pb;
    int c = ee_seq_cmp(a, b);
    if (c) return c;
    if (a.edge != b.edge) return a.edge < b.edge ? -1 : 1; /* BVComp order == index order (input sorted, ties are equal seqs) */
    if (a.rc != b.rc) return a.rc < b.rc ? -1 : 1;
    if (a.distal != b.distal) return a.distal < b.distal ? -1 : 1;
    return 0;
}

int sno_hbv_build(const sno_unitigs* u, uint32_t K, sno_hbv* out) {
    memset(out, 0, sizeof *out);
    uint64_t n = u->n;
    if (n == 0) return 0;
    eectx_t c = {u, K - 1};
    g_ee = &c;
    eend_t* ee = (eend_t*)malloc(sizeof(eend_t) * 4 * n);
    uint64_t ne = 0;
    uint8_t* pal = (uint8_t*)malloc(n);
    for (uint64_t e = 0; e < n; ++e) {
        uint64_t len = u->off[e + 1] - u->off[e];
        pal[e] = seq_form(u->bases + u->off[e], len) == 2;
        eend_t x = {(uint32_t)e, 0, 0}; ee[ne++] = x;
        x.distal = 1; ee[ne++] = x;
        if (!pal[e]) { x.rc = 1; x.distal = 0; ee[ne++] = x; x.distal = 1; ee[ne++] = x; }
    }
    qsort(ee, ne, sizeof(eend_t), ee_cmp);
    /* vertices = runs of equal sequence */
    int32_t* vtx_of = (int32_t*)malloc(sizeof(int32_t) * 4 * n); /* index: edge*4 + rc*2 + distal -> run id */
    uint64_t* run_beg = (uint64_t*)malloc(sizeof(uint64_t) * (ne + 1));
    uint64_t nruns = 0;
    for (uint64_t i = 0; i < ne;) {
        uint64_t j = i + 1;
        while (j < ne && ee_seq_cmp(ee[i], ee[j]) == 0) ++j;
        run_beg[nruns] = i;
        for (uint64_t q = i; q < j; ++q) vtx_of[ee[q].edge * 4 + ee[q].rc * 2 + ee[q].distal] = (int32_t)nruns;
        ++nruns;
        i = j;
    }
    run_beg[nruns] = ne;
    out->n_vertices = (int32_t)nruns;
    int32_t* vid = (int32_t*)malloc(sizeof(int32_t) * nruns);
    for (uint64_t i = 0; i < nruns; ++i) vid[i] = -1;
    out->fwd_xlat = (int32_t*)malloc(sizeof(int32_t) * n);
    out->rev_xlat = (int32_t*)malloc(sizeof(int32_t) * n);
    for (uint64_t i = 0; i < n; ++i) out->fwd_xlat[i] = out->rev_xlat[i] = -1;
    out->v_left = (int32_t*)malloc(sizeof(int32_t) * 2 * n);
    out->v_right = (int32_t*)malloc(sizeof(int32_t) * 2 * n);
    out->src_unitig = (int32_t*)malloc(sizeof(int32_t) * 2 * n);
    out->is_rc = (uint8_t*)malloc(2 * n);
    int32_t next_v = 0, next_e = 0;
    /* FIFO of (edge, rc); bounded by total pushes: each processed edge pushes <= 16 entries */
    uint64_t qcap = 1024, qh = 0, qt = 0;
    uint64_t* q = (uint64_t*)malloc(sizeof(uint64_t) * qcap);
#define Q_PUSH(x) do { if (qt == qcap) { if (qh > 0) { memmove(q, q + qh, sizeof(uint64_t) * (qt - qh)); qt -= qh; qh = 0; } \
                       if (qt == qcap) { qcap *= 2; q = (uint64_t*)realloc(q, sizeof(uint64_t) * qcap); } } q[qt++] = (x); } while (0)
#define IS_DONE(e, r) (((r) ? out->rev_xlat : out->fwd_xlat)[e] != -1)
    for (int pass = 0; pass < 2; ++pass)
        for (uint64_t e0 = 0; e0 < n; ++e0) { /* :285-295: all fwd in edge order, then all rc */
            if (IS_DONE(e0, pass)) continue;
            Q_PUSH(e0 * 2 + pass);
            while (qh < qt) {
                uint64_t x = q[qh++];
                uint64_t e = x >> 1; int rc = (int)(x & 1);
                if (IS_DONE(e, rc)) continue;
                int32_t r1 = vtx_of[e * 4 + rc * 2 + 0], r2 = vtx_of[e * 4 + rc * 2 + 1];
                if (pal[e] && rc) { r1 = vtx_of[e * 4 + 0]; r2 = vtx_of[e * 4 + 1]; } /* palindrome: rc ends == fwd ends */
                if (vid[r1] == -1) vid[r1] = next_v++;
                if (vid[r2] == -1) vid[r2] = next_v++;
                int32_t id = next_e++;
                out->v_left[id] = vid[r1]; out->v_right[id] = vid[r2];
                out->src_unitig[id] = (int32_t)e; out->is_rc[id] = (uint8_t)rc;
                if (!rc || pal[e]) out->fwd_xlat[e] = id;
                if (rc || pal[e]) out->rev_xlat[e] = id;
                for (int side = 0; side < 2; ++side) {
                    int32_t r = side ? r2 : r1;
                    for (uint64_t j = run_beg[r]; j < run_beg[r + 1]; ++j)
                        if (!IS_DONE(ee[j].edge, ee[j].rc)) Q_PUSH((uint64_t)ee[j].edge * 2 + ee[j].rc);
                }
            }
        }
    out->n_edges = next_e;
    free(q); free(vid); free(run_beg); free(vtx_of); free(pal); free(ee);
    return 0;
}
void sno_hbv_free(sno_hbv* h) {
    free(h->v_left); free(h->v_right); free(h->src_unitig); free(h->is_rc); free(h->fwd_xlat); free(h->rev_xlat);
    memset(h, 0, sizeof *h);
}

/* ================================================================================================
 * f1  read pathing: every read onto the unitig graph.
 *   Pather::path                          paths/long/BuildReadQGraph48.cc:705-748   (seed by dictionary, extend by exact match)
 *   HBVPather::algorithmTwo               :1217-1336   (hanging-edge seeds, captured gaps, short last seed, connectivity)
 *   pathPartsToReadPath                   :1393-1428
 *   ExtendReadPath::attemptLeft/RightwardExtension, scoreLeft/RightOverlap   paths/long/ExtendReadPath.cc:15-358
 *   dictionary fill (k-mer -> edge, offset)   :1656-1664 (buildGraphFromMSP) == KDef::set in buildEdges
 * Reads are pathed UNTRIMMED (mReads[readId]).  Edge ids in the result are HBV edge ids (fwd/rev translation of the unitig).
 * ------------------------------------------------------------------------------------------------ */
typedef struct { kmer_t k; uint32_t unitig, offset; int used; } pdict_ent;
typedef struct { pdict_ent* e; uint64_t mask; } pdict;
static uint64_t pd_hash(kmer_t k) { uint64_t x = k.hi * 0x9E3779B97F4A7C15ull ^ (k.lo + 0xD1B54A32D192ED03ull) * 0xC2B2AE3D27D4EB4Full; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; return x ^ (x >> 32); }
static const pdict_ent* pd_find(const pdict* d, kmer_t k, uint32_t K) {       /* KmerDict::findEntry canonicalises, kmers/ReadPather.h:241-245 */
    kmer_t r = kmer_rc(k, K);
    if (kmer_lt(r, k)) k = r;
    for (uint64_t s = pd_hash(k) & d->mask;; s = (s + 1) & d->mask) {
        if (!d->e[s].used) return NULL;
        if (kmer_eq(d->e[s].k, k)) return &d->e[s];
    }
}
typedef struct { int gap; uint32_t unitig; int rc; uint32_t off, len, elen; } ppart;   /* PathPart :622-689 (len of a gap = its k-mers) */
typedef struct {
    uint32_t K;
    const sno_unitigs* u;
    const sno_hbv* h;
    pdict d;
    int32_t *to_off, *to_v, *to_e, *from_off, *from_v, *from_e;   /* To(v)/ToEdgeObj(v), From(v)/FromEdgeObj(v): AddEdge order, graph/DigraphTemplate.h:2572-2582 */
} pctx;
static inline uint32_t ulen(const pctx* c, uint32_t u) { return (uint32_t)(c->u->off[u + 1] - c->u->off[u]); }
static inline uint32_t ubase(const pctx* c, uint32_t u, int rc, uint32_t i) {     /* base i of the unitig in the given orientation */
    const uint8_t* b = c->u->bases + c->u->off[u];
    return rc ? (uint32_t)(b[ulen(c, u) - 1 - i] ^ 3u) : b[i];
}
static inline uint32_t ebase(const pctx* c, int32_t e, uint32_t i) { return ubase(c, (uint32_t)c->h->src_unitig[e], c->h->is_rc[e], i); }
static inline uint32_t elen_bases(const pctx* c, int32_t e) { return ulen(c, (uint32_t)c->h->src_unitig[e]); }
static inline int to_size(const pctx* c, int32_t v) { return c->to_off[v + 1] - c->to_off[v]; }
static inline int from_size(const pctx* c, int32_t v) { return c->from_off[v + 1] - c->from_off[v]; }
static inline int32_t part_edge(const pctx* c, const ppart* p) { return p->rc ? c->h->rev_xlat[p->unitig] : c->h->fwd_xlat[p->unitig]; }

static int pair_cmp(const void* a, const void* b) {
    const int32_t* x = (const int32_t*)a; const int32_t* y = (const int32_t*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1]);
}
static void build_adj(int32_t N, int32_t E, const int32_t* key_v, const int32_t* other_v, int32_t** off_o, int32_t** v_o, int32_t** e_o) {
    int32_t* off = (int32_t*)calloc((size_t)N + 2, 4);
    for (int32_t e = 0; e < E; ++e) off[key_v[e] + 1]++;
    for (int32_t v = 0; v < N; ++v) off[v + 1] += off[v];
    int32_t* tmp = (int32_t*)malloc(((size_t)E + 1) * 8);
    int32_t* cur = (int32_t*)malloc(((size_t)N + 1) * 4);
    memcpy(cur, off, ((size_t)N + 1) * 4);
    for (int32_t e = 0; e < E; ++e) { int32_t p = cur[key_v[e]]++; tmp[2 * p] = other_v[e]; tmp[2 * p + 1] = e; }
    for (int32_t v = 0; v < N; ++v) qsort(tmp + 2 * off[v], (size_t)(off[v + 1] - off[v]), 8, pair_cmp);    /* (other vertex, edge id) ascending */
    int32_t* vv = (int32_t*)malloc(((size_t)E + 1) * 4);
    int32_t* ee = (int32_t*)malloc(((size_t)E + 1) * 4);
    for (int32_t i = 0; i < E; ++i) { vv[i] = tmp[2 * i]; ee[i] = tmp[2 * i + 1]; }
    free(tmp); free(cur);
    *off_o = off; *v_o = vv; *e_o = ee;
}

/* Pather::path :705-748 */
static int path_parts(const pctx* c, const uint8_t* read, uint32_t n, ppart* parts, int cap) {
    const uint32_t K = c->K;
    int np = 0;
    if (n < K) { parts[np].gap = 1; parts[np].len = n; parts[np].elen = 0; parts[np].off = 0; parts[np].rc = 0; parts[np].unitig = 0; return 1; }
    uint32_t i = 0;
    const uint32_t end = n - K + 1;
    while (i != end) {
        kmer_t km = kmer_from(read + i, K);
        const pdict_ent* ent = pd_find(&c->d, km, K);
        if (!ent) {
            uint32_t gap_len = 1, i2 = i + K;
            ++i;
            while (i2 != n) {
                km = kmer_succ(km, K, read[i2] & 3u);
                ++i2;
                if ((ent = pd_find(&c->d, km, K))) break;
                ++gap_len; ++i;
            }
            if (np >= cap) return -1;
            ppart g = {1, 0, 0, 0, gap_len, 0};
            parts[np++] = g;
        }
        if (ent) {
            const uint32_t u = ent->unitig, sz = ulen(c, u);
            uint32_t off = ent->offset, len = 1;
            int rc = 0;
            for (uint32_t q = 0; q < K; ++q) if ((read[i + q] & 3u) != ubase(c, u, 0, off + q)) { rc = 1; break; }      /* CF<K>::isRC, dna/CanonicalForm.h:85-91 */
            if (!rc) {
                uint32_t a = i + K, b = off + K;
                while (a < n && b < sz && (read[a] & 3u) == ubase(c, u, 0, b)) { ++len; ++a; ++b; }
            } else {
                off = sz - off;                                   /* :726-729 */
                uint32_t a = i + K, b = off;
                while (a < n && b < sz && (read[a] & 3u) == ubase(c, u, 1, b)) { ++len; ++a; ++b; }
                off -= K;
            }
            if (np >= cap) return -1;
            ppart p = {0, u, rc, off, len, sz - K + 1};
            parts[np++] = p;
            i += len;
        }
    }
    return np;
}
static inline int same_edge(const ppart* a, const ppart* b) { return a->unitig == b->unitig && a->rc == b->rc; }     /* :656-657; gaps carry unitig 0, rc 0 */
/* PathPart::isConformingCapturedGap :659-666 (unsigned arithmetic as written) */
static int conforming_gap(const ppart* p, uint32_t max_jitter) {
    const ppart *prev = p - 1, *next = p + 1;
    uint32_t graph_dist = next->off - (prev->off + prev->len);
    if (!same_edge(prev, next)) graph_dist += prev->elen;
    int32_t d = (int32_t)(p->len - graph_dist);
    return (uint32_t)(d < 0 ? -d : d) <= max_jitter;
}
/* Pather::isJoinable :808-814: the LAST K-1 bases of both edges, each in its part's orientation (as written in the reference) */
static int joinable(const pctx* c, const ppart* a, const ppart* b) {
    if (a->unitig == b->unitig) return 1;
    const uint32_t K = c->K, la = ulen(c, a->unitig), lb = ulen(c, b->unitig);
    for (uint32_t q = 0; q + 1 < K; ++q)
        if (ubase(c, a->unitig, a->rc, la - (K - 1) + q) != ubase(c, b->unitig, b->rc, lb - (K - 1) + q)) return 0;
    return 1;
}
/* scoreLeftOverlap / scoreRightOverlap, ExtendReadPath.cc:15-106: pDecay 0.2, Q2 counted as Q20, 10 per read base left over */
static uint32_t score_overlap(const pctx* c, const uint8_t* bases, const uint8_t* quals, uint32_t n, uint32_t start, int32_t e, int left) {
    const uint32_t K = c->K, esz = elen_bases(c, e);
    uint32_t qsum = 0, penalty = 0, steps = 0;
    /* right: read positions n-start.., edge positions K-1..;  left: read positions start-1 downwards, edge positions esz-K downwards */
    for (;; ++steps) {
        if (steps >= start) break;
        uint32_t rp, ep;
        if (!left) { rp = n - start + steps; ep = K - 1 + steps; if (ep >= esz) break; }
        else { rp = start - 1 - steps; if (steps + K > esz) break; ep = esz - K - steps; }
        if ((bases[rp] & 3u) != ebase(c, e, ep)) {
            const uint32_t q = quals[rp] == 2 ? 20u : quals[rp];
            penalty += q;
            qsum += penalty;
        } else if (penalty > 0) penalty = (uint32_t)((double)penalty - 0.2 * (double)penalty);      /* penalty -= (pDecay*penalty) */
    }
    qsum += 10u * (start - steps);
    return qsum;
}
/* attemptLeftwardExtension :123-239 / attemptRightwardExtension :242-358 */
static int extend_once(const pctx* c, int32_t* path, int* np, int cap, int32_t* offset, const uint8_t* bases, const uint8_t* quals, uint32_t n, int left) {
    const uint32_t K = c->K;
    if (!*np) return 0;
    uint64_t last_gap;
    if (left) {
        if (*offset >= 0) return 0;
        last_gap = (uint64_t)(-(int64_t)*offset);
    } else {
        int32_t g = (int32_t)n + *offset;
        for (int i = 0; i < *np; ++i) g -= (int32_t)(elen_bases(c, path[i]) - K + 1);
        g -= (int32_t)(K - 1);
        if (g < 10) return 0;
        last_gap = (uint64_t)g;
    }
    if (last_gap < 10) return 0;
    const int32_t v = left ? c->h->v_left[path[0]] : c->h->v_right[path[*np - 1]];
    const int32_t* off = left ? c->to_off : c->from_off;
    const int32_t* ee = (left ? c->to_e : c->from_e) + off[v];
    const int32_t* vd = (left ? c->to_v : c->from_v) + off[v];
    const int ne = off[v + 1] - off[v];
    int nlong = 0, nshort = 0, short_same = 1;
    int32_t short_dest = -1;
    for (int i = 0; i < ne; ++i) {
        const int hanging = left ? (to_size(c, vd[i]) == 0 && from_size(c, vd[i]) == 1) : (from_size(c, vd[i]) == 0 && to_size(c, vd[i]) == 1);
        const int lng = (uint64_t)elen_bases(c, ee[i]) - (K - 1) >= last_gap;
        nlong += lng;
        if (!lng && !hanging) { if (nshort && vd[i] != short_dest) short_same = 0; short_dest = vd[i]; ++nshort; }
    }
    if (ne != 1 && nshort > 0) {
        if (nlong > 0) return 0;
        if (!short_same) return 0;
        if ((left ? to_size(c, short_dest) : from_size(c, short_dest)) != 1) return 0;
    }
    int32_t least_edge = -1;
    uint32_t least = 0xFFFFFFFFu;
    for (int i = 0; i < ne; ++i) {
        const int hanging = left ? (to_size(c, vd[i]) == 0 && from_size(c, vd[i]) == 1) : (from_size(c, vd[i]) == 0 && to_size(c, vd[i]) == 1);
        if (!hanging || ne == 1) {
            const uint32_t sc = score_overlap(c, bases, quals, n, (uint32_t)last_gap, ee[i], left);
            if (sc < least) { least_edge = ee[i]; least = sc; }
        }
    }
    if (least_edge == -1 || (uint64_t)least > last_gap * 10) return 0;
    if (*np >= cap) return 0;
    if (left) {
        memmove(path + 1, path, (size_t)*np * 4);
        path[0] = least_edge;
        *offset += (int32_t)(elen_bases(c, least_edge) - K + 1);
    } else path[*np] = least_edge;
    ++*np;
    return 1;
}

#define SNO_PATH_CAP 512
int sno_path_reads(const uint8_t* bases, const uint8_t* quals, uint32_t stride, const uint32_t* lens, uint64_t n_reads, uint32_t K,
                   const sno_unitigs* u, const sno_hbv* h, int32_t* out_off, int32_t* out_n, int32_t** out_edges, uint64_t* out_total) {
    pctx c;
    memset(&c, 0, sizeof c);
    c.K = K; c.u = u; c.h = h;
    uint64_t nk = 0;
    for (uint64_t i = 0; i < u->n; ++i) nk += u->off[i + 1] - u->off[i] - (K - 1);
    uint64_t cap = 64;
    while (cap < 2 * nk + 2) cap <<= 1;
    c.d.e = (pdict_ent*)calloc(cap, sizeof(pdict_ent));
    c.d.mask = cap - 1;
    if (!c.d.e) return -2;
    for (uint64_t i = 0; i < u->n; ++i) {                /* :1656-1664: every k-mer of every edge -> (edge, offset) */
        const uint8_t* b = u->bases + u->off[i];
        const uint32_t L = (uint32_t)(u->off[i + 1] - u->off[i]);
        kmer_t km = kmer_from(b, K);
        for (uint32_t o = 0; o + K <= L; ++o) {
            if (o) km = kmer_succ(km, K, b[o + K - 1] & 3u);
            kmer_t r = kmer_rc(km, K), ck = kmer_lt(r, km) ? r : km;
            uint64_t s = pd_hash(ck) & c.d.mask;
            while (c.d.e[s].used) s = (s + 1) & c.d.mask;
            c.d.e[s].k = ck; c.d.e[s].unitig = (uint32_t)i; c.d.e[s].offset = o; c.d.e[s].used = 1;
        }
    }
    build_adj(h->n_vertices, h->n_edges, h->v_right, h->v_left, &c.to_off, &c.to_v, &c.to_e);       /* in-edges of w keyed by source vertex */
    build_adj(h->n_vertices, h->n_edges, h->v_left, h->v_right, &c.from_off, &c.from_v, &c.from_e);
    uint64_t ecap = n_reads * 2 + 16, tot = 0;
    int32_t* edges = (int32_t*)malloc(ecap * 4);
    ppart* parts = (ppart*)malloc(sizeof(ppart) * SNO_PATH_CAP);
    ppart* np_ = (ppart*)malloc(sizeof(ppart) * SNO_PATH_CAP);
    int32_t path[SNO_PATH_CAP];
    int rc = 0;
    for (uint64_t r = 0; r < n_reads && !rc; ++r) {
        const uint8_t* rb = bases + r * (uint64_t)stride;
        const uint8_t* rq = quals + r * (uint64_t)stride;
        const uint32_t n = lens[r];
        int m = path_parts(&c, rb, n, parts, SNO_PATH_CAP);
        if (m < 0) { rc = -3; break; }
        /* seeds on hanging edges become gaps, adjacent gaps merge (:1235-1262) */
        int m2 = 0;
        for (int i = 0; i < m; ++i) {
            ppart p = parts[i];
            if (!p.gap) {
                const int32_t e = part_edge(&c, &p), vl = h->v_left[e], vr = h->v_right[e];
                if (to_size(&c, vl) == 0 && to_size(&c, vr) > 1 && from_size(&c, vr) > 0 && p.elen <= 100) { ppart g = {1, 0, 0, 0, p.len, 0}; p = g; }
            }
            if (p.gap && m2 && np_[m2 - 1].gap) np_[m2 - 1].len += p.len;
            else np_[m2++] = p;
        }
        memcpy(parts, np_, sizeof(ppart) * (size_t)m2);
        m = m2;
        /* a captured gap that does not fit the graph (:1268-1292) */
        if (m >= 3) {
            uint32_t seeds = parts[0].gap ? 0u : 1u;
            for (int p = 1; p < m - 1; ++p) {
                if (!parts[p].gap) { ++seeds; continue; }
                if (!conforming_gap(&parts[p], 3) || !joinable(&c, &parts[p - 1], &parts[p + 1])) {
                    if (seeds > 1) {
                        ppart t = {1, 0, 0, 0, parts[p - 1].len, 0};
                        for (int q = p; q < m; ++q) t.len += parts[q].len;
                        parts[p - 1] = t;
                        m = p;
                    } else {
                        for (int q = p + 1; q < m; ++q) parts[p].len += parts[q].len;
                        m = p + 1;
                    }
                    break;
                }
            }
        }
        /* a last seed of <= 5 k-mers at the very start of an edge is not trusted (:1298-1312) */
        if (parts[m - 1].gap && m > 1) {
            const ppart* l2 = &parts[m - 2];
            if (l2->off == 0 && l2->len <= 5) { ppart last = parts[m - 1]; last.len += l2->len; m -= 2; parts[m++] = last; }
        } else if (!parts[m - 1].gap) {
            ppart* l = &parts[m - 1];
            if (l->off == 0 && l->len <= 5) { ppart g = {1, 0, 0, 0, l->len, 0}; *l = g; }
        }
        /* pathPartsToReadPath :1393-1428 */
        int np = 0;
        int32_t offset = 0;
        const ppart* plast = NULL;
        for (int i = 0; i < m; ++i) {
            if (parts[i].gap) continue;
            if (plast && same_edge(plast, &parts[i])) continue;
            if (np < SNO_PATH_CAP) path[np++] = part_edge(&c, &parts[i]);
            plast = &parts[i];
        }
        if (np) offset = !parts[0].gap ? (int32_t)parts[0].off : (int32_t)parts[1].off - (int32_t)parts[0].len;
        /* adjacent edges must share a vertex (:1316-1323) */
        for (int i = 0; i + 1 < np; ++i) if (h->v_right[path[i]] != h->v_left[path[i + 1]]) { np = i + 1; break; }
        /* ExtendReadPath::attemptLeftRightExtension :108-119 */
        while (extend_once(&c, path, &np, SNO_PATH_CAP, &offset, rb, rq, n, 1)) {}
        while (extend_once(&c, path, &np, SNO_PATH_CAP, &offset, rb, rq, n, 0)) {}
        if (tot + (uint64_t)np > ecap) { ecap = ecap * 2 + (uint64_t)np; edges = (int32_t*)realloc(edges, ecap * 4); }
        memcpy(edges + tot, path, (size_t)np * 4);
        out_off[r] = offset;
        out_n[r] = np;
        tot += (uint64_t)np;
    }
    free(parts); free(np_);
    free(c.d.e); free(c.to_off); free(c.to_v); free(c.to_e); free(c.from_off); free(c.from_v); free(c.from_e);
    if (rc) { free(edges); return rc; }
    *out_edges = edges;
    *out_total = tot;
    return 0;
}
void sno_free(void* p) { free(p); }

/* ---------------------------------------------------------------------------------------------------------------
 * f4: MarkDups, lib/assembly/src/10X/SecretOps.cc:413-593 (DF.cc:597-600).  Plain restatement: the records
 * (first edge, offset, head of the mate, read id) as the reference builds them (:430-441), qsort in place of
 * sortInPlaceParallel (:447), the marking walk (:449-474), the quality sums (:480-499), the finalising walk with
 * its artifact check (:505-556).  first_edge[r] < 0 = read r has no path.  dup / art: one byte per PAIR. */
typedef struct { int32_t e, off, head; int64_t id; } md_rec;
static int md_cmp(const void* a, const void* b) {
    const md_rec* x = (const md_rec*)a; const md_rec* y = (const md_rec*)b;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    if (x->off != y->off) return x->off < y->off ? -1 : 1;
    if (x->head != y->head) return x->head < y->head ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
typedef struct { const uint8_t* b; const uint8_t* q; uint32_t len; int64_t pair; } md_qb;
static int md_qb_cmp(const void* a, const void* b) {          /* Sort(qb): bases, then qualities, then the pair (:537) */
    const md_qb* x = (const md_qb*)a; const md_qb* y = (const md_qb*)b;
    uint32_t n = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->b, y->b, n);
    if (c) return c;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    c = memcmp(x->q, y->q, n);
    if (c) return c;
    return x->pair < y->pair ? -1 : (x->pair > y->pair ? 1 : 0);
}
int sno_mark_dups(const uint8_t* bases, const uint8_t* quals, uint32_t stride, const uint32_t* lens, uint64_t n_reads,
                  const int32_t* first_edge, const int32_t* offset, const int32_t* bc, uint8_t* dup, uint8_t* art,
                  double* interdup_rate, uint64_t* n_dups, uint64_t* n_interdups) {
    const int BHEAD = 5;
    if (n_reads & 1ull) return -1;
    md_rec* X = (md_rec*)malloc((n_reads ? n_reads : 1) * sizeof(md_rec));
    uint8_t* dup1 = (uint8_t*)calloc(n_reads ? n_reads : 1, 1);
    int64_t* qsum = (int64_t*)calloc(n_reads ? n_reads : 1, sizeof(int64_t));
    if (!X || !dup1 || !qsum) { free(X); free(dup1); free(qsum); return -2; }
    memset(dup, 0, n_reads / 2);
    memset(art, 0, n_reads / 2);
    for (uint64_t id1 = 0; id1 < n_reads; ++id1) {
        const uint64_t id2 = id1 ^ 1ull;
        if (first_edge[id1] < 0) { X[id1].e = -1; X[id1].off = -1; X[id1].head = -1; X[id1].id = -1; }
        else {
            int n = 0;
            for (int j = 0; j < BHEAD; ++j) n = n * 4 + bases[id2 * stride + j];
            X[id1].e = first_edge[id1]; X[id1].off = offset[id1]; X[id1].head = n; X[id1].id = (int64_t)id1;
        }
    }
    qsort(X, n_reads, sizeof(md_rec), md_cmp);
    uint64_t ndups = 0, interdups = 0;
    for (uint64_t j = 0; j < n_reads; ++j) {
        if (X[j].e < 0) continue;
        uint64_t k;
        for (k = j + 1; k < n_reads; ++k) {
            if (X[k].e != X[j].e || X[k].off != X[j].off) break;
            if (X[k].head != X[j].head) break;
        }
        if (k - j > 1) {
            for (uint64_t l = j; l < k; ++l) dup1[X[l].id] = 1;
            ndups += k - j - 1;
            int inter = 0;
            int32_t b = bc ? bc[X[j].id] : 0;
            for (uint64_t l = j + 1; l < k; ++l) {
                const int32_t c = bc ? bc[X[l].id] : 0;
                if (b == 0) b = c;
                else if (c != b) inter = 1;
            }
            if (inter) interdups += k - j - 1;
        }
        j = k - 1;
    }
    *interdup_rate = ndups ? (double)interdups / (double)ndups : 0.0;
    for (uint64_t id1 = 0; id1 < n_reads; ++id1) {
        if (!dup1[id1]) continue;
        const uint64_t id2 = id1 ^ 1ull;
        for (uint32_t l = 0; l < lens[id1]; ++l) qsum[id1] += quals[id1 * stride + l];
        for (uint32_t l = 0; l < lens[id2]; ++l) qsum[id1] += quals[id2 * stride + l];
    }
    md_qb* qb = NULL;
    uint64_t qcap = 0;
    for (uint64_t j = 0; j < n_reads; ++j) {
        uint64_t k;
        for (k = j + 1; k < n_reads; ++k) {
            if (X[k].e != X[j].e || X[k].off != X[j].off) break;
            if (X[k].head != X[j].head) break;
        }
        if (X[j].e < 0) { j = k - 1; continue; }
        uint64_t best = j;
        int64_t q = qsum[X[j].id];
        int tie = 0;
        for (uint64_t l = j + 1; l < k; ++l) {
            if (qsum[X[l].id] == q) { tie = 1; if (X[l].id < X[best].id) best = l; }
            else if (qsum[X[l].id] > q) { q = qsum[X[l].id]; best = l; }
        }
        if (tie) {
            const uint64_t m = k - j;
            if (m > qcap) { qcap = m * 2; qb = (md_qb*)realloc(qb, qcap * sizeof(md_qb)); }
            for (uint64_t l = j; l < k; ++l) {
                const uint64_t id = (uint64_t)X[l].id;
                qb[l - j].b = bases + id * stride; qb[l - j].q = quals + id * stride; qb[l - j].len = lens[id]; qb[l - j].pair = (int64_t)(id / 2);
            }
            qsort(qb, m, sizeof(md_qb), md_qb_cmp);
            for (uint64_t a = 0; a < m; ++a) {
                uint64_t b2;
                for (b2 = a + 1; b2 < m; ++b2) {
                    if (qb[b2].len != qb[a].len || memcmp(qb[b2].b, qb[a].b, qb[a].len)) break;
                    if (memcmp(qb[b2].q, qb[a].q, qb[a].len)) break;
                }
                for (uint64_t x = a + 1; x < b2; ++x) art[qb[x].pair] = 1;
                a = b2 - 1;
            }
        }
        for (uint64_t l = j; l < k; ++l) if (l != best) dup[X[l].id / 2] = 1;
        j = k - 1;
    }
    *n_dups = ndups;
    *n_interdups = interdups;
    free(qb); free(X); free(dup1); free(qsum);
    return 0;
}
