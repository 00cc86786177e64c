/* snk_oracle.h -- CPU restatement of the reference count+graph path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * nothing under supernova_amd/ imports, links or executes it.  It restates, in plain C, the
 * algorithm of the reference's path B (lib/assembly, C++) -- see the per-function citations in
 * snk_oracle.c -- and is pinned against golden vectors produced by the reference itself
 * (oracle/ref/build_ref.sh -> oracle/_ref/snref_driver -> tests/golden/, generator script
 * tests/golden/make_golden.py) plus the reference's own known-answer tests.
 */
#ifndef SNK_ORACLE_H_
#define SNK_ORACLE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sno_table {
    uint64_t n;        /* retained canonical k-mers, sorted ascending */
    uint32_t* key;     /* n*4 words, MSB-first (word 3 = 0 for K=48) */
    uint32_t* count;   /* u32 observation counts (reference B saturates at 2^24-1) */
    uint8_t* ctx_raw;  /* OR of observed contexts before the prune */
    uint8_t* ctx;      /* after recomputeAdjacencies */
} sno_table;

typedef struct sno_unitigs {
    uint64_t n;
    uint64_t* off;     /* n+1 offsets into bases */
    uint8_t* bases;    /* base codes 0..3, concatenated; canonical orientation; sorted (len desc, lex) */
} sno_unitigs;

typedef struct sno_hbv {
    int32_t n_vertices;
    int32_t n_edges;
    int32_t* v_left;    /* per HBV edge */
    int32_t* v_right;
    int32_t* src_unitig; /* per HBV edge: index of the unitig it came from */
    uint8_t* is_rc;      /* per HBV edge: 1 if it is the reverse complement of that unitig */
    int32_t* fwd_xlat;   /* per unitig */
    int32_t* rev_xlat;
} sno_hbv;

typedef struct sno_slice { uint32_t value, min_pos, start, len; } sno_slice;

uint32_t sno_good_len(const uint8_t* quals, uint32_t len, uint32_t K, uint32_t min_qual);
int sno_msp_scan(uint32_t k, uint32_t p, const uint8_t* seq, uint32_t len, const uint32_t* perm, sno_slice* out,
                 int cap);
int sno_count(const uint8_t* bases, uint32_t stride, const uint32_t* good_len, const int32_t* bc, uint64_t n_reads,
              uint32_t K, uint32_t min_freq, uint32_t min_bc, int64_t ign_bc_below, sno_table* out,
              uint64_t* n_instances);
int sno_unitigs_build(const sno_table* t, uint32_t K, sno_unitigs* out);
int sno_hbv_build(const sno_unitigs* u, uint32_t K, sno_hbv* out);
int sno_write_bv(const char* path, const sno_unitigs* u);
int sno_read_bv(const char* path, sno_unitigs* out);
/* f1: read paths (offset of the read on its first edge, HBV edge ids) of pathReads with the new aligner,
 * paths/long/BuildReadQGraph48.cc:705-748,1217-1336,1393-1469 + paths/long/ExtendReadPath.cc.  Untrimmed reads; unitigs in
 * BVComp order with their HBV (sno_hbv_build).  *out_edges is malloc'ed (sno_free). */
int sno_path_reads(const uint8_t* bases, const uint8_t* quals, uint32_t stride, const uint32_t* lens, uint64_t n_reads, uint32_t K,
                   const sno_unitigs* u, const sno_hbv* h, int32_t* out_off, int32_t* out_n, int32_t** out_edges, uint64_t* out_total);
/* f4: MarkDups (10X/SecretOps.cc:413-593) over read paths given as (first edge or -1, offset) per read; reads 2q, 2q+1 are
 * mates; bc may be NULL (all 0).  dup / art: one byte per pair.  Pinned by the reference's own MarkDups on the golden cases. */
int sno_mark_dups(const uint8_t* bases, const uint8_t* quals, uint32_t stride, const uint32_t* lens, uint64_t n_reads,
                  const int32_t* first_edge, const int32_t* offset, const int32_t* bc, uint8_t* dup, uint8_t* art,
                  double* interdup_rate, uint64_t* n_dups, uint64_t* n_interdups);
void sno_free(void* p);
void sno_table_free(sno_table* t);
void sno_unitigs_free(sno_unitigs* u);
void sno_hbv_free(sno_hbv* h);

#ifdef __cplusplus
}
#endif
#endif
