/* snk.h -- C ABI of libsnk: MI355X-native k-mer count + de Bruijn unitig graph (Supernova hot path).
 *
 * This is the drop-in boundary of SURVEY.md section 8(b), row b5.  The reference has no C ABI on this
 * path; its seams are
 *   - the C++ entry  buildReadQGraph48(...)            lib/assembly/src/paths/long/BuildReadQGraph48.h:24-34
 *   - its tail       buildHBVFromEdges(...)            lib/assembly/src/paths/long/HBVFromEdges.h:27-28
 *   - the unitig hand-off file (.bv) tada -> DF        lib/tada/src/debruijn.rs:895-929,
 *                                                      lib/assembly/src/paths/long/BuildReadQGraph48.cc:1640-1642
 *   - the stage argv/MRO contracts (ASSEMBLER_DF, MSP/SHARD_ASM/MAIN_ASM_SN)
 *                                                      mro/_assembler_stages.mro:24-39, lib/tada/mro/_asm_stages.mro:53-80
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes only.
 *
 * Conventions
 *   - return value: 0 = SNK_OK, negative = error; the message is left in the caller's `err` buffer
 *     (NUL terminated, truncated to errcap) and is also retrievable with snk_last_error().
 *   - base codes A=0 C=1 G=2 T=3; every non-ACGT input character maps to A
 *     (lib/tada/src/kmer/mod.rs:311-319, lib/assembly/src/10X/ParseBarcodedFastqs.cc:87-88).
 *   - packed read rows: `row_words` u32 words per read, base i of a read in word i>>4, bits
 *     31-2*(i&15)..30-2*(i&15) (MSB first; the same order as KMer<K>'s storage words, kmers/KMer.h:153-160).
 *   - k-mer keys are 4 u32 words MSB-first (K=48 uses words 0..2, word 3 = 0; K=60 uses all four with
 *     the low 8 bits of word 3 zero); lexicographic order on words == order on bases (KMer.h:305-311).
 *   - context byte = pred one-hot << 4 | succ one-hot in the k-mer's canonical orientation
 *     (kmers/KMerContext.h:27-28), after the adjacency prune (kmers/ReadPather.h:346-385).
 *   - barcode ids: int32 per read, 0 = no barcode, -1 = "ignore the barcode rule for this read"
 *     (BuildReadQGraph48.cc:108-114,158-159), >0 = barcode ordinal.
 *   - "dev" entry points take device pointers (HBM resident) and a hipStream_t passed as void*.
 */
#ifndef SNK_H_
#define SNK_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNK_OK 0
#define SNK_E_ARG (-1)      /* bad argument */
#define SNK_E_HIP (-2)      /* HIP runtime error (message carries hipGetErrorString) */
#define SNK_E_NOGPU (-3)    /* no gfx950 device visible: the product path never falls back to the CPU */
#define SNK_E_NOMEM (-4)    /* device/host allocation failed (adapter maps it to exit 99, system/RunTime.cc:195-221) */
#define SNK_E_IO (-5)
#define SNK_E_UNSUPPORTED (-6)
#define SNK_E_INTERNAL (-7)

typedef struct snk_ctx snk_ctx; /* one per process per GPU: owns streams, arena, scratch */

/* thresholds of the path; defaults = CS-build constants K=48 MIN_FREQ=3 MIN_BC=2 MIN_QUAL=7
 * (lib/assembly/src/10X/DF.cc:138-141,181-184; mro/_assembler.mro:44; lib/tada/mro/_asm_sn.mro:15) */
typedef struct snk_params {
    uint32_t K;            /* 48 or 60 */
    uint32_t min_qual;     /* 7 */
    uint32_t min_freq;     /* 3 */
    uint32_t min_bc;       /* 0..8 distinct barcodes a k-mer needs (2 = reference default; the count kernel tells up to eight apart
                              per k-mer in LDS: > 8 is SNK_E_UNSUPPORTED) */
    uint32_t n_buckets;    /* 0 = choose from the k-mer instance count */
    uint32_t flags;        /* SNK_F_* */
} snk_params;
#define SNK_F_NO_GRAPH 1u      /* stop after the retained k-mer table (count only) */
#define SNK_F_UNSORTED_TABLE 2u /* leave the retained table in bucket order (the reference's dictionary is an unordered
                                  hash set, kmers/ReadPather.h:189-245); default: keys ascending */
#define SNK_F_GROUPED 8u        /* per-group graphs (BASELINE config 5: per-barcode local graphs): reads carry a group id
                                  (snk_dev_reads.group); k-mers are counted, pruned and walked per (group, k-mer); every
                                  unitig reports its group.  K=48, frequency rule only.  The group id is returned in the
                                  32 low bits of every key. */
#define SNK_F_NO_TABLE 16u      /* snk_count_graph (host pointers): do not download the retained table (kmers/counts/ctx stay
                                  NULL, n_kmers is still reported) -- callers at the .bv seam only need the unitigs */
#define SNK_F_BV_IMAGE 32u      /* snk_count_graph (host pointers): return the unitigs as the bytes of the .bv hand-off file
                                   (snk_result.bv_image, packed on the device) instead of unitig_off / unitig_bases */
#define SNK_F_LONG_MINIMISER 64u /* minimisers of 20 bases instead of 16: for genomes whose minimiser sites outnumber the 2.1 G canonical 16-mers
                                   (human: 3.1 G) -- sites that share a minimiser share a bucket, and at 1-2 sites per value the step loses
                                   5-25 %.  11 % more supermers: slower on small genomes.  Same results, bit for bit. */
#define SNK_F_GLOBAL_GRAPH 4u   /* use the global graph stage (sort + HBM index + list ranking over all k-mers) instead
                                  of the bucket-local one; same results, kept as a cross-check */

const char* snk_version(void);
const char* snk_last_error(void);
void snk_params_default(snk_params* p);

/* ---- context ------------------------------------------------------------------------------------ */
int snk_ctx_create(int device, snk_ctx** out, char* err, size_t errcap);
void snk_ctx_destroy(snk_ctx* ctx);
/* Give the context's cached, currently unused device memory back to the device (results of the last call stay valid).  The
 * context keeps its scratch between calls so that a steady stream of equal-sized calls never allocates; a caller that shares
 * the GPU with another allocator calls this when it changes problem size.  Blocks unused for two calls are dropped anyway. */
void snk_ctx_trim(snk_ctx* ctx);
/* Map `bytes` of device memory into the context's scratch arena now and keep that much mapped between calls (until snk_ctx_trim).  A call
 * that needs more scratch than the context has mapped so far pays the driver for the new memory inside the call (~25-30 ms per GB: 100 M
 * error-rich reads after 100 M clean ones: +20 GB, 745 instead of 228 ms for that one call); a host that owns the GPU reserves its share
 * once, at start-up.  Returns SNK_E_NOMEM when the device cannot give that much (what could be mapped stays usable).  The reservation
 * outlives the arena's own resets (a sealed range, a shrink after much larger calls): it is mapped again at the start of the next call; only a
 * multi-rank step (plain device blocks for the xGMI buffers) runs without it, and the call after that step maps it again. */
int snk_ctx_reserve(snk_ctx* ctx, uint64_t bytes, char* err, size_t errcap);

/* ---- tuning (round 6) ------------------------------------------------------------------------------------------------------------
 * Every choice the library makes by itself -- which count kernel runs, how large a minimiser bucket is, how many partition passes, the
 * minimiser length, the thresholds of the hot-bucket path, how reads are looked up -- can be pinned per context, and read back together
 * with what the context's last call chose.  Nothing on the product path reads the process environment (tracing switches aside); a Martian
 * stage sets what it wants here and logs snk_ctx_get_tuning.  A zero field = the library's own choice.  Results never depend on any of it.
 * The long tail (test hooks, probe modes) is reachable by name: snk_ctx_set_option / snk_option_name enumerate them with a one-line
 * description each; SNK_TUNING="name=value,name=value" in the environment is applied once, when a context is created (shell tools). */
#define SNK_COUNT_KERNEL_AUTO 0u     /* from the data: the distinct-k-mers-per-instance ratio of the first buckets / the previous call */
#define SNK_COUNT_KERNEL_MARGIN 1u   /* 1216 of 2048 table slots usable, no bookkeeping (clean, deep data: the bench's operating point) */
#define SNK_COUNT_KERNEL_BOOKED 2u   /* waves book their slots: count_tight_slots (1920) usable -- error-rich reads, per-barcode groups */
#define SNK_COUNT_KERNEL_SCREEN 3u   /* bit filter in front of a 1024-slot table with booked slots -- reads whose k-mers are mostly singletons */
typedef struct snk_tuning {
    uint32_t count_kernel;            /* SNK_COUNT_KERNEL_* */
    uint32_t count_tight_slots;       /* BOOKED: usable slots (256 .. 1984), 0 = 1920 */
    uint32_t count_screen_ratio_pct;  /* AUTO: the filter goes on above this many distinct k-mers per 100 instances (0 = 30) */
    uint32_t target_inst;             /* k-mer instances per minimiser bucket (0 = 5000 at K=48 / 3500 at K=60, adapted to the data) */
    uint32_t bucket_fill_pct;         /* adaptive buckets aim at this share of the usable slots (0 = 50) */
    uint32_t adaptive_buckets;        /* 0 default (on), 1 on, 2 off: look at the first buckets of unknown data, partition again if their tables run full */
    uint32_t chunk_kmers;             /* retained k-mers per bucket aimed at when the data retain many (0 = 180) */
    uint32_t minimiser_len;           /* 0 = from snk_params.flags (SNK_F_LONG_MINIMISER), else 16 | 20 */
    uint32_t partition_passes;        /* bucket-range passes (0 = as many as the device needs) */
    uint32_t hot_buckets;             /* 0 default (on), 1 on, 2 off: re-partition hot minimiser buckets by k-mer hash */
    uint32_t hot_min, hot_factor, hot_class_inst;   /* 0 = 8192 records, 8 x the slot capacity, 6000 instances per class */
    uint32_t exchange_ranges;         /* sharded step: bucket ranges of the record exchange (0 = 4) */
    uint32_t join_ranking;            /* sharded step: 0 default (partitioned), 1 partitioned, 2 replicated */
    uint32_t path_lookup;             /* read pathing: 0 = dictionary when it fits, 1 minimiser index, 2 dictionary */
    uint32_t unitig_bc_cut;           /* entries a unitig's barcode list is cut at (0 = 20000, cmd_main_asm.rs:115) */
    uint32_t hbv_dev_min, hbv_big;    /* graph ids: below hbv_dev_min unitigs (0 = 65536) / above hbv_big nodes per component (0 = 1024) on the host */
    uint32_t reserved[9];
    /* filled by snk_ctx_get_tuning: what the context's last call ran with */
    uint32_t last_count_kernel, last_count_limit, last_partition_passes, last_minimiser_len;
} snk_tuning;
void snk_tuning_default(snk_tuning* t);
int snk_ctx_set_tuning(snk_ctx* ctx, const snk_tuning* t, char* err, size_t errcap);
void snk_ctx_get_tuning(const snk_ctx* ctx, snk_tuning* t);
int snk_ctx_set_option(snk_ctx* ctx, const char* name, long long value, char* err, size_t errcap);
int snk_ctx_clear_option(snk_ctx* ctx, const char* name);                    /* NULL: every option back to the library's choice */
int snk_ctx_get_option(const snk_ctx* ctx, const char* name, long long* value);   /* 1 set, 0 not set, < 0 no such option */
const char* snk_option_name(uint32_t i);                                      /* NULL past the last one */
const char* snk_option_doc(uint32_t i);

/* ---- synthetic linked reads; counter-based, bit-identical host vs device ---------- */
typedef struct snk_synth_params {
    uint64_t seed;
    uint64_t n_reads;          /* total reads of the data set (pairs are reads 2q, 2q+1) */
    uint64_t genome_len;       /* 0 -> n_reads*read_len/56 (56x coverage)              */
    uint32_t read_len;         /* 150 */
    uint32_t mol_len;          /* 50000 (clamped to genome_len) */
    uint32_t mols_per_bc;      /* 10 */
    uint32_t pairs_per_bc;     /* 400 (=> 800 reads per barcode) */
    uint32_t insert_min;       /* 300 */
    uint32_t insert_span;      /* 101 (insert uniform in [300,400]) */
    uint32_t sub_ppm;          /* substitution probability per base, parts per million (2000) */
    uint32_t unbarcoded_ppm;   /* pairs with bc=0 (20000) */
    uint32_t lowq_tail_ppm;    /* reads with a Q2 tail (50000) */
    uint32_t tail_max;         /* tail length uniform in [0,tail_max] (40) */
    uint32_t err_cdf[4];       /* filled by snk_synth_default: P(#errors<=j)*2^32 for j=0..3 (snk_synth_set_errors refills it) */
    uint32_t repeat_mode;      /* 0: i.i.d. uniform genome.  Bit mask (15 = all) of a repeat-rich genome, every structure a pure function of the
                                  position (csrc/snk_synth.h): bit 0 four interspersed families (a 300-bp element in 60 % of the 4-kb
                                  blocks, 1-3 % divergence from its consensus: ~10^4 copies each at the bench's genome size), bit 1 5-kb
                                  segmental duplications (exact copies, one in four odd 64-kb superblocks), bit 2 short tandem repeats
                                  (unit 1-6 bp, 40-200 bp, 3 % of the blocks), bit 3 poly-A runs (20-80 bp, 2 %); bit 4 (16, not part of 15): every
                                  fifth base is A -- a minimiser space as crowded as a human genome's at a fraction of its size */
    uint32_t reserved[3];
} snk_synth_params;
/* substitution rate of the model: sub_ppm and the error-count table that goes with it */
void snk_synth_set_errors(snk_synth_params* sp, uint32_t sub_ppm);
void snk_synth_default(snk_synth_params* sp, uint64_t n_reads, uint64_t seed, int error_free);
/* host generator: reads [first, first+n).  rows: n*row_words u32; quals: n*qstride bytes (raw phred);
 * bc: n int32.  Any output pointer may be NULL. */
int snk_synth_host(const snk_synth_params* sp, uint64_t first, uint64_t n, uint32_t* rows, uint32_t row_words,
                   uint8_t* quals, uint32_t qstride, int32_t* bc);
int snk_synth_dev(snk_ctx* ctx, const snk_synth_params* sp, uint64_t first, uint64_t n, void* d_rows,
                  uint32_t row_words, void* d_quals, uint32_t qstride, void* d_bc, void* stream);


/* ---- device-resident path -------------------------------------------------------------------------- */
/* K1: replaces GoodLenTailFinder (BuildReadQGraph48.cc:65-89) / find_trim_len (lib/tada/src/cmd_msp.rs:129-146).
 * d_quals: n_reads rows of qstride bytes (raw phred, no +33); d_lens: u16 per read or NULL (= read_len);
 * d_good_len: u16 per read out. */
int snk_dev_trim(snk_ctx* ctx, const void* d_quals, uint32_t qstride, const void* d_lens, uint32_t read_len,
                 uint64_t n_reads, uint32_t K, uint32_t min_qual, void* d_good_len, void* stream);
/* K2: replaces base_to_bits (lib/tada/src/kmer/mod.rs:311-319): ASCII rows -> packed 2-bit rows. */
int snk_dev_pack_ascii(snk_ctx* ctx, const void* d_ascii, uint32_t astride, uint32_t read_len, uint64_t n_reads,
                       void* d_rows, uint32_t row_words, void* stream);

typedef struct snk_dev_reads {
    uint64_t n_reads;
    const void* rows;         /* u32[n_reads*row_words] packed bases (HBM) */
    uint32_t row_words;
    uint32_t read_len;        /* <= 256 */
    const void* lens;         /* u16[n_reads] or NULL (all reads are read_len long) */
    const void* quals;        /* u8[n_reads*qstride] raw phred, or NULL if good_len is given */
    uint32_t qstride;
    uint32_t reserved0;
    const void* good_len;     /* u16[n_reads] precomputed trim, or NULL */
    const void* bc;           /* i32[n_reads] barcode ids, or NULL (no barcode rule) */
    int64_t ign_bc_below;     /* reads with global index < this get bc = -1 (BuildReadQGraph48.cc:158-159) */
    uint64_t read_index_base; /* global index of read 0 of this slab (multi-GPU slabs) */
    const void* group;        /* u32[n_reads] group id per read (SNK_F_GROUPED), or NULL */
} snk_dev_reads;

/* All pointers are device pointers owned by the context; they stay valid until the next
 * snk_dev_count_graph call on the same context or snk_ctx_destroy. */
typedef struct snk_dev_result {
    uint64_t n_reads;
    uint64_t n_instances;        /* sum over reads of max(0, good_len-K+1), reads with good_len < K+1 excluded */
    uint64_t n_supermers;
    uint64_t n_buckets;
    const void* good_len;        /* u16[n_reads] */
    uint64_t n_kmers;            /* retained canonical k-mers */
    const void* keys;            /* n_kmers x 16 bytes: little-endian 128-bit value = {u64 lo, u64 hi}; base i of the
                                    k-mer at bits 127-2i..126-2i; ascending unless SNK_F_UNSORTED_TABLE */
    const void* counts;          /* u32[n_kmers] */
    const void* ctx;             /* u8[n_kmers] pruned context bytes */
    const void* spectrum;        /* u64[spectrum_bins]: retained k-mers per count (histogram_kmer_count.json) */
    uint32_t spectrum_bins;
    uint32_t n_circles;
    uint64_t n_unitigs;
    uint64_t unitig_total_bases;
    const void* unitig_off;      /* u64[n_unitigs+1] */
    const void* unitig_bases;    /* u8 base codes, canonical orientation, unitigs ordered by their first K bases */
    uint32_t rank_rounds;
    uint32_t buckets_split;
    uint32_t max_slots_used;
    uint32_t n_overflow;         /* supermers that did not fit their bucket's fixed capacity (second count segment) */
    uint64_t scratch_bytes;
    float phase_ms[8];           /* trim, partition plan, minimiser partition, count, sort, prune+unitigs, -, total */
    float kernel_ms[4];          /* HIP-event time of single launches: -, minimiser partition, count (LDS reduce), - */
    uint64_t n_boundary;         /* bucket-local graph: k-mers with a neighbour outside their bucket chunk */
    uint64_t n_fragments;        /* bucket-local graph: local unitig fragments joined at the end */
    const void* unitig_group;    /* u32[n_unitigs] group of every unitig (SNK_F_GROUPED; unitigs ordered by group, then by
                                    their first K bases), else NULL */
    float graph_ms[8];           /* bucket-local graph: local prune, boundary resolve, fragments, join, table sort+spectrum */
    uint32_t repartitioned;      /* 1: the first buckets overflowed their tables (error-rich / shallow data) and the reads were
                                    partitioned a second time into smaller buckets; later calls on the context start there */
    uint32_t n_hot_buckets;      /* minimiser buckets far above their capacity (repeat families, homopolymer runs) whose k-mer instances
                                    were re-partitioned by k-mer hash and counted class by class */
} snk_dev_result;

/* Replaces the body of buildReadQGraph48 (BuildReadQGraph48.cc:1688-1774, pPaths==nullptr) up to and
 * including buildEdges; == tada MSP -> SHARD_ASM -> MAIN_ASM_SN.  Inputs already resident in HBM. */
int snk_dev_count_graph(snk_ctx* ctx, const snk_dev_reads* in, const snk_params* p, snk_dev_result* out, void* stream,
                        char* err, size_t errcap);
int snk_dev_download(snk_ctx* ctx, const void* d_src, void* h_dst, size_t bytes, void* stream);

/* Streamed input: the same job with its reads arriving slab by slab (as a FASTH decoder delivers them).  The reference streams its
 * FASTQ chunks through the partitioner (lib/tada/src/cmd_msp.rs:55-69) and re-scans in passes when the keys do not fit
 * (MapReduceEngine.h:452-468); here every slab is partitioned into the job's minimiser buckets as it arrives -- the launch is
 * asynchronous, so the decode / upload of the next slab overlaps it -- and the slab's buffers may be reused once the stream has passed
 * the call (an event, or a synchronise).  The job's reads are never resident as a whole.
 *   begin : total_reads_ub = an upper bound of the job's reads (it sizes the bucket slots; reads beyond it are refused),
 *           has_bc = the slabs carry barcode ids (all of them or none).  Per-group graphs (SNK_F_GROUPED) are not streamed.
 *   append: slab->read_index_base = global index of its first read (ign_bc_below); 0 = numbered in arrival order.
 *   finish: count + graph over everything appended; result as snk_dev_count_graph's (good_len covers the reads in arrival order),
 *           bit-identical to one resident call on the concatenated slabs.
 * A job that cannot look at its first buckets and partition again (its slabs are gone): error-rich data without the context's
 * history of an earlier job of the same size are counted in hash-split sub-passes -- slower, same result. */
int snk_dev_stream_begin(snk_ctx* ctx, const snk_params* p, uint32_t read_len, uint64_t total_reads_ub, int has_bc, void* stream, char* err, size_t errcap);
int snk_dev_stream_append(snk_ctx* ctx, const snk_dev_reads* slab, void* stream, char* err, size_t errcap);
int snk_dev_stream_finish(snk_ctx* ctx, snk_dev_result* out, void* stream, char* err, size_t errcap);

/* ---- minimiser-sharded multi-GPU path: ONE entry point, snk_shard_step (below); its phases are internal
 * (supernova_amd/csrc/snk_shard_phases.h) since the step moved behind the C ABI in round 3. */
/* fragment bases on the wire (the gather to rank 0): 2 bits per base, 16 bases per 32-bit word, base j at bits 2j --
 * the .bv byte packing (lib/tada/src/debruijn.rs:895-929).  snk_pack2_bytes(n) = size of the packed buffer. */
uint64_t snk_pack2_bytes(uint64_t n_bases);
int snk_dev_pack2(snk_ctx* ctx, const void* d_bases, uint64_t n_bases, void* d_packed, void* stream);
int snk_dev_unpack2(snk_ctx* ctx, const void* d_packed, uint64_t n_bases, void* d_bases, void* stream);


/* ---- the sharded step behind ONE call (round 3) ------------------------------------------------------------------------
 * A communicator carries the exchanges; the step itself -- partition, histogram and record exchange (in bucket ranges,
 * overlapped with the count), count, cross-rank prune, fragments, fragment links, owner-side join -- runs inside
 * snk_shard_step, so a C++ host (supernova_amd/csrc/host/snk_asm_sn.cc) runs the N-GPU job without any scripting layer.
 * Replaces tada's MSP -> SHARD_ASM -> MAIN_ASM_SN with its shard files (lib/tada/src/cmd_msp.rs:38-80, cmd_shard_asm.rs:37-94,
 * cmd_main_asm.rs:25-89; lib/tada/external/rust-shardio/src/shard.rs:184-211,488-493) and MapReduceEngine.h:362-385.
 *   snk_comm_unique_id      rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks (file, socket, ...)
 *   snk_comm_create_rccl    every rank: ncclCommInitRank on the context's device; librccl is bound at run time
 *                           (snk_comm_set_rccl_path / $SNK_RCCL_LIB / librccl.so.1 / $ROCM_PATH/lib)
 *   snk_comm_from_nccl      adopt the caller's ncclComm_t (not destroyed by snk_comm_destroy)
 *   snk_comm_create_local   `world` in-process ranks on ONE device (tests): out[r] is rank r's handle, every rank runs on its
 *                           own host thread with its own context; the wire is a device copy */
typedef struct snk_comm snk_comm;
int snk_comm_set_rccl_path(const char* path);
int snk_comm_unique_id(void* id128, char* err, size_t errcap);
int snk_comm_create_rccl(snk_ctx* ctx, const void* id128, uint32_t rank, uint32_t world, snk_comm** out, char* err, size_t errcap);
int snk_comm_from_nccl(snk_ctx* ctx, void* nccl_comm, uint32_t rank, uint32_t world, snk_comm** out, char* err, size_t errcap);
int snk_comm_create_local(uint32_t world, snk_comm** out /* [world] */, char* err, size_t errcap);
/* the exchanges through callbacks of the host (its own transport: MPI, sockets, torch.distributed, ...): a2a moves scnt[p] bytes
 * at send + sbeg[p] to rank p and delivers rcnt[s] bytes from rank s at recv + rbeg[s]; gather collects k u64 of every rank
 * (mine and all are HOST memory: device counters are read back by the library first) into all[world * k].  a2a's buffers are
 * whatever memory the step hands over -- DEVICE memory inside snk_shard_step, not ordered with any stream: the callback
 * synchronises the device before it reads and after it writes.
 * Both return 0 or an error code.  snk_comm_selftest runs the step's exchange patterns on HOST memory with synthetic contents
 * over any communicator whose buffers may be host memory (the CPU tests use it over gloo, world_size 2). */
typedef int (*snk_comm_a2a_fn)(void* user, const void* send, const uint64_t* sbeg, const uint64_t* scnt, void* recv, const uint64_t* rbeg,
                               const uint64_t* rcnt, uint32_t world);
typedef int (*snk_comm_gather_fn)(void* user, const void* mine, uint32_t k, unsigned long long* all, uint32_t world);
int snk_comm_create_callbacks(uint32_t rank, uint32_t world, snk_comm_a2a_fn a2a, snk_comm_gather_fn gather, void* user, snk_comm** out,
                              char* err, size_t errcap);
int snk_comm_selftest(snk_comm* c, uint64_t seed, uint32_t buckets_per_rank, uint32_t ranges, char* err, size_t errcap);
void snk_comm_destroy(snk_comm* c);
void snk_comm_abort(snk_comm* c);          /* in-process ranks: release the others after a failure outside snk_shard_step */
uint32_t snk_comm_rank(const snk_comm* c);
uint32_t snk_comm_world(const snk_comm* c);
const char* snk_comm_kind(const snk_comm* c);   /* "rccl" | "local" */

/* Device pointers owned by the context, valid until its next top-level call.  The table share is in bucket order; the
 * unitigs are the ones whose head fragment this rank owns, ordered by their first K bases (the union over the ranks is the
 * job's unitig set; snk_shard_gather_unitigs brings it to one rank in BVComp order). */
typedef struct snk_shard_result {
    uint32_t rank, world;
    uint64_t n_reads, n_instances, n_supermers, n_buckets_total;
    uint64_t n_kmers;
    const void* keys;
    const void* counts;
    const void* ctx;
    const void* spectrum;
    uint32_t spectrum_bins, n_circles;
    uint64_t n_unitigs, unitig_total_bases;
    const void* unitig_off;          /* u64[n_unitigs+1] */
    const void* unitig_bases;
    const void* unitig_circular;
    uint64_t n_frags, n_frags_total, n_queries, n_link_queries;
    uint64_t exchanged_bytes[8];     /* sent to OTHER ranks: records, prune queries+answers, link queries+answers, link structure,
                                        splitters, ranks, fragments, everything the transport carried */
    uint32_t host_syncs;             /* host waits for the stream during the step (the read-backs of sizes) */
    uint32_t ranking;                /* 1 = the list ranking ran partitioned over the ranks, 0 = replicated */
    uint32_t buckets_split, max_slots_used;
    float phase_ms[8];               /* trim+partition, histograms+compaction, exchange issue, count, prune, fragments, join, total */
    float join_ms[8];                /* links, link structure, rank+place, route, emit */
    float count_kernel_ms;
    uint32_t repartitioned;          /* 1: the first buckets overflowed the count kernel's tables and the step partitioned and exchanged a
                                        second time into smaller buckets (a job-wide decision); later steps on the context start there */
    uint32_t n_hot_buckets;          /* this rank's minimiser buckets far above their capacity (repeat families, homopolymer runs): re-partitioned
                                        by k-mer hash and counted by a launch of their own, like snk_dev_result.n_hot_buckets */
    uint32_t reserved_u;
    uint64_t pair_max_bytes[8];      /* per exchange (the order of exchanged_bytes): the most this rank sent to ONE other rank.  xGMI is point to
                                        point: an exchange takes as long as its fullest pair, so max against exchanged_bytes / (world - 1) is the
                                        link balance of the step */
} snk_shard_result;
/* Bucket-range passes of the last snk_dev_count_graph on the context (1 = the one-pass partition).  A job whose supermer slots would not
 * fit the device is partitioned and counted range by range over one slot array, the reads scanned once per pass -- what the reference
 * does when its k-mer records do not fit (lib/assembly/src/MapReduceEngine.h:452-468, lib/tada/src/utils.rs:329-341).  Same results. */
uint32_t snk_ctx_last_partition_passes(const snk_ctx* ctx);
/* Distinct k-mers one pass over a bucket could hold in the count kernel's table in the last snk_dev_count_graph on the context: 1216 of the 2048
 * slots with the default kernel, 1920 when the call's data run the tables full (error-rich reads, per-barcode groups) and the waves book
 * their slots instead of keeping a round's worth free (DESIGN.md 4 "round 5").  Same results either way. */
uint32_t snk_ctx_last_count_limit(const snk_ctx* ctx);
/* total_reads: reads of the whole job (sizes the bucket count without an exchange; 0 = the ranks exchange their slab sizes,
 * ignored when p->n_buckets is set).  in->read_index_base = global index of the slab's first read. */
int snk_shard_step(snk_ctx* ctx, snk_comm* comm, const snk_dev_reads* in, const snk_params* p, uint64_t total_reads, uint32_t flags,
                   snk_shard_result* out, void* stream, char* err, size_t errcap);
/* The same step with the rank's reads arriving slab by slab (what snk_dev_stream_* is to snk_dev_count_graph; tada streams its FASTQ
 * chunks into the partitioner, lib/tada/src/cmd_msp.rs:55-69): begin sizes the job's buckets -- total_reads = reads of the whole job, the
 * same on every rank -- and this rank's slots (rank_reads_ub: an upper bound of what it will append); append partitions one slab (nothing
 * is waited for; slab->read_index_base = global index of its first read); finish runs the rest of the step.  A streamed step cannot
 * partition twice: without the group's history of a previous step, error-rich data are counted in hash-split sub-passes. */
int snk_shard_stream_begin(snk_ctx* ctx, snk_comm* comm, const snk_params* p, uint32_t read_len, uint64_t rank_reads_ub, uint64_t total_reads,
                           int has_bc, void* stream, char* err, size_t errcap);
int snk_shard_stream_append(snk_ctx* ctx, const snk_dev_reads* slab, void* stream, char* err, size_t errcap);
int snk_shard_stream_finish(snk_ctx* ctx, snk_comm* comm, uint32_t flags, snk_shard_result* out, void* stream, char* err, size_t errcap);
/* ---- host-pointer convenience + graph hand-off -------------- */
typedef struct snk_reads {
    uint64_t n_reads;
    uint32_t read_len;          /* row length in bases (<= 256); per-read lengths in `lens` */
    uint32_t reserved;
    const uint8_t* ascii;       /* n_reads*read_len base characters (non-ACGT -> A), or NULL if rows given */
    const uint32_t* rows;       /* packed rows (row_words = ceil(read_len/16)), or NULL if ascii given */
    const uint16_t* lens;       /* per-read length or NULL */
    const uint8_t* quals;       /* n_reads*read_len raw phred (no +33), or NULL if good_len given */
    const uint16_t* good_len;   /* precomputed trim or NULL */
    const int32_t* bc;          /* barcode ids or NULL */
    int64_t ign_bc_below;
} snk_reads;

typedef struct snk_result {
    uint64_t n_instances;
    uint64_t n_kmers;
    uint32_t* kmers;            /* n_kmers*4 words MSB-first, ascending */
    uint32_t* counts;
    uint8_t* ctx;
    uint64_t n_unitigs;
    uint64_t* unitig_off;       /* n_unitigs+1 */
    uint8_t* unitig_bases;      /* base codes; unitigs canonical, ordered by BVComp (len desc, lexicographic;
                                   lib/assembly/src/paths/long/HBVFromEdges.cc:106-111) */
    uint64_t* spectrum;         /* spectrum_bins */
    uint32_t spectrum_bins;
    uint32_t reserved;
    float phase_ms[8];
    uint8_t* bv_image;          /* SNK_F_BV_IMAGE: the .bv file ("BINWRITE", u64 count, per unitig u32 length + ceil(len/4)
                                   bytes; lib/tada/src/debruijn.rs:895-929), unitigs in BVComp order; else NULL */
    uint64_t bv_bytes;
} snk_result;

/* Replaces buildReadQGraph48(..., pPaths=nullptr) up to the unitigs for host-resident inputs
 * (BuildReadQGraph48.h:24-34).  Uploads over PCIe, runs snk_dev_count_graph, downloads and orders the result.
 * Outputs are malloc'ed by the library and released by snk_free. */
int snk_count_graph(snk_ctx* ctx, const snk_reads* in, const snk_params* p, snk_result* out, char* err, size_t errcap);
void snk_free(snk_result* r);

/* The job's unitig set on ONE rank in the reference's order -- what MAIN_ASM_SN writes to asm_graph.bv
 * (lib/tada/src/cmd_main_asm.rs:184-193, debruijn.rs:895-929): every rank (all must call) ships the unitigs it wrote at 2 bits
 * per base to `root`, which orders the union by BVComp (HBVFromEdges.cc:106-111) on its device.  On root, out holds n_unitigs +
 * unitig_off / unitig_bases, or with flags & SNK_F_BV_IMAGE the file's bytes (bv_image / bv_bytes); release with snk_free.
 * Other ranks get an empty result. */
int snk_shard_gather_unitigs(snk_ctx* ctx, snk_comm* comm, const snk_shard_result* res, uint32_t K, uint32_t root, uint32_t flags,
                             snk_result* out, void* stream, char* err, size_t errcap);

/* Page-locked host memory for the inputs of snk_count_graph (the reference keeps reads/quals in plain vectors, vecbvec /
 * VecPQVec, BuildReadQGraph48.h:24-34; a host that fills pinned buffers instead saves the staging copy: the DMA engine
 * reads them in place).  Pageable inputs are accepted just the same. */
int snk_host_alloc_pinned(size_t bytes, void** out, char* err, size_t errcap);
void snk_host_free_pinned(void* p);

/* a13: the unitig hand-off file tada writes and DF reads through MSPEDGES= ("BINWRITE", u64 count, per entry u32
 * length + ceil(len/4) bytes, base j at bits 2*(j%4)): writer TempGraph::write_to_sn_format
 * lib/tada/src/debruijn.rs:895-929, reader BuildReadQGraph48.cc:1640-1642. */
int snk_write_bv(const char* path, uint64_t n_unitigs, const uint64_t* unitig_off, const uint8_t* unitig_bases,
                 char* err, size_t errcap);
int snk_read_bv(const char* path, uint64_t* n_unitigs, uint64_t** unitig_off, uint8_t** unitig_bases, char* err,
                size_t errcap);   /* outputs malloc'ed; free() them */

/* a13 on the device: the bytes of that file from the device-resident unitigs of snk_dev_count_graph -- BVComp order (length
 * descending, then lexicographic: HBVFromEdges.cc:106-111), 2-bit packing and the "BINWRITE" header all in HBM; *d_image is context
 * memory (valid until the next top-level call), ready for one download or a GPU-direct write.  by_first_kmer: the unitigs are ordered
 * by their first K bases (how snk_dev_count_graph leaves them): one stable sort by length does. */
int snk_dev_bv_image(snk_ctx* ctx, uint32_t K, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases, int by_first_kmer,
                     const void** d_image, uint64_t* image_bytes, void* stream, char* err, size_t errcap);

/* a14: buildHBVFromEdges (lib/assembly/src/paths/long/HBVFromEdges.cc:244-296): vertices = distinct (K-1)-mer
 * unitig ends, HBV edges = every unitig and its reverse complement (palindromes once), ids assigned by the
 * reference's deterministic flood fill over the BVComp edge order.  Unitigs must be in BVComp order.
 * snk_hbv_from_unitigs takes host arrays in BVComp order.  snk_dev_hbv takes the device-resident unitigs of
 * snk_dev_count_graph (any order): BVComp ranking, the
 * (K-1)-mer end keys, their sort and the vertex classes are computed on the device (what the reference runs through
 * its MapReduce engine, HBVFromEdges.cc:136-168,257-262); the id hand-out, a breadth-first flood whose ids ARE the
 * visiting order (:170-238), runs per connected component: components and their id blocks on the device, one thread per
 * component; a component above SNK_HBV_BIG (1024) nodes, and any graph below SNK_HBV_DEV_MIN (65536) unitigs, on the host. */
typedef struct snk_hbv {
    int32_t n_vertices, n_edges;
    int32_t* v_left;            /* per HBV edge */
    int32_t* v_right;
    int32_t* src_unitig;        /* per HBV edge: unitig it is a copy (or reverse complement) of */
    uint8_t* is_rc;
    int32_t* fwd_xlat;          /* per unitig: HBV edge id of the forward / reverse-complement copy */
    int32_t* rev_xlat;
    int32_t* bvcomp_order;      /* snk_dev_hbv: per BVComp rank (the unitig numbering above) the index of that unitig
                                   in the caller's device arrays; NULL from snk_hbv_from_unitigs (input already ranked) */
} snk_hbv;
int snk_hbv_from_unitigs(uint32_t K, uint64_t n_unitigs, const uint64_t* unitig_off, const uint8_t* unitig_bases,
                         snk_hbv* out, char* err, size_t errcap);
/* device_ms (optional): time of the device part, HIP events */
int snk_dev_hbv(snk_ctx* ctx, uint32_t K, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases,
                snk_hbv* out, float* device_ms, void* stream, char* err, size_t errcap);
void snk_hbv_free(snk_hbv* h);
/* f1: read pathing on the device -- pathReads with the new aligner (paths/long/BuildReadQGraph48.cc:1441-1469):
 * Pather::path (:705-748), HBVPather::algorithmTwo (:1217-1336), pathPartsToReadPath (:1393-1428) and
 * ExtendReadPath::attemptLeftRightExtension (paths/long/ExtendReadPath.cc:108-358), over a k-mer -> (edge, offset) dictionary
 * built from the unitigs (:1656-1664).  Reads are pathed untrimmed (packed rows + quality rows + lengths of snk_dev_reads;
 * good_len / bc are not used).  The unitigs are the device arrays of snk_dev_count_graph, h their graph from snk_dev_hbv
 * (bvcomp_order maps its unitig numbering to the device arrays; NULL = the arrays are already in BVComp order).
 * Result (device memory of the context, valid until its next top-level call): per read the offset of the read on its first
 * edge (ReadPath::mOffset, may be negative), its number of edges and their HBV edge ids (start[r] .. start[r] + n_edges[r]). */
typedef struct snk_dev_paths {
    uint64_t n_reads, n_edges_total;
    const void* offset;        /* i32[n_reads] */
    const void* n_edges;       /* u32[n_reads] */
    const void* start;         /* u64[n_reads + 1] */
    const void* edges;         /* i32[n_edges_total] */
    uint64_t dict_slots;
    float dict_ms, path_ms;    /* HIP events: graph tables + packed unitigs + dictionary; pathing + gather */
    /* SNK_PATH_UNITIG_BCS (snk_dev_path_reads2): per unitig (device numbering) the sorted distinct barcodes > 0 of the reads that
     * have a k-mer on it -- the edge -> barcode sets of tada's MAIN_ASM_SN (lib/tada/src/cmd_main_asm.rs:91-151,
     * debruijn.rs:115-131), cut at 20 000 entries per unitig like cmd_main_asm.rs:115 -- the smallest ids are kept; the reference keeps
     * what its shard order delivers first, which needs its shard layout: parity unpinned (Rust) */
    const void* unitig_bc_off; /* u64[n_unitigs + 1] */
    const void* unitig_bcs;    /* u32[n_unitig_bcs] */
    uint64_t n_unitig_bcs;
    float bcs_ms;              /* HIP events: key sort + run heads + per-unitig lists (SNK_PATH_UNITIG_BCS) */
    uint32_t lookup_index;     /* 1: the look-ups went through the minimiser index (the k-mer dictionary did not fit, or SNK_PATH_INDEX=1); dict_slots then counts its places */
    uint64_t n_slow;           /* reads the fast pass left to the full algorithm (a miss, or an exact-match run that ended inside the read) */
} snk_dev_paths;
#define SNK_PATH_UNITIG_BCS 1u
#define SNK_PATH_UNITIG_BCS_EXHAUSTIVE 2u   /* (with SNK_PATH_UNITIG_BCS) derive the lists the slow, literal way -- every k-mer of every barcoded
                                               read looked up, barcodes_for_sedge (debruijn.rs:115-131) -- instead of from the path parts: a
                                               second, independent derivation the tests compare the fast one with */
#define SNK_PATH_UNITIG_BCS_NOCUT 4u        /* do not apply the 20 000-entry cut (cmd_main_asm.rs:115; here: a unitig keeps its 20 000 smallest ids) */
int snk_dev_path_reads(snk_ctx* ctx, uint32_t K, const snk_dev_reads* in, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases,
                       const snk_hbv* h, snk_dev_paths* out, void* stream, char* err, size_t errcap);
int snk_dev_path_reads2(snk_ctx* ctx, uint32_t K, const snk_dev_reads* in, uint64_t n_unitigs, const void* d_unitig_off, const void* d_unitig_bases,
                        const snk_hbv* h, uint32_t flags, snk_dev_paths* out, void* stream, char* err, size_t errcap);
/* f2: hbv.Involution (paths/HyperBasevector.cc:685-697; 10X/runstages/RunStages.cc:418): inv[e] = edge that is e's reverse
 * complement -- and the files DF keeps the graph in: a.hbv = BinaryWriter::writeFile(HyperBasevector)
 * (paths/HyperBasevector.cc:121-125, graph/DigraphTemplate.h:3092-3097) and a.inv (vec<int>), byte for byte.  The unitig arrays
 * are the ones the snk_hbv was built from (BVComp order); path_inv may be NULL. */
int snk_hbv_involution(const snk_hbv* h, uint64_t n_unitigs, int32_t* inv /* [n_edges] */, char* err, size_t errcap);
int snk_write_hbv(const char* path_hbv, const char* path_inv, uint32_t K, uint64_t n_unitigs, const uint64_t* unitig_off,
                  const uint8_t* unitig_bases, const snk_hbv* h, char* err, size_t errcap);

/* ---- f4: duplicate marking over the read paths --------------------------------------------------
 * MarkDups, lib/assembly/src/10X/SecretOps.cc:413-593 (10X/DF.cc:597-600 runs it on the paths just made and writes a.dup):
 * placed reads that share (first edge, offset on it, first five bases of the mate) are duplicates of each other; the one with
 * the largest quality sum over both mates -- the earliest read among equals -- survives, every other member flags its PAIR.
 *   in     the reads the paths were made from (untrimmed rows, quality rows, lens, bc: the raw barcode ids; NULL = all 0);
 *          reads 2q and 2q+1 are mates
 *   dup    u8[n_reads / 2] on the device (owned by the context, valid until its next call): vec<Bool> dup of the reference
 *   interdup_rate   duplicate reads whose group spans more than one barcode / duplicate reads (the reference's rule: a group's
 *          barcode is its first member's, or while that is 0 the next member's)
 *   n_art_pairs     pairs the reference counts as artifactual duplicates (identical bases AND qualities to another member of a
 *          group with a quality-sum tie); it only logs their percentage */
typedef struct snk_dev_dups {
    uint64_t n_pairs;
    const void* dup;
    uint64_t n_placed;            /* reads with a path */
    uint64_t n_dup_reads;         /* sum over groups of (size - 1) */
    uint64_t n_interdup_reads;
    uint64_t n_dup_pairs;         /* pairs flagged */
    uint64_t n_art_pairs;
    double interdup_rate;
    float ms;                     /* HIP events around the whole call */
} snk_dev_dups;
int snk_dev_mark_dups(snk_ctx* ctx, const snk_dev_reads* in, const snk_dev_paths* paths, snk_dev_dups* out, void* stream, char* err, size_t errcap);

/* ---- f3: barcode ids on the device --------------------------------------------------------------
 * BcIndexer, lib/tada/src/utils.rs:101-164: whitelist line -> index (identical lines: the last one wins); a read's
 * barcode field "SEQ[-gg][,raw]" (FASTH line 6, lib/tada/src/multifastq.rs:72-126) gets
 *     id = index(SEQ) + 1 + (gg - 1) * lines(whitelist),   gg = 1 without a "-gg" suffix,   0 = not on the whitelist.
 * snk_bc_index_create takes the bytes of the whitelist file (lines of at most 32 bytes, any characters);
 * snk_dev_bc_ids reads n_reads fields of `stride` bytes each (zero padded, resident in HBM) and writes int32 ids.
 * Errors follow the reference's panics: a gem group that is not a decimal u8 ("invalid gem group string", :138),
 * an id beyond 2^31 ("BC id overflowed", :157). */
typedef struct snk_bc_index snk_bc_index;
int snk_bc_index_create(snk_ctx* ctx, const char* whitelist, size_t bytes, snk_bc_index** out, char* err, size_t errcap);
void snk_bc_index_destroy(snk_bc_index* ix);
uint32_t snk_bc_index_lines(const snk_bc_index* ix);
int snk_dev_bc_ids(snk_ctx* ctx, const snk_bc_index* ix, const void* d_fields, uint32_t stride, uint64_t n_reads, void* d_ids,
                   void* stream, char* err, size_t errcap);

/* ---- stage-input formats of ASSEMBLER_DF (mro/_assembler_stages.mro:24-39) ----------- */
/* reads.fastb = feudal MasterVec<BaseVec>: 24-byte control block (feudal/FeudalControlBlock.h:157-166), the packed
 * bases (2 bits, base j at bits 2*(j%4), feudal/FieldVec.h:586-603), (N+1) u64 file offsets, N u32 lengths.
 * Output: packed rows in libsnk's layout (row_words = ceil(max_len/16)), lengths; malloc'ed, free() them. */
int snk_read_fastb(const char* path, uint64_t* n_reads, uint32_t* max_len, uint16_t** lens, uint32_t** rows, char* err,
                   size_t errcap);
/* reads.qualp = feudal MasterVec<PQVec>: per read a chain of byte-aligned blocks [nQs:8][nBits:3][minQ:6][nQs x nBits]
 * ended by a 0 byte (feudal/PQVec.cc:86-127).  quals: caller buffer n_reads*qstride (raw phred). */
int snk_read_qualp(const char* path, uint64_t n_reads, uint32_t qstride, uint8_t* quals, char* err, size_t errcap);
/* reads.bci = BINWRITE vec<int64_t>: bci[b]..bci[b+1] = read range of barcode ordinal b, b = 0 unbarcoded
 * (10X/ParseBarcodedFastqs.cc:284-293); expanded to one barcode id per read as DF does (10X/DF.cc:464-469). */
int snk_read_bci(const char* path, uint64_t n_reads, int32_t* bc_per_read, uint64_t* n_barcodes, char* err, size_t errcap);

/* The same three files decoded ON THE DEVICE, at the rate the bytes can be moved (supernova_amd/csrc/snk_dfin.hip): the reference loads
 * reads.fastb with bases.ReadAll and walks reads.qualp through VirtualMasterVec<PQVec> (10X/DF.cc:265-272,345,595-597; block codec
 * feudal/PQVec.cc:86-200; barcode index expansion DF.cc:464-469) -- offset-indexed, uncompressed files.  Here the raw byte ranges of a
 * slab of reads go from the page cache into a page-locked ring (a pool of `threads` pread workers, 0 = chosen from the CPU budget), up in
 * one copy per section, and three kernels make packed rows, quality rows, lengths and barcode ids of them; the host parses nothing but the
 * control blocks and the barcode index.
 *   snk_df_open       control blocks + the barcode index (bci may be NULL: no ids); nothing else is read
 *   snk_dev_ingest_df reads [first, first + n) -> resident device arrays as snk_dev_ingest_fasth's (release: snk_dev_ingest_free).  A rank
 *                     of the N-GPU job passes ITS range and touches only those bytes.  read_len = row length in bases, 0 = the longest
 *                     read of the file (one scan of its length table, on the device; snk_df_max_len does the scan for a range);
 *                     slab_reads: reads per slab, 0 = 262144
 *   snk_dev_ingest_df_count_graph   the slabs -> unitigs.  Default: a slab's quality rows are trimmed as soon as they are decoded and dropped;
 *                     packed rows, good lengths and barcode ids of the job stay (46 bytes per read, owned by the context) and the resident
 *                     step runs on them -- with its look at the first buckets, its second partition and its choice of count kernel, which
 *                     is what real (error-rich) reads need from the first call of a process on.  Option df_stream = 2: the slabs are
 *                     appended to a streamed job instead (snk_dev_stream_*: partitioned as they arrive, the reads never resident in any
 *                     form).  res as snk_dev_count_graph's, bit-identical either way.  stats: rows / quals / lens / bc stay NULL;
 *                     text_bytes = file bytes moved; n_files = 4 compact, 3 streamed.
 * Errors follow the host readers': SNK_E_IO for a bad offset / length / truncated quality block, SNK_E_ARG for a quality chain longer
 * than a row (the message names the first offending read). */
typedef struct snk_df_files snk_df_files;
typedef struct snk_df_info {
    uint64_t n_reads, n_barcodes;                 /* n_barcodes = entries of the index - 1 (ordinal 0 = unbarcoded), 0 without one */
    uint64_t fastb_bytes, qualp_bytes, bci_bytes;
    uint64_t reserved[3];
} snk_df_info;
struct snk_dev_ingest;
int snk_df_open(const char* fastb, const char* qualp, const char* bci, snk_df_files** out, snk_df_info* info, char* err, size_t errcap);
void snk_df_close(snk_df_files* f);
int snk_df_max_len(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t* max_len, char* err, size_t errcap);
int snk_dev_ingest_df(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads,
                      struct snk_dev_ingest* out, char* err, size_t errcap);
/* reads [first, first + n) in their compact form: packed rows, GOOD LENGTHS (the trim at K / min_qual runs on every slab as it is decoded, the
 * quality rows are never resident) and barcode ids: 46 instead of 204 bytes per 150-base read, all that count + graph needs (snk_dev_reads.good_len) */
int snk_dev_ingest_df_trimmed(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads, uint32_t K,
                              uint32_t min_qual, struct snk_dev_ingest* out, char* err, size_t errcap);
int snk_dev_ingest_df_count_graph(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads,
                                  const snk_params* p, int64_t ign_bc_below, snk_dev_result* res, struct snk_dev_ingest* stats, char* err, size_t errcap);
/* Writers of the triple (tests, bench.py's df_seam row, tools): feudal/FeudalFileWriter.cc:18-140, PQVec.cc:86-127,
 * BinaryWriter::writeFile(vec<int64_t>).  The reads must be ordered by barcode id (>= 0) -- the index holds one read range per ordinal
 * (10X/ParseBarcodedFastqs.cc:284-293).  The quality block choice is the writer's own (any chain of valid blocks decodes alike);
 * adversarial != 0 seeds random block cuts and wider-than-needed values, 1 = a block per value (decoder tests).  snk_synth_df_write: reads [first, first + n) of
 * the synthetic model (sp->unbarcoded_ppm must be 0); qual_jitter > 1 spreads the model's Q30 over [30, 30 + jitter) -- the same trim,
 * a quality file of realistic entropy. */
int snk_write_df(const char* head, uint64_t n, const uint32_t* rows, uint32_t row_words, const uint16_t* lens, uint32_t read_len, const uint8_t* quals,
                 uint32_t qstride, const int32_t* bc, uint32_t threads, uint64_t adversarial, char* err, size_t errcap);
int snk_synth_df_write(const char* head, const snk_synth_params* sp, uint64_t first, uint64_t n, uint32_t qual_jitter, uint32_t threads, char* err,
                       size_t errcap);

/* FASTH = the barcode-sorted read-pair text format of the tada stages (MultiFastqIter, lib/tada/src/multifastq.rs:69-127):
 * gzip, 9 lines per pair (header, R1, Q1, R2, Q2, barcode field, 3 ignored lines).  Returns malloc'ed arrays (free with
 * snk_host_free): ASCII bases and raw phred values in rows of `stride` bytes (read 2q = R1, 2q+1 = R2, cmd_msp.rs:160-181),
 * lengths, and one zero-padded 64-byte barcode field per PAIR (part before the first ',') for snk_dev_bc_ids. */
int snk_read_fasth(const char* path, uint32_t stride, uint64_t* n_reads, uint32_t* max_len, uint8_t** ascii, uint8_t** quals,
                   uint16_t** lens, uint8_t** bc_fields, char* err, size_t errcap);
void snk_host_free(void* p);

/* f3 at rate: many FASTH files decoded concurrently (the reference: one decode thread per two files, lib/tada/src/cmd_msp.rs:55-69,
 * over MultiFastqIter, multifastq.rs:69-127; like it, a file that is not gzip is refused).  A pool of `threads` workers
 * (0 = one per file up to the host's hardware threads) takes whole files, inflates with zlib's streaming API and parses the
 * records in place into batches of `batch_pairs` read pairs (0 = 32768); snk_fasth_next hands out full batches in any order
 * (file / first_pair say where a batch belongs), n_pairs == 0 = every file has been read to its end.  flags bit 0: the batches
 * live in page-locked memory (needs a GPU).  Rows of `stride` bytes: bases 'A'-padded, raw phred 0-padded. */
typedef struct snk_fasth_stream snk_fasth_stream;
typedef struct snk_fasth_batch {
    uint64_t n_pairs;           /* reads 2q, 2q+1 of the batch = R1, R2 of its pair q (cmd_msp.rs:160-181) */
    uint64_t first_pair;        /* index of the batch's first pair inside its file */
    uint32_t file, max_len;
    const uint8_t* ascii;       /* 2*n_pairs rows */
    const uint8_t* quals;
    const uint16_t* lens;       /* 2*n_pairs */
    const uint8_t* bc_fields;   /* n_pairs x 64 bytes, zero padded: the barcode field up to its first ',' */
    uint64_t text_bytes;        /* inflated bytes this batch was parsed from */
    uint64_t token;
} snk_fasth_batch;
/* CPUs this process may use: the cgroup CPU quota (cpu.max) when there is one, else the affinity mask / the hardware threads.  The
 * default number of decode threads is this minus two (threads = 0). */
uint32_t snk_host_cpu_budget(void);
int snk_fasth_open(const char* const* paths, uint32_t n_files, uint32_t stride, uint32_t batch_pairs, uint32_t threads, uint32_t flags,
                   snk_fasth_stream** out, char* err, size_t errcap);
int snk_fasth_next(snk_fasth_stream* s, snk_fasth_batch* out, char* err, size_t errcap);
void snk_fasth_release(snk_fasth_stream* s, snk_fasth_batch* b);      /* the batch's buffers go back to the workers */
uint64_t snk_fasth_file_pairs(snk_fasth_stream* s, uint32_t file);   /* known once the file has been read to its end */
void snk_fasth_close(snk_fasth_stream* s);
/* ... and into HBM: packed rows (snk_dev_pack_ascii), quality rows, lengths and -- with a whitelist index -- barcode ids
 * (snk_dev_bc_ids, one per read), in file-major order; uploads, pack and id lookup overlap the decode.  The arrays are plain
 * device allocations (not the context's arena: they are the INPUT of snk_dev_count_graph); snk_dev_ingest_free releases them. */
typedef struct snk_dev_ingest {
    uint64_t n_reads;
    uint32_t read_len, row_words, qstride, max_len;
    const void* rows;
    const void* quals;
    const void* lens;
    const void* bc;             /* NULL without an index */
    uint64_t text_bytes, compressed_bytes;
    double seconds, decode_wait_seconds;   /* whole call; of it, time the consumer waited for a decoded batch */
    uint32_t n_files, n_batches;
    double setup_seconds;                  /* of `seconds`: page-locked batch pool + device arrays allocated, workers started */
    const void* good_len;                  /* u16 per read: set by snk_dev_ingest_df_trimmed (quals and lens are NULL there), else NULL */
} snk_dev_ingest;
int snk_dev_ingest_fasth(snk_ctx* ctx, const char* const* paths, uint32_t n_files, uint32_t read_len, const snk_bc_index* ix, uint32_t threads,
                         uint32_t batch_pairs, snk_dev_ingest* out, char* err, size_t errcap);
void snk_dev_ingest_free(snk_dev_ingest* r);
/* FASTH files -> unitigs with the reads never resident as a whole: every decoded batch is uploaded, packed, given its barcode ids and
 * appended to a streamed job (snk_dev_stream_*) -- partitioned while the next batches are being inflated; the wall time is
 * max(ingest, partition) + count + graph.  res: as snk_dev_count_graph's (good_len in arrival order: batches arrive in any order).
 * total_reads_hint: an upper bound of the job's reads (sizes the bucket slots), 0 = from the compressed sizes.  stats: rows / quals /
 * lens / bc stay NULL.  The reading half + MSP of tada in one pass (lib/tada/src/cmd_msp.rs:55-69,100-190). */
int snk_dev_ingest_count_graph(snk_ctx* ctx, const char* const* paths, uint32_t n_files, uint32_t read_len, const snk_bc_index* ix, uint32_t threads,
                               uint32_t batch_pairs, uint64_t total_reads_hint, const snk_params* p, snk_dev_result* res, snk_dev_ingest* stats, char* err,
                               size_t errcap);
/* synthetic FASTH (tests, bench.py --ingest): pairs [first_pair, first_pair + n_pairs) of the synthetic read model as one gzip
 * file; barcode field = snk_synth_bc_seq(id) + "-1" (",raw" appended on every third pair), an off-whitelist sequence for id 0 */
int snk_synth_fasth_write(const char* path, const snk_synth_params* sp, uint64_t first_pair, uint64_t n_pairs, int level, uint64_t* text_bytes,
                          char* err, size_t errcap);
void snk_synth_bc_seq(uint32_t id, char* out16);

#ifdef __cplusplus
}
#endif
#endif /* SNK_H_ */
