// snk_kernels.h -- internal launch interfaces between the translation units of libsnk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snk_common.h"

// minimiser length (bases), by K.  A random-order minimiser starts a supermer every (K - M + 2) / 2 k-mers: the shorter the M-mer the fewer
// supermers -- slot reservations, records -- a read is cut into.  A record holds 107 bases (6 words + 23 bits): K=60 (two flanks + 59 + the
// k-mers of a supermer) has room for M=16 only.  SNK_M48: tuning builds (tools/build_variant.sh).
#ifndef SNK_M48
#define SNK_M48 16
#endif
#define SNK_M_OF(K) ((K) == 48 ? SNK_M48 : 16)
#ifndef SNK_M_LONG
#define SNK_M_LONG 20                    // SNK_F_LONG_MINIMISER (64-bit rolling window)
#endif
#define SNK_M_MIN_OF(K) (SNK_M_OF(K) < SNK_M_LONG ? SNK_M_OF(K) : SNK_M_LONG)
static_assert(SNK_M48 >= 11 && SNK_M48 <= 24, "the M-mer is rolled in one 32-bit (M <= 16) or 64-bit word; 4^M values must cover the buckets");

typedef unsigned __int128 snk_u128;

// ---- snk_msp.hip
size_t snk_msp_lds_bytes(uint32_t K, uint32_t M, uint32_t row_words);
#define SNK_MSP_LCAP 16
struct snk_msp_args {
    const uint32_t* rows;
    uint32_t row_words;
    uint32_t read_len;             // bases per row that are real (good lengths are clamped to it)
    const uint16_t* good_len;
    const int32_t* bc;
    const uint32_t* group;         // grouped runs: group id per read (then record word 7 = group), else NULL
    int64_t ign_bc_below;
    uint64_t read_index_base, n_reads;
    uint32_t NB;
    uint32_t* cursor;              // [NB] supermers of every bucket, counted from 0
    uint4* records;
    // bucket b owns records [b*cap, (b+1)*cap); what does not fit goes to [ovf_base, ovf_base+ovf_cap)
    uint32_t cap;
    uint32_t ovf_cap;              // slots of ONE overflow sub-list (there are SNK_OVF_SUBLISTS, chosen by wave)
    uint64_t ovf_base;
    uint32_t* ovf_bucket;          // [SNK_OVF_SUBLISTS * ovf_cap] bucket of every overflow record
    uint32_t* ovf_cursor;          // [SNK_OVF_SUBLISTS x SNK_OVF_CUR_STRIDE] overflow records wanted per sub-list (keep counting past ovf_cap)
    // buckets far beyond their capacity (a repeat family's or a homopolymer's minimiser: millions of supermers) stop reserving slots:
    // a lane that is handed slot >= hot_thr notes the bucket in hot_tab[bucket % SNK_MSP_HOT_TAB] (bucket + 1), workgroups copy the
    // table into LDS when they start, and a supermer of a noted bucket goes to the overflow list without touching the cursor --
    // same-address atomics are served one at a time, ~10 ns each.  cursor[b] is exact up to cap and a lower bound beyond.
    uint32_t* hot_tab;             // [SNK_MSP_HOT_TAB], zeroed; NULL = every supermer takes its reservation
    uint32_t hot_thr;
    uint32_t dbg;                  // profiling aid (results invalid): 1 = no record stores, 2 = no slot atomics
    // fused quality trim (quals != NULL): the kernel derives every read's good length itself (the rule of snk_trim.hip), writes
    // it to good_out and adds the k-mer instances / contributing reads of its waves to plan[2 * (wave % 256) + {0, 1}]
    const uint8_t* quals;
    uint32_t qstride, min_qual;
    const uint16_t* lens;
    uint16_t* good_out;
    unsigned long long* plan;
    // dense partition (dense_bkt != NULL): no per-bucket slots and no slot reservation -- record p of the pass goes to records[p]
    // (a workgroup reserves its block with one atomic on dense_cursor), its bucket to dense_bkt[p]; cursor / cap / ovf_* are unused
    uint32_t* dense_bkt;
    unsigned long long* dense_cursor;   // [1] records wanted (keeps counting past dense_cap)
    uint64_t dense_cap;
    // bucket-range passes (b_hi != 0): only the supermers of buckets [b_lo, b_hi) are emitted, bucket b owns records
    // [(b - b_lo) * cap, (b - b_lo + 1) * cap) -- a job whose slots do not fit the device is partitioned and counted range by range
    // over the same slot memory, the reads scanned once per pass (what the reference's MapReduce engine does when its records do not
    // fit: lib/assembly/src/MapReduceEngine.h:452-468)
    uint32_t b_lo, b_hi;
};
constexpr int SNK_MSP_PLAN_SLOTS = 256;
constexpr uint32_t SNK_OVF_SUBLISTS = 64;
// the sub-lists' cursors lie 128 bytes apart: atomics on one 64-byte line are served one at a time whatever word they address (64 cursors in
// four lines: 25 M reservations of a repeat-rich genome took 63 ms, a quarter of what ONE cursor took)
constexpr uint32_t SNK_OVF_CUR_STRIDE = 32;
constexpr uint32_t SNK_MSP_HOT_TAB = 256;
int snk_launch_msp(uint32_t K, uint32_t mlen, hipStream_t st, const snk_msp_args& a, char* err, size_t errcap);
int snk_launch_msp_plan(hipStream_t st, const uint16_t* good_len, uint64_t n_reads, uint32_t K, unsigned long long* out2,
                        char* err, size_t errcap);

// ---- snk_count.hip
// the most k-mers one chunk of the bucket-local graph stage holds (snk_local.hip: the big-chunk kernels' LDS); a count sub-pass that retains
// more is counted again in two halves
constexpr uint32_t SNK_GRAPH_CHUNK_MAX = 1280;
struct snk_count_args {
    const uint4* records;          // supermer records, 2 x uint4 each
    const uint64_t* seg_beg;       // [nseg][seg_stride] first record of every bucket in every segment (absolute)
    const uint64_t* seg_end;       // [nseg][seg_stride] one past its last record (offset tables: seg_end = seg_beg + 1)
    uint32_t seg_stride;
    uint32_t nseg;
    const uint2* vmeta;            // virtual buckets (snk_hot.hip): per bucket (real bucket, split_lg << 24 | split_id) the pass starts from; else NULL
    const uint32_t* gidx;          // dense partition: the segment bounds index this list, record v of a bucket is records[gidx[v]]; else NULL
    uint32_t NB;
    uint32_t min_freq;
    uint32_t bc_mode;              // = minBC: 0 no barcode rule, 1: >=1 barcode>0 (or ignore-rule read), 2: >=2 distinct (state machine), 3..8: id sets
    uint32_t bucket0;              // first bucket of this launch (set by the launcher)
    uint32_t bucket_stride;        // set by the launcher (= n_regions): workgroup w counts the buckets == w (mod stride)
    uint32_t screen;               // per-barcode groups: instances pass a three-level bit filter first (level = min(min_freq, 3); 0 / 1 = off; snk_count.hip SCREEN)
    uint32_t tight;                // the table fills to 7/8: waves book their slots (snk_count.hip, TIGHT)
    uint32_t grouped;              // record word 7 is a group id that becomes the low 32 bits of the key (K=48 only)
    snk_u128* out_keys;            // canonical k-mer values (hi<<64|lo); region r owns [r*region_cap, (r+1)*region_cap):
                                   // one region per workgroup of the (persistent) launch, bucket b goes to region b % n_regions
    uint64_t* out_vals;            // count << 8 | raw context byte
    uint64_t region_cap;
    uint32_t n_regions;
    unsigned long long* region_cursor;   // [n_regions] entries used per region
    uint32_t* chunk_n;             // [NB] survivors of an unsplit bucket (0 when it split or has none), or NULL
    uint32_t* chunk_base;          // [NB] their offset inside region (bucket % n_regions)
    uint4* extra;                  // sub-passes of split buckets: (bucket, offset, n | region << 12, split_lg << 24 | split_id); count in status[4]
    uint32_t extra_cap;
    unsigned long long* prof;      // SNK_COUNT_PROF builds: [8] clock cycles of thread 0 per phase, summed over workgroups
    uint32_t dbg;                  // profiling aid: 1 = roll+hash only, 2 = no updates after the probe (results invalid)
    uint32_t* status;              // [0] output overflow, [1] split depth exceeded, [2] buckets split, [3] max slots used, [4] extra chunks
};
int snk_launch_count(uint32_t K, hipStream_t st, const snk_count_args& a, char* err, size_t errcap);
uint32_t snk_count_slots(uint32_t K);   // LDS table slots per workgroup
// output regions (= workgroups of every launch) for a table over NB buckets
int snk_count_regions(uint32_t K, uint32_t grouped, uint32_t nseg, uint32_t NB, uint32_t bc_mode, uint32_t* n_regions, char* err, size_t errcap);
// gather the used prefix of every region into one dense table; region_off = exclusive scan of region_cursor
int snk_launch_compact_regions(hipStream_t st, const snk_u128* keys_in, const uint64_t* vals_in, uint64_t region_cap,
                               uint32_t n_regions, const unsigned long long* region_cursor,
                               const unsigned long long* region_off, snk_u128* keys_out, uint64_t* vals_out, char* err,
                               size_t errcap);
