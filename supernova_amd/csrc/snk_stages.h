// snk_stages.h -- internal stage interfaces shared by snk_pipeline.hip and snk_dist.hip.
#pragma once
#include <vector>

#include "snk_ctx.h"
#include "snk_kernels.h"

struct snk_phase_timer {
    hipStream_t st;
    hipEvent_t ev[16];
    int n = 0;
    bool ok = true;
    explicit snk_phase_timer(hipStream_t s) : st(s) {
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) ok = false;
    }
    ~snk_phase_timer() { for (auto& e : ev) (void)hipEventDestroy(e); }
    void mark() { if (ok && n < 16) (void)hipEventRecord(ev[n++], st); }
    float ms(int a, int b) {
        float t = 0;
        if (!ok || a >= n || b >= n) return 0;
        (void)hipEventSynchronize(ev[b]);
        (void)hipEventElapsedTime(&t, ev[a], ev[b]);
        return t;
    }
};

struct snk_table {
    uint64_t n;
    snk_u128* keys;      // sorted ascending (sorted == true) or in chunk order
    uint64_t* vals;      // count << 8 | raw context
    uint32_t buckets_split, max_slots_used;
    uint64_t distinct;   // distinct k-mers the count kernel's tables held, summed over the buckets (before the filter)
    float count_ms, sort_ms, count_kernel_ms;
    // chunk order (sorted == false): the survivors of every count sub-pass are contiguous; chunk c < NB is bucket c
    // (unsplit), chunk NB + e is extra[e] (a sub-pass of a split bucket).  Dense position of a chunk =
    // region_off[bucket % n_regions] + offset.
    bool sorted;
    uint32_t NB, n_regions, n_extra;
    const uint32_t* chunk_n;
    const uint32_t* chunk_base;
    const uint4* extra;
    const unsigned long long* region_off;
    // deferred compaction (chunk order only): keys points at dense, still UNWRITTEN memory and vals is NULL; the survivors are where the
    // count kernel put them -- region r at [r * region_cap, + region_cursor[r]) of keys_r / vals_r -- until the bucket-local prune,
    // which reads every chunk once anyway, writes the keys densely (snk_local.hip).  keys_r == NULL: the table is dense.
    const snk_u128* keys_r;
    const uint64_t* vals_r;
    uint64_t region_cap;
};

struct snk_hot;
// optional: the first count launch goes out in bucket ranges [bounds[r], bounds[r+1]); ready(user, r) is called before
// range r is launched (the sharded path makes the stream wait for that range's records there)
struct snk_count_ranges {
    uint32_t n;
    const uint32_t* bounds;
    int (*ready)(void* user, uint32_t r);
    void* user;
    // replay: the hook MAKES the records of range r (bucket-range passes: snk_partition_passes) -- a run that has to be repeated (count
    // regions too small) calls it again for every range, and its return code and message are the stage's
    bool replay = false;
    // hot minimiser buckets the hook found and expanded in its ranges (bucket-range passes): counted behind the ranged launches, like `hot`
    const std::vector<snk_hot>* hots = nullptr;
    // the counting is over and will not be repeated (the regions held everything): the hook's record slots can go back to the arena before the
    // dense table is asked for -- a job in passes is short of exactly that memory
    void (*finished)(void* user) = nullptr;
};
// pilot: the first 1/64 of the buckets is counted first; if their tables overflow as a rule (more distinct k-mers than the LDS
// table holds: error-rich reads, shallow coverage) the stage stops there, leaves the distinct k-mers per bucket it saw in
// pilot->per_bucket and returns SNK_RETARGET -- the caller partitions again into smaller buckets instead of hash-splitting nearly
// every bucket.  agree (sharded step): makes per_bucket the job-wide figure (a collective), so that every rank decides alike.
constexpr int SNK_RETARGET = 1000;
struct snk_count_pilot {
    double per_bucket;
    int (*agree)(void* user, double* per_bucket);
    void* user;
};
uint32_t snk_count_limit(uint32_t K, uint32_t grouped, uint32_t tight);
uint32_t snk_count_screen_limit();      // distinct k-mers one pass over a bucket may hold
int snk_stage_count_table(snk_ctx* ctx, hipStream_t st, uint32_t K, const void* records, const uint64_t* seg_beg,
                          const uint64_t* seg_end, uint32_t seg_stride, uint32_t nseg, uint32_t NB, uint32_t min_freq, uint32_t bc_mode, uint32_t grouped, uint64_t n_inst_hint,
                          uint32_t* status, bool want_sort, snk_table* out, char* err, size_t errcap,
                          const snk_count_ranges* ranges = nullptr, snk_count_pilot* pilot = nullptr, const uint32_t* gidx = nullptr,
                          bool defer_compact = false, const snk_hot* hot = nullptr);

// ---- minimiser partition in one pass (fixed bucket capacity + overflow segment)
struct snk_partition {
    uint32_t NB, cap, nseg, n_overflow;
    uint64_t n_supermers;
    void* records;        // bucket b owns [b*cap, (b+1)*cap); overflow records (grouped by bucket) behind them
    uint32_t* cursor;     // [NB] supermers of every bucket (including the overflowed ones)
    uint64_t* seg;        // [begin seg 0 | end seg 0 | begin seg 1 | end seg 1] x NB absolute record offsets
    float kernel_ms;
    // dense partition (snk_stage_partition with allow_dense): records in read order, gidx = their positions sorted by bucket; seg then
    // bounds ranges of gidx, cap / cursor / the overflow segment do not exist
    const uint32_t* gidx;
    float sort_ms;
};
int snk_stage_partition_plan(snk_ctx* ctx, hipStream_t st, uint32_t K, const uint16_t* good_len, uint64_t n_reads,
                             unsigned long long h_plan[2] /* instances, contributing reads */, char* err, size_t errcap,
                             unsigned long long** d_plan_out = nullptr /* given: no read-back, the device counters are returned */);
// The quality trim inside the partition kernel: no trim kernel, no plan kernel; n_inst / n_live are then the caller's upper bounds
// (all bases of all reads) and the exact figures come back in h_plan with the pass's read-back.
struct snk_fused_trim {
    const void* quals; uint32_t qstride; const void* lens; uint32_t min_qual;
    uint16_t* good_out;            // [n_reads] the good lengths, as snk_dev_trim writes them
};
bool snk_fused_trim_ok(const snk_dev_reads* in);
int snk_stage_partition(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_dev_reads* in, const uint16_t* good_len, uint32_t NB,
                        unsigned long long n_inst, unsigned long long n_live, bool grouped, uint32_t* status, snk_partition* out,
                        char* err, size_t errcap, const unsigned long long* d_plan = nullptr, unsigned long long* h_plan = nullptr,
                        const snk_fused_trim* ft = nullptr, bool allow_dense = false);
// ---- the partition in bucket-range passes over ONE slot array sized for a range (a job whose slots do not fit the device; the
// reference re-scans its input in passes when its records do not fit: MapReduceEngine.h:452-468, lib/tada/src/utils.rs:329-341).
// open sizes and allocates; snk_partition_passes_run(user = the object, r) is the count stage's range hook: it scans the reads once
// more and emits the supermers of range r only (the quality trim runs in the first pass), then builds that range's segment tables.
struct snk_partition_passes {
    snk_ctx* ctx; hipStream_t st; uint32_t K, NB, cap, P; bool grouped;
    snk_dev_reads in; const uint16_t* good_len; snk_fused_trim ft; bool fused;
    uint32_t* cursor; uint64_t* seg; void* records; uint32_t* ovf_bucket; uint32_t* ovf_cur; unsigned long long* d_total; unsigned long long* d_fplan;
    uint64_t ovf_cap, slots_per_pass;
    uint32_t bounds[66];
    unsigned long long h_plan[2]; uint64_t n_supermers, n_overflow; float kernel_ms; uint32_t runs;
    std::vector<snk_hot>* hots; uint32_t n_hot;          // the passes' hot buckets, expanded while their records were there (owned by the caller)
    char* err; size_t errcap;
};
int snk_partition_passes_open(snk_ctx* ctx, hipStream_t st, uint32_t K, const snk_dev_reads* in, const uint16_t* good_len, const snk_fused_trim* ft, uint32_t NB,
                              uint32_t passes, unsigned long long n_inst, unsigned long long n_live, bool grouped, snk_partition_passes* S, char* err, size_t errcap);
int snk_partition_passes_run(void* user, uint32_t r);
// passes a job of this size needs so that its slots take at most ~a quarter of the device (1: the one-pass partition)
uint32_t snk_partition_passes_needed(snk_ctx* ctx, uint32_t K, uint32_t NB, unsigned long long n_inst, unsigned long long n_live, bool grouped);
#ifdef SNK_PROBES
int snk_probe_relaunch_msp(snk_ctx* ctx, hipStream_t s2, uint32_t dbg, char* err, size_t errcap);      // measurement aid (tools/overlap_probe*.py)
#endif
// hot minimiser buckets (snk_hot.hip): their records expanded into single-k-mer records, one virtual bucket per (bucket, hash class)
struct snk_hot {
    uint32_t n_hot, NBv;           // hot buckets, virtual buckets (0: nothing is hot)
    uint64_t n_records, n_instances;
    const void* records;           // n_instances single-k-mer records, virtual-bucket-major
    const uint64_t* seg;           // [begin NBv | end NBv]
    const uint2* vmeta;            // [NBv] (real bucket, split_lg << 24 | split_id)
    void* plan;                    // between snk_stage_hot_plan and snk_stage_hot_expand: what the expansion needs (host object)
    // the records are expanded when the count stage gets to them (records == NULL until then): the hook makes the stream wait for what
    // the expansion reads -- a rank of the N-GPU job plans from the exchanged histograms while the records are still on their way
    int (*before_expand)(void* user);
    void* user;
    const void* src_records;       // base of the record array the segment table indexes
};
// after snk_stage_partition: finds the buckets far above their capacity, takes them out of the partition's segment table and builds
// their virtual buckets; snk_stage_count_table counts those in a second launch
int snk_stage_hot(snk_ctx* ctx, hipStream_t st, uint32_t K, bool grouped, snk_partition* part, snk_hot* hot, char* err, size_t errcap);
// the two halves: the plan from a segment table (beg/end [s * stride + b], nseg segments per bucket; the hot buckets are emptied in
// it), the expansion from the records it indexes
int snk_stage_hot_plan(snk_ctx* ctx, hipStream_t st, uint32_t K, bool grouped, uint64_t* seg_beg, uint64_t* seg_end, uint32_t stride, uint32_t nseg, uint32_t NB,
                       uint32_t cap, snk_hot* hot, char* err, size_t errcap);
int snk_stage_hot_expand(snk_ctx* ctx, hipStream_t st, const void* records, snk_hot* hot, char* err, size_t errcap);
void snk_stage_hot_drop(snk_hot* hot);
// the same pass as a job that takes its reads slab by slab (snk_dev_stream_*)
struct snk_partition_job {
    uint32_t K, NB, cap, n_slabs;
    uint64_t ovf_cap, n_reads;
    bool grouped;
    uint32_t* cursor;
    uint64_t* seg;
    unsigned long long *d_total, *d_plan;     // d_plan: SNK_MSP_PLAN_SLOTS x (instances, contributing reads), summed at close
    void* records;
    uint32_t* ovf_bucket;
    uint32_t* ovf_cur;        // [SNK_OVF_SUBLISTS] cursors of the overflow sub-lists
    uint32_t* status;
};
int snk_partition_open(snk_ctx* ctx, hipStream_t st, uint32_t K, uint32_t NB, unsigned long long n_inst_ub, unsigned long long n_live_ub, bool grouped,
                       uint32_t* status, snk_partition_job* J, char* err, size_t errcap);
int snk_partition_add(snk_ctx* ctx, hipStream_t st, snk_partition_job* J, const snk_dev_reads* in, const uint16_t* good_len, const snk_fused_trim* ft, char* err,
                      size_t errcap);
int snk_partition_close(snk_ctx* ctx, hipStream_t st, snk_partition_job* J, snk_partition* out, unsigned long long h_plan[2], char* err, size_t errcap);
// sharded runs: the buckets' records copied to exact offsets (u32 record index per bucket) of a compact buffer
int snk_stage_partition_compact(snk_ctx* ctx, hipStream_t st, const snk_partition* part, const uint32_t* d_offsets, void* d_out, char* err, size_t errcap);
int snk_stage_partition_compact_remote(snk_ctx* ctx, hipStream_t st, const snk_partition* part, const uint32_t* d_offsets, void* d_out,
                                       uint32_t skip_lo, uint32_t skip_hi, char* err, size_t errcap);
