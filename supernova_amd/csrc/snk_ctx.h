// snk_ctx.h -- library-internal context: device, streams, error reporting, scratch arena.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/snk.h"
#include "snk_opts.h"

struct snk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // library-owned default stream
    hipStream_t cur_stream = nullptr;   // the stream of the top-level call at hand (SNK_ARENA_POISON=2 fills handed-back blocks in its order)
    int n_cu = 256;
    size_t lds_per_block = 65536;
    uint64_t device_mem_total = 0;   // HBM of the device (sizing decisions that must not depend on what happens to be free)
    double pass_sigma = 0.0;         // slot capacity (sigmas of the occupancy model) of a job in bucket-range passes: the largest that does not cost a pass (snk_partition_passes_needed); 0 = not planned
    uint64_t plan_mapped = 0;        // what the arena held when the call began: memory a plan can use without asking the device for more (see snk_stages.hip, partition_capacity)
    uint64_t plan_mem = 0;           // ... minus what is not this context's to use -- the caller's reads, other contexts -- in whole 8-GB steps, looked at
                                     //     at the start of every top-level call (snk_ctx_release_scratch): what the slot / pass / region plans of a large job divide
    // caching arena for call-scoped scratch: blocks are handed out by best fit, returned to the cache at the
    // start of the next top-level call (no hipFree/hipMalloc in steady state), released on destroy or OOM
    struct block { void* p; size_t bytes; bool used; uint64_t serial = 0; uint64_t epoch = 0; };
    uint64_t call_epoch = 0;     // top-level calls so far; a cached block no call has taken for two of them is given back to the device
    uint64_t alloc_serial = 0;   // blocks handed out so far (a call's internal scratch = the blocks with a larger serial than at its entry)
    std::vector<block> blocks;
    // Round 4 (SNK_ARENA_VMM=0 switches it off): the arena as ONE growing range of virtual addresses (hipMemAddressReserve) that physical memory is mapped behind
    // on demand (hipMemCreate / hipMemMap): a request that does not fit maps more at the end -- no hipFree, no hipMalloc of a block of
    // another size when the bucket count changes (re-allocating memory the process has freed costs ~30 ms per GB on this stack:
    // tools/probe/vmm.hip; a call that changed its block sizes took 2.7-6.8 s).  Ranges are handed out first-fit and coalesce when
    // they come back; everything is free again at the start of a top-level call.  `blocks` stays as the fallback (VMM refused, the
    // range exhausted, SNK_ARENA_VMM=0, or a multi-rank RCCL step, whose buffers stay plain hipMalloc memory).
    struct vrange { size_t off, bytes; };
    char* va_base = nullptr;
    size_t va_size = 0, va_mapped = 0, va_high = 0, va_high_prev[2] = {0, 0};
    std::vector<hipMemGenericAllocationHandle_t> va_handles;
    std::vector<size_t> va_chunk;           // bytes of every mapped chunk (they follow each other from va_base)
    std::vector<vrange> va_free;            // sorted by offset, coalesced
    std::vector<vrange> va_used;
    size_t va_floor = 0;                    // snk_ctx_reserve: this much stays mapped whatever the last calls used
    int va_state = 0;                       // 0 untried, 1 in use, -1 off
    bool va_sealed = false;                 // chunks were unmapped behind live ranges: no growth until the reservation is replaced
    struct vser { size_t off; uint64_t serial; };
    std::vector<vser> va_serial;            // allocation order of the live ranges (snk_ctx_release_since)
    bool arena_legacy = false;              // this call's scratch comes from plain hipMalloc blocks (multi-rank RCCL steps)
    size_t total_alloc = 0;     // bytes handed out in the current call (minus blocks returned mid-call)
    size_t peak_alloc = 0;      // its maximum during the call
    size_t cached_bytes = 0;    // bytes held by the arena
    uint64_t last_n_kmers = 0, last_n_instances = 0;   // sizing hint from the previous call
    uint64_t last_region_max = 0, last_region_n = 0;   // ... and its fullest count region (with that many regions): survivors that cluster -- duplicated reads of one barcode -- fill regions unevenly
    uint64_t last_bnd = 0, last_bnd_n = 0;              // boundary k-mers the bucket-local prune found for a table of last_bnd_n k-mers
    uint32_t last_ovf = 0, last_ovf_nb = 0;             // overflow supermers of the last partition pass and its bucket count
    uint64_t last_ovf_reads = 0;
    uint64_t last_dense = 0;                            // supermer records of the last dense partition pass (last_ovf_nb == 0xD0000000)
    double retain_ratio = 0.0;                         // retained k-mers per k-mer instance of the last call (same key as claim_ratio)
    uint32_t count_screen = 0;                         // ungrouped call: the count launches run their instances through the bit filter first (level; snk_count.hip SCREEN)
    uint32_t count_tight = 0;                          // this call's count launches book their table slots (error-rich data, per-barcode groups: fuller tables, fewer buckets)
    double screen_ratio = 0.0;                         // the distinct-per-instance ratio an ungrouped screened call was decided on
    double claim_ratio = 0.0;                          // distinct k-mers per k-mer instance the count kernel saw in the last call ...
    uint64_t claim_ratio_reads = 0;                    // ... over this many reads ...
    uint32_t claim_ratio_k = 0;                        // ... in this mode (2 K + grouped + 256 x minimiser length)
    uint32_t mlen = 16;                                // minimiser length of the top-level call at hand (snk_set_mlen)
    uint32_t last_extra = 0;                           // split sub-passes the previous call recorded
    uint64_t last_input_fp = 0;                        // fingerprint of the last resident call's reads (snk_pipeline.hip): other data of the same size must not inherit its sizing history
    bool have_input_fp = false;
    uint32_t last_count_limit = 0;                     // usable table slots of the last resident call's count launches (snk_ctx_last_count_limit)
    uint32_t last_partition_passes = 1;                // bucket-range passes of the last resident call (snk_ctx_last_partition_passes)
    std::vector<unsigned long long> h_region_off;      // host copy of the count regions' dense offsets (source of an async upload)
    snk_opts opts;              // what the host pinned (snk_ctx_set_tuning / snk_ctx_set_option / SNK_TUNING at creation); all clear = the library's own choices
    void* shard = nullptr;      // snk_shard_state (snk_dist.hip)
    void* host_io = nullptr;    // pinned staging + device input buffers of the host-pointer entry point (snk_host.hip)
    void (*host_io_free)(void*) = nullptr;
    void* df_io = nullptr;      // page-locked ring + device mirror of the DF-seam ingest (snk_dfin.hip)
    void (*df_io_free)(void*) = nullptr;
    void* stream_job = nullptr; // the open streamed job of snk_dev_stream_* (snk_pipeline.hip)
    void (*stream_job_free)(void*) = nullptr;
    void (*stream_job_invalidate)(void*) = nullptr;   // the arena the open job lives in is being recycled: append / finish must fail from now on
    void* shard_host = nullptr; // pinned staging, exchange stream and events of snk_shard_step (snk_shard_step.hip)
    void (*shard_host_free)(void*) = nullptr;
};

// every top-level entry point: the context's device is current and its options are the calling thread's
hipError_t snk_enter(snk_ctx* ctx);
void snk_set_error(char* err, size_t errcap, const char* fmt, ...);
void snk_set_mlen(snk_ctx* ctx, const snk_params* p);      // ctx->mlen from the call's K and flags (SNK_MINIMISER_LEN=16|20 overrides: tests)
// every host wait for a stream goes through here: the calling thread's count is what the sharded step reports as
// host_syncs (a host thread = a rank)
hipError_t snk_sync_at(hipStream_t st, const char* file, int line);     // SNK_SYNC_TRACE=1: every wait is logged with its site
uint64_t snk_sync_count();
#define snk_sync(st) snk_sync_at((st), __FILE__, __LINE__)
int snk_fail(int code, char* err, size_t errcap, const char* fmt, ...);

#define SNK_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap,      \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// scratch allocation owned by the ctx; freed by snk_ctx_release_scratch / destroy
int snk_ctx_alloc(snk_ctx* ctx, size_t bytes, void** out, char* err, size_t errcap);
void snk_ctx_release_scratch(snk_ctx* ctx);   // return every block to the cache
void snk_ctx_release_block(snk_ctx* ctx, const void* p);   // return one block to the cache (no-op for unknown pointers)
// return every block handed out after `mark` (= ctx->alloc_serial at the call's entry) except the ones in keep[0..n_keep)
void snk_ctx_release_since(snk_ctx* ctx, uint64_t mark, const void* const* keep, size_t n_keep);
void snk_ctx_trim_cache(snk_ctx* ctx);        // hipFree every unused cached block
void snk_ctx_plan_mem(snk_ctx* ctx);          // ctx->plan_mem from the device's free memory + what the arena already holds (call it with the arena released)
void snk_shard_state_free(void* p);
void snk_shard_state_invalidate_job(void* p);   // an open streamed step dies with the arena it lives in (snk_ctx_release_scratch)
