// snk_comm.h -- library-internal transport of the minimiser-sharded path (snk_shard_step.hip): the few collectives the step
// needs, over RCCL (one process per GPU, xGMI) or between in-process ranks on ONE device (tests: every rank is a host
// thread with its own context, the "wire" is a device copy).  SURVEY.md 8(e); the reference's exchange is a set of shard
// files (lib/tada/external/rust-shardio/src/shard.rs:184-211,488-493) and the in-memory swizzle of MapReduceEngine.h:362-385.
#pragma once
#include "snk_ctx.h"

struct snk_comm {
    uint32_t rank = 0, world = 1;
    uint64_t bytes_sent = 0;          // payload handed to the transport for OTHER ranks since the last reset
    uint64_t n_collectives = 0;
    // Sizing history of the step, kept with the GROUP and not with a rank's context: distinct k-mers per instance the last step over
    // this communicator saw (job-wide, computed by every rank from the same exchanged words).  Every rank of a group has been through
    // the same steps over it, so the decision "partition with the ratio / run a pilot first" -- which decides whether a collective is
    // issued -- is the same everywhere by construction; a context's own history (a one-rank step, a re-created engine) cannot split it.
    double claim_ratio = 0.0;
    uint64_t claim_ratio_reads = 0;
    uint32_t claim_ratio_k = 0;
    virtual ~snk_comm() {}
    virtual const char* kind() const = 0;
    // Variable all-to-all of bytes: scnt[p] bytes at send + sbeg[p] go to rank p, rcnt[s] bytes from rank s land at
    // recv + rbeg[s] (host arrays of `world` entries, known on both sides: rcnt[s] here == scnt[rank] on rank s).
    // Stream-ordered on `st` where the transport allows it.
    virtual int a2a(const void* send, const uint64_t* sbeg, const uint64_t* scnt, void* recv, const uint64_t* rbeg, const uint64_t* rcnt,
                    hipStream_t st, char* err, size_t errcap) = 0;
    // all-gather with per-rank byte counts known to everybody; recv = the ranks' buffers in rank order
    virtual int allgatherv(const void* send, const uint64_t* counts, void* recv, hipStream_t st, char* err, size_t errcap) = 0;
    // k u64 values per rank, resident on the DEVICE (d_mine), arrive on the HOST of every rank: h_all[world * k].  This is the
    // one place where the step learns sizes it could not know: it waits for the stream (one host read-back).
    virtual int gather_counts(const unsigned long long* d_mine, uint32_t k, unsigned long long* h_all, hipStream_t st, char* err, size_t errcap) = 0;
    virtual int barrier(hipStream_t st, char* err, size_t errcap) = 0;
    virtual void abort() {}            // a rank failed: release the others (in-process ranks only)
};

void snk_plan_range_pieces(const unsigned long long* h_rs, uint32_t W, uint32_t R, uint32_t r, uint64_t item_bytes, uint64_t* sbeg, uint64_t* scnt,
                           uint64_t* rbeg, uint64_t* rcnt);
