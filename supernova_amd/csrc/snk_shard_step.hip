// snk_shard_step.hip -- one step of the minimiser-sharded (multi-GPU) count+graph path, entirely behind the C ABI:
// a C++ host (supernova_amd/csrc/host/snk_asm_sn.cc) or any other caller hands in its slab of reads and a communicator
// (snk_comm.hip: RCCL over xGMI, or in-process ranks for tests) and gets its share of the table and its unitigs back.
//
// What it replaces: tada's MSP -> SHARD_ASM -> MAIN_ASM_SN chain with its shard-file exchange
// (lib/tada/src/cmd_msp.rs:38-80, cmd_shard_asm.rs:37-94, cmd_main_asm.rs:25-89;
// lib/tada/external/rust-shardio/src/shard.rs:184-211,488-493) and the in-memory swizzle of MapReduceEngine.h:362-385.
//
//   partition (all NB_total buckets, one pass) -> [histograms] -> compaction of the buckets other ranks own
//   -> [supermer records, in bucket ranges] -> count (range r while range r+1 is on the wire; the rank's own buckets are
//   counted IN PLACE from the partition's slots: no copy, no compaction) -> bucket-local prune -> [membership queries /
//   answers] -> fragments -> [link queries / answers] -> owner-side join: [link structure] -> ranking (partitioned:
//   [splitters], [ranks to owners]) -> placement -> [fragments to the owners of their unitigs] -> unitigs.
// [..] = an exchange.  The host learns sizes through snk_comm::gather_counts only (device counters of all ranks -> host).
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <vector>

#include "snk_comm.h"
#include "snk_common.h"
#include "snk_shard.h"

// the stage entry points of snk_dist.hip (declared in include/snk.h)

namespace {

typedef unsigned long long ull;

__global__ void __launch_bounds__(256) range_sum_kernel(const uint32_t* __restrict__ h, uint32_t NBl, uint32_t R, ull* __restrict__ out) {
    // block (row, r): sum of h[row*NBl + bounds(r) .. bounds(r+1)), bounds(q) = NBl*q/R
    const uint32_t row = blockIdx.x / R, r = blockIdx.x % R;
    const uint64_t lo = (uint64_t)NBl * r / R, hi = (uint64_t)NBl * (r + 1) / R;
    ull v = 0;
    for (uint64_t b = lo + threadIdx.x; b < hi; b += 256) v += h[(uint64_t)row * NBl + b];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __shared__ ull part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// segment tables of the count kernel on a rank of a W > 1 job: T[0..nseg) x NBl starts, then nseg x NBl ends (record indices
// relative to the RECEIVE buffer; the rank's own records stay in the partition's slot array, `delta` records away -- the
// arithmetic wraps, the kernel adds the index to the buffer's address)
__global__ void __launch_bounds__(256) seg_table_kernel(const ull* __restrict__ roff /* [W*NBl] */, const uint32_t* __restrict__ hrecv /* [W][NBl], row me = 0 */,
                                                        const uint32_t* __restrict__ cursor /* [NB_total] */, const uint64_t* __restrict__ pseg /* part.seg */,
                                                        uint32_t W, uint32_t me, uint32_t NBl, uint32_t NB_total, uint32_t cap, uint32_t nseg, ull delta,
                                                        uint64_t* __restrict__ T) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= NBl) return;
    uint64_t* beg = T;
    uint64_t* end = T + (uint64_t)nseg * NBl;
    for (uint32_t s = 0; s < W; ++s) {
        uint64_t x, y;
        if (s == me) {
            const uint32_t gb = me * NBl + b;
            const uint32_t c = cursor[gb];
            x = delta + (uint64_t)gb * cap;
            y = x + (c < cap ? c : cap);
        } else {
            x = roff[(uint64_t)s * NBl + b];
            y = x + hrecv[(uint64_t)s * NBl + b];
        }
        beg[(uint64_t)s * NBl + b] = x;
        end[(uint64_t)s * NBl + b] = y;
    }
    if (nseg > W) {      // the partition's overflow segment of my own buckets
        const uint32_t gb = me * NBl + b;
        const uint64_t x = pseg[2ull * NB_total + gb], y = pseg[3ull * NB_total + gb];
        beg[(uint64_t)W * NBl + b] = y > x ? delta + x : 0;
        end[(uint64_t)W * NBl + b] = y > x ? delta + y : 0;
    }
}

// Measurement aid (SNK_DBG_FAKE_SEGS = n on one rank): every bucket's slots seen as n segments of equal parts -- the record layout a
// rank of an n-GPU job counts (one segment per source), same records, same result
__global__ void __launch_bounds__(256) fake_seg_kernel(const uint32_t* __restrict__ cursor, uint32_t NB, uint32_t cap, uint32_t nseg, uint64_t* __restrict__ T) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= NB) return;
    const uint32_t c = cursor[b] < cap ? cursor[b] : cap;
    for (uint32_t s = 0; s < nseg; ++s) {
        T[(uint64_t)s * NB + b] = (uint64_t)b * cap + (uint64_t)c * s / nseg;
        T[(uint64_t)(nseg + s) * NB + b] = (uint64_t)b * cap + (uint64_t)c * (s + 1) / nseg;
    }
}

}  // namespace

// Piece (q, r) of the ranged record exchange: h_rs = [2][W][R] items per (destination, range) and per (source, range); the
// ranges of one peer follow each other in both buffers, the peers follow each other in rank order.  Byte offsets / counts of
// range r for every peer (host arithmetic shared by the step and by snk_comm_selftest).
void snk_plan_range_pieces(const unsigned long long* h_rs, uint32_t W, uint32_t R, uint32_t r, uint64_t item_bytes, uint64_t* sbeg, uint64_t* scnt,
                           uint64_t* rbeg, uint64_t* rcnt) {
    uint64_t sacc = 0, racc = 0;
    for (uint32_t q = 0; q < W; ++q) {
        uint64_t sd = 0, rd = 0, stot = 0, rtot = 0;
        for (uint32_t x = 0; x < R; ++x) {
            const uint64_t a = h_rs[(size_t)q * R + x], b = h_rs[(size_t)W * R + (size_t)q * R + x];
            if (x < r) { sd += a; rd += b; }
            stot += a; rtot += b;
        }
        sbeg[q] = (sacc + sd) * item_bytes; scnt[q] = h_rs[(size_t)q * R + r] * item_bytes;
        rbeg[q] = (racc + rd) * item_bytes; rcnt[q] = h_rs[(size_t)W * R + (size_t)q * R + r] * item_bytes;
        sacc += stot; racc += rtot;
    }
}

namespace {

struct step_ctx {
    snk_ctx* ctx;
    snk_comm* comm;
    hipStream_t st;
    char* err;
    size_t errcap;
    uint32_t W, me;
    // pinned staging for the small host -> device uploads of the step (offset tables): a bump allocator, reset per step
    ull* pin = nullptr;
    size_t pin_cap = 0, pin_used = 0;
    bool streamed = false;         // the partition is the context's open streamed job (snk_shard_stream_*), not a pass over resident reads
};

struct shard_host {                 // per-context host resources of the step (kept across steps)
    ull* pin = nullptr;
    size_t pin_cap = 0;
    hipStream_t cstream = nullptr;  // the exchange's stream (the count of range r runs while range r+1 is on the wire)
    std::vector<hipEvent_t> ev;
    ~shard_host() {
        if (pin) (void)hipHostFree(pin);
        if (cstream) (void)hipStreamDestroy(cstream);
        for (auto e : ev) (void)hipEventDestroy(e);
    }
};

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define ALLOC(ptr, type, count) do { void* _p = nullptr; int _rc = snk_ctx_alloc(X.ctx, sizeof(type) * (size_t)(count), &_p, X.err, X.errcap); if (_rc) return _rc; ptr = (type*)_p; } while (0)

// host values -> a device array (stream-ordered copy out of the pinned staging area)
int upload(step_ctx& X, const ull* h, size_t n, ull** d_out) {
    char* err = X.err; size_t errcap = X.errcap;
    if (X.pin_used + n > X.pin_cap) return snk_fail(SNK_E_INTERNAL, err, errcap, "shard step: staging area exhausted");
    ull* d;
    ALLOC(d, ull, n + 1);
    ull* p = X.pin + X.pin_used;
    X.pin_used += n;
    memcpy(p, h, n * 8);
    SNK_HIP_TRY(hipMemcpyAsync(d, p, n * 8, hipMemcpyHostToDevice, X.st));
    *d_out = d;
    return SNK_OK;
}

// all-to-all of variable pieces whose per-destination counts (items) every rank holds in a DEVICE array: one read-back tells
// every rank the whole matrix
int exchange_counts(step_ctx& X, const ull* d_counts, uint32_t k, std::vector<ull>& all) {
    all.assign((size_t)X.W * k, 0);
    return X.comm->gather_counts(d_counts, k, all.data(), X.st, X.err, X.errcap);
}

int a2a_items(step_ctx& X, const void* send, const std::vector<ull>& send_items, void* recv, const std::vector<ull>& recv_items, size_t item_bytes) {
    const uint32_t W = X.W;
    std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
    uint64_t a = 0, b = 0;
    for (uint32_t p = 0; p < W; ++p) { sbeg[p] = a; scnt[p] = send_items[p] * item_bytes; a += scnt[p]; rbeg[p] = b; rcnt[p] = recv_items[p] * item_bytes; b += rcnt[p]; }
    return X.comm->a2a(send, sbeg.data(), scnt.data(), recv, rbeg.data(), rcnt.data(), X.st, X.err, X.errcap);
}

struct range_wait { hipStream_t st; hipEvent_t* ev; uint32_t n; };
int range_ready(void* user, uint32_t r) {
    range_wait* w = (range_wait*)user;
    if (r >= w->n) return 1;
    return hipStreamWaitEvent(w->st, w->ev[r], 0) == hipSuccess ? 0 : 1;
}

int all_ranges_ready(void* user) {       // the whole exchange has landed (the hot buckets' expansion reads records of every range)
    range_wait* w = (range_wait*)user;
    for (uint32_t r = 0; r < w->n; ++r) if (hipStreamWaitEvent(w->st, w->ev[r], 0) != hipSuccess) return 1;
    return 0;
}

uint32_t plan_buckets(uint64_t inst_ub, uint32_t world, uint32_t K, uint32_t forced, double ratio, uint32_t* tight_out = nullptr, uint32_t* screen_out = nullptr) {
    uint64_t nb = forced;
    if (tight_out) *tight_out = 0;
    const bool screen_ok = screen_out && *screen_out;       // (in: the call's parameters allow the bit filter; out: it is taken)
    if (screen_out) *screen_out = 0;
    if (!nb) {
        const uint32_t dflt = K == 48 ? 5000u : 3500u;
        uint64_t target = snk_opt_u32("target_inst", dflt);
        if (!snk_opt_is_set("target_inst") && ratio > 0.0 && snk_opt_u32("adaptive_buckets", 1)) {
            // (the rule of the one-GPU path, snk_pipeline.hip: smaller buckets when the tables would run more than ~65 % full)
            double lim = (double)snk_count_limit(K, 0u, 0u);
            // (tables that run full are counted with booked slots, as on the one-GPU path: every rank takes the same turn, the ratio is job-wide)
            const bool tight_off = snk_opt_is_set("count_tight") && snk_opt_u32("count_tight", 1) == 0u;
            if (tight_out && 0.65 * lim / ratio < (double)dflt && !tight_off) {
                *tight_out = (snk_count_slots(K) - snk_count_slots(K) / 16u) | (snk_opt_u32("tight_tries", 48) << 16);
                lim = (double)snk_count_limit(K, 0u, *tight_out);
            }
            if (0.65 * lim / ratio < (double)dflt) { const double t = 0.01 * snk_opt_u32("bucket_fill_pct", 50) * lim / ratio; target = t < 600.0 ? 600u : (uint64_t)t; if (target > dflt) target = dflt; }
            // (... and above 0.3 distinct k-mers per instance behind the bit filter, whose table only sees what can be retained: snk_pipeline.hip)
            const uint32_t ng = snk_opt_u32("count_screen_ng", 1);
            if (screen_ok && tight_out && *tight_out && ng && (ng >= 2 || ratio > 0.01 * snk_opt_u32("screen_ratio_pct", 30))) { *screen_out = 3; target = snk_opt_u32("screen_target", 4000); }
        }
        nb = (inst_ub + target - 1) / target;
        if (nb < 1) nb = 1;
        if (nb > (1ull << 26)) nb = 1ull << 26;
    }
    const uint64_t floor_ = (inst_ub >> 20) + 1;       // at most ~1 M instances per bucket (one workgroup counts a bucket)
    if (nb < floor_) nb = floor_;
    nb = (nb + world - 1) / world * world;
    return (uint32_t)nb;
}

template <typename In, typename Out>
int excl_scan(step_ctx& X, In in, Out* out, size_t n, Out init) {
    char* err = X.err; size_t errcap = X.errcap;
    size_t tb = 0;
    SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb, in, out, init, n, rocprim::plus<Out>(), X.st));
    void* tmp;
    TRY(snk_ctx_alloc(X.ctx, tb, &tmp, err, errcap));
    SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tb, in, out, init, n, rocprim::plus<Out>(), X.st));
    return SNK_OK;
}

int step_impl(step_ctx& X, shard_host& H, const snk_dev_reads* in, const snk_params* p, uint64_t total_reads, uint32_t flags, snk_shard_result* out) {
    X.ctx->count_tight = 0;        // (the ranks size their buckets alike, from one rule: the default count kernel's)
    X.ctx->count_screen = 0;
    snk_ctx* ctx = X.ctx;
    snk_comm* comm = X.comm;
    hipStream_t st = X.st;
    char* err = X.err;
    size_t errcap = X.errcap;
    const uint32_t W = X.W, me = X.me, K = p->K;
    // A multi-rank step takes its scratch from plain hipMalloc blocks, not from the growing arena (snk_ctx.h: mapped through the VMM API):
    // buffers that cross xGMI stay ordinary device memory (RCCL over several devices on VMM ranges is not something this build
    // environment can test), and in-process ranks -- one host thread and one context each -- crashed inside the runtime when one thread
    // unmapped its arena while another copied out of its own (round 4: the suite's in-process worlds with the arena on by default).
    ctx->arena_legacy = W > 1;
    snk_set_mlen(ctx, p);
    const uint64_t syncs0 = snk_sync_count();
    comm->bytes_sent = 0;
    snk_phase_timer tm(st);
    tm.mark();   // 0
    // ---- the job's bucket count: from the caller's read total, or from an exchange of the slabs' sizes
    uint64_t inst_ub = 0;
    const uint64_t kpr = in->read_len >= K ? in->read_len - K + 1 : 0;
    if (p->n_buckets == 0) {
        if (total_reads) inst_ub = total_reads * kpr;
        else if (W == 1) inst_ub = in->n_reads * kpr;
        else {
            ull mine = in->n_reads, *d_mine;
            // (a scratch block next to what the previous step still holds; snk_shard_begin hands everything back)
            TRY(upload(X, &mine, 1, &d_mine));
            std::vector<ull> all;
            TRY(exchange_counts(X, d_mine, 1, all));
            for (ull v : all) inst_ub += v * kpr;
        }
    }
    const bool have_ratio = inst_ub && comm->claim_ratio > 0.0 && comm->claim_ratio_reads == inst_ub && comm->claim_ratio_k == K * 2 + 256u * ctx->mlen;      // (the group's history: snk_comm.h)
    // Bucket size: from the job-wide ratio of distinct k-mers per instance the previous step exchanged; without that history the
    // count stage looks at its first buckets, the ranks agree on what they saw (one more exchange), and if the tables overflow as
    // a rule the reads are partitioned and exchanged once more into smaller buckets (error-rich reads: see snk_pipeline.hip).
    const bool adaptive = inst_ub && !snk_opt_is_set("target_inst") && snk_opt_u32("adaptive_buckets", 1) != 0;
    double ratio = have_ratio ? comm->claim_ratio : 0.0;
    uint32_t NB_total = 0, NBl = 0;
    uint64_t n_inst = 0, inst_hint = 0, exch_records = 0;
    snk_shard_state* S = nullptr;
    const int has_bc = in->bc ? 1 : 0;
    struct agree_ctx { step_ctx* X; } ag{&X};
    snk_count_pilot pilot{0.0, [](void* u, double* pb) -> int {
        step_ctx& Xc = *static_cast<agree_ctx*>(u)->X;
        ull mine = (ull)(*pb * 1024.0), *d_mine;
        int r2 = upload(Xc, &mine, 1, &d_mine);
        if (r2) return r2;
        std::vector<ull> all;
        if ((r2 = exchange_counts(Xc, d_mine, 1, all))) return r2;
        ull sum = 0;
        for (ull v : all) sum += v;
        *pb = (double)sum / 1024.0 / (double)all.size();
        return 0;
    }, &ag};
    uint32_t n_hot_buckets = 0;
    uint64_t pair_max_records = 0;       // the most records (bytes) this rank sends to one other rank
    auto count_pass = [&](bool with_pilot) -> int {
        // ---- trim + one-pass partition over all buckets of the job
        n_inst = 0;
        if (X.streamed) TRY(snk_shard_job_adopt(ctx, &n_inst, st, err, errcap));
        else TRY(snk_shard_begin(ctx, in, p, me, W, NB_total, &n_inst, st, err, errcap));
        S = snk_shard_state_of(ctx);
        const snk_partition& part = S->part;
        if (part.n_supermers >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^32 supermers on one rank");
        tm.mark();   // 1
        inst_hint = total_reads ? total_reads * kpr / W : (inst_ub ? inst_ub / W : n_inst);
        exch_records = 0;
        if (W == 1) {
            // one rank owns every bucket: the slots are counted where they are (exactly the one-GPU path)
            tm.mark();   // 2
            tm.mark();   // 3
            const uint32_t fake = snk_opt_u32("dbg_fake_segs", 0);
            if (fake > 1 && fake <= 32 && part.n_overflow == 0) {
                uint64_t* T;
                ALLOC(T, uint64_t, 2ull * fake * NB_total + 2);
                hipLaunchKernelGGL(fake_seg_kernel, dim3((NB_total + 255) / 256), dim3(256), 0, st, part.cursor, NB_total, part.cap, fake, T);
                TRY(snk_stage_count_table(ctx, st, K, part.records, T, T + (uint64_t)fake * NB_total, NB_total, fake, NB_total, p->min_freq,
                                          has_bc ? p->min_bc : 0u, 0u, inst_hint, S->status, false, &S->tab, err, errcap));
            } else {
            // (hot minimiser buckets -- repeat families -- are re-partitioned by k-mer hash as on the one-GPU path)
            snk_hot hot;
            TRY(snk_stage_hot(ctx, st, K, false, &S->part, &hot, err, errcap));
            n_hot_buckets = hot.n_hot;
            TRY(snk_stage_count_table(ctx, st, K, part.records, part.seg, part.seg + NB_total, 2 * NB_total, part.nseg, NB_total, p->min_freq,
                                      has_bc ? p->min_bc : 0u, 0u, inst_hint, S->status, false, &S->tab, err, errcap, nullptr, with_pilot ? &pilot : nullptr, nullptr, true, &hot));
            }
            snk_ctx_release_block(ctx, part.records);
        } else {
            // ---- histograms: row p of my cursor array goes to rank p
            uint32_t *hrecv, *hsend_x;
            ALLOC(hrecv, uint32_t, (uint64_t)NB_total + 4);
            ALLOC(hsend_x, uint32_t, (uint64_t)NB_total + 4);
            {
                std::vector<uint64_t> beg(W), cnt(W, (uint64_t)NBl * 4);
                for (uint32_t q = 0; q < W; ++q) beg[q] = (uint64_t)q * NBl * 4;
                TRY(comm->a2a(part.cursor, beg.data(), cnt.data(), hrecv, beg.data(), cnt.data(), st, err, errcap));
            }
            // my own buckets take no part in the exchange: zero rows in both directions
            SNK_HIP_TRY(hipMemcpyAsync(hsend_x, part.cursor, (size_t)NB_total * 4, hipMemcpyDeviceToDevice, st));
            SNK_HIP_TRY(hipMemsetAsync(hsend_x + (uint64_t)me * NBl, 0, (size_t)NBl * 4, st));
            SNK_HIP_TRY(hipMemsetAsync(hrecv + (uint64_t)me * NBl, 0, (size_t)NBl * 4, st));
            uint32_t R = snk_opt_u32("exchange_ranges", 4);
            if (R < 1) R = 1;
            if (R > NBl) R = NBl;
            if (R > 64) R = 64;
            ull* d_rs;      // [2][W][R] records per (destination, range) and per (source, range)
            ALLOC(d_rs, ull, 2ull * W * R + 1);
            hipLaunchKernelGGL(range_sum_kernel, dim3(W * R), dim3(256), 0, st, hsend_x, NBl, R, d_rs);
            hipLaunchKernelGGL(range_sum_kernel, dim3(W * R), dim3(256), 0, st, hrecv, NBl, R, d_rs + (size_t)W * R);
            uint32_t* soff32;
            ull* roff;
            ALLOC(soff32, uint32_t, (uint64_t)NB_total + 4);
            ALLOC(roff, ull, (uint64_t)NB_total + 4);
            TRY((excl_scan<const uint32_t*, uint32_t>(X, hsend_x, soff32, (size_t)NB_total + 1, 0u)));
            {
                auto it = rocprim::make_transform_iterator(hrecv, [] __device__(uint32_t v) { return (ull)v; });
                TRY((excl_scan<decltype(it), ull>(X, it, roff, (size_t)NB_total + 1, 0ull)));
            }
            std::vector<ull> h_rs(2ull * W * R);
            SNK_HIP_TRY(hipMemcpyAsync(h_rs.data(), d_rs, h_rs.size() * 8, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));          // read-back: piece sizes of the record exchange (both directions)
            uint64_t n_send = 0, n_recv = 0;
            for (size_t q = 0; q < (size_t)W * R; ++q) { n_send += h_rs[q]; n_recv += h_rs[(size_t)W * R + q]; }
            uint4 *sendb, *recvb;
            ALLOC(sendb, uint4, 2 * n_send + 2);
            ALLOC(recvb, uint4, 2 * n_recv + 2);
            TRY(snk_stage_partition_compact_remote(ctx, st, &part, soff32, sendb, me * NBl, (me + 1) * NBl, err, errcap));
            tm.mark();   // 2
            // ---- the records, range by range on the exchange's stream
            if (H.ev.size() < R + 1) { const size_t o = H.ev.size(); H.ev.resize(R + 1); for (size_t q = o; q < H.ev.size(); ++q) SNK_HIP_TRY(hipEventCreateWithFlags(&H.ev[q], hipEventDisableTiming)); }
            if (!H.cstream) SNK_HIP_TRY(hipStreamCreateWithFlags(&H.cstream, hipStreamNonBlocking));
            SNK_HIP_TRY(hipEventRecord(H.ev[R], st));
            SNK_HIP_TRY(hipStreamWaitEvent(H.cstream, H.ev[R], 0));
            {
                // offsets of piece (q, r): the ranges of one peer follow each other
                std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
                for (uint32_t r = 0; r < R; ++r) {
                    snk_plan_range_pieces(h_rs.data(), W, R, r, 32, sbeg.data(), scnt.data(), rbeg.data(), rcnt.data());
                    TRY(comm->a2a(sendb, sbeg.data(), scnt.data(), recvb, rbeg.data(), rcnt.data(), H.cstream, err, errcap));
                    SNK_HIP_TRY(hipEventRecord(H.ev[r], H.cstream));
                }
            }
            exch_records = n_send * 32;
            pair_max_records = 0;
            for (uint32_t q = 0; q < W; ++q) {
                uint64_t to_q = 0;
                for (uint32_t r = 0; r < R; ++r) to_q += h_rs[(size_t)q * R + r];
                if (q != me && to_q * 32 > pair_max_records) pair_max_records = to_q * 32;
            }
            // ---- segment tables: W sources (mine = the slots, in place) + my overflow segment
            const uint32_t nseg = W + (part.n_overflow ? 1u : 0u);
            uint64_t* T;
            ALLOC(T, uint64_t, 2ull * nseg * NBl + 2);
            const ull delta = (ull)(((intptr_t)part.records - (intptr_t)recvb) / 32);
            hipLaunchKernelGGL(seg_table_kernel, dim3((NBl + 255) / 256), dim3(256), 0, st, roff, hrecv, part.cursor, part.seg, W, me, NBl, NB_total, part.cap,
                               nseg, delta, T);
            SNK_HIP_TRY(hipGetLastError());
            tm.mark();   // 3
            std::vector<uint32_t> bounds(R + 1);
            for (uint32_t r = 0; r <= R; ++r) bounds[r] = (uint32_t)((uint64_t)NBl * r / R);
            range_wait rw{st, H.ev.data(), R};
            snk_count_ranges rg{R, bounds.data(), range_ready, &rw};
            // hot minimiser buckets (a repeat family's, a homopolymer's: all of their records meet on ONE rank, and one workgroup would count
            // them in hundreds of passes): planned here from the segment table -- the histograms say how large every bucket is before a record
            // has arrived --, expanded by the count stage behind its ranged launches, when the exchange is through
            snk_hot hot;
            TRY(snk_stage_hot_plan(ctx, st, K, false, T, T + (uint64_t)nseg * NBl, NBl, nseg, NBl, part.cap, &hot, err, errcap));
            hot.before_expand = all_ranges_ready; hot.user = &rw; hot.src_records = recvb;
            const int rcc = snk_stage_count_table(ctx, st, K, recvb, T, T + (uint64_t)nseg * NBl, NBl, nseg, NBl, p->min_freq, has_bc ? p->min_bc : 0u, 0u, inst_hint,
                                                  S->status, false, &S->tab, err, errcap, &rg, with_pilot ? &pilot : nullptr, nullptr, true, &hot);      // (the region compaction rides in the prune, as on the one-GPU path)
            snk_stage_hot_drop(&hot);
            if (rcc) return rcc;
            n_hot_buckets = hot.n_hot;
            snk_ctx_release_block(ctx, part.records);
            snk_ctx_release_block(ctx, sendb);
            snk_ctx_release_block(ctx, recvb);
        }
        return SNK_OK;
    };
    uint32_t repartitioned = 0;
    for (int pass = 0; pass < 2; ++pass) {
        // (a streamed step sized its buckets when it was opened -- the same rule on the same job-wide figures -- and cannot partition twice: its
        // slabs are gone; error-rich data without the group's history are counted in hash-split sub-passes then)
        {
            uint32_t tight = 0, screen = (K == 48 && p->min_freq >= 3 && (!has_bc || p->min_bc <= 2)) ? 1u : 0u;
            NB_total = X.streamed ? snk_shard_state_of(ctx)->NB_total : plan_buckets(inst_ub, W, K, p->n_buckets, adaptive ? ratio : 0.0, &tight, &screen);
            if (X.streamed) screen = 0;
            ctx->count_tight = tight; ctx->count_screen = screen;
            ctx->last_count_limit = screen ? std::min(snk_count_limit(K, 0u, tight), snk_count_screen_limit()) : snk_count_limit(K, 0u, tight);
            if (screen && ratio > 0.0) { comm->claim_ratio = ratio; comm->claim_ratio_reads = inst_ub; comm->claim_ratio_k = K * 2 + 256u * ctx->mlen; }      // (a screened step reports the table's view: the group keeps the ratio the decision was made on)
        }
        NBl = NB_total / W;
        tm.n = 1;
        const int rcp = count_pass(adaptive && !have_ratio && pass == 0 && p->n_buckets == 0 && !X.streamed);
        if (rcp == SNK_RETARGET) {
            // every rank took this turn (the figure is job-wide); what the exchange still has in flight lands first
            if (H.cstream) SNK_HIP_TRY(hipStreamSynchronize(H.cstream));
            SNK_HIP_TRY(snk_sync(st));
            ratio = pilot.per_bucket * (double)NB_total / (double)inst_ub;
            repartitioned = 1;
            continue;
        }
        if (rcp) return rcp;
        break;
    }
    const snk_partition& part = S->part;
    tm.mark();   // 4
    const uint64_t n_kmers = S->tab.n;

    // ---- bucket-local prune with membership queries for neighbours owned by other ranks
    TRY(snk_shard_prune_plan(ctx, nullptr, st, err, errcap));
    // one read-back: every rank's query counts (still on the device: bl.qcount) and retained k-mers
    ull* d_cnt = S->bl.qcount;
    // ... and, riding along, what the tables of my count kernel held (distinct k-mers) over how many instances: the job-wide ratio
    // sizes the buckets of the NEXT step, the same on every rank because it is computed from the same exchanged words
    constexpr uint32_t QX = 3;
    {
        ull hn[QX] = {n_kmers, S->tab.distinct, n_inst}, *up;
        TRY(upload(X, hn, QX, &up));
        SNK_HIP_TRY(hipMemcpyAsync(d_cnt + W, up, QX * 8, hipMemcpyDeviceToDevice, st));
    }
    std::vector<ull> qall;
    TRY(exchange_counts(X, d_cnt, W + QX, qall));
    std::vector<ull> q_send(W), q_recv(W), all_n(W);
    {
        ull dsum = 0, isum = 0;
        for (uint32_t q = 0; q < W; ++q) {
            q_send[q] = qall[(size_t)me * (W + QX) + q]; q_recv[q] = qall[(size_t)q * (W + QX) + me]; all_n[q] = qall[(size_t)q * (W + QX) + W];
            dsum += qall[(size_t)q * (W + QX) + W + 1]; isum += qall[(size_t)q * (W + QX) + W + 2];
        }
        if (isum && !X.ctx->count_screen) { comm->claim_ratio = (double)dsum / (double)isum; comm->claim_ratio_reads = inst_ub; comm->claim_ratio_k = K * 2 + 256u * ctx->mlen; }      // (job-wide figures, kept with the group: every rank takes the same decision next time)
    }
    uint64_t nq = 0, nq_in = 0;
    for (uint32_t q = 0; q < W; ++q) { nq += q_send[q]; nq_in += q_recv[q]; }
    ull* d_qoff;
    {
        std::vector<ull> qoff(W + 1, 0);
        for (uint32_t q = 0; q < W; ++q) qoff[q + 1] = qoff[q] + q_send[q];
        TRY(upload(X, qoff.data(), W + 1, &d_qoff));
    }
    uint8_t *qbuf, *qin, *ans, *ans_back;
    ALLOC(qbuf, uint8_t, nq * 24 + 32);
    ALLOC(qin, uint8_t, nq_in * 24 + 32);
    ALLOC(ans, uint8_t, nq_in * 4 + 32);
    ALLOC(ans_back, uint8_t, nq * 4 + 32);
    TRY(snk_shard_prune_fill(ctx, d_qoff, qbuf, st, err, errcap));
    TRY(a2a_items(X, qbuf, q_send, qin, q_recv, 24));
    TRY(snk_shard_prune_answer(ctx, qin, nq_in, ans, st, err, errcap));
    TRY(a2a_items(X, ans, q_recv, ans_back, q_send, 4));
    TRY(snk_shard_prune_apply(ctx, qbuf, ans_back, nq, d_qoff, st, err, errcap));
    tm.mark();   // 5

    // ---- fragments of my chunks (global node numbering = the ranks' tables behind each other)
    std::vector<ull> node_off(W + 1, 0);
    for (uint32_t q = 0; q < W; ++q) node_off[q + 1] = node_off[q] + all_n[q];
    ull* d_node_off;
    TRY(upload(X, node_off.data(), W + 1, &d_node_off));
    snk_shard_frags fr;
    TRY(snk_shard_fragments(ctx, d_node_off, node_off[me], &fr, st, err, errcap));
    const uint64_t F = fr.n_frags;
    tm.mark();   // 6

    // ---- links between fragments, decided on the owners: an end asks the rank that owns the state it points at.
    // One read-back: everybody's fragment count and link-query counts (counting needs no global fragment numbering).
    snk_phase_timer jt(st);
    jt.mark();   // j0
    // One pass: destination q's queries go to region q of the buffer (capacity = every end I have), the cursors come back
    // with the fragment count in ONE read-back, the answers return into the same regions.
    ALLOC(S->lq_count, ull, W + 2);
    ALLOC(S->lq_cursor, ull, W + 2);
    const uint64_t lcap = 2 * F + 1;
    uint8_t *lqbuf, *lans_back;
    ALLOC(lqbuf, uint8_t, (uint64_t)W * lcap * 24 + 32);
    ALLOC(lans_back, uint8_t, (uint64_t)W * lcap * 4 + 32);
    {
        std::vector<ull> init(W + 1);
        for (uint32_t q = 0; q < W; ++q) init[q] = (ull)q * lcap;
        init[W] = F;
        ull* d_init;
        TRY(upload(X, init.data(), W + 1, &d_init));
        TRY(snk_shard_links_fill(ctx, d_init, lqbuf, st, err, errcap));      // copies W+1 words: the cursors and, behind them, F
    }
    std::vector<ull> lall;
    TRY(exchange_counts(X, S->lq_cursor, W + 1, lall));
    std::vector<ull> l_send(W), l_recv(W), all_F(W), frag_off(W + 1, 0);
    for (uint32_t q = 0; q < W; ++q) {
        l_send[q] = lall[(size_t)me * (W + 1) + q] - (ull)q * lcap;
        l_recv[q] = lall[(size_t)q * (W + 1) + me] - (ull)me * (2 * lall[(size_t)q * (W + 1) + W] + 1);
        all_F[q] = lall[(size_t)q * (W + 1) + W];
    }
    for (uint32_t q = 0; q < W; ++q) frag_off[q + 1] = frag_off[q] + all_F[q];
    const uint64_t Ft = frag_off[W];
    if (2 * Ft >= (1ull << 32)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than 2^31 fragments in the job");
    S->my_end_base = 2ull * frag_off[me];
    uint64_t nlq = 0, nlq_in = 0;
    for (uint32_t q = 0; q < W; ++q) { nlq += l_send[q]; nlq_in += l_recv[q]; }
    uint8_t *lqin, *lans;
    ALLOC(lqin, uint8_t, nlq_in * 24 + 32);
    ALLOC(lans, uint8_t, nlq_in * 4 + 32);
    {
        std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
        uint64_t acc = 0;
        for (uint32_t q = 0; q < W; ++q) { sbeg[q] = (uint64_t)q * lcap * 24; scnt[q] = l_send[q] * 24; rbeg[q] = acc * 24; rcnt[q] = l_recv[q] * 24; acc += l_recv[q]; }
        TRY(comm->a2a(lqbuf, sbeg.data(), scnt.data(), lqin, rbeg.data(), rcnt.data(), st, err, errcap));
        TRY(snk_shard_links_answer(ctx, lqin, nlq_in, lans, st, err, errcap));
        for (uint32_t q = 0; q < W; ++q) { std::swap(sbeg[q], rbeg[q]); std::swap(scnt[q], rcnt[q]); sbeg[q] /= 6; scnt[q] /= 6; rbeg[q] /= 6; rcnt[q] /= 6; }
        TRY(comm->a2a(lans, sbeg.data(), scnt.data(), lans_back, rbeg.data(), rcnt.data(), st, err, errcap));
    }
    uint32_t* flink_w = nullptr;
    TRY(snk_dist_links_apply_regions(ctx, st, &S->frags, lqbuf, lans_back, W, lcap, l_send.data(), &flink_w, err, errcap));
    const void* flink = flink_w;
    jt.mark();   // j1 links

    // ---- owner-side join: every rank sees the job's LINK structure only (12 bytes per fragment), ranks the fragment lists,
    // places its own fragments and sends each to the rank that owns its unitig's head, which writes the unitig
    uint32_t* nk_all;
    uint32_t* fl_all;
    ALLOC(nk_all, uint32_t, Ft + 4);
    ALLOC(fl_all, uint32_t, 2 * Ft + 4);
    {
        std::vector<uint64_t> c4(W), c8(W);
        for (uint32_t q = 0; q < W; ++q) { c4[q] = all_F[q] * 4; c8[q] = all_F[q] * 8; }
        TRY(comm->allgatherv(fr.nk, c4.data(), nk_all, st, err, errcap));
        TRY(comm->allgatherv(flink, c8.data(), fl_all, st, err, errcap));
    }
    ull* d_frag_off;
    TRY(upload(X, frag_off.data(), W + 1, &d_frag_off));
    jt.mark();   // j2 gather links
    bool ranked = false;
    uint64_t exch_spl = 0, exch_rank = 0, pair_max_rank = 0;
    const bool want_partitioned = snk_opt_u32("join_replicated", 0) == 0;
    for (int attempt = 0; want_partitioned && !ranked && attempt < 3; ++attempt) {
        // (a second attempt follows a circle cut: the links changed, everything derived from them is made again)
        uint64_t m_spl = 0;
        const void* w1p = nullptr;
        TRY(snk_shard_prank_begin(ctx, Ft, nk_all, fl_all, frag_off[me], &m_spl, &w1p, st, err, errcap));
        std::vector<uint64_t> shares(W);
        uint64_t tot16 = 0;
        for (uint32_t q = 0; q < W; ++q) { shares[q] = (m_spl * (q + 1) / W - m_spl * q / W) * 16; tot16 += shares[q]; }
        uint8_t* w1_all;
        ALLOC(w1_all, uint8_t, tot16 + 32);
        TRY(comm->allgatherv(w1p, shares.data(), w1_all, st, err, errcap));
        exch_spl = m_spl * 16;
        uint32_t circ = 0;
        TRY(snk_shard_prank_walk(ctx, w1_all, d_frag_off, nullptr, &circ, st, err, errcap));
        if (circ == 1) break;            // something the partitioned ranking does not handle: every rank ranks the replicated way
        if (circ == 2) continue;         // circles were cut (the same cuts on every rank: the data is replicated): rank again
        {
            // the (state, distance, terminal) records of my walks, routed to the states' owners in one pass (regions of n_rec)
            const uint64_t n_rec = S->pr.n_rec, rcap = n_rec + 1;
            std::vector<ull> init(W);
            for (uint32_t q = 0; q < W; ++q) init[q] = (ull)q * rcap;
            ull* d_init;
            TRY(upload(X, init.data(), W, &d_init));
            uint8_t* rsend;
            ALLOC(rsend, uint8_t, (uint64_t)W * rcap * 16 + 32);
            TRY(snk_shard_prank_route(ctx, d_frag_off, d_init, rsend, st, err, errcap));
            ull* d_cur = S->pr_cursor;
            ull* d_cur2;
            ALLOC(d_cur2, ull, W + 2);
            if (n_rec) SNK_HIP_TRY(hipMemcpyAsync(d_cur2, d_cur, W * 8ull, hipMemcpyDeviceToDevice, st));
            else SNK_HIP_TRY(hipMemcpyAsync(d_cur2, d_init, W * 8ull, hipMemcpyDeviceToDevice, st));      // (no records: the fill was not launched)
            {
                ull hr = rcap, *up;
                TRY(upload(X, &hr, 1, &up));
                SNK_HIP_TRY(hipMemcpyAsync(d_cur2 + W, up, 8, hipMemcpyDeviceToDevice, st));
            }
            std::vector<ull> rall;
            TRY(exchange_counts(X, d_cur2, W + 1, rall));
            std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
            uint64_t n_in = 0, n_out = 0;
            for (uint32_t q = 0; q < W; ++q) {
                const ull cap_q = rall[(size_t)q * (W + 1) + W];
                sbeg[q] = (uint64_t)q * rcap * 16; scnt[q] = (rall[(size_t)me * (W + 1) + q] - (ull)q * rcap) * 16;
                rbeg[q] = n_in * 16; rcnt[q] = (rall[(size_t)q * (W + 1) + me] - (ull)me * cap_q) * 16;
                n_in += rcnt[q] / 16; n_out += scnt[q] / 16;
                if (q != me && scnt[q] > pair_max_rank) pair_max_rank = scnt[q];
            }
            uint8_t* rk_in;
            ALLOC(rk_in, uint8_t, n_in * 16 + 32);
            TRY(comm->a2a(rsend, sbeg.data(), scnt.data(), rk_in, rbeg.data(), rcnt.data(), st, err, errcap));
            TRY(snk_shard_place_ranked(ctx, K, rk_in, n_in, d_frag_off, nullptr, nullptr, st, err, errcap));
            snk_ctx_release_block(ctx, rsend);
            ranked = true;
            exch_rank = n_out * 16;
        }
    }
    if (!ranked)      // a list is a circle (same verdict on every rank: it comes from replicated data), or the replicated mode was asked for
        TRY(snk_shard_place(ctx, K, Ft, nk_all, fl_all, d_frag_off, frag_off[me], nullptr, nullptr, st, err, errcap));
    jt.mark();   // j3 rank + place
    // headers + bases to the owners of the unitigs, one pass into per-owner regions (capacity: all my fragments / all my bases;
    // every region of bases starts 16-byte aligned); the cursors come back in one read-back
    const uint64_t fcap = F + 1, bcap = (S->frags.total_bases + 15) / 16 * 16 + 16;
    uint8_t *hdr, *sb;
    ALLOC(hdr, uint8_t, (uint64_t)W * fcap * 32 + 32);
    ALLOC(sb, uint8_t, (uint64_t)W * bcap + 32);
    {
        std::vector<ull> hb(W), bb(W);
        for (uint32_t q = 0; q < W; ++q) { hb[q] = (ull)q * fcap; bb[q] = (ull)q * bcap; }
        ull *d_hoff, *d_boff;
        TRY(upload(X, hb.data(), W, &d_hoff));
        TRY(upload(X, bb.data(), W, &d_boff));
        TRY(snk_shard_route_fill(ctx, K, d_frag_off, d_hoff, d_boff, hdr, sb, st, err, errcap));
    }
    std::vector<ull> fall;
    {
        ull* d_v;
        ALLOC(d_v, ull, 2 * W + 4);
        SNK_HIP_TRY(hipMemcpyAsync(d_v, S->rt_cursor, 2ull * W * 8, hipMemcpyDeviceToDevice, st));
        ull caps[2] = {fcap, bcap}, *up;
        TRY(upload(X, caps, 2, &up));
        SNK_HIP_TRY(hipMemcpyAsync(d_v + 2 * W, up, 16, hipMemcpyDeviceToDevice, st));
        TRY(exchange_counts(X, d_v, 2 * W + 2, fall));
    }
    const size_t FS = 2 * (size_t)W + 2;
    std::vector<ull> f_send(W), f_recv(W), b_send(W), b_recv(W), hseg(W + 1, 0), bseg(W + 1, 0);
    for (uint32_t q = 0; q < W; ++q) {
        const ull fcap_q = fall[q * FS + 2 * W], bcap_q = fall[q * FS + 2 * W + 1];
        f_send[q] = F ? fall[me * FS + q] - (ull)q * fcap : 0;
        b_send[q] = F ? (fall[me * FS + W + q] - (ull)q * bcap + 15) / 16 * 16 : 0;
        f_recv[q] = fcap_q > 1 ? fall[q * FS + me] - (ull)me * fcap_q : 0;
        b_recv[q] = fcap_q > 1 ? (fall[q * FS + W + me] - (ull)me * bcap_q + 15) / 16 * 16 : 0;
        hseg[q + 1] = hseg[q] + f_recv[q]; bseg[q + 1] = bseg[q] + b_recv[q];
    }
    uint8_t *hdr_in, *b_in;
    ALLOC(hdr_in, uint8_t, hseg[W] * 32 + 32);
    ALLOC(b_in, uint8_t, bseg[W] + 32);
    {
        std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
        for (uint32_t q = 0; q < W; ++q) { sbeg[q] = (uint64_t)q * fcap * 32; scnt[q] = f_send[q] * 32; rbeg[q] = hseg[q] * 32; rcnt[q] = f_recv[q] * 32; }
        TRY(comm->a2a(hdr, sbeg.data(), scnt.data(), hdr_in, rbeg.data(), rcnt.data(), st, err, errcap));
        for (uint32_t q = 0; q < W; ++q) { sbeg[q] = (uint64_t)q * bcap; scnt[q] = b_send[q]; rbeg[q] = bseg[q]; rcnt[q] = b_recv[q]; }
        TRY(comm->a2a(sb, sbeg.data(), scnt.data(), b_in, rbeg.data(), rcnt.data(), st, err, errcap));
    }
    jt.mark();   // j4 route
    ull *d_hseg, *d_bseg;
    TRY(upload(X, hseg.data(), W + 1, &d_hseg));
    TRY(upload(X, bseg.data(), W + 1, &d_bseg));
    snk_shard_unitigs un;
    TRY(snk_shard_emit(ctx, K, hseg[W], hdr_in, d_hseg, d_bseg, b_in, &un, st, err, errcap));
    jt.mark();   // j5 emit
    tm.mark();   // 7
    SNK_HIP_TRY(snk_sync(st));

    memset(out, 0, sizeof *out);
    out->n_reads = in->n_reads;
    out->n_instances = n_inst;
    out->n_supermers = part.n_supermers;
    out->n_buckets_total = NB_total;
    out->n_kmers = n_kmers;
    out->keys = fr.keys; out->counts = fr.counts; out->ctx = fr.ctx;
    out->spectrum = fr.spectrum; out->spectrum_bins = fr.spectrum_bins;
    out->n_unitigs = un.n_unitigs; out->unitig_total_bases = un.total_bases;
    out->unitig_off = un.unitig_off; out->unitig_bases = un.unitig_bases; out->unitig_circular = un.unitig_circular;
    out->n_circles = un.n_circles;
    out->n_frags = F; out->n_frags_total = Ft; out->n_queries = nq; out->n_link_queries = nlq;
    out->ranking = ranked ? 1u : 0u;
    out->repartitioned = repartitioned;
    out->n_hot_buckets = n_hot_buckets;
    out->buckets_split = fr.buckets_split; out->max_slots_used = fr.max_slots_used;
    out->exchanged_bytes[0] = exch_records;
    out->exchanged_bytes[1] = 0; for (uint32_t q = 0; q < W; ++q) if (q != me) out->exchanged_bytes[1] += q_send[q] * 24 + q_recv[q] * 4;
    out->exchanged_bytes[2] = 0; for (uint32_t q = 0; q < W; ++q) if (q != me) out->exchanged_bytes[2] += l_send[q] * 24 + l_recv[q] * 4;
    out->exchanged_bytes[3] = (W - 1) * F * 12;
    out->exchanged_bytes[4] = exch_spl;
    out->exchanged_bytes[5] = exch_rank;
    out->exchanged_bytes[6] = 0; for (uint32_t q = 0; q < W; ++q) if (q != me) out->exchanged_bytes[6] += f_send[q] * 32 + b_send[q];
    out->exchanged_bytes[7] = comm->bytes_sent;
    {
        uint64_t* pm = out->pair_max_bytes;
        pm[0] = pair_max_records;
        for (uint32_t q = 0; q < W; ++q) if (q != me) {
            pm[1] = std::max<uint64_t>(pm[1], q_send[q] * 24 + q_recv[q] * 4);
            pm[2] = std::max<uint64_t>(pm[2], l_send[q] * 24 + l_recv[q] * 4);
            pm[6] = std::max<uint64_t>(pm[6], f_send[q] * 32 + b_send[q]);
        }
        // all-gathers: every other rank gets the same piece
        pm[3] = W > 1 ? F * 12 : 0;
        pm[4] = W > 1 ? exch_spl / W : 0;
        pm[5] = pair_max_rank;
    }
    for (int q = 0; q < 7; ++q) out->phase_ms[q] = tm.ms(q, q + 1);
    out->phase_ms[7] = tm.ms(0, 7);
    for (int q = 0; q < 5; ++q) out->join_ms[q] = jt.ms(q, q + 1);
    out->count_kernel_ms = fr.count_kernel_ms;
    out->host_syncs = (uint32_t)(snk_sync_count() - syncs0);
    out->world = W; out->rank = me;
    (void)flags;
    return SNK_OK;
}

void shard_host_free(void* p) { delete static_cast<shard_host*>(p); }

}  // namespace

int snk_unitigs_to_host(snk_ctx* ctx, hipStream_t st, uint32_t K, uint64_t U, const uint64_t* d_off, const uint8_t* d_bases, bool by_first_kmer,
                        bool want_image, snk_result* out, char* err, size_t errcap);

namespace {
__global__ void __launch_bounds__(256) shift_offsets_kernel(const uint64_t* __restrict__ in, uint64_t n, uint64_t add, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}
}  // namespace

// The job's unitig set on ONE rank, in the reference's order: every rank packs the unitigs it wrote to 2 bits per base
// (what crosses xGMI), `root` receives them, orders the union by BVComp on its device and gets host arrays -- or the bytes of the
// .bv hand-off file (SNK_F_BV_IMAGE) that MAIN_ASM_SN writes (lib/tada/src/cmd_main_asm.rs:184-193, debruijn.rs:895-929).
extern "C" int snk_shard_gather_unitigs(snk_ctx* ctx, snk_comm* comm, const snk_shard_result* res, uint32_t K, uint32_t root, uint32_t flags,
                                        snk_result* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !comm || !res || !out || root >= comm->world) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_gather_unitigs: bad argument");
    SNK_HIP_TRY(snk_enter(ctx));
    memset(out, 0, sizeof *out);
    if (!ctx->shard_host) { ctx->shard_host = new shard_host(); ctx->shard_host_free = shard_host_free; }
    shard_host& H = *static_cast<shard_host*>(ctx->shard_host);
    step_ctx X;
    X.ctx = ctx; X.comm = comm; X.st = stream ? (hipStream_t)stream : ctx->stream; X.err = err; X.errcap = errcap;
    X.W = comm->world; X.me = comm->rank;
    const size_t need = 64ull * (X.W + 2) + 4096;
    if (H.pin_cap < need) {
        if (H.pin) (void)hipHostFree(H.pin);
        H.pin = nullptr; H.pin_cap = 0;
        SNK_HIP_TRY(hipHostMalloc((void**)&H.pin, need * 8, hipHostMallocDefault));
        H.pin_cap = need;
    }
    X.pin = H.pin; X.pin_cap = H.pin_cap; X.pin_used = 0;
    hipStream_t st = X.st;
    const uint32_t W = X.W, me = X.me;
    auto body = [&]() -> int {
        const uint64_t U = res->n_unitigs, TB = res->unitig_total_bases;
        const uint64_t pb = ((TB + 15) / 16) * 4;
        uint8_t* packed;
        ALLOC(packed, uint8_t, pb + 32);
        if (TB) TRY(snk_dev_pack2(ctx, res->unitig_bases, TB, packed, st));
        ull mine[2] = {U, TB}, *d_mine;
        TRY(upload(X, mine, 2, &d_mine));
        std::vector<ull> all;
        TRY(exchange_counts(X, d_mine, 2, all));
        std::vector<uint64_t> sbeg(W, 0), scnt(W, 0), rbeg(W, 0), rcnt(W, 0);
        uint64_t Ut = 0, TBt = 0, pbt = 0;
        std::vector<uint64_t> ubase(W + 1, 0), bbase(W + 1, 0), pbase(W + 1, 0);
        for (uint32_t q = 0; q < W; ++q) {
            ubase[q + 1] = ubase[q] + all[2 * q] + 1;                       // every source ships its U+1 offsets
            bbase[q + 1] = bbase[q] + all[2 * q + 1];
            pbase[q + 1] = pbase[q] + ((all[2 * q + 1] + 15) / 16) * 4;
        }
        Ut = ubase[W] - W; TBt = bbase[W]; pbt = pbase[W];
        uint64_t* off_in = nullptr;
        uint8_t* pk_in = nullptr;
        if (me == root) { ALLOC(off_in, uint64_t, ubase[W] + 2); ALLOC(pk_in, uint8_t, pbt + 32); }
        // offsets, then packed bases: everything flows to `root`
        scnt[root] = (U + 1) * 8;
        if (me == root) for (uint32_t q = 0; q < W; ++q) { rbeg[q] = ubase[q] * 8; rcnt[q] = (all[2 * q] + 1) * 8; }
        TRY(comm->a2a(res->unitig_off, sbeg.data(), scnt.data(), off_in, rbeg.data(), rcnt.data(), st, err, errcap));
        scnt[root] = pb;
        if (me == root) for (uint32_t q = 0; q < W; ++q) { rbeg[q] = pbase[q]; rcnt[q] = ((all[2 * q + 1] + 15) / 16) * 4; }
        TRY(comm->a2a(packed, sbeg.data(), scnt.data(), pk_in, rbeg.data(), rcnt.data(), st, err, errcap));
        if (me != root) { SNK_HIP_TRY(snk_sync(st)); return SNK_OK; }
        uint8_t* bases;
        uint64_t* off;
        ALLOC(bases, uint8_t, TBt + 32);
        ALLOC(off, uint64_t, Ut + 2);
        uint64_t uacc = 0;
        for (uint32_t q = 0; q < W; ++q) {
            const uint64_t uq = all[2 * q], tq = all[2 * q + 1];
            if (tq) TRY(snk_dev_unpack2(ctx, pk_in + pbase[q], tq, bases + bbase[q], st));
            // a source's offsets start at 0: shift them to where its bases landed (the last source also writes the closing offset)
            hipLaunchKernelGGL(shift_offsets_kernel, dim3((unsigned)((uq + 1 + 255) / 256)), dim3(256), 0, st, off_in + ubase[q], uq + (q + 1 == W ? 1 : 0), bbase[q],
                               off + uacc);
            uacc += uq;
        }
        SNK_HIP_TRY(hipGetLastError());
        return snk_unitigs_to_host(ctx, st, K, Ut, off, bases, false, (flags & SNK_F_BV_IMAGE) != 0, out, err, errcap);
    };
    int rc = body();
    if (rc) comm->abort();
    return rc;
}

static int shard_step_run(snk_ctx* ctx, snk_comm* comm, const snk_dev_reads* in, const snk_params* p, uint64_t total_reads, uint32_t flags, snk_shard_result* out,
                          void* stream, bool streamed, char* err, size_t errcap);
extern "C" int snk_shard_step(snk_ctx* ctx, snk_comm* comm, const snk_dev_reads* in, const snk_params* p, uint64_t total_reads, uint32_t flags,
                              snk_shard_result* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !comm || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_step: NULL argument");
    snk_opts_enter(&ctx->opts);
    return shard_step_run(ctx, comm, in, p, total_reads, flags, out, stream, false, err, errcap);
}

// ---- the same step with the rank's reads arriving slab by slab (snk_dev_stream_* of the one-GPU path, under the N-GPU step): begin sizes
// the job's buckets from the job-wide read total (every rank computes the same count: same rule, same figures, same group history) and
// this rank's slots from its own upper bound; append partitions a slab (nothing is waited for: the slab's buffers are free when the
// stream has passed the launch); finish runs the rest of the step -- histograms, exchange, count, prune, fragments, join.
extern "C" int snk_shard_stream_begin(snk_ctx* ctx, snk_comm* comm, const snk_params* p, uint32_t read_len, uint64_t rank_reads_ub, uint64_t total_reads, int has_bc,
                                      void* stream, char* err, size_t errcap) {
    if (!ctx || !comm || !p) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_begin: NULL argument");
    if (p->flags & SNK_F_GROUPED) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_shard_stream_begin: per-group graphs shard by group (replicas), not by minimiser");
    if (total_reads == 0 && p->n_buckets == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_begin: the job's read total (or n_buckets) is needed: the ranks size the buckets alike from it");
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint32_t W = comm->world, K = p->K;
    ctx->arena_legacy = W > 1;
    ctx->count_tight = 0;
    ctx->count_screen = 0;
    snk_set_mlen(ctx, p);
    const uint64_t kpr = read_len >= K ? read_len - K + 1 : 0;
    const uint64_t inst_ub = total_reads * kpr;
    const bool adaptive = inst_ub && !snk_opt_is_set("target_inst") && snk_opt_u32("adaptive_buckets", 1) != 0;
    const bool have_ratio = inst_ub && comm->claim_ratio > 0.0 && comm->claim_ratio_reads == inst_ub && comm->claim_ratio_k == K * 2 + 256u * ctx->mlen;
    const uint32_t NB_total = plan_buckets(inst_ub, W, K, p->n_buckets, adaptive && have_ratio ? comm->claim_ratio : 0.0);
    return snk_shard_job_open(ctx, p, comm->rank, W, NB_total, read_len, rank_reads_ub, total_reads, has_bc, st, err, errcap);
}
extern "C" int snk_shard_stream_append(snk_ctx* ctx, const snk_dev_reads* slab, void* stream, char* err, size_t errcap) {
    if (!ctx || !slab) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_append: NULL argument");
    snk_opts_enter(&ctx->opts);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    return snk_shard_job_add(ctx, slab, st, err, errcap);
}
extern "C" int snk_shard_stream_finish(snk_ctx* ctx, snk_comm* comm, uint32_t flags, snk_shard_result* out, void* stream, char* err, size_t errcap) {
    if (!ctx || !comm || !out || !ctx->shard) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_finish: NULL argument / no open step");
    snk_shard_state* S = snk_shard_state_of(ctx);
    if (!S->job_open) return snk_fail(SNK_E_ARG, err, errcap, "snk_shard_stream_finish: no open step (snk_shard_stream_begin)");
    snk_dev_reads in;
    memset(&in, 0, sizeof in);
    in.n_reads = S->job.n_reads; in.read_len = S->job_read_len; in.row_words = (S->job_read_len + 15) / 16;
    in.bc = S->job_has_bc ? (const void*)S->job_good_len : nullptr;       // (only asked whether there are barcodes)
    const snk_params p = S->params;
    return shard_step_run(ctx, comm, &in, &p, S->job_total_reads, flags, out, stream, true, err, errcap);
}

static int shard_step_run(snk_ctx* ctx, snk_comm* comm, const snk_dev_reads* in, const snk_params* p, uint64_t total_reads, uint32_t flags, snk_shard_result* out,
                          void* stream, bool streamed, char* err, size_t errcap) {
    if (p->flags & SNK_F_GROUPED) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_shard_step: per-group graphs shard by group (replicas), not by minimiser");
    SNK_HIP_TRY(snk_enter(ctx));
    if (!ctx->shard_host) { ctx->shard_host = new shard_host(); ctx->shard_host_free = shard_host_free; }
    shard_host& H = *static_cast<shard_host*>(ctx->shard_host);
    step_ctx X;
    X.ctx = ctx; X.comm = comm; X.st = stream ? (hipStream_t)stream : ctx->stream; X.err = err; X.errcap = errcap;
    X.W = comm->world; X.me = comm->rank; X.streamed = streamed;
    const size_t need = 64ull * (X.W + 2) + 4096;
    if (H.pin_cap < need) {
        if (H.pin) (void)hipHostFree(H.pin);
        H.pin = nullptr; H.pin_cap = 0;
        SNK_HIP_TRY(hipHostMalloc((void**)&H.pin, need * 8, hipHostMallocDefault));
        H.pin_cap = need;
    }
    X.pin = H.pin; X.pin_cap = H.pin_cap; X.pin_used = 0;
    int rc = step_impl(X, H, in, p, total_reads, flags, out);
    if (rc) {
        comm->abort();
        (void)hipStreamSynchronize(X.st);
        if (H.cstream) (void)hipStreamSynchronize(H.cstream);
    }
    return rc;
}
