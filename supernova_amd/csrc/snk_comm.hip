// snk_comm.hip -- transports of the minimiser-sharded path behind the C ABI (snk_comm_* in include/snk.h).
//
//  * RCCL: one process per GPU.  librccl is bound at run time (dlopen) so that libsnk links no second HIP runtime: in a
//    torch process the library torch ships is the one to use (supernova_amd/lib.py passes its path), in a plain C++ host
//    /opt/rocm's.  Every exchange is ONE group of ncclSend / ncclRecv pairs on the caller's stream -- xGMI is point-to-point,
//    an all-to-all is W-1 concurrent pair transfers, which is exactly what a send/recv group expresses -- in pieces of at
//    most 256 MiB (RCCL 2.26 mis-delivers multi-GB messages, tools/dbg_a2a.py), the piece a rank owes itself is a device
//    copy.  The communicator comes from a 128-byte unique id that rank 0 makes and the host hands round (file, socket,
//    torch.distributed broadcast: the host's business), or is adopted from the caller (an ncclComm_t).
//  * local: W in-process ranks on ONE device (every rank a host thread with its own context), the wire replaced by device
//    copies between the ranks' buffers.  The SPMD code of the step is the same; this is how N > 1 is tested on one GPU.
// Reference counterpart of the exchange: shard files written and gathered per shard id
// (lib/tada/external/rust-shardio/src/shard.rs:184-211,488-493), MapReduceEngine.h:362-385.
#include <dlfcn.h>
#include <stdlib.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>      // types and enums only: the functions are bound with dlsym

#include "snk_comm.h"

namespace {

constexpr size_t PIECE = 256ull << 20;

struct rccl_api {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};
rccl_api g_rccl;
std::string g_rccl_path;
std::mutex g_rccl_mu;

int rccl_load(char* err, size_t errcap) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return SNK_OK;
    std::vector<std::string> cands;
    if (!g_rccl_path.empty()) cands.push_back(g_rccl_path);
    if (const char* e = getenv("SNK_RCCL_LIB")) if (*e) cands.push_back(e);
    cands.push_back("librccl.so.1");
    cands.push_back("librccl.so");
    cands.push_back(std::string(getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm") + "/lib/librccl.so");
    void* h = nullptr;
    std::string tried;
    for (auto& c : cands) {
        h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += c + " ";
    }
    if (!h) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "librccl not found (tried: %s)", tried.c_str());
    rccl_api a;
    a.h = h;
#define BIND(f, name) do { *(void**)(&a.f) = dlsym(h, name); if (!a.f) { dlclose(h); return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "librccl has no %s", name); } } while (0)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
    BIND(AllGather, "ncclAllGather");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    *(void**)(&a.CommAbort) = dlsym(h, "ncclCommAbort");      // optional
    g_rccl = a;
    return SNK_OK;
}

#define NCCL_TRY(expr)                                                                                                       \
    do {                                                                                                                     \
        ncclResult_t _r = (expr);                                                                                            \
        if (_r != ncclSuccess) return snk_fail(SNK_E_HIP, err, errcap, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

struct rccl_comm : snk_comm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    unsigned long long* h_pin = nullptr;    // pinned landing area of gather_counts
    unsigned long long* d_all = nullptr;
    size_t cap = 0;                          // u64 entries both hold
    const char* kind() const override { return "rccl"; }
    ~rccl_comm() override {
        if (comm && owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
        if (h_pin) (void)hipHostFree(h_pin);
        if (d_all) (void)hipFree(d_all);
    }
    int a2a(const void* send, const uint64_t* sbeg, const uint64_t* scnt, void* recv, const uint64_t* rbeg, const uint64_t* rcnt, hipStream_t st, char* err,
            size_t errcap) override {
        ++n_collectives;
        const char* s = (const char*)send;
        char* r = (char*)recv;
        if (scnt[rank] != rcnt[rank]) return snk_fail(SNK_E_INTERNAL, err, errcap, "a2a: the piece for myself has two sizes");
        if (scnt[rank]) SNK_HIP_TRY(hipMemcpyAsync(r + rbeg[rank], s + sbeg[rank], scnt[rank], hipMemcpyDeviceToDevice, st));
        bool any = false;
        for (uint32_t p = 0; p < world; ++p) if (p != rank && (scnt[p] || rcnt[p])) any = true;
        if (!any) return SNK_OK;
        NCCL_TRY(g_rccl.GroupStart());
        // a group that was opened is closed on every path: a communicator left in group mode would swallow the next collective of
        // this rank while the peers wait in theirs
        ncclResult_t bad = ncclSuccess;
        const char* what = "";
        for (uint32_t p = 0; p < world && bad == ncclSuccess; ++p) {
            if (p == rank) continue;
            for (uint64_t o = 0; o < scnt[p] && bad == ncclSuccess; o += PIECE) {
                const size_t n = (size_t)(scnt[p] - o < PIECE ? scnt[p] - o : PIECE);
                bad = g_rccl.Send(s + sbeg[p] + o, n, ncclUint8, (int)p, comm, st);
                what = "ncclSend";
            }
            for (uint64_t o = 0; o < rcnt[p] && bad == ncclSuccess; o += PIECE) {
                const size_t n = (size_t)(rcnt[p] - o < PIECE ? rcnt[p] - o : PIECE);
                bad = g_rccl.Recv(r + rbeg[p] + o, n, ncclUint8, (int)p, comm, st);
                what = "ncclRecv";
            }
            bytes_sent += scnt[p];
        }
        const ncclResult_t ge = g_rccl.GroupEnd();
        if (bad != ncclSuccess) return snk_fail(SNK_E_HIP, err, errcap, "%s failed inside an exchange: %s", what, g_rccl.GetErrorString(bad));
        if (ge != ncclSuccess) return snk_fail(SNK_E_HIP, err, errcap, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(ge));
        return SNK_OK;
    }
    // a rank that fails between collectives cannot release peers that already wait inside one: ncclCommAbort tears the
    // communicator down on THIS rank (its queued operations fail instead of waiting), the launcher has to end the job
    void abort() override {
        if (comm && g_rccl.CommAbort) { (void)g_rccl.CommAbort(comm); comm = nullptr; }
    }
    int allgatherv(const void* send, const uint64_t* counts, void* recv, hipStream_t st, char* err, size_t errcap) override {
        std::vector<uint64_t> sbeg(world, 0), scnt(world, counts[rank]), rbeg(world), rcnt(world);
        uint64_t acc = 0;
        for (uint32_t p = 0; p < world; ++p) { rbeg[p] = acc; rcnt[p] = counts[p]; acc += counts[p]; }
        return a2a(send, sbeg.data(), scnt.data(), recv, rbeg.data(), rcnt.data(), st, err, errcap);
    }
    int gather_counts(const unsigned long long* d_mine, uint32_t k, unsigned long long* h_all, hipStream_t st, char* err, size_t errcap) override {
        ++n_collectives;
        const size_t need = (size_t)world * k;
        if (need > cap) {
            if (h_pin) (void)hipHostFree(h_pin);
            if (d_all) (void)hipFree(d_all);
            h_pin = nullptr; d_all = nullptr;
            cap = need + 256;
            SNK_HIP_TRY(hipHostMalloc((void**)&h_pin, cap * 8, hipHostMallocDefault));
            SNK_HIP_TRY(hipMalloc((void**)&d_all, cap * 8));
        }
        if (world == 1) SNK_HIP_TRY(hipMemcpyAsync(h_pin, d_mine, (size_t)k * 8, hipMemcpyDeviceToHost, st));
        else {
            NCCL_TRY(g_rccl.AllGather(d_mine, d_all, k, ncclUint64, comm, st));
            SNK_HIP_TRY(hipMemcpyAsync(h_pin, d_all, need * 8, hipMemcpyDeviceToHost, st));
            bytes_sent += (uint64_t)(world - 1) * k * 8;
        }
        SNK_HIP_TRY(snk_sync(st));
        memcpy(h_all, h_pin, need * 8);
        return SNK_OK;
    }
    int barrier(hipStream_t st, char* err, size_t errcap) override {
        // a stream-ordered rendezvous: an all-gather of one word (nothing on this path needs a host-side barrier)
        if (world == 1) return SNK_OK;
        if (!d_all) { cap = 256 + world; SNK_HIP_TRY(hipHostMalloc((void**)&h_pin, cap * 8, hipHostMallocDefault)); SNK_HIP_TRY(hipMalloc((void**)&d_all, cap * 8)); }
        NCCL_TRY(g_rccl.AllGather(d_all + world, d_all, 1, ncclUint64, comm, st));
        return SNK_OK;
    }
};

// ---------------------------------------------------------------------------------------------------- in-process ranks
struct local_world {
    uint32_t W;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t gen = 0;
    bool aborted = false;
    uint32_t refs;
    std::vector<const void*> sptr;
    std::vector<const uint64_t*> sbeg, scnt;
    std::vector<const unsigned long long*> dvals;
    explicit local_world(uint32_t w) : W(w), refs(w), sptr(w), sbeg(w), scnt(w), dvals(w) {}
    bool wait() {       // false: somebody aborted
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g = gen;
        if (++arrived == W) { arrived = 0; ++gen; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
};

struct local_comm : snk_comm {
    local_world* w = nullptr;
    const char* kind() const override { return "local"; }
    ~local_comm() override {
        bool last;
        { std::lock_guard<std::mutex> lk(w->mu); last = --w->refs == 0; }
        if (last) delete w;
    }
    void abort() override {
        std::lock_guard<std::mutex> lk(w->mu);
        w->aborted = true;
        w->cv.notify_all();
    }
    int gone(char* err, size_t errcap) { return snk_fail(SNK_E_INTERNAL, err, errcap, "another in-process rank failed"); }
    int a2a(const void* send, const uint64_t* sbeg, const uint64_t* scnt, void* recv, const uint64_t* rbeg, const uint64_t* rcnt, hipStream_t st, char* err,
            size_t errcap) override {
        ++n_collectives;
        SNK_HIP_TRY(hipStreamSynchronize(st));        // what I send is complete (the wire's waits are not the step's read-backs: not counted)
        w->sptr[rank] = send; w->sbeg[rank] = sbeg; w->scnt[rank] = scnt;
        if (!w->wait()) return gone(err, errcap);
        for (uint32_t s = 0; s < world; ++s) {
            const uint64_t n = w->scnt[s][rank];
            if (n != rcnt[s]) { abort(); return snk_fail(SNK_E_INTERNAL, err, errcap, "a2a: rank %u sends %llu bytes, rank %u expects %llu", s, (unsigned long long)n, rank, (unsigned long long)rcnt[s]); }
            if (n) SNK_HIP_TRY(hipMemcpyAsync((char*)recv + rbeg[s], (const char*)w->sptr[s] + w->sbeg[s][rank], n, hipMemcpyDeviceToDevice, st));
            if (s != rank) bytes_sent += scnt[s];
        }
        SNK_HIP_TRY(hipStreamSynchronize(st));
        if (!w->wait()) return gone(err, errcap);     // nobody reuses its send buffer before everyone has copied
        return SNK_OK;
    }
    int allgatherv(const void* send, const uint64_t* counts, void* recv, hipStream_t st, char* err, size_t errcap) override {
        std::vector<uint64_t> sbeg(world, 0), scnt(world, counts[rank]), rbeg(world), rcnt(world);
        uint64_t acc = 0;
        for (uint32_t p = 0; p < world; ++p) { rbeg[p] = acc; rcnt[p] = counts[p]; acc += counts[p]; }
        return a2a(send, sbeg.data(), scnt.data(), recv, rbeg.data(), rcnt.data(), st, err, errcap);
    }
    int gather_counts(const unsigned long long* d_mine, uint32_t k, unsigned long long* h_all, hipStream_t st, char* err, size_t errcap) override {
        ++n_collectives;
        SNK_HIP_TRY(snk_sync(st));                    // the step's read-back
        w->dvals[rank] = d_mine;
        if (!w->wait()) return gone(err, errcap);
        for (uint32_t s = 0; s < world; ++s) SNK_HIP_TRY(hipMemcpy(h_all + (size_t)s * k, w->dvals[s], (size_t)k * 8, hipMemcpyDeviceToHost));
        if (!w->wait()) return gone(err, errcap);
        return SNK_OK;
    }
    int barrier(hipStream_t st, char* err, size_t errcap) override {
        SNK_HIP_TRY(hipStreamSynchronize(st));
        if (!w->wait()) return gone(err, errcap);
        return SNK_OK;
    }
};

}  // namespace

extern "C" int snk_comm_set_rccl_path(const char* path) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    g_rccl_path = path ? path : "";
    return SNK_OK;
}

extern "C" int snk_comm_unique_id(void* id128, char* err, size_t errcap) {
    if (!id128) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_unique_id: NULL");
    int rc = rccl_load(err, errcap);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return SNK_OK;
}

extern "C" int snk_comm_create_rccl(snk_ctx* ctx, const void* id128, uint32_t rank, uint32_t world, snk_comm** out, char* err, size_t errcap) {
    if (!ctx || !id128 || !out || world == 0 || rank >= world) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_create_rccl: bad argument");
    int rc = rccl_load(err, errcap);
    if (rc) return rc;
    SNK_HIP_TRY(snk_enter(ctx));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    rccl_comm* c = new rccl_comm();
    c->rank = rank; c->world = world; c->owned = true;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) { c->comm = nullptr; delete c; return snk_fail(SNK_E_HIP, err, errcap, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return SNK_OK;
}

extern "C" int snk_comm_from_nccl(snk_ctx* ctx, void* nccl_comm, uint32_t rank, uint32_t world, snk_comm** out, char* err, size_t errcap) {
    if (!ctx || !nccl_comm || !out || world == 0 || rank >= world) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_from_nccl: bad argument");
    int rc = rccl_load(err, errcap);
    if (rc) return rc;
    rccl_comm* c = new rccl_comm();
    c->rank = rank; c->world = world; c->owned = false; c->comm = (ncclComm_t)nccl_comm;
    *out = c;
    return SNK_OK;
}

extern "C" int snk_comm_create_local(uint32_t world, snk_comm** out /* [world] */, char* err, size_t errcap) {
    if (!out || world == 0 || world > 0x7FFF) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_create_local: bad argument");
    local_world* w = new local_world(world);
    for (uint32_t r = 0; r < world; ++r) {
        local_comm* c = new local_comm();
        c->rank = r; c->world = world; c->w = w;
        out[r] = c;
    }
    return SNK_OK;
}

extern "C" void snk_comm_destroy(snk_comm* c) { delete c; }
extern "C" uint32_t snk_comm_rank(const snk_comm* c) { return c ? c->rank : 0; }
extern "C" uint32_t snk_comm_world(const snk_comm* c) { return c ? c->world : 0; }
extern "C" const char* snk_comm_kind(const snk_comm* c) { return c ? c->kind() : ""; }
extern "C" void snk_comm_abort(snk_comm* c) { if (c) c->abort(); }

// ---------------------------------------------------------------------------------------------------- caller's transport
// The exchanges handed to callbacks of the host (any transport it has: MPI, sockets, torch.distributed).  The CPU tests drive
// the library's exchange planning over gloo with world_size 2 through this (tests/test_sharded_plumbing.py); buffers are
// whatever the caller's transport understands -- host memory there.
namespace {
struct cb_comm : snk_comm {
    snk_comm_a2a_fn a2a_fn = nullptr;
    snk_comm_gather_fn gather_fn = nullptr;
    void* user = nullptr;
    const char* kind() const override { return "callbacks"; }
    int a2a(const void* send, const uint64_t* sbeg, const uint64_t* scnt, void* recv, const uint64_t* rbeg, const uint64_t* rcnt, hipStream_t, char* err,
            size_t errcap) override {
        ++n_collectives;
        for (uint32_t p = 0; p < world; ++p) if (p != rank) bytes_sent += scnt[p];
        const int rc = a2a_fn(user, send, sbeg, scnt, recv, rbeg, rcnt, world);
        return rc ? snk_fail(SNK_E_INTERNAL, err, errcap, "the caller's all-to-all failed (%d)", rc) : SNK_OK;
    }
    int allgatherv(const void* send, const uint64_t* counts, void* recv, hipStream_t st, char* err, size_t errcap) override {
        std::vector<uint64_t> sbeg(world, 0), scnt(world, counts[rank]), rbeg(world), rcnt(world);
        uint64_t acc = 0;
        for (uint32_t p = 0; p < world; ++p) { rbeg[p] = acc; rcnt[p] = counts[p]; acc += counts[p]; }
        return a2a(send, sbeg.data(), scnt.data(), recv, rbeg.data(), rcnt.data(), st, err, errcap);
    }
    int gather_counts(const unsigned long long* d_mine, uint32_t k, unsigned long long* h_all, hipStream_t st, char* err, size_t errcap) override {
        ++n_collectives;
        // the callback always sees HOST memory: counters that live on the device (inside snk_shard_step) are read back here
        std::vector<unsigned long long> mine(k);
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_mine) == hipSuccess && at.type == hipMemoryTypeDevice) {
            SNK_HIP_TRY(hipMemcpyAsync(mine.data(), d_mine, (size_t)k * 8, hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(snk_sync(st));
            d_mine = mine.data();
        } else (void)hipGetLastError();
        const int rc = gather_fn(user, d_mine, k, h_all, world);
        return rc ? snk_fail(SNK_E_INTERNAL, err, errcap, "the caller's gather failed (%d)", rc) : SNK_OK;
    }
    int barrier(hipStream_t st, char* err, size_t errcap) override {
        unsigned long long one = 1;
        std::vector<unsigned long long> all(world);
        return gather_counts(&one, 1, all.data(), st, err, errcap);
    }
};
uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
}  // namespace

extern "C" int snk_comm_create_callbacks(uint32_t rank, uint32_t world, snk_comm_a2a_fn a2a, snk_comm_gather_fn gather, void* user, snk_comm** out,
                                         char* err, size_t errcap) {
    if (!out || !a2a || !gather || world == 0 || rank >= world) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_create_callbacks: bad argument");
    cb_comm* c = new cb_comm();
    c->rank = rank; c->world = world; c->a2a_fn = a2a; c->gather_fn = gather; c->user = user;
    *out = c;
    return SNK_OK;
}

// Exchange patterns of the sharded step on HOST memory with synthetic contents, checked on every rank: (1) bucket histograms
// (equal pieces), (2) the supermer records in R bucket ranges with the step's own piece planner -- every record names its
// (source, destination, bucket, serial) and must land in its (source, bucket) segment in order --, (3) a query / answer round
// trip through per-destination regions with the counts learnt from a gather, (4) a ragged all-gather.  Works over any
// communicator whose buffers may be host memory (callbacks); returns 0 or the number of the check that failed.
extern "C" int snk_comm_selftest(snk_comm* c, uint64_t seed, uint32_t NBl, uint32_t R, char* err, size_t errcap) {
    if (!c || NBl == 0 || R == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_comm_selftest: bad argument");
    const uint32_t W = c->world, me = c->rank;
    if (R > NBl) R = NBl;
    auto hist_of = [&](uint32_t src, uint32_t dst, uint32_t b) -> uint32_t { return (uint32_t)(mix64(seed ^ ((uint64_t)src << 40) ^ ((uint64_t)dst << 20) ^ b) % 7); };
    // (1) histograms
    std::vector<uint32_t> hist((size_t)W * NBl), hrecv((size_t)W * NBl);
    for (uint32_t p = 0; p < W; ++p) for (uint32_t b = 0; b < NBl; ++b) hist[(size_t)p * NBl + b] = hist_of(me, p, b);
    {
        std::vector<uint64_t> beg(W), cnt(W, (uint64_t)NBl * 4);
        for (uint32_t q = 0; q < W; ++q) beg[q] = (uint64_t)q * NBl * 4;
        int rc = c->a2a(hist.data(), beg.data(), cnt.data(), hrecv.data(), beg.data(), cnt.data(), nullptr, err, errcap);
        if (rc) return rc;
    }
    for (uint32_t s = 0; s < W; ++s) for (uint32_t b = 0; b < NBl; ++b) if (hrecv[(size_t)s * NBl + b] != hist_of(s, me, b)) return snk_fail(1, err, errcap, "selftest: histogram of rank %u, bucket %u", s, b);
    // (2) records in ranges: 32-byte records (src, dst, bucket, serial)
    std::vector<unsigned long long> h_rs(2ull * W * R, 0);
    for (uint32_t p = 0; p < W; ++p) for (uint32_t b = 0; b < NBl; ++b) {
        const uint32_t r = (uint32_t)(((uint64_t)(b + 1) * R - 1) / NBl);        // the range whose bounds NBl*r/R .. NBl*(r+1)/R hold b
        uint32_t rr = r;
        while ((uint64_t)NBl * rr / R > b) --rr;
        while ((uint64_t)NBl * (rr + 1) / R <= b) ++rr;
        h_rs[(size_t)p * R + rr] += hist[(size_t)p * NBl + b];
        h_rs[(size_t)W * R + (size_t)p * R + rr] += hrecv[(size_t)p * NBl + b];
    }
    uint64_t n_send = 0, n_recv = 0;
    for (size_t q = 0; q < (size_t)W * R; ++q) { n_send += h_rs[q]; n_recv += h_rs[(size_t)W * R + q]; }
    std::vector<uint64_t> sendb(4 * n_send + 4), recvb(4 * n_recv + 4, ~0ull);
    {
        uint64_t at = 0;
        for (uint32_t p = 0; p < W; ++p) for (uint32_t b = 0; b < NBl; ++b) for (uint32_t k = 0; k < hist[(size_t)p * NBl + b]; ++k) {
            sendb[4 * at] = me; sendb[4 * at + 1] = p; sendb[4 * at + 2] = b; sendb[4 * at + 3] = k; ++at;
        }
    }
    std::vector<uint64_t> sbeg(W), scnt(W), rbeg(W), rcnt(W);
    for (uint32_t r = 0; r < R; ++r) {
        snk_plan_range_pieces(h_rs.data(), W, R, r, 32, sbeg.data(), scnt.data(), rbeg.data(), rcnt.data());
        int rc = c->a2a(sendb.data(), sbeg.data(), scnt.data(), recvb.data(), rbeg.data(), rcnt.data(), nullptr, err, errcap);
        if (rc) return rc;
    }
    {
        uint64_t at = 0;       // the receive buffer is source-major, buckets ascending: exactly the segment table of the count
        for (uint32_t s = 0; s < W; ++s) for (uint32_t b = 0; b < NBl; ++b) for (uint32_t k = 0; k < hrecv[(size_t)s * NBl + b]; ++k) {
            if (recvb[4 * at] != s || recvb[4 * at + 1] != me || recvb[4 * at + 2] != b || recvb[4 * at + 3] != k)
                return snk_fail(2, err, errcap, "selftest: record %llu of segment (source %u, bucket %u)", (unsigned long long)k, s, b);
            ++at;
        }
        if (at != n_recv) return snk_fail(2, err, errcap, "selftest: record count");
    }
    // (3) queries through per-destination regions, answers back into the same regions
    const uint64_t cap = 64 + mix64(seed ^ me) % 64;
    std::vector<unsigned long long> cur(W + 1), all((size_t)W * (W + 1));
    std::vector<uint64_t> qbuf((size_t)W * cap, 0), ans_back((size_t)W * cap, 0);
    for (uint32_t p = 0; p < W; ++p) {
        const uint64_t n = mix64(seed ^ ((uint64_t)me << 8) ^ p ^ 0x51) % cap;
        for (uint64_t i = 0; i < n; ++i) qbuf[(size_t)p * cap + i] = ((uint64_t)me << 48) | ((uint64_t)p << 32) | i;
        cur[p] = (uint64_t)p * cap + n;
    }
    cur[W] = cap;
    { int rc = c->gather_counts(cur.data(), W + 1, all.data(), nullptr, err, errcap); if (rc) return rc; }
    uint64_t n_in = 0;
    for (uint32_t q = 0; q < W; ++q) {
        const uint64_t cap_q = all[(size_t)q * (W + 1) + W];
        sbeg[q] = (uint64_t)q * cap * 8; scnt[q] = (all[(size_t)me * (W + 1) + q] - (uint64_t)q * cap) * 8;
        rbeg[q] = n_in * 8; rcnt[q] = (all[(size_t)q * (W + 1) + me] - (uint64_t)me * cap_q) * 8;
        n_in += rcnt[q] / 8;
    }
    std::vector<uint64_t> qin(n_in + 1), ans(n_in + 1);
    { int rc = c->a2a(qbuf.data(), sbeg.data(), scnt.data(), qin.data(), rbeg.data(), rcnt.data(), nullptr, err, errcap); if (rc) return rc; }
    for (uint64_t i = 0; i < n_in; ++i) {
        if (((qin[i] >> 32) & 0xFFFF) != me) return snk_fail(3, err, errcap, "selftest: a query for rank %llu arrived here", (unsigned long long)((qin[i] >> 32) & 0xFFFF));
        ans[i] = mix64(qin[i]);
    }
    { int rc = c->a2a(ans.data(), rbeg.data(), rcnt.data(), ans_back.data(), sbeg.data(), scnt.data(), nullptr, err, errcap); if (rc) return rc; }
    for (uint32_t p = 0; p < W; ++p) for (uint64_t i = 0; i < scnt[p] / 8; ++i)
        if (ans_back[(size_t)p * cap + i] != mix64(qbuf[(size_t)p * cap + i])) return snk_fail(3, err, errcap, "selftest: answer %llu from rank %u", (unsigned long long)i, p);
    // (4) ragged all-gather
    std::vector<uint64_t> counts(W);
    uint64_t tot = 0;
    for (uint32_t q = 0; q < W; ++q) { counts[q] = (100 + 37 * q + mix64(seed) % 50) * 8; tot += counts[q]; }
    std::vector<uint64_t> mine(counts[me] / 8), gathered(tot / 8 + 1);
    for (size_t i = 0; i < mine.size(); ++i) mine[i] = mix64(seed ^ ((uint64_t)me << 32) ^ i);
    { int rc = c->allgatherv(mine.data(), counts.data(), gathered.data(), nullptr, err, errcap); if (rc) return rc; }
    {
        size_t at = 0;
        for (uint32_t q = 0; q < W; ++q) for (size_t i = 0; i < counts[q] / 8; ++i, ++at)
            if (gathered[at] != mix64(seed ^ ((uint64_t)q << 32) ^ i)) return snk_fail(4, err, errcap, "selftest: all-gather, rank %u word %zu", q, i);
    }
    return SNK_OK;
}
