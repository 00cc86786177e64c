// snk_api.hip -- context, error plumbing, synthetic read generator entry points of libsnk.
#include <math.h>
#include <stdlib.h>
#include <execinfo.h>
#include <dlfcn.h>
#include <time.h>

#include <algorithm>

#include "snk_ctx.h"
#include "snk_kernels.h"
#include "snk_synth.h"

static thread_local char g_last_error[512] = "";

void snk_set_mlen(snk_ctx* ctx, const snk_params* p) {
    uint32_t m = (p->flags & SNK_F_LONG_MINIMISER) ? (uint32_t)SNK_M_LONG : (uint32_t)SNK_M_OF(p->K);
    if (ctx->opts.set[snk_opt_index("minimiser_len")]) { const long long v = ctx->opts.v[snk_opt_index("minimiser_len")]; if (v == SNK_M_LONG || v == SNK_M_OF(p->K)) m = (uint32_t)v; }
    ctx->mlen = m;
}

void snk_set_error(char* err, size_t errcap, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    if (err && errcap) {
        strncpy(err, g_last_error, errcap - 1);
        err[errcap - 1] = 0;
    }
}
int snk_fail(int code, char* err, size_t errcap, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    if (err && errcap) {
        strncpy(err, g_last_error, errcap - 1);
        err[errcap - 1] = 0;
    }
    return code;
}

static thread_local uint64_t g_syncs = 0;
hipError_t snk_sync_at(hipStream_t st, const char* file, int line) {
    static const bool trace = getenv("SNK_SYNC_TRACE") && *getenv("SNK_SYNC_TRACE") == '1';
    ++g_syncs;
    if (trace) { const char* b = strrchr(file, '/'); fprintf(stderr, "[snk sync %llu] %s:%d\n", (unsigned long long)g_syncs, b ? b + 1 : file, line); }
    return hipStreamSynchronize(st);
}
uint64_t snk_sync_count() { return g_syncs; }

extern "C" const char* snk_version(void) { return "libsnk 0.1 (gfx950)"; }
extern "C" const char* snk_last_error(void) { return g_last_error; }

extern "C" void snk_params_default(snk_params* p) {
    memset(p, 0, sizeof *p);
    p->K = 48;
    p->min_qual = 7;
    p->min_freq = 3;
    p->min_bc = 2;
}

extern "C" int snk_ctx_create(int device, snk_ctx** out, char* err, size_t errcap) {
    if (!out) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return snk_fail(SNK_E_NOGPU, err, errcap,
                        "snk_ctx_create: no HIP device visible (%s); libsnk has no CPU fallback",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_create: device %d of %d", device, n);
    SNK_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    SNK_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return snk_fail(SNK_E_NOGPU, err, errcap, "snk_ctx_create: device %d is %s, libsnk is built for gfx950 only",
                        device, prop.gcnArchName);
    snk_ctx* c = new snk_ctx();
    snk_opts_init(&c->opts);
    if (const char* tv = getenv("SNK_TUNING")) {        // shell tools: "name=value,name=value", applied once, here
        char bad[96] = "";
        if (snk_opts_parse(&c->opts, tv, bad, sizeof bad)) { delete c; return snk_fail(SNK_E_ARG, err, errcap, "SNK_TUNING: cannot apply '%s' (unknown option or not an integer)", bad); }
    }
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    c->device_mem_total = (uint64_t)prop.totalGlobalMem;
    c->plan_mem = c->device_mem_total;
    hipError_t se = hipStreamCreate(&c->stream);   // blocking stream: ordered after work on the legacy default stream
    if (se != hipSuccess) {
        delete c;
        return snk_fail(SNK_E_HIP, err, errcap, "hipStreamCreate failed: %s", hipGetErrorString(se));
    }
    *out = c;
    return SNK_OK;
}

// ---- the growing arena (snk_ctx.h)
namespace {
bool va_reserve(snk_ctx* ctx) {
    size_t sz = (size_t)ctx->device_mem_total;
    sz = (sz + ((size_t)1 << 30) - 1) & ~(((size_t)1 << 30) - 1);
    void* base = nullptr;
    if (hipMemAddressReserve(&base, sz, 0, nullptr, 0) != hipSuccess || !base) { (void)hipGetLastError(); return false; }
    ctx->va_base = (char*)base;
    ctx->va_size = sz;
    return true;
}
bool va_init(snk_ctx* ctx) {
    if (ctx->va_state) return ctx->va_state > 0;
    ctx->va_state = -1;
    // option arena_vmm = 0: the cached hipMalloc blocks of rounds 1-3
    { const int ix = snk_opt_index("arena_vmm"); if (ctx->opts.set[ix] && ctx->opts.v[ix] == 0) return false; }
    if (!va_reserve(ctx)) return false;
    ctx->va_state = 1;
    return true;
}
void va_insert_free(std::vector<snk_ctx::vrange>& fr, size_t off, size_t bytes) {
    size_t i = 0;
    while (i < fr.size() && fr[i].off < off) ++i;
    fr.insert(fr.begin() + (long)i, snk_ctx::vrange{off, bytes});
    if (i + 1 < fr.size() && fr[i].off + fr[i].bytes == fr[i + 1].off) { fr[i].bytes += fr[i + 1].bytes; fr.erase(fr.begin() + (long)i + 1); }
    if (i > 0 && fr[i - 1].off + fr[i - 1].bytes == fr[i].off) { fr[i - 1].bytes += fr[i].bytes; fr.erase(fr.begin() + (long)i); }
}
bool va_trace() { static const bool t = getenv("SNK_ARENA_TRACE") && *getenv("SNK_ARENA_TRACE") >= '1'; return t; }
bool va_trace2() { static const bool t = getenv("SNK_ARENA_TRACE") && *getenv("SNK_ARENA_TRACE") >= '2'; return t; }      // every range handed out / taken back
double va_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + 1e-6 * t.tv_nsec; }
bool va_grow(snk_ctx* ctx, size_t need) {
    const double t0 = va_now();
    struct tr { snk_ctx* c; size_t need; double t0; ~tr() { if (va_trace()) fprintf(stderr, "[snk arena] grow by >= %.2f GB -> mapped %.2f GB (%.2f ms)\n", need / 1073741824.0, c->va_mapped / 1073741824.0, va_now() - t0); } } _t{ctx, need, t0};
    // map `need` more bytes behind what is mapped, in uniform 1-GB chunks at 1-GB aligned addresses (chunks of odd sizes at odd offsets
    // were refused by hipMemMap now and then: round 4's trace)
    constexpr size_t CH = (size_t)1 << 30;
    const size_t n_ch = (need + CH - 1) / CH ? (need + CH - 1) / CH : 1;
    if (ctx->va_sealed || ctx->va_mapped + n_ch * CH > ctx->va_size) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = ctx->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t mapped0 = ctx->va_mapped;
    for (size_t q = 0; q < n_ch; ++q) {
        hipMemGenericAllocationHandle_t h;
        char* at = ctx->va_base + ctx->va_mapped;
        bool ok = hipMemCreate(&h, CH, &prop, 0) == hipSuccess;
        if (ok && hipMemMap(at, CH, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); ok = false; }
        if (ok && hipMemSetAccess(at, CH, &acc, 1) != hipSuccess) { (void)hipMemUnmap(at, CH); (void)hipMemRelease(h); ok = false; }
        if (!ok) {
            (void)hipGetLastError();
            if (ctx->va_mapped > mapped0) va_insert_free(ctx->va_free, mapped0, ctx->va_mapped - mapped0);      // what did get mapped is usable
            return false;
        }
        ctx->va_handles.push_back(h);
        ctx->va_chunk.push_back(CH);
        ctx->va_mapped += CH;
        ctx->cached_bytes += CH;
    }
    va_insert_free(ctx->va_free, mapped0, ctx->va_mapped - mapped0);
    return true;
}
// A virtual address that was unmapped is never mapped again: on this stack a range that was unmapped and re-mapped (other physical
// pages behind the same addresses) was read with garbage by the kernels that followed -- the grouped 150 M-read test failed in its
// 15 M-read stage, only with the growing arena, only after a shrink, never with SNK_ARENA_NOSHRINK=1 (round 4).  So memory goes back
// in two ways only: everything at once, with a NEW reservation for what follows (va_reset: nothing may be live), or the chunks behind
// the last live range with the reservation sealed -- it cannot grow any more and is replaced at the start of the next top-level call.
void va_unmap_from(snk_ctx* ctx, size_t keep) {
    while (!ctx->va_chunk.empty() && ctx->va_mapped - ctx->va_chunk.back() >= keep) {
        const size_t c = ctx->va_chunk.back();
        char* at = ctx->va_base + ctx->va_mapped - c;
        (void)hipMemUnmap(at, c);
        (void)hipMemRelease(ctx->va_handles.back());
        ctx->va_handles.pop_back();
        ctx->va_chunk.pop_back();
        ctx->va_mapped -= c;
        ctx->cached_bytes -= c;
        // the free list loses the tail
        if (!ctx->va_free.empty()) {
            snk_ctx::vrange& t = ctx->va_free.back();
            if (t.off >= ctx->va_mapped) ctx->va_free.pop_back();
            else if (t.off + t.bytes > ctx->va_mapped) t.bytes = ctx->va_mapped - t.off;
        }
    }
}
void va_reset(snk_ctx* ctx) {          // nothing is live
    const double t0 = va_now();
    const size_t before = ctx->va_mapped;
    (void)hipDeviceSynchronize();
    va_unmap_from(ctx, 0);
    char* old_base = ctx->va_base;
    const size_t old_size = ctx->va_size;
    ctx->va_base = nullptr;
    if (!va_reserve(ctx)) ctx->va_state = -1;          // (the cached blocks take over)
    if (old_base) (void)hipMemAddressFree(old_base, old_size);      // after the new reservation: other addresses
    ctx->va_free.clear(); ctx->va_used.clear(); ctx->va_serial.clear();
    ctx->va_mapped = 0; ctx->va_high = 0; ctx->va_sealed = false;
    if (va_trace()) fprintf(stderr, "[snk arena] reset: %.2f GB handed back, new reservation (%.2f ms)\n", before / 1073741824.0, va_now() - t0);
}
void va_shrink(snk_ctx* ctx, size_t keep) {
    if (ctx->va_state <= 0 || ctx->va_mapped == 0) return;
    if (keep == 0 && ctx->va_used.empty()) { va_reset(ctx); return; }
    const size_t before = ctx->va_mapped;
    va_unmap_from(ctx, keep);
    if (ctx->va_mapped != before) {
        ctx->va_sealed = true;
        if (va_trace()) fprintf(stderr, "[snk arena] %.2f -> %.2f GB mapped, reservation sealed\n", before / 1073741824.0, ctx->va_mapped / 1073741824.0);
    }
}
void* va_alloc(snk_ctx* ctx, size_t bytes) {
    if (!va_init(ctx)) return nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (size_t i = 0; i < ctx->va_free.size(); ++i) {
            snk_ctx::vrange& f = ctx->va_free[i];
            if (f.bytes < bytes) continue;
            const size_t off = f.off;
            if (f.bytes == bytes) ctx->va_free.erase(ctx->va_free.begin() + (long)i);
            else { f.off += bytes; f.bytes -= bytes; }
            ctx->va_used.push_back(snk_ctx::vrange{off, bytes});
            if (off + bytes > ctx->va_high) ctx->va_high = off + bytes;
            if (va_trace2()) fprintf(stderr, "[snk arena] A %zu %zu\n", off, bytes);
            return ctx->va_base + off;
        }
        // nothing fits: grow by what is missing behind a free tail
        size_t tail = 0;
        if (!ctx->va_free.empty() && ctx->va_free.back().off + ctx->va_free.back().bytes == ctx->va_mapped) tail = ctx->va_free.back().bytes;
        if (attempt || !va_grow(ctx, bytes - (tail < bytes ? tail : 0))) return nullptr;
    }
    return nullptr;
}
bool va_release(snk_ctx* ctx, const void* p) {
    if (ctx->va_state <= 0 || (const char*)p < ctx->va_base || (const char*)p >= ctx->va_base + ctx->va_size) return false;
    const size_t off = (size_t)((const char*)p - ctx->va_base);
    for (size_t i = 0; i < ctx->va_used.size(); ++i)
        if (ctx->va_used[i].off == off) {
            const size_t b = ctx->va_used[i].bytes;
            ctx->va_used[i] = ctx->va_used.back();
            ctx->va_used.pop_back();
            va_insert_free(ctx->va_free, off, b);
            if (va_trace2()) fprintf(stderr, "[snk arena] R %zu %zu\n", off, b);
            ctx->total_alloc -= b;
            return true;
        }
    if (va_trace2()) fprintf(stderr, "[snk arena] R? %zu (not live)\n", off);
    return true;      // inside the range but not live: already released
}
}  // namespace

// SNK_ARENA_POISON=1 (tests): every block handed out is filled with 0xCD first -- scratch is not zero (a cached block holds the last
// call's data, a fresh mapping whatever the device held), and code that reads a word it never wrote shows up at golden size this way
// instead of at 150 M reads in the third test of a process
static void arena_poison(void* p, size_t bytes) {
    static const bool on = getenv("SNK_ARENA_POISON") && *getenv("SNK_ARENA_POISON") == '1';
    // (waited for: the caller may initialise the block on a non-blocking stream, which the null stream's memset is not ordered with)
    if (on) { (void)hipMemset(p, 0xCD, bytes); (void)hipStreamSynchronize(nullptr); }
}

int snk_ctx_alloc(snk_ctx* ctx, size_t bytes, void** out, char* err, size_t errcap) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    if (ctx->device_mem_total && bytes > 2 * ctx->device_mem_total) {
        // a size no device holds is a sizing bug upstream (an unset counter, an underflow): say where it came from -- return addresses
        // relative to the library's load address, for llvm-symbolizer / objdump on the same build
        void* fr[12];
        const int n = backtrace(fr, 12);
        Dl_info di;
        uintptr_t base = 0;
        if (dladdr((void*)&snk_ctx_alloc, &di) && di.dli_fbase) base = (uintptr_t)di.dli_fbase;
        char where[256];
        size_t w = 0;
        for (int i = 1; i < n && w + 20 < sizeof where; ++i) w += (size_t)snprintf(where + w, sizeof where - w, " +0x%zx", (size_t)((uintptr_t)fr[i] - base));
        return snk_fail(SNK_E_INTERNAL, err, errcap, "scratch request of %zu bytes (device: %zu) from libsnk%s", bytes, (size_t)ctx->device_mem_total, where);
    }
    if (ctx->va_state >= 0 && !ctx->arena_legacy) {
        if (void* q = va_alloc(ctx, bytes)) {
            ++ctx->alloc_serial;
            ctx->va_serial.push_back({(size_t)((char*)q - ctx->va_base), ctx->alloc_serial});
            ctx->total_alloc += bytes;
            if (ctx->total_alloc > ctx->peak_alloc) ctx->peak_alloc = ctx->total_alloc;
            *out = q;
            arena_poison(q, bytes);
            return SNK_OK;
        }
    }
    // best fit among the free cached blocks.  A block may be up to 4x the request: the first call of a context sizes its
    // count regions from a guess (instances / 12), later calls from what the first one retained, and a cached block that
    // no request is allowed to take is memory wasted twice (it stays idle AND a new one is allocated -- a 100+ ms
    // hipMalloc in the second call of every run)
    int best = -1;
    for (size_t i = 0; i < ctx->blocks.size(); ++i) {
        const snk_ctx::block& b = ctx->blocks[i];
        if (!b.used && b.bytes >= bytes && b.bytes <= 4 * bytes + (1u << 20) &&
            (best < 0 || b.bytes < ctx->blocks[best].bytes))
            best = (int)i;
    }
    if (best >= 0) {
        ctx->blocks[best].used = true;
        ctx->blocks[best].serial = ++ctx->alloc_serial;
        ctx->blocks[best].epoch = ctx->call_epoch;
        ctx->total_alloc += ctx->blocks[best].bytes;
        if (ctx->total_alloc > ctx->peak_alloc) ctx->peak_alloc = ctx->total_alloc;
        *out = ctx->blocks[best].p;
        arena_poison(*out, bytes);
        return SNK_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        snk_ctx_trim_cache(ctx);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        return snk_fail(SNK_E_NOMEM, err, errcap, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->blocks.push_back({p, bytes, true, ++ctx->alloc_serial, ctx->call_epoch});
    ctx->total_alloc += bytes;
    if (ctx->total_alloc > ctx->peak_alloc) ctx->peak_alloc = ctx->total_alloc;
    ctx->cached_bytes += bytes;
    *out = p;
    arena_poison(p, bytes);
    return SNK_OK;
}
// hand one block back to the cache in the middle of a call (the big ones: supermer slots after the count, count regions
// after the gather) so that later stages of the same call can reuse the memory
// SNK_ARENA_POISON=2 (tests): a block handed back in the middle of a call is filled with 0xDD IN STREAM ORDER on the stream of the top-level call at hand
// -- the kernels that were entitled to read it are in front of the fill; anything launched later that still reads it reads the
// fill, at golden size, on either arena.
static void release_poison(snk_ctx* ctx, const void* p, size_t bytes) {
    static const bool on = getenv("SNK_ARENA_POISON") && *getenv("SNK_ARENA_POISON") == '2';
    if (on && bytes) (void)hipMemsetAsync(const_cast<void*>(p), 0xDD, bytes, ctx->cur_stream ? ctx->cur_stream : ctx->stream);
}
void snk_ctx_release_block(snk_ctx* ctx, const void* p) {
    if (!p) return;
    if (ctx->va_state > 0 && (const char*)p >= ctx->va_base && (const char*)p < ctx->va_base + ctx->va_size) {
        const size_t off = (size_t)((const char*)p - ctx->va_base);
        for (auto& u : ctx->va_used) if (u.off == off) { release_poison(ctx, p, u.bytes); break; }
    }
    if (va_release(ctx, p)) return;
    for (auto& b : ctx->blocks)
        if (b.p == p && b.used) { release_poison(ctx, p, b.bytes); b.used = false; ctx->total_alloc -= b.bytes; return; }
}
void snk_ctx_release_since(snk_ctx* ctx, uint64_t mark, const void* const* keep, size_t n_keep) {
    if (ctx->va_state > 0) {
        std::vector<size_t> drop;
        for (auto& e : ctx->va_serial) {
            if (e.serial <= mark) continue;
            bool kept = false, live = false;
            for (size_t i = 0; i < n_keep; ++i) if (keep[i] == ctx->va_base + e.off) kept = true;
            for (auto& u : ctx->va_used) if (u.off == e.off) live = true;
            if (!kept && live) drop.push_back(e.off);
        }
        for (size_t off : drop) (void)va_release(ctx, ctx->va_base + off);
    }
    for (auto& b : ctx->blocks) {
        if (!b.used || b.serial <= mark) continue;
        bool kept = false;
        for (size_t i = 0; i < n_keep; ++i) if (keep[i] == b.p) kept = true;
        if (!kept) { b.used = false; ctx->total_alloc -= b.bytes; }
    }
}
void snk_ctx_release_scratch(snk_ctx* ctx) {
    // a streamed job (snk_dev_stream_* / snk_shard_stream_*) keeps its slots, cursors, good lengths and status in this arena: once the
    // arena is recycled by ANOTHER top-level call the job is gone, and its append / finish must say so instead of writing into memory
    // that now belongs to someone else
    if (ctx->stream_job && ctx->stream_job_invalidate) ctx->stream_job_invalidate(ctx->stream_job);
    if (ctx->shard) snk_shard_state_invalidate_job(ctx->shard);
    for (auto& b : ctx->blocks) b.used = false;
    ctx->total_alloc = 0;
    ctx->peak_alloc = 0;
    if (ctx->va_state > 0) {
        if (va_trace()) fprintf(stderr, "[snk arena] call done: mapped %.2f GB, high water %.2f GB, live ranges %zu, free ranges %zu\n", ctx->va_mapped / 1073741824.0,
                                ctx->va_high / 1073741824.0, ctx->va_used.size(), ctx->va_free.size());
        // everything is free again; what the last two calls never reached (a 150 M-read run followed by 15 M-read runs) goes back to
        // the device -- only when the range is more than twice what they used: mapping it again would cost ~30 ms per GB
        ctx->va_used.clear();
        ctx->va_serial.clear();
        ctx->va_free.clear();
        if (ctx->va_mapped) ctx->va_free.push_back(snk_ctx::vrange{0, ctx->va_mapped});
        const size_t recent = std::max(ctx->va_high, std::max(ctx->va_high_prev[0], ctx->va_high_prev[1]));
        ctx->va_high_prev[1] = ctx->va_high_prev[0];
        ctx->va_high_prev[0] = ctx->va_high;
        ctx->va_high = 0;
        // (a call that takes plain hipMalloc blocks -- a multi-rank step -- must not sit next to tens of GB this range still maps: ADVICE r4)
        // (the reservation in whole 1-GB chunks, as it is mapped: a floor of 129.6 GB is 130 GB mapped, which must not count as "more than the floor" --
        // a call that needed less than half the reservation would reset and re-map all of it, ~3 s, every time: found by tools/r6_cap_sigma.sh)
        const size_t floor_chunks = (ctx->va_floor + (((size_t)1 << 30) - 1)) & ~(((size_t)1 << 30) - 1);
        if (ctx->va_sealed || (recent && ctx->va_mapped > std::max(2 * recent + ((size_t)1 << 30), floor_chunks)) || (ctx->arena_legacy && ctx->va_mapped)) {
            va_reset(ctx);
            // the host's reservation (snk_ctx_reserve) outlives a reset: mapped again right away -- except under a call that takes plain
            // hipMalloc blocks (a multi-rank step must not sit next to the reserved range; the next call of the other kind maps it again).
            // A failure here is not an error: the call's own requests grow the arena as far as the device allows
            if (ctx->va_floor && !ctx->arena_legacy && ctx->va_state > 0) (void)va_grow(ctx, ctx->va_floor);
        }
        else if (ctx->va_floor > ctx->va_mapped && !ctx->arena_legacy && ctx->va_state > 0 && !ctx->va_sealed) (void)va_grow(ctx, ctx->va_floor - ctx->va_mapped);
    }
    // A new top-level call.  Blocks that neither of the last two calls took are sizes the caller has moved away from (a
    // 150 M-read run followed by 15 M-read runs): they go back to the device, where the caller's own allocator may need them.
    // Steady state (the same sizes call after call) frees nothing.
    ++ctx->call_epoch;
    bool stale = false;
    for (auto& b : ctx->blocks) if (b.epoch + 2 < ctx->call_epoch) stale = true;
    if (stale) {
        std::vector<snk_ctx::block> keep;
        for (auto& b : ctx->blocks) {
            if (b.epoch + 2 < ctx->call_epoch) { (void)hipFree(b.p); ctx->cached_bytes -= b.bytes; }
            else keep.push_back(b);
        }
        ctx->blocks.swap(keep);
    }
    snk_ctx_plan_mem(ctx);      // what this call's plans may count on (everything the arena holds is free at this point)
}
extern "C" uint32_t snk_ctx_last_partition_passes(const snk_ctx* ctx) { return ctx ? ctx->last_partition_passes : 0u; }
extern "C" uint32_t snk_ctx_last_count_limit(const snk_ctx* ctx) { return ctx ? ctx->last_count_limit : 0u; }
extern "C" int snk_ctx_reserve(snk_ctx* ctx, uint64_t bytes, char* err, size_t errcap) {
    if (!ctx) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_reserve: NULL context");
    SNK_HIP_TRY(snk_enter(ctx));
    if (!va_init(ctx)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_ctx_reserve: the growing arena is switched off (SNK_ARENA_VMM=0) or not available");
    if (ctx->va_sealed && ctx->va_used.empty()) va_reset(ctx);
    if (ctx->va_state <= 0) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_ctx_reserve: no address range for the arena");
    if (bytes > ctx->va_mapped && !va_grow(ctx, bytes - ctx->va_mapped))
        return snk_fail(SNK_E_NOMEM, err, errcap, "snk_ctx_reserve: %.1f GB asked for, %.1f GB mapped", bytes / 1073741824.0, ctx->va_mapped / 1073741824.0);
    ctx->va_floor = bytes;
    return SNK_OK;
}
extern "C" void snk_ctx_trim(snk_ctx* ctx) {
    if (!ctx) return;
    ctx->va_floor = 0;
    (void)snk_enter(ctx);
    (void)hipDeviceSynchronize();
    snk_ctx_trim_cache(ctx);
}
void snk_ctx_plan_mem(snk_ctx* ctx) {
    // what this context can count on: the device's free memory and what its own arena holds (all of it free at the start of a call).  In
    // steps of 8 GB so that the plans of a job do not move with a few MB of somebody's allocations; never more than the device
    size_t fr = 0, tot = 0;
    ctx->plan_mem = ctx->device_mem_total;
    ctx->plan_mapped = (uint64_t)ctx->cached_bytes;
    { const int ix = snk_opt_index("plan_mem_mb"); if (ix >= 0 && ctx->opts.set[ix] && ctx->opts.v[ix] > 0) { ctx->plan_mem = (uint64_t)ctx->opts.v[ix] << 20; return; } }      // (tests: a small device)
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return; }
    uint64_t avail = (uint64_t)fr + (uint64_t)ctx->cached_bytes;      // (cached_bytes: the arena's mapped chunks and the plain blocks)
    avail &= ~((8ull << 30) - 1);
    if (avail < (8ull << 30)) avail = 8ull << 30;
    if (avail < ctx->plan_mem) ctx->plan_mem = avail;
}
void snk_ctx_trim_cache(snk_ctx* ctx) {
    if (ctx->va_state > 0) {
        // unmap everything behind the highest live range
        size_t live_end = 0;
        for (auto& u : ctx->va_used) if (u.off + u.bytes > live_end) live_end = u.off + u.bytes;
        va_shrink(ctx, live_end);
    }
    std::vector<snk_ctx::block> keep;
    for (auto& b : ctx->blocks) {
        if (b.used) keep.push_back(b);
        else { (void)hipFree(b.p); ctx->cached_bytes -= b.bytes; }
    }
    ctx->blocks.swap(keep);
}

extern "C" void snk_ctx_destroy(snk_ctx* ctx) {
    if (!ctx) return;
    (void)snk_enter(ctx);
    snk_ctx_release_scratch(ctx);
    snk_ctx_trim_cache(ctx);
    if (ctx->va_base) { va_unmap_from(ctx, 0); (void)hipMemAddressFree(ctx->va_base, ctx->va_size); ctx->va_base = nullptr; }
    if (ctx->shard) snk_shard_state_free(ctx->shard);
    if (ctx->host_io && ctx->host_io_free) ctx->host_io_free(ctx->host_io);
    if (ctx->df_io && ctx->df_io_free) ctx->df_io_free(ctx->df_io);
    if (ctx->shard_host && ctx->shard_host_free) ctx->shard_host_free(ctx->shard_host);
    if (ctx->stream_job && ctx->stream_job_free) ctx->stream_job_free(ctx->stream_job);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------------------------------ synthetic reads
extern "C" void snk_synth_default(snk_synth_params* sp, uint64_t n_reads, uint64_t seed, int error_free) {
    memset(sp, 0, sizeof *sp);
    sp->seed = seed;
    sp->n_reads = n_reads;
    sp->read_len = 150;
    sp->genome_len = n_reads * 150 / 56;
    if (sp->genome_len < 1000) sp->genome_len = 1000;
    sp->mol_len = 50000;
    sp->mols_per_bc = 10;
    sp->pairs_per_bc = 400;
    sp->insert_min = 300;
    sp->insert_span = 101;
    sp->sub_ppm = error_free ? 0 : 2000;
    sp->unbarcoded_ppm = 20000;
    sp->lowq_tail_ppm = error_free ? 0 : 50000;
    sp->tail_max = 40;
    snk_synth_set_errors(sp, sp->sub_ppm);
}
extern "C" void snk_synth_set_errors(snk_synth_params* sp, uint32_t sub_ppm) {
    // Poisson(lambda = read_len*sub_ppm/1e6) cumulative, scaled to 2^32 (a read carries at most four substitutions)
    sp->sub_ppm = sub_ppm;
    double lam = sp->read_len * (double)sp->sub_ppm / 1e6, term = exp(-lam), cum = 0;
    for (int j = 0; j < 4; ++j) {
        cum += term;
        double v = cum * 4294967296.0;
        sp->err_cdf[j] = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
        term *= lam / (j + 1);
    }
}

extern "C" int snk_synth_host(const snk_synth_params* sp, uint64_t first, uint64_t n, uint32_t* rows, uint32_t row_words,
                              uint8_t* quals, uint32_t qstride, int32_t* bc) {
    if (!sp || sp->read_len == 0 || (rows && row_words * 16 < sp->read_len) || (quals && qstride < sp->read_len))
        return snk_fail(SNK_E_ARG, nullptr, 0, "snk_synth_host: bad arguments");
    for (uint64_t i = 0; i < n; ++i)
        snk_synth_read(*sp, first + i, rows ? rows + i * row_words : nullptr, row_words,
                       quals ? quals + i * (uint64_t)qstride : nullptr, bc ? bc + i : nullptr);
    return SNK_OK;
}

__global__ void __launch_bounds__(256) snk_synth_kernel(snk_synth_params sp, uint64_t first, uint64_t n, uint32_t* rows,
                                                        uint32_t row_words, uint8_t* quals, uint32_t qstride,
                                                        int32_t* bc) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step)
        snk_synth_read(sp, first + i, rows ? rows + i * row_words : nullptr, row_words,
                       quals ? quals + i * (uint64_t)qstride : nullptr, bc ? bc + i : nullptr);
}

extern "C" int snk_synth_dev(snk_ctx* ctx, const snk_synth_params* sp, uint64_t first, uint64_t n, void* d_rows,
                             uint32_t row_words, void* d_quals, uint32_t qstride, void* d_bc, void* stream) {
    char* err = nullptr;
    size_t errcap = 0;
    if (!ctx || !sp) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_dev: NULL ctx/params");
    if ((d_rows && row_words * 16 < sp->read_len) || (d_quals && qstride < sp->read_len))
        return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_dev: row stride too small");
    if (n == 0) return SNK_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    uint64_t nb = (n + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(snk_synth_kernel, dim3((unsigned)nb), dim3(256), 0, st, *sp, first, n, (uint32_t*)d_rows, row_words,
                       (uint8_t*)d_quals, qstride, (int32_t*)d_bc);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
