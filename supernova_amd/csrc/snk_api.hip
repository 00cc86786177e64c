// snk_api.hip -- context, error plumbing, synthetic read generator entry points of libsnk.
#include <math.h>
#include <stdlib.h>

#include "snk_ctx.h"
#include "snk_synth.h"

static thread_local char g_last_error[512] = "";

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
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    c->device_mem_total = (uint64_t)prop.totalGlobalMem;
    hipError_t se = hipStreamCreate(&c->stream);   // blocking stream: ordered after work on the legacy default stream
    if (se != hipSuccess) {
        delete c;
        return snk_fail(SNK_E_HIP, err, errcap, "hipStreamCreate failed: %s", hipGetErrorString(se));
    }
    *out = c;
    return SNK_OK;
}

int snk_ctx_alloc(snk_ctx* ctx, size_t bytes, void** out, char* err, size_t errcap) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
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
    return SNK_OK;
}
// hand one block back to the cache in the middle of a call (the big ones: supermer slots after the count, count regions
// after the gather) so that later stages of the same call can reuse the memory
void snk_ctx_release_block(snk_ctx* ctx, const void* p) {
    if (!p) return;
    for (auto& b : ctx->blocks)
        if (b.p == p && b.used) { b.used = false; ctx->total_alloc -= b.bytes; return; }
}
void snk_ctx_release_since(snk_ctx* ctx, uint64_t mark, const void* const* keep, size_t n_keep) {
    for (auto& b : ctx->blocks) {
        if (!b.used || b.serial <= mark) continue;
        bool kept = false;
        for (size_t i = 0; i < n_keep; ++i) if (keep[i] == b.p) kept = true;
        if (!kept) { b.used = false; ctx->total_alloc -= b.bytes; }
    }
}
void snk_ctx_release_scratch(snk_ctx* ctx) {
    for (auto& b : ctx->blocks) b.used = false;
    ctx->total_alloc = 0;
    ctx->peak_alloc = 0;
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
}
extern "C" void snk_ctx_trim(snk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    snk_ctx_trim_cache(ctx);
}
void snk_ctx_trim_cache(snk_ctx* ctx) {
    std::vector<snk_ctx::block> keep;
    for (auto& b : ctx->blocks) {
        if (b.used) keep.push_back(b);
        else { (void)hipFree(b.p); ctx->cached_bytes -= b.bytes; }
    }
    ctx->blocks.swap(keep);
}

extern "C" void snk_ctx_destroy(snk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    snk_ctx_release_scratch(ctx);
    snk_ctx_trim_cache(ctx);
    if (ctx->shard) snk_shard_state_free(ctx->shard);
    if (ctx->host_io && ctx->host_io_free) ctx->host_io_free(ctx->host_io);
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
    uint64_t nb = (n + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(snk_synth_kernel, dim3((unsigned)nb), dim3(256), 0, st, *sp, first, n, (uint32_t*)d_rows, row_words,
                       (uint8_t*)d_quals, qstride, (int32_t*)d_bc);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}
