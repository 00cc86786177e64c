// snk_ingest.hip -- f3: barcode ids of the reads on the device.
//
// Replaces BcIndexer (lib/tada/src/utils.rs:101-164): the whitelist file maps line -> index (a HashMap insert per
// line, so the LAST of two identical lines wins); a read's barcode field "SEQ-gg[,raw]" gets
//     id = index(SEQ) + 1 + (gg - 1) * num_bcs      (gg = 1 when there is no "-gg"; 0 = not on the whitelist)
// (get_bc_parts :129-143, get_bc_id :150-163; the FASTH reader keeps the part before the first ',',
// lib/tada/src/multifastq.rs:72-126).  The whitelist is kept in HBM as (64-bit hash, index) pairs sorted by hash plus
// the line bytes for an exact comparison -- lines are arbitrary strings in the reference, not only ACGT words -- and
// every read does one binary search.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"

struct snk_bc_index {
    uint32_t n = 0;             // distinct whitelist strings
    uint32_t num_bcs = 0;       // lines of the whitelist (the id stride between gem groups)
    uint64_t* d_hash = nullptr; // [n] ascending
    uint32_t* d_line = nullptr; // [n] index of the (last) line holding that string
    uint8_t* d_text = nullptr;  // [n][32] the strings, zero padded, same order
    uint32_t* d_err = nullptr;  // [2] malformed gem group / id overflow
};

namespace {

constexpr int BC_MAXLEN = 32;

SNK_HD uint64_t bc_hash(const uint8_t* s, uint32_t len) {        // FNV-1a, then a finaliser
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < len; ++i) { h ^= s[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    return h;
}

__global__ void __launch_bounds__(256) bc_ids_kernel(const uint8_t* __restrict__ fields, uint32_t stride, uint64_t n_reads,
                                                     const uint64_t* __restrict__ hs, const uint32_t* __restrict__ line,
                                                     const uint8_t* __restrict__ text, uint32_t n, uint32_t num_bcs,
                                                     int32_t* __restrict__ ids, uint32_t* __restrict__ errs) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t* f = fields + r * stride;
    uint8_t s[BC_MAXLEN];
    uint32_t len = 0, p = 0;
    // SEQ = the field up to the first '-' (or ',' or the end)
    for (; p < stride; ++p) {
        const uint8_t c = f[p];
        if (c == 0 || c == ',' || c == '-' || c == '\n') break;
        if (len < BC_MAXLEN) s[len] = c;
        ++len;
    }
    uint32_t gg = 1;
    if (p < stride && f[p] == '-') {            // gem group: decimal u8 up to the next '-', ',' or the end
        uint32_t v = 0, nd = 0;
        bool bad = false;
        ++p;
        if (p < stride && f[p] == '+') ++p;        // u8::from_str takes one leading '+' (core::num: "+1" parses, "-1" and " 1" do not)
        for (; p < stride; ++p) {
            const uint8_t c = f[p];
            if (c == 0 || c == ',' || c == '-' || c == '\n') break;
            if (c < '0' || c > '9') bad = true;
            else { v = v * 10 + (c - '0'); ++nd; if (v > 255) bad = true; }
        }
        if (bad || nd == 0) { atomicAdd(&errs[0], 1u); ids[r] = 0; return; }
        gg = v;
    }
    int32_t id = 0;
    if (len <= BC_MAXLEN && n) {
        const uint64_t h = bc_hash(s, len);
        uint32_t lo = 0, hi = n;                 // first entry with hash >= h
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (hs[mid] < h) lo = mid + 1; else hi = mid; }
        for (uint32_t e = lo; e < n && hs[e] == h; ++e) {
            const uint8_t* t = text + (uint64_t)e * BC_MAXLEN;
            bool same = true;
            for (uint32_t i = 0; i < BC_MAXLEN; ++i) { const uint8_t c = i < len ? s[i] : 0; if (t[i] != c) { same = false; break; } }
            if (same) {
                if (gg == 0) { atomicAdd(&errs[1], 1u); break; }             // (g - 1) underflows in the reference: panic
                const uint64_t v = (uint64_t)(gg - 1) * num_bcs + line[e] + 1ull;
                if (v > 0x7FFFFFFFull) atomicAdd(&errs[1], 1u);
                else id = (int32_t)v;
                break;
            }
        }
    }
    ids[r] = id;
}

}  // namespace

extern "C" int snk_bc_index_create(snk_ctx* ctx, const char* whitelist, size_t bytes, snk_bc_index** out, char* err, size_t errcap) {
    if (!ctx || !out || (bytes && !whitelist)) return snk_fail(SNK_E_ARG, err, errcap, "snk_bc_index_create: NULL argument");
    *out = nullptr;
    SNK_HIP_TRY(snk_enter(ctx));
    struct ent { uint64_t h; uint32_t line; std::string s; };
    std::vector<ent> v;
    uint32_t nlines = 0;
    size_t a = 0;
    while (a < bytes) {                                            // BufRead::lines(): split at '\n', drop one trailing '\r'
        size_t b = a;
        while (b < bytes && whitelist[b] != '\n') ++b;
        size_t e = b;
        if (e > a && whitelist[e - 1] == '\r') --e;
        if (e - a > (size_t)BC_MAXLEN) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "whitelist line %u is longer than %d bytes", nlines, BC_MAXLEN);
        std::string s(whitelist + a, e - a);
        v.push_back({bc_hash((const uint8_t*)s.data(), (uint32_t)s.size()), nlines, s});
        ++nlines;
        a = b + 1;
    }
    // identical lines: the last index wins (HashMap::insert)
    std::sort(v.begin(), v.end(), [](const ent& x, const ent& y) { return x.h != y.h ? x.h < y.h : (x.s != y.s ? x.s < y.s : x.line < y.line); });
    std::vector<ent> u;
    for (size_t i = 0; i < v.size(); ++i) {
        if (i + 1 < v.size() && v[i + 1].h == v[i].h && v[i + 1].s == v[i].s) continue;
        u.push_back(v[i]);
    }
    snk_bc_index* ix = new snk_bc_index();
    ix->n = (uint32_t)u.size();
    ix->num_bcs = nlines;
    std::vector<uint64_t> hh(u.size());
    std::vector<uint32_t> ll(u.size());
    std::vector<uint8_t> tt(u.size() * BC_MAXLEN + 16, 0);
    for (size_t i = 0; i < u.size(); ++i) { hh[i] = u[i].h; ll[i] = u[i].line; memcpy(&tt[i * BC_MAXLEN], u[i].s.data(), u[i].s.size()); }
    hipError_t e1 = hipMalloc((void**)&ix->d_hash, std::max<size_t>(hh.size() * 8, 16));
    hipError_t e2 = hipMalloc((void**)&ix->d_line, std::max<size_t>(ll.size() * 4, 16));
    hipError_t e3 = hipMalloc((void**)&ix->d_text, tt.size());
    hipError_t e4 = hipMalloc((void**)&ix->d_err, 16);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        snk_bc_index_destroy(ix);
        return snk_fail(SNK_E_NOMEM, err, errcap, "snk_bc_index_create: hipMalloc failed");
    }
    if (!hh.empty()) {
        SNK_HIP_TRY(hipMemcpy(ix->d_hash, hh.data(), hh.size() * 8, hipMemcpyHostToDevice));
        SNK_HIP_TRY(hipMemcpy(ix->d_line, ll.data(), ll.size() * 4, hipMemcpyHostToDevice));
    }
    SNK_HIP_TRY(hipMemcpy(ix->d_text, tt.data(), tt.size(), hipMemcpyHostToDevice));
    *out = ix;
    return SNK_OK;
}

extern "C" void snk_bc_index_destroy(snk_bc_index* ix) {
    if (!ix) return;
    if (ix->d_hash) (void)hipFree(ix->d_hash);
    if (ix->d_line) (void)hipFree(ix->d_line);
    if (ix->d_text) (void)hipFree(ix->d_text);
    if (ix->d_err) (void)hipFree(ix->d_err);
    delete ix;
}

extern "C" uint32_t snk_bc_index_lines(const snk_bc_index* ix) { return ix ? ix->num_bcs : 0; }

extern "C" int snk_dev_bc_ids(snk_ctx* ctx, const snk_bc_index* ix, const void* d_fields, uint32_t stride, uint64_t n_reads, void* d_ids,
                              void* stream, char* err, size_t errcap) {
    if (!ctx || !ix || (n_reads && (!d_fields || !d_ids))) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_bc_ids: NULL argument");
    if (stride == 0 || stride > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_bc_ids: stride must be 1..256");
    if (n_reads == 0) return SNK_OK;
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    SNK_HIP_TRY(hipMemsetAsync(ix->d_err, 0, 16, st));
    hipLaunchKernelGGL(bc_ids_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d_fields, stride, n_reads,
                       ix->d_hash, ix->d_line, ix->d_text, ix->n, ix->num_bcs, (int32_t*)d_ids, ix->d_err);
    SNK_HIP_TRY(hipGetLastError());
    uint32_t h_err[2] = {0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h_err, ix->d_err, 8, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if (h_err[0]) return snk_fail(SNK_E_ARG, err, errcap, "invalid gem group string in %u barcode fields (utils.rs:138)", h_err[0]);
    if (h_err[1]) return snk_fail(SNK_E_ARG, err, errcap, "too many gem groups - BC id overflowed (%u fields, utils.rs:157)", h_err[1]);
    return SNK_OK;
}
