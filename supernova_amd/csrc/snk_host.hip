// snk_host.hip -- host-pointer convenience entry point, the .bv hand-off file (the graph-from-unitigs step lives in snk_hbv.hip).
//   snk_count_graph      : buildReadQGraph48 for host-resident reads (lib/assembly/src/paths/long/BuildReadQGraph48.h:24-34)
//   snk_write_bv/read_bv : lib/tada/src/debruijn.rs:895-929 <-> BuildReadQGraph48.cc:1640-1642
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "snk_ctx.h"
#include "snk_common.h"

namespace {

struct dev_buf {
    void* p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
};

// BVComp, HBVFromEdges.cc:106-111: length descending, then lexicographic
struct bv_less {
    const uint64_t* off;
    const uint8_t* b;
    bool operator()(uint64_t x, uint64_t y) const {
        uint64_t lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
        if (lx != ly) return lx > ly;
        int c = memcmp(b + off[x], b + off[y], lx);
        return c < 0;
    }
};

}  // namespace

extern "C" int snk_count_graph(snk_ctx* ctx, const snk_reads* in, const snk_params* p, snk_result* out, char* err, size_t errcap) {
    if (!ctx || !in || !p || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: NULL argument");
    if (!in->ascii && !in->rows) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: need ascii or rows");
    if (!in->quals && !in->good_len) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: need quals or good_len");
    if (in->read_len == 0 || in->read_len > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_count_graph: read_len must be 1..256");
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n = in->n_reads;
    const uint32_t L = in->read_len, rw = (L + 15) / 16;
    dev_buf d_rows, d_ascii, d_quals, d_lens, d_gl, d_bc;
    SNK_HIP_TRY(hipMalloc(&d_rows.p, std::max<size_t>(n * rw * 4, 16)));
    if (in->rows) SNK_HIP_TRY(hipMemcpyAsync(d_rows.p, in->rows, n * rw * 4, hipMemcpyHostToDevice, st));
    else {
        SNK_HIP_TRY(hipMalloc(&d_ascii.p, std::max<size_t>(n * L, 16)));
        SNK_HIP_TRY(hipMemcpyAsync(d_ascii.p, in->ascii, n * L, hipMemcpyHostToDevice, st));
        int rc = snk_dev_pack_ascii(ctx, d_ascii.p, L, L, n, d_rows.p, rw, st);
        if (rc) return snk_fail(rc, err, errcap, "%s", snk_last_error());
    }
    if (in->quals) { SNK_HIP_TRY(hipMalloc(&d_quals.p, std::max<size_t>(n * L, 16))); SNK_HIP_TRY(hipMemcpyAsync(d_quals.p, in->quals, n * L, hipMemcpyHostToDevice, st)); }
    if (in->lens) { SNK_HIP_TRY(hipMalloc(&d_lens.p, std::max<size_t>(n * 2, 16))); SNK_HIP_TRY(hipMemcpyAsync(d_lens.p, in->lens, n * 2, hipMemcpyHostToDevice, st)); }
    if (in->good_len) { SNK_HIP_TRY(hipMalloc(&d_gl.p, std::max<size_t>(n * 2, 16))); SNK_HIP_TRY(hipMemcpyAsync(d_gl.p, in->good_len, n * 2, hipMemcpyHostToDevice, st)); }
    if (in->bc) { SNK_HIP_TRY(hipMalloc(&d_bc.p, std::max<size_t>(n * 4, 16))); SNK_HIP_TRY(hipMemcpyAsync(d_bc.p, in->bc, n * 4, hipMemcpyHostToDevice, st)); }
    snk_dev_reads dr;
    memset(&dr, 0, sizeof dr);
    dr.n_reads = n; dr.rows = d_rows.p; dr.row_words = rw; dr.read_len = L; dr.lens = d_lens.p;
    dr.quals = d_quals.p; dr.qstride = L; dr.good_len = d_gl.p; dr.bc = d_bc.p; dr.ign_bc_below = in->ign_bc_below;
    snk_dev_result r;
    snk_params pp = *p;
    const bool want_table = !(p->flags & SNK_F_NO_TABLE);
    if (!want_table) pp.flags |= SNK_F_UNSORTED_TABLE;          // nobody will look at the table: do not sort it
    int rc = snk_dev_count_graph(ctx, &dr, &pp, &r, st, err, errcap);
    if (rc) return rc;
    out->n_instances = r.n_instances;
    out->n_kmers = r.n_kmers;
    out->spectrum_bins = r.spectrum_bins;
    memcpy(out->phase_ms, r.phase_ms, sizeof out->phase_ms);
    const uint64_t nk = want_table ? r.n_kmers : 0;
    if (want_table) {
        out->kmers = (uint32_t*)malloc(std::max<size_t>(nk * 16, 16));
        out->counts = (uint32_t*)malloc(std::max<size_t>(nk * 4, 16));
        out->ctx = (uint8_t*)malloc(std::max<size_t>(nk, 16));
    }
    out->spectrum = (uint64_t*)malloc(std::max<size_t>((size_t)r.spectrum_bins * 8, 16));
    std::vector<uint64_t> lohi(nk * 2);
    if ((want_table && (!out->kmers || !out->counts || !out->ctx)) || !out->spectrum) { snk_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_count_graph: host allocation failed"); }
    if (nk) {
        SNK_HIP_TRY(hipMemcpyAsync(lohi.data(), r.keys, nk * 16, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->counts, r.counts, nk * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->ctx, r.ctx, nk, hipMemcpyDeviceToHost, st));
    }
    if (r.spectrum_bins) SNK_HIP_TRY(hipMemcpyAsync(out->spectrum, r.spectrum, (size_t)r.spectrum_bins * 8, hipMemcpyDeviceToHost, st));
    std::vector<uint64_t> off(r.n_unitigs + 1, 0);
    std::vector<uint8_t> bases(r.unitig_total_bases);
    if (r.n_unitigs) {
        SNK_HIP_TRY(hipMemcpyAsync(off.data(), r.unitig_off, (r.n_unitigs + 1) * 8, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(bases.data(), r.unitig_bases, r.unitig_total_bases, hipMemcpyDeviceToHost, st));
    }
    SNK_HIP_TRY(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < nk; ++i) {
        uint64_t lo = lohi[2 * i], hi = lohi[2 * i + 1];
        out->kmers[4 * i] = (uint32_t)(hi >> 32); out->kmers[4 * i + 1] = (uint32_t)hi;
        out->kmers[4 * i + 2] = (uint32_t)(lo >> 32); out->kmers[4 * i + 3] = (uint32_t)lo;
    }
    // deterministic unitig order of the reference's graph builder (BVComp)
    const uint64_t U = r.n_unitigs;
    std::vector<uint64_t> order(U);
    std::iota(order.begin(), order.end(), 0ull);
    std::sort(order.begin(), order.end(), bv_less{off.data(), bases.data()});
    out->n_unitigs = U;
    out->unitig_off = (uint64_t*)malloc((U + 1) * 8);
    out->unitig_bases = (uint8_t*)malloc(std::max<size_t>(bases.size(), 16));
    if (!out->unitig_off || !out->unitig_bases) { snk_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_count_graph: host allocation failed"); }
    uint64_t w = 0;
    for (uint64_t u = 0; u < U; ++u) {
        uint64_t s = order[u], len = off[s + 1] - off[s];
        out->unitig_off[u] = w;
        memcpy(out->unitig_bases + w, bases.data() + off[s], len);
        w += len;
    }
    out->unitig_off[U] = w;
    return SNK_OK;
}

extern "C" void snk_free(snk_result* r) {
    if (!r) return;
    free(r->kmers); free(r->counts); free(r->ctx); free(r->unitig_off); free(r->unitig_bases); free(r->spectrum);
    memset(r, 0, sizeof *r);
}

extern "C" int snk_write_bv(const char* path, uint64_t n_unitigs, const uint64_t* off, const uint8_t* bases, char* err, size_t errcap) {
    FILE* f = fopen(path, "wb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_write_bv: cannot open %s", path);
    bool ok = fwrite("BINWRITE", 1, 8, f) == 8 && fwrite(&n_unitigs, 8, 1, f) == 1;
    std::vector<uint8_t> buf;
    for (uint64_t u = 0; ok && u < n_unitigs; ++u) {
        const uint64_t len64 = off[u + 1] - off[u];
        if (len64 > 0xFFFFFFFFull) { fclose(f); return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_write_bv: unitig longer than 2^32 bases"); }
        const uint32_t len = (uint32_t)len64;
        const uint8_t* b = bases + off[u];
        buf.assign((len + 3) / 4, 0);
        for (uint32_t j = 0; j < len; ++j) buf[j >> 2] |= (uint8_t)((b[j] & 3u) << (2 * (j & 3)));
        ok = fwrite(&len, 4, 1, f) == 1 && (buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size());
    }
    if (fclose(f) != 0) ok = false;
    return ok ? SNK_OK : snk_fail(SNK_E_IO, err, errcap, "snk_write_bv: short write to %s", path);
}

extern "C" int snk_read_bv(const char* path, uint64_t* n_unitigs, uint64_t** off_out, uint8_t** bases_out, char* err, size_t errcap) {
    FILE* f = fopen(path, "rb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: cannot open %s", path);
    char magic[8];
    uint64_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BINWRITE", 8) || fread(&n, 8, 1, f) != 1) { fclose(f); return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: %s is not a BINWRITE file", path); }
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint8_t> bases;
    std::vector<uint8_t> buf;
    for (uint64_t u = 0; u < n; ++u) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) { fclose(f); return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: truncated file"); }
        buf.resize((len + 3) / 4);
        if (!buf.empty() && fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); return snk_fail(SNK_E_IO, err, errcap, "snk_read_bv: truncated file"); }
        off[u] = bases.size();
        for (uint32_t j = 0; j < len; ++j) bases.push_back((uint8_t)((buf[j >> 2] >> (2 * (j & 3))) & 3u));
    }
    off[n] = bases.size();
    fclose(f);
    *n_unitigs = n;
    *off_out = (uint64_t*)malloc((n + 1) * 8);
    *bases_out = (uint8_t*)malloc(std::max<size_t>(bases.size(), 16));
    if (!*off_out || !*bases_out) return snk_fail(SNK_E_NOMEM, err, errcap, "snk_read_bv: host allocation failed");
    memcpy(*off_out, off.data(), (n + 1) * 8);
    if (!bases.empty()) memcpy(*bases_out, bases.data(), bases.size());
    return SNK_OK;
}
