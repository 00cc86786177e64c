// snk_formats.hip -- readers for the stage-input files of ASSEMBLER_DF (host side of the b1/b2 seam).
//   fastb : feudal MasterVec<BaseVec>    lib/assembly/src/feudal/FeudalControlBlock.h:27-166, FeudalFileWriter.cc:18-140,
//                                        feudal/FieldVec.h:586-603
//   qualp : feudal MasterVec<PQVec>      feudal/PQVec.cc:86-200 (block codec), PQVec.h:158-171
//   bci   : BINWRITE vec<int64_t>        feudal/BinaryStream.h:33-45,105-108; 10X/DF.cc:464-469 (expansion)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include <zlib.h>

#include "snk_ctx.h"

namespace {

#pragma pack(push, 1)
struct fcb_t {          // FeudalControlBlock, 24 bytes
    uint32_t n;
    uint8_t flags, sizeof_fixed, sizeof_x, sizeof_a;
    uint64_t var_offset, fixed_offset;
};
#pragma pack(pop)
static_assert(sizeof(fcb_t) == 24, "feudal control block is 24 bytes");

struct file_buf {
    std::vector<uint8_t> d;
    bool load(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        d.resize(n > 0 ? (size_t)n : 0);
        bool ok = d.empty() || fread(d.data(), 1, d.size(), f) == d.size();
        fclose(f);
        return ok;
    }
};

// returns 0 and fills n/offs/fixed on success
int open_feudal(const file_buf& fb, const char* path, uint64_t* n, const uint64_t** offs, const uint8_t** fixed, char* err, size_t errcap) {
    if (fb.d.size() < sizeof(fcb_t)) return snk_fail(SNK_E_IO, err, errcap, "%s: too short for a feudal file", path);
    fcb_t h;
    memcpy(&h, fb.d.data(), sizeof h);
    if ((h.flags & 3) != 1 || (h.flags & 4)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "%s: 3-file or compressed feudal files are not supported", path);
    if (h.var_offset < sizeof(fcb_t) || h.fixed_offset < h.var_offset || h.fixed_offset > fb.d.size() || (h.fixed_offset - h.var_offset) % 8 || h.fixed_offset == h.var_offset)
        return snk_fail(SNK_E_IO, err, errcap, "%s: inconsistent feudal control block", path);
    *n = (h.fixed_offset - h.var_offset) / 8 - 1;
    *offs = reinterpret_cast<const uint64_t*>(fb.d.data() + h.var_offset);
    *fixed = fb.d.data() + h.fixed_offset;
    for (uint64_t i = 0; i <= *n; ++i) {
        uint64_t o;
        memcpy(&o, fb.d.data() + h.var_offset + 8 * i, 8);
        if (o < sizeof(fcb_t) || o > h.var_offset) return snk_fail(SNK_E_IO, err, errcap, "%s: element offset out of range", path);
    }
    return SNK_OK;
}

}  // namespace

extern "C" int snk_read_fastb(const char* path, uint64_t* n_reads, uint32_t* max_len, uint16_t** lens_out, uint32_t** rows_out,
                              char* err, size_t errcap) {
    file_buf fb;
    if (!fb.load(path)) return snk_fail(SNK_E_IO, err, errcap, "snk_read_fastb: cannot read %s", path);
    uint64_t n;
    const uint64_t* offs;
    const uint8_t* fixed;
    int rc = open_feudal(fb, path, &n, &offs, &fixed, err, errcap);
    if (rc) return rc;
    if ((size_t)(fixed - fb.d.data()) + 4 * n > fb.d.size()) return snk_fail(SNK_E_IO, err, errcap, "%s: truncated length table", path);
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t L;
        memcpy(&L, fixed + 4 * i, 4);
        uint64_t a, b;
        memcpy(&a, (const uint8_t*)offs + 8 * i, 8);
        memcpy(&b, (const uint8_t*)offs + 8 * (i + 1), 8);
        if (L > 65535 || (uint64_t)(L + 3) / 4 > b - a) return snk_fail(SNK_E_IO, err, errcap, "%s: read %llu has a bad length", path, (unsigned long long)i);
        if (L > mx) mx = L;
    }
    const uint32_t rw = (mx + 15) / 16 ? (mx + 15) / 16 : 1;
    uint16_t* lens = (uint16_t*)malloc(n ? n * 2 : 2);
    uint32_t* rows = (uint32_t*)calloc(n ? n * rw : 1, 4);
    if (!lens || !rows) { free(lens); free(rows); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_read_fastb: host allocation failed"); }
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t L;
        memcpy(&L, fixed + 4 * i, 4);
        uint64_t a;
        memcpy(&a, (const uint8_t*)offs + 8 * i, 8);
        const uint8_t* src = fb.d.data() + a;
        lens[i] = (uint16_t)L;
        uint32_t* row = rows + i * rw;
        for (uint32_t j = 0; j < L; ++j) {
            uint32_t b = (src[j >> 2] >> (2 * (j & 3))) & 3u;        // LSB-first within the byte
            row[j >> 4] |= b << (30 - 2 * (j & 15));                 // MSB-first within the word
        }
    }
    *n_reads = n; *max_len = mx; *lens_out = lens; *rows_out = rows;
    return SNK_OK;
}

extern "C" int snk_read_qualp(const char* path, uint64_t n_reads, uint32_t qstride, uint8_t* quals, char* err, size_t errcap) {
    file_buf fb;
    if (!fb.load(path)) return snk_fail(SNK_E_IO, err, errcap, "snk_read_qualp: cannot read %s", path);
    uint64_t n;
    const uint64_t* offs;
    const uint8_t* fixed;
    int rc = open_feudal(fb, path, &n, &offs, &fixed, err, errcap);
    if (rc) return rc;
    if (n != n_reads) return snk_fail(SNK_E_IO, err, errcap, "%s holds %llu reads, expected %llu", path, (unsigned long long)n, (unsigned long long)n_reads);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t a, b;
        memcpy(&a, (const uint8_t*)offs + 8 * i, 8);
        memcpy(&b, (const uint8_t*)offs + 8 * (i + 1), 8);
        const uint8_t* p = fb.d.data() + a;
        const uint8_t* end = fb.d.data() + b;
        uint8_t* q = quals + i * (uint64_t)qstride;
        uint32_t w = 0;
        while (p < end && *p) {                       // chain of blocks, 0 terminates (PQVec.cc:86-127)
            const uint32_t nqs = *p++;
            if (p + 2 > end) return snk_fail(SNK_E_IO, err, errcap, "%s: truncated quality block (read %llu)", path, (unsigned long long)i);
            // 17-bit header tail: nBits(3) | minQ(6), then the values start at bit 9 of this little-endian bit stream
            const uint32_t nbits = p[0] & 7u;
            const uint32_t minq = ((p[0] >> 3) | ((p[1] & 1u) << 5)) & 63u;
            const uint32_t blk = (nqs * nbits + 9 + 7) / 8;          // bytes after the nQs byte
            if (p + blk > end) return snk_fail(SNK_E_IO, err, errcap, "%s: truncated quality block (read %llu)", path, (unsigned long long)i);
            if (w + nqs > qstride) return snk_fail(SNK_E_ARG, err, errcap, "snk_read_qualp: read %llu longer than qstride", (unsigned long long)i);
            uint64_t bitpos = 9;
            for (uint32_t k = 0; k < nqs; ++k) {
                uint32_t v = 0;
                for (uint32_t t = 0; t < nbits; ++t, ++bitpos) v |= ((p[bitpos >> 3] >> (bitpos & 7)) & 1u) << t;
                q[w++] = (uint8_t)(minq + v);
            }
            p += blk;
        }
    }
    return SNK_OK;
}

extern "C" int snk_read_bci(const char* path, uint64_t n_reads, int32_t* bc, uint64_t* n_barcodes, char* err, size_t errcap) {
    file_buf fb;
    if (!fb.load(path)) return snk_fail(SNK_E_IO, err, errcap, "snk_read_bci: cannot read %s", path);
    if (fb.d.size() < 16 || memcmp(fb.d.data(), "BINWRITE", 8)) return snk_fail(SNK_E_IO, err, errcap, "%s is not a BINWRITE file", path);
    uint64_t m;
    memcpy(&m, fb.d.data() + 8, 8);
    if (16 + 8 * m > fb.d.size() || m < 1) return snk_fail(SNK_E_IO, err, errcap, "%s: truncated index", path);
    const uint8_t* p = fb.d.data() + 16;
    int64_t last;
    memcpy(&last, p + 8 * (m - 1), 8);
    if ((uint64_t)last != n_reads) return snk_fail(SNK_E_IO, err, errcap, "%s indexes %lld reads, expected %llu", path, (long long)last, (unsigned long long)n_reads);
    for (uint64_t i = 0; i < n_reads; ++i) bc[i] = -1;            // vec<int32_t> bc(bci.back(), -1), DF.cc:464
    for (uint64_t b = 0; b + 1 < m; ++b) {
        int64_t s, e;
        memcpy(&s, p + 8 * b, 8);
        memcpy(&e, p + 8 * (b + 1), 8);
        if (s < 0 || e < s || (uint64_t)e > n_reads) return snk_fail(SNK_E_IO, err, errcap, "%s: bad range of barcode %llu", path, (unsigned long long)b);
        for (int64_t j = s; j < e; ++j) bc[j] = (int32_t)b;
    }
    if (n_barcodes) *n_barcodes = m - 1;
    return SNK_OK;
}


// ---------------------------------------------------------------------------------------------- FASTH (f3)
// MultiFastqIter (lib/tada/src/multifastq.rs:69-127): gzip'ed text, 9 lines per read pair -- header, R1, Q1, R2, Q2,
// barcode field ("SEQ-gg[,raw]"), three more lines that the assembler ignores.  R1 becomes read 2q, R2 read 2q+1
// (cmd_msp.rs:160-181).  Output: ASCII bases and raw phred values (quality characters - 33) in rows of `stride` bytes,
// the read lengths, and one zero-padded 64-byte barcode field per PAIR (the input of snk_dev_bc_ids).
namespace {
bool gz_line(gzFile f, std::string& out) {          // one line without its terminator; false at end of file
    out.clear();
    char buf[4096];
    bool any = false;
    while (gzgets(f, buf, sizeof buf)) {
        any = true;
        const size_t n = strlen(buf);
        out.append(buf, n);
        if (n && buf[n - 1] == '\n') break;
    }
    if (!any) return false;
    if (!out.empty() && out.back() == '\n') out.pop_back();
    if (!out.empty() && out.back() == '\r') out.pop_back();
    return true;
}
}  // namespace

extern "C" int snk_read_fasth(const char* path, uint32_t stride, uint64_t* n_reads, uint32_t* max_len, uint8_t** ascii, uint8_t** quals,
                              uint16_t** lens, uint8_t** bc_fields, char* err, size_t errcap) {
    if (!path || !n_reads || !ascii || !quals || !lens || !bc_fields) return snk_fail(SNK_E_ARG, err, errcap, "snk_read_fasth: NULL argument");
    if (stride == 0 || stride > 65535) return snk_fail(SNK_E_ARG, err, errcap, "snk_read_fasth: bad row stride");
    *n_reads = 0; *ascii = nullptr; *quals = nullptr; *lens = nullptr; *bc_fields = nullptr;
    if (max_len) *max_len = 0;
    gzFile f = gzopen(path, "rb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_read_fasth: cannot open %s", path);
    gzbuffer(f, 1 << 20);
    std::vector<uint8_t> A, Q, B;
    std::vector<uint16_t> L;
    std::string head, ln[8];
    uint32_t mx = 0;
    uint64_t pairs = 0;
    int rc = SNK_OK;
    while (gz_line(f, head)) {
        bool ok = true;
        for (int q = 0; q < 8 && ok; ++q) ok = gz_line(f, ln[q]);
        if (!ok) { rc = snk_fail(SNK_E_IO, err, errcap, "%s: truncated record %llu", path, (unsigned long long)pairs); break; }
        for (int m = 0; m < 2 && rc == SNK_OK; ++m) {
            const std::string &r = ln[2 * m], &q = ln[2 * m + 1];
            if (r.size() > stride) { rc = snk_fail(SNK_E_UNSUPPORTED, err, errcap, "%s: a read of %zu bases does not fit rows of %u", path, r.size(), stride); break; }
            if (q.size() != r.size()) { rc = snk_fail(SNK_E_IO, err, errcap, "%s: record %llu: %zu bases but %zu qualities", path, (unsigned long long)pairs, r.size(), q.size()); break; }
            const size_t o = A.size();
            A.resize(o + stride, (uint8_t)'A');
            Q.resize(o + stride, 0);
            memcpy(&A[o], r.data(), r.size());
            for (size_t i = 0; i < q.size(); ++i) Q[o + i] = (uint8_t)(q[i] - 33);
            L.push_back((uint16_t)r.size());
            if (r.size() > mx) mx = (uint32_t)r.size();
        }
        if (rc != SNK_OK) break;
        const size_t ob = B.size();
        B.resize(ob + 64, 0);
        const std::string& bc = ln[4];
        const size_t cut = bc.find(',');                       // only the part before the first ',' is the barcode
        const size_t nb = std::min<size_t>(cut == std::string::npos ? bc.size() : cut, 64);
        memcpy(&B[ob], bc.data(), nb);
        ++pairs;
    }
    if (rc == SNK_OK) {
        // the end of the file inside a gzip member (an interrupted copy) reads as a clean end of data through gzgets; zlib knows
        // (Z_BUF_ERROR at EOF).  The reference panics on it (MultiGzDecoder + unwrap(), multifastq.rs:69-127)
        int zerr = Z_OK;
        (void)gzerror(f, &zerr);
        if (zerr != Z_OK && zerr != Z_STREAM_END) rc = snk_fail(SNK_E_IO, err, errcap, "%s: truncated or corrupt gzip stream (zlib %d)", path, zerr);
    }
    gzclose(f);
    if (rc != SNK_OK) return rc;
    auto dup = [](const void* src, size_t bytes) -> void* { void* p = malloc(bytes ? bytes : 16); if (p && bytes) memcpy(p, src, bytes); return p; };
    *ascii = (uint8_t*)dup(A.data(), A.size());
    *quals = (uint8_t*)dup(Q.data(), Q.size());
    *lens = (uint16_t*)dup(L.data(), L.size() * 2);
    *bc_fields = (uint8_t*)dup(B.data(), B.size());
    if (!*ascii || !*quals || !*lens || !*bc_fields) {
        free(*ascii); free(*quals); free(*lens); free(*bc_fields);
        *ascii = *quals = *bc_fields = nullptr; *lens = nullptr;
        return snk_fail(SNK_E_NOMEM, err, errcap, "snk_read_fasth: host allocation failed");
    }
    *n_reads = 2 * pairs;
    if (max_len) *max_len = mx;
    return SNK_OK;
}

extern "C" void snk_host_free(void* p) { free(p); }
