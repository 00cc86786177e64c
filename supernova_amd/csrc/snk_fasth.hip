// snk_fasth.hip -- f3 at rate: FASTH files -> batches of reads in page-locked host memory, many files in
// flight, and from there into HBM (snk_dev_ingest_fasth) with the upload, the 2-bit pack and the barcode ids overlapped
// with the decode.
//
// What it replaces: the reading half of tada's MSP stage -- one decode thread per two files
// (lib/tada/src/cmd_msp.rs:55-69) over MultiFastqIter (lib/tada/src/multifastq.rs:69-127: gzip text, 9 lines per read pair:
// header, R1, Q1, R2, Q2, barcode field, three ignored lines; R1 = read 2q, R2 = read 2q+1, cmd_msp.rs:160-181) and the
// barcode lookup of BcIndexer (lib/tada/src/utils.rs:101-164, on the device: snk_ingest.hip).
//
// Design: a pool of worker threads takes whole files (a gzip stream is sequential; the parallelism of this format is across
// files: a lane of a flowcell is bucketed into dozens of them).  A worker inflates with zlib's streaming API into a text
// window and parses records in place -- no per-line strings, the bases and qualities of a read are copied once, from the
// inflate window into their row of the batch -- and hands full batches to the consumer through a queue.  Batches live in
// page-locked memory (when a GPU is there), so the DMA engine reads them where they are.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "snk_ctx.h"
#include "snk_synth.h"

struct snk_fasth_stream {
    struct batch {
        uint8_t *ascii = nullptr, *quals = nullptr, *bcf = nullptr;
        uint16_t* lens = nullptr;
        uint64_t n_pairs = 0, text_bytes = 0, first_pair = 0;
        uint32_t file = 0, max_len = 0;
        bool pinned = false;
    };
    std::vector<std::string> paths;
    uint32_t stride = 0, batch_pairs = 0;
    std::vector<batch> pool;
    std::deque<int> free_q, ready_q;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    std::vector<std::thread> workers;
    std::atomic<uint32_t> next_file{0};
    uint32_t live_workers = 0;
    bool stop = false;
    int rc = SNK_OK;
    std::string errmsg;
    std::vector<uint64_t> file_pairs;      // pairs of every file (known once it has been read to its end)
};

namespace {

// libdeflate, when the host has it (bound with dlopen like librccl: libsnk links nothing it can do without): whole-buffer inflate,
// three times zlib's rate (0.64 against 0.21 GB/s of text per thread on the build host).  It cannot stream, so it takes the files
// whose compressed size is under SNK_FASTH_WHOLE_MAX_MB (default 128) -- a lane bucketed into dozens of files gives files of that
// size -- and zlib's streaming inflate takes the rest.  SNK_FASTH_LIBDEFLATE=0 switches it off.
struct deflate_api {
    void* h = nullptr;
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
    bool tried = false;
};
deflate_api g_defl;
std::mutex g_defl_mu;
const deflate_api* libdeflate() {
    std::lock_guard<std::mutex> lk(g_defl_mu);
    if (g_defl.tried) return g_defl.gzip_ex ? &g_defl : nullptr;
    g_defl.tried = true;
    const char* off = getenv("SNK_FASTH_LIBDEFLATE");
    if (off && *off == '0') return nullptr;
    for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        *(void**)(&g_defl.alloc) = dlsym(h, "libdeflate_alloc_decompressor");
        *(void**)(&g_defl.release) = dlsym(h, "libdeflate_free_decompressor");
        *(void**)(&g_defl.gzip_ex) = dlsym(h, "libdeflate_gzip_decompress_ex");
        if (g_defl.alloc && g_defl.release && g_defl.gzip_ex) { g_defl.h = h; return &g_defl; }
        g_defl.gzip_ex = nullptr;
        dlclose(h);
    }
    return nullptr;
}

// CPUs this process may actually use: the cgroup's CPU quota when there is one (a container with 256 visible hardware threads and
// cpu.max = "1600000 100000" gets 16 CPUs' worth of time: 64 decode threads there are SLOWER than 16 -- 6.3 against 8.0 GB/s of text --
// because the scheduler throttles the whole group, the consumer thread included), else the affinity mask / hardware threads.
uint32_t host_cpu_budget() {
    uint32_t hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 8;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (uint32_t)c < hw) hw = (uint32_t)c; }
    auto quota = [](const char* path, bool v2) -> double {
        FILE* f = fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = "", b[64] = "";
        double q = 0.0;
        if (v2) { if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) q = atof(a) / atof(b); }
        else if (fscanf(f, "%63s", a) == 1 && atof(a) > 0) {
            FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (g) { if (fscanf(g, "%63s", b) == 1 && atof(b) > 0) q = atof(a) / atof(b); fclose(g); }
        }
        fclose(f);
        return q;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0.0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    if (q > 0.0) { const uint32_t c = (uint32_t)(q + 0.999); if (c >= 1 && c < hw) hw = c; }
    return hw;
}

void fail(snk_fasth_stream* s, int rc, const std::string& msg) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->rc == SNK_OK) { s->rc = rc; s->errmsg = msg; }
    s->stop = true;
    s->cv_free.notify_all();
    s->cv_ready.notify_all();
}

int take_free(snk_fasth_stream* s) {
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_free.wait(lk, [&] { return s->stop || !s->free_q.empty(); });
    if (s->stop) return -1;
    const int b = s->free_q.front();
    s->free_q.pop_front();
    return b;
}
void push_ready(snk_fasth_stream* s, int b) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->ready_q.push_back(b);
    s->cv_ready.notify_one();
}

// one file, start to end.  Returns false after an error / stop.
bool decode_file(snk_fasth_stream* s, uint32_t fi) {
    const std::string& path = s->paths[fi];
    const uint32_t stride = s->stride;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { fail(s, SNK_E_IO, "cannot open " + path); return false; }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    constexpr size_t IN = 1 << 20, WIN = 4 << 20;
    std::vector<unsigned char> in(IN), win(WIN);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) { close(fd); fail(s, SNK_E_INTERNAL, "inflateInit2 failed"); return false; }
    int cur = -1;                 // batch being filled
    uint64_t pairs_in_file = 0, text_in_batch = 0;
    size_t have = 0;              // bytes of text in the window
    size_t line_beg = 0;          // start of the line being completed
    size_t scan = 0;              // first byte not yet searched for a newline
    uint32_t li = 0;              // line of the record, 0..8
    uint32_t r_len[2] = {0, 0};
    bool ok = true, eof_in = false, first_byte = true, mid_member = false;
    std::string what;
    auto flush = [&](bool last) {
        if (cur < 0) return;
        snk_fasth_stream::batch& b = s->pool[cur];
        if (b.n_pairs == 0 && !last) return;
        b.text_bytes = text_in_batch;
        text_in_batch = 0;
        push_ready(s, cur);
        cur = -1;
    };
    auto on_line = [&](const unsigned char* p, size_t n) -> bool {       // a complete line without its terminator
        if (n && p[n - 1] == '\r') --n;
        if (li == 0) {
            if (cur < 0) {
                cur = take_free(s);
                if (cur < 0) return false;
                snk_fasth_stream::batch& b = s->pool[cur];
                b.n_pairs = 0; b.file = fi; b.first_pair = pairs_in_file; b.max_len = 0;
            }
        } else if (li == 1 || li == 3) {
            snk_fasth_stream::batch& b = s->pool[cur];
            if (n > stride) { what = path + ": a read of " + std::to_string(n) + " bases does not fit rows of " + std::to_string(stride); return false; }
            uint8_t* row = b.ascii + (2 * b.n_pairs + (li >> 1)) * (size_t)stride;
            memcpy(row, p, n);
            memset(row + n, 'A', stride - n);
            r_len[li >> 1] = (uint32_t)n;
            b.lens[2 * b.n_pairs + (li >> 1)] = (uint16_t)n;
            if (n > b.max_len) b.max_len = (uint32_t)n;
        } else if (li == 2 || li == 4) {
            snk_fasth_stream::batch& b = s->pool[cur];
            const uint32_t m = (li >> 1) - 1;
            if (n != r_len[m]) { what = path + ": record " + std::to_string(pairs_in_file) + ": " + std::to_string(r_len[m]) + " bases but " + std::to_string(n) + " qualities"; return false; }
            uint8_t* row = b.quals + (2 * b.n_pairs + m) * (size_t)stride;
            for (size_t i = 0; i < n; ++i) row[i] = (uint8_t)(p[i] - 33);
            memset(row + n, 0, stride - n);
        } else if (li == 5) {
            snk_fasth_stream::batch& b = s->pool[cur];
            const void* comma = memchr(p, ',', n);                      // only the part before the first ',' is the barcode
            size_t nb = comma ? (size_t)((const unsigned char*)comma - p) : n;
            if (nb > 64) nb = 64;
            uint8_t* f = b.bcf + b.n_pairs * 64;
            memcpy(f, p, nb);
            memset(f + nb, 0, 64 - nb);
        }
        if (++li == 9) {
            li = 0;
            snk_fasth_stream::batch& b = s->pool[cur];
            ++b.n_pairs;
            ++pairs_in_file;
            if (b.n_pairs == s->batch_pairs) flush(false);
        }
        return true;
    };
    // ---- whole-file inflate (libdeflate) for files of modest size: every gzip member in turn into one text buffer, lines parsed from it
    bool whole_done = false;
    if (const deflate_api* D = libdeflate()) {
        struct stat sb;
        const char* mx = getenv("SNK_FASTH_WHOLE_MAX_MB");
        const uint64_t max_comp = (uint64_t)(mx && *mx ? strtoull(mx, nullptr, 10) : 128) << 20;
        if (fstat(fd, &sb) == 0 && sb.st_size >= 18 && (uint64_t)sb.st_size <= max_comp) {
            const size_t csz = (size_t)sb.st_size;
            std::vector<unsigned char> comp(csz);
            size_t got = 0;
            while (got < csz) { const ssize_t r = pread(fd, comp.data() + got, csz - got, (off_t)got); if (r <= 0) break; got += (size_t)r; }
            if (got == csz && comp[0] == 0x1f && comp[1] == 0x8b) {
                // the last member's ISIZE says how large its text is (mod 2^32): a first guess for the buffer, doubled while it is too small
                uint32_t isize;
                memcpy(&isize, comp.data() + csz - 4, 4);
                const size_t cap_max = (size_t)24 * csz + (64u << 20);
                size_t cap = (size_t)isize + 64;
                if (cap < 4 * csz) cap = 4 * csz;
                if (cap > cap_max) cap = cap_max;          // (a cut file's last four bytes are not a size)
                void* dc = D->alloc();
                // (plain malloc / realloc: a std::vector would zero 270 MB per file before the inflate overwrites them)
                struct tbuf { unsigned char* p = nullptr; size_t n = 0; ~tbuf() { free(p); } unsigned char* data() { return p; } size_t size() const { return n; }
                              bool grow(size_t m) { unsigned char* q = (unsigned char*)realloc(p, m); if (!q) return false; p = q; n = m; return true; } } text;
                bool bad = !dc, too_big = false;
                size_t ipos = 0, opos = 0;
                while (!bad && !too_big && ipos < csz) {
                    if (text.size() < cap && !text.grow(cap)) { bad = true; break; }
                    size_t ain = 0, aout = 0;
                    const int r = D->gzip_ex(dc, comp.data() + ipos, csz - ipos, text.data() + opos, text.size() - opos, &ain, &aout);
                    if (r == 0) { ipos += ain; opos += aout; if (ain == 0) bad = true; }
                    else if (r == 3) { if (cap >= cap_max) too_big = true; else cap = cap * 2 < cap_max ? cap * 2 : cap_max; }      // LIBDEFLATE_INSUFFICIENT_SPACE
                    else bad = true;
                }
                if (dc) D->release(dc);
                if (bad) { what = path + ": truncated or corrupt gzip stream"; ok = false; whole_done = true; }
                else if (!too_big) {
                    whole_done = true;
                    size_t lb = 0;
                    while (lb < opos && ok) {
                        const unsigned char* nl = (const unsigned char*)memchr(text.data() + lb, '\n', opos - lb);
                        const size_t e = nl ? (size_t)(nl - text.data()) : opos;
                        text_in_batch += e + (nl ? 1 : 0) - lb;
                        if (!on_line(text.data() + lb, e - lb)) { ok = false; break; }
                        lb = e + 1;
                    }
                }
            } else if (got == csz && csz >= 2) { what = path + ": not a gz file"; ok = false; whole_done = true; }
        }
    }
    while (ok && !whole_done) {
        // refill the input
        if (zs.avail_in == 0 && !eof_in) {
            const ssize_t got = read(fd, in.data(), IN);
            if (got < 0) { what = "read error on " + path; ok = false; break; }
            if (got == 0) eof_in = true;
            zs.next_in = in.data();
            zs.avail_in = (uInt)got;
            if (first_byte && got >= 2) { first_byte = false; if (in[0] != 0x1f || in[1] != 0x8b) { what = path + ": not a gz file"; ok = false; break; } }
        }
        if (eof_in && zs.avail_in == 0) break;
        if (have == WIN || line_beg > WIN / 2) {
            // keep the incomplete line, drop what has been parsed
            if (line_beg == 0) { what = path + ": a line longer than " + std::to_string(WIN) + " bytes"; ok = false; break; }
            memmove(win.data(), win.data() + line_beg, have - line_beg);
            have -= line_beg; scan -= line_beg; line_beg = 0;
        }
        zs.next_out = win.data() + have;
        zs.avail_out = (uInt)(WIN - have);
        const uInt in_before = zs.avail_in;
        const int zr = inflate(&zs, Z_NO_FLUSH);
        // a member is open from its first consumed byte to its Z_STREAM_END: the end of the input inside one is a truncated file
        // (the reference's MultiGzDecoder + unwrap() panics on it, multifastq.rs:69-127)
        if (zr == Z_STREAM_END) mid_member = false;
        else if (zs.avail_in < in_before || zs.avail_out < (uInt)(WIN - have)) mid_member = true;
        if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) { what = path + ": corrupt gzip stream (" + std::string(zs.msg ? zs.msg : "zlib error") + ")"; ok = false; break; }
        const size_t now = WIN - zs.avail_out;
        text_in_batch += now - have;
        have = now;
        // complete lines
        while (scan < have) {
            const unsigned char* nl = (const unsigned char*)memchr(win.data() + scan, '\n', have - scan);
            if (!nl) { scan = have; break; }
            const size_t e = (size_t)(nl - win.data());
            if (!on_line(win.data() + line_beg, e - line_beg)) { ok = false; break; }
            line_beg = scan = e + 1;
        }
        if (!ok) break;
        if (zr == Z_STREAM_END) {
            // a further gzip member may follow (concatenated members are one stream)
            if (zs.avail_in == 0 && eof_in) break;
            if (inflateReset(&zs) != Z_OK) { what = "inflateReset failed"; ok = false; break; }
        }
        if (zr == Z_BUF_ERROR && zs.avail_in == 0 && eof_in) break;
    }
    inflateEnd(&zs);
    close(fd);
    if (ok && mid_member) { what = path + ": truncated gzip stream (the input ends inside a member)"; ok = false; }
    if (ok && !whole_done && line_beg < have) {          // a last line without a newline
        if (!on_line(win.data() + line_beg, have - line_beg)) ok = false;
    }
    if (ok && li != 0) { what = path + ": truncated record " + std::to_string(pairs_in_file); ok = false; }
    if (!ok) {
        if (!what.empty()) fail(s, what.find("does not fit") != std::string::npos ? SNK_E_UNSUPPORTED : SNK_E_IO, what);
        return false;
    }
    flush(true);
    { std::lock_guard<std::mutex> lk(s->mu); s->file_pairs[fi] = pairs_in_file; }
    return true;
}

void worker_main(snk_fasth_stream* s) {
    for (;;) {
        const uint32_t fi = s->next_file.fetch_add(1);
        if (fi >= s->paths.size()) break;
        if (!decode_file(s, fi)) break;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    --s->live_workers;
    s->cv_ready.notify_all();
}

void free_batch(snk_fasth_stream::batch& b) {
    auto rel = [&](void* p) { if (!p) return; if (b.pinned) (void)hipHostFree(p); else free(p); };
    rel(b.ascii); rel(b.quals); rel(b.bcf); rel(b.lens);
    b.ascii = b.quals = b.bcf = nullptr; b.lens = nullptr;
}

}  // namespace

extern "C" uint32_t snk_host_cpu_budget(void) { return host_cpu_budget(); }

extern "C" int snk_fasth_open(const char* const* paths, uint32_t n_files, uint32_t stride, uint32_t batch_pairs, uint32_t threads, uint32_t flags,
                              snk_fasth_stream** out, char* err, size_t errcap) {
    if (!paths || !out || n_files == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_fasth_open: no files");
    if (stride == 0 || stride > 65535) return snk_fail(SNK_E_ARG, err, errcap, "snk_fasth_open: bad row stride");
    if (batch_pairs == 0) batch_pairs = 32768;
    if (threads == 0) { const uint32_t b = host_cpu_budget(); threads = b > 3 ? b - 2 : b; }       // (the consumer and the HIP runtime's helpers want CPUs too)
    if (threads > n_files) threads = n_files;
    if (threads > 256) threads = 256;
    snk_fasth_stream* s = new snk_fasth_stream();
    for (uint32_t i = 0; i < n_files; ++i) s->paths.push_back(paths[i] ? paths[i] : "");
    s->stride = stride;
    s->batch_pairs = batch_pairs;
    s->file_pairs.assign(n_files, 0);
    const bool want_pinned = (flags & 1u) != 0;
    const uint32_t n_batches = threads + (threads < 14 ? threads + 2 : 16);      // one per worker being filled + what waits for / is with the consumer
    s->pool.resize(n_batches);
    for (uint32_t i = 0; i < n_batches; ++i) {
        snk_fasth_stream::batch& b = s->pool[i];
        const size_t rows = 2 * (size_t)batch_pairs * stride;
        bool ok = true;
        auto get = [&](size_t bytes) -> void* {
            void* p = nullptr;
            if (want_pinned) { if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } }
            else p = malloc(bytes);
            if (!p) ok = false;
            return p;
        };
        b.pinned = want_pinned;
        b.ascii = (uint8_t*)get(rows);
        b.quals = (uint8_t*)get(rows);
        b.bcf = (uint8_t*)get((size_t)batch_pairs * 64);
        b.lens = (uint16_t*)get((size_t)batch_pairs * 4);
        if (!ok) {
            for (auto& q : s->pool) free_batch(q);
            delete s;
            return snk_fail(SNK_E_NOMEM, err, errcap, "snk_fasth_open: %s host allocation failed (%u batches of %u pairs)", want_pinned ? "page-locked" : "", n_batches, batch_pairs);
        }
        s->free_q.push_back((int)i);
    }
    s->live_workers = threads;
    for (uint32_t t = 0; t < threads; ++t) s->workers.emplace_back(worker_main, s);
    *out = s;
    return SNK_OK;
}

extern "C" int snk_fasth_next(snk_fasth_stream* s, snk_fasth_batch* out, char* err, size_t errcap) {
    if (!s || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_fasth_next: NULL argument");
    memset(out, 0, sizeof *out);
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_ready.wait(lk, [&] { return !s->ready_q.empty() || s->live_workers == 0 || s->rc != SNK_OK; });
    if (s->rc != SNK_OK) return snk_fail(s->rc, err, errcap, "%s", s->errmsg.c_str());
    if (s->ready_q.empty()) return SNK_OK;          // n_pairs == 0: every file has been read to its end
    const int b = s->ready_q.front();
    s->ready_q.pop_front();
    const snk_fasth_stream::batch& B = s->pool[b];
    out->n_pairs = B.n_pairs; out->file = B.file; out->first_pair = B.first_pair; out->max_len = B.max_len;
    out->ascii = B.ascii; out->quals = B.quals; out->lens = B.lens; out->bc_fields = B.bcf; out->text_bytes = B.text_bytes;
    out->token = b + 1;
    return SNK_OK;
}

extern "C" void snk_fasth_release(snk_fasth_stream* s, snk_fasth_batch* b) {
    if (!s || !b || b->token == 0) return;
    std::lock_guard<std::mutex> lk(s->mu);
    s->free_q.push_back((int)b->token - 1);
    b->token = 0;
    s->cv_free.notify_one();
}

extern "C" uint64_t snk_fasth_file_pairs(snk_fasth_stream* s, uint32_t file) {
    if (!s || file >= s->file_pairs.size()) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    return s->file_pairs[file];
}

extern "C" void snk_fasth_close(snk_fasth_stream* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; s->cv_free.notify_all(); s->cv_ready.notify_all(); }
    for (auto& t : s->workers) t.join();
    for (auto& b : s->pool) free_batch(b);
    delete s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Synthetic FASTH (tests, bench.py --ingest): pairs [first_pair, first_pair + n_pairs) of the synthetic linked-read model
// (snk_synth.h) as one gzip file in the layout MultiFastqIter reads.  Barcode field = a whitelist-style 16-mer derived from the
// barcode id + "-1" (+ ",raw" on every third pair), or a sequence off the whitelist for bc 0.
namespace {
void bc_seq(uint32_t id, char* out16) {
    // a fixed bijection id -> 16-mer (id < 2^32): 2 bits per base
    uint32_t x = id * 2654435761u;
    for (int i = 0; i < 16; ++i) { out16[i] = "ACGT"[x & 3u]; x >>= 2; }
}
}  // namespace

extern "C" void snk_synth_bc_seq(uint32_t id, char* out16) { bc_seq(id, out16); }

extern "C" int snk_synth_fasth_write(const char* path, const snk_synth_params* sp, uint64_t first_pair, uint64_t n_pairs, int level, uint64_t* text_bytes,
                                     char* err, size_t errcap) {
    if (!path || !sp) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_fasth_write: NULL argument");
    const uint32_t L = sp->read_len, rw = (L + 15) / 16;
    if (L == 0 || L > 4096) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_fasth_write: bad read length");
    char mode[8];
    snprintf(mode, sizeof mode, "wb%d", level < 0 ? 1 : (level > 9 ? 9 : level));
    gzFile f = gzopen(path, mode);
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_synth_fasth_write: cannot open %s", path);
    gzbuffer(f, 1 << 20);
    std::vector<uint32_t> rows(2 * rw);
    std::vector<uint8_t> q(2 * (size_t)L);
    std::string rec;
    uint64_t total = 0;
    int rc = SNK_OK;
    for (uint64_t pq = first_pair; pq < first_pair + n_pairs; ++pq) {
        int32_t bc[2] = {0, 0};
        for (int m = 0; m < 2; ++m) snk_synth_read(*sp, 2 * pq + m, rows.data() + m * rw, rw, q.data() + (size_t)m * L, &bc[m]);
        rec.clear();
        rec += "@SYN:"; rec += std::to_string(pq); rec += '\n';
        for (int m = 0; m < 2; ++m) {
            for (uint32_t i = 0; i < L; ++i) rec += "ACGT"[(rows[m * rw + (i >> 4)] >> (30 - 2 * (i & 15))) & 3u];
            rec += '\n';
            for (uint32_t i = 0; i < L; ++i) rec += (char)(q[(size_t)m * L + i] + 33);
            rec += '\n';
        }
        char b16[16];
        if (bc[0] > 0) { bc_seq((uint32_t)bc[0], b16); rec.append(b16, 16); rec += "-1"; }
        else rec += "NNNNNNNNNNNNNNNN-1";
        if (pq % 3 == 0) rec += ",RAWRAWRAWRAWRAWR";
        rec += '\n';
        rec += "FFFFFFFFFFFFFFFF\nACGTACGT\nFFFFFFFF\n";
        total += rec.size();
        if (gzwrite(f, rec.data(), (unsigned)rec.size()) != (int)rec.size()) { rc = snk_fail(SNK_E_IO, err, errcap, "snk_synth_fasth_write: write error on %s", path); break; }
    }
    if (gzclose(f) != Z_OK && rc == SNK_OK) rc = snk_fail(SNK_E_IO, err, errcap, "snk_synth_fasth_write: close error on %s", path);
    if (text_bytes) *text_bytes = total;
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// FASTH files -> reads resident in HBM (packed 2-bit rows, quality rows, lengths, barcode ids), file-major order.
// The caller's thread is the consumer of the decode workers: every batch goes up with asynchronous copies out of its
// page-locked buffers -- quality rows and lengths straight to their place, the ASCII rows and barcode fields through a small
// staging ring from which snk_dev_pack_ascii / snk_dev_bc_ids take them -- while the workers inflate the next batches.
// Batches arrive in any order; the arrays are put into file-major order at the end (one device copy per batch and array).
namespace {
__global__ void __launch_bounds__(256) pair_ids_kernel(const int32_t* __restrict__ pair_ids, uint64_t n_pairs, int32_t* __restrict__ read_ids) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * n_pairs) read_ids[i] = pair_ids[i >> 1];
}
struct ingest_arrays {
    uint32_t* rows = nullptr; uint8_t* quals = nullptr; uint16_t* lens = nullptr; int32_t* bc = nullptr;
    uint64_t cap = 0;
    void release() { (void)hipFree(rows); (void)hipFree(quals); (void)hipFree(lens); (void)hipFree(bc); rows = nullptr; quals = nullptr; lens = nullptr; bc = nullptr; cap = 0; }
};
int alloc_arrays(ingest_arrays& a, uint64_t cap, uint32_t row_words, uint32_t qstride, bool want_bc, char* err, size_t errcap) {
    a.cap = cap;
    SNK_HIP_TRY(hipMalloc((void**)&a.rows, (cap + 1) * row_words * 4ull));
    SNK_HIP_TRY(hipMalloc((void**)&a.quals, (cap + 1) * (uint64_t)qstride));
    SNK_HIP_TRY(hipMalloc((void**)&a.lens, (cap + 8) * 2ull));
    if (want_bc) SNK_HIP_TRY(hipMalloc((void**)&a.bc, (cap + 2) * 4ull));
    return SNK_OK;
}
double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
}  // namespace

extern "C" int snk_dev_ingest_fasth(snk_ctx* ctx, const char* const* paths, uint32_t n_files, uint32_t read_len, const snk_bc_index* ix, uint32_t threads,
                                    uint32_t batch_pairs, snk_dev_ingest* out, char* err, size_t errcap) {
    if (!ctx || !paths || !out || n_files == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_fasth: NULL argument");
    if (read_len == 0 || read_len > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_fasth: read_len must be 1..256");
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(snk_enter(ctx));
    const uint32_t stride = (read_len + 15) / 16 * 16, row_words = (read_len + 15) / 16, qstride = stride;
    if (batch_pairs == 0) batch_pairs = 65536;       // (the consumer's per-batch cost -- a dozen runtime calls -- is what limits it once the decode is fast)
    const double t0 = now_s();
    // capacity guess from the compressed sizes (a read is ~330 bytes of text, FASTH deflates ~4x); the arrays grow if it is wrong
    uint64_t comp = 0;
    for (uint32_t i = 0; i < n_files; ++i) { FILE* f = fopen(paths[i], "rb"); if (f) { fseek(f, 0, SEEK_END); const long n = ftell(f); if (n > 0) comp += (uint64_t)n; fclose(f); } }
    uint64_t cap = comp / 70 + 4ull * batch_pairs;
    snk_fasth_stream* fs = nullptr;
    int rc = snk_fasth_open(paths, n_files, stride, batch_pairs, threads, 1u, &fs, err, errcap);
    if (rc) return rc;
    hipStream_t cs = nullptr;
    constexpr int NST = 4;
    uint8_t* st_ascii[NST] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t* st_bcf[NST] = {nullptr, nullptr, nullptr, nullptr};
    int32_t* st_ids[NST] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t st_ev[NST] = {nullptr, nullptr, nullptr, nullptr};
    bool st_busy[NST] = {false, false, false, false};
    ingest_arrays A, Bf;
    struct piece { uint32_t file; uint64_t first_pair, n_pairs, at; };
    std::vector<piece> pieces;
    struct pend { hipEvent_t ev; snk_fasth_batch b; };
    std::deque<pend> pending;
    std::vector<hipEvent_t> ev_pool;
    uint64_t n_reads = 0, text = 0;
    uint32_t max_len = 0;
    auto cleanup = [&]() {
        if (cs) (void)hipStreamSynchronize(cs);
        for (auto& p : pending) { (void)hipEventDestroy(p.ev); snk_fasth_release(fs, &p.b); }
        pending.clear();
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear();
        for (int q = 0; q < NST; ++q) { (void)hipFree(st_ascii[q]); (void)hipFree(st_bcf[q]); (void)hipFree(st_ids[q]); if (st_ev[q]) (void)hipEventDestroy(st_ev[q]); }
        if (cs) { if (ctx->cur_stream == cs) ctx->cur_stream = nullptr; (void)hipStreamDestroy(cs); }
        if (fs) snk_fasth_close(fs);
    };
#define ING_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); A.release(); Bf.release(); return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
#define ING_RC(expr) do { int _r = (expr); if (_r) { cleanup(); A.release(); Bf.release(); return _r; } } while (0)
    ING_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (int q = 0; q < NST; ++q) {
        ING_TRY(hipMalloc((void**)&st_ascii[q], 2ull * batch_pairs * stride));
        ING_TRY(hipMalloc((void**)&st_bcf[q], (size_t)batch_pairs * 64));
        ING_TRY(hipMalloc((void**)&st_ids[q], (size_t)batch_pairs * 4));
        ING_TRY(hipEventCreateWithFlags(&st_ev[q], hipEventDisableTiming));
    }
    ING_RC(alloc_arrays(A, cap, row_words, qstride, ix != nullptr, err, errcap));
    int slot = 0;
    double wait_s = 0, t_pend = 0, t_slot = 0, t_issue = 0;      // SNK_INGEST_TRACE=1: where the consumer thread's time goes
    const bool trace = getenv("SNK_INGEST_TRACE") && *getenv("SNK_INGEST_TRACE") == '1';
    const double t_ready = now_s();          // decode threads running, page-locked batches and device arrays allocated
    for (;;) {
        // hand back the batches whose copies are done (never more than two outstanding: the workers need them)
        const double p0 = now_s();
        while (!pending.empty() && (pending.size() > 2 || hipEventQuery(pending.front().ev) == hipSuccess)) {
            ING_TRY(hipEventSynchronize(pending.front().ev));
            ev_pool.push_back(pending.front().ev);
            snk_fasth_release(fs, &pending.front().b);
            pending.pop_front();
        }
        snk_fasth_batch b;
        const double w0 = now_s();
        t_pend += w0 - p0;
        ING_RC(snk_fasth_next(fs, &b, err, errcap));
        wait_s += now_s() - w0;
        if (b.n_pairs == 0) break;
        const uint64_t nr = 2 * b.n_pairs;
        if (n_reads + nr > A.cap) {
            ingest_arrays N;
            const uint64_t ncap = A.cap + A.cap / 2 + nr;
            int r2 = alloc_arrays(N, ncap, row_words, qstride, ix != nullptr, err, errcap);
            if (r2) { N.release(); snk_fasth_release(fs, &b); cleanup(); A.release(); return r2; }
            ING_TRY(hipMemcpyAsync(N.rows, A.rows, n_reads * row_words * 4ull, hipMemcpyDeviceToDevice, cs));
            ING_TRY(hipMemcpyAsync(N.quals, A.quals, n_reads * (uint64_t)qstride, hipMemcpyDeviceToDevice, cs));
            ING_TRY(hipMemcpyAsync(N.lens, A.lens, n_reads * 2ull, hipMemcpyDeviceToDevice, cs));
            if (ix) ING_TRY(hipMemcpyAsync(N.bc, A.bc, n_reads * 4ull, hipMemcpyDeviceToDevice, cs));
            ING_TRY(hipStreamSynchronize(cs));
            A.release();
            A = N;
        }
        const double s0 = now_s();
        if (st_busy[slot]) { ING_TRY(hipEventSynchronize(st_ev[slot])); st_busy[slot] = false; }
        const double s1 = now_s();
        t_slot += s1 - s0;
        ING_TRY(hipMemcpyAsync(st_ascii[slot], b.ascii, nr * (uint64_t)stride, hipMemcpyHostToDevice, cs));
        ING_TRY(hipMemcpyAsync(A.quals + n_reads * (uint64_t)qstride, b.quals, nr * (uint64_t)stride, hipMemcpyHostToDevice, cs));
        ING_TRY(hipMemcpyAsync(A.lens + n_reads, b.lens, nr * 2ull, hipMemcpyHostToDevice, cs));
        if (ix) ING_TRY(hipMemcpyAsync(st_bcf[slot], b.bc_fields, b.n_pairs * 64ull, hipMemcpyHostToDevice, cs));
        pend pe;
        if (!ev_pool.empty()) { pe.ev = ev_pool.back(); ev_pool.pop_back(); }
        else ING_TRY(hipEventCreateWithFlags(&pe.ev, hipEventDisableTiming));
        ING_TRY(hipEventRecord(pe.ev, cs));
        pe.b = b;
        pending.push_back(pe);
        ING_RC(snk_dev_pack_ascii(ctx, st_ascii[slot], stride, read_len, nr, A.rows + n_reads * row_words, row_words, cs));
        if (ix) {
            ING_RC(snk_dev_bc_ids(ctx, ix, st_bcf[slot], 64, b.n_pairs, st_ids[slot], cs, err, errcap));
            hipLaunchKernelGGL(pair_ids_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, cs, st_ids[slot], b.n_pairs, A.bc + n_reads);
        }
        ING_TRY(hipEventRecord(st_ev[slot], cs));
        st_busy[slot] = true;
        slot = (slot + 1) % NST;
        pieces.push_back({b.file, b.first_pair, b.n_pairs, n_reads});
        n_reads += nr;
        text += b.text_bytes;
        if (b.max_len > max_len) max_len = b.max_len;
        t_issue += now_s() - s1;
    }
    const double t_loop = now_s();
    ING_TRY(hipStreamSynchronize(cs));
    if (trace) fprintf(stderr, "[snk ingest] setup %.3f s | loop %.3f s: wait for decode %.3f, wait for copies (batch hand-back) %.3f, wait for a staging slot %.3f, issue %.3f | drain %.3f s | %zu batches\n",
                       t_ready - t0, t_loop - t_ready, wait_s, t_pend, t_slot, t_issue, now_s() - t_loop, pieces.size());
    while (!pending.empty()) { (void)hipEventDestroy(pending.front().ev); snk_fasth_release(fs, &pending.front().b); pending.pop_front(); }
    // ---- file-major order
    std::vector<uint64_t> fbase(n_files + 1, 0);
    for (uint32_t i = 0; i < n_files; ++i) fbase[i + 1] = fbase[i] + 2 * snk_fasth_file_pairs(fs, i);
    bool in_order = true;
    for (auto& p : pieces) if (fbase[p.file] + 2 * p.first_pair != p.at) { in_order = false; break; }
    if (!in_order) {
        ING_RC(alloc_arrays(Bf, n_reads, row_words, qstride, ix != nullptr, err, errcap));
        for (auto& p : pieces) {
            const uint64_t d = fbase[p.file] + 2 * p.first_pair, s = p.at, nr = 2 * p.n_pairs;
            ING_TRY(hipMemcpyAsync(Bf.rows + d * row_words, A.rows + s * row_words, nr * row_words * 4ull, hipMemcpyDeviceToDevice, cs));
            ING_TRY(hipMemcpyAsync(Bf.quals + d * qstride, A.quals + s * qstride, nr * (uint64_t)qstride, hipMemcpyDeviceToDevice, cs));
            ING_TRY(hipMemcpyAsync(Bf.lens + d, A.lens + s, nr * 2ull, hipMemcpyDeviceToDevice, cs));
            if (ix) ING_TRY(hipMemcpyAsync(Bf.bc + d, A.bc + s, nr * 4ull, hipMemcpyDeviceToDevice, cs));
        }
        ING_TRY(hipStreamSynchronize(cs));
        A.release();
        A = Bf;
        Bf = ingest_arrays();
    }
    cleanup();
    out->n_reads = n_reads; out->read_len = read_len; out->row_words = row_words; out->qstride = qstride; out->max_len = max_len;
    out->rows = A.rows; out->quals = A.quals; out->lens = A.lens; out->bc = A.bc;
    out->text_bytes = text; out->compressed_bytes = comp; out->n_files = n_files;
    out->seconds = now_s() - t0; out->decode_wait_seconds = wait_s; out->setup_seconds = t_ready - t0; out->n_batches = (uint32_t)pieces.size();
    return SNK_OK;
#undef ING_TRY
#undef ING_RC
}

// FASTH files -> count + graph with the reads never resident as a whole: every decoded batch goes up, is packed, gets its barcode ids
// and is appended to a streamed job (snk_dev_stream_*, snk_pipeline.hip) -- partitioned into the job's minimiser buckets while the
// workers inflate the next batches; finish() counts and builds the graph.  Wall time = max(ingest, partition) + count + graph, not
// their sum; the device holds the supermer records, not the reads.  total_reads_hint: an upper bound of the job's reads (it sizes the
// bucket slots); 0 = derived from the compressed sizes.
extern "C" int snk_dev_ingest_count_graph(snk_ctx* ctx, const char* const* paths, uint32_t n_files, uint32_t read_len, const snk_bc_index* ix, uint32_t threads,
                                          uint32_t batch_pairs, uint64_t total_reads_hint, const snk_params* p, snk_dev_result* res, snk_dev_ingest* out, char* err,
                                          size_t errcap) {
    if (!ctx || !paths || !out || !p || !res || n_files == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_count_graph: NULL argument");
    if (read_len == 0 || read_len > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_count_graph: read_len must be 1..256");
    memset(out, 0, sizeof *out);
    SNK_HIP_TRY(snk_enter(ctx));
    const uint32_t stride = (read_len + 15) / 16 * 16, row_words = (read_len + 15) / 16, qstride = stride;
    if (batch_pairs == 0) batch_pairs = 65536;
    const double t0 = now_s();
    uint64_t comp = 0;
    uint64_t isize_sum = 0;      // text bytes by the files' gzip trailers (exact for the one-member files the reference writes, below 4 GB each)
    for (uint32_t i = 0; i < n_files; ++i) {
        struct stat sb;
        if (stat(paths[i], &sb) == 0 && sb.st_size > 0) {
            comp += (uint64_t)sb.st_size;
            if (sb.st_size >= 18) { FILE* f = fopen(paths[i], "rb"); if (f) { uint32_t is = 0; if (fseek(f, -4, SEEK_END) == 0 && fread(&is, 4, 1, f) == 1) isize_sum += is; fclose(f); } }
        }
    }
    // a read pair is ~650 bytes of text that deflate to ~130; 45 compressed bytes per read is a bound with a margin of a third -- unless the
    // files compress better than usual (binned qualities): the trailers' text sizes bound the reads as well (a read is at least its bases
    // and qualities with their line ends), and the larger bound counts (the job used to fail late on such files)
    const uint64_t ub_isize = isize_sum / (2ull * read_len + 2);
    const uint64_t ub = total_reads_hint ? total_reads_hint : std::max(comp / 45, ub_isize) + 8ull * batch_pairs;
    snk_fasth_stream* fs = nullptr;
    int rc = snk_fasth_open(paths, n_files, stride, batch_pairs, threads, 1u, &fs, err, errcap);
    if (rc) return rc;
    hipStream_t cs = nullptr;
    constexpr int NST = 4;
    struct slot_t { uint8_t *ascii = nullptr, *bcf = nullptr, *quals = nullptr; int32_t *ids = nullptr, *bc = nullptr; uint32_t* rows = nullptr; uint16_t* lens = nullptr; hipEvent_t ev = nullptr; bool busy = false; };
    slot_t S[NST];
    struct pend { hipEvent_t ev; snk_fasth_batch b; };
    std::deque<pend> pending;
    std::vector<hipEvent_t> ev_pool;
    auto cleanup = [&]() {
        if (cs) (void)hipStreamSynchronize(cs);
        for (auto& q : pending) { (void)hipEventDestroy(q.ev); snk_fasth_release(fs, &q.b); }
        pending.clear();
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        for (auto& q : S) { (void)hipFree(q.ascii); (void)hipFree(q.bcf); (void)hipFree(q.quals); (void)hipFree(q.ids); (void)hipFree(q.bc); (void)hipFree(q.rows); (void)hipFree(q.lens); if (q.ev) (void)hipEventDestroy(q.ev); }
        if (cs) { if (ctx->cur_stream == cs) ctx->cur_stream = nullptr; (void)hipStreamDestroy(cs); }
        if (fs) snk_fasth_close(fs);
    };
#define ING_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
#define ING_RC(expr) do { int _r = (expr); if (_r) { cleanup(); return _r; } } while (0)
    ING_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    const uint64_t nrb = 2ull * batch_pairs;
    for (auto& q : S) {
        ING_TRY(hipMalloc((void**)&q.ascii, nrb * stride));
        ING_TRY(hipMalloc((void**)&q.quals, nrb * qstride));
        ING_TRY(hipMalloc((void**)&q.rows, nrb * row_words * 4ull));
        ING_TRY(hipMalloc((void**)&q.lens, nrb * 2 + 16));
        ING_TRY(hipMalloc((void**)&q.bcf, (size_t)batch_pairs * 64));
        ING_TRY(hipMalloc((void**)&q.ids, (size_t)batch_pairs * 4));
        ING_TRY(hipMalloc((void**)&q.bc, nrb * 4));
        ING_TRY(hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
    }
    ING_RC(snk_dev_stream_begin(ctx, p, read_len, ub, ix ? 1 : 0, cs, err, errcap));
    int slot = 0;
    double wait_s = 0;
    uint64_t n_reads = 0, text = 0;
    uint32_t max_len = 0, n_batches = 0;
    const double t_ready = now_s();
    for (;;) {
        while (!pending.empty() && (pending.size() > 2 || hipEventQuery(pending.front().ev) == hipSuccess)) {
            ING_TRY(hipEventSynchronize(pending.front().ev));
            ev_pool.push_back(pending.front().ev);
            snk_fasth_release(fs, &pending.front().b);
            pending.pop_front();
        }
        snk_fasth_batch b;
        const double w0 = now_s();
        ING_RC(snk_fasth_next(fs, &b, err, errcap));
        wait_s += now_s() - w0;
        if (b.n_pairs == 0) break;
        const uint64_t nr = 2 * b.n_pairs;
        slot_t& q = S[slot];
        if (q.busy) { ING_TRY(hipEventSynchronize(q.ev)); q.busy = false; }        // the partition launch that read this slot is done
        ING_TRY(hipMemcpyAsync(q.ascii, b.ascii, nr * (uint64_t)stride, hipMemcpyHostToDevice, cs));
        ING_TRY(hipMemcpyAsync(q.quals, b.quals, nr * (uint64_t)stride, hipMemcpyHostToDevice, cs));
        ING_TRY(hipMemcpyAsync(q.lens, b.lens, nr * 2ull, hipMemcpyHostToDevice, cs));
        if (ix) ING_TRY(hipMemcpyAsync(q.bcf, b.bc_fields, b.n_pairs * 64ull, hipMemcpyHostToDevice, cs));
        pend pe;
        if (!ev_pool.empty()) { pe.ev = ev_pool.back(); ev_pool.pop_back(); }
        else ING_TRY(hipEventCreateWithFlags(&pe.ev, hipEventDisableTiming));
        ING_TRY(hipEventRecord(pe.ev, cs));
        pe.b = b;
        pending.push_back(pe);
        ING_RC(snk_dev_pack_ascii(ctx, q.ascii, stride, read_len, nr, q.rows, row_words, cs));
        if (ix) {
            ING_RC(snk_dev_bc_ids(ctx, ix, q.bcf, 64, b.n_pairs, q.ids, cs, err, errcap));
            hipLaunchKernelGGL(pair_ids_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, cs, q.ids, b.n_pairs, q.bc);
        }
        snk_dev_reads slab;
        memset(&slab, 0, sizeof slab);
        slab.n_reads = nr; slab.rows = q.rows; slab.row_words = row_words; slab.read_len = read_len; slab.lens = q.lens; slab.quals = q.quals; slab.qstride = qstride;
        slab.bc = ix ? q.bc : nullptr;
        ING_RC(snk_dev_stream_append(ctx, &slab, cs, err, errcap));
        ING_TRY(hipEventRecord(q.ev, cs));
        q.busy = true;
        slot = (slot + 1) % NST;
        n_reads += nr;
        text += b.text_bytes;
        ++n_batches;
        if (b.max_len > max_len) max_len = b.max_len;
    }
    const double t_decoded = now_s();
    ING_RC(snk_dev_stream_finish(ctx, res, cs, err, errcap));
    ING_TRY(hipStreamSynchronize(cs));
    while (!pending.empty()) { (void)hipEventDestroy(pending.front().ev); snk_fasth_release(fs, &pending.front().b); pending.pop_front(); }
    cleanup();
    out->n_reads = n_reads; out->read_len = read_len; out->row_words = row_words; out->qstride = qstride; out->max_len = max_len;
    out->text_bytes = text; out->compressed_bytes = comp; out->n_files = n_files; out->n_batches = n_batches;
    out->seconds = now_s() - t0; out->decode_wait_seconds = wait_s; out->setup_seconds = t_ready - t0;
    (void)t_decoded;
    return SNK_OK;
#undef ING_TRY
#undef ING_RC
}

extern "C" void snk_dev_ingest_free(snk_dev_ingest* r) {
    if (!r) return;
    (void)hipFree((void*)r->rows); (void)hipFree((void*)r->quals); (void)hipFree((void*)r->lens); (void)hipFree((void*)r->bc); (void)hipFree((void*)r->good_len);
    r->rows = r->quals = r->lens = r->bc = r->good_len = nullptr;
}
