// snk_dfin.hip -- the stage inputs of ASSEMBLER_DF decoded ON THE DEVICE (b1/b2 seam at rate).
//
// The reference loads reads.fastb with bases.ReadAll and walks reads.qualp through VirtualMasterVec<PQVec> (lib/assembly/src/10X/DF.cc:
// 265-272,345,595-597; block codec feudal/PQVec.cc:86-200; control block feudal/FeudalControlBlock.h:27-166; barcode index expansion
// DF.cc:464-469).  Both files are offset-indexed and uncompressed: a read's bytes are found without looking at any other read.  So the
// file bytes themselves go to the device -- raw byte ranges of a slab of reads, preads of a thread pool into a page-locked ring, one
// asynchronous copy per slab -- and three kernels turn them into the arrays the count+graph path takes:
//   df_bases_kernel   fastb bytes (2 bits per base, base j at bits 2(j%4) of byte j/4: LSB first) -> packed rows (MSB-first words): one
//                     output word = one unaligned 32-bit load, a reversal of its sixteen 2-bit groups, a mask behind the read's length
//   df_quals_kernel   PQVec block chains -> raw phred rows: 16 lanes per read walk the chain together (the header bytes are a broadcast
//                     load), a lane per value inside a block; the rows of a workgroup's 16 reads are put together in LDS and stored as
//                     one contiguous run of dwords
//   df_bc_kernel      reads.bci (read range of every barcode ordinal) -> one id per read: a binary search in the index, -1 outside it
// A rank of the N-GPU job maps only the byte ranges of ITS slab (the offset tables say where they are).  Nothing here parses on the
// host: the host's part is the control blocks, the barcode index (a few MB) and moving bytes.
// The old one-thread host readers (snk_formats.hip) stay as the byte-for-byte check of these kernels (tests/test_gpu_dfin.py).
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "snk_common.h"
#include "snk_ctx.h"
#include "snk_synth.h"

namespace {

#pragma pack(push, 1)
struct fcb_t {          // FeudalControlBlock, 24 bytes (feudal/FeudalControlBlock.h:157-166)
    uint32_t n;
    uint8_t flags, sizeof_fixed, sizeof_x, sizeof_a;
    uint64_t var_offset, fixed_offset;
};
#pragma pack(pop)
static_assert(sizeof(fcb_t) == 24, "feudal control block is 24 bytes");

double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

bool pread_all(int fd, void* dst, size_t len, uint64_t off) {
    char* d = (char*)dst;
    while (len) {
        const ssize_t k = pread(fd, d, len, (off_t)off);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        if (k == 0) return false;
        d += k; off += (uint64_t)k; len -= (size_t)k;
    }
    return true;
}

struct feudal_file {
    int fd = -1;
    std::string path;
    uint64_t size = 0, n = 0, var_offset = 0, fixed_offset = 0;
    uint32_t sizeof_fixed = 0;
};

int open_feudal(const char* path, feudal_file* f, char* err, size_t errcap) {
    f->path = path;
    f->fd = open(path, O_RDONLY);
    if (f->fd < 0) return snk_fail(SNK_E_IO, err, errcap, "cannot open %s", path);
    struct stat sb;
    if (fstat(f->fd, &sb) != 0) return snk_fail(SNK_E_IO, err, errcap, "cannot stat %s", path);
    f->size = (uint64_t)sb.st_size;
    fcb_t h;
    if (f->size < sizeof h || !pread_all(f->fd, &h, sizeof h, 0)) return snk_fail(SNK_E_IO, err, errcap, "%s: too short for a feudal file", path);
    if ((h.flags & 3) != 1 || (h.flags & 4)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "%s: 3-file or compressed feudal files are not supported", path);
    if (h.var_offset < sizeof(fcb_t) || h.fixed_offset < h.var_offset || h.fixed_offset > f->size || (h.fixed_offset - h.var_offset) % 8 || h.fixed_offset == h.var_offset)
        return snk_fail(SNK_E_IO, err, errcap, "%s: inconsistent feudal control block", path);
    f->n = (h.fixed_offset - h.var_offset) / 8 - 1;
    f->var_offset = h.var_offset;
    f->fixed_offset = h.fixed_offset;
    f->sizeof_fixed = h.sizeof_fixed;
    return SNK_OK;
}

// ---- a pool of threads that move file bytes (pread out of the page cache is a memcpy by the kernel: one thread moves 3-5 GB/s)
struct io_pool {
    struct task { int fd; uint64_t off; void* dst; size_t len; std::atomic<int>* left; std::atomic<int>* failed; };
    std::vector<std::thread> th;
    std::deque<task> q;
    std::mutex m;
    std::condition_variable cv, done_cv;
    bool stop = false;
    explicit io_pool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) th.emplace_back([this] { run(); });
    }
    ~io_pool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void run() {
        for (;;) {
            task t;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                t = q.front();
                q.pop_front();
            }
            if (!pread_all(t.fd, t.dst, t.len, t.off)) t.failed->store(1);
            if (t.left->fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(m); done_cv.notify_all(); }
        }
    }
    // [off, off + len) of fd -> dst in pieces; `left` counts the pieces still out
    void read(int fd, uint64_t off, void* dst, size_t len, std::atomic<int>* left, std::atomic<int>* failed) {
        constexpr size_t PIECE = 2u << 20;
        const size_t np = (len + PIECE - 1) / PIECE;
        if (!np) return;
        left->fetch_add((int)np);
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < np; ++i) q.push_back({fd, off + i * PIECE, (char*)dst + i * PIECE, std::min(PIECE, len - i * PIECE), left, failed});
        }
        cv.notify_all();
    }
    void wait(std::atomic<int>* left) {
        std::unique_lock<std::mutex> g(m);
        done_cv.wait(g, [left] { return left->load() == 0; });
    }
};

// ---- kernels ------------------------------------------------------------------------------------------------------------------
// errs: [0] reads whose length does not fit their bytes / the rows, [1] quality chains cut off by the end of their element, [2] quality
// chains longer than the row, [3] offset tables that run backwards or out of the section; [4..5] u64: lowest offending read
__device__ __forceinline__ void df_flag(uint32_t* errs, int which, uint64_t read) {
    atomicAdd(&errs[which], 1u);
    atomicMin((unsigned long long*)(errs + 4), (unsigned long long)read);
}

// 32 bits at byte offset s of a 4-byte aligned buffer (the buffer is padded: the second word may be read)
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint32_t* __restrict__ base, uint64_t s) {
    const uint64_t i = s >> 2;
    const uint32_t sh = (uint32_t)(s & 3) * 8;
    const uint32_t lo = base[i];
    if (!sh) return lo;
    return (lo >> sh) | (base[i + 1] << (32 - sh));
}

// thread per output word: row word w of read i
__global__ void __launch_bounds__(256) df_bases_kernel(const uint64_t* __restrict__ offs, const uint32_t* __restrict__ flens, const uint32_t* __restrict__ data,
                                                       uint64_t o0, uint64_t data_bytes, uint64_t n, uint32_t rw, uint32_t max_len, uint64_t first,
                                                       uint32_t* __restrict__ rows, uint16_t* __restrict__ lens, uint32_t* __restrict__ errs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i = t / rw;
    const uint32_t w = (uint32_t)(t - i * rw);
    if (i >= n) return;
    const uint32_t L = flens[i];
    const uint64_t a = offs[i], b = offs[i + 1];
    bool ok = a >= o0 && b >= a && b - o0 <= data_bytes;
    if (!ok) { if (w == 0) df_flag(errs, 3, first + i); }
    else if (L > max_len || (uint64_t)(L + 3) / 4 > b - a) { ok = false; if (w == 0) df_flag(errs, 0, first + i); }
    if (w == 0) lens[i] = ok ? (uint16_t)L : (uint16_t)0;
    uint32_t y = 0;
    if (ok && 16u * w < L) {
        const uint32_t x = load_u32_unaligned(data, a - o0 + 4ull * w);
        // base j of the word sits at bits 2j (bytes LSB first, feudal/FieldVec.h:586-603); the rows want it at bits 30 - 2j
        y = __brev(x);
        y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
        const uint32_t rem = L - 16u * w;
        if (rem < 16) y &= ~0u << (32 - 2 * rem);
    }
    rows[i * rw + w] = y;
}

// 16 lanes per read, 16 reads per workgroup; rows are put together in LDS and leave as one run of dwords
template <int LPR>
__global__ void __launch_bounds__(256) df_quals_kernel(const uint64_t* __restrict__ qoffs, const uint8_t* __restrict__ qd, uint64_t q0, uint64_t data_bytes,
                                                       uint64_t n, uint32_t qstride, uint64_t first, uint8_t* __restrict__ quals, uint32_t* __restrict__ errs) {
    constexpr int RPB = 256 / LPR;
    extern __shared__ uint32_t lds32[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(lds32);
    const uint32_t g = threadIdx.x / LPR, l = threadIdx.x % LPR;
    const uint64_t i0 = (uint64_t)blockIdx.x * RPB, i = i0 + g;
    const uint32_t row_dw = qstride / 4;
    for (uint32_t k = threadIdx.x; k < RPB * row_dw; k += 256) lds32[k] = 0;
    __syncthreads();
    if (i < n) {
        const uint64_t a = qoffs[i], b = qoffs[i + 1];
        if (a < q0 || b < a || b - q0 > data_bytes) { if (l == 0) df_flag(errs, 3, first + i); }
        else {
            uint64_t p = a - q0;
            const uint64_t e = b - q0;
            uint32_t w = 0;
            uint8_t* row = lds + g * qstride;
            while (p < e) {                                   // chain of blocks, a 0 byte ends it (PQVec.cc:86-127)
                const uint32_t nqs = qd[p];
                if (!nqs) break;
                ++p;
                if (p + 2 > e) { if (l == 0) df_flag(errs, 1, first + i); break; }
                const uint32_t h0 = qd[p], h1 = qd[p + 1];
                const uint32_t nbits = h0 & 7u, minq = ((h0 >> 3) | ((h1 & 1u) << 5)) & 63u;
                const uint32_t blk = (nqs * nbits + 9 + 7) / 8;     // bytes after the nQs byte
                if (p + blk > e) { if (l == 0) df_flag(errs, 1, first + i); break; }
                if (w + nqs > qstride) { if (l == 0) df_flag(errs, 2, first + i); break; }
                const uint32_t mask = (1u << nbits) - 1u;
                for (uint32_t k = l; k < nqs; k += LPR) {
                    const uint32_t bit = 9 + k * nbits;
                    const uint64_t at = p + (bit >> 3);
                    const uint32_t two = (uint32_t)qd[at] | ((uint32_t)qd[at + 1] << 8);      // nbits <= 7: a value spans at most two bytes
                    row[w + k] = (uint8_t)(minq + ((two >> (bit & 7)) & mask));
                }
                w += nqs;
                p += blk;
            }
        }
    }
    __syncthreads();
    const uint64_t rows_here = n - i0 < (uint64_t)RPB ? n - i0 : (uint64_t)RPB;
    uint32_t* out = reinterpret_cast<uint32_t*>(quals + i0 * qstride);
    for (uint32_t k = threadIdx.x; k < rows_here * row_dw; k += 256) out[k] = lds32[k];
}

// bc[i] = the ordinal b with bci[b] <= first + i < bci[b + 1], -1 when no range holds the read (vec<int32_t> bc(bci.back(), -1), DF.cc:464-469)
__global__ void __launch_bounds__(256) df_bc_kernel(const long long* __restrict__ bci, uint32_t m, uint64_t first, uint64_t n, int32_t* __restrict__ bc) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long r = (long long)(first + i);
    uint32_t lo = 0, hi = m;                    // first entry > r
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (bci[mid] <= r) lo = mid + 1; else hi = mid; }
    bc[i] = (lo == 0 || lo >= m) ? -1 : (int32_t)(lo - 1);
}

__global__ void __launch_bounds__(256) df_maxlen_kernel(const uint32_t* __restrict__ flens, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t mx = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) mx = max(mx, flens[i]);
    for (int o = 32; o; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}

constexpr int NSLOT = 3;
constexpr uint64_t QUAL_BYTES_PER_READ = 192;      // slot capacity for a read's quality bytes (a 150-base read with 7-bit values: 135; the planner splits slabs that need more)

// page-locked ring + its device mirror, kept by the context between calls (page-locking memory costs ~0.3 ms per MB)
struct df_io {
    uint64_t slab_reads = 0, slot_bytes = 0;
    uint8_t* pin[NSLOT] = {nullptr, nullptr, nullptr};
    uint8_t* dev[NSLOT] = {nullptr, nullptr, nullptr};
    hipEvent_t used[NSLOT] = {nullptr, nullptr, nullptr};   // the device is done with slot s (its raw bytes AND whatever was decoded next to them)
    bool busy[NSLOT] = {false, false, false};
    uint32_t* d_errs = nullptr;
    long long* d_bci = nullptr;
    uint64_t bci_cap = 0;
    // the compact form of a job's reads (packed rows, good lengths, barcode ids: 46 bytes per 150-base read; the quality rows are gone
    // once a slab is trimmed): what snk_dev_ingest_df_count_graph hands to the resident step when the data are not known to be clean
    uint32_t* c_rows = nullptr; uint16_t* c_gl = nullptr; int32_t* c_bc = nullptr;
    uint64_t c_reads = 0; uint32_t c_row_words = 0;
    void release_compact() { (void)hipFree(c_rows); (void)hipFree(c_gl); (void)hipFree(c_bc); c_rows = nullptr; c_gl = nullptr; c_bc = nullptr; c_reads = 0; c_row_words = 0; }
    hipStream_t cs = nullptr;
    io_pool* pool = nullptr;
    unsigned pool_threads = 0;
    void release_slots() {
        for (int s = 0; s < NSLOT; ++s) {
            if (pin[s]) (void)hipHostFree(pin[s]);
            if (dev[s]) (void)hipFree(dev[s]);
            pin[s] = dev[s] = nullptr;
            busy[s] = false;
        }
        slot_bytes = 0;
    }
    ~df_io() {
        if (cs) (void)hipStreamSynchronize(cs);
        release_slots();
        for (int s = 0; s < NSLOT; ++s) if (used[s]) (void)hipEventDestroy(used[s]);
        if (d_errs) (void)hipFree(d_errs);
        if (d_bci) (void)hipFree(d_bci);
        release_compact();
        if (cs) (void)hipStreamDestroy(cs);
        delete pool;
    }
};

uint64_t align16(uint64_t x) { return (x + 15) & ~15ull; }

}  // namespace

struct snk_df_files {
    feudal_file fb, qp;
    std::vector<long long> bci;        // empty: no barcode index
    bool have_bci = false;
    uint64_t bci_bytes = 0;
    uint32_t max_len = 0;              // 0 = not known yet
    bool max_len_known = false;
};

extern "C" void snk_df_close(snk_df_files* f) {
    if (!f) return;
    if (f->fb.fd >= 0) close(f->fb.fd);
    if (f->qp.fd >= 0) close(f->qp.fd);
    delete f;
}

extern "C" int snk_df_open(const char* fastb, const char* qualp, const char* bci, snk_df_files** out, snk_df_info* info, char* err, size_t errcap) {
    if (!fastb || !qualp || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_df_open: NULL argument");
    *out = nullptr;
    snk_df_files* f = new snk_df_files();
    int rc = open_feudal(fastb, &f->fb, err, errcap);
    if (!rc) rc = open_feudal(qualp, &f->qp, err, errcap);
    if (!rc && f->fb.fixed_offset + 4 * f->fb.n > f->fb.size) rc = snk_fail(SNK_E_IO, err, errcap, "%s: truncated length table", fastb);
    if (!rc && f->qp.n != f->fb.n)
        rc = snk_fail(SNK_E_IO, err, errcap, "%s holds %llu reads, expected %llu", qualp, (unsigned long long)f->qp.n, (unsigned long long)f->fb.n);
    if (!rc && bci) {
        int fd = open(bci, O_RDONLY);
        struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0) { if (fd >= 0) close(fd); rc = snk_fail(SNK_E_IO, err, errcap, "snk_df_open: cannot read %s", bci); }
        else {
            const uint64_t sz = (uint64_t)sb.st_size;
            char head[16];
            uint64_t m = 0;
            if (sz < 16 || !pread_all(fd, head, 16, 0) || memcmp(head, "BINWRITE", 8)) rc = snk_fail(SNK_E_IO, err, errcap, "%s is not a BINWRITE file", bci);
            else {
                memcpy(&m, head + 8, 8);
                if (m < 1 || m > (1ull << 31) || 16 + 8 * m > sz) rc = snk_fail(SNK_E_IO, err, errcap, "%s: truncated index", bci);
            }
            if (!rc) {
                f->bci.resize(m);
                if (!pread_all(fd, f->bci.data(), 8 * m, 16)) rc = snk_fail(SNK_E_IO, err, errcap, "%s: read error", bci);
            }
            close(fd);
            if (!rc && (uint64_t)f->bci.back() != f->fb.n)
                rc = snk_fail(SNK_E_IO, err, errcap, "%s indexes %lld reads, expected %llu", bci, f->bci.back(), (unsigned long long)f->fb.n);
            for (uint64_t b = 0; !rc && b + 1 < m; ++b)
                if (f->bci[b] < 0 || f->bci[b + 1] < f->bci[b] || (uint64_t)f->bci[b + 1] > f->fb.n) rc = snk_fail(SNK_E_IO, err, errcap, "%s: bad range of barcode %llu", bci, (unsigned long long)b);
            f->have_bci = !rc;
            f->bci_bytes = sz;
        }
    }
    if (rc) { snk_df_close(f); return rc; }
    if (info) {
        memset(info, 0, sizeof *info);
        info->n_reads = f->fb.n;
        info->n_barcodes = f->have_bci ? f->bci.size() - 1 : 0;
        info->fastb_bytes = f->fb.size; info->qualp_bytes = f->qp.size; info->bci_bytes = f->bci_bytes;
    }
    *out = f;
    return SNK_OK;
}

namespace {

int io_of(snk_ctx* ctx, df_io** out, char* err, size_t errcap) {
    if (!ctx->df_io) {
        // built in a local owner and published only when every resource exists (a half-made ring must not be what the next call finds)
        std::unique_ptr<df_io> own(new df_io());
        df_io* io = own.get();
        SNK_HIP_TRY(hipStreamCreateWithFlags(&io->cs, hipStreamNonBlocking));
        for (int s = 0; s < NSLOT; ++s) SNK_HIP_TRY(hipEventCreateWithFlags(&io->used[s], hipEventDisableTiming));
        SNK_HIP_TRY(hipMalloc((void**)&io->d_errs, 64));
        ctx->df_io = own.release();
        ctx->df_io_free = [](void* p) { delete static_cast<df_io*>(p); };
    }
    *out = static_cast<df_io*>(ctx->df_io);
    return SNK_OK;
}

struct slab_layout {        // byte offsets of the sections inside a slot (all 16-byte aligned), for `cap` reads
    uint64_t foffs, flens, fdata, qoffs, qdata, end, fdata_cap, qdata_cap;
};
slab_layout layout_for(uint64_t cap, uint32_t max_len) {
    slab_layout L;
    uint64_t at = 0;
    // the three tables, then the packed bases, then -- right behind the bases a slab really has -- the quality bytes: what goes up is ONE
    // contiguous range of the slot per slab (five copies of 1-19 MB each before: ~15 % of the ingest was their set-up and the short ones' rate)
    L.foffs = at; at = align16(at + (cap + 1) * 8);
    L.flens = at; at = align16(at + cap * 4);
    L.qoffs = at; at = align16(at + (cap + 1) * 8);
    L.fdata_cap = cap * ((max_len + 3) / 4 + 1) + 64;
    L.fdata = at; at = align16(at + L.fdata_cap + 16);
    L.qdata_cap = cap * QUAL_BYTES_PER_READ + 4096;
    L.qdata = at; at = align16(at + L.qdata_cap + 16);      // (the latest place the quality bytes may start: a slab's own start is qdata_at())
    L.end = at;
    return L;
}

int ensure_slots(df_io* io, uint64_t slab_reads, uint32_t max_len, unsigned threads, char* err, size_t errcap) {
    const slab_layout L = layout_for(slab_reads, max_len);
    if (io->slot_bytes < L.end || io->slab_reads < slab_reads) {
        SNK_HIP_TRY(hipStreamSynchronize(io->cs));
        io->release_slots();
        for (int s = 0; s < NSLOT; ++s) {
            SNK_HIP_TRY(hipHostMalloc((void**)&io->pin[s], L.end, hipHostMallocDefault));
            SNK_HIP_TRY(hipMalloc((void**)&io->dev[s], L.end));
        }
        io->slot_bytes = L.end;
        io->slab_reads = slab_reads;
    }
    if (!io->pool || io->pool_threads != threads) {
        delete io->pool;
        io->pool = new io_pool(threads);
        io->pool_threads = threads;
    }
    return SNK_OK;
}

// the largest length of reads [first, first + n): a scan of the length table on the device (the table is 4 of a read's ~50 file bytes)
int scan_max_len(snk_ctx* ctx, df_io* io, const snk_df_files* f, uint64_t first, uint64_t n, uint32_t* out, char* err, size_t errcap) {
    *out = 0;
    if (!n) return SNK_OK;
    const uint64_t piece = 16ull << 20;          // reads per piece (64 MB)
    void* pin = nullptr;
    uint32_t* d = nullptr;
    SNK_HIP_TRY(hipHostMalloc(&pin, std::min(piece, n) * 4, hipHostMallocDefault));
    hipError_t e = hipMalloc((void**)&d, std::min(piece, n) * 4);
    if (e != hipSuccess) { (void)hipHostFree(pin); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_dfin: hipMalloc failed"); }
    int rc = SNK_OK;
    (void)hipMemsetAsync(io->d_errs + 8, 0, 4, io->cs);
    io_pool local(8);
    for (uint64_t a = 0; a < n && !rc; a += piece) {
        const uint64_t k = std::min(piece, n - a);
        std::atomic<int> left{0}, failed{0};
        local.read(f->fb.fd, f->fb.fixed_offset + 4 * (first + a), pin, k * 4, &left, &failed);
        local.wait(&left);
        if (failed.load()) { rc = snk_fail(SNK_E_IO, err, errcap, "%s: read error in the length table", f->fb.path.c_str()); break; }
        if (hipMemcpyAsync(d, pin, k * 4, hipMemcpyHostToDevice, io->cs) != hipSuccess) { rc = snk_fail(SNK_E_HIP, err, errcap, "snk_dfin: upload failed"); break; }
        hipLaunchKernelGGL(df_maxlen_kernel, dim3(1024), dim3(256), 0, io->cs, d, k, io->d_errs + 8);
        if (hipStreamSynchronize(io->cs) != hipSuccess) { rc = snk_fail(SNK_E_HIP, err, errcap, "snk_dfin: length scan failed"); break; }
    }
    if (!rc && hipMemcpy(out, io->d_errs + 8, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = snk_fail(SNK_E_HIP, err, errcap, "snk_dfin: download failed");
    (void)hipFree(d);
    (void)hipHostFree(pin);
    return rc;
}

struct slab_dev {           // what the consumer of a decoded slab gets (device memory)
    uint64_t first, n;
    uint32_t* rows; uint8_t* quals; uint16_t* lens; int32_t* bc;
};

struct df_stats { uint64_t file_bytes = 0; uint32_t n_slabs = 0; double wait_io = 0, wait_slot = 0; };

// Drives the slabs of reads [first, first + n): target(k, n_k) says where slab k's arrays go (the resident reader: their final place; the
// streamed job: the slot's own output buffers); consume(slab) runs after the decode kernels are enqueued on io->cs.
int run_slabs(snk_ctx* ctx, df_io* io, const snk_df_files* f, uint64_t first, uint64_t n, uint64_t slab_reads, uint32_t max_len, uint32_t row_words, uint32_t qstride,
              const std::function<int(int slot, uint64_t at, uint64_t k, slab_dev*)>& target, const std::function<int(int slot, const slab_dev&)>& consume, df_stats* st,
              char* err, size_t errcap) {
    const slab_layout L = layout_for(slab_reads, max_len);
    hipStream_t cs = io->cs;
    SNK_HIP_TRY(hipMemsetAsync(io->d_errs, 0, 32, cs));
    {   // lowest offending read starts at "none"
        const unsigned long long none = ~0ull;
        SNK_HIP_TRY(hipMemcpyAsync(io->d_errs + 4, &none, 8, hipMemcpyHostToDevice, cs));
        SNK_HIP_TRY(hipStreamSynchronize(cs));
    }
    const uint32_t m_bci = (uint32_t)f->bci.size();
    if (f->have_bci) {
        if (io->bci_cap < m_bci) {
            if (io->d_bci) (void)hipFree(io->d_bci);
            io->d_bci = nullptr; io->bci_cap = 0;
            SNK_HIP_TRY(hipMalloc((void**)&io->d_bci, (size_t)m_bci * 8));
            io->bci_cap = m_bci;
        }
        SNK_HIP_TRY(hipMemcpyAsync(io->d_bci, f->bci.data(), (size_t)m_bci * 8, hipMemcpyHostToDevice, cs));
        SNK_HIP_TRY(hipStreamSynchronize(cs));        // (the vector is pageable: the copy is staged, but the stream order is what the kernels need)
    }
    std::atomic<int> failed{0};
    std::atomic<int> left_offs[NSLOT], left_data[NSLOT];
    for (int s = 0; s < NSLOT; ++s) { left_offs[s].store(0); left_data[s].store(0); }
    // whatever way this function is left, no worker still writes through these counters (the next slab's offset tables are asked for ahead)
    struct drain_t { io_pool* p; std::atomic<int>* a; std::atomic<int>* b; ~drain_t() { for (int s = 0; s < NSLOT; ++s) { p->wait(&a[s]); p->wait(&b[s]); } } } drain{io->pool, left_offs, left_data};
    // slab k covers reads [first + k * slab_reads, ...); its offset tables are fetched one slab ahead of its data
    const uint64_t n_slabs = (n + slab_reads - 1) / slab_reads;
    auto slab_n = [&](uint64_t k) { return std::min(slab_reads, n - k * slab_reads); };
    auto fetch_offs = [&](uint64_t k) {
        const int s = (int)(k % NSLOT);
        const uint64_t a = first + k * slab_reads, c = slab_n(k);
        io->pool->read(f->fb.fd, f->fb.var_offset + 8 * a, io->pin[s] + L.foffs, (c + 1) * 8, &left_offs[s], &failed);
        io->pool->read(f->fb.fd, f->fb.fixed_offset + 4 * a, io->pin[s] + L.flens, c * 4, &left_offs[s], &failed);
        io->pool->read(f->qp.fd, f->qp.var_offset + 8 * a, io->pin[s] + L.qoffs, (c + 1) * 8, &left_offs[s], &failed);
    };
    auto wait_slot = [&](int s) -> int {
        if (io->busy[s]) { const double t0 = now_s(); SNK_HIP_TRY(hipEventSynchronize(io->used[s])); io->busy[s] = false; if (st) st->wait_slot += now_s() - t0; }
        return SNK_OK;
    };
    int rc;
    if (n_slabs) { if ((rc = wait_slot(0))) return rc; fetch_offs(0); }
    for (uint64_t k = 0; k < n_slabs; ++k) {
        const int s = (int)(k % NSLOT);
        const uint64_t a = first + k * slab_reads, c = slab_n(k);
        double t0 = now_s();
        io->pool->wait(&left_offs[s]);
        if (st) st->wait_io += now_s() - t0;
        if (failed.load()) return snk_fail(SNK_E_IO, err, errcap, "%s / %s: read error in the offset tables", f->fb.path.c_str(), f->qp.path.c_str());
        const uint64_t* fo = reinterpret_cast<const uint64_t*>(io->pin[s] + L.foffs);
        const uint64_t* qo = reinterpret_cast<const uint64_t*>(io->pin[s] + L.qoffs);
        // sub-slabs: normally one; a slab whose bytes do not fit the slot's data sections is decoded in pieces over the same offset tables
        uint64_t done = 0;
        while (done < c) {
            uint64_t take = c - done;
            auto bytes_ok = [&](uint64_t t) {
                const uint64_t f0 = fo[done], f1 = fo[done + t], q0 = qo[done], q1 = qo[done + t];
                return f1 >= f0 && q1 >= q0 && f1 - f0 <= L.fdata_cap && q1 - q0 <= L.qdata_cap;
            };
            const uint64_t f0 = fo[done], q0 = qo[done];
            if (f0 < sizeof(fcb_t) || f0 > f->fb.var_offset || q0 < sizeof(fcb_t) || q0 > f->qp.var_offset)
                return snk_fail(SNK_E_IO, err, errcap, "%s / %s: element offset out of range (read %llu)", f->fb.path.c_str(), f->qp.path.c_str(), (unsigned long long)(a + done));
            if (!bytes_ok(take)) {
                uint64_t lo = 0, hi = take;         // largest t with bytes_ok(t); a table that runs backwards ends up at t = 0 -> error
                while (lo < hi) { const uint64_t mid = (lo + hi + 1) / 2; if (bytes_ok(mid)) lo = mid; else hi = mid - 1; }
                take = lo;
                if (!take) return snk_fail(SNK_E_IO, err, errcap, "%s / %s: read %llu has a bad offset or more bytes than a slab holds", f->fb.path.c_str(), f->qp.path.c_str(), (unsigned long long)(a + done));
            }
            const uint64_t f1 = fo[done + take], q1 = qo[done + take];
            if (f1 > f->fb.var_offset || q1 > f->qp.var_offset)
                return snk_fail(SNK_E_IO, err, errcap, "%s / %s: element offset out of range (read %llu)", f->fb.path.c_str(), f->qp.path.c_str(), (unsigned long long)(a + done + take));
            if (done) {        // a later piece of a split slab reuses the slot's data sections: the device must be through with the piece before
                SNK_HIP_TRY(hipEventRecord(io->used[s], cs));
                SNK_HIP_TRY(hipEventSynchronize(io->used[s]));
            }
            const uint64_t qat = align16(L.fdata + (f1 - f0) + 16);            // this piece's quality bytes start right behind its bases
            io->pool->read(f->fb.fd, f0, io->pin[s] + L.fdata, f1 - f0, &left_data[s], &failed);
            io->pool->read(f->qp.fd, q0, io->pin[s] + qat, q1 - q0, &left_data[s], &failed);
            // the next slab's offset tables ride along (its slot must be free of the device first)
            if (done == 0 && k + 1 < n_slabs) { if ((rc = wait_slot((int)((k + 1) % NSLOT)))) return rc; fetch_offs(k + 1); }
            t0 = now_s();
            io->pool->wait(&left_data[s]);
            if (st) st->wait_io += now_s() - t0;
            if (failed.load()) return snk_fail(SNK_E_IO, err, errcap, "%s / %s: read error", f->fb.path.c_str(), f->qp.path.c_str());
            // ---- up and through the kernels
            uint8_t* P = io->pin[s];
            uint8_t* D = io->dev[s];
#ifdef SNK_DF_MULTI_COPY      // (A/B of the single copy, tuning builds)
            if (done == 0) {
                SNK_HIP_TRY(hipMemcpyAsync(D + L.foffs, P + L.foffs, (c + 1) * 8, hipMemcpyHostToDevice, cs));
                SNK_HIP_TRY(hipMemcpyAsync(D + L.flens, P + L.flens, c * 4, hipMemcpyHostToDevice, cs));
                SNK_HIP_TRY(hipMemcpyAsync(D + L.qoffs, P + L.qoffs, (c + 1) * 8, hipMemcpyHostToDevice, cs));
            }
            if (f1 > f0) SNK_HIP_TRY(hipMemcpyAsync(D + L.fdata, P + L.fdata, f1 - f0, hipMemcpyHostToDevice, cs));
            if (q1 > q0) SNK_HIP_TRY(hipMemcpyAsync(D + qat, P + qat, q1 - q0 + 16, hipMemcpyHostToDevice, cs));
#else
            {   // one copy: from the tables (first piece) or from the bases (later pieces of a split slab) to the end of the quality bytes
                const uint64_t from = done == 0 ? 0 : L.fdata, to = qat + (q1 - q0) + 16;
                SNK_HIP_TRY(hipMemcpyAsync(D + from, P + from, to - from, hipMemcpyHostToDevice, cs));
            }
#endif
            slab_dev sd;
            if ((rc = target(s, a - first + done, take, &sd))) return rc;
            sd.first = a + done; sd.n = take;
            const uint64_t words = take * row_words;
            hipLaunchKernelGGL(df_bases_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, cs, reinterpret_cast<const uint64_t*>(D + L.foffs) + done,
                               reinterpret_cast<const uint32_t*>(D + L.flens) + done, reinterpret_cast<const uint32_t*>(D + L.fdata), f0, f1 - f0, take, row_words,
                               max_len, sd.first, sd.rows, sd.lens, io->d_errs);
            hipLaunchKernelGGL((df_quals_kernel<16>), dim3((unsigned)((take + 15) / 16)), dim3(256), 16 * qstride, cs, reinterpret_cast<const uint64_t*>(D + L.qoffs) + done,
                               D + qat, q0, q1 - q0, take, qstride, sd.first, sd.quals, io->d_errs);
            if (sd.bc) hipLaunchKernelGGL(df_bc_kernel, dim3((unsigned)((take + 255) / 256)), dim3(256), 0, cs, io->d_bci, m_bci, sd.first, take, sd.bc);
            SNK_HIP_TRY(hipGetLastError());
            if ((rc = consume(s, sd))) return rc;
            if (st) { st->file_bytes += (f1 - f0) + (q1 - q0) + (done == 0 ? (c + 1) * 16 + c * 4 : 0); ++st->n_slabs; }
            done += take;
        }
        SNK_HIP_TRY(hipEventRecord(io->used[s], cs));
        io->busy[s] = true;
    }
    return SNK_OK;
}

int check_errs(df_io* io, const snk_df_files* f, uint32_t qstride, char* err, size_t errcap) {
    uint32_t h[8];
    SNK_HIP_TRY(hipMemcpyAsync(h, io->d_errs, 32, hipMemcpyDeviceToHost, io->cs));
    SNK_HIP_TRY(snk_sync(io->cs));
    unsigned long long at;
    memcpy(&at, h + 4, 8);
    if (h[3]) return snk_fail(SNK_E_IO, err, errcap, "%s / %s: element offset out of range (%u reads, first: read %llu)", f->fb.path.c_str(), f->qp.path.c_str(), h[3], at);
    if (h[0]) return snk_fail(SNK_E_IO, err, errcap, "%s: %u reads have a bad length (first: read %llu)", f->fb.path.c_str(), h[0], at);
    if (h[1]) return snk_fail(SNK_E_IO, err, errcap, "%s: truncated quality block (%u reads, first: read %llu)", f->qp.path.c_str(), h[1], at);
    if (h[2]) return snk_fail(SNK_E_ARG, err, errcap, "%s: %u reads have more quality values than a row of %u holds (first: read %llu)", f->qp.path.c_str(), h[2], qstride, at);
    return SNK_OK;
}

unsigned pick_threads(uint32_t threads) {
    if (threads) return std::min<uint32_t>(threads, 256);
    const uint32_t b = snk_host_cpu_budget();
    return std::max(2u, std::min(32u, b > 2 ? b - 2 : 2u));
}

int prepare(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t* slab_reads, df_io** io_out, uint32_t* max_len,
            char* err, size_t errcap) {
    if (first > f->fb.n || n > f->fb.n - first) return snk_fail(SNK_E_ARG, err, errcap, "snk_dfin: reads [%llu, +%llu) are not inside the file's %llu", (unsigned long long)first,
                                                               (unsigned long long)n, (unsigned long long)f->fb.n);
    SNK_HIP_TRY(snk_enter(ctx));
    df_io* io;
    int rc = io_of(ctx, &io, err, errcap);
    if (rc) return rc;
    uint32_t mx = read_len;
    if (!mx) {
        if (!f->max_len_known) {       // (the whole file's, so that every rank of a job lays its rows out alike)
            if ((rc = scan_max_len(ctx, io, f, 0, f->fb.n, &f->max_len, err, errcap))) return rc;
            f->max_len_known = true;
        }
        mx = f->max_len ? f->max_len : 1;
    }
    if (mx > 256) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "reads longer than 256 bases are not supported (%u)", mx);
    if (*slab_reads == 0) *slab_reads = 256u << 10;
    *slab_reads = std::max<uint64_t>(16, std::min<uint64_t>(*slab_reads, 4u << 20)) & ~1ull;
    if ((rc = ensure_slots(io, *slab_reads, mx, pick_threads(threads), err, errcap))) return rc;
    *io_out = io;
    *max_len = mx;
    return SNK_OK;
}

// reads [first, first + n) -> their compact form: packed rows, good lengths (a slab's quality rows are trimmed as soon as they are decoded,
// in the caller's per-slot buffers qb / lb, and never kept), barcode ids
int decode_compact(snk_ctx* ctx, df_io* io, const snk_df_files* f, uint64_t first, uint64_t n, uint64_t slab_reads, uint32_t max_len, uint32_t row_words, uint32_t qstride,
                   uint32_t K, uint32_t min_qual, uint8_t* const* qb, uint16_t* const* lb, uint32_t* rows, uint16_t* gl, int32_t* bc, df_stats* st, char* err, size_t errcap) {
    int rc = run_slabs(ctx, io, f, first, n, slab_reads, max_len, row_words, qstride,
                       [&](int s, uint64_t at, uint64_t, slab_dev* sd) { sd->rows = rows + at * row_words; sd->quals = qb[s]; sd->lens = lb[s]; sd->bc = bc ? bc + at : nullptr; return SNK_OK; },
                       [&](int, const slab_dev& sd) {      // the slab's quality rows are used here and never again
                           const int r2 = snk_dev_trim(ctx, sd.quals, qstride, sd.lens, max_len, sd.n, K, min_qual, gl + (sd.first - first), io->cs);
                           return r2 ? snk_fail(r2, err, errcap, "%s", snk_last_error()) : SNK_OK;
                       },
                       st, err, errcap);
    if (!rc) rc = check_errs(io, f, qstride, err, errcap);
    return rc;
}

}  // namespace

extern "C" int snk_df_max_len(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t* out, char* err, size_t errcap) {
    if (!ctx || !f || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_df_max_len: NULL argument");
    if (first > f->fb.n || n > f->fb.n - first) return snk_fail(SNK_E_ARG, err, errcap, "snk_df_max_len: range outside the file");
    SNK_HIP_TRY(snk_enter(ctx));
    df_io* io;
    int rc = io_of(ctx, &io, err, errcap);
    if (rc) return rc;
    return scan_max_len(ctx, io, f, first, n, out, err, errcap);
}

// reads [first, first + n) of the triple -> resident device arrays (plain device allocations: they are the INPUT of snk_dev_count_graph /
// snk_shard_step; snk_dev_ingest_free releases them).  read_len: row length in bases, 0 = the longest read of the FILE.
extern "C" int snk_dev_ingest_df(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads,
                                 snk_dev_ingest* out, char* err, size_t errcap) {
    if (!ctx || !f || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_df: NULL argument");
    memset(out, 0, sizeof *out);
    const double t0 = now_s();
    df_io* io;
    uint32_t max_len;
    int rc = prepare(ctx, f, first, n, read_len, threads, &slab_reads, &io, &max_len, err, errcap);
    if (rc) return rc;
    const uint32_t row_words = (max_len + 15) / 16, qstride = row_words * 16;
    uint32_t* rows = nullptr; uint8_t* quals = nullptr; uint16_t* lens = nullptr; int32_t* bc = nullptr;
    auto drop = [&]() { (void)hipStreamSynchronize(io->cs); (void)hipFree(rows); (void)hipFree(quals); (void)hipFree(lens); (void)hipFree(bc); };
#define DF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { drop(); return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    DF_TRY(hipMalloc((void**)&rows, (n + 1) * row_words * 4ull));
    DF_TRY(hipMalloc((void**)&quals, (n + 16) * (uint64_t)qstride));
    DF_TRY(hipMalloc((void**)&lens, (n + 8) * 2ull));
    if (f->have_bci) DF_TRY(hipMalloc((void**)&bc, (n + 2) * 4ull));
#undef DF_TRY
    const double t_ready = now_s();
    df_stats st;
    rc = run_slabs(ctx, io, f, first, n, slab_reads, max_len, row_words, qstride,
                   [&](int, uint64_t at, uint64_t, slab_dev* sd) { sd->rows = rows + at * row_words; sd->quals = quals + at * qstride; sd->lens = lens + at; sd->bc = bc ? bc + at : nullptr; return SNK_OK; },
                   [&](int, const slab_dev&) { return SNK_OK; }, &st, err, errcap);
    if (!rc) rc = check_errs(io, f, qstride, err, errcap);
    if (rc) { drop(); return rc; }
    out->n_reads = n; out->read_len = max_len; out->row_words = row_words; out->qstride = qstride; out->max_len = max_len;
    out->rows = rows; out->quals = quals; out->lens = lens; out->bc = bc;
    out->text_bytes = st.file_bytes; out->compressed_bytes = st.file_bytes; out->n_files = 3; out->n_batches = st.n_slabs;
    out->seconds = now_s() - t0; out->decode_wait_seconds = st.wait_io; out->setup_seconds = t_ready - t0;
    return SNK_OK;
}

// the same reads in their COMPACT form: packed rows, GOOD LENGTHS (the quality trim at K / min_qual, GoodLenTailFinder BuildReadQGraph48.cc:65-89, run on
// every slab as it is decoded; the quality rows are never resident) and barcode ids -- 46 instead of 204 bytes per 150-base read, and all the
// count+graph step needs (snk_dev_reads.good_len).  What a rank of the N-GPU job holds of its share of the stage inputs (snk_asm_sn LR=).
// out->quals and out->lens stay NULL, out->good_len is set; release with snk_dev_ingest_free.
extern "C" int snk_dev_ingest_df_trimmed(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads, uint32_t K,
                                         uint32_t min_qual, snk_dev_ingest* out, char* err, size_t errcap) {
    if (!ctx || !f || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_df_trimmed: NULL argument");
    memset(out, 0, sizeof *out);
    if (K != 48 && K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
    const double t0 = now_s();
    df_io* io;
    uint32_t max_len;
    int rc = prepare(ctx, f, first, n, read_len, threads, &slab_reads, &io, &max_len, err, errcap);
    if (rc) return rc;
    const uint32_t row_words = (max_len + 15) / 16, qstride = row_words * 16;
    uint32_t* rows = nullptr; uint16_t* gl = nullptr; int32_t* bc = nullptr;
    uint8_t* qb[NSLOT] = {nullptr, nullptr, nullptr}; uint16_t* lb[NSLOT] = {nullptr, nullptr, nullptr};
    auto drop_slots = [&]() { (void)hipStreamSynchronize(io->cs); for (int q = 0; q < NSLOT; ++q) { (void)hipFree(qb[q]); (void)hipFree(lb[q]); } if (ctx->cur_stream == io->cs) ctx->cur_stream = nullptr; };
    auto drop = [&]() { drop_slots(); (void)hipFree(rows); (void)hipFree(gl); (void)hipFree(bc); };
#define DF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { drop(); return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    DF_TRY(hipMalloc((void**)&rows, (n + 1) * row_words * 4ull));
    DF_TRY(hipMalloc((void**)&gl, (n + 8) * 2ull));
    if (f->have_bci) DF_TRY(hipMalloc((void**)&bc, (n + 2) * 4ull));
    for (int q = 0; q < NSLOT; ++q) {
        DF_TRY(hipMalloc((void**)&qb[q], (slab_reads + 16) * (uint64_t)qstride));
        DF_TRY(hipMalloc((void**)&lb[q], (slab_reads + 8) * 2ull));
    }
#undef DF_TRY
    const double t_ready = now_s();
    df_stats st;
    rc = decode_compact(ctx, io, f, first, n, slab_reads, max_len, row_words, qstride, K, min_qual, qb, lb, rows, gl, bc, &st, err, errcap);
    if (rc) { drop(); return rc; }
    drop_slots();
    out->n_reads = n; out->read_len = max_len; out->row_words = row_words; out->qstride = qstride; out->max_len = max_len;
    out->rows = rows; out->good_len = gl; out->bc = bc;
    out->text_bytes = st.file_bytes; out->compressed_bytes = st.file_bytes; out->n_files = 3; out->n_batches = st.n_slabs;
    out->seconds = now_s() - t0; out->decode_wait_seconds = st.wait_io; out->setup_seconds = t_ready - t0;
    return SNK_OK;
}

// the triple -> unitigs.  Two ways (option df_stream; DfFiles.count_graph reports which):
//  * COMPACT (default): a slab's quality rows are trimmed as soon as they are decoded and dropped; the packed rows, good lengths and barcode
//    ids of the job stay (46 bytes per read, context-owned, grow-only) and the RESIDENT step runs on them -- it looks at its first buckets,
//    partitions a second time when their tables run full and picks the count kernel the data want, none of which a streamed job can do once
//    its slabs are gone (100 M reads with 0.6 % / 1.5 % errors, first call of a process: 0.72-0.88 / 1.29-1.32 s streamed against 0.50 / 0.59 s
//    compact; clean reads: 0.46 against 0.43 s; tools/r6_df_errors.py, profiles/r06_df_errors.log);
//  * STREAMED (df_stream = 2): every decoded slab is appended to a streamed job (snk_dev_stream_*), i.e. partitioned while the next slab's
//    bytes are read and copied: the reads are never resident in any form (a job whose 46 bytes per read do not fit next to its records).
// first / n: this caller's reads (the whole file: 0, n_reads); ign_bc_below as snk_dev_reads'.  res: as snk_dev_count_graph's.  stats: rows /
// quals / lens / bc stay NULL; n_files = 3 streamed, 4 compact.
extern "C" int snk_dev_ingest_df_count_graph(snk_ctx* ctx, snk_df_files* f, uint64_t first, uint64_t n, uint32_t read_len, uint32_t threads, uint64_t slab_reads,
                                             const snk_params* p, int64_t ign_bc_below, snk_dev_result* res, snk_dev_ingest* out, char* err, size_t errcap) {
    if (!ctx || !f || !p || !res || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_df_count_graph: NULL argument");
    memset(out, 0, sizeof *out);
    if (n == 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_df_count_graph: no reads");
    if (p->K != 48 && p->K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", p->K);
    if ((p->flags & SNK_F_GROUPED) && !f->have_bci) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_ingest_df_count_graph: per-barcode graphs need the barcode index (reads.bci)");
    const double t0 = now_s();
    df_io* io;
    uint32_t max_len;
    int rc = prepare(ctx, f, first, n, read_len, threads, &slab_reads, &io, &max_len, err, errcap);
    if (rc) return rc;
    const uint32_t row_words = (max_len + 15) / 16, qstride = row_words * 16;
    // (measured, tools/r6_df_errors.py: on clean data the compact form is as fast as the streamed job -- 0.43 against 0.46 s per 100 M reads:
    // 382 small partition launches on the copies' stream cost 52 ms, one resident launch 30 -- and on error-rich data much faster.  So the
    // compact form is what runs unless the caller asks for the streamed job, whose point is that the reads are never resident in any form.)
    const bool streamed = snk_opt_u32("df_stream", 0) == 2 && !(p->flags & SNK_F_GROUPED);
    struct obuf { uint32_t* rows = nullptr; uint8_t* quals = nullptr; uint16_t* lens = nullptr; int32_t* bc = nullptr; } O[NSLOT];
    auto drop = [&]() {
        (void)hipStreamSynchronize(io->cs);
        for (auto& o : O) { (void)hipFree(o.rows); (void)hipFree(o.quals); (void)hipFree(o.lens); (void)hipFree(o.bc); }
        if (ctx->cur_stream == io->cs) ctx->cur_stream = nullptr;
    };
#define DF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { drop(); return snk_fail(_e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, err, errcap, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    for (auto& o : O) {
        if (streamed) DF_TRY(hipMalloc((void**)&o.rows, (slab_reads + 1) * row_words * 4ull));
        DF_TRY(hipMalloc((void**)&o.quals, (slab_reads + 16) * (uint64_t)qstride));
        DF_TRY(hipMalloc((void**)&o.lens, (slab_reads + 8) * 2ull));
        if (streamed && f->have_bci) DF_TRY(hipMalloc((void**)&o.bc, (slab_reads + 2) * 4ull));
    }
    if (!streamed && (io->c_reads < n || io->c_row_words != row_words || (f->have_bci && !io->c_bc))) {
        SNK_HIP_TRY(hipStreamSynchronize(io->cs));
        io->release_compact();
        DF_TRY(hipMalloc((void**)&io->c_rows, (n + 1) * row_words * 4ull));
        DF_TRY(hipMalloc((void**)&io->c_gl, (n + 8) * 2ull));
        DF_TRY(hipMalloc((void**)&io->c_bc, (n + 2) * 4ull));
        io->c_reads = n; io->c_row_words = row_words;
    }
#undef DF_TRY
    df_stats st;
    double t_ready;
    if (streamed) {
        if ((rc = snk_dev_stream_begin(ctx, p, max_len, n, f->have_bci ? 1 : 0, io->cs, err, errcap))) { drop(); return rc; }
        t_ready = now_s();
        rc = run_slabs(ctx, io, f, first, n, slab_reads, max_len, row_words, qstride,
                       [&](int s, uint64_t, uint64_t, slab_dev* sd) { sd->rows = O[s].rows; sd->quals = O[s].quals; sd->lens = O[s].lens; sd->bc = O[s].bc; return SNK_OK; },
                       [&](int, const slab_dev& sd) {
                           snk_dev_reads slab;
                           memset(&slab, 0, sizeof slab);
                           slab.n_reads = sd.n; slab.rows = sd.rows; slab.row_words = row_words; slab.read_len = max_len; slab.lens = sd.lens; slab.quals = sd.quals;
                           slab.qstride = qstride; slab.bc = sd.bc; slab.ign_bc_below = ign_bc_below; slab.read_index_base = sd.first;
                           return snk_dev_stream_append(ctx, &slab, io->cs, err, errcap);
                       },
                       &st, err, errcap);
        // a bad file must not reach the count: the flags are read before the job is finished
        if (!rc) rc = check_errs(io, f, qstride, err, errcap);
        if (!rc) rc = snk_dev_stream_finish(ctx, res, io->cs, err, errcap);
    } else {
        t_ready = now_s();
        uint8_t* qb[NSLOT]; uint16_t* lb[NSLOT];
        for (int q = 0; q < NSLOT; ++q) { qb[q] = O[q].quals; lb[q] = O[q].lens; }
        rc = decode_compact(ctx, io, f, first, n, slab_reads, max_len, row_words, qstride, p->K, p->min_qual, qb, lb, io->c_rows, io->c_gl, f->have_bci ? io->c_bc : nullptr, &st, err, errcap);
        if (!rc) {
            snk_dev_reads in;
            memset(&in, 0, sizeof in);
            in.n_reads = n; in.rows = io->c_rows; in.row_words = row_words; in.read_len = max_len; in.good_len = io->c_gl; in.bc = f->have_bci ? io->c_bc : nullptr;
            in.ign_bc_below = ign_bc_below; in.read_index_base = first;
            // per-barcode graphs (SNK_F_GROUPED, BASELINE config 5): the group of a read is its barcode's ordinal in reads.bci (reads outside
            // every range -- ordinal -1 -- are one group of their own)
            if (p->flags & SNK_F_GROUPED) { in.group = in.bc; in.bc = nullptr; }
            rc = snk_dev_count_graph(ctx, &in, p, res, io->cs, err, errcap);
        }
    }
    if (!rc && hipStreamSynchronize(io->cs) != hipSuccess) rc = snk_fail(SNK_E_HIP, err, errcap, "snk_dev_ingest_df_count_graph: the stream failed");
    drop();
    if (rc) return rc;
    out->n_reads = n; out->read_len = max_len; out->row_words = row_words; out->qstride = qstride; out->max_len = max_len;
    out->text_bytes = st.file_bytes; out->compressed_bytes = st.file_bytes; out->n_files = streamed ? 3 : 4; out->n_batches = st.n_slabs;
    out->seconds = now_s() - t0; out->decode_wait_seconds = st.wait_io; out->setup_seconds = t_ready - t0;
    return SNK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Writers of the triple (tests, bench.py --df-seam, tools): the layouts of feudal/FeudalFileWriter.cc:18-140 (control block, variable data,
// (N + 1) offsets, fixed data), feudal/PQVec.cc:86-127 (block chain) and BinaryWriter::writeFile(vec<int64_t>).  The block choice is this
// file's own (runs of equal values, merged left to right while one block is not dearer than two: pq_encode); any chain of valid blocks decodes to the same
// values, and the reference's own choice is covered by files its writer made (tests/golden/formats, snref_driver ... formats).
namespace {

// one read's qualities -> PQVec bytes appended to `out`; adversarial != 0: random block cuts and more bits than needed (decoder tests)
bool pq_encode(const uint8_t* q, uint32_t L, std::vector<uint8_t>& out, uint64_t adversarial) {
    auto emit = [&](uint32_t s, uint32_t cnt, uint32_t mn, uint32_t nb) {
        out.push_back((uint8_t)cnt);
        uint64_t bits = nb | ((uint64_t)mn << 3);
        uint32_t have = 9;
        for (uint32_t k = 0; k < cnt && nb; ++k) {
            bits |= (uint64_t)(q[s + k] - mn) << have;
            have += nb;
            while (have >= 8) { out.push_back((uint8_t)bits); bits >>= 8; have -= 8; }
        }
        while (have > 0) { out.push_back((uint8_t)bits); bits >>= 8; have = have > 8 ? have - 8 : 0; }
    };
    auto width = [](uint32_t x) { uint32_t b = 0; while (x >> b) ++b; return b; };
    auto cost = [](uint32_t cnt, uint32_t nb) { return 1u + (cnt * nb + 9 + 7) / 8; };
    if (adversarial) {
        uint32_t s = 0;
        uint64_t h = adversarial;
        while (s < L) {
            uint32_t mn = q[s], mx = q[s], cnt = 1;
            h = snk_mix64(h + s);
            const uint32_t limit = adversarial == 1 ? 1u : 1 + (uint32_t)(h % 40);       // 1: a block per value (the longest chains there are)
            while (s + cnt < L && cnt < limit) {
                const uint32_t v = q[s + cnt], nmn = std::min(mn, v), nmx = std::max(mx, v);
                if (width(nmx - std::min(nmn, 63u)) > 7) break;
                mn = nmn; mx = nmx; ++cnt;
            }
            const uint32_t base = std::min(mn, 63u);
            uint32_t nb = width(mx - base);
            if (nb > 7) return false;
            h = snk_mix64(h);
            nb = std::min(7u, nb + (uint32_t)(h % 3));
            emit(s, cnt, base, nb);
            s += cnt;
        }
    } else {
        // runs of equal values are blocks of width 0; two neighbours are joined whenever one block is not dearer than the two (a stack,
        // left to right: linear time, within a few percent of the optimal chain on sequencer-like rows)
        struct blk { uint32_t s, n, mn, mx; };
        blk st[256 + 8];
        std::vector<blk> big;
        blk* stack = st;
        uint32_t top = 0;
        if (L > 256) { big.resize(L + 8); stack = big.data(); }
        auto wid = [&](const blk& b) { return width(b.mx - std::min(b.mn, 63u)); };
        uint32_t i = 0;
        while (i < L) {
            uint32_t j = i + 1;
            while (j < L && q[j] == q[i] && j - i < 255) ++j;
            stack[top++] = {i, j - i, q[i], q[i]};
            while (top >= 2) {
                const blk &a = stack[top - 2], &b = stack[top - 1];
                if (a.n + b.n > 255) break;
                const blk m = {a.s, a.n + b.n, std::min(a.mn, b.mn), std::max(a.mx, b.mx)};
                if (wid(m) > 7 || cost(m.n, wid(m)) > cost(a.n, wid(a)) + cost(b.n, wid(b))) break;
                stack[top - 2] = m;
                --top;
            }
            i = j;
        }
        for (uint32_t k = 0; k < top; ++k) {
            const uint32_t nb = wid(stack[k]);
            if (nb > 7) return false;
            emit(stack[k].s, stack[k].n, std::min(stack[k].mn, 63u), nb);
        }
    }
    out.push_back(0);
    return true;
}

bool write_all(int fd, const void* src, size_t len, uint64_t off) {
    const char* p = (const char*)src;
    while (len) {
        const ssize_t k = pwrite(fd, p, len, (off_t)off);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; off += (uint64_t)k; len -= (size_t)k;
    }
    return true;
}

// get(i, row, qual, &len, &bc): fills read i (row: packed MSB-first words, qual: raw phred).  Reads are encoded by `threads` workers in
// chunks; the chunks' bytes are laid out by a prefix over their sizes, so the files are written in parallel too.
int write_triple(const char* head, uint64_t n, uint32_t row_words, uint32_t max_len, uint32_t threads, uint64_t adversarial,
                 const std::function<void(uint64_t, uint32_t*, uint8_t*, uint32_t*, int32_t*)>& get, char* err, size_t errcap) {
    const std::string H(head);
    const int ffd = open((H + ".fastb").c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    const int qfd = open((H + ".qualp").c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (ffd < 0 || qfd < 0) { if (ffd >= 0) close(ffd); if (qfd >= 0) close(qfd); return snk_fail(SNK_E_IO, err, errcap, "snk_write_df: cannot create %s.fastb / .qualp", head); }
    const uint64_t CH = 1u << 16;
    const uint64_t n_ch = (n + CH - 1) / CH;
    struct chunk { std::vector<uint8_t> fb, qp; std::vector<uint32_t> flen; std::vector<uint32_t> fsz, qsz; };
    std::vector<uint64_t> fb_at(n_ch + 1, 0), qp_at(n_ch + 1, 0);
    std::vector<uint64_t> foffs(n + 1), qoffs(n + 1);
    std::vector<uint32_t> flens(n);
    std::vector<int32_t> bcs(n);
    std::atomic<int> bad{0};
    if (!threads) threads = std::max(1u, std::min(64u, snk_host_cpu_budget()));
    // pass 1: encode every chunk into memory (sizes), pass 2: write at the prefix offsets.  Memory: the files' size; fine for tests and the bench.
    std::vector<chunk> chunks(n_ch);
    {
        std::atomic<uint64_t> next{0};
        auto work = [&]() {
            std::vector<uint32_t> row(row_words ? row_words : 1);
            std::vector<uint8_t> ql(max_len ? max_len : 1);
            for (;;) {
                const uint64_t c = next.fetch_add(1);
                if (c >= n_ch) return;
                chunk& k = chunks[c];
                const uint64_t a = c * CH, b = std::min(n, a + CH);
                k.fsz.resize(b - a); k.qsz.resize(b - a);
                k.fb.reserve((b - a) * ((max_len + 3) / 4));
                for (uint64_t i = a; i < b; ++i) {
                    uint32_t L = 0;
                    int32_t bc = 0;
                    get(i, row.data(), ql.data(), &L, &bc);
                    if (L > max_len) { bad.store(1); L = max_len; }
                    flens[i] = L; bcs[i] = bc;
                    const size_t f0 = k.fb.size();
                    for (uint32_t j = 0; j < L; j += 4) {
                        uint8_t by = 0;
                        for (uint32_t t = 0; t < 4 && j + t < L; ++t) by |= (uint8_t)(((row[(j + t) >> 4] >> (30 - 2 * ((j + t) & 15))) & 3u) << (2 * t));
                        k.fb.push_back(by);
                    }
                    k.fsz[i - a] = (uint32_t)(k.fb.size() - f0);
                    const size_t q0 = k.qp.size();
                    if (!pq_encode(ql.data(), L, k.qp, adversarial == 1 ? 1 : (adversarial ? snk_mix64(adversarial + i) | 2 : 0))) bad.store(2);
                    k.qsz[i - a] = (uint32_t)(k.qp.size() - q0);
                }
            }
        };
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < threads; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    if (bad.load()) { close(ffd); close(qfd); return snk_fail(SNK_E_ARG, err, errcap, "snk_write_df: %s", bad.load() == 1 ? "a read is longer than max_len" : "a quality value does not fit a PQVec block (> 190)"); }
    for (uint64_t c = 0; c < n_ch; ++c) { fb_at[c + 1] = fb_at[c] + chunks[c].fb.size(); qp_at[c + 1] = qp_at[c] + chunks[c].qp.size(); }
    {
        std::atomic<uint64_t> next{0};
        std::atomic<int> io_bad{0};
        auto work = [&]() {
            for (;;) {
                const uint64_t c = next.fetch_add(1);
                if (c >= n_ch) return;
                chunk& k = chunks[c];
                const uint64_t a = c * CH;
                uint64_t fo = 24 + fb_at[c], qo = 24 + qp_at[c];
                for (size_t i = 0; i < k.fsz.size(); ++i) { foffs[a + i] = fo; fo += k.fsz[i]; qoffs[a + i] = qo; qo += k.qsz[i]; }
                if (!write_all(ffd, k.fb.data(), k.fb.size(), 24 + fb_at[c]) || !write_all(qfd, k.qp.data(), k.qp.size(), 24 + qp_at[c])) io_bad.store(1);
                std::vector<uint8_t>().swap(k.fb);
                std::vector<uint8_t>().swap(k.qp);
            }
        };
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < threads; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
        if (io_bad.load()) { close(ffd); close(qfd); return snk_fail(SNK_E_IO, err, errcap, "snk_write_df: write error on %s.*", head); }
    }
    const uint64_t fvar = 24 + fb_at[n_ch], qvar = 24 + qp_at[n_ch];
    foffs[n] = fvar; qoffs[n] = qvar;
    fcb_t fh = {(uint32_t)n, 1, 4, 16, 1, fvar, fvar + (n + 1) * 8};
    fcb_t qh = {(uint32_t)n, 1, 0, 8, 1, qvar, qvar + (n + 1) * 8};
    bool ok = write_all(ffd, &fh, 24, 0) && write_all(ffd, foffs.data(), (n + 1) * 8, fvar) && write_all(ffd, flens.data(), n * 4, fh.fixed_offset);
    ok = ok && write_all(qfd, &qh, 24, 0) && write_all(qfd, qoffs.data(), (n + 1) * 8, qvar);
    ok = (close(ffd) == 0) & ok;
    ok = (close(qfd) == 0) & ok;
    if (!ok) return snk_fail(SNK_E_IO, err, errcap, "snk_write_df: write error on %s.*", head);
    // reads.bci: read range of every barcode ordinal (10X/ParseBarcodedFastqs.cc:284-293): the reads must be ordered by barcode
    int32_t maxbc = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (bcs[i] < 0 || (i && bcs[i] < bcs[i - 1])) return snk_fail(SNK_E_ARG, err, errcap, "snk_write_df: reads must be ordered by barcode id (>= 0) for the .bci index (read %llu)", (unsigned long long)i);
        maxbc = std::max(maxbc, bcs[i]);
    }
    std::vector<long long> bci((size_t)maxbc + 2, 0);
    for (uint64_t i = 0; i < n; ++i) bci[(size_t)bcs[i] + 1]++;
    for (size_t b = 1; b < bci.size(); ++b) bci[b] += bci[b - 1];
    FILE* bf = fopen((H + ".bci").c_str(), "wb");
    const uint64_t m = bci.size();
    if (!bf || fwrite("BINWRITE", 1, 8, bf) != 8 || fwrite(&m, 8, 1, bf) != 1 || fwrite(bci.data(), 8, m, bf) != m || fclose(bf) != 0)
        return snk_fail(SNK_E_IO, err, errcap, "snk_write_df: cannot write %s.bci", head);
    return SNK_OK;
}

}  // namespace

extern "C" int snk_write_df(const char* head, uint64_t n, const uint32_t* rows, uint32_t row_words, const uint16_t* lens, uint32_t read_len, const uint8_t* quals,
                            uint32_t qstride, const int32_t* bc, uint32_t threads, uint64_t adversarial, char* err, size_t errcap) {
    if (!head || (n && (!rows || !quals))) return snk_fail(SNK_E_ARG, err, errcap, "snk_write_df: NULL argument");
    if (read_len == 0 || read_len > row_words * 16 || read_len > qstride) return snk_fail(SNK_E_ARG, err, errcap, "snk_write_df: read_len does not fit the rows");
    return write_triple(head, n, row_words, read_len, threads, adversarial,
                        [&](uint64_t i, uint32_t* row, uint8_t* q, uint32_t* L, int32_t* b) {
                            *L = lens ? lens[i] : read_len;
                            memcpy(row, rows + i * row_words, row_words * 4);
                            memcpy(q, quals + i * (uint64_t)qstride, std::min<uint32_t>(*L, read_len));
                            *b = bc ? bc[i] : 0;
                        },
                        err, errcap);
}

// reads [first, first + n) of the synthetic model as a triple.  The model's unbarcoded pairs are scattered; a .bci index needs the reads
// ordered by barcode, so sp->unbarcoded_ppm must be 0 (barcode ids then rise with the read index).  qual_jitter: a quality of 30 becomes
// 30 + hash % jitter -- values the trim never looks at differently (all >= min_qual), but a quality file of realistic entropy.
extern "C" int snk_synth_df_write(const char* head, const snk_synth_params* sp, uint64_t first, uint64_t n, uint32_t qual_jitter, uint32_t threads, char* err,
                                  size_t errcap) {
    if (!head || !sp) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_df_write: NULL argument");
    if (sp->unbarcoded_ppm) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_df_write: unbarcoded_ppm must be 0 (the .bci index needs reads ordered by barcode)");
    const uint32_t L = sp->read_len, rw = (L + 15) / 16;
    if (L == 0 || L > 256) return snk_fail(SNK_E_ARG, err, errcap, "snk_synth_df_write: bad read length");
    const snk_synth_params P = *sp;
    return write_triple(head, n, rw, L, threads, 0,
                        [=](uint64_t i, uint32_t* row, uint8_t* q, uint32_t* len, int32_t* b) {
                            snk_synth_read(P, first + i, row, rw, q, b);
                            *len = L;
                            if (qual_jitter > 1) {
                                for (uint32_t j = 0; j < L; j += 16) {
                                    uint64_t h = snk_rng(P.seed ^ 0x51A17ull, 77, (first + i) * 16 + (j >> 4));
                                    for (uint32_t t = 0; t < 16 && j + t < L; ++t, h >>= 4) if (q[j + t] == 30) q[j + t] = (uint8_t)(30 + (h & 15) % qual_jitter);
                                }
                            }
                        },
                        err, errcap);
}
