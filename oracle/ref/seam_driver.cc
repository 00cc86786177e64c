// oracle/ref/seam_driver.cc -- TEST INFRASTRUCTURE ONLY: the binding of INTEGRATION.md section 1 as CODE.
//
// buildReadQGraph48_snk() below is the function a maintainer of the reference would put into
// lib/assembly/src/10X/runstages/RunStages.cc in place of the buildReadQGraph48(...) call of StageBuildGraph (:404-413): it takes the
// reference's own types (vecbvec reads, VecPQVec quals, vec<int32_t> barcodes, BuildReadQGraph48.h:24-40), hands them to libsnk
// through the C ABI of include/snk.h (device arrays in, unitigs out) and finishes with the reference's own buildHBVFromEdges
// (paths/long/HBVFromEdges.cc:244-296) -- so what comes back is a HyperBasevector built by reference code from GPU unitigs.
// It is compiled against the reference's headers where they lie (oracle/ref/build_ref.sh, same overlay and flags as snref_driver),
// linked with libref.a + libsnk.so + the HIP runtime, and run on the GPU box by tests/test_gpu_seam.py, which compares its dump
// (unitigs in BVComp order, vertices / edges / fwd-rev translation of the graph) with the reference's own run on the same input.
//
// usage: snref_seam <in.snkrd> <outdir> [minQual=7 minFreq=3 minBC=2]
#include "Basevector.h"
#include "feudal/ObjectManager.h"
#include "feudal/PQVec.h"
#include "paths/HyperBasevector.h"
#include "paths/long/HBVFromEdges.h"
#include "system/RunTime.h"
#include "system/System.h"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/snk.h"

namespace {

void die(const char* what, const char* detail = "") { fprintf(stderr, "snref_seam: %s %s\n", what, detail); exit(2); }
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) die(#x, hipGetErrorString(e_)); } while (0)

// ---- the stub -----------------------------------------------------------------------------------
void buildReadQGraph48_snk(vecbvec const& reads, VecPQVec const& quals, vec<int32_t> const* bcp, int64_t ignBcBelow, unsigned minQual, unsigned minFreq,
                           unsigned minBC, HyperBasevector* pHBV, vec<int>* fwd, vec<int>* rev, vecbvec* pEdges) {
    char err[512] = "";
    snk_ctx* ctx = nullptr;
    if (snk_ctx_create(0, &ctx, err, sizeof err)) die("no usable MI355X (there is no CPU path):", err);
    const size_t n = reads.size();
    unsigned L = 0;
    for (size_t r = 0; r < n; ++r) if (reads[r].size() > L) L = reads[r].size();
    if (L == 0) L = 1;
    const uint32_t rw = (L + 15) / 16, qs = (L + 3) / 4 * 4;
    // 1. rows MSB-first 2-bit (KMer.h:153-160 word order), quality rows raw phred, lengths, barcodes -> HBM
    std::vector<uint32_t> rows(n * rw, 0);
    std::vector<uint8_t> q8(n * (size_t)qs, 0);
    std::vector<uint16_t> lens(n);
    qvec qv;
    for (size_t r = 0; r < n; ++r) {
        bvec const& b = reads[r];
        lens[r] = (uint16_t)b.size();
        for (unsigned i = 0; i < b.size(); ++i) rows[r * rw + (i >> 4)] |= (uint32_t)b[i] << (30 - 2 * (i & 15));
        quals[r].unpack(&qv);
        for (unsigned i = 0; i < qv.size(); ++i) q8[r * (size_t)qs + i] = qv[i];
    }
    void *d_rows, *d_quals, *d_lens, *d_bc = nullptr;
    HIP_OK(hipMalloc(&d_rows, rows.size() * 4 + 64));
    HIP_OK(hipMalloc(&d_quals, q8.size() + 64));
    HIP_OK(hipMalloc(&d_lens, n * 2 + 64));
    HIP_OK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_quals, q8.data(), q8.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_lens, lens.data(), n * 2, hipMemcpyHostToDevice));
    if (bcp) {
        HIP_OK(hipMalloc(&d_bc, n * 4 + 64));
        HIP_OK(hipMemcpy(d_bc, &(*bcp)[0], n * 4, hipMemcpyHostToDevice));
    }
    snk_dev_reads in;
    memset(&in, 0, sizeof in);
    in.n_reads = n; in.rows = d_rows; in.row_words = rw; in.read_len = L; in.lens = d_lens; in.quals = d_quals; in.qstride = qs; in.bc = d_bc;
    in.ign_bc_below = ignBcBelow;
    snk_params p;
    snk_params_default(&p);                      // K=48 MIN_QUAL=7 MIN_FREQ=3 MIN_BC=2 (DF.cc:138-141)
    p.min_qual = minQual; p.min_freq = minFreq; p.min_bc = minBC;
    snk_dev_result r;
    if (snk_dev_count_graph(ctx, &in, &p, &r, nullptr, err, sizeof err)) die("snk_dev_count_graph:", err);
    // 2. unitigs back as vecbvec (canonical orientation == EdgeBuilder::addEdge, BuildReadQGraph48.cc:478-506)
    std::vector<uint64_t> off(r.n_unitigs + 1);
    std::vector<uint8_t> bases(r.unitig_total_bases + 1);
    if (snk_dev_download(ctx, r.unitig_off, off.data(), off.size() * 8, nullptr)) die("download");
    if (r.unitig_total_bases && snk_dev_download(ctx, r.unitig_bases, bases.data(), r.unitig_total_bases, nullptr)) die("download");
    vecbvec edges;
    edges.reserve(r.n_unitigs);
    for (size_t u = 0; u < r.n_unitigs; ++u) {
        bvec b;
        b.resize(off[u + 1] - off[u]);
        for (size_t i = 0; i < b.size(); ++i) b.Set(i, bases[off[u] + i]);
        edges.push_back(b);
    }
    // 3. the unchanged reference tail
    buildHBVFromEdges(edges, 48, pHBV, fwd, rev);
    if (pEdges) *pEdges = edges;
    snk_ctx_destroy(ctx);
    (void)hipFree(d_rows); (void)hipFree(d_quals); (void)hipFree(d_lens); (void)hipFree(d_bc);
}

// ---- input / dump (the formats of ref_driver.cc) ---------------------------------------------------------------------------------
struct Input {
    uint64_t n = 0;
    uint32_t stride = 0, has_bc = 0;
    int64_t ign_bc_below = 0;
    std::vector<uint16_t> len;
    std::vector<uint8_t> bases, quals;
    std::vector<int32_t> bc;
};
void rd(FILE* f, void* p, size_t n) { if (n && fread(p, 1, n, f) != n) die("short read"); }
Input load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) die("cannot open input");
    char magic[8];
    rd(f, magic, 8);
    if (memcmp(magic, "SNKRD001", 8)) die("bad magic");
    Input in;
    rd(f, &in.n, 8); rd(f, &in.stride, 4); rd(f, &in.has_bc, 4); rd(f, &in.ign_bc_below, 8);
    in.len.resize(in.n); rd(f, in.len.data(), in.n * 2);
    in.bases.resize(in.n * in.stride); rd(f, in.bases.data(), in.bases.size());
    in.quals.resize(in.n * in.stride); rd(f, in.quals.data(), in.quals.size());
    if (in.has_bc) { in.bc.resize(in.n); rd(f, in.bc.data(), in.n * 4); }
    fclose(f);
    return in;
}
std::string bvstr(bvec const& b) {
    std::string s(b.size(), 'A');
    for (unsigned i = 0; i < b.size(); ++i) s[i] = "ACGT"[b[i]];
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    RunTime();
    if (argc < 3) die("usage: snref_seam <in.snkrd> <outdir> [minQual minFreq minBC]");
    const std::string outdir = argv[2];
    const unsigned minQual = argc > 3 ? atoi(argv[3]) : 7, minFreq = argc > 4 ? atoi(argv[4]) : 3, minBC = argc > 5 ? atoi(argv[5]) : 2;
    Input in = load(argv[1]);
    Mkpath(String(outdir.c_str()));
    vecbvec reads;
    VecPQVec pq;
    reads.reserve(in.n);
    pq.reserve(in.n);
    {
        bvec b;
        qvec q;
        for (uint64_t r = 0; r < in.n; ++r) {
            const unsigned L = in.len[r];
            b.resize(L);
            q.resize(L);
            for (unsigned i = 0; i < L; ++i) {
                unsigned code;
                switch (in.bases[r * in.stride + i]) {
                    case 'C': code = 1; break;
                    case 'G': code = 2; break;
                    case 'T': code = 3; break;
                    default: code = 0;
                }
                b.Set(i, code);
                q[i] = in.quals[r * in.stride + i];
            }
            reads.push_back(b);
            pq.push_back(PQVec(q));
        }
    }
    vec<int32_t> bc;
    if (in.has_bc) bc.assign(in.bc.begin(), in.bc.end());
    HyperBasevector hbv0;
    vec<int> f0, r0;
    vecbvec edges;
    buildReadQGraph48_snk(reads, pq, in.has_bc ? &bc : nullptr, in.ign_bc_below, minQual, minFreq, minBC, &hbv0, &f0, &r0, &edges);
    // the dump of ref_driver.cc: unitigs in BVComp order, the graph built from them in that order
    std::vector<size_t> order(edges.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        bvec const& x = edges[a]; bvec const& y = edges[b];
        if (x.size() != y.size()) return x.size() > y.size();
        return x < y;
    });
    vecbvec sorted;
    sorted.reserve(edges.size());
    for (size_t i : order) sorted.push_back(edges[i]);
    {
        FILE* f = fopen((outdir + "/unitigs.txt").c_str(), "w");
        for (size_t i = 0; i < sorted.size(); ++i) fprintf(f, "%s\n", bvstr(sorted[i]).c_str());
        fclose(f);
    }
    HyperBasevector hbv;
    vec<int> fwd, rev;
    buildHBVFromEdges(sorted, 48, &hbv, &fwd, &rev);
    {
        vec<int> to_left, to_right;
        hbv.ToLeft(to_left);
        hbv.ToRight(to_right);
        FILE* f = fopen((outdir + "/hbv.txt").c_str(), "w");
        fprintf(f, "N %d E %d U %lu\n", hbv.N(), hbv.EdgeObjectCount(), (unsigned long)sorted.size());
        for (int e = 0; e < hbv.EdgeObjectCount(); ++e) fprintf(f, "E %d %d %d %s\n", e, to_left[e], to_right[e], bvstr(hbv.EdgeObject(e)).c_str());
        for (size_t u = 0; u < sorted.size(); ++u) fprintf(f, "X %lu %d %d\n", (unsigned long)u, fwd[u], rev[u]);
        fclose(f);
    }
    printf("SNREF_SEAM reads=%lu unitigs=%lu hbv_edges=%d hbv_vertices=%d (graph of the stub's own call: %d edges)\n", (unsigned long)in.n,
           (unsigned long)sorted.size(), hbv.EdgeObjectCount(), hbv.N(), hbv0.EdgeObjectCount());
    return 0;
}
