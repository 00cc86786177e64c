// snk_hbv.hip -- a14: the graph-from-unitigs step, buildHBVFromEdges (lib/assembly/src/paths/long/HBVFromEdges.cc:244-296).
//
//   VertexDictBuilder::map (:136-145) emits 4 edge ends per unitig (2 for a palindrome); a vertex is one distinct
//   (K-1)-mer and lists its incident (edge, rc) pairs in EEComp order (:113-122); HBVBuilder (:170-238) floods from
//   every edge in BVComp order (:106-111: length descending, then lexicographic), forward copies first, and hands out
//   vertex and edge ids in visiting order.
//
// Two entry points produce the same snk_hbv:
//   snk_hbv_from_unitigs : host arrays that are already in BVComp order (the .bv file, snk_count_graph's output)
//   snk_dev_hbv          : unitigs resident in HBM, in any order (the output of snk_dev_count_graph).
//        The data-parallel part -- what the reference runs through its MapReduce engine -- is done on the device:
//        BVComp rank of every unitig (radix sort by first k-mer -- two unitigs never share it, so it decides the
//        lexicographic comparison -- then a stable one by length), the palindrome test, the 4U (K-1)-mer end keys
//        (128-bit, MSB first), their stable radix sort (EEComp == generation order), vertex classes by a flag scan.
//        The id hand-out is a breadth-first flood whose ids are the visiting order -- sequential inside a connected
//        component, independent between components: components by union-find on the device, one thread floods one
//        component into its own id block, and only components above SNK_HBV_BIG nodes (the connected bulk of a genome
//        graph) are flooded on the host.  Graphs below SNK_HBV_DEV_MIN (65536) unitigs take the host flood directly (the
//        hot path's few thousand: faster than the launches).  Every loop of the device flood is bounded; one that runs out
//        hands the graph to the host flood.
#include <stdlib.h>
#include <sys/mman.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"

namespace {

// ee[]: edge ends in vertex-major order, code = rank*4 + rc*2 + distal; run_beg[v..v+1] delimits vertex class v;
// vtx_of[code] = class (or -1).
struct hbv_tables {
    uint64_t U;
    const uint8_t* pal;
    const uint32_t* ee;
    const int32_t* vtx_of;
    const uint64_t* run_beg;
    bool short_q = true;      // option hbv_short_queue, read by the caller (the floods run on host threads of their own)
};
// The host flood walks a few hundred MB of tables at random: with 4-KB pages every access is a TLB miss on top of the cache miss.  Large
// host arrays are asked for on 2-MB boundaries with MADV_HUGEPAGE BEFORE they are touched (transparent huge pages are "madvise" on these
// hosts); free() releases them.  Best effort: without huge pages the arrays are what they were.
bool hbv_huge_pages = true;          // option hbv_huge_pages (a measurement switch: set by the entry points, read here)
void* huge_alloc(size_t bytes) {
    if (!hbv_huge_pages || bytes < ((size_t)4 << 20)) return malloc(bytes ? bytes : 1);
    void* p = nullptr;
    const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, len) != 0) return malloc(bytes);
    (void)madvise(p, len, MADV_HUGEPAGE);
    return p;
}
template <typename T>
struct huge_vec {                      // the little of std::vector the tables need; never value-initialised
    T* p = nullptr;
    size_t n = 0;
    huge_vec() = default;
    huge_vec(const huge_vec&) = delete;
    huge_vec& operator=(const huge_vec&) = delete;
    ~huge_vec() { free(p); }
    bool resize(size_t m) { free(p); p = (T*)huge_alloc((m ? m : 1) * sizeof(T)); n = p ? m : 0; return p != nullptr; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

int hbv_alloc_out(uint64_t U, uint64_t nruns, snk_hbv* out, char* err, size_t errcap) {
    out->n_vertices = (int32_t)nruns;
    out->fwd_xlat = (int32_t*)huge_alloc(U * 4);
    out->rev_xlat = (int32_t*)huge_alloc(U * 4);
    out->v_left = (int32_t*)huge_alloc(2 * U * 4);
    out->v_right = (int32_t*)huge_alloc(2 * U * 4);
    out->src_unitig = (int32_t*)huge_alloc(2 * U * 4);
    out->is_rc = (uint8_t*)huge_alloc(2 * U);
    if (!out->fwd_xlat || !out->rev_xlat || !out->v_left || !out->v_right || !out->src_unitig || !out->is_rc) {
        snk_hbv_free(out);
        return snk_fail(SNK_E_NOMEM, err, errcap, "snk_hbv: host allocation failed");
    }
    return SNK_OK;
}
// the flood of one component from its seed (HBVBuilder::processQueue); ids continue from next_e / next_v
void hbv_flood_component(const hbv_tables& t, uint64_t e0, int rc0, int32_t* vid, std::vector<uint64_t>& q, int32_t& next_e,
                         int32_t& next_v, snk_hbv* out) {
    auto done = [&](uint64_t e, int rc) { return (rc ? out->rev_xlat : out->fwd_xlat)[e] != -1; };
    q.clear();            // FIFO: [head, size); emptied whenever a component is finished
    size_t head = 0;
    q.push_back(e0 * 2 + rc0);
    // measurement aid (SNK_HBV_DEPTH=1): the levels of this breadth-first flood -- what a level-synchronous reproduction on the device
    // would take one round (a launch, or a grid-wide barrier: >= 3 us either way) for each
    static const bool want_depth = getenv("SNK_HBV_DEPTH") != nullptr;
    const bool short_q = t.short_q;
    std::vector<uint32_t> lvl;
    uint64_t visited = 0;
    uint32_t depth = 0;
    if (want_depth) lvl.push_back(0);
    while (head < q.size()) {
        // The flood is a chain of cache misses (4U + U + V words touched at random: 250 ns per edge on a 6 M-unitig graph); the queue
        // says what will be touched next, so the lines of the entries 16, 8 and 4 places ahead are asked for now, one level of
        // indirection each (classes of an entry; runs and ids of its classes; the ends behind the runs).
        // The bulk of a genome graph is flooded through a SHORT queue (4.7 edge copies per breadth-first level, profiles/r05_hbv_depth.log): most
        // of the time there is nothing 16, 8 or 4 places ahead.  An entry's class words are asked for when it is pushed, and with fewer entries
        // behind the head the same stages run one and two places ahead.
        auto stage_b = [&](uint64_t y) {
            const uint64_t ye = y >> 1, yrc = (t.pal[ye] && (y & 1)) ? 0 : (y & 1);
            const int32_t a = t.vtx_of[ye * 4 + yrc * 2], b = t.vtx_of[ye * 4 + yrc * 2 + 1];
            __builtin_prefetch(&t.run_beg[a]); __builtin_prefetch(&t.run_beg[b]);
            __builtin_prefetch(&vid[a]); __builtin_prefetch(&vid[b]);
            __builtin_prefetch(&(y & 1 ? out->rev_xlat : out->fwd_xlat)[ye]);
        };
        auto stage_c = [&](uint64_t y) {
            const uint64_t ye = y >> 1, yrc = (t.pal[ye] && (y & 1)) ? 0 : (y & 1);
            __builtin_prefetch(&t.ee[t.run_beg[t.vtx_of[ye * 4 + yrc * 2]]]);
            __builtin_prefetch(&t.ee[t.run_beg[t.vtx_of[ye * 4 + yrc * 2 + 1]]]);
        };
        if (head + 16 < q.size()) __builtin_prefetch(&t.vtx_of[(q[head + 16] >> 1) * 4 + (q[head + 16] & 1) * 2]);
        if (head + 8 < q.size()) stage_b(q[head + 8]);
        else if (short_q && head + 2 < q.size()) stage_b(q[head + 2]);
        if (head + 4 < q.size()) stage_c(q[head + 4]);
        else if (short_q && head + 1 < q.size()) stage_c(q[head + 1]);
        const uint64_t x = q[head++];
        const uint64_t e = x >> 1;
        const int rc = (int)(x & 1);
        if (done(e, rc)) continue;
        int32_t r1 = t.vtx_of[e * 4 + rc * 2 + 0], r2 = t.vtx_of[e * 4 + rc * 2 + 1];
        if (t.pal[e] && rc) { r1 = t.vtx_of[e * 4 + 0]; r2 = t.vtx_of[e * 4 + 1]; }
        if (vid[r1] == -1) vid[r1] = next_v++;
        if (vid[r2] == -1) vid[r2] = next_v++;
        const int32_t id = next_e++;
        if (want_depth) { ++visited; if (lvl[head - 1] > depth) depth = lvl[head - 1]; }
        out->v_left[id] = vid[r1]; out->v_right[id] = vid[r2];
        out->src_unitig[id] = (int32_t)e; out->is_rc[id] = (uint8_t)rc;
        if (!rc || t.pal[e]) out->fwd_xlat[e] = id;
        if (rc || t.pal[e]) out->rev_xlat[e] = id;
        for (int side = 0; side < 2; ++side) {
            const int32_t r = side ? r2 : r1;
            for (uint64_t j = t.run_beg[r]; j < t.run_beg[r + 1]; ++j) {
                const uint64_t ed = t.ee[j] >> 2;
                const int erc = (int)((t.ee[j] >> 1) & 1u);
                if (!done(ed, erc)) {
                    q.push_back(ed * 2 + erc);
                    if (short_q) __builtin_prefetch(&t.vtx_of[ed * 4 + ((t.pal[ed] && erc) ? 0 : erc) * 2]);
                    if (want_depth) lvl.push_back(lvl[head - 1] + 1);
                }
            }
        }
    }
    if (want_depth && visited >= 100000)
        fprintf(stderr, "[snk hbv] component of %llu edge copies: %u breadth-first levels (%.1f copies per level)\n", (unsigned long long)visited, depth + 1, (double)visited / (depth + 1));
}
// Fills every array of `out`: every component in seed order.
int hbv_flood(uint64_t U, const uint8_t* pal, const uint32_t* ee, uint64_t n_ee, const int32_t* vtx_of, const uint64_t* run_beg,
              uint64_t nruns, snk_hbv* out, char* err, size_t errcap) {
    int rc = hbv_alloc_out(U, nruns, out, err, errcap);
    if (rc) return rc;
    for (uint64_t i = 0; i < U; ++i) out->fwd_xlat[i] = out->rev_xlat[i] = -1;
    huge_vec<int32_t> vidv;
    if (!vidv.resize(nruns + 1)) { snk_hbv_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_hbv: host allocation failed"); }
    int32_t* vid = vidv.data();
    for (uint64_t i = 0; i < nruns; ++i) vid[i] = -1;
    int32_t next_v = 0, next_e = 0;
    std::vector<uint64_t> q;
    const hbv_tables t{U, pal, ee, vtx_of, run_beg, snk_opt_u32("hbv_short_queue", 1) != 0};
    (void)n_ee;
    for (int pass = 0; pass < 2; ++pass)
        for (uint64_t e0 = 0; e0 < U; ++e0)
            if ((pass ? out->rev_xlat : out->fwd_xlat)[e0] == -1) hbv_flood_component(t, e0, pass, vid, q, next_e, next_v, out);
    out->n_edges = next_e;
    return SNK_OK;
}

}  // namespace

extern "C" int snk_hbv_from_unitigs(uint32_t K, uint64_t U, const uint64_t* off, const uint8_t* bases, snk_hbv* out, char* err, size_t errcap) {
    if (!out) return snk_fail(SNK_E_ARG, err, errcap, "snk_hbv_from_unitigs: NULL argument");
    memset(out, 0, sizeof *out);
    if (U == 0) return SNK_OK;
    if (U >= (1ull << 30)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_hbv_from_unitigs: too many unitigs");
    const uint32_t kl = K - 1;
    auto base_of = [&](uint32_t code, uint32_t j) -> uint8_t {
        const uint64_t e = code >> 2;
        const uint8_t* b = bases + off[e];
        const uint64_t len = off[e + 1] - off[e];
        const uint64_t p = (code & 1u) ? len - kl + j : j;
        return (code & 2u) ? (uint8_t)(b[len - 1 - p] ^ 3) : b[p];
    };
    auto seq_cmp = [&](uint32_t a, uint32_t b) -> int {
        for (uint32_t j = 0; j < kl; ++j) { uint8_t x = base_of(a, j), y = base_of(b, j); if (x != y) return x < y ? -1 : 1; }
        return 0;
    };
    std::vector<uint8_t> pal(U);
    std::vector<uint32_t> ee;
    ee.reserve(4 * U);
    for (uint64_t e = 0; e < U; ++e) {
        const uint8_t* b = bases + off[e];
        const uint64_t len = off[e + 1] - off[e];
        if (len < K) return snk_fail(SNK_E_ARG, err, errcap, "snk_hbv_from_unitigs: unitig %llu shorter than K", (unsigned long long)e);
        bool p = (len & 1) == 0;                       // getCanonicalForm == PALINDROME (dna/CanonicalForm.h:35-48)
        for (uint64_t i = 0, j = len; p && i < j; ++i) { --j; if (b[i] != (uint8_t)(b[j] ^ 3)) p = false; }
        pal[e] = p;
        ee.push_back((uint32_t)e * 4 + 0);
        ee.push_back((uint32_t)e * 4 + 1);
        if (!p) { ee.push_back((uint32_t)e * 4 + 2); ee.push_back((uint32_t)e * 4 + 3); }
    }
    // EEComp: BVComp of the containers (== index order, the input is sorted), then rc, then position == code order
    std::sort(ee.begin(), ee.end(), [&](uint32_t a, uint32_t b) {
        int c = seq_cmp(a, b);
        return c ? c < 0 : a < b;
    });
    std::vector<int32_t> vtx_of(4 * U, -1);
    std::vector<uint64_t> run_beg;
    for (uint64_t i = 0; i < ee.size();) {
        uint64_t j = i + 1;
        while (j < ee.size() && seq_cmp(ee[i], ee[j]) == 0) ++j;
        for (uint64_t q = i; q < j; ++q) vtx_of[ee[q]] = (int32_t)run_beg.size();
        run_beg.push_back(i);
        i = j;
    }
    const uint64_t nruns = run_beg.size();
    run_beg.push_back(ee.size());
    return hbv_flood(U, pal.data(), ee.data(), ee.size(), vtx_of.data(), run_beg.data(), nruns, out, err, errcap);
}

// ---- f2: the involution and the files DF keeps the graph in.
// hbv.Involution(inv) (paths/HyperBasevector.cc:685-697, called at 10X/runstages/RunStages.cc:418): inv[e] = the edge whose
// sequence is the reverse complement of edge e's.  The reference finds the pairs by sorting the edges and their reverse
// complements; here they are known by construction: the two copies of a unitig (a palindrome is its own partner).
extern "C" int snk_hbv_involution(const snk_hbv* h, uint64_t n_unitigs, int32_t* inv, char* err, size_t errcap) {
    if (!h || (!inv && h->n_edges)) return snk_fail(SNK_E_ARG, err, errcap, "snk_hbv_involution: NULL argument");
    for (int32_t e = 0; e < h->n_edges; ++e) inv[e] = -1;
    for (uint64_t u = 0; u < n_unitigs; ++u) {
        const int32_t f = h->fwd_xlat[u], r = h->rev_xlat[u];
        if (f < 0 || r < 0 || f >= h->n_edges || r >= h->n_edges) return snk_fail(SNK_E_ARG, err, errcap, "snk_hbv_involution: inconsistent translation tables");
        inv[f] = r;
        inv[r] = f;
    }
    for (int32_t e = 0; e < h->n_edges; ++e) if (inv[e] < 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_hbv_involution: edge %d has no partner", e);
    return SNK_OK;
}

// a.hbv = BinaryWriter::writeFile(HyperBasevector): "BINWRITE", int K, from_ (vec<vec<int>>: u64 n, per vertex u64 m + m ints),
// from_edge_obj_, to_edge_obj_ (same shape), edges_ (u64 E, per edge u32 bases + ceil(bases/4) bytes, base j at bits 2(j%4))
// (paths/HyperBasevector.cc:121-125, graph/Digraph.h:381-382, graph/DigraphTemplate.h:3092-3097, feudal/BinaryStream.h:488-493).
// Adjacency order is AddEdge's (DigraphTemplate.h:2572-2582): a vertex's out-edges ascending by target vertex, equal targets
// in edge-id order (upper_bound insertion); in-edges likewise by source vertex.  a.inv = "BINWRITE", u64 E, E ints.
static int write_hbv_impl(const char* path_hbv, const char* path_inv, uint32_t K, uint64_t U, const uint64_t* off, const uint8_t* bases,
                          const snk_hbv* h, char* err, size_t errcap) {
    const int32_t N = h->n_vertices, E = h->n_edges;
    std::vector<std::vector<std::pair<int32_t, int32_t>>> from(N), to(N);      // (other vertex, edge)
    for (int32_t e = 0; e < E; ++e) {
        const int32_t v = h->v_left[e], w = h->v_right[e];
        if (v < 0 || w < 0 || v >= N || w >= N || h->src_unitig[e] < 0 || (uint64_t)h->src_unitig[e] >= U)
            return snk_fail(SNK_E_ARG, err, errcap, "snk_write_hbv: edge %d is out of range", e);
        from[v].push_back({w, e});
        to[w].push_back({v, e});
    }
    for (auto& l : from) std::stable_sort(l.begin(), l.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) { return a.first < b.first; });
    for (auto& l : to) std::stable_sort(l.begin(), l.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) { return a.first < b.first; });
    FILE* f = fopen(path_hbv, "wb");
    if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_write_hbv: cannot open %s", path_hbv);
    bool ok = fwrite("BINWRITE", 1, 8, f) == 8;
    const int32_t k32 = (int32_t)K;
    ok = ok && fwrite(&k32, 4, 1, f) == 1;
    auto put_lists = [&](const std::vector<std::vector<std::pair<int32_t, int32_t>>>& ls, bool second) {
        const uint64_t n = ls.size();
        ok = ok && fwrite(&n, 8, 1, f) == 1;
        std::vector<int32_t> tmp;
        for (const auto& l : ls) {
            const uint64_t m = l.size();
            tmp.resize(m);
            for (uint64_t i = 0; i < m; ++i) tmp[i] = second ? l[i].second : l[i].first;
            ok = ok && fwrite(&m, 8, 1, f) == 1 && (m == 0 || fwrite(tmp.data(), 4, m, f) == m);
        }
    };
    put_lists(from, false);
    put_lists(from, true);
    put_lists(to, true);
    const uint64_t e64 = (uint64_t)E;
    ok = ok && fwrite(&e64, 8, 1, f) == 1;
    std::vector<uint8_t> buf;
    for (int32_t e = 0; ok && e < E; ++e) {
        const uint64_t u = (uint64_t)h->src_unitig[e], len64 = off[u + 1] - off[u];
        if (len64 > 0xFFFFFFFFull) { fclose(f); return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_write_hbv: edge longer than 2^32 bases"); }
        const uint32_t len = (uint32_t)len64;
        const uint8_t* b = bases + off[u];
        buf.assign((len + 3) / 4, 0);
        if (!h->is_rc[e]) for (uint32_t j = 0; j < len; ++j) buf[j >> 2] |= (uint8_t)((b[j] & 3u) << (2 * (j & 3)));
        else for (uint32_t j = 0; j < len; ++j) buf[j >> 2] |= (uint8_t)(((b[len - 1 - j] & 3u) ^ 3u) << (2 * (j & 3)));
        ok = fwrite(&len, 4, 1, f) == 1 && (buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size());
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) return snk_fail(SNK_E_IO, err, errcap, "snk_write_hbv: short write to %s", path_hbv);
    if (path_inv) {
        std::vector<int32_t> inv((size_t)E);
        int rc = snk_hbv_involution(h, U, inv.data(), err, errcap);
        if (rc) return rc;
        f = fopen(path_inv, "wb");
        if (!f) return snk_fail(SNK_E_IO, err, errcap, "snk_write_hbv: cannot open %s", path_inv);
        ok = fwrite("BINWRITE", 1, 8, f) == 8 && fwrite(&e64, 8, 1, f) == 1 && (E == 0 || fwrite(inv.data(), 4, (size_t)E, f) == (size_t)E);
        if (fclose(f) != 0) ok = false;
        if (!ok) return snk_fail(SNK_E_IO, err, errcap, "snk_write_hbv: short write to %s", path_inv);
    }
    return SNK_OK;
}
extern "C" int snk_write_hbv(const char* path_hbv, const char* path_inv, uint32_t K, uint64_t n_unitigs, const uint64_t* unitig_off,
                             const uint8_t* unitig_bases, const snk_hbv* h, char* err, size_t errcap) {
    if (!path_hbv || !h || (n_unitigs && (!unitig_off || !unitig_bases))) return snk_fail(SNK_E_ARG, err, errcap, "snk_write_hbv: NULL argument");
    try { return write_hbv_impl(path_hbv, path_inv, K, n_unitigs, unitig_off, unitig_bases, h, err, errcap); }
    catch (const std::bad_alloc&) { return snk_fail(SNK_E_NOMEM, err, errcap, "snk_write_hbv: host allocation failed"); }
    catch (...) { return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_write_hbv: unexpected exception"); }
}

extern "C" void snk_hbv_free(snk_hbv* h) {
    if (!h) return;
    free(h->v_left); free(h->v_right); free(h->src_unitig); free(h->is_rc); free(h->fwd_xlat); free(h->rev_xlat); free(h->bvcomp_order);
    memset(h, 0, sizeof *h);
}

// ------------------------------------------------------------------------------------------------ device part
namespace {

constexpr int HB = 256;

// per unitig: its first K bases as a 128-bit key (the lexicographic part of BVComp: two unitigs never share their
// first k-mer), the palindrome flag, the "shorter than K" flag
__global__ void __launch_bounds__(HB) hbv_head_kernel(const uint64_t* __restrict__ off, const uint8_t* __restrict__ bases, uint64_t U,
                                                      uint32_t K, snk_u128* __restrict__ fkey, uint32_t* __restrict__ idx,
                                                      uint8_t* __restrict__ pal, uint32_t* __restrict__ flags /* [0] short, [1] #palindromes */) {
    const uint64_t u = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (u >= U) return;
    const uint64_t b0 = off[u], len = off[u + 1] - b0;
    idx[u] = (uint32_t)u;
    const uint8_t* b = bases + b0;
    if (len < K) { atomicOr(&flags[0], 1u); fkey[u] = 0; pal[u] = 0; return; }
    snk_u128 k = 0;
    for (uint32_t j = 0; j < K; ++j) k = (k << 2) | (snk_u128)(b[j] & 3u);
    fkey[u] = k;
    // getCanonicalForm == PALINDROME (dna/CanonicalForm.h:35-48); a mismatch shows up within a few bases
    bool p = (len & 1) == 0;
    for (uint64_t i = 0, j = len; p && i < j; ++i) { --j; if (b[i] != (uint8_t)(b[j] ^ 3)) p = false; }
    pal[u] = p ? 1 : 0;
    if (p) atomicAdd(&flags[1], 1u);
}

// second key of the ranking: length descending, taken in first-k-mer order (the sort that follows is stable)
__global__ void __launch_bounds__(HB) hbv_lenkey_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ idx1, uint64_t U,
                                                        uint64_t* __restrict__ lkey) {
    const uint64_t i = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= U) return;
    const uint32_t u = idx1[i];
    lkey[i] = ~(off[u + 1] - off[u]);
}

// one thread per (BVComp rank, end): the (K-1)-mer of that end of that strand, MSB first; ends a palindrome does not
// have sort last (all ones: a real key leaves the low 128 - 2(K-1) bits zero)
__global__ void __launch_bounds__(HB) hbv_ends_kernel(const uint64_t* __restrict__ off, const uint8_t* __restrict__ bases,
                                                      const uint32_t* __restrict__ order, const uint8_t* __restrict__ pal, uint64_t U,
                                                      uint32_t K, snk_u128* __restrict__ keys, uint32_t* __restrict__ codes,
                                                      uint8_t* __restrict__ pal_ranked) {
    const uint64_t t = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (t >= 4 * U) return;
    const uint64_t r = t >> 2;
    const uint32_t rc = (uint32_t)(t >> 1) & 1u, distal = (uint32_t)t & 1u;
    const uint32_t u = order[r];
    const uint64_t b0 = off[u], len = off[u + 1] - b0;
    const uint8_t* b = bases + b0;
    const uint32_t kl = K - 1;
    codes[t] = (uint32_t)t;
    if ((t & 3) == 0) pal_ranked[r] = pal[u];
    if (rc && pal[u]) { keys[t] = ~(snk_u128)0; return; }
    // forward strand: bases [p0, p0+kl); reverse strand: complement of the mirrored range, read backwards
    const uint64_t p0 = distal ? len - kl : 0;
    snk_u128 k = 0;
    if (!rc) for (uint32_t j = 0; j < kl; ++j) k = (k << 2) | (snk_u128)(b[p0 + j] & 3u);
    else for (uint32_t j = 0; j < kl; ++j) k = (k << 2) | (snk_u128)((b[len - 1 - (p0 + j)] ^ 3u) & 3u);
    keys[t] = k << (128 - 2 * kl);
}

__global__ void __launch_bounds__(HB) hbv_flag_kernel(const snk_u128* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// cls[i] = inclusive scan of the flags = class + 1
__global__ void __launch_bounds__(HB) hbv_class_kernel(const uint32_t* __restrict__ cls, const uint32_t* __restrict__ flag,
                                                       const uint32_t* __restrict__ codes, uint64_t n, int32_t* __restrict__ vtx_of,
                                                       uint64_t* __restrict__ run_beg) {
    const uint64_t i = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cls[i] - 1u;
    vtx_of[codes[i]] = (int32_t)c;
    if (flag[i]) run_beg[c] = i;
}

// ---- id hand-out on the device.  HBVBuilder (HBVFromEdges.cc:170-238) floods one connected component of the (edge, strand)
// graph after the other -- components in the order of their smallest seed (forward copies in BVComp order, then reverse
// copies), inside a component in first-push order of a FIFO -- so a component's ids are a block [first edge id, +size) x
// [first vertex id, +classes) that depends on nothing but the components in front of it.  Components: lock-free union-find
// over the 2U nodes (node = rc * U + rank, the seed order; larger roots hook under smaller ones, so a root IS its
// component's seed), sizes by wave-aggregated atomics, block starts by two scans over the node space.  Then one thread
// floods one component, its queue being its block of the output arrays (an id is handed out at the first push: visiting
// order == first-push order in a FIFO that skips what is done).  Components above a size limit (the connected bulk of
// a genome graph: sequential by definition) are left to the host, which floods them into their blocks.
__device__ __forceinline__ uint32_t uf_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (every loop of the device flood is bounded: a logic error must end in the error flag and the host's flood, never in a kernel that
// runs for minutes)
__device__ __forceinline__ uint32_t uf_find(uint32_t* par, uint32_t x, uint32_t* errflag) {
    for (uint32_t it = 0; it < (1u << 24); ++it) {
        const uint32_t p = uf_ld(par + x);
        if (p == x) return x;
        const uint32_t gp = uf_ld(par + p);
        if (gp != p) __hip_atomic_store(par + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving: any ancestor will do
        x = gp;
    }
    atomicOr(errflag, 1u);
    return x;
}
__device__ __forceinline__ uint32_t hbv_node(uint32_t code, uint32_t U) { return ((code >> 1) & 1u) * U + (code >> 2); }

__global__ void __launch_bounds__(HB) hbv_cc_init_kernel(uint32_t* __restrict__ par, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (i < n) par[i] = (uint32_t)i;
}
// the ends of one vertex class are mutually adjacent: every entry joins the class's first
__global__ void __launch_bounds__(HB) hbv_cc_union_kernel(const uint32_t* __restrict__ ee, const uint32_t* __restrict__ cls,
                                                          const uint64_t* __restrict__ run_beg, uint64_t n_ee, uint32_t U, uint32_t* par, uint32_t* errflag) {
    const uint64_t i = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n_ee) return;
    const uint64_t first = run_beg[cls[i] - 1u];
    if (first == i) return;
    uint32_t a = hbv_node(ee[i], U), b = hbv_node(ee[first], U);
    for (uint32_t it = 0; it < (1u << 16); ++it) {
        a = uf_find(par, a, errflag);
        b = uf_find(par, b, errflag);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }
        if (atomicCAS(par + a, a, b) == a) return;
    }
    atomicOr(errflag, 2u);
}
// counter[root] += 1 for every valid lane, one atomic per distinct root of a wave (a genome graph's bulk is ONE root)
__device__ __forceinline__ void hbv_count_root(uint32_t* counter, bool valid, uint32_t root) {
    unsigned long long act = __ballot(valid);
    const uint32_t lane = __lane_id();
    while (act) {
        const int lead = __ffsll((long long)act) - 1;
        const uint32_t r0 = (uint32_t)__shfl((int)root, lead);
        const unsigned long long same = __ballot(valid && root == r0);
        if ((int)lane == lead) atomicAdd(counter + r0, (uint32_t)__popcll(same));
        act &= ~same;
    }
}
__global__ void __launch_bounds__(HB) hbv_cc_nodes_kernel(uint32_t* par, const uint8_t* __restrict__ palr, uint32_t U, uint32_t* ce, uint32_t* errflag) {
    const uint64_t n = (uint64_t)blockIdx.x * HB + threadIdx.x;
    const bool valid = n < 2ull * U && !(n >= U && palr[n - U]);
    hbv_count_root(ce, valid, valid ? uf_find(par, (uint32_t)n, errflag) : 0u);
}
__global__ void __launch_bounds__(HB) hbv_cc_classes_kernel(uint32_t* par, const uint32_t* __restrict__ ee, const uint64_t* __restrict__ run_beg,
                                                            uint64_t nruns, uint32_t U, uint32_t* cv, uint32_t* errflag) {
    const uint64_t r = (uint64_t)blockIdx.x * HB + threadIdx.x;
    const bool valid = r < nruns;
    hbv_count_root(cv, valid, valid ? uf_find(par, hbv_node(ee[run_beg[r]], U), errflag) : 0u);
}
struct hbv_big { uint32_t root, be, bv, ce; };
// one thread per root: floods its component when that is small, lists it for the host otherwise
__global__ void __launch_bounds__(HB) hbv_flood_kernel(const uint32_t* __restrict__ par, const uint8_t* __restrict__ palr, uint32_t U,
                                                       const uint32_t* __restrict__ ce, const uint32_t* __restrict__ be,
                                                       const uint32_t* __restrict__ bv, const uint32_t* __restrict__ ee,
                                                       const int32_t* __restrict__ vtx_of, const uint64_t* __restrict__ run_beg,
                                                       uint32_t big_limit, int32_t* fwd, int32_t* rev, int32_t* vid, int32_t* v_left,
                                                       int32_t* v_right, int32_t* src, uint8_t* isrc, hbv_big* big, uint32_t big_cap,
                                                       uint32_t* n_big, uint32_t nruns, uint32_t* errflag) {
    const uint64_t n = (uint64_t)blockIdx.x * HB + threadIdx.x;
    if (n >= 2ull * U || par[n] != (uint32_t)n || (n >= U && palr[n - U])) return;
    const uint32_t size = ce[n], e_base = be[n], v_base = bv[n];
    if (size > big_limit) {
        const uint32_t q = atomicAdd(n_big, 1u);
        if (q < big_cap) big[q] = hbv_big{(uint32_t)n, e_base, v_base, size};
        return;
    }
    uint32_t head = 0, tail = 0, nv = 0;
    bool bad = false;
    auto push = [&](uint32_t e, uint32_t rc) {
        const bool p = palr[e] != 0;
        int32_t* x = (rc && !p) ? rev : fwd;
        if (x[e] != -1) return;
        if (tail >= size) { bad = true; return; }         // more nodes than the component was counted to hold: never outside its block
        const int32_t id = (int32_t)(e_base + tail++);
        x[e] = id;
        if (p) rev[e] = id;
        src[id] = (int32_t)e;
        isrc[id] = (uint8_t)rc;
    };
    push(n >= U ? (uint32_t)(n - U) : (uint32_t)n, n >= U ? 1u : 0u);
    while (head < tail && !bad) {
        const uint32_t id = e_base + head++;
        const uint32_t e = (uint32_t)src[id], rc = isrc[id];
        const int32_t r1 = vtx_of[4ull * e + 2 * rc], r2 = vtx_of[4ull * e + 2 * rc + 1];
        if ((uint32_t)r1 >= nruns || (uint32_t)r2 >= nruns) { bad = true; break; }
        if (vid[r1] == -1) vid[r1] = (int32_t)(v_base + nv++);
        if (vid[r2] == -1) vid[r2] = (int32_t)(v_base + nv++);
        v_left[id] = vid[r1];
        v_right[id] = vid[r2];
        for (int side = 0; side < 2; ++side) {
            const int32_t r = side ? r2 : r1;
            const uint64_t j0 = run_beg[r], j1 = run_beg[r + 1];
            if (j1 - j0 > 4096) { bad = true; break; }         // (a (K-1)-mer has at most eight edge ends)
            for (uint64_t j = j0; j < j1; ++j) { const uint32_t c = ee[j]; push(c >> 2, (c >> 1) & 1u); }
        }
    }
    if (bad || tail != size) atomicOr(errflag, 4u);
}

// scratch of one call: handed back to the arena when the call returns (the unitigs it reads are scratch of the
// preceding snk_dev_count_graph and must stay)
struct scratch_list {
    snk_ctx* ctx;
    std::vector<void*> p;
    ~scratch_list() { for (void* q : p) snk_ctx_release_block(ctx, q); }
};


template <typename T>
int dalloc(scratch_list& sl, size_t n, T** out, char* err, size_t errcap) {
    void* q = nullptr;
    int rc = snk_ctx_alloc(sl.ctx, std::max<size_t>(n * sizeof(T), 16), &q, err, errcap);
    if (!rc) sl.p.push_back(q);
    *out = (T*)q;
    return rc;
}

}  // namespace

extern "C" int snk_dev_hbv(snk_ctx* ctx, uint32_t K, uint64_t U, const void* d_unitig_off, const void* d_unitig_bases, snk_hbv* out,
                           float* device_ms, void* stream, char* err, size_t errcap) {
    if (!ctx || !out) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_hbv: NULL argument");
    memset(out, 0, sizeof *out);
    if (device_ms) *device_ms = 0.f;
    if (K != 48 && K != 60) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
    if (U == 0) return SNK_OK;
    if (!d_unitig_off || !d_unitig_bases) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_hbv: NULL unitig arrays");
    if (U >= (1ull << 30)) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "snk_dev_hbv: too many unitigs");
    SNK_HIP_TRY(snk_enter(ctx));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->cur_stream = st;
    const uint64_t* off = (const uint64_t*)d_unitig_off;
    const uint8_t* bases = (const uint8_t*)d_unitig_bases;
    const uint64_t n4 = 4 * U;
    int rc;
    scratch_list sl{ctx, {}};
    uint64_t *lkey, *lkey2, *run_beg;
    uint32_t *idx, *order, *codes, *codes2, *flag, *cls, *flags;
    uint8_t *pal, *palr;
    snk_u128 *keys, *keys2;
    int32_t* vtx_of;
    if ((rc = dalloc(sl, U, &lkey, err, errcap)) || (rc = dalloc(sl, U, &lkey2, err, errcap)) || (rc = dalloc(sl, U, &idx, err, errcap)) ||
        (rc = dalloc(sl, U, &order, err, errcap)) || (rc = dalloc(sl, U, &pal, err, errcap)) || (rc = dalloc(sl, U, &palr, err, errcap)) ||
        (rc = dalloc(sl, n4, &keys, err, errcap)) || (rc = dalloc(sl, n4, &keys2, err, errcap)) || (rc = dalloc(sl, n4, &codes, err, errcap)) ||
        (rc = dalloc(sl, n4, &codes2, err, errcap)) || (rc = dalloc(sl, n4, &flag, err, errcap)) || (rc = dalloc(sl, n4, &cls, err, errcap)) ||
        (rc = dalloc(sl, n4, &vtx_of, err, errcap)) || (rc = dalloc(sl, n4 + 1, &run_beg, err, errcap)) || (rc = dalloc(sl, 4, &flags, err, errcap)))
        return rc;
    hipEvent_t e0, e1;
    SNK_HIP_TRY(hipEventCreate(&e0));
    SNK_HIP_TRY(hipEventCreate(&e1));
    SNK_HIP_TRY(hipEventRecord(e0, st));
    SNK_HIP_TRY(hipMemsetAsync(flags, 0, 16, st));
    const unsigned gU = (unsigned)((U + HB - 1) / HB), g4 = (unsigned)((n4 + HB - 1) / HB);
    hipLaunchKernelGGL(hbv_head_kernel, dim3(gU), dim3(HB), 0, st, off, bases, U, K, keys, idx, pal, flags);
    // BVComp rank: sort by first k-mer, then stable sort by descending length
    size_t tmp_bytes = 0, tb2 = 0, tb3 = 0;
    SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, lkey, lkey2, idx, order, (size_t)U, 0u, 64u, st));
    SNK_HIP_TRY(rocprim::radix_sort_pairs((void*)nullptr, tb2, keys, keys2, codes, codes2, (size_t)n4, 0u, 128u, st));
    SNK_HIP_TRY(rocprim::inclusive_scan((void*)nullptr, tb3, flag, cls, (size_t)n4, rocprim::plus<uint32_t>(), st));
    size_t tb4 = 0;
    SNK_HIP_TRY(rocprim::exclusive_scan((void*)nullptr, tb4, flag, cls, 0u, (size_t)(2 * U), rocprim::plus<uint32_t>(), st));
    tmp_bytes = std::max(std::max(tmp_bytes, tb4), std::max(tb2, tb3));
    uint8_t* tmp = nullptr;
    if ((rc = dalloc(sl, tmp_bytes, &tmp, err, errcap))) return rc;
    size_t tbx = tmp_bytes;
    SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tbx, keys, keys2, idx, codes, (size_t)U, 0u, 128u, st));   // codes = first-k-mer order
    hipLaunchKernelGGL(hbv_lenkey_kernel, dim3(gU), dim3(HB), 0, st, off, codes, U, lkey);
    tbx = tmp_bytes;
    SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tbx, lkey, lkey2, codes, order, (size_t)U, 0u, 64u, st));
    hipLaunchKernelGGL(hbv_ends_kernel, dim3(g4), dim3(HB), 0, st, off, bases, order, pal, U, K, keys, codes, palr);
    // vertex-major order of the ends: stable sort by the (K-1)-mer; inside a vertex the generation order (rank, rc,
    // position) is EEComp
    tbx = tmp_bytes;
    SNK_HIP_TRY(rocprim::radix_sort_pairs(tmp, tbx, keys, keys2, codes, codes2, (size_t)n4, 0u, 128u, st));
    uint32_t h_flags[4] = {0, 0, 0, 0};
    SNK_HIP_TRY(hipMemcpyAsync(h_flags, flags, 16, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    if (h_flags[0]) return snk_fail(SNK_E_ARG, err, errcap, "snk_dev_hbv: a unitig is shorter than K");
    const uint64_t n_ee = n4 - 2ull * h_flags[1];            // the missing ends of palindromes sorted last
    const unsigned ge = (unsigned)((n_ee + HB - 1) / HB);
    hipLaunchKernelGGL(hbv_flag_kernel, dim3(ge), dim3(HB), 0, st, keys2, n_ee, flag);
    tbx = tmp_bytes;
    SNK_HIP_TRY(rocprim::inclusive_scan(tmp, tbx, flag, cls, (size_t)n_ee, rocprim::plus<uint32_t>(), st));
    SNK_HIP_TRY(hipMemsetAsync(vtx_of, 0xFF, n4 * 4, st));
    hipLaunchKernelGGL(hbv_class_kernel, dim3(ge), dim3(HB), 0, st, cls, flag, codes2, n_ee, vtx_of, run_beg);
    SNK_HIP_TRY(hipGetLastError());
    uint32_t nruns = 0;
    SNK_HIP_TRY(hipMemcpyAsync(&nruns, cls + (n_ee - 1), 4, hipMemcpyDeviceToHost, st));
    SNK_HIP_TRY(snk_sync(st));
    hbv_huge_pages = snk_opt_u32("hbv_huge_pages", 1) != 0;
    std::vector<uint32_t> h_order(U);
    huge_vec<uint32_t> h_ee;
    huge_vec<int32_t> h_vtx;
    huge_vec<uint8_t> h_pal;
    huge_vec<uint64_t> h_run;
    auto fetch_tables = [&]() -> hipError_t {       // what a flood on the host reads
        if (!h_ee.resize(n_ee) || !h_vtx.resize(n4) || !h_pal.resize(U) || !h_run.resize((size_t)nruns + 1)) return hipErrorOutOfMemory;
        hipError_t e;
        if ((e = hipMemcpyAsync(h_ee.data(), codes2, n_ee * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(h_vtx.data(), vtx_of, n4 * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(h_pal.data(), palr, U, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(h_run.data(), run_beg, (size_t)nruns * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        h_run[nruns] = n_ee;
        return hipSuccess;
    };
    // small graphs (the hot path's few thousand unitigs) are flooded on the host in less time than the launches below take.
    // Measured (tools/hbv_scale_probe.py, profiles/r04_hbv_scale.log): 9.5 M unitigs of per-barcode graphs (every component small) 4.8 s
    // -> 0.19 s per call, the flood itself ~5 ms behind the 30 ms of sorts; 6.1 M unitigs of ONE genome (the connected bulk goes to
    // the host either way) 3.15 -> 2.82 s.
    const uint64_t dev_min = snk_opt_u64("hbv_dev_min", 1ull << 16);
    if (U < dev_min) {
        SNK_HIP_TRY(hipEventRecord(e1, st));
        SNK_HIP_TRY(fetch_tables());
        SNK_HIP_TRY(hipMemcpyAsync(h_order.data(), order, U * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        if (device_ms) (void)hipEventElapsedTime(device_ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        rc = hbv_flood(U, h_pal.data(), h_ee.data(), n_ee, h_vtx.data(), h_run.data(), nruns, out, err, errcap);
        if (rc) return rc;
    } else {
        // the sort buffers are dead: node arrays live in `keys` (64 U bytes), the outputs in `keys2`, the vertex ids in `flag`
        const uint64_t n2 = 2 * U;
        uint32_t* par = (uint32_t*)keys;
        uint32_t *ce = par + n2, *cv = ce + n2, *be = cv + n2, *bv = be + n2;
        int32_t* d_fwd = (int32_t*)keys2;
        int32_t *d_rev = d_fwd + U, *d_vl = d_rev + U, *d_vr = d_vl + n2, *d_src = d_vr + n2;
        uint8_t* d_isrc = (uint8_t*)(d_src + n2);
        int32_t* d_vid = (int32_t*)flag;
        const uint32_t big_limit = (uint32_t)snk_opt_u64("hbv_big", 1024);
        const uint32_t big_cap = (uint32_t)(n2 / ((uint64_t)big_limit + 1) + 1);
        hbv_big* d_big;
        uint32_t* d_nbig;                      // [0] components for the host, [1] error flag
        if ((rc = dalloc(sl, big_cap, &d_big, err, errcap)) || (rc = dalloc(sl, 4, &d_nbig, err, errcap))) return rc;
        const uint64_t h_nee = n_ee;
        SNK_HIP_TRY(hipMemcpyAsync(run_beg + nruns, &h_nee, 8, hipMemcpyHostToDevice, st));
        SNK_HIP_TRY(hipMemsetAsync(ce, 0, n2 * 8, st));                       // ce, cv
        SNK_HIP_TRY(hipMemsetAsync(d_fwd, 0xFF, U * 8, st));                  // fwd, rev
        SNK_HIP_TRY(hipMemsetAsync(d_vid, 0xFF, (size_t)nruns * 4, st));
        SNK_HIP_TRY(hipMemsetAsync(d_nbig, 0, 8, st));
        const unsigned g2 = (unsigned)((n2 + HB - 1) / HB), gr = (unsigned)(((uint64_t)nruns + HB - 1) / HB);
        hipLaunchKernelGGL(hbv_cc_init_kernel, dim3(g2), dim3(HB), 0, st, par, n2);
        hipLaunchKernelGGL(hbv_cc_union_kernel, dim3(ge), dim3(HB), 0, st, codes2, cls, run_beg, n_ee, (uint32_t)U, par, d_nbig + 1);
        hipLaunchKernelGGL(hbv_cc_nodes_kernel, dim3(g2), dim3(HB), 0, st, par, palr, (uint32_t)U, ce, d_nbig + 1);
        hipLaunchKernelGGL(hbv_cc_classes_kernel, dim3(gr), dim3(HB), 0, st, par, codes2, run_beg, (uint64_t)nruns, (uint32_t)U, cv, d_nbig + 1);
        tbx = tmp_bytes;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tbx, ce, be, 0u, (size_t)n2, rocprim::plus<uint32_t>(), st));
        tbx = tmp_bytes;
        SNK_HIP_TRY(rocprim::exclusive_scan(tmp, tbx, cv, bv, 0u, (size_t)n2, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(hbv_flood_kernel, dim3(g2), dim3(HB), 0, st, par, palr, (uint32_t)U, ce, be, bv, codes2, vtx_of, run_beg, big_limit,
                           d_fwd, d_rev, d_vid, d_vl, d_vr, d_src, d_isrc, d_big, big_cap, d_nbig, nruns, d_nbig + 1);
        SNK_HIP_TRY(hipGetLastError());
        SNK_HIP_TRY(hipEventRecord(e1, st));
        uint32_t h_nb[2] = {0, 0};
        SNK_HIP_TRY(hipMemcpyAsync(h_nb, d_nbig, 8, hipMemcpyDeviceToHost, st));
        if ((rc = hbv_alloc_out(U, nruns, out, err, errcap))) return rc;
        SNK_HIP_TRY(hipMemcpyAsync(out->fwd_xlat, d_fwd, U * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->rev_xlat, d_rev, U * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->v_left, d_vl, n2 * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->v_right, d_vr, n2 * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->src_unitig, d_src, n2 * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(out->is_rc, d_isrc, n2, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(hipMemcpyAsync(h_order.data(), order, U * 4, hipMemcpyDeviceToHost, st));
        SNK_HIP_TRY(snk_sync(st));
        if (device_ms) (void)hipEventElapsedTime(device_ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        const uint32_t n_big = h_nb[0];
        if (h_nb[1]) {              // a bounded loop of the device flood ran out (never seen; the flood on the host does not depend on it)
            snk_hbv_free(out);
            if (snk_opt_u32("hbv_strict", 0)) return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_dev_hbv: the device flood gave up (flag %u)", h_nb[1]);
            SNK_HIP_TRY(fetch_tables());
            SNK_HIP_TRY(snk_sync(st));
            if ((rc = hbv_flood(U, h_pal.data(), h_ee.data(), n_ee, h_vtx.data(), h_run.data(), nruns, out, err, errcap))) return rc;
            goto flooded;
        }
        out->n_edges = (int32_t)(n2 - h_flags[1]);
        if (n_big) {               // the connected bulk: flooded here into the blocks the scans gave it
            if (n_big > big_cap) { snk_hbv_free(out); return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_dev_hbv: component list overflow"); }
            std::vector<hbv_big> h_big(n_big);
            SNK_HIP_TRY(hipMemcpyAsync(h_big.data(), d_big, (size_t)n_big * sizeof(hbv_big), hipMemcpyDeviceToHost, st));
            SNK_HIP_TRY(fetch_tables());
            SNK_HIP_TRY(snk_sync(st));
            huge_vec<int32_t> vidv;
            if (!vidv.resize((size_t)nruns + 1)) { snk_hbv_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_dev_hbv: host allocation failed"); }
            int32_t* vid = vidv.data();
            for (uint64_t i = 0; i < nruns; ++i) vid[i] = -1;
            const hbv_tables t{U, h_pal.data(), h_ee.data(), h_vtx.data(), h_run.data(), snk_opt_u32("hbv_short_queue", 1) != 0};
            // Components are independent once their id blocks are known (the device's scans): a host thread each, largest first.  The bulk
            // of a genome graph is TWO components -- the forward copies' and its mirror image, the reverse copies' -- whose floods are NOT
            // each other's mirror (a flood pushes the left vertex's edges before the right vertex's), so both are run, side by side.
            // (Why not on the device: profiles/r05_hbv_depth.log -- 6.1 M edge copies in 1.3 M breadth-first levels, 4.7 per level: a
            // level-synchronous reproduction is >= 1.3 M rounds of >= 3 us.)
            std::vector<uint32_t> ord(n_big);
            for (uint32_t i = 0; i < n_big; ++i) ord[i] = i;
            std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return h_big[a].ce > h_big[b].ce; });
            const uint32_t nthr = std::min<uint32_t>(n_big, std::min<uint32_t>(8u, std::max(1u, (uint32_t)snk_host_cpu_budget())));
            std::atomic<uint32_t> next{0};
            std::atomic<int> bad{0};
            auto work = [&]() {
                std::vector<uint64_t> q;
                for (;;) {
                    const uint32_t i = next.fetch_add(1);
                    if (i >= n_big) break;
                    const hbv_big& b = h_big[ord[i]];
                    int32_t next_e = (int32_t)b.be, next_v = (int32_t)b.bv;
                    hbv_flood_component(t, b.root >= U ? b.root - U : b.root, b.root >= U ? 1 : 0, vid, q, next_e, next_v, out);
                    if ((uint32_t)next_e != b.be + b.ce) bad = 1;
                }
            };
            if (nthr <= 1) work();
            else {
                std::vector<std::thread> th;
                for (uint32_t i = 0; i < nthr; ++i) th.emplace_back(work);
                for (auto& x : th) x.join();
            }
            if (bad) { snk_hbv_free(out); return snk_fail(SNK_E_INTERNAL, err, errcap, "snk_dev_hbv: a component's flood left its block"); }
        }
    }
flooded:
    out->bvcomp_order = (int32_t*)malloc(U * 4);
    if (!out->bvcomp_order) { snk_hbv_free(out); return snk_fail(SNK_E_NOMEM, err, errcap, "snk_dev_hbv: host allocation failed"); }
    for (uint64_t i = 0; i < U; ++i) out->bvcomp_order[i] = (int32_t)h_order[i];
    return SNK_OK;
}
