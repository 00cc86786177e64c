// oracle/ref/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Driver around the *real* reference path B (lib/assembly, C++): it #includes the
// reference translation unit so that the anonymous-namespace functions
//   createDict   (paths/long/BuildReadQGraph48.cc:218-325)
//   buildEdges   (paths/long/BuildReadQGraph48.cc:514-541)
// are callable, then mirrors the body of buildReadQGraph48 (:1688-1774, pPaths==nullptr
// branch) and dumps every intermediate the parity tests pin:
//   goodlens.u32   per-read good length            (GoodLenTailFinder :65-89)
//   kmers.bin      sorted retained k-mers: 3xu32 key, u32 count, u8 ctx(+3 pad)
//   unitigs.txt    canonical unitigs sorted by BVComp (HBVFromEdges.cc:106-111)
//   hbv.txt        buildHBVFromEdges result on the *sorted* unitigs
//   a.hbv, a.inv   that graph and its involution as DF writes them (BinaryWriter::writeFile; RunStages.cc:418, DF a.base files)
//   markdups.txt   (K=48, mode dump, even read count) MarkDups (10X/SecretOps.cc:413-593) over those paths: inter-barcode
//                  duplicate rate, logged artifactual-duplicate percentage, one flag per read pair
//   paths.txt      (K=48, mode dump) read paths of pathReads(..., NEW_ALIGNER=True) (BuildReadQGraph48.cc:1441-1469,
//                  HBVPather::algorithmTwo :1217-1336): per read "offset n e0 e1 ..." with HBV edge ids
//   stats/histogram_kmer_count.json  (written by the reference itself)
// Built only by oracle/ref/build_ref.sh from the sources where they lie in
// /root/reference; the binary lands in oracle/_ref/ (git-ignored).
//
// usage: snref_driver <in.snkr> <outdir> [threads=8] [mode=dump|time] [minQual=7 minFreq=3 minBC=2]

#ifdef SNK_REF_K60
// K=60 variant of the reference (paths/long/BuildReadQGraph60.cc: createDict :148, buildEdges :378); it has no
// barcode rule, so the barcode vector of the input is ignored.
#include "paths/long/BuildReadQGraph60.cc"
#include "system/RunTime.h"
#include "system/System.h"
#include "feudal/ObjectManager.h"
#include "feudal/PQVec.h"
#include "paths/HyperBasevector.h"
#include "paths/long/HBVFromEdges.h"
#include "ParallelVecUtilities.h"
#define SNK_KW 4
#else
#include "paths/long/BuildReadQGraph48.cc"
#include "10X/SecretOps.h"
#include <sstream>
#define SNK_KW 3
#endif

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

struct Input {
    uint64_t n = 0;
    uint32_t stride = 0;
    uint32_t has_bc = 0;
    int64_t ign_bc_below = 0;
    std::vector<uint16_t> len;
    std::vector<uint8_t> bases, quals;
    std::vector<int32_t> bc;
};

void die(const char* msg) { fprintf(stderr, "snref_driver: %s\n", msg); exit(2); }

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

struct KRec { uint32_t k[SNK_KW]; uint32_t count; uint8_t ctx; uint8_t pad[3]; };

std::string bvstr(bvec const& b) {
    std::string s(b.size(), 'A');
    for (unsigned i = 0; i < b.size(); ++i) s[i] = "ACGT"[b[i]];
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    RunTime();
    if (argc < 3) die("usage: snref_driver <in.snkr> <outdir> [threads] [dump|time] [minQual minFreq minBC]");
    std::string inpath = argv[1], outdir = argv[2];
    unsigned nt = argc > 3 ? atoi(argv[3]) : 8;
    std::string mode = argc > 4 ? argv[4] : "dump";
    unsigned minQual = argc > 5 ? atoi(argv[5]) : 7;
    unsigned minFreq = argc > 6 ? atoi(argv[6]) : 3;
    unsigned minBC = argc > 7 ? atoi(argv[7]) : 2;
    SetThreads(nt, False);

    Input in = load(inpath.c_str());
    String work(outdir.c_str());
    Mkpath(work + "/data");

    vecbvec reads;
    reads.reserve(in.n);
    ObjectManager<VecPQVec> quals(work + "/data/frag_reads_orig.qualp");
    VecPQVec& pq = quals.create();
    pq.reserve(in.n);
    size_t nInst = 0;
    {
        bvec b;
        qvec q;
        for (uint64_t r = 0; r < in.n; ++r) {
            unsigned L = in.len[r];
            b.resize(L);
            q.resize(L);
            const uint8_t* bp = &in.bases[r * in.stride];
            const uint8_t* qp = &in.quals[r * in.stride];
            for (unsigned i = 0; i < L; ++i) {
                unsigned code;
                switch (bp[i]) {
                    case 'C': code = 1; break;
                    case 'G': code = 2; break;
                    case 'T': code = 3; break;
                    default: code = 0;  // A and every non-ACGT (N->A, 10X/ParseBarcodedFastqs.cc:87-88)
                }
                b.Set(i, code);
                q[i] = qp[i];
            }
            reads.push_back(b);
            pq.push_back(PQVec(q));
        }
    }
    quals.store();          // the pathing stage reads the quality file back (VirtualMasterVec), as DF does
    vec<int32_t> bc;
    if (in.has_bc) bc.assign(in.bc.begin(), in.bc.end());
    vec<int32_t> const* bcp = in.has_bc ? &bc : nullptr;

    if (mode == "formats") {
        // the stage-input files of ASSEMBLER_DF written by the reference's own writers (data fixtures for the
        // product's format readers): <out>/reads.fastb (MasterVec<BaseVec>), reads.qualp (VecPQVec), reads.bci
        // (BINWRITE vec<int64_t>: read range of every barcode ordinal, 10X/ParseBarcodedFastqs.cc:284-293)
        reads.WriteAll(work + "/reads.fastb");
        quals.newFile(work + "/reads.qualp");
        quals.store();
        vec<int64_t> bci;
        int32_t maxbc = 0;
        for (uint64_t r = 0; r < in.n; ++r) maxbc = std::max(maxbc, in.bc[r]);
        bci.resize(maxbc + 2, 0);
        for (uint64_t r = 0; r < in.n; ++r) {
            if (r && in.bc[r] < in.bc[r - 1]) die("formats mode needs reads sorted by barcode");
            bci[in.bc[r] + 1]++;
        }
        for (size_t b = 1; b < bci.size(); ++b) bci[b] += bci[b - 1];
        BinaryWriter::writeFile(work + "/reads.bci", bci);
        printf("SNREF_FORMATS reads=%lu barcodes=%d\n", (unsigned long)in.n, maxbc);
        return 0;
    }
    if (mode == "time") {
        // whole reference path (count + unitigs + HBV, no read pathing), timed as the CPU baseline
        HyperBasevector hbv;
        {   // k-mer instances of the sample (untimed): sum over reads with goodlen >= K+1 of goodlen-K+1
            std::vector<unsigned> goodLens(reads.size());
            parallelForBatch(0ul, reads.size(), 100000, GoodLenTailFinder(quals.load(), minQual, &goodLens));
            for (unsigned g : goodLens) if (g >= K + 1) nInst += g - K + 1;
        }
        auto t0 = std::chrono::steady_clock::now();
#ifdef SNK_REF_K60
        buildReadQGraph60(reads, quals, False, False, minQual, minFreq, .75, 0, "", True, False, &hbv, nullptr, 0.5, False);
#else
        buildReadQGraph48(work, "/data/frag_reads_orig", "", reads, quals, False, False, minQual, minFreq,
                          in.ign_bc_below, minBC, bcp, .75, 0, "", True, False, &hbv, nullptr, 0.5, False);
#endif
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("SNREF_TIME seconds=%.6f threads=%u reads=%lu kmer_instances=%lu hbv_edges=%d hbv_vertices=%d\n", s, nt,
               (unsigned long)in.n, (unsigned long)nInst, hbv.EdgeObjectCount(), hbv.N());
        return 0;
    }

    // ---- good lengths (same functor the reference uses inside createDict)
    {
        std::vector<unsigned> goodLens(reads.size());
        parallelForBatch(0ul, reads.size(), 100000, GoodLenTailFinder(quals.load(), minQual, &goodLens));
        FILE* f = fopen((outdir + "/goodlens.u32").c_str(), "wb");
        fwrite(goodLens.data(), 4, goodLens.size(), f);
        fclose(f);
        for (unsigned g : goodLens) if (g >= K + 1) nInst += g - K + 1;
    }

    // ---- count + filter + contexts + adjacency prune
#ifdef SNK_REF_K60
    Dict* pDict = createDict(reads, quals, minQual, minFreq, 0.5);
#else
    Dict<BCWrapper>* pDict =
        createDict(work, reads, quals, minQual, minFreq, in.ign_bc_below, 0.5, minBC, bcp);
#endif
    {
        std::vector<KRec> recs;
        recs.reserve(pDict->size());
        for (auto const& hhs : *pDict)
            for (auto const& e : hhs) {
                KRec r;
                memset(&r, 0, sizeof r);
                for (unsigned w = 0; w < SNK_KW; ++w) {
                    uint32_t v = 0;
                    for (unsigned i = 0; i < 16; ++i) v = (v << 2) | (w * 16 + i < K ? e[w * 16 + i] : 0);
                    r.k[w] = v;
                }
                r.count = e.getKDef().getCount();
                KMerContext c = e.getKDef().getContext();
                r.ctx = (uint8_t)((c.getPredecessors() << 4) | c.getSuccessors());
                recs.push_back(r);
            }
        std::sort(recs.begin(), recs.end(), [](KRec const& a, KRec const& b) {
            for (int w = 0; w < SNK_KW; ++w) if (a.k[w] != b.k[w]) return a.k[w] < b.k[w];
            return false;
        });
        FILE* f = fopen((outdir + "/kmers.bin").c_str(), "wb");
        fwrite(recs.data(), sizeof(KRec), recs.size(), f);
        fclose(f);
    }

    // ---- unitigs
    vecbvec edges;
    edges.reserve(pDict->size() / 100);
    buildEdges(*pDict, &edges);
#ifndef SNK_REF_K60
    // ---- read paths: the pPaths != nullptr branch of buildReadQGraph48 (:1749-1769) -- graph from the edges as built, then
    //      pathReads with the new aligner (RunStages.cc:405-406 passes useNewAligner = True)
    {
        HyperBasevector hbvp;
        vec<int> fwdp, revp;
        buildHBVFromEdges(edges, K, &hbvp, &fwdp, &revp);
        reads.WriteAll(work + "/data/frag_reads_orig.fastb");
        {
            VirtualMasterVec<PQVec> vquals(quals.filename());
            VirtualMasterVec<BaseVec> vreads(work + "/data/frag_reads_orig.fastb");
            pathReads(vreads, vquals, *pDict, edges, hbvp, fwdp, revp, work + "/tmp.paths", True, False);
        }
        ReadPathVec paths(work + "/tmp.paths");
        FILE* f = fopen((outdir + "/paths.txt").c_str(), "w");
        for (size_t r = 0; r < paths.size(); ++r) {
            ReadPath const& rp = paths[r];
            fprintf(f, "%d %lu", rp.getOffset(), (unsigned long)rp.size());
            for (size_t i = 0; i < rp.size(); ++i) fprintf(f, " %d", rp[i]);
            fprintf(f, "\n");
        }
        fclose(f);
        // ---- duplicate marking: MarkDups over the read paths just made (10X/SecretOps.cc:413-593; DF.cc:597-600
        //      runs it right after the pathing).  dup = one flag per read PAIR; the rate of duplicates that involve more than
        //      one barcode; the artifactual-duplicate percentage only exists as a logged statistic, so it is taken from the log.
        if (reads.size() % 2 == 0 && reads.size() > 0) {
            VecPQVec pq2 = quals.load();
            vec<int32_t> bcm(reads.size(), 0);
            if (in.has_bc) bcm.assign(in.bc.begin(), in.bc.end());
            vec<Bool> dup;
            double interdup = 0;
            std::stringstream log;
            std::streambuf* old = std::cout.rdbuf(log.rdbuf());
            MarkDups(reads, pq2, paths, bcm, dup, interdup, False);
            std::cout.rdbuf(old);
            std::string art = "0";
            {
                const std::string l = log.str();
                size_t at = l.find("art_dup_perc=");
                if (at != std::string::npos) { size_t e = l.find('\n', at); art = l.substr(at + 13, e - at - 13); }
            }
            FILE* g = fopen((outdir + "/markdups.txt").c_str(), "w");
            fprintf(g, "%.17g %s %lu\n", interdup, art.c_str(), (unsigned long)dup.size());
            for (size_t i = 0; i < dup.size(); ++i) fputc(dup[i] ? '1' : '0', g);
            fputc('\n', g);
            fclose(g);
        }
    }
#endif
    delete pDict;

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

    // ---- graph from (sorted) unitigs
    HyperBasevector hbv;
    vec<int> fwd, rev;
    buildHBVFromEdges(sorted, K, &hbv, &fwd, &rev);
    {
        vec<int> to_left, to_right;
        hbv.ToLeft(to_left);
        hbv.ToRight(to_right);
        FILE* f = fopen((outdir + "/hbv.txt").c_str(), "w");
        fprintf(f, "N %d E %d U %lu\n", hbv.N(), hbv.EdgeObjectCount(), (unsigned long)sorted.size());
        for (int e = 0; e < hbv.EdgeObjectCount(); ++e)
            fprintf(f, "E %d %d %d %s\n", e, to_left[e], to_right[e], bvstr(hbv.EdgeObject(e)).c_str());
        for (size_t u = 0; u < sorted.size(); ++u) fprintf(f, "X %lu %d %d\n", (unsigned long)u, fwd[u], rev[u]);
        fclose(f);
    }
    {
        vec<int> inv;
        hbv.Involution(inv);
        BinaryWriter::writeFile(work + "/a.hbv", hbv);
        BinaryWriter::writeFile(work + "/a.inv", inv);
    }
    printf("SNREF_DUMP reads=%lu kmer_instances=%lu unitigs=%lu hbv_edges=%d hbv_vertices=%d\n",
           (unsigned long)in.n, (unsigned long)nInst, (unsigned long)sorted.size(), hbv.EdgeObjectCount(), hbv.N());
    return 0;
}
