// snk_mspedges -- C++ host program at the ASSEMBLER_DF seam: reads.fastb / .qualp / .bci  ->  MI355X  ->  asm_graph.bv
//
// What it stands in for: the createDict + buildEdges half of DF's StageBuildGraph
// (lib/assembly/src/10X/runstages/RunStages.cc:392-437, paths/long/BuildReadQGraph48.cc:1688-1774) and tada's
// MSP -> SHARD_ASM -> MAIN_ASM_SN chain (lib/tada/mro/_asm_stages.mro:53-80).  The stock DF then takes the unitigs
// through MSPEDGES=<file.bv> (10X/DF.cc:86-207, RunStages.cc:407-413) and carries on unchanged.
//
// Arguments are KEY=VALUE like DF's own (10X/DF.cc:86-207; mro/stages/denovo/df/__init__.py:123-139):
//   LR=<head>.fastb        reads; <head>.qualp and <head>.bci must exist next to it (DF.cc:265-272)
//   OUT=<file.bv>          unitigs in the .bv hand-off format (lib/tada/src/debruijn.rs:895-929)
//   K=48|60  MIN_QUAL=7  MIN_FREQ=3  MIN_BC=2      CS-build defaults, 10X/DF.cc:138-141
//   BC_START=<n>           reads below this index ignore the barcode rule (BuildReadQGraph48.cc:158-159); default: what DF
//                          derives from <head>.dti -- the start of the first 10X dataset (DF.cc:358-363) -- or 0 without one
//   DEVICE=0               GPU ordinal
//   SPECTRUM=<file.json>   optional: k-mer spectrum as DF writes it to stats/histogram_kmer_count.json
// Exit codes follow the reference's conventions: 0 ok, 1 fatal (FatalErr), 99 out of memory (system/RunTime.cc:195-221).
// Only the C ABI of include/snk.h is used -- this file is plain C++ (g++), no HIP, no torch.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <map>
#include <string>
#include <vector>

#include "../../../include/snk.h"

namespace {

[[noreturn]] void fatal(int code, const char* what, const char* detail) {
    fprintf(stderr, "snk_mspedges: %s: %s\n", what, detail);
    exit(code == SNK_E_NOMEM ? 99 : 1);
}

std::string head_of(const std::string& path, const char* ext) {
    const size_t n = strlen(ext);
    if (path.size() < n || path.compare(path.size() - n, n, ext) != 0) fatal(SNK_E_ARG, "file has incorrect extension", path.c_str());
    return path.substr(0, path.size() - n);
}

// <head>.dti = BINWRITE vec<DataSet>, DataSet = {ReadDataType dt (u8, padded to 8 bytes); int64 start} (10X/DfTools.h:23-46):
// start of the first UNBAR_10X (2) / BAR_10X (3) dataset, 0 when there is none or no file
long long bc_start_from_dti(const std::string& head) {
    FILE* f = fopen((head + ".dti").c_str(), "rb");
    if (!f) return 0;
    char magic[8];
    uint64_t n = 0;
    long long out = 0;
    if (fread(magic, 1, 8, f) == 8 && memcmp(magic, "BINWRITE", 8) == 0 && fread(&n, 8, 1, f) == 1) {
        for (uint64_t i = 0; i < n && i < (1u << 20); ++i) {
            unsigned char rec[16];
            if (fread(rec, 1, 16, f) != 16) fatal(SNK_E_IO, "dataset index", "truncated .dti");
            if (rec[0] == 2 || rec[0] == 3) { int64_t st; memcpy(&st, rec + 8, 8); out = st; break; }
        }
    } else { fclose(f); fatal(SNK_E_IO, "dataset index", "not a BINWRITE file"); }
    fclose(f);
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    std::map<std::string, std::string> kv = {{"K", "48"}, {"MIN_QUAL", "7"}, {"MIN_FREQ", "3"}, {"MIN_BC", "2"}, {"DEVICE", "0"}};
    for (int i = 1; i < argc; ++i) {
        const char* eq = strchr(argv[i], '=');
        if (!eq) fatal(SNK_E_ARG, "argument is not KEY=VALUE", argv[i]);
        kv[std::string(argv[i], eq - argv[i])] = eq + 1;
    }
    if (!kv.count("LR") || !kv.count("OUT")) {
        fprintf(stderr, "usage: snk_mspedges LR=<reads.fastb> OUT=<asm_graph.bv> [K=48] [MIN_QUAL=7] [MIN_FREQ=3] [MIN_BC=2] "
                        "[BC_START=<from .dti>] [DEVICE=0] [SPECTRUM=<file.json>]\n");
        return 1;
    }
    const std::string head = head_of(kv["LR"], ".fastb");
    char err[512] = "";
    int rc;

    snk_ctx* ctx = nullptr;
    if ((rc = snk_ctx_create(atoi(kv["DEVICE"].c_str()), &ctx, err, sizeof err))) fatal(rc, "no usable MI355X (there is no CPU path)", err);

    uint64_t n_reads = 0;
    uint32_t max_len = 0;
    uint16_t* lens = nullptr;
    uint32_t* rows = nullptr;
    if ((rc = snk_read_fastb(kv["LR"].c_str(), &n_reads, &max_len, &lens, &rows, err, sizeof err))) fatal(rc, "reads", err);
    if (max_len == 0) max_len = 1;
    if (max_len > 256) fatal(SNK_E_UNSUPPORTED, "reads", "reads longer than 256 bases are not supported");
    // the quality rows are three quarters of what crosses PCIe: read them into page-locked memory so the DMA engine takes
    // them from where they are (pageable memory would be staged through the library's pinned ring)
    void* qp = nullptr;
    if ((rc = snk_host_alloc_pinned((size_t)n_reads * max_len, &qp, err, sizeof err))) fatal(rc, "quals", err);
    uint8_t* quals = (uint8_t*)qp;
    if ((rc = snk_read_qualp((head + ".qualp").c_str(), n_reads, max_len, quals, err, sizeof err))) fatal(rc, "quals", err);
    std::vector<int32_t> bc(n_reads);
    uint64_t n_bc = 0;
    if ((rc = snk_read_bci((head + ".bci").c_str(), n_reads, bc.data(), &n_bc, err, sizeof err))) fatal(rc, "barcode index", err);
    fprintf(stderr, "snk_mspedges: %llu reads (max %u bases), %llu barcodes\n", (unsigned long long)n_reads, max_len, (unsigned long long)n_bc);

    snk_params p;
    snk_params_default(&p);
    p.K = (uint32_t)atoi(kv["K"].c_str());
    p.min_qual = (uint32_t)atoi(kv["MIN_QUAL"].c_str());
    p.min_freq = (uint32_t)atoi(kv["MIN_FREQ"].c_str());
    p.min_bc = (uint32_t)atoi(kv["MIN_BC"].c_str());
    p.flags |= SNK_F_NO_TABLE | SNK_F_BV_IMAGE;     // the hand-off is the unitig file (+ the spectrum), not the dictionary
    snk_reads in;
    memset(&in, 0, sizeof in);
    in.n_reads = n_reads;
    in.read_len = max_len;
    in.rows = rows;
    in.lens = lens;
    in.quals = quals;
    in.bc = bc.data();
    in.ign_bc_below = kv.count("BC_START") ? atoll(kv["BC_START"].c_str()) : bc_start_from_dti(head);
    snk_result r;
    const int reps = kv.count("REPEAT") ? atoi(kv["REPEAT"].c_str()) : 1;       // timing aid: the first call sizes the buffers
    for (int rep = 0; rep < reps; ++rep) {
        if (rep) snk_free(&r);
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if ((rc = snk_count_graph(ctx, &in, &p, &r, err, sizeof err))) fatal(rc, "count+graph", err);
        FILE* f = fopen(kv["OUT"].c_str(), "wb");
        if (!f || fwrite(r.bv_image, 1, r.bv_bytes, f) != r.bv_bytes || fclose(f) != 0) fatal(SNK_E_IO, "OUT", "cannot write the unitig file");
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        fprintf(stderr, "snk_mspedges: %llu k-mer instances, %llu retained k-mers, %llu unitigs; device %.1f ms; host arrays -> %s in %.3f s = %.2f Gk-mers/s\n",
                (unsigned long long)r.n_instances, (unsigned long long)r.n_kmers, (unsigned long long)r.n_unitigs, r.phase_ms[7], kv["OUT"].c_str(), s,
                r.n_instances / s / 1e9);
    }
    if (kv.count("SPECTRUM")) {
        // same shape as WriteHistToJson(kmerspec, 0, max_count, 1, ...) (BuildReadQGraph48.cc:199-216)
        FILE* f = fopen(kv["SPECTRUM"].c_str(), "w");
        if (!f) fatal(SNK_E_IO, "SPECTRUM", "cannot open for writing");
        uint32_t last = 0;
        for (uint32_t i = 0; i < r.spectrum_bins; ++i) if (r.spectrum[i]) last = i;
        fprintf(f, "{\"name\": \"kmer_count\", \"min\": 0, \"max\": %u, \"binsize\": 1, \"vals\": [", last);
        for (uint32_t i = 0; i <= last; ++i) fprintf(f, "%s%llu", i ? ", " : "", (unsigned long long)r.spectrum[i]);
        fprintf(f, "]}\n");
        fclose(f);
    }
    snk_free(&r);
    snk_ctx_destroy(ctx);
    snk_host_free_pinned(qp);
    free(lens);
    free(rows);
    return 0;
}
