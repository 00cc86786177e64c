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
//   IO_THREADS=<n> SLAB_READS=<n> READ_LEN=<n>   tuning of the ingest (pread workers, reads per slab, row length; defaults: CPU budget, 262144, longest read)
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

    // the three stage inputs are decoded on the device (include/snk.h, snk_df_*): the host moves file bytes, slab by slab, into a streamed job
    snk_df_files* files = nullptr;
    snk_df_info info;
    if ((rc = snk_df_open(kv["LR"].c_str(), (head + ".qualp").c_str(), (head + ".bci").c_str(), &files, &info, err, sizeof err))) fatal(rc, "reads", err);
    fprintf(stderr, "snk_mspedges: %llu reads, %llu barcodes, %.2f GB of stage inputs\n", (unsigned long long)info.n_reads, (unsigned long long)info.n_barcodes,
            (info.fastb_bytes + info.qualp_bytes + info.bci_bytes) / 1e9);
    if (info.n_reads == 0) fatal(SNK_E_ARG, "reads", "the read file is empty");

    snk_params p;
    snk_params_default(&p);
    p.K = (uint32_t)atoi(kv["K"].c_str());
    p.min_qual = (uint32_t)atoi(kv["MIN_QUAL"].c_str());
    p.min_freq = (uint32_t)atoi(kv["MIN_FREQ"].c_str());
    p.min_bc = (uint32_t)atoi(kv["MIN_BC"].c_str());
    p.flags |= SNK_F_UNSORTED_TABLE;                 // the hand-off is the unitig file (+ the spectrum), not the dictionary
    const long long ign_bc_below = kv.count("BC_START") ? atoll(kv["BC_START"].c_str()) : bc_start_from_dti(head);
    const uint32_t threads = kv.count("IO_THREADS") ? (uint32_t)atoi(kv["IO_THREADS"].c_str()) : 0u;
    const uint64_t slab = kv.count("SLAB_READS") ? strtoull(kv["SLAB_READS"].c_str(), nullptr, 10) : 0ull;
    const uint32_t read_len = kv.count("READ_LEN") ? (uint32_t)atoi(kv["READ_LEN"].c_str()) : 0u;      // 0: the longest read of the file
    snk_dev_result r;
    std::vector<uint64_t> spectrum;
    const int reps = kv.count("REPEAT") ? atoi(kv["REPEAT"].c_str()) : 1;       // timing aid: the first call sizes the buffers
    for (int rep = 0; rep < reps; ++rep) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        snk_dev_ingest st;
        if ((rc = snk_dev_ingest_df_count_graph(ctx, files, 0, info.n_reads, read_len, threads, slab, &p, ign_bc_below, &r, &st, err, sizeof err))) fatal(rc, "count+graph", err);
        const void* d_image = nullptr;
        uint64_t image_bytes = 0;
        if ((rc = snk_dev_bv_image(ctx, p.K, r.n_unitigs, r.unitig_off, r.unitig_bases, 1, &d_image, &image_bytes, nullptr, err, sizeof err))) fatal(rc, "unitig file image", err);
        std::vector<uint8_t> image(image_bytes);
        if ((rc = snk_dev_download(ctx, d_image, image.data(), image_bytes, nullptr))) fatal(rc, "download", snk_last_error());
        spectrum.assign(r.spectrum_bins, 0);
        if (r.spectrum_bins && (rc = snk_dev_download(ctx, r.spectrum, spectrum.data(), 8ull * r.spectrum_bins, nullptr))) fatal(rc, "download", snk_last_error());
        FILE* f = fopen(kv["OUT"].c_str(), "wb");
        if (!f || fwrite(image.data(), 1, image_bytes, f) != image_bytes || fclose(f) != 0) fatal(SNK_E_IO, "OUT", "cannot write the unitig file");
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        fprintf(stderr, "snk_mspedges: %llu k-mer instances, %llu retained k-mers, %llu unitigs; %.2f GB of file bytes in %u slabs, %.3f s ingest+partition (%.1f GB/s, "
                        "%.3f s waiting for file bytes, %.3f s setup), count+graph %.1f ms; files -> %s in %.3f s = %.2f Gk-mers/s\n",
                (unsigned long long)r.n_instances, (unsigned long long)r.n_kmers, (unsigned long long)r.n_unitigs, st.text_bytes / 1e9, st.n_batches, st.seconds,
                st.text_bytes / 1e9 / (st.seconds > 0 ? st.seconds : 1), st.decode_wait_seconds, st.setup_seconds, r.phase_ms[7], kv["OUT"].c_str(), s, r.n_instances / s / 1e9);
        if (kv.count("PHASES")) fprintf(stderr, "snk_mspedges: phases (ms): trim %.1f plan %.1f partition %.1f count %.1f sort %.1f graph %.1f | buckets %llu split %llu repartitioned %u scratch %.1f GB\n",
                                        r.phase_ms[0], r.phase_ms[1], r.phase_ms[2], r.phase_ms[3], r.phase_ms[4], r.phase_ms[5], (unsigned long long)r.n_buckets,
                                        (unsigned long long)r.buckets_split, (unsigned)r.repartitioned, r.scratch_bytes / 1e9);
    }
    if (kv.count("SPECTRUM")) {
        // same shape as WriteHistToJson(kmerspec, 0, max_count, 1, ...) (BuildReadQGraph48.cc:199-216)
        FILE* f = fopen(kv["SPECTRUM"].c_str(), "w");
        if (!f) fatal(SNK_E_IO, "SPECTRUM", "cannot open for writing");
        uint32_t last = 0;
        for (size_t i = 0; i < spectrum.size(); ++i) if (spectrum[i]) last = (uint32_t)i;
        fprintf(f, "{\"name\": \"kmer_count\", \"min\": 0, \"max\": %u, \"binsize\": 1, \"vals\": [", last);
        for (uint32_t i = 0; i <= last && i < spectrum.size(); ++i) fprintf(f, "%s%llu", i ? ", " : "", (unsigned long long)spectrum[i]);
        fprintf(f, "]}\n");
        fclose(f);
    }
    snk_df_close(files);
    snk_ctx_destroy(ctx);
    return 0;
}
