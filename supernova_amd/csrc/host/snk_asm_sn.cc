// snk_asm_sn -- C++ host of the minimiser-sharded N-GPU job: one process per GPU, RCCL over xGMI, nothing but the C ABI.
//
// What it stands in for: tada's _ASM_SN pipeline run as ONE job over N GPUs -- MSP (lib/tada/src/cmd_msp.rs:38-245), the shardio
// exchange (rust-shardio/src/shard.rs:184-211,488-493), SHARD_ASM (cmd_shard_asm.rs:37-94) and MAIN_ASM_SN, which writes
// asm_graph.bv (cmd_main_asm.rs:25-89,184-193) -- FASTH files in, the unitig hand-off file out.  Every process runs
//     snk_shard_step   (partition -> exchanges -> count -> prune -> fragments -> owner-side join, include/snk.h)
// on its slab, then snk_shard_gather_unitigs brings the unitigs to rank 0, which writes OUT in BVComp order.
//
// Arguments are KEY=VALUE like DF's (10X/DF.cc:86-207):
//   WORLD=<n> RANK=<r>      the job's size and this process's rank (default 1 / 0; $WORLD_SIZE / $RANK are honoured)
//   ID_FILE=<path>          rendezvous: rank 0 writes the 128-byte RCCL unique id there, the others wait for it (shared filesystem)
//   DEVICE=<d>              GPU ordinal (default: RANK modulo the visible devices)
//   one input of
//     FASTH=<f1,f2,...>|@<list file>  WHITELIST=<barcode whitelist>   rank r takes files r, r+WORLD, ... (MSP's chunks)
//     LR=<head>.fastb                 (+ .qualp, .bci)  rank r takes reads [n*r/WORLD, n*(r+1)/WORLD) rounded to pairs
//     SYNTH=<total reads> [SEED=<s>]  the synthetic linked-read model, generated in HBM
//   OUT=<asm_graph.bv>      written by rank 0;  K MIN_QUAL MIN_FREQ MIN_BC as for snk_mspedges;  BC_START=<n>
//   STEPS=<k>               repeat the step (timing); STATS=<file> rank 0 appends one JSON line per step
// Exit codes: 0 ok, 1 fatal, 99 out of memory (system/RunTime.cc:195-221).  Plain C++ + the HIP runtime API for device buffers.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <fstream>
#include <map>
#include <string>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "../../../include/snk.h"

namespace {

int g_rank = 0;
[[noreturn]] void fatal(int code, const char* what, const char* detail) {
    fprintf(stderr, "snk_asm_sn[%d]: %s: %s\n", g_rank, what, detail);
    exit(code == SNK_E_NOMEM ? 99 : 1);
}
void hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) fatal(e == hipErrorOutOfMemory ? SNK_E_NOMEM : SNK_E_HIP, what, hipGetErrorString(e)); }
double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

std::vector<std::string> split_list(const std::string& v) {
    std::vector<std::string> out;
    if (!v.empty() && v[0] == '@') {
        std::ifstream f(v.substr(1));
        if (!f) fatal(SNK_E_IO, "cannot read the file list", v.c_str() + 1);
        std::string ln;
        while (std::getline(f, ln)) if (!ln.empty()) out.push_back(ln);
        return out;
    }
    size_t a = 0;
    while (a <= v.size()) {
        const size_t b = v.find(',', a);
        const std::string s = v.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (!s.empty()) out.push_back(s);
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}

// The file is the 128-byte unique id followed by a 64-byte job nonce (JOB_ID=..., default: MASTER_PORT of the launcher, else empty).
// Rank 0 removes whatever a crashed run left behind BEFORE it makes the id; the other ranks only take a file that carries this job's
// nonce -- a stale id from another job would send them into ncclCommInitRank with nobody on the other side.
void rendezvous(const std::string& path, const std::string& job, int rank, int world, unsigned char id[128]) {
    char err[512] = "";
    char nonce[64];
    memset(nonce, 0, sizeof nonce);
    memcpy(nonce, job.data(), job.size() < 63 ? job.size() : 63);
    if (rank == 0) {
        if (world > 1 && !path.empty()) (void)unlink(path.c_str());
        int rc = snk_comm_unique_id(id, err, sizeof err);
        if (rc) fatal(rc, "RCCL unique id", err);
        if (world > 1) {
            if (path.empty()) fatal(SNK_E_ARG, "WORLD > 1", "ID_FILE=<path> is needed for the rendezvous");
            const std::string tmp = path + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128 || fwrite(nonce, 1, 64, f) != 64 || fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0)
                fatal(SNK_E_IO, "ID_FILE", "cannot write the unique id");
        }
        return;
    }
    if (path.empty()) fatal(SNK_E_ARG, "WORLD > 1", "ID_FILE=<path> is needed for the rendezvous");
    for (int tries = 0; tries < 6000; ++tries) {         // ten minutes
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            unsigned char buf[192];
            const size_t n = fread(buf, 1, 192, f);
            fclose(f);
            if (n == 192 && memcmp(buf + 128, nonce, 64) == 0) { memcpy(id, buf, 128); return; }
        }
        usleep(100000);
    }
    fatal(SNK_E_IO, "ID_FILE", "rank 0's unique id (with this job's JOB_ID) did not appear");
}

}  // namespace

int main(int argc, char** argv) {
    std::map<std::string, std::string> kv = {{"K", "48"}, {"MIN_QUAL", "7"}, {"MIN_FREQ", "3"}, {"MIN_BC", "2"}, {"STEPS", "1"}, {"SEED", "1592590337"}};
    if (getenv("WORLD_SIZE")) kv["WORLD"] = getenv("WORLD_SIZE");
    if (getenv("RANK")) kv["RANK"] = getenv("RANK");
    for (int i = 1; i < argc; ++i) {
        const char* eq = strchr(argv[i], '=');
        if (!eq) fatal(SNK_E_ARG, "argument is not KEY=VALUE", argv[i]);
        kv[std::string(argv[i], eq - argv[i])] = eq + 1;
    }
    const int world = kv.count("WORLD") ? atoi(kv["WORLD"].c_str()) : 1;
    const int rank = kv.count("RANK") ? atoi(kv["RANK"].c_str()) : 0;
    g_rank = rank;
    if (world < 1 || rank < 0 || rank >= world) fatal(SNK_E_ARG, "WORLD / RANK", "out of range");
    const int n_in = (int)kv.count("FASTH") + (int)kv.count("LR") + (int)kv.count("SYNTH");
    if (n_in != 1 || (rank == 0 && !kv.count("OUT"))) {
        fprintf(stderr, "usage: snk_asm_sn [WORLD=n RANK=r ID_FILE=<path> JOB_ID=<nonce>] (FASTH=<files> WHITELIST=<txt> | LR=<reads.fastb> | SYNTH=<reads> [SEED=s]) OUT=<asm_graph.bv> "
                        "[K=48] [MIN_QUAL=7] [MIN_FREQ=3] [MIN_BC=2] [LONG_MINIMISER=1] [BC_START=n] [DEVICE=d] [STEPS=k] [STATS=<file>]\n");
        return 1;
    }
    char err[512] = "";
    int rc, ndev = 0;
    hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (ndev < 1) fatal(SNK_E_NOGPU, "no usable MI355X", "no HIP device visible (there is no CPU path)");
    const int device = kv.count("DEVICE") ? atoi(kv["DEVICE"].c_str()) : rank % ndev;
    snk_ctx* ctx = nullptr;
    if ((rc = snk_ctx_create(device, &ctx, err, sizeof err))) fatal(rc, "no usable MI355X (there is no CPU path)", err);
    hip_ok(hipSetDevice(device), "hipSetDevice");

    snk_params p;
    snk_params_default(&p);
    p.K = (uint32_t)atoi(kv["K"].c_str());
    p.min_qual = (uint32_t)atoi(kv["MIN_QUAL"].c_str());
    p.min_freq = (uint32_t)atoi(kv["MIN_FREQ"].c_str());
    if (kv.count("LONG_MINIMISER") && atoi(kv["LONG_MINIMISER"].c_str())) p.flags |= SNK_F_LONG_MINIMISER;      // genomes of human size (snk.h)
    p.min_bc = (uint32_t)atoi(kv["MIN_BC"].c_str());

    // ---- this rank's slab of reads, resident in HBM
    snk_dev_reads in;
    memset(&in, 0, sizeof in);
    snk_dev_ingest ing;
    memset(&ing, 0, sizeof ing);
    uint64_t total_reads = 0;
    void *d_rows = nullptr, *d_quals = nullptr, *d_bc = nullptr, *d_lens = nullptr;
    const double t_in0 = now_s();
    if (kv.count("SYNTH")) {
        total_reads = strtoull(kv["SYNTH"].c_str(), nullptr, 10) & ~1ull;
        snk_synth_params sp;
        snk_synth_default(&sp, total_reads, strtoull(kv["SEED"].c_str(), nullptr, 0), 0);
        const uint64_t lo = (total_reads / 2 * rank / world) * 2, hi = (total_reads / 2 * (rank + 1) / world) * 2, n = hi - lo;
        const uint32_t rw = (sp.read_len + 15) / 16, qs = rw * 16;
        hip_ok(hipMalloc(&d_rows, (n + 1) * rw * 4ull), "hipMalloc rows");
        hip_ok(hipMalloc(&d_quals, (n + 1) * (uint64_t)qs), "hipMalloc quals");
        hip_ok(hipMalloc(&d_bc, (n + 1) * 4ull), "hipMalloc bc");
        if ((rc = snk_synth_dev(ctx, &sp, lo, n, d_rows, rw, d_quals, qs, d_bc, nullptr))) fatal(rc, "synthetic reads", snk_last_error());
        hip_ok(hipDeviceSynchronize(), "synth");
        in.n_reads = n; in.rows = d_rows; in.row_words = rw; in.read_len = sp.read_len; in.quals = d_quals; in.qstride = qs; in.bc = d_bc;
        in.read_index_base = lo;
    } else if (kv.count("FASTH")) {
        const std::vector<std::string> all = split_list(kv["FASTH"]);
        std::vector<const char*> mine;
        for (size_t i = (size_t)rank; i < all.size(); i += (size_t)world) mine.push_back(all[i].c_str());
        snk_bc_index* ix = nullptr;
        if (kv.count("WHITELIST")) {
            std::ifstream f(kv["WHITELIST"], std::ios::binary);
            if (!f) fatal(SNK_E_IO, "WHITELIST", "cannot open");
            const std::string wl((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if ((rc = snk_bc_index_create(ctx, wl.data(), wl.size(), &ix, err, sizeof err))) fatal(rc, "barcode whitelist", err);
        }
        const uint32_t read_len = kv.count("READ_LEN") ? (uint32_t)atoi(kv["READ_LEN"].c_str()) : 150u;
        if (!mine.empty()) {
            if ((rc = snk_dev_ingest_fasth(ctx, mine.data(), (uint32_t)mine.size(), read_len, ix, 0, 0, &ing, err, sizeof err))) fatal(rc, "FASTH ingest", err);
        } else { ing.read_len = read_len; ing.row_words = (read_len + 15) / 16; ing.qstride = ing.row_words * 16; }
        if (ix) snk_bc_index_destroy(ix);
        in.n_reads = ing.n_reads; in.rows = ing.rows; in.row_words = ing.row_words; in.read_len = ing.read_len; in.lens = ing.lens;
        in.quals = ing.quals; in.qstride = ing.qstride; in.bc = ing.bc;
        total_reads = 0;              // the ranks exchange their slab sizes
    } else {
        const std::string lr = kv["LR"];
        if (lr.size() < 6 || lr.compare(lr.size() - 6, 6, ".fastb") != 0) fatal(SNK_E_ARG, "file has incorrect extension", lr.c_str());
        const std::string head = lr.substr(0, lr.size() - 6);
        // the stage inputs are decoded on the device, and this rank touches only the bytes of ITS reads (the offset tables say where they are)
        snk_df_files* files = nullptr;
        snk_df_info info;
        if ((rc = snk_df_open(lr.c_str(), (head + ".qualp").c_str(), (head + ".bci").c_str(), &files, &info, err, sizeof err))) fatal(rc, "reads", err);
        const uint64_t n_all = info.n_reads;
        total_reads = n_all;
        const uint64_t lo = (n_all / 2 * rank / world) * 2, hi = rank + 1 == world ? n_all : (n_all / 2 * (rank + 1) / world) * 2, n = hi - lo;
        // every rank lays its rows out alike: READ_LEN, or the longest read of the FILE (one scan of its length table, 4 of a read's ~50 bytes)
        const uint32_t read_len = kv.count("READ_LEN") ? (uint32_t)atoi(kv["READ_LEN"].c_str()) : 0u;
        // ... in their compact form: packed rows, good lengths (every slab trimmed as it is decoded, its quality rows dropped), barcode ids --
        // 46 instead of 204 bytes per read of device memory, and all the step needs
        if ((rc = snk_dev_ingest_df_trimmed(ctx, files, lo, n, read_len, 0, 0, p.K, p.min_qual, &ing, err, sizeof err))) fatal(rc, "reads", err);
        snk_df_close(files);
        in.n_reads = ing.n_reads; in.rows = ing.rows; in.row_words = ing.row_words; in.read_len = ing.read_len; in.good_len = ing.good_len;
        in.qstride = ing.qstride; in.bc = ing.bc;
        in.read_index_base = lo;
        fprintf(stderr, "snk_asm_sn[%d/%d]: reads [%llu, %llu) of %llu: %.2f GB of file bytes in %.3f s (%.1f GB/s)\n", rank, world, (unsigned long long)lo,
                (unsigned long long)hi, (unsigned long long)n_all, ing.text_bytes / 1e9, ing.seconds, ing.text_bytes / 1e9 / (ing.seconds > 0 ? ing.seconds : 1));
    }
    if (kv.count("BC_START")) in.ign_bc_below = atoll(kv["BC_START"].c_str());
    const double t_in = now_s() - t_in0;

    // ---- the communicator
    unsigned char id[128];
    const std::string job = kv.count("JOB_ID") ? kv["JOB_ID"] : std::string(getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "");
    rendezvous(kv.count("ID_FILE") ? kv["ID_FILE"] : std::string(), job, rank, world, id);
    snk_comm* comm = nullptr;
    if ((rc = snk_comm_create_rccl(ctx, id, (uint32_t)rank, (uint32_t)world, &comm, err, sizeof err))) fatal(rc, "RCCL communicator", err);

    // ---- the step(s), the gather, the file
    const int steps = atoi(kv["STEPS"].c_str()) > 0 ? atoi(kv["STEPS"].c_str()) : 1;
    snk_shard_result res;
    snk_result un;
    memset(&un, 0, sizeof un);
    for (int s = 0; s < steps; ++s) {
        const double t0 = now_s();
        if ((rc = snk_shard_step(ctx, comm, &in, &p, total_reads, 0, &res, nullptr, err, sizeof err))) fatal(rc, "count+graph step", err);
        const double t1 = now_s();
        if (s + 1 == steps) {
            if ((rc = snk_shard_gather_unitigs(ctx, comm, &res, p.K, 0, SNK_F_BV_IMAGE, &un, nullptr, err, sizeof err))) fatal(rc, "unitig gather", err);
        }
        const double t2 = now_s();
        fprintf(stderr, "snk_asm_sn[%d/%d]: step %d: %llu reads, %llu k-mer instances, %llu retained k-mers, %llu fragments, %llu unitigs written here; "
                        "%.1f ms on the device (%.1f ms wall), %u host read-backs, %.1f MB sent%s\n",
                rank, world, s, (unsigned long long)res.n_reads, (unsigned long long)res.n_instances, (unsigned long long)res.n_kmers,
                (unsigned long long)res.n_frags, (unsigned long long)res.n_unitigs, res.phase_ms[7], 1e3 * (t1 - t0), res.host_syncs,
                res.exchanged_bytes[7] / 1e6, s + 1 == steps ? "" : "");
        if (rank == 0 && kv.count("STATS")) {
            FILE* f = fopen(kv["STATS"].c_str(), "a");
            if (f) {
                fprintf(f, "{\"step\": %d, \"world\": %d, \"reads_rank0\": %llu, \"kmer_instances_rank0\": %llu, \"retained_kmers_rank0\": %llu, \"step_ms\": %.3f, "
                           "\"wall_ms\": %.3f, \"gather_ms\": %.3f, \"host_syncs\": %u, \"ranking\": \"%s\", \"input_s\": %.3f, "
                           "\"phase_ms\": [%.3f, %.3f, %.3f, %.3f, %.3f, %.3f, %.3f]}\n",
                        s, world, (unsigned long long)res.n_reads, (unsigned long long)res.n_instances, (unsigned long long)res.n_kmers, res.phase_ms[7],
                        1e3 * (t1 - t0), 1e3 * (t2 - t1), res.host_syncs, res.ranking ? "partitioned" : "replicated", t_in, res.phase_ms[0], res.phase_ms[1],
                        res.phase_ms[2], res.phase_ms[3], res.phase_ms[4], res.phase_ms[5], res.phase_ms[6]);
                fclose(f);
            }
        }
    }
    if (rank == 0) {
        FILE* f = fopen(kv["OUT"].c_str(), "wb");
        if (!f || fwrite(un.bv_image, 1, un.bv_bytes, f) != un.bv_bytes || fclose(f) != 0) fatal(SNK_E_IO, "OUT", "cannot write the unitig file");
        fprintf(stderr, "snk_asm_sn[0/%d]: %llu unitigs of the whole job -> %s (%llu bytes)\n", world, (unsigned long long)un.n_unitigs, kv["OUT"].c_str(),
                (unsigned long long)un.bv_bytes);
    }
    snk_free(&un);
    snk_comm_destroy(comm);
    if (ing.rows) snk_dev_ingest_free(&ing);
    (void)hipFree(d_rows); (void)hipFree(d_quals); (void)hipFree(d_bc); (void)hipFree(d_lens);
    snk_ctx_destroy(ctx);
    if (rank == 0 && world > 1 && kv.count("ID_FILE")) unlink(kv["ID_FILE"].c_str());
    return 0;
}
