// snk_opts.hip -- the option registry behind snk_ctx_set_option / snk_ctx_set_tuning (snk_opts.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "snk_ctx.h"
#include "snk_opts.h"

const snk_opt_def snk_opt_defs[] = {
    // ---- which count kernel runs, how full its tables may get, how large a bucket is (snk_pipeline.hip, snk_shard_step.hip)
    {"count_tight", "count kernel with booked table slots: 0 never, n = always, n usable slots of the table (256 .. slots - 64); unset: chosen from the data"},
    {"count_screen", "per-barcode groups: bit filter in front of the table: 0 off, 1 on for min_freq >= 3 (default), 2 on for min_freq >= 2"},
    {"count_screen_ng", "ungrouped reads: bit filter in front of a 1024-slot table: 0 never, 1 when the tables run full (default), 2 always"},
    {"screen_ratio_pct", "count_screen_ng = 1: distinct k-mers per 100 instances above which the filter goes on (30)"},
    {"screen_target", "k-mer instances per bucket behind the ungrouped filter (4000)"},
    {"tight_tries", "booked slots: how often a wave looks again before it gives the pass up (48)"},
    {"target_inst", "k-mer instances per minimiser bucket; unset: 5000 (K=48) / 3500 (K=60), adapted to the data"},
    {"bucket_fill_pct", "adaptive buckets aim at this share of the table's usable slots (50)"},
    {"adaptive_buckets", "look at the first buckets of unknown data and partition a second time if their tables run full (1)"},
    {"chunk_kmers", "retained k-mers per bucket the bucket count aims at when the data retain many (180)"},
    {"count_persist", "residency waves of count workgroups (32)"},
    {"input_fp", "fingerprint the reads so that other data of the same size do not inherit sizing history (1)"},
    {"pilot_est", "size the count regions from the pilot launch (1)"},
    {"minimiser_len", "16 or 20: overrides SNK_F_LONG_MINIMISER (tools, tests)"},
    {"global_graph", "1: the global graph stage (as SNK_F_GLOBAL_GRAPH)"},
    // ---- partition
    {"partition_passes", "bucket-range passes of the partition: 0 = as many as the device needs (default)"},
    {"msp_cap_pct", "bucket slot capacity in percent of the occupancy model's (100; tests shrink it to force the overflow segment)"},
    {"msp_sigmas_x10", "bucket slot capacity = mean + this/10 sigma of the occupancy model; unset: 5, or 3 / 1.5 when the slots would take more than 30 % of the device"},
    {"msp_site_records", "supermers per minimiser site in the occupancy model (48; groups 3)"},
    {"msp_dense", "1: reservation-free partition (dense records + sorted index list)"},
    {"msp_hot_factor", "a bucket is noted hot at this multiple of its capacity (32)"},
    {"msp_hot_min", "... and at least this many reservations (4096)"},
    {"trim_fused", "quality trim inside the partition kernel (1)"},
    {"trim_rowwise", "1: the row-wise trim kernel for every layout"},
    {"defer_compact", "leave the count regions in place until the prune reads them (1)"},
    // ---- hot minimiser buckets
    {"hot", "re-partition hot minimiser buckets by k-mer hash (1)"},
    {"hot_min", "a bucket is hot above this many records (8192) ..."},
    {"hot_factor", "... and this multiple of the slot capacity (8)"},
    {"hot_class_inst", "k-mer instances per hash class of a hot bucket (6000)"},
    // ---- bucket-local graph, join
    {"chunk_merge", "graph chunks are merged up to this many k-mers (256; 0 off)"},
    {"bl_cpw", "graph chunks per workgroup of the prune (4)"},
    {"bl_index_fused", "boundary index built by the prune (1)"},
    {"bl_noclassify", "1: every miss of the prune is pending (no neighbour classification)"},
    {"bl_pool", "slots of the in-chunk circle pool (4096; 0 forces the exact re-run)"},
    {"split_log2", "log2 of the ranking's splitter spacing + 1 (5)"},
    {"rank_wyllie", "1: plain pointer jumping instead of the sparse ruling set"},
    {"emit_grid_log2", "log2 of the largest grid of the join's fragment copy (22; tests make it small: the kernel strides)"},
    {"lean_cold", "0: a context whose arena has not mapped the memory yet still sizes its record slots at 5 sigma (1: 1.5 sigma until the arena has the slack)"},
    {"plan_mem_mb", "the memory (MB) the slot / pass / region plans of a call divide instead of what the device has free (tests: bucket-range passes and their region probe at fixture size)"},
    {"hbv_huge_pages", "0: the host tables of a14's flood are plain malloc memory (1: 2-MB aligned with MADV_HUGEPAGE)"},
    {"hbv_short_queue", "0: the host flood of a14 prefetches 16 / 8 / 4 queue places ahead only (1: also at push time and one / two places ahead: the bulk of a genome graph has a short queue)"},
    {"join_dbg", "1: the join counts the bytes of its unitig buffers that nothing wrote (stderr; debugging aid)"},
    {"rank_round_batch", "jumping rounds per read-back, sharded ranking (8)"},
    {"rank_round_batch0", "jumping rounds before the first read-back, one-GPU ranking (12)"},
    // ---- sharded step
    {"exchange_ranges", "bucket ranges of the record exchange (4)"},
    {"join_replicated", "1: replicated list ranking instead of the partitioned one"},
    {"dbg_fake_segs", "count kernel: see the slots as this many record segments (measurement of the N-rank read pattern)"},
    // ---- read pathing, duplicates, HBV
    {"path_index", "look-ups through the minimiser index: 1 always, 0 never; unset: when the k-mer dictionary does not fit"},
    {"path_dict_max_kb", "a k-mer dictionary above this size 'does not fit' (tests)"},
    {"path_slots_x10", "dictionary slots per unitig k-mer x 10 (30)"},
    {"path_two_pass", "second pass with 16 lanes per read (1)"},
    {"path_fast_gs", "lanes per read of the first pass (8)"},
    {"path_fused", "1: one kernel for both passes"},
    {"path_redo_all", "1: every read takes the second pass (tests)"},
    {"path_fp_mask", "mask of the dictionary's fingerprints (tests: collisions)"},
    {"path_idx_dbg", "index look-up debug mode"},
    {"unitig_bc_cut", "entries a unitig's barcode list is cut at (20000)"},
    {"dups_two_sorts", "1: the two-pass sort of MarkDups"},
    {"hbv_dev_min", "graphs below this many unitigs take the host id hand-out (65536)"},
    {"hbv_big", "components above this many nodes take the host flood (1024)"},
    {"hbv_strict", "1: fail instead of falling back when the device flood gives up"},
    {"df_stream", "stage-input files -> unitigs: 2 = through a streamed job (the reads never resident in any form); else the compact resident form (rows + good lengths + barcode ids, the adaptive resident step)"},
    // ---- memory
    {"arena_vmm", "growing virtual-memory arena (1); 0 = cached hipMalloc blocks"},
    // ---- kernel debug modes (results invalid unless stated)
    {"count_dbg", "count kernel probe mode"},
    {"msp_dbg", "partition kernel probe mode"},
    {"overlap_probe", "builds with -DSNK_PROBES: a second kernel next to the count kernel (tools/overlap_probe*.py)"},
    {"overlap_probe_dbg", "... which one"},
};

int snk_opt_count() { return (int)(sizeof snk_opt_defs / sizeof snk_opt_defs[0]); }
static_assert(sizeof snk_opt_defs / sizeof snk_opt_defs[0] <= SNK_MAX_OPTS, "raise SNK_MAX_OPTS");

int snk_opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < snk_opt_count(); ++i) if (strcmp(snk_opt_defs[i].name, name) == 0) return i;
    return -1;
}
void snk_opts_init(snk_opts* o) { memset(o, 0, sizeof *o); }

int snk_opts_parse(snk_opts* o, const char* text, char* bad, unsigned badcap) {
    if (!text) return 0;
    const char* p = text;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == ';') ++p;
        if (!*p) break;
        const char* e = p;
        while (*e && *e != ',' && *e != ';' && *e != ' ') ++e;
        char item[96];
        const size_t n = (size_t)(e - p) < sizeof item - 1 ? (size_t)(e - p) : sizeof item - 1;
        memcpy(item, p, n);
        item[n] = 0;
        char* eq = strchr(item, '=');
        int ix = -1;
        char* endp = nullptr;
        long long v = 0;
        if (eq) { *eq = 0; ix = snk_opt_index(item); v = strtoll(eq + 1, &endp, 0); }
        if (ix < 0 || !eq || endp == eq + 1 || *endp) {
            if (eq) *eq = '=';
            if (bad && badcap) { strncpy(bad, item, badcap - 1); bad[badcap - 1] = 0; }
            return -1;
        }
        o->v[ix] = v; o->set[ix] = true;
        p = e;
    }
    return 0;
}

static thread_local const snk_opts* g_opts = nullptr;
void snk_opts_enter(const snk_opts* o) { g_opts = o; }

static int lookup(const char* name) {
    const int ix = snk_opt_index(name);
    if (ix < 0) { fprintf(stderr, "libsnk: internal: option '%s' is not in the registry (snk_opts.hip)\n", name); abort(); }
    return ix;
}
uint32_t snk_opt_u32(const char* name, uint32_t dflt) {
    const int ix = lookup(name);
    return (g_opts && g_opts->set[ix]) ? (uint32_t)g_opts->v[ix] : dflt;
}
unsigned long long snk_opt_u64(const char* name, unsigned long long dflt) {
    const int ix = lookup(name);
    return (g_opts && g_opts->set[ix]) ? (unsigned long long)g_opts->v[ix] : dflt;
}
bool snk_opt_is_set(const char* name) {
    const int ix = lookup(name);
    return g_opts && g_opts->set[ix];
}

hipError_t snk_enter(snk_ctx* ctx) {
    snk_opts_enter(ctx ? &ctx->opts : nullptr);
    return hipSetDevice(ctx->device);
}

// ---- C ABI
extern "C" int snk_ctx_set_option(snk_ctx* ctx, const char* name, long long value, char* err, size_t errcap) {
    if (!ctx || !name) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_option: NULL argument");
    const int ix = snk_opt_index(name);
    if (ix < 0) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_option: no option '%s' (snk_option_name lists them)", name);
    ctx->opts.v[ix] = value; ctx->opts.set[ix] = true;
    return SNK_OK;
}
extern "C" int snk_ctx_clear_option(snk_ctx* ctx, const char* name) {
    if (!ctx) return SNK_E_ARG;
    if (!name) { snk_opts_init(&ctx->opts); return SNK_OK; }
    const int ix = snk_opt_index(name);
    if (ix < 0) return SNK_E_ARG;
    ctx->opts.set[ix] = false; ctx->opts.v[ix] = 0;
    return SNK_OK;
}
extern "C" int snk_ctx_get_option(const snk_ctx* ctx, const char* name, long long* value) {
    if (!ctx) return SNK_E_ARG;
    const int ix = snk_opt_index(name);
    if (ix < 0) return SNK_E_ARG;
    if (value) *value = ctx->opts.v[ix];
    return ctx->opts.set[ix] ? 1 : 0;
}
extern "C" const char* snk_option_name(uint32_t i) { return i < (uint32_t)snk_opt_count() ? snk_opt_defs[i].name : nullptr; }
extern "C" const char* snk_option_doc(uint32_t i) { return i < (uint32_t)snk_opt_count() ? snk_opt_defs[i].doc : nullptr; }

// the documented knobs as one struct: 0 in a field = the library's own choice (the option is cleared)
namespace {
struct tfield { const char* opt; size_t off; };
#define TF(f, o) {o, offsetof(snk_tuning, f)}
const tfield tfields[] = {
    TF(count_screen_ratio_pct, "screen_ratio_pct"), TF(target_inst, "target_inst"), TF(bucket_fill_pct, "bucket_fill_pct"),
    TF(minimiser_len, "minimiser_len"), TF(partition_passes, "partition_passes"), TF(hot_min, "hot_min"), TF(hot_factor, "hot_factor"),
    TF(hot_class_inst, "hot_class_inst"), TF(exchange_ranges, "exchange_ranges"), TF(hbv_dev_min, "hbv_dev_min"), TF(hbv_big, "hbv_big"),
    TF(chunk_kmers, "chunk_kmers"), TF(unitig_bc_cut, "unitig_bc_cut"),
};
#undef TF
}  // namespace

extern "C" void snk_tuning_default(snk_tuning* t) { if (t) memset(t, 0, sizeof *t); }

extern "C" int snk_ctx_set_tuning(snk_ctx* ctx, const snk_tuning* t, char* err, size_t errcap) {
    if (!ctx || !t) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_tuning: NULL argument");
    if (t->count_kernel > SNK_COUNT_KERNEL_SCREEN) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_tuning: count_kernel %u", t->count_kernel);
    if (t->minimiser_len && t->minimiser_len != 16 && t->minimiser_len != 20) return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_tuning: minimiser_len is 0, 16 or 20");
    if (t->path_lookup > 2 || t->join_ranking > 2 || t->adaptive_buckets > 2 || t->hot_buckets > 2)
        return snk_fail(SNK_E_ARG, err, errcap, "snk_ctx_set_tuning: a 0/1/2 field is out of range");
    auto put = [&](const char* o, bool on, long long v) { const int ix = snk_opt_index(o); ctx->opts.set[ix] = on; ctx->opts.v[ix] = on ? v : 0; };
    for (const tfield& f : tfields) { const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t) + f.off); put(f.opt, v != 0, v); }
    // count kernel: auto | margin (the default kernel) | booked slots | bit filter + booked slots
    const uint32_t slots = t->count_tight_slots ? t->count_tight_slots : 1920u;
    switch (t->count_kernel) {
        case SNK_COUNT_KERNEL_AUTO: put("count_tight", false, 0); put("count_screen_ng", false, 0); break;
        case SNK_COUNT_KERNEL_MARGIN: put("count_tight", true, 0); put("count_screen_ng", true, 0); break;
        case SNK_COUNT_KERNEL_BOOKED: put("count_tight", true, slots); put("count_screen_ng", true, 0); break;
        case SNK_COUNT_KERNEL_SCREEN: put("count_tight", false, 0); put("count_screen_ng", true, 2); break;
    }
    put("adaptive_buckets", t->adaptive_buckets != 0, t->adaptive_buckets == 1 ? 1 : 0);       // 1 on, 2 off
    put("hot", t->hot_buckets != 0, t->hot_buckets == 1 ? 1 : 0);
    put("path_index", t->path_lookup != 0, t->path_lookup == 1 ? 1 : 0);                        // 1 index, 2 dictionary
    put("join_replicated", t->join_ranking != 0, t->join_ranking == 2 ? 1 : 0);                 // 1 partitioned, 2 replicated
    return SNK_OK;
}

extern "C" void snk_ctx_get_tuning(const snk_ctx* ctx, snk_tuning* t) {
    if (!ctx || !t) return;
    memset(t, 0, sizeof *t);
    auto get = [&](const char* o, long long* v) { const int ix = snk_opt_index(o); *v = ctx->opts.v[ix]; return ctx->opts.set[ix]; };
    long long v;
    for (const tfield& f : tfields) if (get(f.opt, &v)) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(t) + f.off) = (uint32_t)v;
    long long tight = 0, ng = 0;
    const bool ts = get("count_tight", &tight), ns = get("count_screen_ng", &ng);
    if (ns && ng >= 2) t->count_kernel = SNK_COUNT_KERNEL_SCREEN;
    else if (ts && tight) { t->count_kernel = SNK_COUNT_KERNEL_BOOKED; t->count_tight_slots = (uint32_t)tight; }
    else if (ts) t->count_kernel = SNK_COUNT_KERNEL_MARGIN;
    if (get("adaptive_buckets", &v)) t->adaptive_buckets = v ? 1 : 2;
    if (get("hot", &v)) t->hot_buckets = v ? 1 : 2;
    if (get("path_index", &v)) t->path_lookup = v ? 1 : 2;
    if (get("join_replicated", &v)) t->join_ranking = v ? 2 : 1;
    // what the last call on the context chose
    t->last_count_limit = ctx->last_count_limit;
    t->last_count_kernel = ctx->count_screen ? SNK_COUNT_KERNEL_SCREEN : (ctx->count_tight ? SNK_COUNT_KERNEL_BOOKED : SNK_COUNT_KERNEL_MARGIN);
    t->last_partition_passes = ctx->last_partition_passes;
    t->last_minimiser_len = ctx->mlen;
}
