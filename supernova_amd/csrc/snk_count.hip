// snk_count.hip -- K5..K8: supermers of one minimiser bucket -> canonical k-mers -> counts, barcode
// rule, OR of contexts -> filtered table.  One workgroup per bucket, the whole reduce in LDS.
//
// What it replaces:
//   Kmerizer::map           lib/assembly/src/paths/long/BuildReadQGraph48.cc:155-172 (roll, canonicalise, context)
//   MapReduceEngine reduce  lib/assembly/src/MapReduceEngine.h:574-584 (std::sort + run detection)
//   summarizeEntries        BuildReadQGraph48.cc:92-105   (sum of counts, OR of contexts)
//   areIgnoredBarcodes / areEnoughBarcodes  :108-137       (>= minBC distinct barcodes > 0, or any bc == -1)
//   Kmerizer::reduce        :174-181                       (keep iff count >= minFreq && bc test)
//   == process_kmer_shard_opt, lib/tada/src/utils.rs:322-408 (sort + group_by per shard).
//
// The reference materialises one 24-byte record per k-mer instance and comparison-sorts them.  Here
// the instances never leave the CU: a bucket's supermers (32 B per ~15 k-mers) come from HBM once, by LDS-DMA and one bucket
// ahead; identical supermers are folded; every remaining k-mer instance is extracted by its own lane, canonicalised, hashed
// and inserted into an open-addressing hash table in LDS (key 96/120 bit, count, barcode state, context byte).  A workgroup
// walks a strided list of buckets and writes the survivors of each behind a cursor in its own output region.  A bucket whose
// distinct k-mers do not fit is re-run split by a second hash (2, 4, ... passes) -- correctness never depends on the bucket
// sizing.  Barcode state machine (minBC <= 2): 0 = none yet, id = exactly one barcode id seen, MULTI = two different ids,
// IGN = a bc == -1 read contributed; minBC 3..8: six more id words per slot.
// The kernel is bound by VALU issue in its insert phase; DESIGN.md 4 ("round 2") says what was tried around that.
#include "snk_ctx.h"
#include "snk_common.h"
#include "snk_kernels.h"
#include "snk_stages.h"

namespace {

constexpr uint32_t BC_MULTI = 0xFFFFFFFEu;
constexpr uint32_t BC_IGN = 0xFFFFFFFFu;      // (a read under the ignore rule: larger than BC_MULTI, so the atomicMax of the merges keeps it)
static_assert(BC_IGN > BC_MULTI, "barcode states are merged with atomicMax");
#ifndef SNK_TIGHT_TRIES
#define SNK_TIGHT_TRIES 48
#endif
constexpr int MAX_SPLIT_LOG2 = 20;      // (split ids are 24-bit fields; the split stack in ctl[] holds 24 entries)
#define SNK_COUNT_MAXSEG 32      // record segments per bucket (sharded runs: one per source rank)
#ifndef SNK_COUNT_THREADS
#define SNK_COUNT_THREADS 768
#endif
// ---- build parameters of the SCREEN instantiations (tuning builds: tools/build_variant.sh)
#ifndef SNK_SCREEN_ROUNDS
#define SNK_SCREEN_ROUNDS 6
#endif
#ifndef SNK_GSCREEN_SLOTS
#define SNK_GSCREEN_SLOTS 1024          // table slots of the grouped SCREEN instantiation: the table sees a tenth of the instances, half of it pays for 512-record batches (2048: 256-record batches, 104.6 instead of 92.2 ms)
#endif
#ifndef SNK_GSCREEN_BATCH
#define SNK_GSCREEN_BATCH 512
#endif
#ifndef SNK_GSCREEN_ROUNDS
#define SNK_GSCREEN_ROUNDS (SNK_SCREEN_ROUNDS + 4)
#endif
#ifndef SNK_NGSCREEN_PLANE_WORDS
#define SNK_NGSCREEN_PLANE_WORDS 512
#endif
#ifndef SNK_COUNT_SLOTS
#define SNK_COUNT_SLOTS 2048
#endif
#define LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// Workgroup barrier that orders LDS traffic only.  Inside this kernel the threads of a workgroup talk to each other through
// LDS exclusively; __syncthreads() also drains every outstanding HBM access of the wave (s_waitcnt vmcnt(0)): the prefetched
// records of the next bucket, the survivors' stores.  Loaded values are still waited for where they are used (the compiler's
// own s_waitcnt).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef SNK_COUNT_PROF
// thread 0's cycles per phase, accumulated in registers and added to the global counters once per workgroup
#define PROF(n) do { if (tid == 0) { const long long _t = clock64(); prof_acc[n] += (unsigned long long)(_t - prof_t); prof_t = _t; } } while (0)
#else
#define PROF(n) do {} while (0)
#endif

// high word of (hi:lo) << sh, sh in 0..31
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) {
#ifdef SNK_FUNNEL_ALIGNBIT
    return sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
#else
    return (uint32_t)(((((uint64_t)hi << 32) | lo) << sh) >> 32);
#endif
}

template <int K> struct lo_t { typedef uint32_t type; };       // K<=48: only the top 32 bits of lo are used
template <> struct lo_t<60> { typedef uint64_t type; };

template <int K> __device__ __forceinline__ typename lo_t<K>::type lo_pack(uint64_t lo);
template <> __device__ __forceinline__ uint32_t lo_pack<48>(uint64_t lo) { return (uint32_t)(lo >> 32); }
template <> __device__ __forceinline__ uint64_t lo_pack<60>(uint64_t lo) { return lo; }
template <int K> __device__ __forceinline__ uint64_t lo_unpack(typename lo_t<K>::type v);
template <> __device__ __forceinline__ uint64_t lo_unpack<48>(uint32_t v) { return (uint64_t)v << 32; }
template <> __device__ __forceinline__ uint64_t lo_unpack<60>(uint64_t v) { return v; }

template <int K, bool G> struct klo_t { typedef typename lo_t<K>::type type; };
template <int K> struct klo_t<K, true> { typedef uint64_t type; };      // grouped: the low 32 bits carry the group id
template <int K, bool G> __device__ __forceinline__ typename klo_t<K, G>::type klo_pack(uint64_t lo) {
    if constexpr (G) return lo; else return lo_pack<K>(lo);
}
template <int K, bool G> __device__ __forceinline__ uint64_t klo_unpack(typename klo_t<K, G>::type v) {
    if constexpr (G) return v; else return lo_unpack<K>(v);
}

typedef __attribute__((address_space(3))) void* snk_lptr;

// MULTI: more than one record segment per bucket.  GATHER (dense partition, snk_stages.hip): a bucket is a range of an index list, record v of
// the bucket is records[gidx[v]] -- the LDS-DMA fetch takes a per-lane address, so a gathered batch costs what a contiguous one does
// plus the (coalesced) read of its indices.
// SCREEN (per-barcode groups, a.screen = min(min_freq, 3) >= 2): inside one barcode a locus is read once or twice -- nine instances in ten are
// the only one of their (group, k-mer) and every one of them is a claim, the expensive path, that the filter throws away.  A one-batch bucket
// of up to SNK_SCREEN_ROUNDS rounds first runs all its instances through a three-level bit filter (three planes of 8192 bits in the memory of the
// de-duplication table and the weights, which grouped runs do not use: an instance sets its cell's bit in the first plane that does not have it yet), remembers each
// instance's cell in registers (two per register), and then inserts only the instances whose cell reached the level -- compacted into a list first, so that
// the insert rounds run with full waves.  A k-mer whose own instances do not reach min_freq cannot be retained; what shares a cell only
// over-admits (5 % at 3000 instances); the table still counts what is admitted and the filter still decides.  Buckets of more than one batch,
// of more rounds, or with more candidates than the list holds are counted as before.
template <int K, int THREADS, int SLOTS, bool GROUPED, bool MULTI, bool GATHER = false, bool TIGHT = false, bool SCREEN = false>
__global__ void __launch_bounds__(THREADS, 6) snk_count_kernel(snk_count_args a) {
    // supermers staged per batch.  A 4000-instance bucket holds ~270 (sigma ~100): with 512 slots nearly every bucket is
    // ONE batch (with 256, 55 % of the buckets ran a second, mostly empty batch through all the phases below).  K=60 and
    // grouped runs have 64-bit low key words: 256 keeps the workgroup under 80 KB of LDS, i.e. two per CU.
    // (grouped SCREEN with a 1024-slot table -- the table only sees a tenth of the instances -- has the room for 512-record batches as well)
    constexpr int BATCH = (GROUPED && SCREEN && SLOTS <= 1024 && K == 48) ? SNK_GSCREEN_BATCH : ((K == 48 && !GROUPED && (SLOTS >= 2048 || SCREEN)) ? 512 : 256);
    constexpr int DD = 2 * BATCH;                        // de-duplication table slots
    typedef typename klo_t<K, GROUPED>::type lo_type;
    constexpr int WMAX = K - SNK_M_MIN_OF(K) + 1;                  // k-mers per supermer, at most
    constexpr int NCI = BATCH * WMAX / 32 + 2;           // coarse instance index: one entry per 32 k-mer instances
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* khi = reinterpret_cast<uint64_t*>(smem_raw);                         // [SLOTS]
    lo_type* klo = reinterpret_cast<lo_type*>(khi + SLOTS);                         // [SLOTS]
    uint32_t* tag = reinterpret_cast<uint32_t*>(klo + SLOTS);                       // [SLOTS] 0 empty | fingerprint(31) | ready(1)
    uint32_t* cnt = tag + SLOTS;                                                    // [SLOTS] observations
    uint32_t* bcs = cnt + SLOTS;                                                    // [SLOTS] barcode state
    uint32_t* ctxw = bcs + SLOTS;                                                   // [SLOTS/4] context bytes (a claimer stores its byte, later observations OR into the word)
    // [BATCH][8] the staged supermer records as they are in HBM (32 bytes each; word 6 = bases | k-mers, flank flags; word 7 =
    // barcode state / group id).  They are not copied through registers: the LDS-DMA load of gfx950 (global_load_lds_dwordx4)
    // puts the 16 bytes lane l asked for at (wave's LDS base) + 16 l -- two lanes per record, consecutive lanes = consecutive
    // addresses on both sides.  16-byte aligned (every array before it is a multiple of 16 bytes).
    uint32_t* rec = ctxw + SLOTS / 4;
    uint32_t* ctl = rec + 8 * BATCH;                                                // [64] control words
    uint32_t* dd = ctl + (TIGHT ? 72 : 64);                                                        // [DD] supermer de-duplication table (leader index + 1)
    uint32_t* wgt = dd + DD;                                                        // [BATCH] copies folded into each leader
    uint16_t* lead = reinterpret_cast<uint16_t*>(wgt + BATCH);                      // [BATCH] r-th leading (non-folded) supermer
    uint16_t* lpre = lead + BATCH;                                                  // [BATCH+2] one past its last k-mer instance
    uint16_t* cidx = lpre + BATCH + 2;                                              // [NCI] leader rank that owns instance 32*w
    uint32_t* segi = reinterpret_cast<uint32_t*>(smem_raw + (((size_t)(reinterpret_cast<unsigned char*>(cidx + NCI) - smem_raw) + 15) & ~(size_t)15));   // [2][3][MAXSEG] segment start (lo, hi), end (lo; a segment holds < 2^32 records); one copy per bucket parity
    uint16_t* olist = reinterpret_cast<uint16_t*>(segi + 2 * 3 * SNK_COUNT_MAXSEG);  // [LIMIT] claimed slots in claim order (the filter walks these, not the table)
    // minBC > 2 (areEnoughBarcodes counts DISTINCT barcodes for any minBC, BuildReadQGraph48.cc:117-137): six more barcode ids
    // per slot behind the one in bcs[] -- up to seven distinct ids are told apart exactly, an eighth turns the state into MULTI
    // (so min_bc <= 8).  Only allocated for such runs (a.bc_mode > 2); they give up the second workgroup per CU.
    uint32_t* bcx = reinterpret_cast<uint32_t*>(smem_raw + (((size_t)(reinterpret_cast<unsigned char*>(olist + (TIGHT ? SLOTS - 64 : SLOTS - THREADS - 64)) - smem_raw) + 15) & ~(size_t)15));   // [SLOTS][6]
    const bool bcset = a.bc_mode > 2;
    constexpr int SROUNDS = (SCREEN && SLOTS <= 1024) ? SNK_GSCREEN_ROUNDS : SNK_SCREEN_ROUNDS;                        // SCREEN: instances per lane (their cells ride in two registers)
    static_assert(SROUNDS >= 1 && SROUNDS <= 16, "two cells per register");
    // words per bit plane.  Grouped runs: the de-duplication table and the weights hold the three planes (neither is used there); ungrouped
    // runs (error-rich reads; both are in use): their own array behind the candidate list
    constexpr int SPW = GROUPED ? ((DD + BATCH) / 3 >= 512 ? 512 : 256) : SNK_NGSCREEN_PLANE_WORDS;
    constexpr uint32_t SCB = SPW == 1024 ? 17u : (SPW == 512 ? 18u : 19u);                // cell = the top 15 / 14 / 13 bits of h1
    // SCREEN: candidate list (instance indices), in bcx's place (SCREEN runs keep up to two barcodes per slot: no bcx)
    constexpr uint32_t SCAND = GROUPED ? (SLOTS <= 1024 ? 2048 : 1024) : 4096;
    uint16_t* cand = reinterpret_cast<uint16_t*>(bcx);
    uint32_t* planes = GROUPED ? dd : reinterpret_cast<uint32_t*>(cand + SCAND);
    static_assert(!SCREEN || (GROUPED && 3 * SPW <= DD + BATCH) || (!GROUPED && K == 48 && SLOTS <= 1024), "no room for the screen's bit planes");
    // ctl[1..2] occupied slots (by pass parity; more than LIMIT = the pass overflows), ctl[3] most slots used so far,
    // ctl[4..7] / ctl[12..15] record bounds of the bucket (segment 0; by bucket parity), ctl[8..9] placement counter (by pass
    // parity), ctl[10..11] instances | leaders << 16 of the batch (by batch parity), ctl[16..16+2*MAX) split stack (17 levels)
    static_assert(THREADS >= BATCH && THREADS % 64 == 0, "the first BATCH threads own the staged records");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // distinct k-mers one sub-pass may hold.  TIGHT: a wave books a slot for every lane that is about to probe (ctl[64..65], by pass
    // parity; bit 31 = the pass is over capacity) and hands back what its lanes did not claim, so the table fills to 7/8 and the probe loops
    // still always find a free slot; otherwise nobody books anything and the margin is one round of every wave (below)
    const uint32_t LIMIT = TIGHT ? min(a.tight & 0xFFFFu, (uint32_t)SLOTS - 64u) : SLOTS - THREADS - 64;
    const int tight_tries = (int)(a.tight >> 16);
#ifdef SNK_COUNT_PROF
    long long prof_t = clock64();
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Fetch one batch of records (virtual indices base .. base+BATCH-1 of a bucket whose bounds are in LDS: bnd = segment 0's
    // start and end, seg = all segments') into rec[] by LDS-DMA.  Nothing is waited for here.
    auto dma_batch = [&](uint64_t base, uint64_t vend, const uint32_t* bnd, const uint32_t* seg) {
        constexpr int CH = 2 * BATCH;                     // 16-byte chunks
        uint32_t seg_incl = 0;
        if (MULTI) {
            const uint32_t mylen = (uint32_t)lane < a.nseg ? LDS_LOAD(&seg[2 * SNK_COUNT_MAXSEG + lane]) - LDS_LOAD(&seg[lane]) : 0u;
            seg_incl = snk_wave_scan_incl(mylen);
        }
        // GATHER: every index of the batch is asked for before the first fetch goes out (vmcnt counts in order: an index load behind
        // a fetch would wait for that fetch to land)
        constexpr int NR = (CH + THREADS - 1) / THREADS;
        uint32_t gx[NR];
        if (GATHER) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int c = r * THREADS + tid;
                const uint64_t v = base + (uint32_t)(c >> 1);
                gx[r] = (c < CH && v < vend) ? a.gidx[v] : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r * THREADS < CH; ++r) {
            const int c = r * THREADS + tid;
            const uint64_t v = base + (uint32_t)(c >> 1);
            if (c < CH && v < vend) {
                uint64_t gi;
                if (GATHER) gi = gx[r];
                else if (MULTI) {
                    // the segments are read as ONE concatenated record stream (batches stay full, identical supermers from
                    // different sources fold): find the segment by a short prefix walk
                    // the segment of virtual record x: the inclusive prefix of the segment lengths sits in the wave's first lanes
                    // (seg_incl, below) -- one scalar read per segment and a compare, no chain of dependent LDS reads (the walk
                    // over eight segments was +5 ms on the kernel: its latency delays every batch's fetch)
                    const uint32_t x = (uint32_t)v;
                    uint32_t acc = 0, sg = 0;
                    for (uint32_t q = 0; q + 1 < a.nseg; ++q) { const uint32_t e = __builtin_amdgcn_readlane(seg_incl, q); if (x >= e) { sg = q + 1; acc = e; } }
                    gi = (((uint64_t)seg[SNK_COUNT_MAXSEG + sg] << 32) | seg[sg]) + (x - acc);
                } else gi = v;
                (void)bnd;
                // (inline assembly on purpose: behind the builtin the compiler drains vmcnt before the next LDS read, whatever
                // it reads -- the fetch would be waited for right where it is issued)
                const uint32_t lds_at = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(snk_lptr)(rec + 4 * (r * THREADS + (tid & ~63))));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_at), "v"(a.records + gi * 2 + (uint32_t)(c & 1)) : "memory");
            }
        }
    };
    // iteration space of a bucket: absolute record indices of its one segment, or offsets into the concatenation of all segments
    auto bucket_range = [&](const uint32_t* bnd, const uint32_t* seg, uint64_t& vbeg, uint64_t& vend) {
        if (MULTI) {
            // (sum of the segment lengths: one segment per lane and a wave scan, not a loop of 2 nseg LDS reads in every thread)
            const uint32_t mylen = (uint32_t)lane < a.nseg ? LDS_LOAD(&seg[2 * SNK_COUNT_MAXSEG + lane]) - LDS_LOAD(&seg[lane]) : 0u;
            vbeg = 0;
            vend = (uint32_t)__builtin_amdgcn_readlane((int)snk_wave_scan_incl(mylen), 63);
        }
        else {
            vbeg = ((uint64_t)LDS_LOAD(&bnd[1]) << 32) | LDS_LOAD(&bnd[0]);
            vend = ((uint64_t)LDS_LOAD(&bnd[3]) << 32) | LDS_LOAD(&bnd[2]);
        }
    };
    // The grid is a few residency waves of workgroups; workgroup w counts the buckets b == w (mod grid) of [bucket0, NB)
    // (no dispatch gap between buckets) and OWNS output region w: its survivors go out behind each other at a cursor the
    // workgroup keeps itself.  The first version reserved every sub-pass's chunk with a device-scope atomic on one of 4096
    // region cursors: a round trip of microseconds per bucket that all twelve waves waited for at a barrier.  A region is
    // continued across launches (the ranged launches of the sharded path): the cursor is read at the start, written at the end.
    const uint32_t G = a.bucket_stride;
    unsigned long long rcur = a.region_cursor[blockIdx.x];          // uniform: every thread keeps the same value
    uint32_t bucket = a.bucket0 + (blockIdx.x + G - a.bucket0 % G) % G;
    // The bounds of a bucket travel through LDS, one bucket ahead: they are loaded at the top of the bucket before, put down
    // (ctl[4..7] / ctl[12..15] and the two copies of segi, by bucket parity) once that bucket's first batch is staged, and read
    // when its last insert phase is over -- which is where the next bucket's first batch is fetched, behind the survivors'
    // stores, so that neither the bounds nor the records cost an HBM latency at the top of a bucket.
    uint32_t par = 0;
    {
        uint64_t b0 = 0, e0 = 0;
        if (tid == 0) {
            if (bucket < a.NB) { b0 = a.seg_beg[bucket]; e0 = a.seg_end[bucket]; }
            ctl[4] = (uint32_t)b0; ctl[5] = (uint32_t)(b0 >> 32); ctl[6] = (uint32_t)e0; ctl[7] = (uint32_t)(e0 >> 32);
            ctl[3] = 0;
            ctl[1] = 0; ctl[2] = 0; ctl[8] = 0; ctl[9] = 0; ctl[10] = 0; ctl[11] = 0; ctl[0] = 0;      // (ctl[0]: SCREEN's candidate counter)
            if (TIGHT) { ctl[64] = 0; ctl[65] = 0; }
        }
        for (int s = tid; s < SLOTS; s += THREADS) tag[s] = 0;      // from here on a pass leaves the table empty behind it
        if (MULTI && tid < (int)a.nseg) {
            uint64_t b = 0, e = 0;
            if (bucket < a.NB) { b = a.seg_beg[(uint64_t)tid * a.seg_stride + bucket]; e = a.seg_end[(uint64_t)tid * a.seg_stride + bucket]; }
            segi[tid] = (uint32_t)b; segi[SNK_COUNT_MAXSEG + tid] = (uint32_t)(b >> 32); segi[2 * SNK_COUNT_MAXSEG + tid] = (uint32_t)e;
        }
        lds_barrier();
        uint64_t vb, ve;
        bucket_range(ctl + 4, segi, vb, ve);
        dma_batch(vb, ve, ctl + 4, segi);
    }
    bool prefetched = true;         // rec[] holds (or is about to hold) the first batch of the bucket at hand
    // Four barriers per pass (staged / mapped / inserted / written), none between passes: the counters a pass uses (occupancy,
    // placement; the batch's instance scan) exist twice and the copy the NEXT pass / batch will use is zeroed behind the
    // 'staged' barrier of this one, when every wave is past its last look at it; the table is emptied by the lanes that write
    // its entries out.
    uint32_t q = 0, bq = 0;         // parity of the pass / of the batch
    unsigned long long claims_sum = 0;
    for (; bucket < a.NB; bucket += G) {
    // depth of the split stack: every thread keeps its own copy (the control flow is uniform), so the sub-pass loop needs
    // no barrier-protected LDS read to decide whether it is done
    uint32_t sp = 1;
    bool fresh = true;              // the first pass of a bucket: nothing on the split stack
    uint32_t splits_done = 0;
    const uint32_t nbucket = bucket + G;
    bool seg_pending = true;        // the next bucket's bounds are not in LDS yet
    const uint32_t* bcur = ctl + 4 + 8 * par;
    uint32_t* bnext = ctl + 4 + 8 * (par ^ 1u);
    const uint32_t* segc = segi + par * (3 * SNK_COUNT_MAXSEG);          // this bucket's copy
    uint32_t* segn = segi + (par ^ 1u) * (3 * SNK_COUNT_MAXSEG);        // the next bucket's
    // Every lane loads them (the same address in all lanes is one transaction per wave) and every lane puts them down (the same
    // value to the same word): no divergent branch around either side, so the one wait the compiler needs sits in front of the
    // stores and nowhere else (a load that is consumed on some paths only is waited for at the next loop header).
    const uint32_t sseg = MULTI ? min((uint32_t)tid, a.nseg - 1u) : 0u;
    auto load_next = [&](uint64_t& nbeg, uint64_t& nend, uint64_t& nsb, uint32_t& nse) {
        nbeg = 0; nend = 0; nsb = 0; nse = 0;
        if (nbucket < a.NB) {
            nbeg = a.seg_beg[nbucket]; nend = a.seg_end[nbucket];
            if (MULTI) { nsb = a.seg_beg[(uint64_t)sseg * a.seg_stride + nbucket]; nse = (uint32_t)a.seg_end[(uint64_t)sseg * a.seg_stride + nbucket]; }
        }
    };
    auto put_down = [&](uint64_t nbeg, uint64_t nend, uint64_t nsb, uint32_t nse) {
        bnext[0] = (uint32_t)nbeg; bnext[1] = (uint32_t)(nbeg >> 32); bnext[2] = (uint32_t)nend; bnext[3] = (uint32_t)(nend >> 32);
        if (MULTI) { segn[sseg] = (uint32_t)nsb; segn[SNK_COUNT_MAXSEG + sseg] = (uint32_t)(nsb >> 32); segn[2 * SNK_COUNT_MAXSEG + sseg] = nse; }
        seg_pending = false;
    };
    // virtual buckets (snk_hot.hip): the k-mer instances of one hash class of a hot minimiser bucket -- the pass starts in that class
    // (further splits add hash bits above it) and its chunks are reported under the real bucket
    uint32_t lg0 = 0, id0 = 0, real_bucket = bucket;
    if (a.vmeta) {
        const uint2 vm = a.vmeta[bucket];
        real_bucket = __builtin_amdgcn_readfirstlane(vm.x);
        lg0 = __builtin_amdgcn_readfirstlane(vm.y >> 24);
        id0 = __builtin_amdgcn_readfirstlane(vm.y & 0xFFFFFFu);
    }
    while (sp) {
        PROF(0);
        --sp;
        uint32_t split_lg = lg0, split_id = id0;
        if (!fresh) {
            lds_barrier();      // the stack entries and the cleared table of the pass that overflowed are visible
            split_lg = LDS_LOAD(&ctl[16 + 2 * sp]); split_id = LDS_LOAD(&ctl[17 + 2 * sp]);
        }
        fresh = false;
        const uint32_t split_mask = (1u << split_lg) - 1u;
        uint32_t* occ = ctl + 1 + q;            // distinct k-mers of this pass
        uint32_t* place = ctl + 8 + q;          // its survivors
        uint32_t* resv = ctl + 64 + q;          // (TIGHT) slots booked: claimed + asked for by the rounds in flight
        uint64_t vbeg, vend;
        bucket_range(bcur, segc, vbeg, vend);
        if (vbeg >= vend) {                     // a bucket without records: nothing is staged, what happens behind 'staged' happens here
            lds_barrier();
            if (tid == 0) { ctl[1 + (q ^ 1u)] = 0; ctl[8 + (q ^ 1u)] = 0; if (TIGHT) ctl[64 + (q ^ 1u)] = 0; }
            if (seg_pending) {
                uint64_t nbeg, nend, nsb; uint32_t nse;
                load_next(nbeg, nend, nsb, nse);
                put_down(nbeg, nend, nsb, nse);
            }
            lds_barrier();
        }
        PROF(1);
        {
            for (uint64_t base = vbeg; base < vend; base += BATCH) {
                // ---- stage one batch of supermer records
                if (!prefetched) dma_batch(base, vend, bcur, segc);
                prefetched = false;
                if (tid < BATCH) wgt[tid] = (SCREEN && GROUPED) ? 0u : 1u;        // (grouped SCREEN: the third bit plane; nothing folds in grouped runs, every weight is 1)
                if (SCREEN && !GROUPED) { for (int qq = tid; qq < 3 * SPW; qq += THREADS) planes[qq] = 0; }
                for (int q = tid; q < DD; q += THREADS) dd[q] = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my share of the batch is in LDS (and everything older has landed)
                lds_barrier();                                       // 'staged'
                PROF(2);
                if (tid == 0) { ctl[1 + (q ^ 1u)] = 0; ctl[8 + (q ^ 1u)] = 0; ctl[10 + (bq ^ 1u)] = 0; if (TIGHT) ctl[64 + (q ^ 1u)] = 0; if (SCREEN) ctl[0] = 0; }
                // the next bucket's bounds: asked for here (behind the wait for the batch -- vmcnt counts in order), put down
                // before the insert phase, read when it is over
                const bool fetch_next = seg_pending;
                uint64_t nbeg = 0, nend = 0, nsb = 0; uint32_t nse = 0;
                if (fetch_next) load_next(nbeg, nend, nsb, nse);
                // ---- fold identical supermers: at 56x coverage ~3 of 4 reads over a locus yield the SAME record (same
                //      bases, same flanks); only their barcodes differ.  The first one becomes the leader and carries a
                //      weight and a merged barcode state, the copies insert nothing (2.5x fewer k-mer insertions).
                uint32_t nkm = 0;
                if (tid < BATCH && base + tid < vend) {
                    const uint4 ra = *reinterpret_cast<const uint4*>(rec + 8 * tid), rb = *reinterpret_cast<const uint4*>(rec + 8 * tid + 4);
                    nkm = rb.z & 0x7Fu;
                    if (a.dbg == 5 && ((rb.z >> 7) & 3u) != 3u) nkm = 0;      // probe (results invalid): what the records of read starts / ends cost (DESIGN 4 "round 4")
                    // (per-barcode groups: a group sees a locus once or twice, identical supermers inside one group are rare -- looking for
                    // them cost 11 of the grouped bench's 160 ms and found next to nothing)
                    if (!GROUPED && a.dbg != 4 && nkm) {
                        // (seven independent multiplies, one finaliser: this lane is on the critical path of the batch)
                        uint32_t h = (ra.x * 0x9E3779B1u) ^ snk_rotl32(ra.y * 0x85EBCA77u, 7) ^ (ra.z * 0xC2B2AE3Du) ^ snk_rotl32(ra.w * 0x27D4EB2Fu, 13)
                                   ^ (rb.x * 0x165667B1u) ^ snk_rotl32(rb.y * 0xcc9e2d51u, 19) ^ (rb.z * 0x1b873593u);
                        h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 13;
                        uint32_t s = h & (DD - 1);
                        for (;;) {
                            uint32_t v = LDS_LOAD(&dd[s]);
                            if (v == 0) {
                                v = atomicCAS(&dd[s], 0u, (uint32_t)tid + 1u);
                                if (v == 0) break;                               // I lead this record
                            }
                            const uint32_t L = v - 1u;
                            const uint4 la = *reinterpret_cast<const uint4*>(rec + 8 * L);
                            const uint32_t lb0 = rec[8 * L + 4], lb1 = rec[8 * L + 5], lb2 = rec[8 * L + 6];     // (word 7 of a leader changes under us)
                            bool same = ((la.x ^ ra.x) | (la.y ^ ra.y) | (la.z ^ ra.z) | (la.w ^ ra.w) | (lb0 ^ rb.x) | (lb1 ^ rb.y) | (lb2 ^ rb.z)) == 0;
                            if (GROUPED || bcset) same = same && LDS_LOAD(&rec[8 * L + 7]) == rb.w;     // same bases in another group: not a copy; minBC > 2: a folded supermer could not say how many barcodes it stands for
                            if (same) {
                                atomicAdd(&wgt[L], 1u);
                                const uint32_t mine = GROUPED ? 0u : rb.w;
                                if (mine >= BC_MULTI) atomicMax(&rec[8 * L + 7], mine);
                                else if (mine) {
                                    uint32_t ob = atomicCAS(&rec[8 * L + 7], 0u, mine);
                                    if (ob != 0 && ob != mine && ob < BC_MULTI) atomicMax(&rec[8 * L + 7], BC_MULTI);
                                }
                                nkm = 0;
                                break;
                            }
                            s = (s + 1) & (DD - 1);
                        }
                    }
                }
                // one packed scan over the supermers: k-mer instances (low 16 bits) and leaders (high 16 bits); a wave takes its
                // base from a counter (no barrier, no wave order: an instance range and a leader rank come from the same atomic,
                // so they are ordered alike)
                const uint32_t sv = nkm | (nkm ? 0x10000u : 0u);
                uint32_t incl = sv;
                incl = snk_wave_scan_incl(incl);
                uint32_t woff = 0;
                if (lane == 63 && incl) woff = atomicAdd(&ctl[10 + bq], incl);
                woff = __builtin_amdgcn_readlane(woff, 63);
                if (tid < BATCH && nkm) {
                    // instance -> supermer map without a byte per instance: leaders in order, their first instance, and
                    // for every 32nd instance the leader that owns it (a lane then walks 0-3 leaders forward)
                    const uint32_t ex = woff + incl - sv;
                    const uint32_t off = ex & 0xFFFFu, r = ex >> 16;
                    lead[r] = (uint16_t)tid;
                    lpre[r] = (uint16_t)(off + nkm);                 // one past its last instance
                    for (uint32_t w = (off + 31u) >> 5; (w << 5) < off + nkm; ++w) cidx[w] = (uint16_t)r;
                }
                lds_barrier();                                       // 'mapped'
                PROF(4);
                const uint32_t tot = LDS_LOAD(&ctl[10 + bq]);
                const uint32_t total = tot & 0xFFFFu;
                if (fetch_next) put_down(nbeg, nend, nsb, nse);
                // ---- one lane per k-mer instance: a wave inserts 64 different k-mers of consecutive supermers, so
                //      copies of the same k-mer (identical supermers of other reads) are spread over time, not lanes.
                // A sub-pass overflows when it holds more than LIMIT distinct k-mers.  Nobody polls a flag while probing: a wave
                // looks at the occupancy before each round of 64 insertions and stops above LIMIT, so at most THREADS claims
                // can follow the one that crossed the line -- LIMIT + THREADS < SLOTS, the probe loops always find a free slot.
                bool screened = false;
                uint32_t n_iter = total;
                if constexpr (SCREEN) {
                    // (uniform) the whole bucket is this batch, and every lane has at most three instances
                    screened = a.screen >= 2u && vend - vbeg <= (uint64_t)BATCH && total <= (uint32_t)SROUNDS * THREADS;
                    if (screened) {
                        uint32_t cpk[(SROUNDS + 1) / 2], inpass = 0;
#pragma unroll
                        for (int r = 0; r < (SROUNDS + 1) / 2; ++r) cpk[r] = 0;
#pragma unroll
                        for (uint32_t r = 0; r < (uint32_t)SROUNDS; ++r) {
                            const uint32_t g = r * THREADS + tid;
                            if (g < total) {
                                uint32_t lr = cidx[g >> 5];
                                uint32_t lend = lpre[lr];
                                while (lend <= g) lend = lpre[++lr];
                                const uint32_t i = lead[lr];
                                const uint32_t* rp = rec + 8 * i;
                                const uint32_t m6 = rp[6];
                                const uint32_t n_i = m6 & 0x7Fu, hasL = (m6 >> 7) & 1u;
                                const uint32_t o = hasL + (g + n_i - lend);
                                const uint32_t wi = o >> 4, sh = (2u * o) & 31u;
                                const uint32_t* wp = rp + wi;
                                const uint32_t W0 = wp[0], W1 = wp[1], W2 = wp[2], W3 = wp[3];
                                snk_kmer f;
                                f.hi = ((uint64_t)funnel(W0, W1, sh) << 32) | funnel(W1, W2, sh);
                                if constexpr (K == 48) f.lo = (uint64_t)funnel(W2, W3, sh) << 32;
                                else f.lo = ((uint64_t)funnel(W2, W3, sh) << 32) | (funnel(W3, wp[4], sh) & 0xFFFFFF00u);
                                const snk_kmer rk = snk_kmer_rc<K>(f);
                                snk_kmer c = snk_kmer_lt(rk, f) ? rk : f;
                                if (GROUPED) c.lo |= (uint64_t)rp[7];
                                uint32_t h1, h2;
                                snk_kmer_hash_count<(K > 48) || GROUPED>(c, &h1, &h2);
                                if ((h2 & split_mask) == split_id) {
                                    const uint32_t cell = h1 >> SCB;                // 8192 / 16384 cells (the slot comes from h1's low bits)
                                    const uint32_t bit = 1u << (cell & 31u);
                                    uint32_t* w = planes + (cell >> 5);
                                    // (a folded record stands for wgt copies: as many steps, the level at most)
                                    const uint32_t steps = GROUPED ? 1u : min(wgt[i], a.screen);
                                    for (uint32_t k = 0; k < steps; ++k)
                                        if (atomicOr(w, bit) & bit) { if ((atomicOr(w + SPW, bit) & bit) && a.screen > 2u) atomicOr(w + 2 * SPW, bit); }
                                    cpk[r >> 1] |= cell << (16u * (r & 1u));
                                    inpass |= 1u << r;
                                }
                            }
                        }
                        lds_barrier();
                        const uint32_t* plane = planes + SPW * (a.screen > 2u ? 2 : 1);
#pragma unroll
                        for (uint32_t r = 0; r < (uint32_t)SROUNDS; ++r) {
                            const uint32_t cell = (cpk[r >> 1] >> (16u * (r & 1u))) & 0xFFFFu;
                            const bool adm = ((inpass >> r) & 1u) && ((plane[cell >> 5] >> (cell & 31u)) & 1u);
                            const unsigned long long am = __ballot(adm);
                            if (am) {
                                const int leader = __ffsll((long long)am) - 1;
                                uint32_t base = 0;
                                if (lane == leader) base = atomicAdd(&ctl[0], (uint32_t)__popcll(am));
                                base = __shfl(base, leader) + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                                if (adm && base < SCAND) cand[base] = (uint16_t)(r * THREADS + tid);
                            }
                        }
                        lds_barrier();
                        n_iter = LDS_LOAD(&ctl[0]);
                        if (n_iter > SCAND) { screened = false; n_iter = total; }      // (too many candidates for the list: every instance goes to the table)
                    }
                }
                for (uint32_t g0 = 0; g0 < n_iter; g0 += THREADS) {
                    if (TIGHT ? (LDS_LOAD(resv) >> 31) != 0u : LDS_LOAD(occ) > LIMIT) break;
                    uint32_t g = g0 + tid;
                    const bool have = g < n_iter;
                    if (SCREEN && screened && have) g = cand[g];
                    if (have) {
                        uint32_t lr = cidx[g >> 5];
                        uint32_t lend = lpre[lr];
                        while (lend <= g) lend = lpre[++lr];         // (the last leader ends at total > g)
                        const uint32_t i = lead[lr];
                        const uint32_t* rp = rec + 8 * i;
                        const uint32_t m6 = rp[6], w7 = rp[7];
                        const uint32_t wt = (SCREEN && GROUPED) ? 1u : wgt[i];
                        const uint32_t n_i = m6 & 0x7Fu, hasL = (m6 >> 7) & 1u, hasR = (m6 >> 8) & 1u;
                        const uint32_t j = g + n_i - lend;
                        const uint32_t bst = GROUPED ? 0u : w7;                   // merged barcode state of the supermer
                        const uint32_t o = hasL + j;                     // first base of the k-mer inside the record
                        const uint32_t wi = o >> 4, sh = (2u * o) & 31u;
                        // the k-mer's words: words wi .. wi+3 (K=48; +4 at K=60) of the record -- always inside its eight words
                        // (o <= K-M+1, so wi <= 2; the flag bits of word 6 and word 7 only reach the base after a k-mer that has
                        // no successor, which is not looked at) -- and the word before for the preceding base (the word before a
                        // record is other LDS data; it only matters when o > 0, and then wi > 0 or sh > 0)
                        const uint32_t* wp = rp + wi;
                        const uint32_t P = wp[-1], W0 = wp[0], W1 = wp[1], W2 = wp[2], W3 = wp[3];
                        const uint32_t F0 = funnel(W0, W1, sh), F1 = funnel(W1, W2, sh), F2 = funnel(W2, W3, sh);
                        const uint32_t pb = funnel(P, W0, sh) & 3u;       // base before the k-mer
                        uint32_t F3 = 0, nb;                              // base after the k-mer
                        if constexpr (K == 48) nb = (W3 >> (30u - sh)) & 3u;
                        else {
                            static_assert(K == 60, "K is 48 or 60");
                            const uint32_t W4 = wp[4];
                            const uint32_t f3 = funnel(W3, W4, sh);
                            nb = (f3 >> 6) & 3u;
                            F3 = f3 & 0xFFFFFF00u;
                        }
                        const uint32_t havesucc = hasR | ((j + 1u - n_i) >> 31), havepred = min(o, 1u);      // 0 / 1
                        snk_kmer f;
                        f.hi = ((uint64_t)F0 << 32) | F1;
                        f.lo = ((uint64_t)F2 << 32) | F3;
                        const snk_kmer r = snk_kmer_rc<K>(f);
                        const bool rev = snk_kmer_lt(r, f);          // isRev(): store the reverse complement (:164)
                        snk_kmer c = rev ? r : f;
                        // context of the stored orientation: the reverse complement's successor is the complement of the
                        // base before, its predecessor the complement of the base after (KMerContext.cc:19)
                        const uint32_t cfw = (havepred << (4u + pb)) | (havesucc << nb);
                        const uint32_t crv = (havesucc << (7u - nb)) | (havepred << (3u - pb));
                        const uint32_t ctx = rev ? crv : cfw;
                        if (GROUPED) c.lo |= (uint64_t)w7;           // (group, k-mer) is the counted entity
                        uint32_t h1, h2;
                        snk_kmer_hash_count<(K > 48) || GROUPED>(c, &h1, &h2);
                        if (a.dbg == 1) { if (h1 == 0x12345u && h2 == 0x54321u) a.status[3] = 7; }
                        else if ((h2 & split_mask) == split_id) {
                            // double hashing (odd stride, power-of-two table): a wave waits for its slowest lane, so the
                            // tail of the probe-length distribution matters more than its mean
                            uint32_t slot = h1 & (SLOTS - 1);
                            const uint32_t stride = ((h1 >> 16) ^ (h2 >> 20)) | 1u;
                            const lo_type clo = klo_pack<K, GROUPED>(c.lo);
                            const uint32_t mytag = (h2 & ~1u) | 2u;     // never 0; bit 0 = the slot's fields are in place
                            bool claimed = false;
                            // TIGHT: book one slot per probing lane of this wave (the lanes in here); a booking that does not fit ends the pass
                            // (the lanes of this round insert nothing: the pass is counted again in halves)
                            bool go = true;
                            uint32_t need = 0;
                            int rl = 0;
                            if constexpr (TIGHT) {
                                const unsigned long long pm = __ballot(1);
                                rl = __ffsll((long long)pm) - 1;
                                need = (uint32_t)__popcll(pm);
                                uint32_t r = 0;
                                if (lane == rl) {
                                    r = atomicAdd(resv, need);
                                    if ((r & 0x7FFFFFFFu) + need > LIMIT && !(r >> 31)) {
                                        // no room next to what the rounds in flight have booked: most of that comes back (their lanes find their
                                        // k-mers in the table).  Hand the booking back and look again a few times -- a bounded wait, nobody
                                        // waits while holding a booking -- unless the claims alone leave no room
                                        atomicSub(resv, need);
                                        r = 0x80000000u;
                                        for (int tries = 0; tries < tight_tries; ++tries) {
                                            if (LDS_LOAD(occ) + need > LIMIT) break;
                                            const uint32_t rv = LDS_LOAD(resv);
                                            if (rv >> 31) break;
                                            if (rv + need <= LIMIT) {
                                                const uint32_t r2 = atomicAdd(resv, need);
                                                if (r2 >> 31) break;
                                                if (r2 + need <= LIMIT) { r = r2; break; }
                                                atomicSub(resv, need);
                                            }
                                            __builtin_amdgcn_s_sleep(2);
                                        }
                                        if (r >> 31) atomicOr(resv, 0x80000000u);
                                    }
                                }
                                go = (__shfl(r, rl) >> 31) == 0u;
                            }
                            if (go) {
                            // find-or-claim.  The inner loop is the common case and nothing else: step over slots that hold
                            // other fingerprints (one tag load each).  It stops at an empty slot or at my fingerprint.
                            for (;;) {
                                uint32_t t;
                                for (;;) {
                                    t = __hip_atomic_load(&tag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (t == 0 || ((t ^ mytag) & ~1u) == 0) break;
                                    slot = (slot + stride) & (SLOTS - 1);
                                }
                                if (t == 0) {
                                    t = atomicCAS(&tag[slot], 0u, mytag);
                                    if (t == 0) {
                                        // mine: the first observation is stored, not added (no atomics; nobody touches the
                                        // fields before the ready bit is up)
                                        khi[slot] = c.hi;
                                        klo[slot] = clo;
                                        cnt[slot] = a.dbg == 2 ? 0u : wt;
                                        bcs[slot] = bst;
                                        reinterpret_cast<uint8_t*>(ctxw)[slot] = (uint8_t)ctx;
                                        if (bcset) {
#pragma unroll
                                            for (int q = 0; q < 6; ++q) bcx[slot * 6 + q] = 0;
                                        }
                                        __hip_atomic_store(&tag[slot], mytag | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        claimed = true;
                                        break;
                                    }
                                    if (((t ^ mytag) & ~1u) != 0) { slot = (slot + stride) & (SLOTS - 1); continue; }   // lost it to another k-mer
                                }
                                if (!(t & 1u)) continue;                 // my fingerprint, fields not written yet: look again
                                if (khi[slot] == c.hi && klo[slot] == clo) break;
                                slot = (slot + stride) & (SLOTS - 1);
                            }
                            if constexpr (TIGHT) {
                                const uint32_t nc = (uint32_t)__popcll(__ballot(claimed));
                                if (lane == rl && need > nc) atomicSub(resv, need - nc);      // (bit 31 stays: need - nc <= the count below it)
                            }
                            if (claimed) {
                                // one reservation per wave for the lanes that claimed in this round (same-address LDS atomics are served one lane at a time: 43.2 -> 42.5 ms)
                                const unsigned long long cm = __ballot(1);
                                const int leader = __ffsll((long long)cm) - 1;
                                uint32_t base = 0;
                                if (lane == leader) base = atomicAdd(occ, (uint32_t)__popcll(cm));
                                const uint32_t at = __shfl(base, leader) + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
                                if (at < LIMIT) olist[at] = (uint16_t)slot;
                            } else if (a.dbg != 2) {
                                atomicAdd(&cnt[slot], wt);
                                if (ctx) atomicOr(&ctxw[slot >> 2], ctx << (8 * (slot & 3)));
                                if (bst >= BC_MULTI) atomicMax(&bcs[slot], bst);
                                else if (bst) {
                                    uint32_t ob = atomicCAS(&bcs[slot], 0u, bst);
                                    if (ob != 0 && ob != bst && ob < BC_MULTI) {
                                        if (!bcset) atomicMax(&bcs[slot], BC_MULTI);
                                        else {
                                            bool stored = false;
                                            for (int q = 0; q < 6 && !stored; ++q) {
                                                const uint32_t ov = atomicCAS(&bcx[slot * 6 + q], 0u, bst);
                                                stored = ov == 0u || ov == bst;
                                            }
                                            if (!stored) atomicMax(&bcs[slot], BC_MULTI);       // an eighth distinct barcode
                                        }
                                    }
                                }
                            }
                            }
                        }
                    }
                }
                lds_barrier();   // 'inserted': rec and the map are rewritten by the next batch
                PROF(5);
                bq ^= 1u;
            }
        }
        // (every batch ends with a barrier: the occupancy and the table are final here)
        if (TIGHT ? (LDS_LOAD(resv) >> 31) != 0u : LDS_LOAD(occ) > LIMIT) {   // too many distinct k-mers for one table: split this pass in two by one more hash bit
            if (split_lg >= MAX_SPLIT_LOG2) { if (tid == 0) atomicExch(&a.status[1], 1u); }
            else {
                if (tid == 0) {
                    ctl[16 + 2 * sp] = split_lg + 1; ctl[17 + 2 * sp] = split_id;
                    ctl[18 + 2 * sp] = split_lg + 1; ctl[19 + 2 * sp] = split_id | (1u << split_lg);
                }
                sp += 2;
            }
            for (int s = tid; s < SLOTS; s += THREADS) tag[s] = 0;      // nothing is written out: the table is emptied wholesale
            ++splits_done;
            q ^= 1u;
            continue;
        }
        // K8: filter + write the surviving entries behind the workgroup's cursor in its own output region, in ONE pass over the
        // slots that were claimed (olist; a 5000-instance bucket claims ~500 of the 2048 slots, most of them by k-mers seen
        // once): the first version scanned the whole table twice (count the survivors, then rank and write them) with a
        // device-scope reservation and two barriers in between -- a quarter of the kernel.  A survivor's position is the
        // running cursor + its rank (one LDS atomic per wave); the cursor moves on after the one barrier that follows.
        PROF(6);
        // the last sub-pass of the bucket is past its insert phase: rec[] is free, the next bucket's first batch can be on its
        // way while this one's survivors are written (no registers involved; the next bucket's stage waits for it)
        if (sp == 0) {
            uint64_t nvb, nve;
            bucket_range(bnext, segn, nvb, nve);
            dma_batch(nvb, nve, bnext, segn);
            prefetched = true;
        }
        const uint32_t nocc = LDS_LOAD(occ);              // <= LIMIT here (the overflow case went the other way)
        const uint64_t rbase = rcur;
        const uint64_t gbase = (uint64_t)blockIdx.x * a.region_cap + rbase;
        for (uint32_t e0 = 0; e0 < nocc; e0 += THREADS) {
            const uint32_t e = e0 + tid;
            uint32_t s = 0, c = 0;
            bool ok = false;
            if (e < nocc) {
                s = olist[e];
                c = cnt[s];
                tag[s] = 0;                     // the slot is free for the next pass (nobody probes before the barrier below)
                ok = c >= a.min_freq && c != 0;
                if (ok && a.bc_mode) {
                    const uint32_t b = bcs[s];
                    if (!bcset) ok = a.bc_mode == 1 ? (b != 0) : (b >= BC_MULTI);
                    else {
                        uint32_t distinct = b ? 1u : 0u;
                        for (int j = 0; j < 6; ++j) distinct += bcx[s * 6 + j] ? 1u : 0u;
                        ok = b >= BC_MULTI || distinct >= a.bc_mode;
                    }
                }
            }
            const unsigned long long m = __ballot(ok);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t b = 0;
                if (lane == leader) b = atomicAdd(place, (uint32_t)__popcll(m));
                b = __builtin_amdgcn_readlane(b, leader);
                const uint64_t pos = rbase + b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (ok && pos < a.region_cap) {
                    const uint64_t at = gbase - rbase + pos;
                    const uint32_t cx = (ctxw[s >> 2] >> (8 * (s & 3))) & 0xFFu;
                    a.out_keys[at] = ((snk_u128)khi[s] << 64) | (snk_u128)klo_unpack<K, GROUPED>(klo[s]);
                    a.out_vals[at] = ((uint64_t)(c < 0xFFFFFFu ? c : 0xFFFFFFu) << 8) | cx;   // KDef::setCount saturates at 2^24-1 (kmers/ReadPather.h:127-131,145)
                }
            }
        }
        lds_barrier();                                               // 'written'
        const uint32_t nvalid = LDS_LOAD(place);
        if constexpr (TIGHT) {
            // more survivors than one chunk of the bucket-local graph stage holds: count the pass again in halves (the survivors just written
            // are overwritten: the cursor stays; the table is empty, the other parity's counters are zero as behind any pass)
            if (a.chunk_n && nvalid > SNK_GRAPH_CHUNK_MAX && split_lg >= MAX_SPLIT_LOG2) { if (tid == 0) atomicExch(&a.status[1], 1u); }      // (the host fails the call)
            else if (a.chunk_n && nvalid > SNK_GRAPH_CHUNK_MAX) {
                if (tid == 0) {
                    ctl[16 + 2 * sp] = split_lg + 1; ctl[17 + 2 * sp] = split_id;
                    ctl[18 + 2 * sp] = split_lg + 1; ctl[19 + 2 * sp] = split_id | (1u << split_lg);
                }
                sp += 2;
                ++splits_done;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the next bucket's first batch, if it was asked for, has landed before rec[] is fetched again)
                prefetched = false;
                q ^= 1u;
                continue;
            }
        }
        if (nvalid) {
            rcur += nvalid;                       // keeps counting past the capacity: the host learns what the region needs
            if (tid == 0) {
                if (rbase + nvalid > a.region_cap) a.status[0] = 1;
                else if (a.chunk_n) {
                    // chunk descriptor for the bucket-local graph stage: the survivors of one sub-pass are contiguous
                    if (split_lg == 0) { a.chunk_n[bucket] = nvalid; a.chunk_base[bucket] = (uint32_t)rbase; }
                    else {
                        // (the region travels with the descriptor: a virtual bucket's region is not its real bucket's)
                        const uint32_t e = atomicAdd(&a.status[4], 1u);
                        if (e < a.extra_cap) a.extra[e] = make_uint4(real_bucket, (uint32_t)rbase, nvalid | (blockIdx.x << 12), (split_lg << 24) | split_id);
                    }
                }
            }
        }
        claims_sum += nocc;                                  // (uniform) distinct k-mers this workgroup has held in its table
        if (tid == 0 && nocc > ctl[3]) ctl[3] = nocc;      // running maximum, reported once at the end (a device atomic per bucket: 2 M on one address are 20 ms)
        q ^= 1u;
        PROF(7);
    }
    if (tid == 0 && splits_done) atomicAdd(&a.status[2], 1u);
    par ^= 1u;
    }
    // status[5]: distinct k-mers over all buckets, in units of 16 -- with the instance count it tells the host how full the
    // tables run (the bucket size of the next call, snk_pipeline.hip)
    if (tid == 0) { a.region_cursor[blockIdx.x] = rcur; atomicMax(&a.status[3], ctl[3]); atomicAdd(&a.status[5], (uint32_t)(claims_sum >> 4)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA in flight when the workgroup's LDS is handed back
#ifdef SNK_COUNT_PROF
    if (tid == 0) for (int q = 0; q < 8; ++q) atomicAdd(&a.prof[q], prof_acc[q]);
#endif
}

template <int K> struct cfg;
template <> struct cfg<48> { static constexpr int THREADS = SNK_COUNT_THREADS; static constexpr int SLOTS = SNK_COUNT_SLOTS; };
template <> struct cfg<60> { static constexpr int THREADS = SNK_COUNT_THREADS; static constexpr int SLOTS = SNK_COUNT_SLOTS; };

template <int K, bool G>
size_t lds_bytes(uint32_t bc_mode = 0, bool tight = false, bool screen = false) {
    const size_t S = screen ? SNK_GSCREEN_SLOTS : cfg<K>::SLOTS, B = (G && screen && S <= 1024 && K == 48) ? SNK_GSCREEN_BATCH : ((K == 48 && !G && (S >= 2048 || screen)) ? 512 : 256), DD = 2 * B, NCI = B * (K - SNK_M_MIN_OF(K) + 1) / 32 + 2;
    return S * (8 + sizeof(typename klo_t<K, G>::type) + 4 + 4 + 4) + S + 4 * (8 * B + (tight ? 72 : 64) + DD + B) + 2 * (B + B + 2 + NCI) + 16 + 2 * 4 * 3 * SNK_COUNT_MAXSEG + 2 * (tight ? S - 64 : S - cfg<K>::THREADS - 64) + 16 + (bc_mode > 2 ? S * 24 + 16 : 0) + (screen ? (G ? (S <= 1024 ? 4096 : 2048) : 8192 + 3 * SNK_NGSCREEN_PLANE_WORDS * 4) + 16 : 0);
}

template <int K, bool G>
int launch(hipStream_t st, const snk_count_args& a, char* err, size_t errcap) {
    auto kern = a.nseg > 1 ? snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, true> : snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, false>;
    if (a.gidx) {
        if (a.nseg != 1) return snk_fail(SNK_E_INTERNAL, err, errcap, "count: an index list comes with one segment per bucket");
        kern = snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, false, true>;
    }
    const bool tight = a.tight && !a.gidx;
    const bool screen = K == 48 && tight && a.screen >= 2u && a.bc_mode <= 2u;
    static_assert(SNK_GSCREEN_SLOTS <= 1024, "the ungrouped SCREEN instantiation needs the LDS half a table frees");
    if constexpr (K == 48) {
        if (screen) kern = a.nseg > 1 ? snk_count_kernel<K, cfg<K>::THREADS, SNK_GSCREEN_SLOTS, G, true, false, true, true> : snk_count_kernel<K, cfg<K>::THREADS, SNK_GSCREEN_SLOTS, G, false, false, true, true>;
    }
    if (tight && !screen) kern = a.nseg > 1 ? snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, true, false, true> : snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, false, false, true>;
    size_t lds = lds_bytes<K, G>(a.bc_mode, tight, screen);
    SNK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (a.bucket0 >= a.NB) return SNK_OK;          // the launch covers buckets [bucket0, NB)
    // one workgroup per output region, every launch of a table (the ranged launches of the sharded path) with the same grid:
    // workgroup w counts the buckets == w (mod n_regions) and appends to region w
    snk_count_args b = a;
    b.bucket_stride = a.n_regions;
    hipLaunchKernelGGL(kern, dim3(a.n_regions), dim3(cfg<K>::THREADS), lds, st, b);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

// Workgroups walk strided bucket lists: 2 M one-bucket workgroups spend ~8 % of the kernel in dispatch (73.7 ms at 1e8
// reads); exactly one residency wave (grid = 2 x CUs) is no better (73.5: whoever finishes early idles to the end); 32-64
// waves keep both the dispatch cost and the tail small (67.3 ms).  SNK_COUNT_PERSIST sets the number of residency waves.
template <int K, bool G>
int regions(uint32_t nseg, uint32_t NB, uint32_t bc_mode, uint32_t* out, char* err, size_t errcap) {
    auto kern = nseg > 1 ? snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, true> : snk_count_kernel<K, cfg<K>::THREADS, cfg<K>::SLOTS, G, false>;
    size_t lds = lds_bytes<K, G>(bc_mode);
    SNK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t persist = snk_opt_u32("count_persist", 32);
    if (persist == 0) persist = 1;
    int per_cu = 0, dev = 0, n_cu = 256;
    SNK_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, cfg<K>::THREADS, lds));
    SNK_HIP_TRY(hipGetDevice(&dev));
    SNK_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    uint64_t grid = (uint64_t)(per_cu > 0 ? per_cu : 1) * (uint64_t)n_cu * persist;
    if (grid > NB) grid = NB;
    if (grid > (1u << 20)) grid = 1u << 20;
    *out = grid ? (uint32_t)grid : 1u;
    return SNK_OK;
}

}  // namespace

namespace {
__global__ void __launch_bounds__(256) compact_regions_kernel(const snk_u128* __restrict__ keys_in, const uint64_t* __restrict__ vals_in,
                                                              uint64_t region_cap, const unsigned long long* __restrict__ cursor,
                                                              const unsigned long long* __restrict__ off,
                                                              snk_u128* __restrict__ keys_out, uint64_t* __restrict__ vals_out) {
    const uint32_t r = blockIdx.x;
    const uint64_t n = cursor[r], src = (uint64_t)r * region_cap, dst = off[r];
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        keys_out[dst + i] = keys_in[src + i];
        vals_out[dst + i] = vals_in[src + i];
    }
}
}  // namespace

int snk_launch_compact_regions(hipStream_t st, const snk_u128* keys_in, const uint64_t* vals_in, uint64_t region_cap,
                               uint32_t n_regions, const unsigned long long* region_cursor,
                               const unsigned long long* region_off, snk_u128* keys_out, uint64_t* vals_out, char* err,
                               size_t errcap) {
    hipLaunchKernelGGL(compact_regions_kernel, dim3(n_regions), dim3(256), 0, st, keys_in, vals_in, region_cap, region_cursor,
                       region_off, keys_out, vals_out);
    SNK_HIP_TRY(hipGetLastError());
    return SNK_OK;
}

uint32_t snk_count_screen_limit() { return SNK_GSCREEN_SLOTS - 64; }      // usable slots of the SCREEN instantiations' table
uint32_t snk_count_slots(uint32_t K) { return K == 60 ? cfg<60>::SLOTS : cfg<48>::SLOTS; }

uint32_t snk_count_limit(uint32_t K, uint32_t grouped, uint32_t tight) {
    (void)grouped;
    const uint32_t S = K == 48 ? cfg<48>::SLOTS : cfg<60>::SLOTS, T = K == 48 ? cfg<48>::THREADS : cfg<60>::THREADS;
    return tight ? (tight & 0xFFFFu) : S - T - 64;
}
int snk_count_regions(uint32_t K, uint32_t grouped, uint32_t nseg, uint32_t NB, uint32_t bc_mode, uint32_t* n_regions, char* err, size_t errcap) {
    if (grouped) return regions<48, true>(nseg, NB, bc_mode, n_regions, err, errcap);
    if (K == 60) return regions<60, false>(nseg, NB, bc_mode, n_regions, err, errcap);
    return regions<48, false>(nseg, NB, bc_mode, n_regions, err, errcap);
}

int snk_launch_count(uint32_t K, hipStream_t st, const snk_count_args& a, char* err, size_t errcap) {
    if (a.NB == 0) return SNK_OK;
    if (a.bc_mode > 8) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "min_bc=%u: up to eight distinct barcodes are told apart per k-mer", a.bc_mode);
    if (a.nseg > SNK_COUNT_MAXSEG) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "more than %d record segments per bucket (nseg=%u)", SNK_COUNT_MAXSEG, a.nseg);
    if (a.grouped) {
        if (K != 48 || a.bc_mode) return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "grouped counting needs K=48 and no barcode rule");
        return launch<48, true>(st, a, err, errcap);
    }
    if (K == 48) return launch<48, false>(st, a, err, errcap);
    if (K == 60) return launch<60, false>(st, a, err, errcap);
    return snk_fail(SNK_E_UNSUPPORTED, err, errcap, "K=%u is not supported (48 or 60)", K);
}
