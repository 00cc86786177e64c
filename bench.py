#!/usr/bin/env python3
"""bench.py -- Gk-mers/s through count+graph at k=48 on synthetic linked reads (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (trim -> minimiser partition -> [all-to-all] -> LDS count/filter ->
bucket-local prune/links/fragments -> fragment join -> canonical unitigs) over one batch of synthetic reads that
is already resident in HBM.  Outputs of a step, all on the device: the retained k-mer table (keys, counts, pruned
contexts; in minimiser-bucket order unless --sorted-table), the k-mer spectrum and the canonical unitigs.
N=1 workload: BASELINE.json configs[1], 100 M x 150 bp, k=48 (override with --reads).  N>1: weak
scaling, every rank owns --reads reads of one (N x reads)-read data set, k-mer space sharded by
minimiser bucket, one all-to-all of supermer records per step.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, algorithmic bytes
per launch / HIP-event launch time, SURVEY.md 8(d): 32.65 B per k-mer instance at K=48) and, at N=1,
`cpu_baseline` (the reference's own C++ path -- oracle/_ref/snref_driver -- timed on this box's host
cores on a bounded sample; falls back to the single-thread C port in oracle/ when the binary is absent).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALG_BYTES_PER_KMER = {48: 32.65, 60: 40.8}
HBM_PEAK_GBS = 8000.0                          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_STREAM_GBS = 6290.0                        # what a float4 copy reaches (same guide): the bound a streaming kernel is priced against
# G wave64 VALU instructions per second: 1024 SIMDs x 2.4 GHz / 4 cycles each.  Four, not two: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.004
# quad-cycles per instruction on the count kernel's own mix (profiles/r05_pmc_instmix_1e8.csv) -- a wave64 integer instruction holds its SIMD
# for four cycles (tools/probe/valu_rate.hip: the multiplies, rotates and bit-field ops the hashes use all run at that full rate)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4
ATOMICS_PEAK_G = 27.0                          # random device-scope atomics per second, any flavour (tools/probe/atomics.hip, G/s)
METRIC = "Gk-mers/s through count+graph at k=48, 1.2B×150bp; bit-exact counts"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=float, default=1e8, help="reads per GPU")
    ap.add_argument("--k", type=int, default=48)
    ap.add_argument("--error-free", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=1e6, help="reads of the workload timed on the host cores")
    ap.add_argument("--cpu-threads", default="16,64,256", help="thread counts the CPU baseline is timed at (the best is reported)")
    ap.add_argument("--cpu-sample-10m", action="store_true",
                    help="also time the reference on BASELINE config 1 (10 M reads, SURVEY 8(d)(i)) at the best thread count of the sweep: "
                         "minutes of host time, so not part of the default run (cpu_baseline.sample_10m)")
    ap.add_argument("--sharded", action="store_true", help="force the sharded (multi-GPU) code path even with one rank")
    ap.add_argument("--transport", choices=("rccl", "gloo"), default=os.environ.get("SNK_BENCH_TRANSPORT", "rccl"),
                    help="rccl: one process per GPU over RCCL/xGMI (the measured configuration).  gloo: the same multi-process step with the exchanges "
                         "staged through the host and ranks sharing the visible GPUs round-robin -- a functional check of the N > 1 path on a box with "
                         "fewer GPUs than ranks, not a performance number")
    ap.add_argument("--sorted-table", action="store_true",
                    help="also sort the retained k-mer table by key (+~40 ms; the reference's dictionary is an unordered hash set, "
                         "the default leaves the table in minimiser-bucket order = SNK_F_UNSORTED_TABLE)")
    ap.add_argument("--global-graph", action="store_true", help="global graph stage instead of the bucket-local one")
    ap.add_argument("--grouped", action="store_true",
                    help="BASELINE config 5: per-barcode local graphs (group = barcode, frequency rule only, --min-freq); "
                         "replicas only for N>1 (every rank owns whole barcodes, no collective)")
    ap.add_argument("--minimiser", choices=["auto", "16", "20"], default="auto",
                    help="minimiser length of the partition: 20 = SNK_F_LONG_MINIMISER; auto = 20 when the job's genome has more minimiser sites than "
                         "~0.7 per canonical 16-mer (genome >= 1.5 Gb: the 6..8-GPU lines of this bench, whose genome grows with the reads), else 16")
    ap.add_argument("--min-freq", type=int, default=3)
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed self-check of the sharded path")
    ap.add_argument("--no-next-rows", action="store_true", help="N=1: skip the untimed f1/f4 rows (read pathing, MarkDups, barcode lists) on the bench workload")
    ap.add_argument("--no-ingest", action="store_true", help="N=1: skip the untimed f3 row (FASTH files -> HBM)")
    ap.add_argument("--no-robust", action="store_true", help="N=1: skip config.robust (the step off its operating point: more errors, half the coverage, repeat-rich genome)")
    ap.add_argument("--reserve-gb", type=float, default=-1.0,
                    help="one-GPU path: device memory mapped into the context's arena before the first call (snk_ctx_reserve), as a host that owns the GPU "
                         "does at start-up; -1 = 1.3 GB per million reads up to 45 %% of the device, 0 = none (every call that outgrows the arena pays "
                         "the driver ~25-30 ms per new GB inside the call)")
    ap.add_argument("--no-df-seam", action="store_true", help="N=1: skip the untimed b1/b2 row (reads.fastb / .qualp / .bci -> device decode -> unitigs)")
    ap.add_argument("--no-large-job", action="store_true", help="skip config.large_job (800 M reads on this GPU in its own process, ~40 s; only the default run has it)")
    ap.add_argument("--large-job-reads", type=float, default=8e8)
    ap.add_argument("--df-reads", type=float, default=1e8, help="reads of the df_seam row's stage-input files")
    ap.add_argument("--df-dir", default="", help="where the df_seam row writes its files (default: $TMPDIR or /tmp)")
    ap.add_argument("--df-threads", type=int, default=0, help="pread workers of the df_seam row (0 = from the CPU budget, at most 32)")
    ap.add_argument("--ingest-files", type=int, default=64)
    ap.add_argument("--ingest-pairs", type=int, default=100_000, help="read pairs per FASTH file of the f3 row")
    ap.add_argument("--ingest-threads", type=int, default=0, help="decode threads (0 = one per file up to the host's hardware threads)")
    return ap.parse_args()


def cpu_baseline(sp_full, K: int, sample_reads: int, threads=(16, 64, 256), big: bool = False):
    """Reference CPU path on a bounded sample (first `sample_reads` reads of a data set with the same coverage)."""
    import numpy as np
    from supernova_amd import synth
    n = int(sample_reads)
    sp = synth.synth_params(n, seed=sp_full.seed, error_free=(sp_full.sub_ppm == 0))
    rows, quals, bc = synth.synth_host(sp)
    cores = os.cpu_count() or 1
    drv = ROOT / "oracle" / "_ref" / "snref_driver"
    if K == 48 and drv.exists():
        sys.path.insert(0, str(ROOT / "tests"))
        import refio
        asc = synth.codes_to_ascii(synth.unpack_rows(rows, sp.read_len))
        # the reference's MapReduce engine does not scale with the core count (its best is a few dozen threads): the sample
        # is timed at several thread counts and the best one is reported, with the whole sweep in `sample`
        sweep = sorted({t for t in threads if 0 < t <= cores} or {cores})
        best, runs = None, []
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
            refio.write_snkrd(Path(td) / "in.snkrd", np.full(n, sp.read_len), asc, quals, bc)
            for t in sweep:
                out = refio.run_ref(Path(td) / "in.snkrd", Path(td) / f"out{t}", threads=t, mode="time", timeout=1800)
                m = re.search(r"SNREF_TIME seconds=([0-9.]+) threads=(\d+) reads=(\d+) kmer_instances=(\d+)", out)
                secs, inst = float(m.group(1)), int(m.group(4))
                runs.append((t, secs))
                if best is None or secs < best[1]:
                    best = (t, secs, inst)
        t, secs, inst = best
        out = {"value": inst / secs / 1e9, "unit": "Gk-mers/s", "cores": t, "kind": "reference",
               "sample": f"{n} reads x {sp.read_len} bp of the same synthetic model ({inst} k-mer instances), "
                         f"buildReadQGraph48 (count+unitigs+HBV, no read pathing); best of "
                         + ", ".join(f"{tt} threads {ss:.2f} s" for tt, ss in runs) + f" on a {cores}-thread host"}
        if big:
            # BASELINE config 1 in time mode at the sweep's best thread count (the MapReduce engine's passes grow with the input)
            nb = 10_000_000
            spb = synth.synth_params(nb, seed=0x5EED0001, error_free=(sp_full.sub_ppm == 0))
            rb, qb, bb = synth.synth_host(spb)
            ab = synth.codes_to_ascii(synth.unpack_rows(rb, spb.read_len))
            with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
                refio.write_snkrd(Path(td) / "in.snkrd", np.full(nb, spb.read_len), ab, qb, bb)
                o = refio.run_ref(Path(td) / "in.snkrd", Path(td) / "out", threads=t, mode="time", timeout=7200)
                m = re.search(r"SNREF_TIME seconds=([0-9.]+) threads=(\d+) reads=(\d+) kmer_instances=(\d+)", o)
                out["sample_10m"] = {"reads": nb, "seconds": float(m.group(1)), "threads": t, "kmer_instances": int(m.group(4)),
                                     "value": int(m.group(4)) / float(m.group(1)) / 1e9, "unit": "Gk-mers/s"}
        return out
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    n = min(n, 200_000)
    gl = oracle_lib.good_lens(quals[:n], sp.read_len, K=K)
    t0 = time.time()
    o = oracle_lib.OracleResult(synth.unpack_rows(rows[:n], sp.read_len), gl, bc[:n], K=K, hbv=False)
    secs = time.time() - t0
    return {"value": o.n_instances / secs / 1e9, "unit": "Gk-mers/s", "cores": 1, "kind": "port",
            "sample": f"{n} reads x {sp.read_len} bp, oracle/snk_oracle.c count+unitigs, {secs:.2f} s single thread"}


def next_rows(eng, res, rows, quals, bc, read_len, K):
    """f1 / f4 on the bench workload, untimed rows next to the contract line: HIP-event times of the library's own
    phases, each with the bytes its data model moves once and the fraction of the HBM peak that is."""
    n = int(rows.shape[0])
    # (like the timed step: one call to size the arena, the second is reported)
    for _ in range(2):
        _, _, _, info = res.path_reads(rows, read_len, quals, mark_dups=True, bc=bc, unitig_bcs=True, download=False)
    rw, qs = int(rows.shape[1]) * 4, int(quals.shape[1])
    frac = lambda nbytes, ms: (nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None
    path_bytes = n * (rw + qs + 16) + info["n_edges_total"] * 4
    dict_bytes = info["dict_slots"] * 16 + int(res.unitig_total_bases)      # the slots cleared and written (8 B each) + the unitig bases read
    dup_bytes = n * (16 + rw + 2 * 12)          # path head (edge, offset), mate head row, two sort passes over 12-byte (key, id) records
    bcs_bytes = n * 8 * 4                       # one 8-byte (unitig, barcode) key per barcoded read through a 64-bit radix sort (write + read, twice)
    d = info["dups"]
    return {
        "reads": n, "calls": "second of two (arena warm, like the timed step)",
        "f1_dictionary_build": {"ms": round(info["dict_ms"], 3), "slots": info["dict_slots"], "alg_bytes": dict_bytes, "hbm_frac": frac(dict_bytes, info["dict_ms"])},
        "f1_read_pathing": {"ms": round(info["path_ms"], 3), "reads_per_s": n / (info["path_ms"] * 1e-3), "edges": info["n_edges_total"],
                            "reads_in_second_pass": info["n_slow"],
                            "alg_bytes": path_bytes, "hbm_frac": frac(path_bytes, info["path_ms"])},
        "f2_hbv_device_ms": round(info["hbv_device_ms"], 3),
        "f4_mark_dups": {"ms": round(d["ms"], 3), "dup_pairs": d["n_dup_pairs"], "interdup_rate": d["interdup_rate"], "alg_bytes": dup_bytes,
                         "hbm_frac": frac(dup_bytes, d["ms"])},
        "f4_unitig_barcode_lists": {"ms": round(info["bcs_ms"], 3), "entries": info["n_unitig_bcs"], "alg_bytes": bcs_bytes, "hbm_frac": frac(bcs_bytes, info["bcs_ms"])},
    }


def ingest_row(eng, args, K, step_ms_per_read):
    """f3: synthetic FASTH files -> reads resident in HBM (parallel inflate, uploads / pack / barcode ids overlapped),
    then the same count+graph step on what arrived."""
    import shutil
    from supernova_amd import ingest, synth
    from supernova_amd.engine import Params
    nf, ppf = args.ingest_files, args.ingest_pairs
    n = 2 * nf * ppf
    sp = synth.synth_params(n, seed=0x5EED0F33)
    td = Path(tempfile.mkdtemp(prefix="snk_fasth_", dir=os.environ.get("TMPDIR", "/tmp")))
    try:
        t0 = time.perf_counter()
        paths, text = ingest.write_synth_fasth(td, sp, nf, ppf, workers=os.cpu_count() or 8)
        t_write = time.perf_counter() - t0
        nbc = n // (2 * sp.pairs_per_bc) + 8
        wl = ingest.synth_whitelist(nbc)
        best = None
        for _ in range(2):                      # the second pass has the files in the page cache for certain
            dr = ingest.ingest_fasth(eng, paths, sp.read_len, wl, threads=args.ingest_threads)
            st = dr.stats
            if best is None or st["seconds"] < best["seconds"]:
                best = dict(st)
            res = eng.count_graph_reads(dr.dev_reads(), Params(K=K, sorted_table=False))
            n_k, n_u = int(res.n_kmers), int(res.n_unitigs)
            dr.close()
        # end to end with the reads never resident: batches are partitioned as they arrive (snk_dev_ingest_count_graph); the wall is
        # max(ingest, partition) + count + graph, not ingest + step
        e2e = None
        for _ in range(2):
            t0 = time.perf_counter()
            r2, st2 = ingest.ingest_count_graph(eng, paths, sp.read_len, wl, params=Params(K=K, sorted_table=False), threads=args.ingest_threads, total_reads_hint=n)
            wall = time.perf_counter() - t0
            if e2e is None or wall < e2e["wall_seconds"]:
                e2e = {"wall_seconds": wall, "text_GB_per_s": text / wall / 1e9, "reads_per_s": n / wall, "decode_wait_share": st2["decode_wait_seconds"] / st2["seconds"],
                       "retained_kmers": int(r2.n_kmers), "unitigs": int(r2.n_unitigs), "same_result_as_resident": bool(int(r2.n_kmers) == n_k and int(r2.n_unitigs) == n_u),
                       "count_graph_ms_after_last_batch": round(r2.phase_ms["count"] + r2.phase_ms["graph"], 2)}
        secs = best["seconds"]
        return {"files": nf, "reads": n, "text_GB": text / 1e9, "compressed_GB": best["compressed_bytes"] / 1e9, "seconds": secs, "fasth_to_unitigs_streamed": e2e,
                "text_GB_per_s": text / secs / 1e9, "reads_per_s": n / secs, "decode_wait_share": best["decode_wait_seconds"] / secs,
                "setup_seconds": best["setup_seconds"], "text_GB_per_s_after_setup": text / max(secs - best["setup_seconds"], 1e-9) / 1e9,
                "decode_threads": args.ingest_threads or min(nf, max(1, int(eng.lib.snk_host_cpu_budget()) - 2)), "host_threads": os.cpu_count(),
                "host_cpu_budget": int(eng.lib.snk_host_cpu_budget()),      # the cgroup's CPU quota when there is one: what host-side inflate can use at all
                "count_graph_on_ingested": {"retained_kmers": n_k, "unitigs": n_u},
                # how long the device step of the same reads is against their ingest: the share of the ingest the step hides behind
                "step_over_ingest": (step_ms_per_read * n * 1e-3) / secs, "synth_files_written_in_s": t_write}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def large_job_row(args):
    """How far ONE GPU goes: --large-job-reads (800 M = two thirds of BASELINE config 3's whole job) through the resident step in bucket-range
    passes, reads held in the DF seam's compact form, with the size-independent checks (tools/r6_full_job.py; DESIGN 4 "round 6").  Runs in
    its own process after everything else of this run has given its device memory back; a failure is reported, never raised."""
    import subprocess
    import torch
    try:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        def one(reads, env):
            t0 = time.perf_counter()
            pr = subprocess.run([sys.executable, str(ROOT / "tools" / "r6_full_job.py"), str(reads), "5e7", "2"], capture_output=True, text=True, timeout=420, env=dict(os.environ, **env))
            rows_ = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
            if pr.returncode != 0 or not rows_:
                return {"failed": (pr.stderr or pr.stdout).strip().splitlines()[-1][:300] if (pr.stderr or pr.stdout).strip() else f"exit {pr.returncode}"}
            d = rows_[-1]
            return {"reads": int(float(reads)), "held_as": "packed rows + good lengths + barcode ids (46 B per read)", "call": "second of two",
                    "wall_s": d["wall_s"], "Gkmers_per_s": d["Gkmers_per_s"], "instances": d["instances"], "retained_kmers": d["retained_kmers"], "unitigs": d["unitigs"],
                    "bucket_range_passes": d["passes"], "buckets": d["buckets"], "phase_ms": d["phase_ms"], "scratch_gib": d["scratch_gb"], "fragments": d["n_fragments"],
                    "first_call_s": rows_[0]["wall_s"], "checks": {"every_count_at_least_min_freq": d["min_count"] >= 3, "spectrum_adds_up": d["spectrum_adds_up"],
                                                                   "unitig_lengths_add_up": d["unitig_lengths_add_up"], "same_as_first_call": d["same_as_first_call"]},
                    "row_seconds": round(time.perf_counter() - t0, 1)}
        out = one(args.large_job_reads, {})
        out["note"] = "one MI355X; the north star asks 50 Gk-mers/s of eight for 1.2 B reads; 2^31 retained k-mers (32-bit node states) is the one-GPU limit"
        # BASELINE config 5's WHOLE job -- 1.2 B reads as per-barcode local graphs (a barcode's k-mers are few: 0.45 G retained) -- on this one GPU
        out["config5_whole_job_per_barcode_graphs"] = one(1.2e9, {"GROUPED": "1"})
        return out
    except Exception as ex:
        return {"failed": str(ex)[:300]}


def df_seam_row(eng, args, K):
    """b1/b2 at rate: the ASSEMBLER_DF stage inputs -- reads.fastb / reads.qualp / reads.bci -- of a synthetic data set
    decoded on the device, slab by slab, and counted+graphed (snk_df_open / snk_dev_ingest_df_count_graph, what supernova_amd/df_stage.py and
    snk_mspedges run), against the same reads decoded into resident arrays and counted by one resident call.  Files are written once by the
    library's own writer (qualities jittered over [30, 38): a quality file of sequencer-like entropy with the same trim), so they sit in the
    page cache.  Beside it: the rate of the old one-thread host readers on a prefix of the same data."""
    import hashlib
    import shutil
    from supernova_amd import dfin, formats, synth
    from supernova_amd.engine import Params
    n = int(args.df_reads) & ~1
    sp = synth.synth_params(n, seed=0x5EED0DF5, unbarcoded_ppm=0)
    td = Path(tempfile.mkdtemp(prefix="snk_df_", dir=args.df_dir or os.environ.get("TMPDIR", "/tmp")))
    params = Params(K=K, sorted_table=False)
    try:
        t0 = time.perf_counter()
        dfin.write_synth_df(td / "reads", sp, qual_jitter=8)
        t_write = time.perf_counter() - t0
        n_small = min(n, 2_000_000)
        dfin.write_synth_df(td / "small", sp, n=n_small, qual_jitter=8)
        t0 = time.perf_counter()
        rows_h, lens_h, mx = formats.read_fastb(td / "small.fastb")
        t_fb = time.perf_counter() - t0
        t0 = time.perf_counter()
        formats.read_qualp(td / "small.qualp", n_small, mx)
        t_qp = time.perf_counter() - t0
        del rows_h, lens_h
        with dfin.DfFiles(td / "reads") as f:
            fb = f.file_bytes
            # resident: the decode alone (file bytes -> rows / quality rows / lengths / barcode ids in HBM), then one resident count+graph
            ing = []
            for _ in range(2):
                dr = f.ingest(eng, read_len=sp.read_len, threads=args.df_threads)
                ing.append(dict(dr.stats))
                if _ == 0:
                    dr.close()
            best_ing = min(ing, key=lambda d: d["seconds"])
            ref = eng.count_graph_reads(dr.dev_reads(), params)
            want = (int(ref.n_kmers), int(ref.n_unitigs), int(ref.n_instances), hashlib.sha256(ref.bv_image()).hexdigest(), ref.spectrum().tobytes())
            del ref
            dr.close()
            calls, same = [], True
            for rep in range(3):
                t0 = time.perf_counter()
                res, st = f.count_graph(eng, params, read_len=sp.read_len, threads=args.df_threads)
                img = res.bv_image()
                wall = time.perf_counter() - t0
                got = (int(res.n_kmers), int(res.n_unitigs), int(res.n_instances), hashlib.sha256(img).hexdigest(), res.spectrum().tobytes())
                same = same and got == want
                calls.append({"wall_s": round(wall, 4), "ingest_partition_s": round(st["seconds"] - (res.phase_ms["count"] + res.phase_ms["graph"]) * 1e-3, 4),
                              "io_wait_s": round(st["io_wait_seconds"], 4), "setup_s": round(st["setup_seconds"], 4),
                              "count_graph_ms": round(res.phase_ms["count"] + res.phase_ms["graph"], 2)})
                n_inst, n_slabs, moved = int(res.n_instances), st["n_slabs"], st["file_bytes"]
                del res
            # ... and through a streamed job (option df_stream = 2: every slab partitioned as it arrives, the reads never resident in any form)
            eng.set_option("df_stream", 2)
            streamed = []
            for rep in range(2):
                t0 = time.perf_counter()
                res, stc = f.count_graph(eng, params, read_len=sp.read_len, threads=args.df_threads)
                imgc = res.bv_image()
                streamed.append(round(time.perf_counter() - t0, 4))
                same = same and stc["mode"] == "streamed" and hashlib.sha256(imgc).hexdigest() == want[3]
                del res
            eng.clear_option("df_stream")
            # the decode read_len = 0 (the row length found by a scan of the file's length table), as the stage adapter calls it
            t0 = time.perf_counter()
            res, st0 = f.count_graph(eng, params, threads=args.df_threads)
            res.bv_image()
            wall_scan = time.perf_counter() - t0
            del res
        best = min(calls[1:], key=lambda c: c["wall_s"])
        return {"reads": n, "dir": str(td.parent), "file_GB": round(fb / 1e9, 3), "bytes_per_read": round(fb / n, 1), "files_written_in_s": round(t_write, 2),
                "resident_decode": {"seconds": round(best_ing["seconds"], 4), "file_GB_per_s": round(best_ing["text_bytes"] / best_ing["seconds"] / 1e9, 2),
                                    "reads_per_s": n / best_ing["seconds"], "io_wait_s": round(best_ing["decode_wait_seconds"], 4),
                                    "first_call_seconds": round(ing[0]["seconds"], 4), "slabs": best_ing["n_batches"],
                                    "of_it_device_arrays_allocated_s": round(best_ing["setup_seconds"], 4)},
                "fastb_to_unitigs": {"wall_s": best["wall_s"], "file_GB_per_s": round(moved / best["wall_s"] / 1e9, 2), "reads_per_s": n / best["wall_s"],
                                     "Gkmers_per_s": round(n_inst / best["wall_s"] / 1e9, 2), "slabs": n_slabs, "calls": calls,
                                     "wall_s_with_length_scan": round(wall_scan, 4), "includes": "open files .. .bv image on the host",
                                     "mode": "compact (rows + good lengths + barcode ids stay: 46 B per read; the adaptive resident step)", "streamed_job_wall_s": streamed},
                "same_result_as_resident": bool(same),
                "io_threads": args.df_threads or "auto", "host_threads": os.cpu_count(), "host_cpu_budget": int(eng.lib.snk_host_cpu_budget()),
                "old_host_readers": {"reads": n_small, "fastb_reads_per_s": n_small / t_fb, "qualp_reads_per_s": n_small / t_qp,
                                     "both_reads_per_s": n_small / (t_fb + t_qp), "note": "snk_read_fastb / snk_read_qualp, one thread (tests' byte-for-byte check of the device decode)"}}
    finally:
        shutil.rmtree(td, ignore_errors=True)


ROBUST = (("errors_0.6pct", dict(sub_ppm=6000)),
          ("errors_1.5pct_tails_50pct", dict(sub_ppm=15000, lowq_tail_ppm=500000)),
          ("coverage_28x", None),                # genome_len = reads * read_len / 28
          ("repeat_rich_genome", dict(repeat_mode=15)),
          # every fifth base of the genome an A: ~1 site per canonical 16-mer value, the minimiser-sharing regime of a human genome at the
          # bench's size; the row also times the step with SNK_F_LONG_MINIMISER (20-base minimisers)
          ("crowded_minimiser_space", dict(repeat_mode=16)))


def robust_rows(eng, per_gpu, K, headline_ms, arena_bytes=0):
    """The same step off the bench's operating point, 100 M reads each, on the SAME engine (arena warm, like the timed
    steps): the first call on the new data (it may look at the first buckets and partition a second time) and the better of the next two
    (the figure: the second call still sizes the arena for the new bucket count).  Every model also exists as a 200 k-read digest of the REFERENCE's result (tests/golden/big_hashes.json robust_*) that
    tests/test_gpu_bigparity.py compares the HIP path with."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    out = {}
    arena_seen = int(arena_bytes)          # the arena the timed steps left behind
    for name, ov in ROBUST:
        ov = dict(ov) if ov is not None else dict(genome_len=per_gpu * 150 // 28)
        sp = synth.synth_params(per_gpu, seed=0x5EED0042, **ov)
        rows, quals, bc = eng.synth(sp)
        torch.cuda.synchronize()
        calls = []
        arena_before = arena_seen
        for rep in range(3):        # the first call meets new data (it may partition twice), the second sizes the arena for it, the third is steady state
            t0 = time.perf_counter()
            r = eng.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=Params(K=K, sorted_table=False))
            torch.cuda.synchronize()
            calls.append(((time.perf_counter() - t0) * 1e3, int(r.repartitioned)))
            if rep == 0:
                first_phases = {k: round(v, 1) for k, v in r.phase_ms.items() if k in ("partition", "count", "graph", "total")}
                # (a first call that needs more scratch than anything the context has seen, on a context whose arena was not reserved ahead
                # -- --reserve-gb 0 -- also pays the driver for the arena's growth, ~25-30 ms per GB of freshly mapped memory, inside
                # whichever stage asks for it: 745 instead of 227 ms for the 1.5 % row behind the 0.6 % one)
                first_phases.update(buckets=int(r.n_buckets), buckets_split=int(r.buckets_split), scratch_gb=round(r.scratch_bytes / 2**30, 1),
                                    scratch_over_earlier_calls_gb=round(max(0, r.scratch_bytes - arena_before) / 2**30, 1))
            if calls[-1][0] > 20000:       # a pathological case is reported, not repeated
                break
        arena_seen = max(arena_seen, int(r.scratch_bytes))
        ms = min(c[0] for c in calls[1:]) if len(calls) > 1 else calls[0][0]
        out[name] = {"ms": round(ms, 2), "Gkmers_per_s": round(r.n_instances / ms / 1e6, 2), "vs_headline_ms": round(ms / headline_ms, 3),
                     "first_call_ms": round(calls[0][0], 2), "first_call_repartitioned": calls[0][1], "first_call_phases": first_phases, "calls_ms": [round(c[0], 1) for c in calls],
                     "phase_ms": {k: round(v, 2) for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")},
                     "graph_ms": {k: round(v, 2) for k, v in r.graph_ms.items()}, "hot_buckets": int(r.n_hot_buckets),
                     "buckets": int(r.n_buckets), "buckets_split": int(r.buckets_split), "overflow_supermers": int(r.n_overflow),
                     "retained_kmers": int(r.n_kmers), "unitigs": int(r.n_unitigs)}
        if name == "crowded_minimiser_space":
            lm = []
            for rep in range(3):
                t0 = time.perf_counter()
                r = eng.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=Params(K=K, sorted_table=False, long_minimiser=True))
                torch.cuda.synchronize()
                lm.append((time.perf_counter() - t0) * 1e3)
            out[name]["long_minimiser_ms"] = round(min(lm[1:]), 2)
            out[name]["long_minimiser_phase_ms"] = {k: round(v, 2) for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")}
        del rows, quals, bc, r
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: run the N ranks under torch.distributed.run on this node (127.0.0.1, a free
    port), exactly as the driver's own N > 1 command does, and return their exit code.  Rank 0's JSON line is the children's stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the driver's own command
            # line), hand their stdout through -- rank 0 prints the one JSON line -- and leave with their exit code
            if args.transport == "rccl" and torch.cuda.device_count() < args.gpus:
                raise SystemExit(f"--gpus {args.gpus} over RCCL needs {args.gpus} visible GPUs, this node shows {torch.cuda.device_count()} "
                                 "(--transport gloo runs the same N-process job on fewer GPUs as a functional check)")
            raise SystemExit(self_launch(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if args.transport == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    cdev = torch.device("cpu") if args.transport == "gloo" else torch.device("cuda", local_rank)      # where the bench's own tiny collectives live
    use_dist = world > 1 or args.sharded
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.transport == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params

    K = args.k
    per_gpu = int(args.reads)
    total_reads = per_gpu * world
    eng = Engine(local_rank)
    reserved_gb = 0.0
    if world == 1 and args.reserve_gb != 0:      # (a one-rank step takes the growing arena too; ranks of a larger job keep plain blocks: DESIGN 6)
        cap_gb = 0.45 * torch.cuda.get_device_properties(local_rank).total_memory / 2**30
        reserved_gb = min(cap_gb, 1.3 * per_gpu / 1e6) if args.reserve_gb < 0 else args.reserve_gb
        try:
            eng.reserve(int(reserved_gb * 2**30))
        except Exception:
            reserved_gb = 0.0
    sp = synth.synth_params(total_reads, seed=0x5EED0000 + (1 if world == 1 else 2), error_free=args.error_free)
    rows, quals, bc = eng.synth(sp, first=rank * per_gpu, n=per_gpu)
    torch.cuda.synchronize()
    long_min = args.minimiser == "20" or (args.minimiser == "auto" and int(sp.genome_len) >= 1_500_000_000)      # sites that share a minimiser share a bucket
    params = Params(K=K, sorted_table=args.sorted_table, global_graph=args.global_graph, min_freq=args.min_freq, long_minimiser=long_min)
    if args.grouped:
        assert per_gpu % (2 * sp.pairs_per_bc) == 0, "--grouped: reads per GPU must be a multiple of the reads per barcode"
        params = Params(K=K, sorted_table=False, grouped=True, min_bc=0, min_freq=args.min_freq)

    if args.grouped:
        use_dist_collectives = False

        def step():      # replicas only: the barcodes of this rank's slab belong to nobody else
            return eng.count_graph(rows, sp.read_len, quals=quals, bc=None, group=bc, params=params)
    elif not use_dist:
        tail_ms = []

        def step():
            # the whole hand-off is inside the step: unitigs -> BVComp order + the .bv file's bytes (a13) and the graph from the unitigs
            # (a14: end keys, vertex classes, ids, fwd/rev translation), all from the device-resident result
            r = eng.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=params)
            t0 = time.perf_counter()
            r.bv = r.bv_image_device()
            r.hbv_graph = r.hbv()
            tail_ms.append((time.perf_counter() - t0) * 1e3)
            return r
    else:
        from supernova_amd.sharded import ShardedEngine
        sh = ShardedEngine(eng, dist)

        def step():
            return sh.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=params, read_index_base=rank * per_gpu, total_reads=total_reads)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed self-check of the sharded path: a small data set through the SAME exchanges, compared on rank 0 with the
    # one-GPU path over all of its reads (retained table as a multiset checksum, unitigs as a set).  A transport that
    # drops or garbles bytes shows up here, not as a fast wrong number.
    verified = None
    if use_dist and not args.grouped and not args.no_verify:
        import hashlib
        import numpy as np
        per_v = 2_000_000
        spv = synth.synth_params(per_v * world, seed=0x5EED0F00 + world, error_free=args.error_free)
        rv, qv, bv = eng.synth(spv, first=rank * per_v, n=per_v)
        resv = sh.count_graph(rv, spv.read_len, quals=qv, bc=bv, params=params, read_index_base=rank * per_v, total_reads=per_v * world)
        kv = resv.keys().astype(np.uint64)
        mixed = (kv[:, 0] * np.uint64(0x9E3779B97F4A7C15) + kv[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) + kv[:, 2] * np.uint64(0x165667B19E3779F9)
                 + resv.counts().astype(np.uint64) * np.uint64(0x27D4EB2F165667C5) + resv.ctx().astype(np.uint64) * np.uint64(0x85EBCA77C2B2AE63))
        # every rank wrote the unitigs whose head fragment it owns: their union is compared as an order-independent sum of
        # 64-bit hashes (fetched now: the next call on this engine recycles the result buffers)
        uh = sum(int.from_bytes(hashlib.sha256(u.encode()).digest()[:8], "little") for u in resv.unitigs()) & 0xFFFFFFFFFFFFFFFF
        nu_loc = int(resv.n_unitigs)
        # exact 64-bit multiset checksum over an int64 transport: the two 32-bit halves are reduced separately
        msum = int(mixed.sum(dtype=np.uint64))
        loc = torch.tensor([int(kv.shape[0]), msum & 0xFFFFFFFF, msum >> 32, nu_loc, uh & 0xFFFFFFFF, uh >> 32], dtype=torch.int64, device=cdev)
        if world > 1:
            dist.all_reduce(loc)
        if rank == 0:
            ra, qa, ba = eng.synth(spv, first=0, n=per_v * world)
            ref = eng.count_graph(ra, spv.read_len, quals=qa, bc=ba, params=params)
            kr = ref.keys().astype(np.uint64)
            mr = (kr[:, 0] * np.uint64(0x9E3779B97F4A7C15) + kr[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) + kr[:, 2] * np.uint64(0x165667B19E3779F9)
                  + ref.counts().astype(np.uint64) * np.uint64(0x27D4EB2F165667C5) + ref.ctx().astype(np.uint64) * np.uint64(0x85EBCA77C2B2AE63))
            got = (int(loc[1]) + (int(loc[2]) << 32)) & 0xFFFFFFFFFFFFFFFF
            same_table = int(loc[0]) == kr.shape[0] and got == int(mr.sum(dtype=np.uint64))
            ur = ref.unitigs()
            uhr = sum(int.from_bytes(hashlib.sha256(u.encode()).digest()[:8], "little") for u in ur) & 0xFFFFFFFFFFFFFFFF
            same_unitigs = int(loc[3]) == len(ur) and ((int(loc[4]) + (int(loc[5]) << 32)) & 0xFFFFFFFFFFFFFFFF) == uhr
            verified = bool(same_table and same_unitigs)
            del ra, qa, ba, ref
        del rv, qv, bv, resv
        barrier()

    # setup: one untimed call allocates the library's caching arena and the exchange buffer pool (tens of GB of
    # hipMalloc, ~2 s); from the second call on a step allocates nothing.  Then the W warm-up steps the caller asked for.
    res = step()
    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(dict(res.kernel_ms))
    barrier()
    elapsed = time.perf_counter() - t0
    inst_local = res.n_instances_input if hasattr(res, "n_instances_input") else res.n_instances
    if world > 1:
        t = torch.tensor([elapsed, float(inst_local)], dtype=torch.float64, device=cdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        inst_total = float(tsum[1])
    else:
        inst_total = float(inst_local)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = inst_total * args.steps / elapsed / 1e9
        # dominant kernel: the LDS count/filter kernel (one launch per step per GPU); units per launch =
        # the k-mer instances this rank's launch reduces
        count_ms = sum(k["count"] for k in kernel_ms) / len(kernel_ms)
        units = float(res.n_instances)
        achieved = units * ALG_BYTES_PER_KMER[K] / (count_ms * 1e-3) / 1e9
        # HBM bytes of one count-kernel launch from the PMC passes (profiles/traffic.json, keyed by reads / K / mode)
        traffic = ptraffic = None
        ent = {}
        tf = ROOT / "profiles" / "traffic.json"
        mode = "grouped" if args.grouped else ("sharded" if use_dist else "single")
        if tf.exists():
            try:
                ent = json.loads(tf.read_text()).get("entries", {}).get(f"{per_gpu}_k{K}_{mode}") or {}
                traffic = ent.get("count_kernel_hbm_bytes_per_launch")
                ptraffic = ent.get("partition_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic, ent = None, {}
        part_ms = (sum(k.get("partition", 0.0) for k in kernel_ms) / len(kernel_ms)) or None
        valu_insts = ent.get("count_kernel_SQ_INSTS_VALU_per_launch")
        salu_insts = ent.get("count_kernel_SQ_INSTS_SALU_per_launch")
        lds_insts = ent.get("count_kernel_SQ_INSTS_LDS_per_launch")
        # ---- every big kernel against its OWN bound (VERDICT r3 #1b).  The count kernel moves a tenth of SURVEY 8(d)'s write-once /
        # read-once bytes (supermers, not k-mer records, cross HBM): what binds it is VALU issue -- wave64 instructions (PMC
        # SQ_INSTS_VALU of the same workload, profiles/traffic.json) over the launch time measured here, against 1024 SIMD-32s x 2.4 GHz /
        # 2 cycles.  The partition kernel: its payload bytes against the streaming rate, and its slot reservations against the device's
        # atomic throughput (27 G/s, tools/probe/atomics.hip) -- the second one is the wall it stands at.
        kernels = {}
        if valu_insts:
            ach = valu_insts / (count_ms * 1e-3) / 1e9
            kernels["snk_count_kernel"] = {"bound": "valu", "launch_ms": count_ms, "valu_wave_insts": valu_insts, "achieved": ach, "peak": VALU_PEAK_GINST,
                                           "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINST,
                                           # (rounds 2-4 priced the same count against a 2-cycle issue, twice this peak: kept so that rounds stay comparable)
                                           "frac_at_2_cycle_issue_r4_definition": ach / (2 * VALU_PEAK_GINST),
                                           "valu_insts_per_kmer_instance": valu_insts * 64 / units if units else None,
                                           "salu_wave_insts": salu_insts, "lds_wave_insts": lds_insts,
                                           "hbm_traffic_bytes": traffic, "hbm_frac": (traffic / (count_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}
        if part_ms:
            n_super = float(getattr(res, "n_supermers", 0))
            payload = per_gpu * (int(rows.shape[1]) * 4 + 64) + n_super * 32      # packed rows + the last K quals of every read (64 B) + the 32-byte records
            kernels["snk_msp_kernel"] = {"bound": "atomics", "launch_ms": part_ms, "slot_reservations": n_super,
                                         "achieved": n_super / (part_ms * 1e-3) / 1e9, "peak": ATOMICS_PEAK_G, "unit": "G atomics/s",
                                         "frac": n_super / (part_ms * 1e-3) / 1e9 / ATOMICS_PEAK_G,
                                         "payload_bytes": payload, "stream_frac": payload / (part_ms * 1e-3) / 1e9 / HBM_STREAM_GBS,
                                         "hbm_traffic_bytes": ptraffic}
        dom = kernels.get("snk_count_kernel")
        out = {
            "metric": METRIC, "value": value, "unit": "Gk-mers/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{total_reads} x {sp.read_len} bp synthetic linked reads "
                                   f"({'error-free' if args.error_free else '0.2% substitutions, Q2 tails on 5%'}), "
                                   f"k={K}, {'1xMI355X count+graph' if world == 1 else f'{world}xMI355X minimiser-sharded all-to-all'}",
                       "reads_per_gpu": per_gpu, "k": K, "kmer_instances": int(inst_total),
                       "scratch_gb": round(getattr(res, "scratch_bytes", 0) / 2**30, 1), "overflow_supermers": int(getattr(res, "n_overflow", 0)),
                       "arena_reserved_gb": round(reserved_gb, 1),      # mapped before the first call (snk_ctx_reserve; --reserve-gb 0 = grow on demand)
                       "retained_kmers_rank0": int(res.n_kmers), "unitigs_rank0": int(res.n_unitigs),
                       "phase_ms_rank0": {k: round(v, 3) for k, v in res.phase_ms.items()},
                       "graph_ms_rank0": {k: round(v, 3) for k, v in getattr(res, "graph_ms", {}).items()},
                       "table_order": "key" if args.sorted_table else "bucket", "minimiser_len": 20 if (long_min and not args.grouped) else 16, "genome_len": int(sp.genome_len),
                       "fragments_rank0": int(getattr(res, "n_fragments", 0) or getattr(res, "n_frags", 0))},
            # THE figure is pipeline_frac: the whole step (value x SURVEY 8(d)'s algorithmic bytes) against the chips' HBM peak.
            # `bound` / `achieved` / `peak` / `frac` are SURVEY 8(d)'s formula for the dominant kernel, as the contract asks: algorithmic bytes of
            # the launch (32.65 B x the k-mer instances it reduces) / its HIP-event time against the HBM peak.  It says how fast the WORK goes
            # through that kernel, not that the kernel is near an HBM roofline: its real traffic (`traffic`, PMC) is a tenth of the algorithmic
            # bytes -- supermers, not k-mer records, cross HBM -- and what it runs into is instruction issue (`kernels.snk_count_kernel`: VALU
            # wave-instructions against 1024 SIMDs x 2.4 GHz / 4 cycles).  The whole step against the chip is `pipeline_frac`.
            "roofline": {"pipeline_frac": value * ALG_BYTES_PER_KMER[K] / (world * HBM_PEAK_GBS),
                         "bound": "hbm", "kernel": "snk_count_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": count_ms, "units_per_launch": units, "alg_bytes_per_unit": ALG_BYTES_PER_KMER[K],
                         "own_bound": ({"bound": "valu", "achieved": dom["achieved"], "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s", "frac": dom["frac"]} if dom else None),
                         "kernels": kernels,
                         "real_traffic_GBs": {"snk_count_kernel": (traffic / (count_ms * 1e-3) / 1e9) if traffic else None,
                                              "snk_msp_kernel": (ptraffic / (part_ms * 1e-3) / 1e9) if (ptraffic and part_ms) else None,
                                              "snk_msp_kernel_launch_ms": part_ms}},
        }
        out["config"]["path"] = "grouped-replicas" if args.grouped else ("sharded" if use_dist else "single")
        if use_dist and not args.grouped:
            # the multi-GPU breakdown of rank 0's last step: sections of the join, bytes this rank put on the links per exchange,
            # host read-backs of the step, ranks of the RCCL group
            out["config"]["multi_gpu"] = {
                "rccl_ranks": dist.get_world_size(), "transport": sh.kind, "host": "snk_shard_step (C++ behind the C ABI)",
                "host_syncs": int(res.host_syncs), "join": "owner", "join_ranking": res.join_ranking,
                "join_ms_rank0": {k: round(v, 3) for k, v in res.join_ms.items()},
                "exchange_bytes_rank0": res.exchange_bytes,
                "fragments_rank0": int(res.n_frags), "unitigs_written_by_rank0": int(res.n_unitigs)}
            if args.transport == "gloo":
                out["config"]["multi_gpu"]["note"] = ("functional run: exchanges staged through the host over gloo, ranks share the visible GPUs "
                                                      f"({torch.cuda.device_count()} for {world} ranks) -- not a performance number")
        if verified is not None:
            out["config"]["sharded_self_check"] = "passed" if verified else "FAILED"
            if not verified:        # a wrong result is not a performance number
                out["value"] = None
                out["invalid"] = "the sharded path's self-check against the one-GPU path failed"

        if world == 1 and not use_dist and not args.grouped:
            # nothing of the path is outside the timed step any more: a13 (.bv image in BVComp order, packed on the device) and a14 (the
            # graph from the unitigs) run inside it; only the optional key-sorted table (--sorted-table) is an extra
            out["config"]["step_includes"] = {"a13_bv_image_bytes": int(res.bv[1]), "a14_hbv_edges": int(res.hbv_graph["n_edges"]),
                                              "a14_hbv_vertices": int(res.hbv_graph["n_vertices"]),
                                              "a13_a14_ms": round(sum(tail_ms[-args.steps:]) / max(1, args.steps), 3)}
            if not args.no_robust and not args.error_free and per_gpu >= 10_000_000:
                try:
                    out["config"]["robust"] = robust_rows(eng, per_gpu, K, ms_per_step, int(getattr(res, "scratch_bytes", 0)))
                except Exception as ex:
                    out["config"]["robust"] = {"failed": str(ex)}
            if not args.no_next_rows:
                try:
                    out["config"]["next_rows"] = next_rows(eng, step(), rows, quals, bc, sp.read_len, K)
                except Exception as ex:
                    out["config"]["next_rows"] = {"failed": str(ex)}
            if not args.no_ingest:
                try:
                    out["config"]["f3_ingest"] = ingest_row(eng, args, K, ms_per_step / per_gpu)
                except Exception as ex:
                    out["config"]["f3_ingest"] = {"failed": str(ex)}
            if not args.no_df_seam:
                try:
                    out["config"]["df_seam"] = df_seam_row(eng, args, K)
                except Exception as ex:
                    out["config"]["df_seam"] = {"failed": str(ex)}
            if not args.no_robust and not args.error_free and per_gpu >= 10_000_000 and isinstance(out["config"].get("robust"), dict):
                try:
                    # the first call on a NEW context, same reads, right after the bench's own engine was closed: what it pays for is
                    # mostly the driver clearing the ~150 GB that engine has just handed back (~30 ms per GB; 0.19 s on a clean device:
                    # tools/first_call_probe.py, DESIGN 4 "round 4")
                    from supernova_amd.engine import Engine, Params
                    del res
                    eng.close()
                    torch.cuda.empty_cache()
                    e2 = Engine(local_rank)
                    ts = []
                    for rep in range(2):
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        r2 = e2.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=Params(K=K, sorted_table=False))
                        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
                    out["config"]["robust"]["new_context_after_close_calls_ms"] = ts
                    out["config"]["robust"]["new_context_note"] = ("first two calls of a NEW context right after the bench's own context handed its ~150 GB back: the "
                                                                   "driver clears freed memory before it hands it out again (~30 ms per GB); 0.19 s on a clean device "
                                                                   "(tools/first_call_probe.py, profiles/r04_first_call.log)")
                    del r2
                    e2.close()
                except Exception as ex:
                    out["config"]["robust"]["new_context_after_close_calls_ms"] = {"failed": str(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sp, K, args.cpu_sample, tuple(int(x) for x in args.cpu_threads.split(",")), big=args.cpu_sample_10m)
                c10 = ROOT / "profiles" / "cpu_baseline_10m.json"
                if args.cpu_sample_10m and out["cpu_baseline"].get("sample_10m"):
                    (ROOT / "gpurun_out").mkdir(exist_ok=True)
                    (ROOT / "gpurun_out" / "cpu_baseline_10m.json").write_text(json.dumps(out["cpu_baseline"]["sample_10m"], indent=1) + "\n")
                elif c10.exists() and K == 48:
                    # BASELINE config 1 (10 M reads) takes the reference about a minute per run: timed once per round on a GPU box
                    # (bench.py --cpu-sample-10m) and carried here from the committed record
                    out["cpu_baseline"]["sample_10m"] = dict(json.loads(c10.read_text()), cached_from="profiles/cpu_baseline_10m.json")
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "Gk-mers/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": f"failed: {ex}"}
        if (world == 1 and not use_dist and not args.grouped and K == 48 and not args.no_large_job and per_gpu == 100_000_000
                and not (args.no_robust or args.no_next_rows or args.no_ingest or args.no_df_seam or args.no_cpu_baseline)):
            # this run's reads, results and arena go back to the device first (the row's process plans with what is free)
            try:
                del res
            except NameError:
                pass
            try:
                del rows, quals, bc
            except NameError:
                pass
            try:
                eng.close()
            except Exception:
                pass
            out["config"]["large_job"] = large_job_row(args)
        line = json.dumps(out)
    if use_dist:
        if use_dist and not args.grouped and not args.no_verify:         # every rank leaves with the same exit code
            vt = torch.tensor([1 if verified else 0], dtype=torch.int64, device=cdev)
            dist.broadcast(vt, 0)
            verified = bool(int(vt.item()))
        dist.destroy_process_group()
    if rank == 0:
        # the contract line is the LAST thing on stdout: RCCL writes its version banner through C stdio, which would otherwise
        # be flushed behind it at exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)
    if verified is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
