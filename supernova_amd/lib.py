"""ctypes binding of libsnk.so (the C ABI declared in include/snk.h).

This is what a reference-side maintainer's ctypes/cffi stub would look like; the
package's host logic sits on top of it.  There is no CPU fallback: if the shared object is missing
or no gfx950 device is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os as _os

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(_os.environ.get("SNK_LIB_PATH", str(_HERE / "libsnk.so")))   # override = tuning builds (tools/build_variant.sh)


class SnkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsnk error {code}: {msg}")
        self.code = code


class SnkParams(C.Structure):
    _fields_ = [("K", C.c_uint32), ("min_qual", C.c_uint32), ("min_freq", C.c_uint32), ("min_bc", C.c_uint32),
                ("n_buckets", C.c_uint32), ("flags", C.c_uint32)]


class SnkSynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint64), ("genome_len", C.c_uint64), ("read_len", C.c_uint32),
                ("mol_len", C.c_uint32), ("mols_per_bc", C.c_uint32), ("pairs_per_bc", C.c_uint32),
                ("insert_min", C.c_uint32), ("insert_span", C.c_uint32), ("sub_ppm", C.c_uint32),
                ("unbarcoded_ppm", C.c_uint32), ("lowq_tail_ppm", C.c_uint32), ("tail_max", C.c_uint32),
                ("err_cdf", C.c_uint32 * 4), ("repeat_mode", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class SnkDevReads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("rows", C.c_void_p), ("row_words", C.c_uint32), ("read_len", C.c_uint32),
                ("lens", C.c_void_p), ("quals", C.c_void_p), ("qstride", C.c_uint32), ("reserved0", C.c_uint32),
                ("good_len", C.c_void_p), ("bc", C.c_void_p), ("ign_bc_below", C.c_int64),
                ("read_index_base", C.c_uint64), ("group", C.c_void_p)]


class SnkDevResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_instances", C.c_uint64), ("n_supermers", C.c_uint64),
                ("n_buckets", C.c_uint64), ("good_len", C.c_void_p), ("n_kmers", C.c_uint64), ("keys", C.c_void_p),
                ("counts", C.c_void_p), ("ctx", C.c_void_p), ("spectrum", C.c_void_p), ("spectrum_bins", C.c_uint32),
                ("n_circles", C.c_uint32), ("n_unitigs", C.c_uint64), ("unitig_total_bases", C.c_uint64),
                ("unitig_off", C.c_void_p), ("unitig_bases", C.c_void_p), ("rank_rounds", C.c_uint32),
                ("buckets_split", C.c_uint32), ("max_slots_used", C.c_uint32), ("n_overflow", C.c_uint32),
                ("scratch_bytes", C.c_uint64), ("phase_ms", C.c_float * 8), ("kernel_ms", C.c_float * 4),
                ("n_boundary", C.c_uint64), ("n_fragments", C.c_uint64), ("unitig_group", C.c_void_p),
                ("graph_ms", C.c_float * 8), ("repartitioned", C.c_uint32), ("n_hot_buckets", C.c_uint32)]


class SnkReads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("reserved", C.c_uint32), ("ascii", C.c_void_p),
                ("rows", C.c_void_p), ("lens", C.c_void_p), ("quals", C.c_void_p), ("good_len", C.c_void_p),
                ("bc", C.c_void_p), ("ign_bc_below", C.c_int64)]


class SnkResult(C.Structure):
    _fields_ = [("n_instances", C.c_uint64), ("n_kmers", C.c_uint64), ("kmers", C.POINTER(C.c_uint32)),
                ("counts", C.POINTER(C.c_uint32)), ("ctx", C.POINTER(C.c_uint8)), ("n_unitigs", C.c_uint64),
                ("unitig_off", C.POINTER(C.c_uint64)), ("unitig_bases", C.POINTER(C.c_uint8)),
                ("spectrum", C.POINTER(C.c_uint64)), ("spectrum_bins", C.c_uint32), ("reserved", C.c_uint32),
                ("phase_ms", C.c_float * 8), ("bv_image", C.POINTER(C.c_uint8)), ("bv_bytes", C.c_uint64)]


class SnkHbv(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_edges", C.c_int32), ("v_left", C.POINTER(C.c_int32)),
                ("v_right", C.POINTER(C.c_int32)), ("src_unitig", C.POINTER(C.c_int32)),
                ("is_rc", C.POINTER(C.c_uint8)), ("fwd_xlat", C.POINTER(C.c_int32)),
                ("rev_xlat", C.POINTER(C.c_int32)), ("bvcomp_order", C.POINTER(C.c_int32))]


class SnkDevPaths(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_edges_total", C.c_uint64), ("offset", C.c_void_p), ("n_edges", C.c_void_p),
                ("start", C.c_void_p), ("edges", C.c_void_p), ("dict_slots", C.c_uint64), ("dict_ms", C.c_float),
                ("path_ms", C.c_float), ("unitig_bc_off", C.c_void_p), ("unitig_bcs", C.c_void_p), ("n_unitig_bcs", C.c_uint64),
                ("bcs_ms", C.c_float), ("lookup_index", C.c_uint32), ("n_slow", C.c_uint64)]


class SnkDevDups(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("dup", C.c_void_p), ("n_placed", C.c_uint64), ("n_dup_reads", C.c_uint64),
                ("n_interdup_reads", C.c_uint64), ("n_dup_pairs", C.c_uint64), ("n_art_pairs", C.c_uint64),
                ("interdup_rate", C.c_double), ("ms", C.c_float)]


class SnkFasthBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("first_pair", C.c_uint64), ("file", C.c_uint32), ("max_len", C.c_uint32),
                ("ascii", C.c_void_p), ("quals", C.c_void_p), ("lens", C.c_void_p), ("bc_fields", C.c_void_p),
                ("text_bytes", C.c_uint64), ("token", C.c_uint64)]


class SnkDevIngest(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("row_words", C.c_uint32), ("qstride", C.c_uint32),
                ("max_len", C.c_uint32), ("rows", C.c_void_p), ("quals", C.c_void_p), ("lens", C.c_void_p), ("bc", C.c_void_p),
                ("text_bytes", C.c_uint64), ("compressed_bytes", C.c_uint64), ("seconds", C.c_double),
                ("decode_wait_seconds", C.c_double), ("n_files", C.c_uint32), ("n_batches", C.c_uint32), ("setup_seconds", C.c_double), ("good_len", C.c_void_p)]


class SnkTuning(C.Structure):
    _fields_ = [("count_kernel", C.c_uint32), ("count_tight_slots", C.c_uint32), ("count_screen_ratio_pct", C.c_uint32), ("target_inst", C.c_uint32),
                ("bucket_fill_pct", C.c_uint32), ("adaptive_buckets", C.c_uint32), ("chunk_kmers", C.c_uint32), ("minimiser_len", C.c_uint32),
                ("partition_passes", C.c_uint32), ("hot_buckets", C.c_uint32), ("hot_min", C.c_uint32), ("hot_factor", C.c_uint32),
                ("hot_class_inst", C.c_uint32), ("exchange_ranges", C.c_uint32), ("join_ranking", C.c_uint32), ("path_lookup", C.c_uint32),
                ("unitig_bc_cut", C.c_uint32), ("hbv_dev_min", C.c_uint32), ("hbv_big", C.c_uint32), ("reserved", C.c_uint32 * 9),
                ("last_count_kernel", C.c_uint32), ("last_count_limit", C.c_uint32), ("last_partition_passes", C.c_uint32),
                ("last_minimiser_len", C.c_uint32)]


class SnkDfInfo(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_barcodes", C.c_uint64), ("fastb_bytes", C.c_uint64), ("qualp_bytes", C.c_uint64),
                ("bci_bytes", C.c_uint64), ("reserved", C.c_uint64 * 3)]


class SnkShardResult(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("n_reads", C.c_uint64), ("n_instances", C.c_uint64),
                ("n_supermers", C.c_uint64), ("n_buckets_total", C.c_uint64), ("n_kmers", C.c_uint64), ("keys", C.c_void_p),
                ("counts", C.c_void_p), ("ctx", C.c_void_p), ("spectrum", C.c_void_p), ("spectrum_bins", C.c_uint32),
                ("n_circles", C.c_uint32), ("n_unitigs", C.c_uint64), ("unitig_total_bases", C.c_uint64),
                ("unitig_off", C.c_void_p), ("unitig_bases", C.c_void_p), ("unitig_circular", C.c_void_p),
                ("n_frags", C.c_uint64), ("n_frags_total", C.c_uint64), ("n_queries", C.c_uint64), ("n_link_queries", C.c_uint64),
                ("exchanged_bytes", C.c_uint64 * 8), ("host_syncs", C.c_uint32), ("ranking", C.c_uint32),
                ("buckets_split", C.c_uint32), ("max_slots_used", C.c_uint32), ("phase_ms", C.c_float * 8),
                ("join_ms", C.c_float * 8), ("count_kernel_ms", C.c_float), ("repartitioned", C.c_uint32), ("n_hot_buckets", C.c_uint32),
                ("reserved_u", C.c_uint32), ("pair_max_bytes", C.c_uint64 * 8)]


COMM_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                       C.POINTER(C.c_uint64), C.c_uint32)
COMM_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32)
RANGE_READY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)      # int ready(void* user, uint32_t range)

_lib = None


def load() -> C.CDLL:
    """Load libsnk.so (built in-tree by supernova_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -m supernova_amd.build` (hipcc, gfx950). "
            "supernova_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(str(LIB_PATH))
    _declare(lib)
    _lib = lib
    return lib


def _preload_hip_runtime() -> None:
    """libsnk.so leaves the hip* symbols undefined; bind them to the ONE HIP runtime of this process:
    torch's bundled libamdhip64 when torch is installed (so device pointers/streams are shared with torch),
    else the system ROCm one."""
    import os
    cands = []
    try:
        import torch  # noqa: F401  (loads its libamdhip64 first)
        cands.append(Path(torch.__file__).parent / "lib" / "libamdhip64.so")
    except Exception:  # pragma: no cover
        pass
    cands += [Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "lib" / "libamdhip64.so"]
    for c in cands:
        if c.exists():
            C.CDLL(str(c), mode=C.RTLD_GLOBAL)
            return
    raise FileNotFoundError("no libamdhip64.so found (torch bundle or $ROCM_PATH/lib)")


def _declare(lib: C.CDLL) -> None:
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    cp, sz = C.c_char_p, C.c_size_t
    P = C.POINTER
    sig = {
        "snk_version": (cp, []),
        "snk_last_error": (cp, []),
        "snk_params_default": (None, [P(SnkParams)]),
        "snk_ctx_create": (C.c_int, [C.c_int, P(vp), cp, sz]),
        "snk_ctx_destroy": (None, [vp]),
        "snk_ctx_trim": (None, [vp]),
        "snk_ctx_reserve": (C.c_int, [vp, u64, cp, sz]),
        "snk_tuning_default": (None, [P(SnkTuning)]),
        "snk_ctx_set_tuning": (C.c_int, [vp, P(SnkTuning), cp, sz]),
        "snk_ctx_get_tuning": (None, [vp, P(SnkTuning)]),
        "snk_ctx_set_option": (C.c_int, [vp, cp, C.c_longlong, cp, sz]),
        "snk_ctx_clear_option": (C.c_int, [vp, cp]),
        "snk_ctx_get_option": (C.c_int, [vp, cp, P(C.c_longlong)]),
        "snk_option_name": (cp, [u32]),
        "snk_option_doc": (cp, [u32]),
        "snk_synth_default": (None, [P(SnkSynthParams), u64, u64, C.c_int]),
        "snk_synth_set_errors": (None, [P(SnkSynthParams), u32]),
        "snk_synth_host": (C.c_int, [P(SnkSynthParams), u64, u64, vp, u32, vp, u32, vp]),
        "snk_synth_dev": (C.c_int, [vp, P(SnkSynthParams), u64, u64, vp, u32, vp, u32, vp, vp]),
        "snk_dev_trim": (C.c_int, [vp, vp, u32, vp, u32, u64, u32, u32, vp, vp]),
        "snk_dev_pack_ascii": (C.c_int, [vp, vp, u32, u32, u64, vp, u32, vp]),
        "snk_dev_count_graph": (C.c_int, [vp, P(SnkDevReads), P(SnkParams), P(SnkDevResult), vp, cp, sz]),
        "snk_dev_stream_begin": (C.c_int, [vp, P(SnkParams), u32, u64, C.c_int, vp, cp, sz]),
        "snk_dev_stream_append": (C.c_int, [vp, P(SnkDevReads), vp, cp, sz]),
        "snk_dev_stream_finish": (C.c_int, [vp, P(SnkDevResult), vp, cp, sz]),
        "snk_dev_download": (C.c_int, [vp, vp, vp, sz, vp]),
        "snk_count_graph": (C.c_int, [vp, P(SnkReads), P(SnkParams), P(SnkResult), cp, sz]),
        "snk_free": (None, [P(SnkResult)]),
        "snk_host_alloc_pinned": (C.c_int, [sz, P(vp), cp, sz]),
        "snk_host_free_pinned": (None, [vp]),
        "snk_write_bv": (C.c_int, [cp, u64, vp, vp, cp, sz]),
        "snk_read_bv": (C.c_int, [cp, P(u64), P(P(u64)), P(P(C.c_uint8)), cp, sz]),
        "snk_hbv_from_unitigs": (C.c_int, [u32, u64, vp, vp, P(SnkHbv), cp, sz]),
        "snk_dev_hbv": (C.c_int, [vp, u32, u64, vp, vp, P(SnkHbv), P(C.c_float), vp, cp, sz]),
        "snk_dev_bv_image": (C.c_int, [vp, u32, u64, vp, vp, C.c_int, P(C.c_void_p), P(u64), vp, cp, sz]),
        "snk_hbv_free": (None, [P(SnkHbv)]),
        "snk_dev_path_reads": (C.c_int, [vp, u32, P(SnkDevReads), u64, vp, vp, P(SnkHbv), P(SnkDevPaths), vp, cp, sz]),
        "snk_dev_path_reads2": (C.c_int, [vp, u32, P(SnkDevReads), u64, vp, vp, P(SnkHbv), u32, P(SnkDevPaths), vp, cp, sz]),
        "snk_dev_mark_dups": (C.c_int, [vp, P(SnkDevReads), P(SnkDevPaths), P(SnkDevDups), vp, cp, sz]),
        "snk_hbv_involution": (C.c_int, [P(SnkHbv), u64, vp, cp, sz]),
        "snk_write_hbv": (C.c_int, [cp, cp, u32, u64, vp, vp, P(SnkHbv), cp, sz]),
        "snk_read_fastb": (C.c_int, [cp, P(u64), P(u32), P(P(C.c_uint16)), P(P(u32)), cp, sz]),
        "snk_read_qualp": (C.c_int, [cp, u64, u32, vp, cp, sz]),
        "snk_read_bci": (C.c_int, [cp, u64, vp, P(u64), cp, sz]),
        "snk_read_fasth": (C.c_int, [cp, u32, P(u64), P(u32), P(P(C.c_uint8)), P(P(C.c_uint8)), P(P(C.c_uint16)), P(P(C.c_uint8)), cp, sz]),
        "snk_host_free": (None, [vp]),
        "snk_bc_index_create": (C.c_int, [vp, cp, sz, P(vp), cp, sz]),
        "snk_bc_index_destroy": (None, [vp]),
        "snk_bc_index_lines": (u32, [vp]),
        "snk_dev_bc_ids": (C.c_int, [vp, vp, vp, u32, u64, vp, vp, cp, sz]),
        "snk_pack2_bytes": (u64, [u64]),
        "snk_dev_pack2": (C.c_int, [vp, vp, u64, vp, vp]),
        "snk_dev_unpack2": (C.c_int, [vp, vp, u64, vp, vp]),
        "snk_host_cpu_budget": (u32, []),
        "snk_ctx_last_partition_passes": (u32, [vp]),
        "snk_ctx_last_count_limit": (u32, [vp]),
        "snk_fasth_open": (C.c_int, [P(cp), u32, u32, u32, u32, u32, P(vp), cp, sz]),
        "snk_fasth_next": (C.c_int, [vp, P(SnkFasthBatch), cp, sz]),
        "snk_fasth_release": (None, [vp, P(SnkFasthBatch)]),
        "snk_fasth_file_pairs": (u64, [vp, u32]),
        "snk_fasth_close": (None, [vp]),
        "snk_dev_ingest_fasth": (C.c_int, [vp, P(cp), u32, u32, vp, u32, u32, P(SnkDevIngest), cp, sz]),
        "snk_dev_ingest_free": (None, [P(SnkDevIngest)]),
        "snk_dev_ingest_count_graph": (C.c_int, [vp, P(cp), u32, u32, vp, u32, u32, u64, P(SnkParams), P(SnkDevResult), P(SnkDevIngest), cp, sz]),
        "snk_df_open": (C.c_int, [cp, cp, cp, P(vp), P(SnkDfInfo), cp, sz]),
        "snk_df_close": (None, [vp]),
        "snk_df_max_len": (C.c_int, [vp, vp, u64, u64, P(u32), cp, sz]),
        "snk_dev_ingest_df": (C.c_int, [vp, vp, u64, u64, u32, u32, u64, P(SnkDevIngest), cp, sz]),
        "snk_dev_ingest_df_trimmed": (C.c_int, [vp, vp, u64, u64, u32, u32, u64, u32, u32, P(SnkDevIngest), cp, sz]),
        "snk_dev_ingest_df_count_graph": (C.c_int, [vp, vp, u64, u64, u32, u32, u64, P(SnkParams), C.c_int64, P(SnkDevResult), P(SnkDevIngest), cp, sz]),
        "snk_write_df": (C.c_int, [cp, u64, vp, u32, vp, u32, vp, u32, vp, u32, u64, cp, sz]),
        "snk_synth_df_write": (C.c_int, [cp, P(SnkSynthParams), u64, u64, u32, u32, cp, sz]),
        "snk_synth_fasth_write": (C.c_int, [cp, P(SnkSynthParams), u64, u64, C.c_int, P(u64), cp, sz]),
        "snk_synth_bc_seq": (None, [u32, vp]),
        "snk_comm_set_rccl_path": (C.c_int, [cp]),
        "snk_comm_unique_id": (C.c_int, [vp, cp, sz]),
        "snk_comm_create_rccl": (C.c_int, [vp, vp, u32, u32, P(vp), cp, sz]),
        "snk_comm_from_nccl": (C.c_int, [vp, vp, u32, u32, P(vp), cp, sz]),
        "snk_comm_create_local": (C.c_int, [u32, P(vp), cp, sz]),
        "snk_comm_create_callbacks": (C.c_int, [u32, u32, COMM_A2A, COMM_GATHER, vp, P(vp), cp, sz]),
        "snk_comm_selftest": (C.c_int, [vp, u64, u32, u32, cp, sz]),
        "snk_comm_destroy": (None, [vp]),
        "snk_comm_abort": (None, [vp]),
        "snk_comm_rank": (u32, [vp]),
        "snk_comm_world": (u32, [vp]),
        "snk_comm_kind": (cp, [vp]),
        "snk_shard_step": (C.c_int, [vp, vp, P(SnkDevReads), P(SnkParams), u64, u32, P(SnkShardResult), vp, cp, sz]),
        "snk_shard_stream_begin": (C.c_int, [vp, vp, P(SnkParams), u32, u64, u64, C.c_int, vp, cp, sz]),
        "snk_shard_stream_append": (C.c_int, [vp, P(SnkDevReads), vp, cp, sz]),
        "snk_shard_stream_finish": (C.c_int, [vp, vp, u32, P(SnkShardResult), vp, cp, sz]),
        "snk_shard_gather_unitigs": (C.c_int, [vp, vp, P(SnkShardResult), u32, u32, u32, P(SnkResult), vp, cp, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._snk_declared = tuple(sig)


def check(code: int) -> None:
    if code != 0:
        raise SnkError(code, load().snk_last_error().decode(errors="replace"))


def exported_symbols() -> tuple[str, ...]:
    return load()._snk_declared
