"""ctypes binding of libnmscan.so (C ABI: include/nmscan.h).

The library is built in-tree by ``nanomotif_amd.build.build()`` (hipcc, gfx950) and must be present:
there is NO CPU fallback — importing the product path without it raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NM_LIB", os.path.join(_HERE, "libnmscan.so"))   # NM_LIB: A/B kernel experiments

SYMBOLS = [
    "nm_abi_version", "nm_last_error", "nm_set_device_allocator", "nm_ctx_create", "nm_ctx_destroy", "nm_set_stream", "nm_set_score_lanes", "nm_sync", "nm_upload_contigs",
    "nm_upload_contigs_device", "nm_upload_pileup", "nm_upload_pileup_device", "nm_score_batch", "nm_score_batch_device", "nm_score_batch_wide", "nm_block_cache", "nm_hit_positions", "nm_stats",
    "nm_last_kernel_ms", "nm_timing_reset", "nm_timing_total_ms", "nm_timing_intervals", "nm_warm_file_parsers", "nm_parse_motifs",
    "nm_win_clear", "nm_win_add_task", "nm_win_batch", "nm_win_batch_w", "nm_win_add_task_rows", "nm_win_add_task_contigs", "nm_plan_windows", "nm_methylated_row_counts", "nm_contig_base_counts", "nm_bg_counts", "nm_bg_counts_runs", "nm_assembly_other_letters", "nm_ingest_pileup", "nm_ingest_pileup_part", "nm_ingest_results", "nm_py_random_sample", "nm_py_random_sample_many", "nm_py_random_sample_groups", "nm_window_letter_counts", "nm_bed_open", "nm_bed_open_indexed", "nm_bed_shape", "nm_bed_contig_name", "nm_bed_mod_code", "nm_bed_columns", "nm_bed_ingest_columns", "nm_bed_close", "nm_fasta_open", "nm_fasta_shape", "nm_fasta_record", "nm_fasta_sequence", "nm_fasta_close",
    "nm_comm_unique_id", "nm_comm_init", "nm_allreduce_counts", "nm_allreduce_counts_async", "nm_comm_wait", "nm_allreduce_counts_host",
    "nm_comm_sync", "nm_comm_info", "nm_comm_destroy",
    "nm_score_batch_per_contig", "nm_bin_contigs", "nm_readstats_upload", "nm_contig_methylation", "nm_bed_open_counts", "nm_bed_count_columns", "nm_bed_parse_device", "nm_bed_parse_device_indexed", "nm_bedcols_shape", "nm_bedcols_contig_name", "nm_bedcols_mod_code", "nm_bedcols_runs", "nm_bedcols_map_contigs", "nm_bedcols_device_columns", "nm_bedcols_close", "nm_device_read", "nm_score_batch_begin", "nm_score_batch_end", "nm_win_batch_w_begin", "nm_win_batch_w_end", "nm_search_run", "nm_search_run_custom", "nm_search_result_sizes", "nm_search_result_speculation", "nm_search_result_export", "nm_search_result_gml", "nm_search_result_free", "nm_post_run", "nm_post_run_custom", "nm_post_run_rows_custom", "nm_post_sizes", "nm_post_export", "nm_post_tables", "nm_post_free", "nm_psi_posint",
    "nm_bedcols_phase_seconds", "nm_bed_plan_indexed", "nm_bed_parse_device_planned", "nm_bedplan_close", "nm_fasta_parse_device", "nm_fastadev_shape", "nm_fastadev_record", "nm_fastadev_table", "nm_fastadev_sequence_device", "nm_upload_contigs_fasta", "nm_fastadev_close",
    "nm_tabix_regions",
]

class SearchParams(C.Structure):
    """include/nmscan.h: nm_search_params."""
    _fields_ = [("padding", C.c_uint32), ("max_dead_ends", C.c_uint32), ("max_rounds_since_new_best", C.c_uint32),
                ("max_motif_length", C.c_uint32), ("min_kl", C.c_double), ("score_threshold", C.c_double),
                ("remaining_threshold", C.c_double), ("freq_threshold", C.c_double)]


ALLOC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t)
FREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
SEARCH_SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_char), C.POINTER(C.c_int64))
SEARCH_WINDOW_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_char),
                               C.POINTER(C.c_int32))
SEARCH_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_uint64)
POST_SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_char), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                            C.POINTER(C.c_int64))

_lib = None
early_engine_thread = None     # __main__.py: the thread that creates the engine context while the interpreter imports ...
early_pin_thread = None        # ... the thread that pins the file parsers' host buffers meanwhile (nm_warm_file_parsers) ...
early_engine = None            # ... and what it made: (device, nm_ctx *, block cache installed); main.find_motifs_bin adopts it
loaded_with_torch = False      # torch's HIP runtime was in the process when the library was loaded (one runtime for both)


class NmScanError(RuntimeError):
    """An error of libnmscan; ``code``: its nm_status (include/nmscan.h) when the library returned one."""
    code = None


def text_of(raw: bytes, what: str) -> str:
    """A name the library read from a file, as text.  polars (the reference's reader, dataload.py:72-100) refuses a file that
    is not UTF-8; so does this, with the place named."""
    try:
        return raw.decode()
    except UnicodeDecodeError:
        raise NmScanError(f"{what} is not valid UTF-8: {raw[:60]!r}") from None


def _cli_process() -> bool:
    """True when this interpreter was started as ``python -m nanomotif_amd`` (the only torch-free entry point)."""
    import sys
    main = sys.modules.get("__main__")
    return getattr(getattr(main, "__spec__", None), "name", "") in ("nanomotif_amd.__main__", "nanomotif_amd")


_load_lock = __import__("threading").Lock()


def load():
    """Load libnmscan.so (after torch, so both share one HIP runtime) and declare prototypes."""
    if _lib is not None:
        return _lib
    with _load_lock:                        # (the command line's warm-up thread and the main thread may both come here first)
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NmScanError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nanomotif_amd has no CPU fallback.")
    # In a process that uses torch (tests, bench, multi-rank runs) torch's bundled libamdhip64 must be loaded first so
    # that both share ONE HIP runtime (the SONAME matches ours).  A single-GPU CLI run never touches torch and loads
    # the system runtime through the library's RUNPATH instead — a second of start-up saved.
    import sys
    if "torch" in sys.modules or os.environ.get("WORLD_SIZE", "1") != "1" or os.environ.get("NANOMOTIF_WITH_TORCH") == "1" \
            or not _cli_process():
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing only
            pass
    global loaded_with_torch
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as first:
        # no system HIP runtime reachable through the library's RUNPATH (a host that only has torch's bundled ROCm, or
        # a library built elsewhere): torch's libamdhip64 has the same SONAME — load it first and try once more
        try:
            import torch  # noqa: F401
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except Exception as second:
            raise NmScanError(
                f"cannot load {LIB_PATH}: {first}.  It needs the ROCm HIP runtime (libamdhip64.so): install ROCm under "
                f"/opt/rocm (or set ROCM_PATH when building), or make PyTorch-ROCm importable (retry after importing torch: "
                f"{second}).  nanomotif_amd has no CPU fallback.") from first
    loaded_with_torch = "torch" in sys.modules
    p = C.c_void_p
    u8p, u32p, u64p, i64p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int64, C.c_double))
    lib.nm_abi_version.restype = C.c_int
    lib.nm_last_error.restype = C.c_char_p
    lib.nm_ctx_create.argtypes = [C.c_int, C.POINTER(p)]
    lib.nm_set_device_allocator.argtypes = [ALLOC_FN, FREE_FN, p]
    lib.nm_block_cache.argtypes = [C.c_int, C.c_uint64, u64p]
    lib.nm_ctx_destroy.argtypes = [p]
    lib.nm_set_stream.argtypes = [p, p]
    lib.nm_set_score_lanes.argtypes = [p, C.c_int]
    lib.nm_sync.argtypes = [p]
    lib.nm_upload_contigs.argtypes = [p, C.c_uint32, u64p, u32p, C.c_uint32, u8p]
    lib.nm_upload_contigs_device.argtypes = [p, C.c_uint32, u64p, u32p, C.c_uint32, p]
    lib.nm_upload_pileup.argtypes = [p, C.c_uint32, C.c_uint8, C.c_double, C.c_double, C.c_uint64, u32p, u32p, u8p,
                                     f64p, C.c_int]
    lib.nm_upload_pileup_device.argtypes = [p, C.c_uint32, C.c_uint8, C.c_double, C.c_double, C.c_uint64, p, p, p, p,
                                            C.c_int]
    for name in ("nm_score_batch", "nm_score_batch_device"):
        getattr(lib, name).argtypes = [p, C.c_uint32, u32p, u8p, u8p, u8p, u32p, u8p, p]
    lib.nm_score_batch_wide.argtypes = [p, C.c_uint32, u32p, u8p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), u32p, u8p, i64p]
    lib.nm_score_batch_begin.argtypes = [p, C.c_uint32, u32p, u8p, u8p, u8p, u32p, u8p]
    lib.nm_score_batch_end.argtypes = [p, i64p]
    lib.nm_win_batch_w_begin.argtypes = [p, C.c_uint32, u32p, u8p, u8p, C.c_uint32]
    lib.nm_win_batch_w_end.argtypes = [p, C.POINTER(C.c_int32)]
    lib.nm_score_batch_per_contig.argtypes = [p, C.c_uint32, u32p, u8p, u8p, u8p, u32p, u8p, u64p, i64p]
    lib.nm_bin_contigs.argtypes = [p, C.c_uint32, u32p, C.c_uint32, u32p]
    lib.nm_readstats_upload.argtypes = [p, C.c_uint32, C.c_uint64, p, p, p, p, p, p, C.c_int32, C.c_double, C.c_int, u64p]
    lib.nm_contig_methylation.argtypes = [p, C.c_uint32, u8p, u8p, u8p, u32p, u8p, u32p, f64p, f64p, f64p]
    lib.nm_bed_open_counts.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(p)]
    lib.nm_bed_count_columns.argtypes = [p, C.POINTER(p), C.POINTER(p)]
    lib.nm_bed_parse_device.argtypes = [p, C.c_char_p, C.c_uint32, C.POINTER(p)]
    lib.nm_bed_parse_device_indexed.argtypes = [p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(p), C.POINTER(C.c_uint64)]
    lib.nm_bedcols_shape.argtypes = [p, u64p, u32p, u32p, f64p]
    lib.nm_bedcols_contig_name.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p)]
    lib.nm_bedcols_mod_code.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p)]
    lib.nm_bedcols_runs.argtypes = [p, u64p, u32p]
    lib.nm_bedcols_map_contigs.argtypes = [p, u32p, C.c_uint32]
    lib.nm_bedcols_device_columns.argtypes = [p] + [C.POINTER(p)] * 7
    lib.nm_bedcols_close.argtypes = [p]
    lib.nm_bedcols_phase_seconds.argtypes = [p, f64p]
    lib.nm_bed_plan_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(p), C.POINTER(C.c_uint64)]
    lib.nm_bed_parse_device_planned.argtypes = [p, p, C.c_uint32, C.POINTER(p)]
    lib.nm_bedplan_close.argtypes = [p]
    lib.nm_device_read.argtypes = [p, p, p, C.c_uint64]
    lib.nm_fasta_parse_device.argtypes = [p, C.c_char_p, C.c_uint32, C.POINTER(p)]
    lib.nm_fastadev_shape.argtypes = [p, u32p, u64p, f64p]
    lib.nm_fastadev_record.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p), u64p, u64p]
    lib.nm_fastadev_sequence_device.argtypes = [p, C.POINTER(p)]
    lib.nm_fastadev_table.argtypes = [p, C.POINTER(p), u64p, C.POINTER(p)]
    lib.nm_upload_contigs_fasta.argtypes = [p, p, C.c_uint32, u32p, u32p, C.c_uint32]
    lib.nm_fastadev_close.argtypes = [p]
    lib.nm_hit_positions.argtypes = [p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, u8p, C.c_int, i64p, C.c_uint64,
                                     u64p]
    lib.nm_stats.argtypes = [p, u64p]
    lib.nm_last_kernel_ms.argtypes = [p, C.POINTER(C.c_float)]
    lib.nm_parse_motifs.argtypes = [C.c_uint32, C.c_char_p, u32p, C.POINTER(C.c_int32), u8p, u8p, u32p, u8p, C.c_uint64,
                                    u64p]
    lib.nm_win_clear.argtypes = [p]
    lib.nm_win_add_task.argtypes = [p, C.c_uint32, C.c_uint32, u8p, u32p]
    lib.nm_win_batch.argtypes = [p, C.c_uint32, u32p, u8p, u8p, C.POINTER(C.c_int32)]
    lib.nm_win_batch_w.argtypes = [p, C.c_uint32, u32p, u8p, u8p, C.c_uint32, C.POINTER(C.c_int32)]
    lib.nm_win_add_task_rows.argtypes = [p, C.c_uint32, u32p, u32p, u8p, C.c_uint32, u32p]
    lib.nm_win_add_task_contigs.argtypes = [p, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p, u64p]
    lib.nm_plan_windows.argtypes = [p, C.c_uint32, u32p, u8p, u32p, u32p, u32p, C.c_uint32, C.c_double, C.c_uint32, u32p, C.c_int, u8p, u32p,
                                    u64p, u64p, i64p, u32p]
    lib.nm_methylated_row_counts.argtypes = [p, C.c_uint32, C.c_uint32, u64p]
    lib.nm_contig_base_counts.argtypes = [p, C.c_uint8, C.c_uint32, u64p]
    lib.nm_bg_counts.argtypes = [p, C.c_uint8, C.c_uint32, C.c_uint64, u32p, u32p, C.c_uint32, u64p, i64p]
    lib.nm_bg_counts_runs.argtypes = [p, C.c_uint8, C.c_uint32, C.c_uint32, u32p, u32p, u32p, C.c_uint32, u32p, i64p]
    lib.nm_assembly_other_letters.argtypes = [p, u64p]
    lib.nm_ingest_pileup.argtypes = [p, C.c_uint64, p, p, p, p, p, p, C.POINTER(C.c_int32), u8p, C.c_double, C.c_double,
                                     C.c_int, u64p, u64p]
    lib.nm_ingest_pileup_part.argtypes = [p, C.c_uint64, p, p, p, p, p, p, C.POINTER(C.c_int32), u8p, C.c_double, C.c_double,
                                          C.c_int, C.c_int, C.c_uint32, u32p, u64p, u64p]
    lib.nm_ingest_results.argtypes = [p, u32p, u32p, u8p, C.POINTER(C.c_int8), C.c_uint64, u32p]
    lib.nm_py_random_sample.argtypes = [u32p, C.c_uint64, C.c_uint64, u32p]
    lib.nm_py_random_sample_many.argtypes = [u32p, C.c_uint32, u64p, u64p, u32p]
    lib.nm_py_random_sample_groups.argtypes = [C.c_uint32, u32p, u64p, u64p, u64p, u32p, u32p]
    lib.nm_window_letter_counts.argtypes = [u8p, C.c_uint64, i64p, C.c_uint64, C.c_uint32, i64p]
    lib.nm_bed_open.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(p)]
    lib.nm_bed_open_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, u32p, C.c_uint32, C.POINTER(p), u64p]
    lib.nm_tabix_regions.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, u32p, u64p, u64p, u8p]
    lib.nm_bed_shape.argtypes = [p, u64p, u32p]
    lib.nm_bed_contig_name.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p)]
    lib.nm_bed_mod_code.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p)]
    lib.nm_bed_columns.argtypes = [p] + [C.POINTER(p)] * 6
    lib.nm_bed_ingest_columns.argtypes = [p, u32p, C.c_uint32] + [C.POINTER(p)] * 6
    lib.nm_bed_close.argtypes = [p]
    lib.nm_fasta_open.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(p)]
    lib.nm_fasta_shape.argtypes = [p, u32p, u64p]
    lib.nm_fasta_record.argtypes = [p, C.c_uint32, C.POINTER(C.c_char_p), u64p, u64p]
    lib.nm_fasta_sequence.argtypes = [p, C.POINTER(p)]
    lib.nm_fasta_close.argtypes = [p]
    lib.nm_timing_reset.argtypes = [p, C.c_int]
    lib.nm_search_run.argtypes = [p, C.c_uint32, u32p, u32p, u32p, C.POINTER(SearchParams), f64p, u64p, u8p, SEARCH_REDUCE_FN, p, C.POINTER(p)]
    lib.nm_search_run_custom.argtypes = [C.c_uint32, C.POINTER(SearchParams), f64p, u64p, u8p, SEARCH_SCORE_FN, SEARCH_WINDOW_FN, p, C.POINTER(p)]
    lib.nm_search_result_sizes.argtypes = [p, u64p, u64p, u64p, u64p]
    lib.nm_search_result_speculation.argtypes = [p, u64p]
    lib.nm_search_result_export.argtypes = [p, u64p, u64p, u64p, u8p, C.c_char_p, i64p, f64p, f64p, C.POINTER(C.c_int32), u8p,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nm_search_result_free.argtypes = [p]
    lib.nm_post_run.argtypes = [p, p, u32p, u32p, SEARCH_REDUCE_FN, p, C.POINTER(p)]
    lib.nm_post_run_custom.argtypes = [p, POST_SCORE_FN, p, C.POINTER(p)]
    lib.nm_post_run_rows_custom.argtypes = [C.c_uint32, C.c_uint32, u64p, C.c_char_p, i64p, f64p, POST_SCORE_FN, p, C.POINTER(p)]
    lib.nm_post_sizes.argtypes = [p, u64p, u64p, u64p]
    lib.nm_search_result_gml.argtypes = [p, C.POINTER(C.c_void_p), C.POINTER(u64p), u64p]
    lib.nm_post_tables.argtypes = [p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(u64p), u64p]
    lib.nm_post_export.argtypes = [p, u32p, u8p, u64p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), f64p, C.POINTER(C.c_int64)]
    lib.nm_post_free.argtypes = [p]
    lib.nm_psi_posint.argtypes = [C.c_int64, f64p]
    lib.nm_comm_unique_id.argtypes = [u8p]
    lib.nm_comm_init.argtypes = [p, C.c_int, C.c_int, u8p]
    lib.nm_allreduce_counts.argtypes = [p, p, C.c_uint64]
    lib.nm_allreduce_counts_async.argtypes = [p, p, C.c_uint64, C.c_int]
    lib.nm_comm_wait.argtypes = [p, C.c_int]
    lib.nm_allreduce_counts_host.argtypes = [p, i64p, C.c_uint64]
    lib.nm_comm_sync.argtypes = [p]
    lib.nm_comm_info.argtypes = [p, C.POINTER(C.c_int32)]
    lib.nm_comm_destroy.argtypes = [p]
    lib.nm_timing_total_ms.argtypes = [p, C.POINTER(C.c_double), u64p]
    lib.nm_warm_file_parsers.argtypes = [p, C.c_uint64, C.c_uint32]
    lib.nm_timing_intervals.argtypes = [p, p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p]
    for s in SYMBOLS:
        if s != "nm_last_error":
            getattr(lib, s).restype = C.c_int
    _lib = lib
    return lib


_torch_pool = None


def use_block_cache(max_idle_bytes: int = 16 << 30):
    """The library's own block cache behind its device allocations (nm_block_cache) — for a process without a pool of its own (the
    command line).  Call before the first engine is created; ``max_idle_bytes`` 0 uninstalls it."""
    lib = load()
    check(lib.nm_block_cache(1 if max_idle_bytes > 0 else 0, int(max_idle_bytes), None))


def block_cache_stats() -> dict:
    stats = (C.c_uint64 * 4)()
    check(load().nm_block_cache(-1, 0, stats))
    return dict(served_from_cache=int(stats[0]), went_to_hipmalloc=int(stats[1]), idle_bytes=int(stats[2]), blocks_in_use=int(stats[3]))


def use_torch_allocator(enable: bool = True):
    """Serve the library's device allocations from torch's caching allocator (nm_set_device_allocator): for processes
    that hold a torch pool anyway (bench.py, the synthetic end-to-end runs).  Call before the first engine is created
    and switch it off only after the last one is closed."""
    global _torch_pool
    lib = load()
    if not enable:
        check(lib.nm_set_device_allocator(C.cast(None, ALLOC_FN), C.cast(None, FREE_FN), None))
        _torch_pool = None
        return
    import torch

    def _alloc(_user, out, nbytes):
        try:
            out[0] = torch.cuda.caching_allocator_alloc(int(nbytes))
            return 0
        except Exception:
            return -4

    def _free(_user, ptr):
        try:
            torch.cuda.caching_allocator_delete(ptr)
            return 0
        except Exception:
            return -1
    _torch_pool = (ALLOC_FN(_alloc), FREE_FN(_free))              # kept alive: the library calls them
    check(lib.nm_set_device_allocator(_torch_pool[0], _torch_pool[1], None))


NM_EINDEX = -6          # include/nmscan.h: the tabix index cannot be used with this pileup (read the whole file)
NM_EDECLINED = -7       # a device parser declines an input its host twin reads (use nm_bed_open / nm_fasta_open)
NM_ESEQUENCE = -8       # a FASTA record is empty or holds a letter outside the IUPAC nucleotides


def check(rc: int):
    if rc != 0:
        e = NmScanError(f"libnmscan error {rc}: {load().nm_last_error().decode()}")
        e.code = int(rc)
        raise e
