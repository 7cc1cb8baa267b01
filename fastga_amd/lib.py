"""ctypes binding of libfastga_amd.so (include/fastga_amd.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FgaError(RuntimeError):
    pass


def lib_path():
    # FGA_LIBRARY: another build of this same library (kernel A/B experiments, tools/merge_knockout.sh)
    return os.environ.get("FGA_LIBRARY") or os.path.join(_HERE, "libfastga_amd.so")


def load_library():
    """Load the in-tree shared library; raise loudly if it has not been built (no fallback).

    One HIP runtime per process: the PyTorch wheel brings its own libamdhip64 / libhsa-runtime64, and a process that
    first loads this library (ROCm's runtime under /opt/rocm) and then initialises torch.cuda ends up with two HSA
    runtimes -- torch then reports "No HIP GPUs are available".  A process that needs torch.cuda beside this library
    (bench.py under torchrun: RCCL collectives on device buffers) must therefore import torch FIRST; the library's
    libamdhip64.so.7 dependency then resolves to the runtime torch loaded and device pointers are interchangeable.
    FGA_TORCH_RUNTIME=1 makes this function do that import itself."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if os.environ.get("FGA_TORCH_RUNTIME") == "1":
        import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise FgaError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C fastga_amd/csrc`")
    L = C.CDLL(path)
    L.fga_last_error.restype = C.c_char_p
    _declare(L)
    _LIB = L
    return L


def check(status, what=""):
    if status != 0:
        L = load_library()
        msg = L.fga_last_error().decode(errors="replace")
        raise FgaError(f"{what}: {msg}" if what else msg)


class MergeParams(C.Structure):
    _fields_ = [("freq", C.c_int), ("soft_mask", C.c_int), ("flip", C.c_int),
                ("prefix_begin", C.c_int64), ("prefix_end", C.c_int64)]


class ChainParams(C.Structure):
    _fields_ = [("chain_break", C.c_int64), ("chain_min", C.c_int64), ("amxpos", C.c_int64),
                ("bmxpos", C.c_int64), ("alen", C.c_void_p), ("nalen", C.c_int64)]


class Hits(C.Structure):
    _fields_ = [("nhits", C.c_int64), ("nunits", C.c_int64), ("hits", C.c_void_p), ("units", C.c_void_p)]


class ExtendParams(C.Structure):
    _fields_ = [("tspace", C.c_int), ("path_ave", C.c_int), ("table", C.c_void_p), ("score", C.c_void_p),
                ("self", C.c_int), ("aln_min", C.c_int), ("aln_rate", C.c_double),
                ("cell_cap", C.c_int64), ("aln_cap", C.c_int64), ("trace_cap", C.c_int64)]


class Alns(C.Structure):
    _fields_ = [("naln", C.c_int64), ("ntrace", C.c_int64), ("ncalls", C.c_int64), ("nwaves", C.c_int64),
                ("alns", C.c_void_p), ("tbytes", C.c_void_p),
                ("ncells", C.c_int64), ("nbases", C.c_int64), ("busy_waves", C.c_double),
                ("ctg_waves", C.c_void_p), ("nctg_waves", C.c_int)]


class Traces(C.Structure):
    _fields_ = [("naln", C.c_int64), ("ntrace", C.c_int64), ("npanels", C.c_int64), ("toff", C.c_void_p),
                ("tlen", C.c_void_p), ("diffs", C.c_void_p), ("trace", C.c_void_p), ("resume", C.c_void_p)]


class RunParams(C.Structure):
    _fields_ = [("device", C.c_int), ("freq", C.c_int), ("soft_mask", C.c_int), ("symmetric", C.c_int),
                ("chain_break", C.c_int), ("chain_min", C.c_int), ("align_min", C.c_int),
                ("align_rate", C.c_double), ("nthreads", C.c_int), ("out_path", C.c_char_p),
                ("command_line", C.c_char_p), ("paf_path", C.c_char_p), ("paf_flags", C.c_int),
                ("pass_seeds", C.c_int64), ("build_index", C.c_int),
                ("masks1", C.POINTER(C.c_char_p)), ("nmasks1", C.c_int), ("masks2", C.POINTER(C.c_char_p)), ("nmasks2", C.c_int),
                ("reference_threads", C.c_int)]


class RunStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("nseeds", "seed_len_sum", "nhits", "nunits", "nalns", "nlive", "cover",
                                         "ncalls", "nwaves")] + \
               [(n, C.c_double) for n in ("load_s", "upload_s", "merge_s", "sort_s", "download_s", "chain_s",
                                          "extend_s", "filter_s", "write_s", "phase23_s", "trace_s", "paf_s")] + \
               [(n, C.c_float) for n in ("merge_kernel_ms", "sort_kernel_ms", "extend_kernel_ms",
                                         "trace_kernel_ms")] + [("nparts", C.c_int), ("bases1", C.c_int64), ("bases2", C.c_int64), ("ext_cells", C.c_int64),
                                                     ("ext_bases", C.c_int64), ("ext_trace", C.c_int64),
                                                     ("ext_busy_waves", C.c_double), ("hbm_peak_bytes", C.c_int64),
                                                     ("sort_keys", C.c_int64), ("sort_passes", C.c_int), ("streamed_parts", C.c_int),
                                                     ("chain_kernel_ms", C.c_float)]


class SortParams(C.Structure):
    _fields_ = [("amxpos", C.c_int64), ("bmxpos", C.c_int64), ("nctg_a", C.c_int), ("nctg_b", C.c_int),
                ("anti_order_only", C.c_int)]


STAGE_MERGE_PARTITION, STAGE_MERGE, STAGE_SORT, STAGE_CHAIN, STAGE_EXTEND = 0, 1, 2, 3, 4
STAGE_GIX, STAGE_TRACE = 5, 6


def _declare(L):
    vp, i32, i64, cp = C.c_void_p, C.c_int, C.c_int64, C.c_char_p
    P = C.POINTER
    sig = {
        "fga_fasta_to_gdb": (i32, [cp, cp, i32]),
        "fga_gdb_open": (i32, [cp, P(vp)]),
        "fga_gdb_close": (None, [vp]),
        "fga_gdb_ncontig": (i32, [vp]),
        "fga_gdb_nscaff": (i32, [vp]),
        "fga_gdb_seqtot": (i64, [vp]),
        "fga_gdb_maxctg": (i64, [vp]),
        "fga_gdb_contig_len": (i64, [vp, i32]),
        "fga_gdb_freq": (None, [vp, P(C.c_float)]),
        "fga_gdb_get_contig": (vp, [vp, i32, vp]),
        "fga_gix_build": (i32, [vp, cp, i32]),
        "fga_gix_build_masked": (i32, [vp, cp, i32, i32]),
        "fga_gdb_nmask": (i64, [vp]),
        "fga_gix_open": (i32, [cp, P(vp)]),
        "fga_gix_close": (None, [vp]),
        "fga_gix_nents": (i64, [vp]),
        "fga_gix_ebytes": (i32, [vp]),
        "fga_gix_postbytes": (i32, [vp]),
        "fga_gix_contbytes": (i32, [vp]),
        "fga_gix_nctg": (i32, [vp]),
        "fga_gix_nparts": (i32, [vp]),
        "fga_gix_part_begin": (i64, [vp, i32]),
        "fga_gix_maxpre": (i64, [vp]),
        "fga_gix_perm": (P(C.c_int), [vp]),
        "fga_gix_legacy_cutoff": (i32, [vp]),
        "fga_gix_index": (P(C.c_int64), [vp]),
        "fga_gix_table": (P(C.c_uint8), [vp]),
        "fga_dev_open": (i32, [i32, P(vp)]),
        "fga_dev_close": (None, [vp]),
        "fga_dev_sync": (i32, [vp]),
        "fga_dev_stage_ms": (C.c_float, [vp, i32]),
        "fga_dev_malloc": (i32, [vp, C.c_size_t, P(vp)]),
        "fga_dev_free": (None, [vp, vp]),
        "fga_dev_download": (i32, [vp, vp, vp, C.c_size_t]),
        "fga_dev_peak_bytes": (i64, [vp]),
        "fga_dev_trim": (None, [vp]),
        "fga_dev_available": (C.c_size_t, [vp]),
        "fga_dev_set_host_threads": (None, [vp, i32]),
        "fga_dgix_build": (i32, [vp, vp, i32, i32, P(vp), P(vp)]),
        "fga_gix_write_files": (i32, [vp, cp]),
        "fga_dgix_upload": (i32, [vp, vp, P(vp)]),
        "fga_dgix_free": (None, [vp]),
        "fga_seed_merge": (i32, [vp, vp, vp, P(MergeParams), i64, P(vp)]),
        "fga_seeds_count": (i64, [vp]),
        "fga_seeds_plen_sum": (i64, [vp]),
        "fga_seeds_download": (i32, [vp, vp, i64]),
        "fga_seeds_free": (None, [vp]),
        "fga_seed_sort": (i32, [vp, vp, P(SortParams), P(vp)]),
        "fga_keys_count": (i64, [vp]),
        "fga_keys_layout": (None, [vp, P(i32), P(i32), P(i32), P(i32)]),
        "fga_keys_download": (i32, [vp, vp, i64]),
        "fga_keys_download_pinned": (vp, [vp]),
        "fga_keys_free": (None, [vp]),
        "fga_chain_scan": (i32, [vp, i64, i32, i32, i32, i32, P(ChainParams), i32, P(P(Hits))]),
        "fga_chain_scan_device": (i32, [vp, vp, P(ChainParams), P(P(Hits))]),
        "fga_hits_create": (i32, [vp, i64, vp, i64, P(P(Hits))]),
        "fga_hits_free": (None, [P(Hits)]),
        "fga_hits_count": (i64, [P(Hits)]),
        "fga_hits_nunits": (i64, [P(Hits)]),
        "fga_hits_array": (vp, [P(Hits)]),
        "fga_hits_units": (vp, [P(Hits)]),
        "fga_align_spec": (i32, [C.c_double, i32, P(C.c_float), P(i32), vp, vp]),
        "fga_dgenome_upload": (i32, [vp, vp, vp, i32, i32, P(vp)]),
        "fga_dgenome_free": (None, [vp]),
        "fga_extend": (i32, [vp, vp, vp, P(Hits), P(ExtendParams), P(P(Alns))]),
        "fga_alns_free": (None, [P(Alns)]),
        "fga_seed_merge_append": (i32, [vp, vp, vp, P(MergeParams), vp]),
        "fga_filter_alignments": (i32, [P(Alns), P(P(Alns))]),
        "fga_filter_alignments_mt": (i32, [P(Alns), i32, P(P(Alns))]),
        "fga_write_1aln": (i32, [cp, vp, vp, P(Alns), i32, cp, cp, cp]),
        "fga_write_1aln_binary": (i32, [cp, vp, vp, P(Alns), i32, cp, cp, cp]),
        "fga_aln_stream_open": (i32, [cp, vp, vp, i32, cp, cp, cp, P(vp)]),
        "fga_aln_stream_append": (i32, [vp, P(Alns)]),
        "fga_aln_stream_preformats": (i32, [vp]),
        "fga_aln_stream_format": (i32, [vp, P(Alns), P(vp)]),
        "fga_aln_stream_commit": (i32, [vp, vp]),
        "fga_aln_block_free": (None, [vp]),
        "fga_aln_stream_records": (i64, [vp]),
        "fga_aln_stream_close": (i32, [vp, i32]),
        "fga_trace_pts": (i32, [vp, vp, vp, P(Alns), i32, i32, P(P(Traces))]),
        "fga_trace_pts_regrouped": (i32, [vp, vp, vp, P(Alns), i32, i32, P(P(Traces))]),
        "fga_gap_core_check": (i32, [vp, vp, P(Alns), P(Traces), i32, C.c_int64]),
        "fga_dev_driver_seconds": (C.c_double, []),
        "fga_traces_free": (None, [P(Traces)]),
        "fga_write_paf": (i32, [cp, vp, vp, P(Alns), P(Traces), i32, i32]),
        "fga_write_psl": (i32, [cp, vp, vp, P(Alns), P(Traces), i32]),
        "fga_read_1aln": (i32, [cp, P(P(Alns)), P(i32), P(cp), P(cp)]),
        "fga_gap_improve": (i32, [vp, vp, P(Alns), P(Traces)]),
        "fga_run": (i32, [cp, cp, P(RunParams), P(RunStats)]),
        "fga_run_multi": (i32, [cp, cp, P(RunParams), i32, P(i32), P(RunStats)]),
        "fga_multi_open": (i32, [cp, cp, P(RunParams), i32, P(i32), P(vp)]),
        "fga_multi_run": (i32, [vp, P(RunParams), P(RunStats)]),
        "fga_multi_close": (None, [vp]),
        "fga_multi_ndev": (i32, [vp]),
        "fga_multi_info": (i32, [vp, P(i64), P(i32), P(i64), P(i64)]),
        "fga_multi_rank_stats": (i32, [vp, i32, P(C.c_double), P(C.c_double), P(i64)]),
        "fga_seeds_import_peer": (i32, [vp, P(vp), P(i32), P(i64), i32, P(vp)]),
        "fga_dev_enable_peer": (i32, [vp, i32]),
        "fga_dev_device_count": (i32, []),
        "fga_session_open": (i32, [cp, cp, i32, P(vp)]),
        "fga_session_open_threads": (i32, [cp, cp, i32, i32, P(vp)]),
        "fga_session_open_flags": (i32, [cp, cp, i32, i32, i32, P(vp)]),
        "fga_gdb_apply_masks": (i32, [vp, P(cp), i32]),
        "fga_session_run": (i32, [vp, P(RunParams), P(RunStats)]),
        "fga_session_close": (None, [vp]),
        "fga_session_device": (vp, [vp]),
        "fga_session_table_bytes": (i64, [vp]),
        "fga_session_seed_bytes": (i32, [vp]),
        "fga_session_bases": (i64, [vp, i32]),
        "fga_session_nctg": (i32, [vp]),
        "fga_session_contig_perm": (P(i32), [vp]),
        "fga_session_merge": (i32, [vp, P(RunParams), i64, i64, P(vp), P(RunStats)]),
        "fga_session_align": (i32, [vp, P(RunParams), vp, P(P(Alns)), P(RunStats)]),
        "fga_session_finish": (i32, [vp, P(RunParams), P(P(Alns)), i32, P(RunStats)]),
        "fga_seeds_contig_histogram": (i32, [vp, vp, i32, P(i64)]),
        "fga_partition_contigs": (i32, [P(i64), i32, i32, P(i32)]),
        "fga_partition_contigs_in_order": (i32, [P(i64), P(i32), i32, i32, P(i32)]),
        "fga_seeds_split_to": (i32, [vp, vp, P(i32), i32, i32, vp, P(i64)]),
        "fga_seeds_import": (i32, [vp, P(vp), P(i64), i32, P(vp)]),
        "fga_seeds_view": (i32, [vp, vp, i64, P(vp)]),
        "fga_seeds_device_ptr": (vp, [vp]),
        "fga_merge_prefix_cuts": (i32, [vp, vp, vp, i32, P(i64)]),
        "fga_session_prefix_cuts": (i32, [vp, i32, P(i64)]),
        "fga_session_strand_counts": (i32, [vp, P(i64)]),
        "fga_session_set_strand_counts": (i32, [vp, P(i64)]),
        "fga_session_clear_strand_counts": (None, [vp]),
        "fga_seeds_strand_histogram": (i32, [vp, vp, i32, P(i64)]),
        "fga_rmsd_ranges": (i32, [P(i64), i32, i64, i32, P(i32), P(i32), P(i64)]),
        "fga_reference_slots": (i32, [P(i64), P(i64), i32, i32, i32, P(i32)]),
        "fga_alns_concat": (i32, [P(P(Alns)), i32, P(P(Alns))]),
        "fga_dev_upload": (i32, [vp, vp, vp, C.c_size_t]),
        "fga_session_open_sliced": (i32, [cp, cp, i32, i32, i32, i32, P(vp)]),
        "fga_dgix_build_range": (i32, [vp, vp, i32, i32, i64, i64, P(vp), P(vp)]),
        "fga_dgix_upload_range": (i32, [vp, vp, i64, i64, P(vp)]),
        "fga_dgix_prefix_counts": (i32, [vp, vp, i32, vp]),
        "fga_alns_merge_filtered": (i32, [P(P(Alns)), i32, P(P(Alns))]),
        "fga_alns_merge_filtered_mt": (i32, [P(P(Alns)), i32, i32, P(P(Alns))]),
        "fga_session_finish_filtered": (i32, [vp, P(RunParams), P(P(Alns)), i32, P(RunStats)]),
        "fga_session_stream_open": (i32, [vp, P(RunParams), P(vp)]),
        "fga_session_reference_order": (i32, [vp, P(RunParams), P(Alns)]),
        "fga_shim_New_Work_Data": (vp, []),
        "fga_shim_Free_Work_Data": (None, [vp]),
        "fga_shim_New_Align_Spec": (vp, [C.c_double, i32, P(C.c_float), i32]),
        "fga_shim_Free_Align_Spec": (None, [vp]),
        "fga_shim_Local_Alignment": (i32, [vp, vp, vp, i32, i32, i32, i32, i32]),
        "fga_shim_rmsd_sort": (i32, [vp, i64, i32, i32, i32, P(i64), i32, vp]),
        "fga_shim_Compute_Trace_PTS": (i32, [vp, vp, i32, i32, i32, i32]),
        "fga_shim_Gap_Improver": (i32, [vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._declared = sorted(sig)
