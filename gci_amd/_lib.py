"""ctypes binding of libgci_hip.so (include/gci_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at load, and
creating a context without an MI355X raises GciError.  (The CPU oracle lives under oracle/
and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCI_LIB_PATH") or os.path.join(_HERE, "csrc", "libgci_hip.so")   # GCI_LIB_PATH: an experimental build

GCI_OK, GCI_E_INVALID, GCI_E_HIP, GCI_E_NO_NM, GCI_E_ZERO_DIV = 0, -1, -2, -3, -4
GCI_E_BAD_NM_TYPE, GCI_E_NO_END, GCI_E_MALFORMED, GCI_E_CAPACITY, GCI_E_NOMEM, GCI_E_NO_LAYOUT = -5, -6, -7, -8, -9, -10
GCI_TILE = 4096
GCI_MAX_JOIN_FILES = 16
PAGE_MAX_REC, PAGE_MAX_BYTES, PAGE_BYTES_DEFAULT = 1024, 32768, 24576
REC_PASS, REC_HQ = 1, 2
PROF_COUNT = 19
PROF_DEPTH_SCAN = 5          # k_tile_build: the pass that writes the depth track (+ text)
PROF_TILE_PASS1 = 13
PROF_TILE_DENSE = 14         # k_tile_dense<1>, <2>: the tiles the event-list kernels left over


class GciError(RuntimeError):
    def __init__(self, status: int, msg: str, rec: int = -1):
        super().__init__(msg)
        self.status, self.rec = status, rec


class JoinFile(ctypes.Structure):
    _fields_ = [("d_recs", c_void_p), ("n_recs", c_uint32), ("name_delta", c_uint32),
                ("d_name_base", c_void_p), ("d_name_off", c_void_p)]


class Window(ctypes.Structure):
    _fields_ = [("begin", c_int64), ("end", c_int64)]


class BuildOpts(ctypes.Structure):
    _fields_ = [("flank", c_int), ("want_text", c_int), ("d_contig_text_off", c_void_p), ("d_sums", c_void_p),
                ("d_n_keys", c_void_p), ("d_keys", c_void_p), ("key_cap", c_uint32), ("issue_flank", c_int),
                ("lo", c_double), ("hi", c_double), ("counted", c_int), ("want_runs", c_int)]


# every symbol include/gci_hip.h declares: (name, restype, argtypes)
EXPORTS = [
    ("gci_abi_version", c_int, []),
    ("gci_ctx_create", c_int, [c_int, c_void_p, c_int, POINTER(c_void_p)]),
    ("gci_ctx_destroy", c_int, [c_void_p]),
    ("gci_sync", c_int, [c_void_p]),
    ("gci_strerror", c_char_p, [c_int]),
    ("gci_last_error", c_char_p, [c_void_p]),
    ("gci_malloc", c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    ("gci_free", c_int, [c_void_p, c_void_p]),
    ("gci_memcpy_h2d", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("gci_memcpy_d2h", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("gci_memset", c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    ("gci_dev_last_error", c_char_p, []),
    ("gci_dev_count", c_int, [POINTER(c_int)]),
    ("gci_dev_malloc", c_int, [c_int, c_size_t, POINTER(c_void_p)]),
    ("gci_dev_free", c_int, [c_int, c_void_p]),
    ("gci_dev_reserve", c_int, [c_int, c_uint64, POINTER(c_uint64)]),
    ("gci_dev_arena_info", c_int, [c_int, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint32)]),
    ("gci_dev_mem_info", c_int, [c_int, POINTER(c_uint64), POINTER(c_uint64)]),
    ("gci_dev_sync", c_int, [c_int]),
    ("gci_dev_host_alloc", c_int, [c_int, c_size_t, POINTER(c_void_p)]),
    ("gci_dev_host_free", c_int, [c_int, c_void_p]),
    ("gci_dev_stream_create", c_int, [c_int, POINTER(c_void_p)]),
    ("gci_dev_stream_destroy", c_int, [c_int, c_void_p]),
    ("gci_dev_stream_sync", c_int, [c_int, c_void_p]),
    ("gci_dev_event_create", c_int, [c_int, c_int, POINTER(c_void_p)]),
    ("gci_dev_event_destroy", c_int, [c_int, c_void_p]),
    ("gci_dev_event_record", c_int, [c_int, c_void_p, c_void_p]),
    ("gci_dev_event_sync", c_int, [c_int, c_void_p]),
    ("gci_dev_event_elapsed_ms", c_int, [c_int, c_void_p, c_void_p, POINTER(c_double)]),
    ("gci_dev_stream_wait_event", c_int, [c_int, c_void_p, c_void_p]),
    ("gci_dev_memcpy_async", c_int, [c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    ("gci_dev_memset_async", c_int, [c_int, c_void_p, c_int, c_size_t, c_void_p]),
    ("gci_dev_i64_add", c_int, [c_int, c_void_p, c_uint64, c_int64, c_void_p, c_void_p]),
    ("gci_dev_rec_flags_and", c_int, [c_int, c_void_p, c_uint64, c_uint32, c_void_p]),
    ("gci_dev_u32_scan_u64", c_int, [c_int, c_void_p, c_uint32, c_void_p, c_void_p]),
    ("gci_profile_enable", c_int, [c_void_p, c_int]),
    ("gci_profile_read", c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_uint64), c_int]),
    ("gci_profile_name", c_char_p, [c_int]),
    ("gci_layout_set", c_int, [c_void_p, c_int32, c_void_p]),
    ("gci_layout_total", c_int64, [c_void_p]),
    ("gci_layout_offsets", c_int, [c_void_p, c_void_p]),
    ("gci_bam_filter", c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_int32, c_int, c_int,
                               c_double, c_double, c_uint32, c_void_p, c_void_p]),
    ("gci_bam_pages_size", c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_int, c_uint32, c_void_p]),
    ("gci_bam_pages_write", c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_int, c_void_p, c_uint64]),
    ("gci_bam_filter_pages", c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_uint32, c_void_p, c_int32, c_int, c_int,
                                     c_double, c_double, c_uint32, c_void_p, c_void_p, c_void_p]),
    ("gci_decode_status", c_int, [c_uint64, POINTER(c_uint32)]),
    ("gci_name_hash", c_uint64, [c_void_p, c_uint32]),
    ("gci_pack_names", c_int, [c_void_p, POINTER(JoinFile), c_void_p, c_uint64, c_void_p]),
    ("gci_name_join", c_int, [c_void_p, POINTER(JoinFile), c_int, c_double, c_void_p, c_void_p, c_uint32, c_void_p,
                              c_void_p]),
    ("gci_join_mode", c_int, [c_void_p, c_int]),
    ("gci_name_join_count", c_int, [c_void_p, POINTER(JoinFile), c_int, c_double, c_void_p, c_void_p, c_uint32, c_void_p,
                              c_void_p, c_int]),
    ("gci_depth_deflate_from_build", c_int, [c_void_p, c_void_p]),
    ("gci_depth_deflate_size", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("gci_depth_deflate_write", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_uint64]),
    ("gci_hash_bucket", c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_void_p, c_void_p]),
    ("gci_hash_conflicts", c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_void_p]),
    ("gci_route_records", c_int, [c_void_p, POINTER(JoinFile), c_uint32, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p]),
    ("gci_route_seal_records", c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_void_p]),
    ("gci_route_intervals", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_int32, c_uint32, c_uint32, c_void_p, c_void_p]),
    ("gci_route_seal_intervals", c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_void_p, c_int32, c_void_p]),
    ("gci_depth_build", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_int, c_void_p]),
    ("gci_depth_build_begin", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, POINTER(BuildOpts)]),
    ("gci_depth_build_finish", c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    ("gci_gap_mask", c_int, [c_void_p, c_void_p, c_void_p, c_uint32]),
    ("gci_max2", c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("gci_two_type_tail", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_double, c_double, c_int, c_void_p,
                                  c_uint32, c_void_p, c_void_p]),
    ("gci_issue_scan", c_int, [c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_uint32, c_void_p]),
    ("gci_issue_scan_windows", c_int, [c_void_p, c_void_p, POINTER(Window), c_uint32, c_double, c_double, c_void_p,
                                       c_uint32, c_void_p]),
    ("gci_depth_text_size", c_int, [c_void_p, c_void_p, c_void_p]),
    ("gci_depth_text_write", c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    ("gci_depth_sum", c_int, [c_void_p, c_void_p, c_void_p]),
    ("gci_range_sums", c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    ("gci_fasta_n_scan", c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p]),
    ("gci_paf_filter", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, ctypes.c_double, c_int, c_void_p, c_void_p]),
    ("gci_paf_count", c_uint64, [c_void_p, c_int]),
    ("gci_paf_name_bytes", c_uint64, [c_void_p, c_int]),
    ("gci_paf_export", c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    ("gci_paf_free", c_int, [c_void_p]),
    ("gci_paf_filter_device", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, ctypes.c_double, c_void_p,
                                      c_void_p]),
    ("gci_paf_dev_count", c_uint64, [c_void_p, c_int]),
    ("gci_paf_dev_export", c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    ("gci_paf_dev_free", c_int, [c_void_p]),
    ("gci_paf_hits_device", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, ctypes.c_double, c_void_p,
                                    c_void_p]),
    ("gci_paf_hits_count", c_uint64, [c_void_p, c_int]),
    ("gci_paf_hits_export", c_int, [c_void_p, c_int, c_void_p]),
    ("gci_paf_hits_free", c_int, [c_void_p]),
    ("gci_route_hits", c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p]),
    ("gci_paf_pool_release", c_int, [c_void_p]),
    ("gci_paf_score_device", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    ("gci_stage_create", c_int, [c_void_p, c_uint64, c_int, c_int, POINTER(c_void_p)]),
    ("gci_stage_send", c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_int]),
    ("gci_stage_send_fd", c_int, [c_void_p, c_void_p, c_int, c_uint64, c_uint64, c_void_p, c_void_p, c_int]),
    ("gci_stage_free", c_int, [c_void_p]),
    ("gci_bgzf_inflate_device", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_uint64, c_int, c_void_p]),
    ("gci_bgzf_inflate_streams", c_int, [c_void_p, c_int]),
    ("gci_bgzf_inflate_last_stats", c_int, [c_void_p, c_void_p]),
    ("gci_bgzf_inflate_round", c_uint32, [c_void_p]),
    ("gci_bam_record_offsets_device", c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_int32, c_void_p, c_uint64, c_void_p]),
    ("gci_bgzf_scan", c_int, [c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    ("gci_bgzf_blocks", c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    ("gci_bgzf_table_build", c_int, [c_void_p, c_uint64, c_int, POINTER(c_void_p)]),
    ("gci_bgzf_table_build_prefix", c_int, [c_void_p, c_uint64, c_uint64, c_int, POINTER(c_void_p)]),
    ("gci_bgzf_table_build_fd", c_int, [c_int, c_uint64, c_int, POINTER(c_void_p)]),
    ("gci_bgzf_table_count", c_uint64, [c_void_p]),
    ("gci_bgzf_table_export", c_int, [c_void_p, c_void_p, c_void_p]),
    ("gci_bgzf_table_free", c_int, [c_void_p]),
    ("gci_bam_chunk_offsets", c_int, [c_void_p, c_uint64, c_uint64, c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    ("gci_bgzf_inflate", c_int, [c_void_p, c_uint64, c_void_p, c_uint64, c_int, c_int]),
    ("gci_bam_record_offsets", c_int, [c_void_p, c_uint64, c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    ("gci_bam_heads", c_int, [c_void_p, c_uint64, c_int, c_uint64, c_int, POINTER(c_void_p)]),
    ("gci_bam_heads_bytes", c_uint64, [c_void_p]),
    ("gci_bam_heads_count", c_uint64, [c_void_p]),
    ("gci_bam_heads_first", c_uint64, [c_void_p]),
    ("gci_bam_heads_stream", c_void_p, [c_void_p]),
    ("gci_bam_heads_offsets", c_void_p, [c_void_p]),
    ("gci_bam_heads_free", c_int, [c_void_p]),
    ("gci_fasta_titles", c_int, [c_void_p, c_uint64, c_int, c_void_p, c_uint64, POINTER(c_uint64)]),
    ("gci_gzip_bound", c_uint64, [c_uint64, c_uint64]),
    ("gci_gzip_members", c_int, [c_void_p, c_uint64, c_uint64, c_int, c_int, c_void_p, c_uint64, POINTER(c_uint64)]),
]

_lib = None


def _one_hip_runtime() -> None:
    """libgci_hip.so is linked against the system's libamdhip64; a torch wheel brings a copy of its own and loads THAT one by path.
    A process that ends up with both (this library first, `import torch` later: a test session, GCI_HBM=torch, an embedding
    application) has two runtimes, and the second finds no device.  So when a torch installation is present, its copy of the
    runtime is mapped first -- found through the module spec, torch itself is NOT imported (a few ms) -- and this library's
    DT_NEEDED entry resolves to it by soname: one runtime whoever comes first.  GCI_HIP_RUNTIME=system skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("GCI_HIP_RUNTIME") == "system" or os.environ.get("GCI_HOST_ONLY") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        path = spec and spec.origin and os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if path and os.path.isfile(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:                                      # noqa: BLE001  (no torch, an unusual layout: the system's runtime)
        pass


_load_lock = __import__("threading").Lock()


def load() -> ctypes.CDLL:
    """Load the library and bind every export; raises if the .so or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        return _load_locked()


def _load_locked() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GciError(GCI_E_INVALID, "libgci_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` -- there is no CPU fallback" % LIB_PATH)
        _one_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        # GCI_HOST_ONLY=1: a build of the host-side entry points alone (host_io.cpp through g++ with the sanitizers,
        # tools/asan_host.sh): the device exports are absent, nothing that needs them can run
        host_only = os.environ.get("GCI_HOST_ONLY") == "1"
        for name, res, args in EXPORTS:
            if host_only and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)           # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if not host_only and lib.gci_abi_version() != 1:
            raise GciError(GCI_E_INVALID, "libgci_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def name_hash(name: bytes) -> int:
    return int(load().gci_name_hash(name, len(name)))
