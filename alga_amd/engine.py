"""ctypes binding of libalga_amd.so (include/alga_amd.h) + the host-side mirror of the reference's
GraphCreator interface for this path (include/GraphCreators/GraphCreator.h:12-62 of the reference).

Nothing here computes an overlap: every build call goes through the C ABI into the HIP kernels and
raises AlgaError when the library or the device is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EDGE_DTYPE = np.dtype([("src", np.int32), ("dst", np.int32), ("offset", np.int32)])


class AlgaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("alga_amd error %d: %s" % (code, msg))
        self.code = code


class _Nodes(C.Structure):
    _fields_ = [("words", C.c_void_p), ("stride_words", C.c_int32), ("len", C.c_void_p), ("n", C.c_int32),
                ("align_from", C.c_void_p), ("align_to", C.c_void_p)]


class PrefSufParams(C.Structure):
    """alga_prefsuf_params"""
    _fields_ = [("min_overlap", C.c_int32), ("rsoe_min_overlap", C.c_int32), ("soes", C.c_int32),
                ("max_len_cap", C.c_int32), ("collect_stats", C.c_int32), ("reduction", C.c_int32),
                ("keys_shared", C.c_int32), ("twin_rows", C.c_int32)]


class CompactEdges(C.Structure):
    """alga_compact_edges"""
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_uint64), ("degree", C.c_void_p), ("dst", C.c_void_p), ("offset", C.c_void_p)]


class PrefSufStats(C.Structure):
    """alga_prefsuf_stats"""
    _fields_ = [("raw_overlaps", C.c_uint64), ("transitive_listed", C.c_uint64), ("transitive_compares", C.c_uint64),
                ("transitive_removed", C.c_uint64), ("windows_probed", C.c_uint64), ("slots_scanned", C.c_uint64),
                ("records", C.c_uint64), ("edges", C.c_uint64), ("table_slots", C.c_uint64),
                ("max_in_records", C.c_uint64), ("ms_total", C.c_double), ("ms_seed", C.c_double),
                ("ms_probe", C.c_double), ("ms_group", C.c_double), ("ms_reduce", C.c_double), ("ms_emit", C.c_double),
                ("nodes_live", C.c_uint64), ("reduction_used", C.c_uint64), ("generic_sources", C.c_uint64),
                ("big_sources", C.c_uint64), ("probe_used", C.c_uint64), ("deferred_sources", C.c_uint64), ("ms_probe_pairs", C.c_double),
                ("ms_keys", C.c_double), ("ms_sort", C.c_double), ("ms_gather", C.c_double), ("ms_dir", C.c_double), ("probe_rounds", C.c_uint64), ("ms_pile", C.c_double),
                ("pile_buckets", C.c_uint64), ("pile_irregular", C.c_uint64), ("pile_list_checked", C.c_uint64), ("pile_list_mismatch", C.c_uint64),
                ("pile_own_lists", C.c_uint64), ("host_ms_check", C.c_double), ("host_ms_upload", C.c_double), ("host_ms_build", C.c_double),
                ("host_ms_download", C.c_double), ("pile_mixed", C.c_uint64), ("pile_deferred", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class IngestParams(C.Structure):
    """alga_ingest_params"""
    _fields_ = [("trim_left", C.c_int32), ("trim_right", C.c_int32), ("remove_reads_with_n", C.c_int32), ("rna", C.c_int32),
                ("scale", C.c_float), ("min_overlap", C.c_int32), ("rsoemo", C.c_int32), ("remove_pref_reads", C.c_int32),
                ("threads", C.c_int32)]


class NodeSet(C.Structure):
    """alga_node_set"""
    _fields_ = [("n", C.c_int32), ("stride_words", C.c_int32), ("words", C.POINTER(C.c_uint32)), ("len", C.POINTER(C.c_int32)),
                ("pair_off", C.POINTER(C.c_uint8)), ("LEN", C.c_int32), ("min_overlap", C.c_int32), ("rsoemo", C.c_int32),
                ("li_kmer_length", C.c_int32), ("records", C.c_int64), ("removed_n", C.c_int32), ("removed_str", C.c_int32),
                ("removed_prefix", C.c_int32), ("removed_short", C.c_int32), ("avg_len", C.c_double)]


class ParsedReads(C.Structure):
    """alga_parsed_reads"""
    _fields_ = [("n_nodes", C.c_int64), ("stride_words", C.c_int32), ("rows", C.POINTER(C.c_uint32)), ("len", C.POINTER(C.c_int32)),
                ("paired", C.c_int32), ("records", C.c_int64), ("removed_n", C.c_int32), ("removed_str", C.c_int32), ("LEN", C.c_int32),
                ("min_overlap", C.c_int32), ("rsoemo", C.c_int32), ("li_kmer_length", C.c_int32), ("avg_len", C.c_double),
                ("owner", C.c_void_p)]


class PreprocessInput(C.Structure):
    """alga_preprocess_input"""
    _fields_ = [("rows", C.c_void_p), ("stride_words", C.c_int32), ("len", C.c_void_p), ("n_nodes", C.c_int64),
                ("remove_pref_reads", C.c_int32), ("min_keep_len", C.c_int32)]


class DeviceNodeSet(C.Structure):
    """alga_device_node_set"""
    _fields_ = [("d_words", C.c_void_p), ("d_len", C.c_void_p), ("d_pair_off", C.c_void_p), ("n", C.c_int32), ("stride_words", C.c_int32),
                ("removed_prefix", C.c_int32), ("removed_short", C.c_int32), ("max_len", C.c_int32), ("ms_device", C.c_double)]


class IngestInfo(C.Structure):
    """alga_ingest_info"""
    _fields_ = [("records", C.c_int64), ("removed_n", C.c_int32), ("removed_str", C.c_int32), ("LEN", C.c_int32), ("min_overlap", C.c_int32),
                ("rsoemo", C.c_int32), ("li_kmer_length", C.c_int32), ("paired", C.c_int32), ("avg_len", C.c_double), ("ms_parse", C.c_double),
                ("ms_preprocess", C.c_double), ("ms_upload", C.c_double)]


class PkbParams(C.Structure):
    """alga_pkb_params"""
    _fields_ = [("min_overlap_area", C.c_int32), ("max_offset_pct", C.c_int32), ("min_identity_pct", C.c_int32),
                ("same_ends", C.c_int32), ("li_k", C.c_int32), ("li_intervals", C.c_int32), ("rounds", C.c_int32),
                ("kmer_length_bucket", C.c_int32)]


class PkbStats(C.Structure):
    """alga_pkb_stats"""
    _fields_ = [("kmers", C.c_uint64 * 4), ("groups", C.c_uint64 * 4), ("can_align_calls", C.c_uint64 * 4),
                ("edges_after", C.c_uint64 * 4), ("max_group", C.c_uint64), ("ms_total", C.c_double), ("group_hist", (C.c_uint64 * 8) * 4)]


PROBE = {"auto": 0, "table": 1, "cluster": 2}                  # alga_probe
PILE_DECLINE_ONE_IN = 20                                       # ALGA_PILE_DECLINE_ONE_IN: the pile path keeps a build iff pile_irregular * this <= pile_buckets (round 5; between the two: the mixed form)
PILE_IRREGULAR_ONE_IN = 250                                    # ALGA_PILE_IRREGULAR_ONE_IN (include/alga_amd.h): the pile path keeps a build iff pile_irregular * this <= pile_buckets

class MultiStats(C.Structure):
    """alga_multi_stats"""
    _fields_ = [("n_ranks", C.c_int32), ("transport", C.c_int32), ("fell_back_to_one_gpu", C.c_int32), ("form", C.c_int32), ("edges", C.c_uint64),
                ("ms_upload", C.c_double), ("ms_download", C.c_double), ("ms_keys", C.c_double), ("ms_share", C.c_double), ("ms_build", C.c_double),
                ("ms_gather", C.c_double), ("ms_total", C.c_double),
                ("xbytes_keys", C.c_uint64), ("xbytes_descriptors", C.c_uint64), ("xbytes_pending", C.c_uint64), ("xbytes_small_keys", C.c_uint64),
                ("xbytes_edges", C.c_uint64), ("xbytes_gather", C.c_uint64),
                ("ms_shard_index", C.c_double), ("ms_shard_exchange", C.c_double), ("ms_shard_join", C.c_double), ("ms_shard_cap", C.c_double),
                ("ms_shard_place", C.c_double)]


class ShardStats(C.Structure):
    """alga_shard_stats"""
    _fields_ = [(k, C.c_uint64) for k in ("targets_owned", "descriptors_out", "descriptors_in", "flagged_sources", "records", "pending", "pending_sources",
                                          "small_keys_out", "small_keys_in", "dropped", "edges_out", "edges_in", "edges", "join_passes", "join_passes_serial")] + \
               [(k, C.c_double) for k in ("ms_index", "ms_export", "ms_sort", "ms_join", "ms_cap", "ms_edges_out", "ms_place")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


MULTI_FORM = {"auto": 0, "replicated": 1, "bucket_sharded": 2}   # alga_multi_form


TRANSPORT = {"auto": 0, "rccl": 1, "copy": 2}                  # alga_transport


class NodeKeys(C.Structure):
    _fields_ = [("d_keys", C.c_void_p), ("d_meta", C.c_void_p), ("n", C.c_int32), ("eligible", C.c_int32), ("meta_needed", C.c_int32), ("reserved", C.c_int32)]


EXPORTS = ["alga_abi_version", "alga_engine_set_option", "alga_engine_create", "alga_engine_destroy", "alga_last_error",
           "alga_engine_device_name", "alga_prefsuf_default_params", "alga_prefsuf_build_host", "alga_free_edges",
           "alga_prefsuf_build_device", "alga_prefsuf_last_stats", "alga_prefsuf_discover_device",
           "alga_prefsuf_reduce_device", "alga_prefsuf_build_range_device", "alga_prefsuf_keys_device", "alga_write_graph", "alga_ingest_default_params", "alga_ingest_files",
           "alga_free_node_set", "alga_sort_records_device", "alga_sort_edges_device", "alga_pkb_derive_params",
           "alga_can_align_batch_host", "alga_li_kmers_host", "alga_pkb_supplement_host", "alga_pkb_supplement_device",
           "alga_pkb_last_stats", "alga_parse_files", "alga_free_parsed_reads", "alga_preprocess_nodes", "alga_copy_to_host", "alga_device_alloc", "alga_device_free", "alga_copy_to_device", "alga_cut_triangles_device", "alga_cut_triangles_host", "alga_ingest_device", "alga_contig_trim_host", "alga_engine_reserve", "alga_upload_nodes", "alga_download_edges",
           "alga_multi_create", "alga_multi_destroy", "alga_multi_last_error", "alga_multi_engine", "alga_multi_prefsuf_build_host", "alga_multi_prefsuf_build_device",
           "alga_multi_free_edges", "alga_multi_last_stats", "alga_multi_set_option", "alga_upload_twin_nodes",
           "alga_shard_index_device", "alga_shard_join_device", "alga_shard_small_keys_device", "alga_shard_resolve_device", "alga_shard_place_device",
           "alga_shard_last_stats", "alga_sort_u32_pairs_device", "alga_sort_u64_pairs_device", "alga_multi_pkb_supplement_device", "alga_pkb_shard_begin", "alga_pkb_shard_round", "alga_pkb_shard_merge", "alga_pkb_shard_end",
           "alga_prefsuf_build_host_compact", "alga_download_edges_compact", "alga_free_compact_edges", "alga_host_alloc", "alga_host_free"]


def library_path():
    return os.path.join(_HERE, "lib", "libalga_amd.so")


def source_fingerprint():
    """sha256 (first 16 hex digits) over the kernel / engine sources the library is built from (alga_amd/csrc/*, include/alga_amd.h):
    what profiles/ records next to a counter measurement, so that a number is only ever quoted for the code it was measured on.
    (The hash of the binary itself changes with the build directory.)"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".hpp", ".cpp")) or f == "Makefile")
    files.append(os.path.join(os.path.dirname(_HERE), "include", "alga_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_library():
    """Load libalga_amd.so; raises AlgaError (never falls back to anything else) if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64.  If the
    # system copy under /opt/rocm is initialised first, torch's copy later finds "No HIP GPUs".  Importing
    # torch first makes the loader resolve this library's NEEDED libamdhip64.so.7 to the copy already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise AlgaError(-2, "HIP extension %s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    lib = C.CDLL(path)
    lib.alga_abi_version.restype = C.c_int
    lib.alga_engine_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.alga_engine_destroy.argtypes = [C.c_void_p]
    lib.alga_engine_destroy.restype = None
    lib.alga_last_error.argtypes = [C.c_void_p]
    lib.alga_last_error.restype = C.c_char_p
    lib.alga_engine_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.alga_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.alga_engine_reserve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]
    lib.alga_prefsuf_default_params.argtypes = [C.POINTER(PrefSufParams)]
    lib.alga_prefsuf_default_params.restype = None
    lib.alga_prefsuf_build_host.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_free_edges.argtypes = [C.c_void_p, C.c_void_p]
    lib.alga_free_edges.restype = None
    lib.alga_prefsuf_build_host_compact.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.POINTER(CompactEdges)]
    lib.alga_free_compact_edges.argtypes = [C.c_void_p, C.POINTER(CompactEdges)]
    lib.alga_free_compact_edges.restype = None
    lib.alga_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    lib.alga_host_alloc.restype = C.c_void_p
    lib.alga_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.alga_host_free.restype = None
    lib.alga_prefsuf_build_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.c_void_p,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_prefsuf_last_stats.argtypes = [C.c_void_p, C.POINTER(PrefSufStats)]
    lib.alga_prefsuf_discover_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.c_int32,
                                                 C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_uint64)]
    lib.alga_prefsuf_reduce_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.c_void_p,
                                               C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_prefsuf_build_range_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.c_int32, C.c_int32,
                                                    C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_shard_index_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.alga_shard_join_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_shard_small_keys_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_shard_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.alga_shard_place_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_write_graph.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_uint64]
    lib.alga_sort_records_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_sort_edges_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.alga_sort_u64_pairs_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
    lib.alga_sort_u32_pairs_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_double)]
    lib.alga_pkb_shard_begin.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PkbParams), C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]
    lib.alga_pkb_shard_round.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_pkb_shard_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.alga_pkb_shard_end.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_pkb_derive_params.argtypes = [C.c_double, C.c_float, C.c_double, C.c_int32, C.POINTER(PkbParams)]
    lib.alga_pkb_derive_params.restype = None
    lib.alga_can_align_batch_host.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PkbParams), C.c_void_p, C.c_uint64, C.c_void_p]
    lib.alga_li_kmers_host.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PkbParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.alga_pkb_supplement_host.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PkbParams), C.c_void_p, C.c_uint64,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_pkb_supplement_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PkbParams), C.c_void_p, C.c_uint64, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_pkb_last_stats.argtypes = [C.c_void_p, C.POINTER(PkbStats)]
    lib.alga_ingest_default_params.argtypes = [C.POINTER(IngestParams)]
    lib.alga_ingest_default_params.restype = None
    lib.alga_ingest_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(IngestParams), C.POINTER(NodeSet), C.c_char_p, C.c_size_t]
    lib.alga_free_node_set.argtypes = [C.POINTER(NodeSet)]
    lib.alga_free_node_set.restype = None
    lib.alga_parse_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(IngestParams), C.POINTER(ParsedReads), C.c_char_p, C.c_size_t]
    lib.alga_free_parsed_reads.argtypes = [C.POINTER(ParsedReads)]
    lib.alga_free_parsed_reads.restype = None
    lib.alga_preprocess_nodes.argtypes = [C.c_void_p, C.POINTER(PreprocessInput), C.POINTER(DeviceNodeSet)]
    lib.alga_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.alga_contig_trim_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.alga_ingest_device.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(IngestParams), C.POINTER(DeviceNodeSet), C.POINTER(IngestInfo)]
    lib.alga_cut_triangles_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.alga_cut_triangles_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _LIB = lib
    return lib


def ingest_files(file1, file2=None, threads=1, **kw):
    """Host input stages (C++, alga_amd/host/ingest.cpp) through the C ABI -> dict with numpy copies."""
    lib = load_library()
    p = IngestParams()
    lib.alga_ingest_default_params(C.byref(p))
    p.threads = int(threads)
    for k, v in kw.items():
        setattr(p, k, v)
    ns = NodeSet()
    err = C.create_string_buffer(512)
    rc = lib.alga_ingest_files(file1.encode(), (file2 or "").encode() or None, C.byref(p), C.byref(ns), err, 512)
    if rc:
        raise AlgaError(rc, err.value.decode())
    n, st = ns.n, ns.stride_words
    out = dict(n=n, stride=st,
               words=np.ctypeslib.as_array(ns.words, shape=(max(n, 1) * st,))[: n * st].reshape(n, st).copy(),
               len=np.ctypeslib.as_array(ns.len, shape=(max(n, 1),))[:n].copy(),
               pair_off=np.ctypeslib.as_array(ns.pair_off, shape=(max(n, 1),))[:n].copy(),
               LEN=ns.LEN, min_overlap=ns.min_overlap, rsoemo=ns.rsoemo, li_kmer_length=ns.li_kmer_length, records=ns.records,
               removed_n=ns.removed_n, removed_str=ns.removed_str, removed_prefix=ns.removed_prefix, removed_short=ns.removed_short)
    lib.alga_free_node_set(C.byref(ns))
    return out


def parse_files(file1, file2=None, threads=1, **kw):
    """Stage 1 of the input (C++ host): every record's two nodes in node order, before duplicate / prefix removal.
    -> dict(rows[2R, stride] u32, len[2R] i32 (-1 removed), params); numpy copies."""
    lib = load_library()
    p = IngestParams()
    lib.alga_ingest_default_params(C.byref(p))
    p.threads = int(threads)
    for k, v in kw.items():
        setattr(p, k, v)
    pr = ParsedReads()
    err = C.create_string_buffer(512)
    rc = lib.alga_parse_files(file1.encode(), (file2 or "").encode() or None, C.byref(p), C.byref(pr), err, 512)
    if rc:
        raise AlgaError(rc, err.value.decode())
    n, st = int(pr.n_nodes), int(pr.stride_words)
    out = dict(n_nodes=n, stride=st, rows=np.ctypeslib.as_array(pr.rows, shape=(max(n, 1) * st,))[: n * st].reshape(n, st).copy(),
               len=np.ctypeslib.as_array(pr.len, shape=(max(n, 1),))[:n].copy(), paired=bool(pr.paired), records=pr.records,
               removed_n=pr.removed_n, removed_str=pr.removed_str, LEN=pr.LEN, min_overlap=pr.min_overlap, rsoemo=pr.rsoemo,
               li_kmer_length=pr.li_kmer_length, avg_len=pr.avg_len, remove_pref_reads=int(p.remove_pref_reads))
    lib.alga_free_parsed_reads(C.byref(pr))
    return out


def pack_reads(codes, lens=None, stride_words=None):
    """codes[n, maxlen] uint8 in {0,1,2,3} (A C G T) -> words[n, stride] uint32 in the reference's layout
    (nucleotide i in bits 2i,2i+1 of a little-endian bit string; src/DataStructures/Read.cpp:40-68)."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, m = codes.shape
    W = (2 * m + 31) // 32
    if stride_words is None:
        stride_words = W
    out = np.zeros((n, stride_words), dtype=np.uint32)
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    lens = None if lens is None else np.asarray(lens)
    CH = 1 << 18                                                   # bounded temporaries (64 B of uint32 per nucleotide row chunk)
    for s0 in range(0, n, CH):
        c = codes[s0:s0 + CH]
        if lens is not None:
            c = np.where(np.arange(m)[None, :] < lens[s0:s0 + CH, None], c, 0).astype(np.uint8)
        pad = W * 16 - m
        if pad:
            c = np.concatenate([c, np.zeros((c.shape[0], pad), np.uint8)], axis=1)
        c = c.reshape(c.shape[0], W, 16).astype(np.uint32)
        out[s0:s0 + CH, :W] = np.bitwise_or.reduce(c << shifts, axis=2)
    return out


def derive_params(avg_len, trim_left=3, trim_right=3, scale=0.55):
    """(min_overlap, rsoe_min_overlap) as src/main.cpp:93-108 of the reference derives them
    (mixed float/int arithmetic with truncation, reproduced literally in float32)."""
    LEN = int(avg_len + trim_left + trim_right)
    sc = np.float32(scale)
    L = int(np.float32(LEN) * sc)
    rsoemo = int(np.float32(LEN) * (sc + np.float32(1)) / np.float32(2))
    return L, rsoemo


REDUCTION = {"auto": 0, "per_target": 1, "source_side": 2}     # alga_reduction
ERR_UNSUPPORTED = -7


class Engine:
    """One engine handle == one HIP device.  Mirrors the reference's GraphCreator life cycle:
    construct -> (setAlignFrom/To masks) -> build -> read the graph."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.alga_engine_create(int(device), C.byref(h))
        if rc:
            raise AlgaError(rc, "alga_engine_create(device=%d) failed -- no usable HIP device" % device)
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.alga_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise AlgaError(rc, (self._lib.alga_last_error(self._h) or b"").decode())

    def set_option(self, name, value):
        """alga_engine_set_option: "probe" ("auto" | "table" | "cluster"), "cluster_bucket_bias", "cluster_pairs", "cluster_order",
        "local_big_max", "auto_reduction_per_target"."""
        if name == "probe" and isinstance(value, str):
            value = PROBE[value]
        self._check(self._lib.alga_engine_set_option(self._h, name.encode(), int(value)))

    def reserve(self, n_nodes, max_len, min_overlap, n_edges_hint=0):
        """alga_engine_reserve: every device buffer of a build for this shape, ahead of the build."""
        self._check(self._lib.alga_engine_reserve(self._h, int(n_nodes), int(max_len), int(min_overlap), int(n_edges_hint)))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._lib.alga_engine_device_name(self._h, buf, 256)
        return buf.value.decode()

    @staticmethod
    def params(min_overlap, rsoe_min_overlap, collect_stats=False, reduction="auto"):
        p = PrefSufParams()
        load_library().alga_prefsuf_default_params(C.byref(p))
        p.min_overlap, p.rsoe_min_overlap, p.collect_stats = int(min_overlap), int(rsoe_min_overlap), int(bool(collect_stats))
        p.reduction = REDUCTION[reduction] if isinstance(reduction, str) else int(reduction)
        return p

    def last_stats(self):
        st = PrefSufStats()
        self._check(self._lib.alga_prefsuf_last_stats(self._h, C.byref(st)))
        return st.as_dict()

    # ---- duplicate / prefix-read removal + id compaction on the GPU ---------------------------
    def preprocess_nodes(self, rows, lens, remove_pref_reads=2, min_keep_len=0):
        """rows[2R, stride] u32 / lens[2R] i32 (-1 = removed) on the host -> DeviceNodeSet (device pointers, engine-owned)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        inp = PreprocessInput(rows.ctypes.data, int(rows.shape[1]) if rows.ndim == 2 else 1, lens.ctypes.data, n, int(remove_pref_reads), int(min_keep_len))
        out = DeviceNodeSet()
        self._check(self._lib.alga_preprocess_nodes(self._h, C.byref(inp), C.byref(out)))
        return out

    def ingest_device(self, file1, file2=None, **kw):
        """The whole input stage on the GPU (alga_ingest_device): files -> (DeviceNodeSet, info dict); raises AlgaError -7 for the
        inputs that stage does not take (file types other than .fasta / .fastq / .fq, remove_reads_with_n = 0)."""
        p = IngestParams()
        self._lib.alga_ingest_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        ds, info = DeviceNodeSet(), IngestInfo()
        self._check(self._lib.alga_ingest_device(self._h, file1.encode(), (file2 or "").encode() or None, C.byref(p), C.byref(ds), C.byref(info)))
        return ds, {k: getattr(info, k) for k, _ in IngestInfo._fields_}

    # ---- host buffers in, edges out (the drop-in call) --------------------------------------
    def prefsuf_host(self, words, lens, min_overlap, rsoe_min_overlap, align_from=None, align_to=None, collect_stats=False,
                     reduction="auto", twin_rows=False):
        """twin_rows: `words` holds the rows of the ODD nodes alone (n / 2 rows; alga_prefsuf_params.twin_rows)."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        stride = int(words.shape[1]) if words.ndim == 2 else (words.size // max(n // 2 if twin_rows else n, 1))
        keep = [words, lens]
        nd = _Nodes(words.ctypes.data, stride, lens.ctypes.data, n, None, None)
        if align_from is not None:
            af = np.ascontiguousarray(align_from, dtype=np.uint8); keep.append(af); nd.align_from = af.ctypes.data
        if align_to is not None:
            at = np.ascontiguousarray(align_to, dtype=np.uint8); keep.append(at); nd.align_to = at.ctypes.data
        p = self.params(min_overlap, rsoe_min_overlap, collect_stats, reduction)
        p.twin_rows = 1 if twin_rows else 0
        out = C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_prefsuf_build_host(self._h, C.byref(nd), C.byref(p), C.byref(out), C.byref(m)))
        try:
            e = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3]
            return e.reshape(-1, 3).copy()
        finally:
            self._lib.alga_free_edges(self._h, out)

    def host_array(self, shape, dtype):
        """numpy array over pinned host memory (alga_host_alloc); released when the array's base object goes (alga_host_free)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self._lib.alga_host_alloc(self._h, n)
        if not p:
            raise AlgaError(-1, (self._lib.alga_last_error(self._h) or b"").decode())
        lib, h = self._lib, self._h

        class _Owner:
            def __del__(self_inner):
                lib.alga_host_free(h, p)
        buf = (C.c_char * n).from_address(p)
        buf._owner = _Owner()
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def prefsuf_host_compact(self, words, lens, min_overlap, rsoe_min_overlap, twin_rows=False, repeat=1, as_triples=True):
        """alga_prefsuf_build_host_compact -> (edges int32[m, 3] rebuilt from the compact form (or None), best seconds of the C call alone)"""
        import time
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        nd = _Nodes(words.ctypes.data, int(words.shape[1]), lens.ctypes.data, n, None, None)
        p = self.params(min_overlap, rsoe_min_overlap)
        p.twin_rows = 1 if twin_rows else 0
        best, edges = None, None
        for it in range(repeat):
            out = CompactEdges()
            t = time.perf_counter()
            self._check(self._lib.alga_prefsuf_build_host_compact(self._h, C.byref(nd), C.byref(p), C.byref(out)))
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
            if it == repeat - 1 and as_triples:
                m = int(out.n_edges)
                deg = np.ctypeslib.as_array(C.cast(out.degree, C.POINTER(C.c_uint8)), shape=(n,)) if n else np.zeros(0, np.uint8)
                edges = np.empty((m, 3), dtype=np.int32)
                if m:
                    edges[:, 0] = np.repeat(np.arange(n, dtype=np.int32), deg)
                    edges[:, 1] = np.ctypeslib.as_array(C.cast(out.dst, C.POINTER(C.c_uint32)), shape=(m,)).view(np.int32)
                    edges[:, 2] = np.ctypeslib.as_array(C.cast(out.offset, C.POINTER(C.c_uint8)), shape=(m,))
            self._lib.alga_free_compact_edges(self._h, C.byref(out))
        return edges, best

    def prefsuf_host_timed(self, words, lens, min_overlap, rsoe_min_overlap, repeat=3, twin_rows=False, digest=False):
        """Wall time of the C call alone (alga_prefsuf_build_host + alga_free_edges; no Python-side copy of the result):
        -> (best seconds, n_edges, digest).  twin_rows: `words` holds the rows of the odd nodes alone (alga_prefsuf_params.twin_rows).
        digest: [count, position-weighted checksum] of the last repeat's host list (host_edges_digest; outside the timing), else None."""
        import time
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        nd = _Nodes(words.ctypes.data, int(words.shape[1]), lens.ctypes.data, n, None, None)
        p = self.params(min_overlap, rsoe_min_overlap)
        p.twin_rows = 1 if twin_rows else 0
        best, m_out, dg = None, 0, None
        for it in range(repeat):
            out, m = C.c_void_p(), C.c_uint64()
            t = time.perf_counter()
            self._check(self._lib.alga_prefsuf_build_host(self._h, C.byref(nd), C.byref(p), C.byref(out), C.byref(m)))
            dt = time.perf_counter() - t
            if digest and it == repeat - 1:
                dg = host_edges_digest(np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(int(m.value), 3))) if m.value else [0, 0]
            self._lib.alga_free_edges(self._h, out)
            best = dt if best is None else min(best, dt)
            m_out = int(m.value)
        return best, m_out, dg

    # ---- device-resident node set (torch tensors on this engine's device) --------------------
    @staticmethod
    def _nodes_from_torch(words, lens, align_from=None, align_to=None):
        assert words.is_cuda and lens.is_cuda and words.is_contiguous() and lens.is_contiguous()
        n = int(lens.shape[0])
        stride = int(words.shape[1])
        nd = _Nodes(words.data_ptr(), stride, lens.data_ptr(), n, None, None)
        if align_from is not None:
            nd.align_from = align_from.data_ptr()
        if align_to is not None:
            nd.align_to = align_to.data_ptr()
        return nd

    def prefsuf_device(self, words, lens, min_overlap, rsoe_min_overlap, align_from=None, align_to=None, stream=None,
                       collect_stats=False, reduction="auto"):
        """words: int32/uint32-viewed torch tensor [n, stride] on the device, lens: int32 [n].
        Returns (device pointer of alga_edge[n_edges], n_edges); the memory belongs to the engine."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, collect_stats, reduction)
        out = C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_prefsuf_build_device(self._h, C.byref(nd), C.byref(p), C.c_void_p(stream or 0),
                                                        C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def keys_device(self, words, lens, min_overlap, rsoe_min_overlap, node_begin, node_end, align_from=None, align_to=None, stream=None, want_meta_flag=False):
        """Minimizer keys + runs of the nodes [node_begin, node_end) -> (d_keys ptr, d_meta ptr) over all n nodes (this range
        filled; the caller all-gathers the rest in place), or None when the clustered probe does not take the input."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, False)
        out = NodeKeys()
        self._check(self._lib.alga_prefsuf_keys_device(self._h, C.byref(nd), C.byref(p), int(node_begin), int(node_end), C.c_void_p(stream or 0),
                                                       C.byref(out)))
        if not out.eligible:
            return None
        if want_meta_flag:
            return out.d_keys, out.d_meta, bool(out.meta_needed)
        return out.d_keys, out.d_meta

    def build_range_device(self, words, lens, min_overlap, rsoe_min_overlap, src_begin, src_end, align_from=None, align_to=None,
                           stream=None, collect_stats=False, keys_shared=False):
        """Final edges of the sources [src_begin, src_end) by the source-side reduction -> (ptr, n_edges), or None when
        that form is not exact for the input (ALGA_ERR_UNSUPPORTED): the caller then takes discover/exchange/reduce.
        keys_shared: 1/True = keys_device + the all-gather of its arrays came first; 2 = reuse the entry array of the previous build
        of the same node set (include/alga_amd.h)."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, collect_stats)
        p.keys_shared = int(keys_shared)
        out = C.c_void_p()
        m = C.c_uint64()
        rc = self._lib.alga_prefsuf_build_range_device(self._h, C.byref(nd), C.byref(p), int(src_begin), int(src_end),
                                                       C.c_void_p(stream or 0), C.byref(out), C.byref(m))
        if rc == ERR_UNSUPPORTED:
            return None
        self._check(rc)
        return out.value, int(m.value)

    # ---- the bucket-sharded N-GPU form, phase by phase (include/alga_amd.h: alga_shard_*); None = ALGA_ERR_UNSUPPORTED ----
    def shard_index_device(self, words, lens, min_overlap, rsoe_min_overlap, rank, n_ranks, align_from=None, align_to=None, stream=None):
        """-> (d_desc ptr [3 x u32 per descriptor], counts[n_ranks], offsets[n_ranks]) after keys_device + the key all-gather."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, False)
        out = C.c_void_p()
        cnt, off = (C.c_uint64 * n_ranks)(), (C.c_uint64 * n_ranks)()
        rc = self._lib.alga_shard_index_device(self._h, C.byref(nd), C.byref(p), int(rank), int(n_ranks), C.c_void_p(stream or 0), C.byref(out), cnt, off)
        if rc == ERR_UNSUPPORTED:
            return None
        self._check(rc)
        return out.value, [int(x) for x in cnt], [int(x) for x in off]

    def shard_join_device(self, words, lens, desc_in, n_desc, align_from=None, align_to=None, stream=None):
        """desc_in: device tensor / pointer of n_desc received descriptors -> (d_pending_src ptr [u32], n_pending) or None."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        out, m = C.c_void_p(), C.c_uint64()
        rc = self._lib.alga_shard_join_device(self._h, C.byref(nd), C.c_void_p(_ptr(desc_in)), C.c_uint64(int(n_desc)), C.c_void_p(stream or 0), C.byref(out), C.byref(m))
        if rc == ERR_UNSUPPORTED:
            return None
        self._check(rc)
        return out.value, int(m.value)

    def shard_small_keys_device(self, pending_all, n_all, stream=None):
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_shard_small_keys_device(self._h, C.c_void_p(_ptr(pending_all)), C.c_uint64(int(n_all)), C.c_void_p(stream or 0), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def shard_resolve_device(self, small_all, n_small_all, n_ranks, stream=None):
        """-> (d_edges ptr, counts[n_ranks], offsets[n_ranks]): final edges grouped by the rank that owns the source id"""
        out = C.c_void_p()
        cnt, off = (C.c_uint64 * n_ranks)(), (C.c_uint64 * n_ranks)()
        self._check(self._lib.alga_shard_resolve_device(self._h, C.c_void_p(_ptr(small_all)), C.c_uint64(int(n_small_all)), C.c_void_p(stream or 0), C.byref(out), cnt, off))
        return out.value, [int(x) for x in cnt], [int(x) for x in off]

    def shard_place_device(self, edges_in, n_in, src_begin, src_end, stream=None):
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_shard_place_device(self._h, C.c_void_p(_ptr(edges_in)), C.c_uint64(int(n_in)), int(src_begin), int(src_end), C.c_void_p(stream or 0),
                                                      C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def shard_stats(self):
        st = ShardStats()
        self._lib.alga_shard_last_stats.argtypes = [C.c_void_p, C.POINTER(ShardStats)]
        self._lib.alga_shard_last_stats(self._h, C.byref(st))
        return st.as_dict()

    def discover_device(self, words, lens, min_overlap, rsoe_min_overlap, src_begin, src_end, align_from=None,
                        align_to=None, stream=None, collect_stats=False):
        """-> (d_dst ptr [u32], d_val ptr [u64], n_record_slots); slots with dst == 0xFFFFFFFF are padding."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, collect_stats)
        d, v = C.c_void_p(), C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_prefsuf_discover_device(self._h, C.byref(nd), C.byref(p), int(src_begin), int(src_end),
                                                           C.c_void_p(stream or 0), C.byref(d), C.byref(v), C.byref(m)))
        return d.value, v.value, int(m.value)

    def reduce_device(self, words, lens, min_overlap, rsoe_min_overlap, rec_dst, rec_val, n_records, dst_begin,
                      dst_end, align_from=None, align_to=None, stream=None, collect_stats=False):
        """rec_dst (int32) / rec_val (int64): device pointers (ints) or torch tensors."""
        nd = self._nodes_from_torch(words, lens, align_from, align_to)
        p = self.params(min_overlap, rsoe_min_overlap, collect_stats)

        def ptr(x):
            return C.c_void_p(x if isinstance(x, int) else x.data_ptr())
        out = C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_prefsuf_reduce_device(self._h, C.byref(nd), C.byref(p), ptr(rec_dst), ptr(rec_val),
                                                         int(n_records), int(dst_begin), int(dst_end),
                                                         C.c_void_p(stream or 0), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def sort_records_device(self, rec_dst, rec_val, n_records, n_nodes, stream=None):
        """-> (d_dst_sorted ptr, d_val_sorted ptr, n_valid): records ordered by target id, padding dropped."""
        def ptr(x):
            return C.c_void_p(x if isinstance(x, int) else x.data_ptr())
        d, v = C.c_void_p(), C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_sort_records_device(self._h, ptr(rec_dst), ptr(rec_val), int(n_records), int(n_nodes),
                                                       C.c_void_p(stream or 0), C.byref(d), C.byref(v), C.byref(m)))
        return d.value, v.value, int(m.value)

    def sort_u32_pairs_device(self, keys, vals, begin_bit=0, own=True, repeat=1, stream=None):
        """alga_sort_u32_pairs_device: (key, value) int32 / uint32 tensors on this device, stable on the key bits [begin_bit, 32) ->
        (keys ptr, vals ptr, best ms).  own: the engine's radix sort (radix_sort.hip), else rocPRIM's."""
        n = int(keys.shape[0])
        ko, vo, ms = C.c_void_p(), C.c_void_p(), C.c_double()
        self._check(self._lib.alga_sort_u32_pairs_device(self._h, keys.data_ptr() if n else None, vals.data_ptr() if n else None, n, int(begin_bit), int(bool(own)),
                                                         int(repeat), C.c_void_p(stream or 0), C.byref(ko), C.byref(vo), C.byref(ms)))
        return ko.value, vo.value, ms.value

    def sort_u64_pairs_device(self, keys, vals, bits, own=True, repeat=1, stream=None):
        """alga_sort_u64_pairs_device: (key, value) int64 tensors on this device, stable on the key bits [0, bits) -> (keys ptr, vals ptr, best ms)"""
        n = int(keys.shape[0])
        ko, vo, ms = C.c_void_p(), C.c_void_p(), C.c_double()
        self._check(self._lib.alga_sort_u64_pairs_device(self._h, keys.data_ptr() if n else None, vals.data_ptr() if n else None, n, int(bits), int(bool(own)),
                                                         int(repeat), C.c_void_p(stream or 0), C.byref(ko), C.byref(vo), C.byref(ms)))
        return ko.value, vo.value, ms.value

    def sort_edges_device(self, edges, n_edges, n_nodes, stream=None):
        """edges: device pointer or int32 tensor [n_edges, 3] -> device pointer of the list ordered by (src, dst)."""
        p = C.c_void_p(edges if isinstance(edges, int) else edges.data_ptr())
        out = C.c_void_p()
        self._check(self._lib.alga_sort_edges_device(self._h, p, int(n_edges), int(n_nodes), C.c_void_p(stream or 0), C.byref(out)))
        return out.value

    # ---- approximate supplement ---------------------------------------------------------------
    @staticmethod
    def pkb_params(avg_len, error_rate, kmer_length_bucket, scale=0.55):
        p = PkbParams()
        load_library().alga_pkb_derive_params(float(avg_len), float(scale), float(error_rate), int(kmer_length_bucket), C.byref(p))
        return p

    @staticmethod
    def _host_nodes(words, lens):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        stride = int(words.shape[1]) if words.ndim == 2 else (words.size // max(n, 1))
        return _Nodes(words.ctypes.data, stride, lens.ctypes.data, n, None, None), (words, lens)

    def can_align_batch(self, words, lens, triples, p):
        nd, keep = self._host_nodes(words, lens)
        t = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(len(t), dtype=np.uint8)
        self._check(self._lib.alga_can_align_batch_host(self._h, C.byref(nd), C.byref(p), t.ctypes.data, len(t), out.ctypes.data))
        return out

    def li_kmers(self, words, lens, p, prio):
        nd, keep = self._host_nodes(words, lens)
        n, I = nd.n, p.li_intervals
        h = np.zeros((n, I), dtype=np.uint64)
        ind = np.zeros((n, I), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        pr = np.ascontiguousarray(prio, dtype=np.int32)
        self._check(self._lib.alga_li_kmers_host(self._h, C.byref(nd), C.byref(p), pr.ctypes.data, h.ctypes.data, ind.ctypes.data, cnt.ctypes.data))
        return h, ind, cnt

    def pkb_supplement_host(self, words, lens, edges_in, p):
        nd, keep = self._host_nodes(words, lens)
        e = np.ascontiguousarray(edges_in, dtype=np.int32).reshape(-1, 3)
        out = C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_pkb_supplement_host(self._h, C.byref(nd), C.byref(p), e.ctypes.data, len(e), C.byref(out), C.byref(m)))
        try:
            r = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3]
            return r.reshape(-1, 3).copy()
        finally:
            self._lib.alga_free_edges(self._h, out)

    def pkb_supplement_device(self, words, lens, d_edges, n_edges, p, stream=None):
        """node set (torch device tensors) and edge list (device pointer, sorted by (src, dst)) in HBM -> (ptr, n_edges) engine-owned"""
        nd = self._nodes_from_torch(words, lens, None, None)
        out = C.c_void_p()
        m = C.c_uint64()
        self._check(self._lib.alga_pkb_supplement_device(self._h, C.byref(nd), C.byref(p), C.c_void_p(d_edges), C.c_uint64(int(n_edges)),
                                                         C.c_void_p(stream or 0), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    # ---- the supplement on N ranks: begin -> { round -> [all-gather of the additions] -> merge } x rounds -> end (include/alga_amd.h) ----
    def pkb_shard_begin(self, words, lens, d_edges, n_edges, p, rank, n_ranks, stream=None):
        nd = self._nodes_from_torch(words, lens)
        self._check(self._lib.alga_pkb_shard_begin(self._h, C.byref(nd), C.byref(p), C.c_void_p(d_edges), int(n_edges), int(rank), int(n_ranks), C.c_void_p(stream or 0)))

    def pkb_shard_round(self, stream=None):
        """-> (device pointer of this rank's additions (uint64 keys), how many)"""
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_pkb_shard_round(self._h, C.c_void_p(stream or 0), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def pkb_shard_merge(self, d_all, n_all, stream=None):
        self._check(self._lib.alga_pkb_shard_merge(self._h, C.c_void_p(d_all), int(n_all), C.c_void_p(stream or 0)))

    def pkb_shard_end(self, stream=None):
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_pkb_shard_end(self._h, C.c_void_p(stream or 0), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def pkb_last_stats(self):
        st = PkbStats()
        self._check(self._lib.alga_pkb_last_stats(self._h, C.byref(st)))
        return dict(kmers=list(st.kmers), groups=list(st.groups), can_align_calls=list(st.can_align_calls),
                    edges_after=list(st.edges_after), max_group=st.max_group, ms_total=st.ms_total,
                    group_hist=[list(r) for r in st.group_hist])

    # ---- first simplifier step ----------------------------------------------------------------
    def cut_triangles_host(self, n_nodes, edges, max_offset_parallel_paths):
        """edges [m, 3] sorted by (src, dst) -> the graph after sortEdgesByIncreasingOffset + cutNonAndWeaklyMetricTriangles,
        lists in the reference's order."""
        e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 3)
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_cut_triangles_host(self._h, int(n_nodes), e.ctypes.data, len(e), int(max_offset_parallel_paths), C.byref(out), C.byref(m)))
        try:
            r = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3]
            return r.reshape(-1, 3).copy()
        finally:
            self._lib.alga_free_edges(self._h, out)

    def cut_triangles_device(self, n_nodes, d_edges_ptr, n_edges, max_offset_parallel_paths, stream=None):
        """device edge list (pointer) -> (device pointer, n_edges_out, n_removed); engine-owned."""
        out, m, rem = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.alga_cut_triangles_device(self._h, int(n_nodes), C.c_void_p(d_edges_ptr), int(n_edges), int(max_offset_parallel_paths),
                                                        C.c_void_p(stream or 0), C.byref(out), C.byref(m), C.byref(rem)))
        return out.value, int(m.value), int(rem.value)

    def contig_trim(self, words, lens, threshold=25):
        """src/main.cpp:636-697 on the GPU: packed contigs -> trim_left[n_contigs] (int32)."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        out = np.zeros(len(lens), dtype=np.int32)
        self._check(self._lib.alga_contig_trim_host(self._h, words.ctypes.data, int(words.shape[1]) if words.ndim == 2 else 1, lens.ctypes.data, len(lens),
                                                    int(threshold), out.ctypes.data))
        return out

    def write_graph(self, path, n_nodes, edges):
        edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 3)
        rc = self._lib.alga_write_graph(path.encode(), int(n_nodes), edges.ctypes.data, len(edges))
        if rc:
            raise AlgaError(rc, "alga_write_graph(%s) failed" % path)


class MultiEngine:
    """alga_multi_*: the N GPUs of one node behind one handle -- one process, one host thread and one engine per rank
    (alga_amd/csrc/engine_multi.hip).  devices may name the same GPU several times with transport="copy" (how a one-GPU box tests it)."""

    def __init__(self, devices, transport="auto"):
        self._lib = load_library()
        self._lib.alga_multi_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        self._lib.alga_multi_destroy.argtypes = [C.c_void_p]
        self._lib.alga_multi_destroy.restype = None
        self._lib.alga_multi_last_error.argtypes = [C.c_void_p]
        self._lib.alga_multi_last_error.restype = C.c_char_p
        self._lib.alga_multi_prefsuf_build_host.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        self._lib.alga_multi_prefsuf_build_device.argtypes = [C.c_void_p, C.POINTER(_Nodes), C.POINTER(PrefSufParams), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        self._lib.alga_multi_free_edges.argtypes = [C.c_void_p, C.c_void_p]
        self._lib.alga_multi_free_edges.restype = None
        self._lib.alga_multi_last_stats.argtypes = [C.c_void_p, C.POINTER(MultiStats), C.c_void_p]
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.alga_multi_create(devs, len(devices), TRANSPORT[transport] if isinstance(transport, str) else int(transport), C.byref(h))
        if rc:
            raise AlgaError(rc, "alga_multi_create(devices=%s, transport=%s) failed" % (list(devices), transport))
        self._h, self.n = h, len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.alga_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise AlgaError(rc, (self._lib.alga_multi_last_error(self._h) or b"").decode())

    def set_option(self, name, value):
        """alga_multi_set_option: "form" ("auto" | "replicated" | "bucket_sharded")."""
        if name == "form" and isinstance(value, str):
            value = MULTI_FORM[value]
        self._lib.alga_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self._check(self._lib.alga_multi_set_option(self._h, name.encode(), int(value)))

    def rank_shard_stats(self, rank):
        """alga_shard_last_stats of one rank's engine (the bucket-sharded form's counters and device times)."""
        self._lib.alga_multi_engine.argtypes = [C.c_void_p, C.c_int32]
        self._lib.alga_multi_engine.restype = C.c_void_p
        self._lib.alga_shard_last_stats.argtypes = [C.c_void_p, C.POINTER(ShardStats)]
        st = ShardStats()
        self._lib.alga_shard_last_stats(C.c_void_p(self._lib.alga_multi_engine(self._h, int(rank))), C.byref(st))
        return st.as_dict()

    def set_rank_option(self, rank, name, value):
        """alga_engine_set_option on ONE rank's engine (alga_multi_engine): tests make a single rank decline this way."""
        self._lib.alga_multi_engine.argtypes = [C.c_void_p, C.c_int32]
        self._lib.alga_multi_engine.restype = C.c_void_p
        self._lib.alga_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        eh = self._lib.alga_multi_engine(self._h, int(rank))
        if not eh:
            raise AlgaError(-1, "no such rank")
        rc = self._lib.alga_engine_set_option(C.c_void_p(eh), name.encode(), int(value))
        if rc:
            raise AlgaError(rc, "alga_engine_set_option(%s) on rank %d" % (name, rank))

    def prefsuf_host(self, words, lens, min_overlap, rsoe_min_overlap, align_from=None, align_to=None, collect_stats=False, reduction="auto", twin_rows=False):
        """twin_rows: `words` holds the rows of the odd nodes alone (alga_prefsuf_params.twin_rows)"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        n = int(lens.shape[0])
        stride = int(words.shape[1]) if words.ndim == 2 else (words.size // max(n // 2 if twin_rows else n, 1))
        keep = [words, lens]
        nd = _Nodes(words.ctypes.data, stride, lens.ctypes.data, n, None, None)
        if align_from is not None:
            af = np.ascontiguousarray(align_from, dtype=np.uint8); keep.append(af); nd.align_from = af.ctypes.data
        if align_to is not None:
            at = np.ascontiguousarray(align_to, dtype=np.uint8); keep.append(at); nd.align_to = at.ctypes.data
        p = Engine.params(min_overlap, rsoe_min_overlap, collect_stats, reduction)
        p.twin_rows = 1 if twin_rows else 0
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_multi_prefsuf_build_host(self._h, C.byref(nd), C.byref(p), C.byref(out), C.byref(m)))
        try:
            e = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3]
            return e.reshape(-1, 3).copy()
        finally:
            self._lib.alga_multi_free_edges(self._h, out)

    def prefsuf_device(self, per_rank, min_overlap, rsoe_min_overlap):
        """per_rank: [(words, lens)] torch tensors on each rank's device (the same node set) -> (device pointer on rank 0's GPU, n_edges)."""
        arr = (_Nodes * self.n)(*[Engine._nodes_from_torch(w, l) for w, l in per_rank])
        p = Engine.params(min_overlap, rsoe_min_overlap)
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_multi_prefsuf_build_device(self._h, arr, C.byref(p), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def pkb_supplement_device(self, per_rank, d_edges_rank0, n_edges, p):
        """alga_multi_pkb_supplement_device: the approximate supplement on the handle's ranks; the exact graph on rank 0's GPU (what prefsuf_device
        returned) -> (device pointer on rank 0's GPU, n_edges)."""
        self._lib.alga_multi_pkb_supplement_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PkbParams), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        arr = (_Nodes * self.n)(*[Engine._nodes_from_torch(w, l) for w, l in per_rank])
        out, m = C.c_void_p(), C.c_uint64()
        self._check(self._lib.alga_multi_pkb_supplement_device(self._h, arr, C.byref(p), C.c_void_p(d_edges_rank0), int(n_edges), C.byref(out), C.byref(m)))
        return out.value, int(m.value)

    def last_stats(self):
        st = MultiStats()
        per = (PrefSufStats * self.n)()
        self._check(self._lib.alga_multi_last_stats(self._h, C.byref(st), C.cast(per, C.c_void_p)))
        d = {k: getattr(st, k) for k, _ in st._fields_}
        d["ranks"] = [x.as_dict() for x in per]
        return d


def _ptr(x):
    """device pointer of a torch tensor (or an int that already is one; None / empty -> 0)"""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    return x.data_ptr() if x.numel() else 0


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, shape, device=None, typestr="<i4"):
    """torch tensor (int32, or int64 with typestr "<i8") aliasing `ptr` (no copy).  Valid until the engine reuses
    the buffer."""
    import torch
    if int(np.prod(shape)) == 0:
        return torch.empty(tuple(shape), dtype=torch.int32 if typestr == "<i4" else torch.int64, device=device or "cuda")
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device or "cuda")


def host_edges_digest(e):
    """alga_amd.multigpu.edges_digest for a HOST edge list (numpy int32 [m, 3]): the same wrap-around int64 arithmetic -> [count, checksum]"""
    k = int(e.shape[0])
    acc = np.uint64(0)
    with np.errstate(over="ignore"):
        for s0 in range(0, k, 1 << 24):
            c = e[s0:s0 + (1 << 24)].astype(np.int64)
            w = np.arange(s0 + 1, s0 + c.shape[0] + 1, dtype=np.int64) | 1
            v = ((c[:, 0] * 1000003 + c[:, 1]) * 10007 + c[:, 2]) * w
            acc = acc + v.view(np.uint64).sum(dtype=np.uint64)
    return [k, int(np.array([acc], dtype=np.uint64).view(np.int64)[0])]


def device_edges_to_numpy(ptr, n_edges):
    """Copy an engine-owned device edge list to host (torch is only the memcpy)."""
    if n_edges == 0:
        return np.zeros((0, 3), np.int32)
    return device_view(ptr, (n_edges, 3)).cpu().numpy().copy()
