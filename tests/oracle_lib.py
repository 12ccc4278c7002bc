"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import gzip
import json
import os
import shutil
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class Nodes(C.Structure):
    _fields_ = [("n", C.c_int32), ("W", C.c_int32), ("words", C.POINTER(C.c_uint32)), ("len", C.POINTER(C.c_int32)),
                ("pair_off", C.POINTER(C.c_uint8)), ("LEN", C.c_int32), ("min_overlap", C.c_int32),
                ("rsoemo", C.c_int32), ("li_kmer_length", C.c_int32), ("reads_in_file", C.c_int64),
                ("removed_n", C.c_int32), ("removed_str", C.c_int32), ("removed_prefix", C.c_int32),
                ("avg_len", C.c_double)]


class IngestParams(C.Structure):
    _fields_ = [("trim_left", C.c_int32), ("trim_right", C.c_int32), ("remove_reads_with_n", C.c_int32),
                ("rna", C.c_int32), ("scale", C.c_float), ("min_overlap", C.c_int32), ("rsoemo", C.c_int32),
                ("remove_pref_reads", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("n_edges", C.c_int64), ("edges_after_iter", C.POINTER(C.c_int64)),
                ("n_iters", C.c_int32), ("bucket_entries_scanned", C.c_int64), ("hash_equal_pairs", C.c_int64),
                ("transitive_checks", C.c_int64), ("transitive_removed", C.c_int64)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        _lib.oracle_ingest.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(IngestParams), C.POINTER(Nodes)]
        _lib.oracle_prefsuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_int32, C.POINTER(Graph)]
        _lib.oracle_min_period.argtypes = [C.c_char_p, C.c_int]
        _lib.oracle_pack.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    return _lib


def ingest(file1, file2=None, min_overlap=-1, rsoemo=-1, **kw):
    """-> dict(words[n,W] u32, len[n] i32, pair_off[n] u8, min_overlap, rsoemo, ...)"""
    L = lib()
    p = IngestParams()
    L.oracle_default_ingest_params(C.byref(p))
    p.min_overlap, p.rsoemo = min_overlap, rsoemo
    for k, v in kw.items():
        setattr(p, k, v)
    nd = Nodes()
    rc = L.oracle_ingest(file1.encode(), (file2 or "").encode(), C.byref(p), C.byref(nd))
    if rc:
        raise RuntimeError("oracle_ingest failed rc=%d" % rc)
    n, W = nd.n, nd.W
    out = dict(n=n, W=W,
               words=np.ctypeslib.as_array(nd.words, shape=(max(n, 1) * W,))[: n * W].reshape(n, W).copy(),
               len=np.ctypeslib.as_array(nd.len, shape=(max(n, 1),))[:n].copy(),
               pair_off=np.ctypeslib.as_array(nd.pair_off, shape=(max(n, 1),))[:n].copy(),
               LEN=nd.LEN, min_overlap=nd.min_overlap, rsoemo=nd.rsoemo, li_kmer_length=nd.li_kmer_length,
               removed_prefix=nd.removed_prefix, removed_n=nd.removed_n, removed_str=nd.removed_str)
    L.oracle_free_nodes(C.byref(nd))
    return out


def prefsuf(words, lens, min_overlap, rsoemo, align_from=None, align_to=None):
    """-> (edges[m,3] i32 sorted by (src,dst,off), edges_after_iter, counters dict)"""
    L = lib()
    words = np.ascontiguousarray(words, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n = lens.shape[0]
    W = words.shape[1] if words.ndim == 2 else (words.size // max(n, 1))
    g = Graph()
    af = None if align_from is None else np.ascontiguousarray(align_from, dtype=np.uint8)
    at = None if align_to is None else np.ascontiguousarray(align_to, dtype=np.uint8)
    rc = L.oracle_prefsuf(words.ctypes.data, lens.ctypes.data, n, W,
                          None if af is None else af.ctypes.data, None if at is None else at.ctypes.data,
                          min_overlap, rsoemo, C.byref(g))
    if rc:
        raise RuntimeError("oracle_prefsuf failed")
    m = g.n_edges
    e = np.ctypeslib.as_array(C.cast(g.edges, C.POINTER(C.c_int32)), shape=(max(m, 1) * 3,))[: m * 3].reshape(m, 3).copy()
    it = np.ctypeslib.as_array(g.edges_after_iter, shape=(max(g.n_iters, 1),))[: g.n_iters].copy()
    cnt = dict(scanned=g.bucket_entries_scanned, hash_equal=g.hash_equal_pairs,
               transitive_checks=g.transitive_checks, transitive_removed=g.transitive_removed)
    L.oracle_free_graph(C.byref(g))
    return e, it, cnt


def revcomp_rows(words, lens):
    """2-bit rows of the reverse complements (MyUtils::getComplimentaryString(getReverse(.)))"""
    out = np.zeros_like(words)
    for i in range(len(lens)):
        n = int(lens[i])
        c = np.array([(int(words[i, k >> 4]) >> ((k & 15) << 1)) & 3 for k in range(n)], dtype=np.uint8)
        rc = (3 - c)[::-1]
        for k in range(n):
            out[i, k >> 4] |= np.uint32(int(rc[k]) << ((k & 15) << 1))
    return out


def contig_trim(words, lens, threshold=25):
    """src/main.cpp:636-697 with the oracle's creator: nodes = contigs then their reverse complements; trimLeft per contig"""
    M = len(lens)
    w2 = np.concatenate([words, revcomp_rows(words, lens)], axis=0)
    l2 = np.concatenate([lens, lens]).astype(np.int32)
    e, _, _ = prefsuf(w2, l2, threshold, threshold)
    trim = np.zeros(M, dtype=np.int32)
    for a, d, off in e:
        if a < M and d < M:
            trim[d] = max(trim[d], int(l2[a]) - int(off))
    return trim


def cut_triangles(n, edges, mopp):
    """first simplifier step (oracle_cut_triangles) -> edges [m, 3] grouped by src, lists in the reference's order"""
    L = lib()
    L.oracle_cut_triangles.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 3)
    out, m = C.c_void_p(), C.c_int64()
    rc = L.oracle_cut_triangles(int(n), e.ctypes.data, len(e), int(mopp), C.byref(out), C.byref(m))
    assert rc == 0
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3].reshape(-1, 3).copy()
    C.CDLL(None).free(out)
    return res


def graph_bytes(n, edges):
    """Graph::serializeGraph wire format (src/DataStructures/Graph.cpp:269-297) from sorted edge triples."""
    edges = np.asarray(edges, dtype=np.int32).reshape(-1, 3)
    deg = np.bincount(edges[:, 0], minlength=n).astype(np.int32) if len(edges) else np.zeros(n, np.int32)
    out = np.empty(1 + 2 * n + 2 * len(edges), dtype=np.int32)
    out[0] = n
    start = np.concatenate([[0], np.cumsum(deg)[:-1]]) if n else np.zeros(0, np.int64)
    hdr = 1 + 2 * np.arange(n, dtype=np.int64) + 2 * start
    out[hdr] = np.arange(n, dtype=np.int32)
    out[hdr + 1] = deg
    if len(edges):
        pos = np.arange(len(edges), dtype=np.int64)
        base = hdr[edges[:, 0]] + 2 + 2 * (pos - start[edges[:, 0]])
        out[base] = edges[:, 1]
        out[base + 1] = edges[:, 2]
    return out.tobytes()


def parse_graph(buf):
    """inverse of graph_bytes -> (n, edges[m,3])"""
    a = np.frombuffer(buf, dtype=np.int32)
    n = int(a[0])
    p = 1
    src, dst, off = [], [], []
    for _ in range(n):
        i, t = int(a[p]), int(a[p + 1])
        p += 2
        if t:
            blk = a[p: p + 2 * t].reshape(t, 2)
            src.append(np.full(t, i, np.int32)); dst.append(blk[:, 0]); off.append(blk[:, 1])
            p += 2 * t
    if not src:
        return n, np.zeros((0, 3), np.int32)
    return n, np.stack([np.concatenate(src), np.concatenate(dst), np.concatenate(off)], axis=1).astype(np.int32)


class Fixture:
    """One golden fixture (tests/golden/<name>.json + inputs + the reference's graph dump)."""

    def __init__(self, golden_dir, name):
        self.name = name
        self.dir = golden_dir
        with open(os.path.join(golden_dir, name + ".json")) as f:
            self.meta = json.load(f)
        self._tmp = None

    def inputs(self):
        if self._tmp is None:
            self._tmp = tempfile.mkdtemp(prefix="alga_fx_")
            for gz in self.meta["inputs"]:
                with gzip.open(os.path.join(self.dir, gz), "rb") as fi, open(os.path.join(self._tmp, gz[:-3]), "wb") as fo:
                    shutil.copyfileobj(fi, fo)
        ps = [os.path.join(self._tmp, gz[:-3]) for gz in self.meta["inputs"]]
        return ps[0], (ps[1] if len(ps) > 1 else None)

    def ref_graph(self):
        with gzip.open(os.path.join(self.dir, self.meta["graph"]), "rb") as f:
            return f.read()

    def explicit_params(self):
        lo, rs = -1, -1
        ex = self.meta.get("extra_args", [])
        for i, a in enumerate(ex):
            if a == "-l":
                lo = int(ex[i + 1])
            if a.startswith("--rsoemo="):
                rs = int(a.split("=")[1])
        return lo, rs

    def cleanup(self):
        if self._tmp:
            shutil.rmtree(self._tmp, ignore_errors=True)
            self._tmp = None


FIXTURES = ["f1_cfg1", "f2_err2", "f3_paired", "f4_varlen", "f5_messy", "f6_l40"]


# ---- approximate supplement (oracle/alga_oracle_pkb.cpp) -----------------------------------------------------------
class PkbParams(C.Structure):
    _fields_ = [("min_overlap_area", C.c_int32), ("max_offset_pct", C.c_int32), ("min_identity_pct", C.c_int32),
                ("same_ends", C.c_int32), ("li_k", C.c_int32), ("li_intervals", C.c_int32), ("rounds", C.c_int32)]


def pkb_params(avg_len, scale=0.55, error_rate_percent=2):
    L = lib()
    L.oracle_pkb_derive_params.argtypes = [C.c_double, C.c_float, C.c_int, C.POINTER(PkbParams)]
    p = PkbParams()
    L.oracle_pkb_derive_params(float(avg_len), float(scale), int(error_rate_percent), C.byref(p))
    return p


def can_align(words, lens, triples, p):
    L = lib()
    L.oracle_can_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PkbParams)]
    words = np.ascontiguousarray(words, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    W = words.shape[1]
    out = np.zeros(len(triples), dtype=np.uint8)
    for i, (a, b, o) in enumerate(np.asarray(triples).tolist()):
        out[i] = L.oracle_can_align(words.ctypes.data, lens.ctypes.data, W, a, b, o, C.byref(p))
    return out


def li_kmers(row, length, k, intervals, prio):
    L = lib()
    L.oracle_li_kmers.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    row = np.ascontiguousarray(row, dtype=np.uint32)
    pr = np.ascontiguousarray(prio, dtype=np.int32)
    h = np.zeros(intervals, dtype=np.uint64)
    ind = np.zeros(intervals, dtype=np.int32)
    c = L.oracle_li_kmers(row.ctypes.data, int(length), k, intervals, pr.ctypes.data, h.ctypes.data, ind.ctypes.data)
    return h[:c].copy(), ind[:c].copy()


def supplement(words, lens, edges_in, p, kmer_length_bucket, flags=0):
    """-> (edges[m,3] sorted, number of canAlign calls); flags: 1 = ties by read id, 2 = round-snapshot semantics"""
    L = lib()
    L.oracle_supplement.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(PkbParams),
                                    C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    words = np.ascontiguousarray(words, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    e = np.ascontiguousarray(edges_in, dtype=np.int32).reshape(-1, 3)
    out = C.c_void_p()
    m = C.c_int64()
    calls = C.c_int64()
    rc = L.oracle_supplement(words.ctypes.data, lens.ctypes.data, len(lens), words.shape[1], e.ctypes.data, len(e), C.byref(p),
                             int(kmer_length_bucket), int(flags), C.byref(out), C.byref(m), C.byref(calls))
    if rc:
        raise RuntimeError("oracle_supplement failed")
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(m.value, 1) * 3,))[: m.value * 3].reshape(-1, 3).copy()
    C.CDLL(None).free(out)
    return res, int(calls.value)


def load_nodes_bin(path_gz):
    with gzip.open(path_gz, "rb") as f:
        buf = f.read()
    n, W = np.frombuffer(buf[:8], dtype=np.int32)
    lens = np.frombuffer(buf[8: 8 + 4 * n], dtype=np.int32).copy()
    words = np.frombuffer(buf[8 + 4 * n:], dtype=np.uint32).reshape(n, W).copy()
    return words, lens
