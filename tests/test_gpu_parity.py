"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the reference dumps.

Bar: bit-exact edge sets (integer work).  The exact path of the reference decides on two modular hashes
(src/GraphCreators/GraphCreatorPrefSuf.cpp:386-387) where the engine compares 2-bit words exactly, so the
two can differ only on a double hash collision (p ~ 1e-27 per pair).
"""
import os

import numpy as np
import pytest

import alga_amd
import gen_reads
import oracle_lib as O
from source_side_rule import preconditions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = alga_amd.Engine(0)
    yield e
    e.close()


def _check(eng, words, lens, lo, rs, af=None, at=None, stats=True, source_side=None):
    """Both forms of the transitive reduction against the oracle: the per-target replay always, the source-side form
    whenever it claims to be exact for the input (source_side=False: expected to decline, e.g. capacity)."""
    want, _, cnt = O.prefsuf(words, lens, lo, rs, af, at)
    got = eng.prefsuf_host(words, lens, lo, rs, af, at, collect_stats=stats, reduction="per_target")
    assert got.shape == want.shape, (got.shape, want.shape)
    assert (got == want).all()
    st = eng.last_stats()
    assert st["reduction_used"] == 1
    if stats:
        assert st["raw_overlaps"] == cnt["hash_equal"]
        assert st["transitive_listed"] == cnt["transitive_checks"]
        assert st["transitive_removed"] == cnt["transitive_removed"]
        assert st["edges"] == len(want)
    exact = preconditions(lens, lo, rs, af, at) and int(np.max(lens, initial=0)) - lo <= 127
    if source_side is None:
        source_side = exact
    if source_side:
        # the source-side form through BOTH probes: the bucketised seed table and the clustered minimizer join (round 4: up to 128 suffix
        # windows -- 250-bp reads, the two-word form -- and rows of up to 17 words; anything else goes to the table probe)
        maxlen = int(np.max(lens, initial=0))
        nwin = maxlen - lo + 1
        kk = min(32, max(lo - 63, min(lo, 16)))              # cluster_plan (alga_amd/csrc/prefsuf_cluster.hip)
        if min(nwin, 64) > min(lo - kk + 1, 64):
            kk = lo - min(nwin, 64) + 1
        clusterable = 1 <= nwin <= 128 and (2 * maxlen + 31) // 32 <= 17 and 8 <= kk <= min(32, lo)
        # the clustered probe with k_probe_stream first (the default), sources in key order and in id order, and its general kernel alone
        for probe, first_kernel, order in (("table", 1, 1), ("cluster", 1, 1), ("cluster", 1, 0), ("cluster", 0, 1)):
            eng.set_option("probe", probe)
            eng.set_option("cluster_pairs", first_kernel)
            eng.set_option("cluster_order", order)
            try:
                got2 = eng.prefsuf_host(words, lens, lo, rs, af, at, collect_stats=stats, reduction="source_side")
            finally:
                eng.set_option("probe", "auto")
                eng.set_option("cluster_pairs", 1)
                eng.set_option("cluster_order", 1)
            assert got2.shape == want.shape and (got2 == want).all(), (probe, first_kernel, order)
            st = eng.last_stats()
            assert st["reduction_used"] == 2 and st["edges"] == len(want)
            assert st["probe_used"] == (2 if probe == "cluster" and clusterable else 1)
            if stats:
                assert st["raw_overlaps"] == cnt["hash_equal"], probe
    else:
        with pytest.raises(alga_amd.AlgaError) as ei:
            eng.prefsuf_host(words, lens, lo, rs, af, at, reduction="source_side")
        assert ei.value.code == -7
    got3 = eng.prefsuf_host(words, lens, lo, rs, af, at)                      # automatic choice
    assert got3.shape == want.shape and (got3 == want).all()
    assert eng.last_stats()["reduction_used"] == (2 if source_side else 1)
    return got


@pytest.mark.parametrize("name", O.FIXTURES)
def test_golden_fixture_bit_exact(eng, golden_dir, name):
    fx = O.Fixture(golden_dir, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        nd = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
    finally:
        fx.cleanup()
    got = _check(eng, nd["words"], nd["len"], nd["min_overlap"], nd["rsoemo"])
    assert O.graph_bytes(nd["n"], got) == fx.ref_graph()          # the reference's own dump, byte for byte


def _nodes(n, length, G, seed, err=0.0, min_length=None, both_strands=True, stride=None):
    codes, lens = gen_reads.sample_reads(n, length, G, seed, err, min_length)
    if both_strands:  # node 2i = reverse complement, 2i+1 = forward, as the reference's input stage lays them out
        rc = np.zeros_like(codes)
        for i in range(n):
            rc[i, : lens[i]] = 3 - codes[i, : lens[i]][::-1]
        codes = np.stack([rc, codes], axis=1).reshape(2 * n, length)
        lens = np.repeat(lens, 2)
    return alga_amd.pack_reads(codes, lens, stride), lens.astype(np.int32)


@pytest.mark.parametrize("n,length,G,seed,err,minlen,lo,rs", [
    (3000, 100, 6000, 11, 0.0, None, 55, 77),
    (3000, 150, 9000, 12, 0.01, None, 82, 116),
    (2500, 150, 5000, 13, 0.0, 90, 70, 100),      # variable length, duplicates and prefix reads NOT removed
    (2000, 64, 3000, 14, 0.0, None, 20, 40),      # W=4, short seed (40 bits)
    (1500, 150, 3000, 15, 0.03, 120, 16, 60),     # one-word seed
    (1200, 250, 6000, 16, 0.0, 200, 140, 190),    # W=16
    (3000, 150, 9000, 19, 0.01, None, 90, 120),   # 150 nt, span 60: source-side form with errors
    (2500, 144, 5000, 20, 0.0, 100, 82, 116),     # variable length incl. contained / prefix reads, span 62
    (2500, 250, 9000, 28, 0.0, None, 137, 190),   # 250 nt: two-word offset masks, 256-bit overhangs (span 113)
    (2500, 250, 6000, 29, 0.003, 180, 125, 180),  # same with errors, variable length, span 125
    (900, 250, 9000, 30, 0.0, None, 137, 190),    # low coverage at 250 nt: several survivors per source
    (700, 100, 6000, 17, 0.0, None, 55, 77),      # low coverage: gaps too long for any big via (all-pairs path)
    (3000, 100, 5000, 18, 0.004, 80, 50, 70),     # errors + variable length: several items per offset
])
def test_random_sets_bit_exact(eng, n, length, G, seed, err, minlen, lo, rs):
    words, lens = _nodes(n, length, G, seed, err, minlen)
    _check(eng, words, lens, lo, rs)


def test_masks_and_removed_nodes(eng):
    words, lens = _nodes(2500, 120, 5000, 21)
    rng = np.random.default_rng(21)
    af = (rng.random(len(lens)) < 0.7).astype(np.uint8)
    at = (rng.random(len(lens)) < 0.7).astype(np.uint8)
    dead = rng.random(len(lens)) < 0.1
    lens = lens.copy(); lens[dead] = 0; words = words.copy(); words[dead] = 0
    _check(eng, words, lens, 66, 90, af, at)
    _check(eng, words, lens, 66, 90, af, np.maximum(af, at))     # alignFrom implies alignTo: the source-side form applies


def test_row_stride_padding(eng):
    words, lens = _nodes(1500, 100, 3000, 22, stride=12)
    assert words.shape[1] == 12
    _check(eng, words, lens, 55, 77)


def test_long_reads_cap_500(eng):
    # reads longer than 500 nt: overlap lengths stop at 501 (GraphCreatorPrefSuf.cpp:92-100), several 64-window chunks
    words, lens = _nodes(500, 640, 4000, 23, min_length=420)
    _check(eng, words, lens, 300, 420)
    _check(eng, words, lens, 120, 330)


def test_rsoemo_outside_iteration_range(eng):
    words, lens = _nodes(1500, 100, 3000, 24)
    _check(eng, words, lens, 55, 40)       # rsoemo < min_overlap: no small phase at all
    _check(eng, words, lens, 55, 55)       # reversal in the very first iteration
    _check(eng, words, lens, 55, 101)      # rsoemo == last iteration (L = maxlen+1)
    _check(eng, words, lens, 55, 150)      # never reversed mid-way: the reference hands back the reversed graph
    _check(eng, words, lens, 55, 0)


def test_degenerate_inputs(eng):
    z = np.zeros((0, 4), np.uint32)
    assert eng.prefsuf_host(z, np.zeros(0, np.int32), 55, 77).shape == (0, 3)
    words, lens = _nodes(200, 60, 2000, 25)
    assert eng.prefsuf_host(words, np.zeros_like(lens), 30, 40).shape == (0, 3)      # every node removed
    _check(eng, words, lens, 61, 70)                                                   # min_overlap > every read
    _check(eng, words, lens, 60, 60)                                                   # only full-length overlaps
    # heavy repeats: 400 copies of 5 distinct reads (long seed-table chains, big in-lists, ties in the cap)
    codes, l5 = gen_reads.sample_reads(5, 80, 120, 26)
    codes80 = np.repeat(codes, 80, axis=0)
    w = alga_amd.pack_reads(codes80)
    # more raw overlaps per source than a wave's LDS holds (160): those sources go through the second pass (items in global memory)
    _check(eng, w, np.full(len(w), 80, np.int32), 40, 60)
    eng.prefsuf_host(w, np.full(len(w), 80, np.int32), 40, 60, reduction="source_side")
    assert eng.last_stats()["big_sources"] > 0
    # 25 copies: 65..160 raw overlaps per source, the multi-round all-pairs path of the source-side form in LDS
    w25 = alga_amd.pack_reads(np.repeat(codes, 25, axis=0))
    _check(eng, w25, np.full(len(w25), 80, np.int32), 40, 60)
    # the same with 250 nt reads: second pass in the two-word form (offsets up to 113, 256-bit overhangs)
    codes250, _ = gen_reads.sample_reads(6, 250, 420, 27)
    w250 = alga_amd.pack_reads(np.repeat(codes250, 60, axis=0))
    _check(eng, w250, np.full(len(w250), 250, np.int32), 137, 190)
    eng.prefsuf_host(w250, np.full(len(w250), 250, np.int32), 137, 190, reduction="source_side")
    assert eng.last_stats()["big_sources"] > 0
    # beyond the largest item slice the engine allocates for the second pass (4096 per wave; lowered here) the source-side form
    # declines and AUTO falls back to the per-target pipeline
    small = alga_amd.Engine(0)
    small.set_option("local_big_max", 200)
    try:
        _check(small, w, np.full(len(w), 80, np.int32), 40, 60, source_side=False)
    finally:
        small.close()


def _tandem_nodes(seed, n_reads, genome_len, length, period):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, genome_len, dtype=np.uint8)
    for s0 in range(0, genome_len - 400, 1500):
        for k in range(1, 300 // period):
            g[s0 + k * period: s0 + (k + 1) * period] = g[s0: s0 + period]
    starts = np.unique(rng.integers(0, genome_len - length, n_reads))
    fw = np.stack([g[p: p + length] for p in starts])
    fw = np.unique(fw, axis=0)
    codes = np.stack([3 - fw[:, ::-1], fw], axis=1).reshape(-1, length)
    return alga_amd.pack_reads(codes), np.full(len(codes), length, np.int32)


@pytest.mark.parametrize("seed,n_reads,period", [(41, 700, 5), (42, 2500, 7), (43, 1200, 23)])
def test_tandem_repeats_same_target_at_several_offsets(eng, seed, n_reads, period):
    words, lens = _tandem_nodes(seed, n_reads, 6000, 60, period)
    _check(eng, words, lens, 33, 46)
    st_needed = eng.prefsuf_host(words, lens, 33, 46, collect_stats=True, reduction="source_side")
    assert eng.last_stats()["generic_sources"] > 0           # the all-pairs path of the source-side form really ran


def test_source_side_range_build_needs_no_exchange(eng):
    import torch
    from alga_amd.engine import device_edges_to_numpy
    words, lens = _nodes(4000, 150, 9000, 32, err=0.002)
    want, _, _ = O.prefsuf(words, lens, 90, 120)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    n = len(lens)
    parts = []
    for a, b in ((0, n // 3), (n // 3, n // 3), (n // 3, n // 2 + 1), (n // 2 + 1, n)):
        r = eng.build_range_device(dw, dl, 90, 120, a, b)
        assert r is not None
        parts.append(device_edges_to_numpy(*r))
        assert len(parts[-1]) == 0 or (parts[-1][:, 0].min() >= a and parts[-1][:, 0].max() < b)
    assert (np.concatenate(parts) == want).all()              # ranges in order: already the single-GPU byte order
    # an input the source-side form declines (reads longer than min_overlap + 127)
    words, lens = _nodes(300, 250, 3000, 33)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    assert eng.build_range_device(dw, dl, 100, 190, 0, len(lens)) is None
    r = eng.build_range_device(dw, dl, 137, 190, 0, len(lens))       # 250 nt, span 113: the two-word form of the source-side reduction
    want250, _, _ = O.prefsuf(words, lens, 137, 190)
    assert r is not None and (device_edges_to_numpy(*r) == want250).all()


@pytest.mark.parametrize("pairs", [1, 0])
def test_corrupt_index_fails_closed(pairs):
    """A clustered index built over UNSORTED keys (test-only option: the sort is skipped, as if a library sort had compared the wrong
    bits -- round 3's GPU fault): the directory pass flags the key order, both probe kernels clamp their entry reads, and the build
    answers ALGA_ERR_HIP instead of faulting the GPU.  The same engine then builds the right graph."""
    words, lens = _nodes(20000, 100, 40000, 91)
    want, _, _ = O.prefsuf(words, lens, 55, 77)
    e = alga_amd.Engine(0)
    try:
        e.set_option("probe", "cluster")
        e.set_option("cluster_pairs", pairs)
        e.set_option("test_unsorted_index", 1)
        with pytest.raises(alga_amd.AlgaError) as ei:
            e.prefsuf_host(words, lens, 55, 77, reduction="source_side")
        assert ei.value.code == -3 and "not in order" in str(ei.value)
        e.set_option("test_unsorted_index", 0)
        got = e.prefsuf_host(words, lens, 55, 77, reduction="source_side")
        assert got.shape == want.shape and (got == want).all()
        assert e.last_stats()["probe_used"] == 2
    finally:
        e.close()


@pytest.mark.parametrize("n,length,G,seed,err,minlen,lo,rs", [(3000, 100, 6000, 92, 0.0, None, 55, 77), (2500, 144, 5000, 93, 0.004, 100, 82, 116),
                                                             (1200, 250, 9000, 94, 0.0, 180, 125, 180)])
def test_twin_rows_upload_equals_full_upload(eng, n, length, G, seed, err, minlen, lo, rs):
    """ALGA's node set in twin layout (node 2k = reverse complement of node 2k + 1), given by the rows of its ODD nodes alone
    (alga_prefsuf_params.twin_rows): the even rows are rebuilt on the device (k_expand_twins) -- same graph as the full upload, also with
    variable lengths, removed nodes (even alone, or both), rows of 16 words; a pair of different lengths is refused."""
    words, lens = _nodes(n, length, G, seed, err, minlen)
    lens = lens.copy()
    rng = np.random.default_rng(seed)
    gone = rng.random(len(lens) // 2) < 0.05
    lens[0::2][gone] = 0                                             # the even node alone removed ...
    lens[1::2][gone & (rng.random(len(gone)) < 0.5)] = 0             # ... or the pair
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    for red in ("auto", "per_target"):
        got = eng.prefsuf_host(np.ascontiguousarray(words[1::2]), lens, lo, rs, reduction=red, twin_rows=True)
        assert got.shape == want.shape and (got == want).all(), red
    bad = lens.copy()
    k = int(np.flatnonzero(bad[0::2] > 0)[0])
    bad[2 * k + 1] -= 1
    with pytest.raises(alga_amd.AlgaError) as ei:
        eng.prefsuf_host(np.ascontiguousarray(words[1::2]), bad, lo, rs, twin_rows=True)
    assert ei.value.code == -1


def test_invalid_arguments_are_errors(eng):
    words, lens = _nodes(50, 60, 500, 27)
    with pytest.raises(alga_amd.AlgaError):
        eng.prefsuf_host(words, lens, 0, 10)
    with pytest.raises(alga_amd.AlgaError):
        eng.prefsuf_host(words[:, :2].copy(), lens, 30, 40)   # stride too small for the reads


def test_device_resident_and_sharded_paths_agree(eng):
    import torch
    from alga_amd.engine import device_edges_to_numpy, device_view
    words, lens = _nodes(4000, 150, 9000, 31, err=0.005)
    want, _, _ = O.prefsuf(words, lens, 82, 116)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    ptr, m = eng.prefsuf_device(dw, dl, 82, 116)
    got = device_edges_to_numpy(ptr, m)
    assert (got == want).all()
    # sharded: sources in 3 ranges, targets in 2 ranges, as the multi-GPU driver does it
    n = len(lens)
    recs = []
    for a, b in ((0, n // 3), (n // 3, n // 2), (n // 2, n)):
        d, v, k = eng.discover_device(dw, dl, 82, 116, a, b)
        if k:
            recs.append((device_view(d, (k,)).clone(), device_view(v, (k,), typestr="<i8").clone()))   # incl. padding
    rd = torch.cat([r[0] for r in recs]).contiguous()
    rv = torch.cat([r[1] for r in recs]).contiguous()
    parts = []
    for a, b in ((0, n // 2 + 7), (n // 2 + 7, n)):
        ptr, m = eng.reduce_device(dw, dl, 82, 116, rd, rv, rd.shape[0], a, b)
        parts.append(device_edges_to_numpy(ptr, m))
    allp = np.concatenate(parts)
    order = np.lexsort((allp[:, 2], allp[:, 1], allp[:, 0]))
    assert (allp[order] == want).all()


def test_cli_front_end_writes_the_reference_dump(golden_dir, tmp_path):
    """alga_hip (C++ host over the C ABI): FASTA in, `<TEST_NAME>_beforeSimplifier.graph` out, byte-identical to the
    dump the reference wrote for the same input."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(alga_amd.library_path()), "..", "bin", "alga_hip")
    for name in ("f1_cfg1", "f3_paired", "f5_messy"):
        fx = O.Fixture(golden_dir, name)
        try:
            f1, f2 = fx.inputs()
            # one GPU; and `--gpu-list=0,0,0`: the N-GPU path of the command line (alga_multi_*: three ranks, here on the one GPU
            # of the box with the copy transport), every rank running the input stage itself
            for extra in ([], ["--gpu-list=0,0,0"]):
                cmd = [exe, "--file1=" + f1, "--threads=4", "--output=o.fasta"] + (["--file2=" + f2] if f2 else []) + extra
                r = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                assert r.returncode == 0, r.stderr[-2000:]
                stem = os.path.basename(f1).rsplit(".", 1)[0]
                dump = os.path.join(str(tmp_path), "ALGA_%s_scale55_noN_beforeSimplifier.graph" % stem)
                got = open(dump, "rb").read()
                os.unlink(dump)
                assert got == fx.ref_graph(), extra
                assert ("Before first simplifier graph has %d edges" % fx.meta["edges_before_simplifier"]) in r.stderr
                assert not extra or "3 GPUs (peer copies)" in r.stderr
        finally:
            fx.cleanup()


def test_drop_in_produces_identical_contigs(golden_dir, tmp_path):
    """File-level drop-in: alga_hip builds the graph, stock ALGA consumes it with
    --deserialize_graph=1; the contigs must equal those of a plain ALGA run on the same input."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "oracle", "_ref", "ALGA")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/ALGA not built (needs /root/reference at build time)")
    exe = os.path.join(root, "alga_amd", "bin", "alga_hip")
    fx = O.Fixture(golden_dir, "f1_cfg1")
    try:
        f1, _ = fx.inputs()
        a, b = tmp_path / "plain", tmp_path / "dropin"
        a.mkdir(); b.mkdir()
        r = subprocess.run([ref, "--file1=" + f1, "--threads=1", "--output=o.fasta"], cwd=str(a), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
        assert r.returncode == 0
        r = subprocess.run([exe, "--file1=" + f1, "--threads=1", "--output=o.fasta", "--alga=" + ref], cwd=str(b),
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(str(a / "o.fasta")).read() == open(str(b / "o.fasta")).read()
        assert os.path.getsize(str(a / "o.fasta")) > 1000
    finally:
        fx.cleanup()
    # with sequencing errors and --error-rate=0.02 (both spellings): exact path + approximate supplement on the GPU
    fx = O.Fixture(golden_dir, "f2_err2")
    try:
        f1, _ = fx.inputs()
        a, b = tmp_path / "plain_err", tmp_path / "dropin_err"
        a.mkdir(); b.mkdir()
        r = subprocess.run([ref, "--file1=" + f1, "--threads=1", "--output=o.fasta", "--error_rate=0.02"], cwd=str(a),
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
        assert r.returncode == 0
        want_edges = int(re.search(r"After supplement G has (\d+) edges", r.stderr).group(1))
        r = subprocess.run([exe, "--file1=" + f1, "--threads=1", "--output=o.fasta", "--error-rate=0.02", "--alga=" + ref], cwd=str(b),
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("After supplement G has %d edges" % want_edges) in r.stderr
        assert open(str(a / "o.fasta")).read() == open(str(b / "o.fasta")).read()
    finally:
        fx.cleanup()


def test_exchange_helpers_emulated_ranks(eng):
    """The multi-GPU driver's compute steps (HipBackend: discover+order, reduce, edge ordering) with the collectives
    emulated in one process for 3 'ranks': the result must be the single-GPU graph, byte for byte."""
    import torch
    from alga_amd.multigpu import HipBackend, shard_bounds
    words, lens = _nodes(3000, 150, 7000, 41, err=0.004, stride=12)
    want, _, _ = O.prefsuf(words, lens, 82, 116)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    be = HipBackend(eng, dw, dl, 82, 116)
    with be.stream_scope():                                    # torch ops and engine calls on one stream, as in ShardedPrefSuf.step
        nr = 3
        b = shard_bounds(len(lens), nr)
        sent = []
        for r in range(nr):
            d, v = be.discover_sorted(b[r], b[r + 1])
            assert bool((d[1:] >= d[:-1]).all()) and int(d.min()) >= 0          # ordered by target, padding dropped
            cuts = torch.searchsorted(d, torch.tensor(b, dtype=d.dtype, device=d.device)).tolist()
            sent.append([(d[cuts[q]:cuts[q + 1]].clone(), v[cuts[q]:cuts[q + 1]].clone()) for q in range(nr)])
        parts = []
        for q in range(nr):
            rd = torch.cat([sent[r][q][0] for r in range(nr)]).contiguous()
            rv = torch.cat([sent[r][q][1] for r in range(nr)]).contiguous()
            parts.append(be.reduce(rd, rv, b[q], b[q + 1]).clone())
        allp = torch.cat(parts).contiguous()
        got = be.sort_edges(allp).cpu().numpy()
        assert got.shape == want.shape and (got == want).all()
        assert (be.build().cpu().numpy() == want).all()


@pytest.mark.parametrize("n,length,G,seed,err,minlen,lo,rs", [
    (4000, 150, 6000, 71, 0.0, None, 90, 120),     # 100x coverage: clusters of 60+ entries
    (3000, 144, 5000, 72, 0.004, 110, 82, 116),    # variable length + errors
    (2500, 100, 1200, 73, 0.0, None, 55, 77),      # 200x coverage on a short genome: hundreds of entries per bucket
])
def test_cluster_directory_geometries(n, length, G, seed, err, minlen, lo, rs):
    """the clustered probe with every shape of its bucket directory: few huge buckets (more than 255 entries: the directory's byte
    offsets saturate and a run reads the whole bucket), one cluster per bucket, more buckets than the key has bits for (clamped);
    with k_probe_stream first or the general kernel alone.  Always the same graph."""
    words, lens = _nodes(n, length, G, seed, err, minlen)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    e = alga_amd.Engine(0)
    try:
        e.set_option("probe", "cluster")
        for bias in (-8, -4, 0, 3, 8):
            for pairs, order in ((1, 1), (1, 0), (0, 1)):      # k_probe_stream first (sources in key / id order); the general kernel alone
                e.set_option("cluster_bucket_bias", bias)
                e.set_option("cluster_pairs", pairs)
                e.set_option("cluster_order", order)
                got = e.prefsuf_host(words, lens, lo, rs, reduction="source_side")
                st = e.last_stats()
                assert st["probe_used"] == 2, (bias, pairs, order)
                assert got.shape == want.shape and (got == want).all(), (bias, pairs, order)
    finally:
        e.close()


def test_shared_keys_protocol(eng):
    """alga_prefsuf_keys_device + keys_shared: two 'ranks' on one engine pair -- each computes the keys of its own nodes, the slices are
    exchanged by hand, each builds its source range: together the single-GPU graph.  Misuse (no key pass, a source range outside
    the keyed node range, another node set) is an error, not a wrong graph; inputs the clustered probe declines report so."""
    import torch
    from alga_amd.engine import device_view
    words, lens = _nodes(3000, 150, 8000, 83, err=0.002, stride=16)
    want, _, _ = O.prefsuf(words, lens, 90, 120)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    n, h = len(lens), (len(lens) // 4) * 2
    e2 = alga_amd.Engine(0)
    try:
        k1 = eng.keys_device(dw, dl, 90, 120, 0, h)
        k2 = e2.keys_device(dw, dl, 90, 120, h, n)
        assert k1 is not None and k2 is not None
        v1 = [device_view(p, (n,)) for p in k1]
        v2 = [device_view(p, (n,)) for p in k2]
        torch.cuda.synchronize()
        for a, b in zip(v1, v2):
            a[h:] = b[h:]
            b[:h] = a[:h]
        torch.cuda.synchronize()
        # rank 0 builds its range in two pieces (the second reuses the entry array: keys_shared = 2), rank 1 in one
        q = (h // 4) * 2
        r1a = eng.build_range_device(dw, dl, 90, 120, 0, q, keys_shared=1)
        e1a = device_view(r1a[0], (r1a[1], 3)).cpu().numpy()              # the next build reuses the edge buffer
        r1b = eng.build_range_device(dw, dl, 90, 120, q, h, keys_shared=2)
        e1b = device_view(r1b[0], (r1b[1], 3)).cpu().numpy()
        r2 = e2.build_range_device(dw, dl, 90, 120, h, n, keys_shared=True)
        got = np.concatenate([e1a, e1b, device_view(r2[0], (r2[1], 3)).cpu().numpy()])
        assert got.shape == want.shape and (got == want).all()
        with pytest.raises(alga_amd.AlgaError) as ei:                 # a piece outside the range the entry array has runs for
            eng.build_range_device(dw, dl, 90, 120, h, n, keys_shared=2)
        assert ei.value.code == -1
        fresh = alga_amd.Engine(0)
        try:
            with pytest.raises(alga_amd.AlgaError) as ei:             # no build came before
                fresh.build_range_device(dw, dl, 90, 120, 0, q, keys_shared=2)
            assert ei.value.code == -1
        finally:
            fresh.close()
        with pytest.raises(alga_amd.AlgaError) as ei:                 # the keys were consumed by the build
            eng.build_range_device(dw, dl, 90, 120, 0, h, keys_shared=True)
        assert ei.value.code == -1
        eng.keys_device(dw, dl, 90, 120, 0, h)
        with pytest.raises(alga_amd.AlgaError) as ei:                 # sources outside the keyed node range
            eng.build_range_device(dw, dl, 90, 120, 0, n, keys_shared=True)
        assert ei.value.code == -1
        # 250-nt reads take the clustered probe too (round 4: two-word offset masks, 80-byte entries): there are keys to share;
        # reads beyond 272 nt (rows of more than 17 words) do not
        # (at this size AUTO keeps the seed-table probe for them -- cache-resident, faster --: nothing to share; asked for, the clustered probe shares)
        w2, l2 = _nodes(600, 250, 4000, 84)
        d2, dl2 = torch.from_numpy(w2.view(np.int32)).cuda(), torch.from_numpy(l2).cuda()
        assert eng.keys_device(d2, dl2, 137, 190, 0, len(l2)) is None
        eng.set_option("probe", "cluster")
        try:
            assert eng.keys_device(d2, dl2, 137, 190, 0, len(l2)) is not None
            w3, l3 = _nodes(300, 280, 4000, 85)
            assert eng.keys_device(torch.from_numpy(w3.view(np.int32)).cuda(), torch.from_numpy(l3).cuda(), 160, 200, 0, len(l3)) is None
        finally:
            eng.set_option("probe", "auto")
    finally:
        e2.close()


@pytest.mark.parametrize("lo,rs,replicate", [(90, 120, False), (90, 120, True), (82, 116, False)])
def test_sharded_driver_with_hip_backend_three_ranks_one_gpu(lo, rs, replicate):
    """alga_amd.multigpu.ShardedPrefSuf exactly as bench.py --gpus N drives it (real HipBackend, device tensors), the
    collectives replaced by a thread rendezvous: 3 ranks, one engine each, on this one GPU.  (90, 120): source-side form,
    no record exchange; (82, 116): reads of 150 nt are too long for it, every rank declines and all take the exchange."""
    import torch
    from alga_amd.multigpu import HipBackend, ShardedPrefSuf
    from fake_dist import run_ranks
    words, lens = _nodes(3000, 150, 7000, 51, err=0.002, stride=16)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()

    def rank_main(rank, dist):
        e = alga_amd.Engine(0)
        try:
            out = []
            for kw in (dict(), dict(shard_keys=True, pieces=None)):        # the driver's defaults; sharded key pass + pieces (opt-in)
                run = ShardedPrefSuf(HipBackend(e, dw, dl, lo, rs), rank, 3, dist, replicate=replicate, **kw)
                m, st = run.step(collect_stats=True)
                m2, _ = run.step()
                assert m == m2 == len(want)
                out.append((run.edges_numpy(), st["raw_overlaps"]))
            return out
        finally:
            e.close()
    res = run_ranks(3, rank_main)
    _, _, cnt = O.prefsuf(words, lens, lo, rs)
    for r, outs in enumerate(res):
        for got, raw in outs:
            assert raw == cnt["hash_equal"]                          # whole-job counter, all-reduced
            if r == 0 or replicate:
                assert got.shape == want.shape and (got == want).all()
            else:
                assert len(got) == 0


@pytest.mark.parametrize("ranks,n,length,G,seed,err,minlen,lo,rs", [(3, 3000, 100, 6000, 52, 0.0, None, 55, 77), (2, 2500, 144, 6000, 53, 0.004, 110, 82, 116),
                                                                      (3, 1500, 100, 30000, 54, 0.0, None, 55, 77)])
def test_python_driver_bucket_sharded_form_thread_ranks_one_gpu(ranks, n, length, G, seed, err, minlen, lo, rs):
    """alga_amd.multigpu.ShardedPrefSuf(bucket_sharded=True) as bench.py --gpus N --multi-form bucket_sharded drives it (real HipBackend,
    device tensors; the collectives a thread rendezvous): the index sharded by seed bucket, five exchanges, == the oracle; the last case
    at 5x coverage, where the per-source cap decides most edges (pending edges, small-key exchange)."""
    import torch
    from alga_amd.multigpu import HipBackend, ShardedPrefSuf
    from fake_dist import run_ranks
    words, lens = _nodes(n, length, G, seed, err, minlen, stride=16)      # (the second case: variable lengths + errors, the meta array travels too)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()

    def rank_main(rank, dist):
        e = alga_amd.Engine(0)
        try:
            run = ShardedPrefSuf(HipBackend(e, dw, dl, lo, rs), rank, ranks, dist, bucket_sharded=True)
            m, st = run.step()
            m2, _ = run.step()
            torch.cuda.synchronize()
            return m, m2, run.form_used, run.edges_numpy(), dict(run.exchange_bytes)
        finally:
            e.close()

    res = run_ranks(ranks, rank_main)
    for r, (m, m2, form, edges, xb) in enumerate(res):
        assert m == len(want) and m2 == len(want) and form == "bucket_sharded"
        assert set(xb) == {"descriptors", "pending", "small_keys", "edges"}
        if r == 0:
            assert edges.shape == want.shape and (edges == want).all()


def test_sharded_driver_three_ranks_one_gpu_at_a_payload_that_can_race():
    """The same driver at 1 M reads (1.7 M nodes): key arrays of 7 MB, edge pieces of several MB per rank, four pieces per rank with
    their transfers outstanding while the next piece is probed -- the size at which the stream-ordering bug of round 2's rehearsal
    showed (wrong edge counts at 2 and 4 ranks; HipBackend.stream_scope is the fix this test guards).  Reference = the one-GPU
    graph of the same engine (itself byte-equal to the reference binary at this size: tests/test_gpu_fullsize.py)."""
    import torch
    from alga_amd import workload
    from alga_amd.multigpu import HipBackend, ShardedPrefSuf
    from fake_dist import run_ranks
    wl = workload.build("cfg2_1M_150bp", stride_words="aligned")
    lo, rs = wl["min_overlap"], wl["rsoemo"]
    dw = torch.from_numpy(wl["words"].view(np.int32)).cuda()
    dl = torch.from_numpy(wl["lens"]).cuda()
    e0 = alga_amd.Engine(0)
    try:
        ptr, m = e0.prefsuf_device(dw, dl, lo, rs)
        want = alga_amd.engine.device_view(ptr, (m, 3), dw.device).cpu().numpy().astype(np.int32).copy()
    finally:
        e0.close()
    assert len(want) > 1_000_000

    def rank_main(rank, dist):
        e = alga_amd.Engine(0)
        try:
            got = []
            for kw in (dict(shard_keys=True, pieces=4), dict()):
                run = ShardedPrefSuf(HipBackend(e, dw, dl, lo, rs), rank, 3, dist, **kw)
                for _ in range(2):
                    m_r, _ = run.step()
                    assert m_r == len(want)
                got.append(run.edges_numpy())
            return got
        finally:
            e.close()
    res = run_ranks(3, rank_main)
    for got in res[0]:
        assert got.shape == want.shape and (got == want).all()


def test_validated_runner_three_ranks_one_gpu():
    """What bench.py --gpus N times (alga_amd.multigpu.validated_runner): the source range in pieces -- up to four ranks with all keys computed
    on every rank (the pieces then go through the pile path), from five on with the sharded key pass -- after their graph equalled the plain
    form's, on the real HipBackend at 1 M reads, three and five ranks on this GPU."""
    import torch
    from alga_amd import workload
    from alga_amd.multigpu import HipBackend, validated_runner
    from fake_dist import run_ranks
    wl = workload.build("cfg2_1M_150bp", stride_words="aligned")
    lo, rs = wl["min_overlap"], wl["rsoemo"]
    dw = torch.from_numpy(wl["words"].view(np.int32)).cuda()
    dl = torch.from_numpy(wl["lens"]).cuda()
    e0 = alga_amd.Engine(0)
    try:
        ptr, m = e0.prefsuf_device(dw, dl, lo, rs)
        want = alga_amd.engine.device_view(ptr, (m, 3), dw.device).cpu().numpy().astype(np.int32).copy()
    finally:
        e0.close()

    for world in (3, 5):
        def rank_main(rank, dist):
            e = alga_amd.Engine(0)
            try:
                run, form = validated_runner(HipBackend(e, dw, dl, lo, rs), rank, world, dist)
                if world <= 4:                            # every rank computes all keys: its build goes through the pile path for its id range, in pieces
                    assert form["form"].startswith("all keys on every rank") and run.pieces == 4 and not run.shard_keys, form
                else:
                    assert form["form"].startswith("keys of own nodes") and run.pieces == 3 and run.shard_keys, form
                m_r, _ = run.step()
                assert m_r == len(want)
                return run.edges_numpy(), e.last_stats()
            finally:
                e.close()
        res = run_ranks(world, rank_main)
        assert res[0][0].shape == want.shape and (res[0][0] == want).all()
        # (the last piece's statistics: the piles served it at three ranks, the pairwise kernels at five)
        assert all((st["pile_buckets"] > 0) == (world <= 4) for _, st in res), [st["pile_buckets"] for _, st in res]


def test_bench_run_as_two_thread_ranks_equals_one_gpu():
    """bench.py's measurement body (bench.run: what `bench.py --gpus N` executes per rank after the process group is up) as two
    thread-ranks on this GPU, the collectives replaced by the thread rendezvous: the N > 1 branch of the script -- runner validation,
    counted pass, timing, max over ranks, the JSON line -- runs in the suite, and its graph has the one-GPU edge count."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from fake_dist import run_ranks
    common = ["--steps", "2", "--warmup", "1", "--config", "cfg3_5M_150bp", "--no-cpu-baseline", "--no-pcie", "--no-first-call"]
    one = bench.run(bench.parse_args(["--gpus", "1"] + common), 0, 1, 0, None)
    args2 = bench.parse_args(["--gpus", "2"] + common)
    res = run_ranks(2, lambda rank, dist: bench.run(args2, rank, 2, 0, dist))
    assert res[1] is None
    two = res[0]
    assert two["n_gpus"] == 2 and two["config"]["edges"] == one["config"]["edges"] > 5_000_000 and two["config"]["nodes"] == one["config"]["nodes"]
    assert two["multi_gpu_form"]["form"].startswith("all keys on every rank") and ", 4 pieces" in two["multi_gpu_form"]["form"] and "byte-identical" in two["multi_gpu_form"]["validated"]   # (two ranks: no key all-gather; the pieces through the pile path)
    assert two["value"] > 0 and two["roofline"]["frac"] > 0 and two["scaling"] == "strong"
    plain = run_ranks(2, lambda rank, dist: bench.run(bench.parse_args(["--gpus", "2", "--multi-plain"] + common), rank, 2, 0, dist))[0]
    assert plain["config"]["edges"] == one["config"]["edges"] and plain["multi_gpu_form"]["form"].startswith("plain")


def test_contig_like_inputs_second_call_of_the_reference(eng):
    """src/main.cpp:633-656 calls the same creator once more on the CONTIGS (kb-long "reads", min_overlap = rsoemo = 25, overlap
    lengths capped at 501): not on the benchmark path, but the engine takes it (per-target form: the reads are far too long for
    the source-side one)."""
    rng = np.random.default_rng(61)
    g = rng.integers(0, 4, 120000, dtype=np.uint8)
    seqs, p = [], 0
    while p < len(g) - 9000:                                   # contigs of 3-8 kb whose ends share 30-450 nt with the next one
        L = int(rng.integers(3000, 8000))
        seqs.append(g[p: p + L].copy())
        p += L - int(rng.integers(30, 450))
    maxlen = max(len(s) for s in seqs)
    codes = np.zeros((2 * len(seqs), maxlen), dtype=np.uint8)
    lens = np.zeros(2 * len(seqs), dtype=np.int32)
    for i, s in enumerate(seqs):
        codes[2 * i, : len(s)] = (3 - s)[::-1]
        codes[2 * i + 1, : len(s)] = s
        lens[2 * i] = lens[2 * i + 1] = len(s)
    words = alga_amd.pack_reads(codes, lens)
    got = _check(eng, words, lens, 25, 25)
    assert len(got) >= len(seqs) - 1
