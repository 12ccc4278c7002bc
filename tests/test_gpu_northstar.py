"""BASELINE.json's full sizes (configs[3]: 50 M x 150 bp; configs[4]: 10 M x 150 bp with 2 % errors) through properties that do not
need the reference at that size (the byte-for-byte comparisons against the real reference binary are tests/test_gpu_fullsize.py
at 1 M reads in this suite and the builder-run dumps under profiles/ at 10 M and 50 M):

  * two independent discovery algorithms -- the seed-table probe (one hash probe per suffix window) and the clustered minimizer
    join (sorted entry array + bucket directory) -- give the SAME edge list, byte for byte;
  * two independent reductions -- the source-side form and the per-target replay of the reference's insertion order -- give the same list;
  * the list is strictly (src, dst)-ordered, ids and offsets are in range, no self edges;
  * every sampled edge IS a suffix-prefix overlap of the packed reads (checked on the host against the 2-bit rows), of length
    >= min_overlap;
  * a second build of the same engine returns the same bytes (no dependence on what a previous build left in its buffers);
  * the counts equal the ones of the runs whose dumps were compared with the reference's (profiles/, cited below).
The node sets are generated on the device (alga_amd.workload.device_build, what bench.py measures on)."""
import numpy as np
import pytest

import alga_amd
from alga_amd import workload
from alga_amd.engine import device_view

pytestmark = pytest.mark.gpu


def _codes(words_row, n):
    """2-bit codes of one packed row (A0 C1 G2 T3, nucleotide i at bits 2i, 2i+1 of the LSB-first words)."""
    w = words_row.astype(np.uint64)
    i = np.arange(n)
    return ((w[(2 * i) >> 5] >> ((2 * i) & 31).astype(np.uint64)) & 3).astype(np.uint8)


def _check_list(edges, n_nodes, lens_max, lo):
    import torch
    e = edges.to(torch.int64)
    assert int(e[:, 0].min()) >= 0 and int(e[:, 0].max()) < n_nodes and int(e[:, 1].min()) >= 0 and int(e[:, 1].max()) < n_nodes
    assert not bool((e[:, 0] == e[:, 1]).any())
    assert int(e[:, 2].min()) >= 0 and int(e[:, 2].max()) <= lens_max - lo
    key = e[:, 0] * n_nodes + e[:, 1]
    assert bool((key[1:] > key[:-1]).all()), "not strictly (src, dst)-ordered"


def _check_overlaps(edges, d_words, d_lens, lo, sample, seed):
    import torch
    m = int(edges.shape[0])
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    pick = torch.randint(0, m, (sample,), generator=g).to(edges.device)
    es = edges[pick].cpu().numpy()
    rows_a = d_words[es[:, 0].astype(np.int64)].cpu().numpy().view(np.uint32)
    rows_b = d_words[es[:, 1].astype(np.int64)].cpu().numpy().view(np.uint32)
    la = d_lens[es[:, 0].astype(np.int64)].cpu().numpy()
    lb = d_lens[es[:, 1].astype(np.int64)].cpu().numpy()
    for k in range(sample):
        off = int(es[k, 2])
        ov = min(int(la[k]) - off, int(lb[k]))
        assert ov >= lo
        a, b = _codes(rows_a[k], int(la[k])), _codes(rows_b[k], int(lb[k]))
        assert (a[off:off + ov] == b[:ov]).all(), "edge %s is not an overlap" % (es[k],)


def _build(eng, wl, probe, reduction="auto", collect_stats=True, **opts):
    import torch
    eng.set_option("probe", probe)
    ptr, m = eng.prefsuf_device(wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"], reduction=reduction, collect_stats=collect_stats)
    torch.cuda.synchronize()
    return device_view(ptr, (m, 3), wl["words"].device).clone(), eng.last_stats()


def _pile_kept(st):
    """the pile path sampled the key order and KEPT the build (prefsuf_pile.hip: k_pile_sample_close)"""
    return st["pile_buckets"] > 0 and st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN <= st["pile_buckets"] and st["ms_pile"] > 0


@pytest.mark.parametrize("config,nodes,edges", [("cfg4_50M_150bp", 90_621_096, 92_350_115), ("x2_100M_150bp", 181_269_292, 184_711_310)])
def test_north_star_50M_reads_properties(config, nodes, edges):
    """BASELINE configs[3] on one GPU: 90 621 096 nodes -> 92 350 115 edges (profiles/r03_cfg4_50M_bench.json; the host-generated set
    of the same shape is byte-equal to the reference's dump: profiles/r02_d_cfg4_50M_dump_vs_reference.log) -- and the same shape at
    twice the reads (181 M nodes: the bucket directory at its 2^26 limit, two entries per bucket)."""
    import torch
    n_reads, read_len, G, seed, err = workload.CONFIGS[config]
    wl = workload.device_build(n_reads, read_len, G, seed, err=err)
    torch.cuda.synchronize()                       # the engine works on its own stream: its inputs must be complete
    n = int(wl["lens"].shape[0])
    lo = wl["min_overlap"]
    eng = alga_amd.Engine(0)
    try:
        cl, st = _build(eng, wl, "cluster")
        assert st["probe_used"] == 2 and st["reduction_used"] == 2
        assert n == nodes and int(cl.shape[0]) == edges == st["edges"]
        _check_list(cl, n, int(wl["lens"].max().item()), lo)
        _check_overlaps(cl, wl["words"], wl["lens"], lo, 4000, 1)
        cl2, st2 = _build(eng, wl, "cluster")
        assert torch.equal(cl, cl2) and st2["raw_overlaps"] == st["raw_overlaps"]
        del cl2
        tb, st3 = _build(eng, wl, "table")
        assert st3["probe_used"] == 1
        assert st3["raw_overlaps"] == st["raw_overlaps"]                     # both probes verified the same set of overlaps
        assert torch.equal(cl, tb), "seed-table probe and clustered join disagree"
        del tb
        # WHAT bench.py TIMES: a build without the work counters takes the probe through PILES (k_pile_build / k_pile_runs /
        # k_pile_probe; a build that collects the counters runs the pairwise kernels, which is what the three lists above came
        # from).  Same bytes, at the size where the 2^26-bucket directory, two k-mers per bucket, 5-bit tag collisions and the
        # ~0.5 % of sources handed to the general kernel all occur.  Semantics held: GraphCreatorPrefSuf.cpp:369-483.
        assert st["pile_buckets"] == 0 and st["ms_pile"] == 0.0 and st3["pile_buckets"] == 0     # (the counted builds did not pile)
        for rep in range(2):                                                              # twice: epoch-tagged bucket table reused
            pl, stp = _build(eng, wl, "auto", collect_stats=False)
            assert stp["probe_used"] == 2 and stp["reduction_used"] == 2
            assert _pile_kept(stp), stp
            assert 0 < stp["deferred_sources"] < n // 20                                   # the general kernel finished what the piles handed on (0.54 % at 50 M reads, 2.1 % at 100 M)
            assert int(pl.shape[0]) == edges
            assert torch.equal(cl, pl), "pile path and pairwise kernels disagree at %s (build %d)" % (config, rep)
            del pl
        # the parity harness of the consensus-derived run lists at this size: every first-group member's own list (k_node_runs) against its
        # pile's list clipped to the member's windows -- not one may differ (tests/test_gpu_pile.py: the same on the small inputs)
        eng.set_option("pile_check", 1)
        try:
            pl, stp = _build(eng, wl, "auto", collect_stats=False)
        finally:
            eng.set_option("pile_check", 0)
        assert _pile_kept(stp) and torch.equal(cl, pl)
        assert stp["pile_list_checked"] > n // 2 and stp["pile_list_mismatch"] == 0, stp
        del pl
        # a rank's share through the piles at this size (round 5): the sources as three ranks get them, each range a build of
        # its own, the second cut in two pieces (the later one reuses the piles: keys_shared 2) -- concatenated: the same list
        if config == "cfg4_50M_150bp":
            b = [0, (n // 3) | 1, 2 * (n // 3), n]                                          # (an odd border: ranges need not respect the twin pairs)
            mid = (b[1] + b[2]) // 2
            at = 0
            for a0, a1, ks in ((b[0], b[1], 0), (b[1], mid, 0), (mid, b[2], 2), (b[2], b[3], 0)):
                eng.set_option("probe", "auto")
                ptr, m = eng.build_range_device(wl["words"], wl["lens"], lo, wl["rsoemo"], a0, a1, keys_shared=ks)
                torch.cuda.synchronize()
                stp = eng.last_stats()
                part = device_view(ptr, (m, 3), wl["words"].device)
                assert stp["pile_buckets"] > 0 and stp["probe_used"] == 2, stp
                assert torch.equal(part, cl[at:at + m]), "the range [%d, %d) through the piles differs from the whole build" % (a0, a1)
                assert m == 0 or (int(part[0, 0]) >= a0 and int(part[-1, 0]) < a1)
                at += m
            assert at == edges
    finally:
        eng.close()


def test_pile_path_at_4x_takes_the_mixed_form_and_equals_pairwise():
    """Four times the north-star read set (200 M reads of a 1 Gb genome, 363 M nodes): a 19-mer sits at a second locus often enough
    that the sample finds more than 1 irregular bucket in 250 (1.5 %).  Until round 4 the pile kernels left such a build to the pairwise
    ones; since round 5 they keep it in the MIXED form -- what they hand on (one source in eight) goes through k_probe_stream by list and
    only the rest to the general kernel -- decided on the device.  The list must be the one a build with the pile path switched off gives,
    and a valid overlap list; the mixed build must be the faster one."""
    import torch
    n_reads, read_len, G, seed, err = workload.CONFIGS["x4_200M_150bp"]
    wl = workload.device_build(n_reads, read_len, G, seed, err=err)
    torch.cuda.synchronize()
    n = int(wl["lens"].shape[0])
    lo = wl["min_overlap"]
    eng = alga_amd.Engine(0)
    try:
        a, sta = _build(eng, wl, "auto", collect_stats=False)
        assert sta["probe_used"] == 2 and sta["pile_buckets"] > 0
        kept = _pile_kept(sta)
        _check_list(a, n, int(wl["lens"].max().item()), lo)
        _check_overlaps(a, wl["words"], wl["lens"], lo, 2000, 3)
        eng.set_option("pile", 0)
        b, stb = _build(eng, wl, "auto", collect_stats=False)
        eng.set_option("pile", 1)
        assert stb["pile_buckets"] == 0
        assert torch.equal(a, b), "4x set: build with the pile path %s differs from the pairwise build" % ("kept" if kept else "declined")
        assert kept and sta["pile_mixed"] == 1, sta
        assert sta["pile_irregular"] * alga_amd.engine.PILE_IRREGULAR_ONE_IN > sta["pile_buckets"]
        assert sta["deferred_sources"] < sta["pile_deferred"] // 2, sta        # the stream kernel finished most of what the pile kernel handed on
        del a, b
        _, sta2 = _build(eng, wl, "auto", collect_stats=False)                 # (the first build of an engine allocates inside its timed region: both forms once more, warm)
        eng.set_option("pile", 0)
        _, stb2 = _build(eng, wl, "auto", collect_stats=False)
        eng.set_option("pile", 1)
        print("4x set: mixed form %.1f ms (handed on %d, to the general kernel %d), pairwise %.1f ms" % (sta2["ms_total"], sta["pile_deferred"], sta["deferred_sources"], stb2["ms_total"]))
        assert sta2["ms_total"] < stb2["ms_total"], (sta2["ms_total"], stb2["ms_total"])
    finally:
        eng.close()


@pytest.mark.parametrize("n_reads", [2_250_000, 2_400_000])
def test_both_sides_of_the_partial_sort_threshold(n_reads):
    """The index build sorts only the key bits the directory needs (bucket and m_C >> 3) once there are 2^22 nodes or more, all 32 bits
    below that (rocPRIM's merge-sort path compares the wrong bits for a sort on [b, 32): alga_amd/csrc/sort_records.hip).  Node sets just
    under and just over the threshold, the clustered join against the seed-table probe."""
    import torch
    wl = workload.device_build(n_reads, 150, 5 * n_reads, 23)
    torch.cuda.synchronize()
    n = int(wl["lens"].shape[0])
    assert (n < (1 << 22)) == (n_reads == 2_250_000) and abs(n - (1 << 22)) < 300_000
    eng = alga_amd.Engine(0)
    try:
        cl, st = _build(eng, wl, "cluster")
        assert st["probe_used"] == 2
        _check_list(cl, n, int(wl["lens"].max().item()), wl["min_overlap"])
        tb, st2 = _build(eng, wl, "table")
        assert st2["probe_used"] == 1 and st2["raw_overlaps"] == st["raw_overlaps"]
        assert torch.equal(cl, tb)
    finally:
        eng.close()


def test_configs4_10M_reads_with_errors_properties():
    """BASELINE configs[4]: exact graph through both probes and both reductions, then the approximate supplement: 19 989 632 nodes,
    4 617 669 exact edges, 8 991 578 after the supplement (profiles/r03_cfg5_10M_bench.json; the exact graph of this shape equals the
    reference's --threads=1 dump, the supplement equals the oracle in the engine's semantics edge for edge:
    profiles/r02_d_cfg5_10M_exact_path_vs_reference_threads1.json, profiles/r02_cfg5_10M_err2_oracle_modes.json)."""
    import torch
    n_reads, read_len, G, seed, err = workload.CONFIGS["cfg5_10M_150bp_err2"]
    wl = workload.device_build(n_reads, read_len, G, seed, err=err)
    torch.cuda.synchronize()                       # the engine works on its own stream: its inputs must be complete
    n = int(wl["lens"].shape[0])
    lo = wl["min_overlap"]
    eng = alga_amd.Engine(0)
    try:
        cl, st = _build(eng, wl, "cluster")
        assert st["probe_used"] == 2
        assert n == 19_989_632 and int(cl.shape[0]) == 4_617_669
        _check_list(cl, n, int(wl["lens"].max().item()), lo)
        _check_overlaps(cl, wl["words"], wl["lens"], lo, 2000, 2)
        tb, st2 = _build(eng, wl, "table")
        assert st2["probe_used"] == 1 and torch.equal(cl, tb), "seed-table probe and clustered join disagree"
        del tb
        pt, st3 = _build(eng, wl, "auto", reduction="per_target")
        assert st3["reduction_used"] == 1 and torch.equal(cl, pt), "per-target replay and source-side form disagree"
        del pt
        # the supplement on the resident exact graph: same result twice, a superset of nothing it removes wrongly -- every exact
        # (src, dst) pair survives (addDirectedEdge only adds pairs or lowers offsets), list ordered, count as recorded
        eng.set_option("probe", "auto")
        mean_len = float(wl["lens"][wl["lens"] > 0].float().mean().item())
        pkb = alga_amd.Engine.pkb_params(mean_len, err, min(2 * lo // 3, 60))
        outs = []
        for _ in range(2):
            ptr, m = eng.prefsuf_device(wl["words"], wl["lens"], lo, wl["rsoemo"])
            p2, m2 = eng.pkb_supplement_device(wl["words"], wl["lens"], ptr, m, pkb)
            torch.cuda.synchronize()
            outs.append(device_view(p2, (m2, 3), wl["words"].device).clone())
        assert torch.equal(outs[0], outs[1])
        sup = outs[0]
        assert int(sup.shape[0]) == 8_991_578
        _check_list(sup, n, int(wl["lens"].max().item()), 0)
        ks = sup[:, 0].to(torch.int64) * n + sup[:, 1].to(torch.int64)
        ke = cl[:, 0].to(torch.int64) * n + cl[:, 1].to(torch.int64)
        pos = torch.searchsorted(ks, ke)
        assert bool((ks[pos.clamp(max=ks.shape[0] - 1)] == ke).all()), "an exact edge is missing after the supplement"
        assert bool((sup[pos, 2] <= cl[:, 2]).all())                       # offsets only ever go down (keep-min)
    finally:
        eng.close()
