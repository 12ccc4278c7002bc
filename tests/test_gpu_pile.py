"""GPU parity of the probe through PILES (alga_amd/csrc/prefsuf_pile.hip; engine option "pile", default on).

The pile path replaces the pairwise verification of the clustered probe (src/GraphCreators/GraphCreatorPrefSuf.cpp:238-395 and the via
compare :434-451) by one compare of a source against the consensus of a minimizer's targets.  It may only ever change HOW a graph is
computed: every case here is built three ways -- pile path, pairwise kernels, CPU oracle -- and the three edge lists must be identical.
The cases are chosen to hit what the pile path hands on instead of deciding: duplicate reads (two members at one coordinate), repeats
(one k-mer, two loci: members that differ from the consensus), tandem repeats (one minimizer twice in a source), coverage gaps (two
items stand), reads with errors (the build is left to the pairwise kernels), and inputs it does not take at all (several lengths,
masks); and what the last step of round 4 added: no entry array for a build the pile path keeps (rows by id, any row stride, nothing stale read)."""
import numpy as np
import pytest

import alga_amd
import gen_reads
import oracle_lib as O
from alga_amd import workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = alga_amd.Engine(0)
    yield e
    e.close()


def _three_ways(eng, words, lens, lo, rs, af=None, at=None):
    want, _, _ = O.prefsuf(words, lens, lo, rs, af, at)
    out = {}
    # pile 2 (tests only): without the sample -- the pile kernels take the build however irregular its buckets are.  pile_runs 0: round 4's form (every
    # node its own run list, a pile's list joined from its outer members'); 1: the list from the pile's consensus, target keys only in the key pass.
    # pile_check: every node's own list as well, and every first-group member's compared with its pile's list clipped to its windows.
    # pile 3 (tests only, round 5): no sample either, the build in the MIXED form -- what k_pile_probe hands on goes through k_probe_stream (list mode:
    # the entry array is built for it), the general kernel gets what is left.
    for pile, pile_runs, check in ((1, 1, 0), (2, 1, 0), (3, 1, 0), (3, 0, 0), (1, 0, 0), (2, 1, 1), (0, 1, 0)):
        eng.set_option("pile", pile)
        eng.set_option("pile_runs", pile_runs)
        eng.set_option("pile_check", check)
        try:
            got = eng.prefsuf_host(words, lens, lo, rs, af, at, reduction="source_side")
        finally:
            eng.set_option("pile", 1)
            eng.set_option("pile_runs", 1)
            eng.set_option("pile_check", 0)
        st = eng.last_stats()
        assert got.shape == want.shape and (got == want).all(), ("pile", pile, "pile_runs", pile_runs, got.shape, want.shape)
        assert st["edges"] == len(want)
        if check:
            assert st["pile_list_mismatch"] == 0, st
            assert (st["pile_list_checked"] > 0) == (st["ms_pile"] > 0)
            continue
        if pile == 3:
            if st["ms_pile"] > 0:                                            # an input the pile path takes: the mixed form ran
                assert st["pile_mixed"] == 1 and st["pile_deferred"] >= st["deferred_sources"], st
            continue
        if pile_runs == 0:
            assert (st["ms_pile"] > 0) == (out[1]["ms_pile"] > 0)
            continue
        assert st["pile_mixed"] == 0 or pile == 1, st
        out[pile] = st
    assert out[0]["pile_buckets"] == 0 and out[0]["ms_pile"] == 0.0
    assert (out[2]["ms_pile"] > 0) == (out[1]["ms_pile"] > 0)       # forced or not, the same inputs are the pile path's
    return out[1], len(want)


def _nodes(n, length, G, seed, err=0.0, genome=None):
    if genome is None:
        codes, _ = gen_reads.sample_reads(n, length, G, seed, err)
    else:
        rng = np.random.default_rng(seed)
        starts = rng.integers(0, len(genome) - length + 1, size=n)
        codes = np.stack([genome[s:s + length] for s in starts]).astype(np.uint8)
        flip = rng.random(n) < 0.5
        codes[flip] = (3 - codes[flip])[:, ::-1]
    words, lens, _ = workload.make_nodes(codes)
    return words, lens


@pytest.mark.parametrize("length,coverage", [(150, 30), (150, 8), (100, 40), (126, 25), (150, 120)])
def test_pile_path_equals_pairwise_and_oracle(eng, length, coverage):
    G = 60_000
    words, lens = _nodes(G * coverage // length, length, G, 5 + length + coverage)
    lo, rs = alga_amd.derive_params(float(length - 6))
    st, E = _three_ways(eng, words, lens, lo, rs)
    assert st["probe_used"] == 2 and st["pile_buckets"] > 0, st              # the pile path ran ...
    assert st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN <= st["pile_buckets"]                     # ... and kept the build
    live = int((lens > 0).sum())
    assert st["deferred_sources"] <= live // 10, (st["deferred_sources"], live)   # nearly every source finished there
    assert E > 0


@pytest.mark.parametrize("rs", [82, 116, 140, 144, 145])
def test_pile_path_all_reduction_gaps(eng, rs):
    """G = len - max(rsoemo, Lmin): from the whole window range (rsoemo = min_overlap) down to 0 (rsoemo past the read length: no overlap is
    big, every item stands and every source with more than two is handed on); the source-side form takes min_overlap <= rsoemo <= len + 1."""
    words, lens = _nodes(9000, 150, 40_000, 77)
    lo, _ = alga_amd.derive_params(144.0)
    assert lo == 82
    _three_ways(eng, words, lens, lo, rs)


def test_pile_path_with_duplicate_reads(eng):
    """The same read several times (no duplicate removal in front of the build): two members of a pile at one coordinate."""
    codes, _ = gen_reads.sample_reads(6000, 150, 30_000, 3)
    codes = np.concatenate([codes, codes[:1500], codes[100:400]])
    fw = alga_amd.pack_reads(np.ascontiguousarray(codes[:, 3:147]))
    rv = alga_amd.pack_reads(np.ascontiguousarray((3 - codes[:, 3:147])[:, ::-1]))
    words = np.empty((2 * len(codes), fw.shape[1]), dtype=np.uint32)
    words[0::2], words[1::2] = fw, rv
    lens = np.full(2 * len(codes), 144, dtype=np.int32)
    lo, rs = alga_amd.derive_params(144.0)
    st, _ = _three_ways(eng, words, lens, lo, rs)
    assert st["pile_buckets"] > 0


def test_pile_path_on_repeats_and_tandems(eng):
    """A genome made of a few units repeated with point differences, plus a short-period stretch: one minimizer at several loci
    (members that differ from the consensus), one minimizer twice inside a read."""
    rng = np.random.default_rng(11)
    unit = rng.integers(0, 4, size=700).astype(np.uint8)
    parts = []
    for k in range(12):
        u = unit.copy()
        pos = rng.integers(0, len(u), size=3)
        u[pos] = (u[pos] + 1 + rng.integers(0, 3, size=3)) % 4
        parts.append(u)
        parts.append(rng.integers(0, 4, size=300).astype(np.uint8))
    motif = rng.integers(0, 4, size=37).astype(np.uint8)
    parts.append(np.tile(motif, 30))
    genome = np.concatenate(parts)
    words, lens = _nodes(9000, 150, None, 19, genome=genome)
    lo, rs = alga_amd.derive_params(144.0)
    st, _ = _three_ways(eng, words, lens, lo, rs)
    assert st["pile_buckets"] > 0


def test_mixed_form_is_chosen_on_the_device_for_a_genome_with_repeats(eng):
    """A genome whose repeats (a twentieth of it: 1 kb units copied with a point difference or two) make 1 - 4 % of the buckets irregular: too many for
    the pile kernels + general kernel alone (more than 1 in ALGA_PILE_IRREGULAR_ONE_IN), too few to leave the build to the pairwise kernels (fewer
    than 1 in ALGA_PILE_DECLINE_ONE_IN).  The SAMPLE decides, on the device: the mixed form (alga_prefsuf_stats.pile_mixed) -- the sources
    k_pile_probe hands on go through k_probe_stream by list, the general kernel gets what is left.  Error-free reads, 30x; the list must be the one
    the pairwise kernels give (option pile = 0), which the rest of the suite holds to the oracle."""
    import torch
    from alga_amd.engine import device_view
    rng = np.random.default_rng(404)
    G = 1_500_000
    genome = rng.integers(0, 4, size=G).astype(np.uint8)
    for _ in range(75):                                                       # 75 x 1 kb copied elsewhere, two point differences each
        a, b = (int(x) for x in rng.integers(0, G - 1000, size=2))
        unit = genome[a:a + 1000].copy()
        pos = rng.integers(0, 1000, size=2)
        unit[pos] = (unit[pos] + 1 + rng.integers(0, 3, size=2)) % 4
        genome[b:b + 1000] = unit
    words, lens = _nodes(G * 30 // 150, 150, None, 405, genome=genome)
    lo, rs = alga_amd.derive_params(144.0)
    dw = torch.from_numpy(np.ascontiguousarray(words).view(np.int32)).cuda()
    dl = torch.from_numpy(lens.astype(np.int32)).cuda()
    n = len(lens)
    ptr, m = eng.build_range_device(dw, dl, lo, rs, 0, n)
    st = eng.last_stats()
    got = device_view(ptr, (m, 3), dw.device).clone()
    assert st["probe_used"] == 2 and st["pile_buckets"] > 0 and st["ms_pile"] > 0, st
    assert st["pile_irregular"] * alga_amd.engine.PILE_IRREGULAR_ONE_IN > st["pile_buckets"], st        # too irregular for the pure form ...
    assert st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN <= st["pile_buckets"], st         # ... not enough to decline
    assert st["pile_mixed"] == 1 and 0 < st["pile_deferred"] and st["deferred_sources"] < st["pile_deferred"], st
    eng.set_option("pile", 0)
    try:
        ptr, m2 = eng.build_range_device(dw, dl, lo, rs, 0, n)
        want = device_view(ptr, (m2, 3), dw.device).clone()
    finally:
        eng.set_option("pile", 1)
    assert m == m2 and torch.equal(got, want)


def test_reads_with_errors_leave_the_build_to_the_pairwise_kernels(eng):
    words, lens = _nodes(12_000, 150, 40_000, 23, err=0.02)
    lo, rs = alga_amd.derive_params(144.0)
    st, _ = _three_ways(eng, words, lens, lo, rs)
    assert st["pile_buckets"] > 0 and st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN > st["pile_buckets"], st      # sampled, found irregular, declined on the device


def test_inputs_the_pile_path_does_not_take(eng):
    """Several read lengths, or an alignFrom / alignTo mask: the reduction is no function of the offset set alone."""
    codes, lens_nt = gen_reads.sample_reads(5000, 144, 30_000, 31, 0.0, min_length=130)
    rc = np.zeros_like(codes)
    for i, l in enumerate(lens_nt):
        rc[i, :l] = (3 - codes[i, :l])[::-1]
    both = np.empty((2 * len(codes), codes.shape[1]), dtype=np.uint8)
    both[0::2], both[1::2] = codes, rc
    vlens = np.repeat(lens_nt, 2).astype(np.int32)
    vwords = alga_amd.pack_reads(both, vlens)
    lo, rs = alga_amd.derive_params(144.0)
    want, _, _ = O.prefsuf(vwords, vlens, lo, rs)
    got = eng.prefsuf_host(vwords, vlens, lo, rs)
    assert got.shape == want.shape and (got == want).all()
    assert eng.last_stats()["pile_buckets"] == 0
    words, lens = _nodes(5000, 150, 30_000, 37)
    lo, rs = alga_amd.derive_params(144.0)
    rng = np.random.default_rng(2)
    at = (rng.random(len(lens)) < 0.7).astype(np.uint8)
    want, _, _ = O.prefsuf(words, lens, lo, rs, None, at)
    got = eng.prefsuf_host(words, lens, lo, rs, None, at)
    assert got.shape == want.shape and (got == want).all()
    assert eng.last_stats()["pile_buckets"] == 0


def test_no_entry_array_is_left_over(eng):
    """A build the pile path keeps makes no entry array (its kernels and the general kernel read the rows by id): whatever an EARLIER build
    left in that buffer must not be read.  First a pairwise build of other reads (which fills the entry array), then reads of a length whose
    shape takes the generic instantiation of the general kernel (120 nt: no compile-time word count), through the pile path -- with and
    without the entry array -- against the oracle."""
    words, lens = _nodes(9000, 150, 40_000, 301)
    lo, rs = alga_amd.derive_params(144.0)
    eng.set_option("pile", 0)
    try:
        eng.prefsuf_host(words, lens, lo, rs)
    finally:
        eng.set_option("pile", 1)
    words, lens = _nodes(12_000, 126, 60_000, 302)
    lo, rs = alga_amd.derive_params(120.0)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    for skip in (1, 0):
        eng.set_option("pile_skip_gather", skip)
        try:
            got = eng.prefsuf_host(words, lens, lo, rs)
        finally:
            eng.set_option("pile_skip_gather", 1)
        st = eng.last_stats()
        assert got.shape == want.shape and (got == want).all(), ("pile_skip_gather", skip, got.shape, want.shape)
        assert st["pile_buckets"] > 0 and st["deferred_sources"] > 0          # the general kernel had sources to finish


@pytest.mark.parametrize("stride", [9, 10, 12, 13, 16])
def test_rows_by_id_with_any_row_stride(eng, stride):
    """The pile kernels and the general kernel behind them read the rows straight from the caller's node array: tight rows (9 words per 144-nt node, the
    Bitset's own layout), strides that are no multiple of four words (scalar loads) and padded ones (16-byte loads) must give the same graph."""
    import torch
    from alga_amd.engine import device_view
    words, lens = _nodes(9000, 150, 40_000, 401)
    lo, rs = alga_amd.derive_params(144.0)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    W = 9
    wide = np.zeros((len(lens), stride), dtype=np.uint32)
    wide[:, :W] = words[:, :W]
    wide[:, W:] = 0xDEADBEEF                                                 # what lies between the rows is never looked at
    dw = torch.from_numpy(wide.view(np.int32)).cuda()
    dl = torch.from_numpy(lens.astype(np.int32)).cuda()
    ptr, m = eng.prefsuf_device(dw, dl, lo, rs, reduction="source_side")
    st = eng.last_stats()
    got = device_view(ptr, (m, 3), dw.device).cpu().numpy()
    assert got.shape == want.shape and (got == want).all(), (stride, got.shape, want.shape)
    assert st["pile_buckets"] > 0 and st["ms_pile"] > 0


def test_bucket_table_is_valid_by_epoch(eng):
    """The bucket table of the pile path is never cleared between builds: a record counts only when it carries the epoch of the build at hand.
    One engine builds two different read sets of different sizes alternately -- every record the other set left behind is stale -- and far
    more often than the nine bits of the epoch hold, so that the table is cleared and the epoch starts over at least once."""
    sets = []
    for n, G, seed in ((1500, 8000, 101), (2600, 11_000, 102)):
        words, lens = _nodes(n, 150, G, seed)
        lo, rs = alga_amd.derive_params(144.0)
        want, _, _ = O.prefsuf(words, lens, lo, rs)
        sets.append((words, lens, lo, rs, want))
    for it in range(540):
        words, lens, lo, rs, want = sets[it & 1]
        got = eng.prefsuf_host(words, lens, lo, rs)
        assert got.shape == want.shape and (got == want).all(), it
        if it < 2:
            assert eng.last_stats()["pile_buckets"] > 0


def test_out_of_memory_in_the_pile_buffers_falls_back_to_the_pairwise_kernels(eng):
    """The pile path needs ~180 B per node of its own (bucket table, group records, side records).  An input that fits without them must not
    fail because of them: when their allocation answers out of memory the build gives them back and finishes on the
    pairwise kernels -- same graph, no error left behind -- and the next build, with memory again, takes the pile path as before."""
    words, lens = _nodes(9000, 150, 40_000, 501)
    lo, rs = alga_amd.derive_params(144.0)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    eng.set_option("test_pile_oom", 1)
    try:
        got = eng.prefsuf_host(words, lens, lo, rs)
    finally:
        eng.set_option("test_pile_oom", 0)
    st = eng.last_stats()
    assert got.shape == want.shape and (got == want).all()
    assert st["probe_used"] == 2 and st["pile_buckets"] == 0 and st["ms_pile"] == 0.0      # the pairwise kernels took it
    got = eng.prefsuf_host(words, lens, lo, rs)
    st = eng.last_stats()
    assert got.shape == want.shape and (got == want).all()
    assert st["pile_buckets"] > 0 and st["ms_pile"] > 0


def test_a_further_piece_after_a_build_the_pile_path_kept_or_declined(eng):
    """params.keys_shared = 2 (a further piece of the same node set reuses the index of the build before it): a build the pile path DECLINED --
    reads with errors: decided on the device -- made an entry array and serves the next piece as before the pile path existed;
    one it KEPT left none and serves the piece through its piles (round 5: k_pile_probe over the piece's id range; with option pile_range 0 such a
    piece is refused, as it was until then)."""
    import torch
    from alga_amd.engine import device_view
    lo, rs = alga_amd.derive_params(144.0)
    for err, kept in ((0.02, False), (0.0, True)):
        words, lens = _nodes(12_000, 150, 40_000, 23, err=err)
        want, _, _ = O.prefsuf(words, lens, lo, rs)
        dw = torch.from_numpy(words.view(np.int32)).cuda()
        dl = torch.from_numpy(lens.astype(np.int32)).cuda()
        n = len(lens)
        ptr, m = eng.build_range_device(dw, dl, lo, rs, 0, n)
        st = eng.last_stats()
        assert (st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN <= st["pile_buckets"]) == kept and st["pile_buckets"] > 0
        full = device_view(ptr, (m, 3), dw.device).cpu().numpy()
        assert full.shape == want.shape and (full == want).all()
        half = (n // 4) * 2
        if kept:
            eng.set_option("pile_range", 0)
            try:
                with pytest.raises(alga_amd.AlgaError):
                    eng.build_range_device(dw, dl, lo, rs, 0, half, keys_shared=2)
            finally:
                eng.set_option("pile_range", 1)
            ptr, m = eng.build_range_device(dw, dl, lo, rs, 0, n)            # (the refusal forgot nothing, but the piles are those of a build: again)
        for a, b in ((0, half), (half, n), (half - 6, half + 10)):
            ptr, m = eng.build_range_device(dw, dl, lo, rs, a, b, keys_shared=2)
            got = device_view(ptr, (m, 3), dw.device).cpu().numpy()
            sel = want[(want[:, 0] >= a) & (want[:, 0] < b)]
            assert got.shape == sel.shape and (got == sel).all(), (err, a, b)


@pytest.mark.parametrize("pile,what", [(1, "sampled"), (2, "forced, pure form"), (3, "forced, mixed form")])
def test_pile_path_for_a_rank_s_id_range(eng, pile, what):
    """The strong-scaling N-GPU build deals the SOURCES out by id range while every rank holds the index: the pile path takes such a build too
    (round 5: k_pile_side_range compacts the side records of the range's sources, k_pile_probe walks those).  Ranges as 2, 3 and 7 ranks would
    get them, odd borders, a range of one twin pair and the whole: each equals the oracle's edges of those sources, the pairwise kernels' (option
    pile 0) and the build before round 5 (option pile_range 0: pairwise for a range); on plain reads and on a genome with repeats (deferred
    sources: the general kernel by list, in the mixed form the stream kernel first)."""
    import torch
    from alga_amd.engine import device_view
    lo, rs = alga_amd.derive_params(144.0)
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, 3000, dtype=np.uint8)
    genome = np.concatenate([rng.integers(0, 4, 60_000, dtype=np.uint8), unit, rng.integers(0, 4, 20_000, dtype=np.uint8), unit, rng.integers(0, 4, 30_000, dtype=np.uint8)])
    for words, lens in (_nodes(20_000, 150, 100_000, 41), _nodes(24_000, 150, 0, 42, genome=genome)):
        want, _, _ = O.prefsuf(words, lens, lo, rs)
        dw = torch.from_numpy(words.view(np.int32)).cuda()
        dl = torch.from_numpy(lens.astype(np.int32)).cuda()
        n = len(lens)
        cuts = sorted({0, n} | {(n * k // w) for w in (2, 3, 7) for k in range(1, w)} | {n // 2 + 1, n // 3 - 1})
        ranges = list(zip(cuts[:-1], cuts[1:])) + [(0, n), (n // 2, n // 2 + 2), (1, n - 1)]
        eng.set_option("pile", pile)
        try:
            for a, b in ranges:
                ptr, m = eng.build_range_device(dw, dl, lo, rs, a, b)
                st = eng.last_stats()
                got = device_view(ptr, (m, 3), dw.device).cpu().numpy()
                sel = want[(want[:, 0] >= a) & (want[:, 0] < b)]
                assert got.shape == sel.shape and (got == sel).all(), (what, a, b)
                assert st["ms_pile"] > 0 and (pile != 1 or st["pile_buckets"] > 0), (what, a, b)       # the piles were built for the range's build (forced: no sample, no count)
                if pile == 3:
                    assert st["pile_mixed"] == 1, (a, b, st)
            eng.set_option("pile_range", 0)
            for a, b in ranges[:3]:
                ptr, m = eng.build_range_device(dw, dl, lo, rs, a, b)
                st = eng.last_stats()
                got = device_view(ptr, (m, 3), dw.device).cpu().numpy()
                sel = want[(want[:, 0] >= a) & (want[:, 0] < b)]
                assert got.shape == sel.shape and (got == sel).all() and st["pile_buckets"] == 0, (a, b)
        finally:
            eng.set_option("pile", 1)
            eng.set_option("pile_range", 1)


@pytest.mark.parametrize("length,coverage,G", [(150, 30, 400_000), (150, 8, 300_000), (100, 40, 200_000), (150, 120, 60_000), (132, 25, 200_000)])
def test_consensus_run_lists_equal_the_members_own(eng, length, coverage, G):
    """The parity harness of the consensus-derived run lists (k_pile_runs_consensus): with option pile_check every node gets its own run list
    (k_node_runs, as in round 4) AND the piles get theirs from the consensus; a kernel then clips the pile's list to the windows of each
    first-group member and compares it with the member's own, run for run (cluster key, k-mer position, window range).  Not one may differ:
    a pile list that differs from a member's own would make the probe look in the wrong buckets and lose overlaps silently."""
    words, lens = _nodes(G * coverage // length, length, G, 900 + length + coverage)
    lo, rs = alga_amd.derive_params(float(length - 6))
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    eng.set_option("pile_check", 1)
    try:
        got = eng.prefsuf_host(words, lens, lo, rs)
    finally:
        eng.set_option("pile_check", 0)
    st = eng.last_stats()
    assert got.shape == want.shape and (got == want).all()
    live = int((lens > 0).sum())
    assert st["pile_list_checked"] > live // 2, (st["pile_list_checked"], live)      # most nodes are members of a first group with a list
    assert st["pile_list_mismatch"] == 0, st
    # and the build as it is timed (target keys only in the key pass): same graph, nearly every source finished by the pile kernels
    got = eng.prefsuf_host(words, lens, lo, rs)
    st = eng.last_stats()
    assert got.shape == want.shape and (got == want).all()
    assert st["pile_buckets"] > 0 and st["deferred_sources"] <= live // 10


def test_declined_then_kept_then_declined(eng):
    """Where the run lists come from is laid out from the verdict of the build BEFORE (a declined build makes every node's list up front, a kept
    one only where a list is read; the first build of either kind makes them behind the sample): reads with errors, clean reads and reads with
    errors again through one engine -- every transition -- against the oracle."""
    lo, rs = alga_amd.derive_params(144.0)
    sets = []
    for err, seed in ((0.02, 61), (0.0, 62), (0.02, 63), (0.02, 64), (0.0, 65), (0.0, 66)):
        words, lens = _nodes(10_000, 150, 40_000, seed, err=err)
        want, _, _ = O.prefsuf(words, lens, lo, rs)
        sets.append((err, words, lens, want))
    for err, words, lens, want in sets:
        got = eng.prefsuf_host(words, lens, lo, rs)
        st = eng.last_stats()
        assert got.shape == want.shape and (got == want).all(), err
        kept = st["pile_irregular"] * alga_amd.engine.PILE_DECLINE_ONE_IN <= st["pile_buckets"]
        assert st["pile_buckets"] > 0 and kept == (err == 0.0)
