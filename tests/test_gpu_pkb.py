"""GPU parity of the approximate supplement through the C ABI.

canAlign and the LI k-mers are pure functions: bit-exact against vectors made by the reference's own code.
The supplement as a whole is order dependent in the reference: the engine must equal the oracle run
with the engine's order-independent semantics (ORACLE_PKB_TIES_BY_ID | ORACLE_PKB_SNAPSHOT) bit for bit, and on the
golden fixture that also equals the reference's own post-supplement graph byte for byte."""
import gzip
import json
import os

import numpy as np
import pytest

import alga_amd
import gen_reads
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = alga_amd.Engine(0)
    yield e
    e.close()


def _f7(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "f7_pkb.json")))
    words, lens = O.load_nodes_bin(os.path.join(golden_dir, "f7_pkb.nodes.bin.gz"))
    return meta, words, lens


def test_can_align_batch_bit_exact(eng, golden_dir):
    meta, words, lens = _f7(golden_dir)
    z = np.load(os.path.join(golden_dir, "f7_pkb.canalign.npz"))
    p = eng.pkb_params(meta["avg_len"], 0.02, meta["kmer_length_bucket"])
    assert (p.min_overlap_area, p.max_offset_pct, p.min_identity_pct) == (meta["min_overlap_area"], meta["max_offset_pct"], meta["min_identity_pct"])
    got = eng.can_align_batch(words, lens, z["triples"], p)
    assert (got == z["verdict"]).all()
    # other thresholds on the same triples, and unequal read lengths (reads cut at random): against the oracle
    rng = np.random.default_rng(5)
    l2 = lens.copy()
    cut = rng.random(len(l2)) < 0.5
    l2[cut] = np.maximum(60, l2[cut] - rng.integers(0, 60, int(cut.sum()))) * (lens[cut] > 0)
    w2 = words.copy()
    for i in np.flatnonzero(cut):                       # keep the tail bits zero
        nb = 2 * int(l2[i])
        w2[i, (nb + 31) // 32:] = 0
        if nb % 32:
            w2[i, nb // 32] &= (1 << (nb % 32)) - 1
    tri = z["triples"]
    for (ww, ll) in ((words, lens), (w2, l2)):
        for moa, mo, mi in ((60, 40, 90), (100, 32, 97), (30, 70, 80), (111, 32, 99)):
            p2 = eng.pkb_params(144.0, 0.02, 54)
            p2.min_overlap_area, p2.max_offset_pct, p2.min_identity_pct = moa, mo, mi
            op = O.pkb_params(144.0)
            op.min_overlap_area, op.max_offset_pct, op.min_identity_pct = moa, mo, mi
            want = O.can_align(ww, ll, tri, op)
            got = eng.can_align_batch(ww, ll, tri, p2)
            assert want.sum() > 20 and (got == want).all()


def test_li_kmers_bit_exact(eng, golden_dir):
    meta, words, lens = _f7(golden_dir)
    p = eng.pkb_params(meta["avg_len"], 0.02, meta["kmer_length_bucket"])
    n = meta["likmer_nodes"]
    with gzip.open(os.path.join(golden_dir, "f7_pkb.likmers.bin.gz"), "rb") as f:
        buf = f.read()
    pos = 0
    prio = [0, 1, 2, 3]
    for rot in range(4):
        h, ind, cnt = eng.li_kmers(words[:n], lens[:n], p, prio)
        for i in range(n):
            if lens[i] < meta["li_k"]:
                assert cnt[i] == 0
                continue
            c = int(np.frombuffer(buf[pos:pos + 4], dtype=np.int32)[0]); pos += 4
            rec = np.frombuffer(buf[pos:pos + 12 * c], dtype=np.dtype([("h", "<u8"), ("i", "<i4")])); pos += 12 * c
            assert cnt[i] == c
            assert h[i, :c].tolist() == rec["h"].tolist() and ind[i, :c].tolist() == rec["i"].tolist()
        prio = prio[1:] + prio[:1]
    assert pos == len(buf)
    # other k / interval counts / lengths against the oracle
    codes, l2 = gen_reads.sample_reads(300, 120, 900, 56, min_length=40)
    w2 = alga_amd.pack_reads(codes, l2)
    for k, iv in ((35, 6), (21, 3), (60, 4), (40, 1)):
        p.li_k, p.li_intervals = k, iv
        h, ind, cnt = eng.li_kmers(w2, l2, p, [2, 0, 3, 1])
        for i in range(len(l2)):
            wh, wi = O.li_kmers(w2[i], l2[i], k, iv, [2, 0, 3, 1])
            assert cnt[i] == len(wh) and h[i, :cnt[i]].tolist() == wh.tolist() and ind[i, :cnt[i]].tolist() == wi.tolist()


def test_supplement_equals_reference_on_golden_fixture(eng, golden_dir):
    meta, words, lens = _f7(golden_dir)
    with gzip.open(os.path.join(golden_dir, meta["pre_graph"]), "rb") as f:
        n, pre = O.parse_graph(f.read())
    p = eng.pkb_params(meta["avg_len"], 0.02, meta["kmer_length_bucket"])
    got = eng.pkb_supplement_host(words, lens, pre, p)
    st = eng.pkb_last_stats()
    assert len(got) == meta["edges_after"] == st["edges_after"][3]
    with gzip.open(os.path.join(golden_dir, "f7_pkb.supplement.graph.gz"), "rb") as f:
        assert O.graph_bytes(n, got) == f.read()


@pytest.mark.parametrize("n,G,seed,err", [(4000, 9000, 61, 0.02), (3000, 3000, 62, 0.03), (2500, 12000, 63, 0.015)])
def test_supplement_equals_oracle_with_engine_semantics(eng, n, G, seed, err):
    codes, lens = gen_reads.sample_reads(n, 150, G, seed, err)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    pre = eng.prefsuf_host(words, lens, 82, 116)
    op = O.pkb_params(144.0, error_rate_percent=2)
    p = eng.pkb_params(144.0, 0.02, 54)
    want, _ = O.supplement(words, lens, pre, op, 54, flags=3)
    ref_order, _ = O.supplement(words, lens, pre, op, 54, flags=0)
    got = eng.pkb_supplement_host(words, lens, pre, p)
    assert len(want) > len(pre)
    assert got.shape == want.shape and (got == want).all()
    # how far the order-independent semantics is from the reference's sequential order on this input (reported, bounded)
    a, b = set(map(tuple, ref_order.tolist())), set(map(tuple, want.tolist()))
    assert len(a ^ b) <= 0.005 * len(a)


def _supplement_vs_oracle(eng, codes, stride_words=None):
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, codes.shape[1])[:, 3:-3]
    L = codes.shape[1]
    lens = np.full(len(codes), L, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    if stride_words:
        words = np.concatenate([words, np.zeros((len(words), stride_words - words.shape[1]), np.uint32)], axis=1)
    lo, rs = alga_amd.derive_params(float(L))
    pre = eng.prefsuf_host(words, lens, lo, rs)
    kb = min(2 * lo // 3, 60)
    op = O.pkb_params(float(L), error_rate_percent=2)
    p = eng.pkb_params(float(L), 0.02, kb)
    want, _ = O.supplement(words, lens, pre, op, kb, flags=3)
    got = eng.pkb_supplement_host(words, lens, pre, p)
    assert len(want) > len(pre)
    assert got.shape == want.shape and (got == want).all()
    return eng.pkb_last_stats()


def test_supplement_group_shapes(eng):
    """the three group kernels: deep coverage (groups of 8-64 entries: one wave each; above 64: the serial kernel), a tandem-repeat
    genome (a read twice in a group: handed to the serial kernel) and rows too long to stage in LDS (everything serial)"""
    rng = np.random.default_rng(91)

    def reads(genome, n, L, err):
        st = rng.integers(0, len(genome) - L + 1, n)
        c = genome[st[:, None] + np.arange(L)[None, :]].copy()
        m = rng.random(c.shape) < err
        c[m] = (c[m] + rng.integers(1, 4, int(m.sum()))) & 3
        flip = rng.random(n) < 0.5
        c[flip] = (3 - c[flip])[:, ::-1]
        return c.astype(np.uint8)
    g = rng.integers(0, 4, 1500, dtype=np.uint8)
    st = _supplement_vs_oracle(eng, reads(g, 4000, 150, 0.02))                       # ~400x coverage
    assert st["max_group"] > 64
    unit = rng.integers(0, 4, 43, dtype=np.uint8)
    g2 = np.concatenate([rng.integers(0, 4, 400, dtype=np.uint8), np.tile(unit, 12), rng.integers(0, 4, 400, dtype=np.uint8), np.tile(unit[::-1].copy(), 9),
                         rng.integers(0, 4, 300, dtype=np.uint8)])
    _supplement_vs_oracle(eng, reads(g2, 1500, 150, 0.02))
    g3 = rng.integers(0, 4, 9000, dtype=np.uint8)
    _supplement_vs_oracle(eng, reads(g3, 2500, 150, 0.02), stride_words=20)          # device rows of 32 words: not staged
    _supplement_vs_oracle(eng, reads(g3, 1200, 300, 0.02))                            # 294-nt reads: 19 words


@pytest.mark.parametrize("G,n,seed", [(4000, 1200, 93), (2500, 1500, 94), (6000, 5000, 95)])
def test_supplement_groups_of_8_to_16_four_per_wave(eng, G, n, seed):
    """k-mer groups of 8 .. 16 entries go four to a wave (k_pkb_groups_quarter, round 5): against the oracle in the engine's semantics, and against
    round 4's form of the same join (a wave per group, option pkb_legacy bit 0) down to the number of canAlign calls"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    st = rng.integers(0, G - 150 + 1, n)
    c = g[st[:, None] + np.arange(150)[None, :]].copy()
    m = rng.random(c.shape) < 0.02
    c[m] = (c[m] + rng.integers(1, 4, int(m.sum()))) & 3
    flip = rng.random(n) < 0.5
    c[flip] = (3 - c[flip])[:, ::-1]
    s_new = _supplement_vs_oracle(eng, c.astype(np.uint8))
    assert s_new["group_hist"][0][4] > 20, s_new["group_hist"]               # there are groups of 8 .. 15
    for legacy in (1, 6, 248, 256, 512):                                                  # bit 0: a wave per group of 8 .. 16; bits 1, 2: the library's k-mer sort on 32 bits, the head list in three kernels; bits 3, 4, 5: replay inside the pair kernel, the library's unique after the merge, the 128-bit k-mer walk
        # bit 8: a device-to-host copy per count the host waits for; bit 9: no look-ahead (the next round's sort beside this round's joins)
        eng.set_option("pkb_legacy", legacy)
        try:
            s_old = _supplement_vs_oracle(eng, c.astype(np.uint8))
        finally:
            eng.set_option("pkb_legacy", 0)
        for k in ("kmers", "groups", "can_align_calls", "edges_after", "group_hist", "max_group"):
            assert s_new[k] == s_old[k], (legacy, k)


def test_supplement_reads_of_several_lengths(eng):
    """reads of 110 .. 150 nt in one set: the k-mer walk of a wave runs to its longest read with every lane stopping at its own end, the interval
    borders differ from read to read (Read::getLIKmers, src/DataStructures/Read.cpp:145-226); against the oracle, and the 96-bit walk against the 128-bit one"""
    rng = np.random.default_rng(97)
    G, n = 5000, 2500
    g = rng.integers(0, 4, G, dtype=np.uint8)
    L = rng.integers(110, 151, n)
    st = rng.integers(0, G - 150 + 1, n)
    rows, lens = [], []
    for k in range(n):
        r = g[st[k]:st[k] + L[k]].copy()
        m = rng.random(len(r)) < 0.02
        r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) & 3
        for x in ((3 - r)[::-1], r):                                          # the pair (reverse complement, read): nodes 2 k, 2 k + 1
            row = np.zeros(150, np.uint8); row[:len(x)] = x
            rows.append(row); lens.append(len(x))
    codes, lens = np.stack(rows), np.array(lens, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    lo, rs = alga_amd.derive_params(float(lens.mean()))
    pre = eng.prefsuf_host(words, lens, lo, rs)
    kb = min(2 * lo // 3, 60)
    op = O.pkb_params(float(lens.mean()), error_rate_percent=2)
    p = eng.pkb_params(float(lens.mean()), 0.02, kb)
    want, _ = O.supplement(words, lens, pre, op, kb, flags=3)
    got = eng.pkb_supplement_host(words, lens, pre, p)
    s_new = eng.pkb_last_stats()
    assert len(want) > len(pre)
    assert got.shape == want.shape and (got == want).all()
    for legacy in (32, 128, 512):                                                   # the 128-bit walk per round; the 96-bit walk per round (default: all rounds in one walk)
        eng.set_option("pkb_legacy", legacy)
        try:
            old = eng.pkb_supplement_host(words, lens, pre, p)
            s_old = eng.pkb_last_stats()
        finally:
            eng.set_option("pkb_legacy", 0)
        assert (old == got).all() and s_old["kmers"] == s_new["kmers"] and s_old["groups"] == s_new["groups"] and s_old["group_hist"] == s_new["group_hist"], legacy


@pytest.mark.parametrize("rounds", [1, 2, 3])
def test_supplement_look_ahead_with_fewer_rounds_and_a_sequence_given_up(eng, rounds):
    """The look-ahead (engine_pkb.hip: pkb_presort -- the next round's k-mer entries sorted and their groups listed on the engine's side stream while
    this round's joins run) with 1 .. 3 rounds (the k-mer walk of all rounds is then the kernel for one or three), against the run without it (option
    pkb_legacy bit 9) and the oracle; then a sharded sequence that is given up after its first round -- its look-ahead is in flight -- followed by a
    full one on the same engine: the abandoned work must be waited out, not read."""
    import torch
    from alga_amd.engine import device_view
    codes, lens = gen_reads.sample_reads(5000, 150, 12000, 300 + rounds, 0.02)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    pre = eng.prefsuf_host(words, lens, 82, 116)
    p = eng.pkb_params(144.0, 0.02, 54)
    p.rounds = rounds
    got = eng.pkb_supplement_host(words, lens, pre, p)
    s_new = eng.pkb_last_stats()
    eng.set_option("pkb_legacy", 512)
    try:
        old = eng.pkb_supplement_host(words, lens, pre, p)
        s_old = eng.pkb_last_stats()
    finally:
        eng.set_option("pkb_legacy", 0)
    assert got.shape == old.shape and (got == old).all() and len(got) > len(pre)
    for k in ("kmers", "groups", "can_align_calls", "edges_after", "group_hist"):
        assert s_new[k] == s_old[k], k
    op = O.pkb_params(144.0, error_rate_percent=2)
    op.rounds = rounds
    want, _ = O.supplement(words, lens, pre, op, 54, flags=3)
    assert got.shape == want.shape and (got == want).all()
    # a sequence given up after its first round, then a whole one
    stride = 16
    wide = np.zeros((len(lens), stride), dtype=np.uint32)
    wide[:, :words.shape[1]] = words
    dw = torch.from_numpy(wide.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    d_pre = torch.from_numpy(np.ascontiguousarray(pre, dtype=np.int32)).cuda()
    torch.cuda.synchronize()
    p4 = eng.pkb_params(144.0, 0.02, 54)
    eng.pkb_shard_begin(dw, dl, d_pre.data_ptr(), len(pre), p4, 0, 1)
    eng.pkb_shard_round()                                                      # (rounds 1 .. 3 never come)
    ptr, m = eng.pkb_supplement_device(dw, dl, d_pre.data_ptr(), len(pre), p4)
    whole = device_view(ptr, (m, 3), dw.device).cpu().numpy()
    ref = eng.pkb_supplement_host(words, lens, pre, p4)
    assert whole.shape == ref.shape and (whole == ref).all()


def test_supplement_goes_on_serially_when_the_look_ahead_buffers_do_not_fit(eng):
    """The look-ahead needs a second set of sorted k-mer entries and sort scratch: when their allocation fails (option test_presort_oom stands in for a
    real out-of-memory there) the sequence gives them back and runs its rounds one after the other -- same graph, no error left behind."""
    codes, lens = gen_reads.sample_reads(4000, 150, 9000, 77, 0.02)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    pre = eng.prefsuf_host(words, lens, 82, 116)
    p = eng.pkb_params(144.0, 0.02, 54)
    want = eng.pkb_supplement_host(words, lens, pre, p)
    s_want = eng.pkb_last_stats()
    eng.set_option("test_presort_oom", 1)
    try:
        got = eng.pkb_supplement_host(words, lens, pre, p)
        s_got = eng.pkb_last_stats()
    finally:
        eng.set_option("test_presort_oom", 0)
    assert got.shape == want.shape and (got == want).all() and len(got) > len(pre)
    assert s_got["groups"] == s_want["groups"] and s_got["can_align_calls"] == s_want["can_align_calls"]
    again = eng.pkb_supplement_host(words, lens, pre, p)                      # (and with the look-ahead again afterwards)
    assert (again == want).all()


def test_supplement_rejects_offsets_it_cannot_represent(eng):
    """the edge merge packs (src, dst, offset) into 64 bits with 9 bits of offset: an edge outside that range is an error, not a
    silently different graph"""
    words = np.zeros((4, 40), np.uint32)
    lens = np.array([600, 600, 600, 600], np.int32)
    p = eng.pkb_params(600.0, 0.02, 60)
    with pytest.raises(alga_amd.AlgaError) as ei:
        eng.pkb_supplement_host(words, lens, np.array([[0, 2, 530]], np.int32), p)
    assert ei.value.code in (-1, -5)        # the host entry point validates its edge list, the device one counts misfits in the key kernel
    out = eng.pkb_supplement_host(words, lens, np.array([[0, 2, 300]], np.int32), p)       # in range: accepted
    assert len(out) >= 1


def test_engine_destroy_releases_every_device_buffer():
    """an engine that has run the clustered build, the supplement, the simplifier step and the sharded protocol gives all of its
    device memory back when it is destroyed (every buffer goes through one allocator that remembers it)"""
    import torch
    codes, lens = gen_reads.sample_reads(3000, 150, 6000, 77, 0.02)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens, 16)
    dw, dl = torch.from_numpy(words.view(np.int32)).cuda(), torch.from_numpy(lens).cuda()

    def use_one():
        e = alga_amd.Engine(0)
        try:
            pre = e.prefsuf_host(words, lens, 82, 116)
            e.pkb_supplement_host(words, lens, pre, e.pkb_params(144.0, 0.02, 54))
            e.cut_triangles_host(len(lens), pre, 250)
            e.keys_device(dw, dl, 82, 116, 0, len(lens))
            e.build_range_device(dw, dl, 82, 116, 0, len(lens), keys_shared=1)
        finally:
            e.close()
    use_one()                                               # runtime-side pools and code objects settle in the first round
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        use_one()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), (free0, free1)


@pytest.mark.parametrize("ranks,n,G,seed,err", [(2, 4000, 9000, 71, 0.02), (3, 3000, 3000, 72, 0.03), (5, 6000, 20000, 73, 0.02)])
def test_supplement_on_n_ranks_equals_one_gpu(eng, ranks, n, G, seed, err):
    """The supplement with its k-mer groups dealt out over N ranks by hash (alga_pkb_shard_*; the reference spreads its
    k-mer buckets over worker threads, src/GraphCreators/GraphCreatorKmerBased.cpp:108-136): alga_amd.multigpu.ShardedSupplement as bench.py
    --gpus N drives it -- real engines, one per rank, on this one GPU, the collectives a thread rendezvous.  Every rank must end with the graph
    the one-GPU supplement gives, which equals the oracle in the engine's semantics."""
    import torch
    from alga_amd.engine import device_view
    from alga_amd.multigpu import HipBackend, ShardedSupplement
    from fake_dist import run_ranks
    codes, lens = gen_reads.sample_reads(n, 150, G, seed, err)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    pre = eng.prefsuf_host(words, lens, 82, 116)
    p = eng.pkb_params(144.0, 0.02, 54)
    want = eng.pkb_supplement_host(words, lens, pre, p)
    op = O.pkb_params(144.0, error_rate_percent=2)
    orc, _ = O.supplement(words, lens, pre, op, 54, flags=3)
    assert want.shape == orc.shape and (want == orc).all() and len(want) > len(pre)
    stride = 16
    wide = np.zeros((len(lens), stride), dtype=np.uint32)
    wide[:, :words.shape[1]] = words
    dw = torch.from_numpy(wide.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    d_pre = torch.from_numpy(np.ascontiguousarray(pre, dtype=np.int32)).cuda()
    torch.cuda.synchronize()

    def rank_main(rank, dist):
        e = alga_amd.Engine(0)
        try:
            be = HipBackend(e, dw, dl, 82, 116)
            be.pkb = p
            sup = ShardedSupplement(be, rank, ranks, dist)
            with be.stream_scope():
                out = sup.run(d_pre if rank == 0 else torch.empty((0, 3), dtype=torch.int32, device=dw.device))
                got = out.clone()
            torch.cuda.synchronize()
            st = e.pkb_last_stats()
            return got.cpu().numpy(), st["groups"], sup.exchange_bytes
        finally:
            e.close()
    res = run_ranks(ranks, rank_main)
    tot_groups = [sum(r[1][k] for r in res) for k in range(4)]
    for r, (got, groups, xb) in enumerate(res):
        assert got.shape == want.shape and (got == want).all(), r
        assert len(xb) == 4
    # the groups were dealt out: every rank joined some of them, together all of them (the one-GPU count)
    eng.pkb_supplement_host(words, lens, pre, p)
    assert tot_groups == list(eng.pkb_last_stats()["groups"])
    assert all(sum(r[1]) > 0 for r in res)
