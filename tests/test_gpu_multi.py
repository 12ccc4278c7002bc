"""alga_multi_* (alga_amd/csrc/engine_multi.hip): the C++ N-GPU driver behind the C ABI -- one process, one host thread and one engine
per rank.  On a one-GPU box the ranks share the GPU and the COPY transport stands in for RCCL (the same driver code, the
collectives as peer copies + host barriers); RCCL itself is exercised as far as one GPU allows: loaded, a one-rank communicator
made, its in-place all-gather called.  The N-rank graph must be the one-GPU graph byte for byte."""
import numpy as np
import pytest

import alga_amd
import oracle_lib as O
from test_gpu_parity import _nodes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks", [2, 3, 5])
@pytest.mark.parametrize("n,length,G,seed,err,minlen,lo,rs", [
    (3000, 100, 6000, 81, 0.0, None, 55, 77),          # uniform length: only the key array is shared
    (3000, 144, 5000, 82, 0.004, 110, 82, 116),        # variable length + errors: the meta array travels too, some sources go the general way
])
def test_multi_copy_transport_equals_oracle(ranks, n, length, G, seed, err, minlen, lo, rs):
    words, lens = _nodes(n, length, G, seed, err, minlen)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    m = alga_amd.MultiEngine([0] * ranks, transport="copy")
    try:
        m.set_option("form", "replicated")               # round 3's form: every rank the whole index, its own source ids
        for _ in range(2):                               # warm buffers the second time
            got = m.prefsuf_host(words, lens, lo, rs)
            assert got.shape == want.shape and (got == want).all()
        st = m.last_stats()
        assert st["n_ranks"] == ranks and st["transport"] == 2 and st["fell_back_to_one_gpu"] == 0 and st["edges"] == len(want) and st["form"] == 1
        assert all(r["probe_used"] == 2 and r["reduction_used"] == 2 for r in st["ranks"])
        assert sum(r["edges"] for r in st["ranks"]) == len(want)
    finally:
        m.close()


@pytest.mark.parametrize("ranks", [2, 3, 5])
@pytest.mark.parametrize("n,length,G,seed,err,minlen,lo,rs", [
    (3000, 100, 6000, 81, 0.0, None, 55, 77),          # uniform length, error-free
    (3000, 144, 5000, 82, 0.004, 110, 82, 116),        # variable length + errors: irregular targets, pending small overlaps, the meta array travels
    (4000, 100, 2500, 86, 0.01, None, 55, 77),         # 160x coverage with errors: buckets of more than 64 descriptors (the chunked path of the join)
    (1500, 100, 30000, 87, 0.0, None, 55, 77),         # 5x coverage: most survivors are SMALL overlaps -- the per-source cap decides (pending edges, top-3 exchange)
])
def test_multi_bucket_sharded_form_equals_oracle(ranks, n, length, G, seed, err, minlen, lo, rs):
    """The index sharded by seed bucket (alga_shard_*: every rank 1 / N of the entry array, run descriptors to the bucket's owner, the
    reduction per target there, the per-source cap through the pending / small-key exchange, edges back to the source's owner): the
    same bytes as the oracle for 2, 3 and 5 ranks."""
    words, lens = _nodes(n, length, G, seed, err, minlen)
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    m = alga_amd.MultiEngine([0] * ranks, transport="copy")
    try:
        m.set_option("form", "bucket_sharded")
        for _ in range(2):
            got = m.prefsuf_host(words, lens, lo, rs)
            assert got.shape == want.shape and (got == want).all()
        st = m.last_stats()
        assert st["form"] == 2 and st["fell_back_to_one_gpu"] == 0 and st["edges"] == len(want)
        sh = [m.rank_shard_stats(r) for r in range(ranks)]
        assert sum(x["edges"] for x in sh) == len(want)
        assert sum(x["targets_owned"] for x in sh) > 0 and sum(x["descriptors_out"] for x in sh) == sum(x["descriptors_in"] for x in sh)
        assert sum(x["edges_out"] for x in sh) == len(want)
    finally:
        m.close()


def test_multi_bucket_sharded_masks_tandem_repeats_and_declines():
    import gen_reads
    from test_gpu_parity import _tandem_nodes
    rng = np.random.default_rng(88)
    m = alga_amd.MultiEngine([0, 0, 0], transport="copy")
    try:
        m.set_option("form", "bucket_sharded")
        # masks (alignFrom => alignTo) and removed nodes
        words, lens = _nodes(2500, 100, 5000, 88, 0.0, None)
        lens = lens.copy()
        lens[rng.random(len(lens)) < 0.05] = 0
        at = (rng.random(len(lens)) > 0.1).astype(np.uint8)
        af = (at & (rng.random(len(lens)) > 0.1)).astype(np.uint8)
        want, _, _ = O.prefsuf(words, lens, 55, 77, af, at)
        got = m.prefsuf_host(words, lens, 55, 77, af, at)
        assert got.shape == want.shape and (got == want).all() and m.last_stats()["form"] == 2
        # tandem repeats: the same target at several offsets of a source (two descriptors of one source in a bucket: supersede)
        words, lens = _tandem_nodes(89, 1500, 6000, 100, 7)
        want, _, _ = O.prefsuf(words, lens, 55, 77)
        got = m.prefsuf_host(words, lens, 55, 77)
        assert got.shape == want.shape and (got == want).all()
        # a bucket with more descriptors than ONE rank's join takes (its limit lowered to 48; 160x coverage): that rank declines the phase,
        # all ranks continue in the replicated form with the keys already gathered
        words, lens = _nodes(4000, 100, 2500, 86, 0.01, None)
        want, _, _ = O.prefsuf(words, lens, 55, 77)
        m.set_rank_option(1, "shard_bucket_max", 48)
        got = m.prefsuf_host(words, lens, 55, 77)
        assert got.shape == want.shape and (got == want).all()
        assert m.last_stats()["form"] == 1
        m.set_rank_option(1, "shard_bucket_max", 4096)
        got = m.prefsuf_host(words, lens, 55, 77)
        assert got.shape == want.shape and (got == want).all() and m.last_stats()["form"] == 2
        # 250-nt reads: the clustered probe takes them (two-word form), the bucket-sharded join (one-word offsets) does not: declined at its first phase
        words, lens = _nodes(1200, 250, 8000, 90, 0.0, None)
        want, _, _ = O.prefsuf(words, lens, 140, 190)
        got = m.prefsuf_host(words, lens, 140, 190)
        assert got.shape == want.shape and (got == want).all() and m.last_stats()["form"] == 1
        assert m.prefsuf_host(np.zeros((0, 8), np.uint32), np.zeros(0, np.int32), 55, 77).shape == (0, 3)
    finally:
        m.close()


def test_multi_falls_back_when_a_rank_declines():
    """250-nt reads at the default scale take the source-side form (two-word masks; since round 4 through the clustered probe, keys
    shared); reads beyond 288 nt do not -- every rank declines and rank 0 builds the whole graph the general way.  Same graph as one engine."""
    for length, lo, rs, fell in ((250, 140, 190, 0), (400, 60, 90, 1)):
        words, lens = _nodes(1500, length, 9000, 83, 0.0, None)
        want, _, _ = O.prefsuf(words, lens, lo, rs)
        m = alga_amd.MultiEngine([0, 0, 0], transport="copy")
        try:
            got = m.prefsuf_host(words, lens, lo, rs)
            assert got.shape == want.shape and (got == want).all()
            assert m.last_stats()["fell_back_to_one_gpu"] == fell
        finally:
            m.close()


@pytest.mark.parametrize("who", [0, 1, 2])
def test_multi_exactly_one_rank_declines(who):
    """The capacity case: ONE rank's build answers UNSUPPORTED (its per-wave item slice lowered; every source of this repeat-rich
    input has more raw overlaps than a wave's LDS holds), the other two succeed.  All three must agree on the fallback from one
    snapshot taken at the rendezvous (a rank re-reading a peer's flags after the barrier could see them already reset, take the
    gather branch and wait there for ever): rank 0 builds the whole graph, same edges as the oracle."""
    import gen_reads
    codes, _ = gen_reads.sample_reads(5, 80, 120, 26)
    w = alga_amd.pack_reads(np.tile(codes, (80, 1)))     # copies interleaved: every rank's id range holds every one of the five reads
    lens = np.full(len(w), 80, np.int32)
    want, _, _ = O.prefsuf(w, lens, 40, 60)
    m = alga_amd.MultiEngine([0, 0, 0], transport="copy")
    try:
        m.set_rank_option(who, "local_big_max", 200)
        for _ in range(2):
            got = m.prefsuf_host(w, lens, 40, 60)
            assert got.shape == want.shape and (got == want).all()
            assert m.last_stats()["fell_back_to_one_gpu"] == 1
    finally:
        m.close()


def test_multi_masks_and_empty_input():
    words, lens = _nodes(2000, 100, 4000, 84, 0.0, None)
    rng = np.random.default_rng(84)
    at = (rng.random(len(lens)) > 0.1).astype(np.uint8)
    af = (at & (rng.random(len(lens)) > 0.1)).astype(np.uint8)          # alignFrom => alignTo (the source-side form's precondition)
    want, _, _ = O.prefsuf(words, lens, 55, 77, af, at)
    m = alga_amd.MultiEngine([0, 0], transport="copy")
    try:
        got = m.prefsuf_host(words, lens, 55, 77, af, at)
        assert got.shape == want.shape and (got == want).all()
        empty = m.prefsuf_host(np.zeros((0, 8), np.uint32), np.zeros(0, np.int32), 55, 77)
        assert empty.shape == (0, 3)
    finally:
        m.close()


def test_multi_rccl_one_rank():
    """RCCL as far as one GPU goes: librccl loaded, ncclCommInitAll for one device, the in-place ncclAllGather of the key arrays."""
    words, lens = _nodes(3000, 100, 6000, 85, 0.0, None)
    want, _, _ = O.prefsuf(words, lens, 55, 77)
    try:
        m = alga_amd.MultiEngine([0], transport="rccl")
    except alga_amd.AlgaError as e:
        if e.code == -7:
            pytest.skip("librccl.so.1 is not loadable on this box")
        raise
    try:
        got = m.prefsuf_host(words, lens, 55, 77)
        assert got.shape == want.shape and (got == want).all()
        assert m.last_stats()["transport"] == 1
    finally:
        m.close()
    with pytest.raises(alga_amd.AlgaError):                # RCCL wants one GPU per rank
        alga_amd.MultiEngine([0, 0], transport="rccl")


def test_multi_device_entry_point_at_1m_reads():
    """the device-resident entry point with three ranks on one GPU at 1 M reads (1.7 M nodes): == the one-engine graph"""
    import torch
    from alga_amd import workload
    wl = workload.build("cfg2_1M_150bp", stride_words="aligned")
    lo, rs = wl["min_overlap"], wl["rsoemo"]
    dw = torch.from_numpy(wl["words"].view(np.int32)).cuda()
    dl = torch.from_numpy(wl["lens"]).cuda()
    e0 = alga_amd.Engine(0)
    try:
        ptr, k = e0.prefsuf_device(dw, dl, lo, rs)
        want = alga_amd.engine.device_view(ptr, (k, 3), dw.device).cpu().numpy().astype(np.int32).copy()
    finally:
        e0.close()
    m = alga_amd.MultiEngine([0, 0, 0], transport="copy")
    try:
        ptr, k = m.prefsuf_device([(dw, dl)] * 3, lo, rs)
        got = alga_amd.engine.device_view(ptr, (k, 3), dw.device).cpu().numpy().astype(np.int32)
        assert got.shape == want.shape and (got == want).all()
    finally:
        m.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_supplement_equals_one_gpu(ranks):
    """configs[4]'s scored path on N ranks through the C++ driver: exact graph (alga_multi_prefsuf_build_device), then the approximate supplement with
    its k-mer groups dealt out by hash (alga_multi_pkb_supplement_device: the exact graph to every rank, a round's additions all-gathered over the
    handle's transport) -- the post-supplement graph of one engine, edge for edge."""
    import torch
    import gen_reads
    from alga_amd.engine import device_view
    codes, _ = gen_reads.sample_reads(5000, 150, 12000, 91, 0.02)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 150)[:, 3:147]
    lens = np.full(len(codes), 144, dtype=np.int32)
    words = alga_amd.pack_reads(codes, lens)
    wide = np.zeros((len(lens), 16), dtype=np.uint32)
    wide[:, :words.shape[1]] = words
    e1 = alga_amd.Engine(0)
    try:
        pre = e1.prefsuf_host(words, lens, 82, 116)
        p = e1.pkb_params(144.0, 0.02, 54)
        want = e1.pkb_supplement_host(words, lens, pre, p)
    finally:
        e1.close()
    assert len(want) > len(pre)
    dw = torch.from_numpy(wide.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    torch.cuda.synchronize()
    m = alga_amd.MultiEngine([0] * ranks, transport="copy")
    try:
        for _ in range(2):
            ptr, k = m.prefsuf_device([(dw, dl)] * ranks, 82, 116)
            assert k == len(pre)
            p2, k2 = m.pkb_supplement_device([(dw, dl)] * ranks, ptr, k, p)
            got = device_view(p2, (k2, 3), dw.device).cpu().numpy()
            assert got.shape == want.shape and (got == want).all()
    finally:
        m.close()


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_multi_host_entry_uploads_a_slice_per_rank(ranks):
    """alga_multi_prefsuf_build_host brings the rows over PCIe ONCE: every rank its 1 / N of the caller's row array, the slices all-gathered
    between the GPUs, the engine's layout made on each rank (twin expansion / re-stride).  Odd row counts (the last slice is short), rows at
    the Bitset's tight stride (9 words: re-strided on the device), twin rows (the odd nodes' rows alone), masks -- against the oracle."""
    words, lens = _nodes(3001, 144, 6000, 95, 0.0, None)
    tight = np.ascontiguousarray(words[:, :9])
    want, _, _ = O.prefsuf(words, lens, 82, 116)
    m = alga_amd.MultiEngine([0] * ranks, transport="copy")
    try:
        for _ in range(2):
            got = m.prefsuf_host(tight, lens, 82, 116)
            assert got.shape == want.shape and (got == want).all()
        got = m.prefsuf_host(np.ascontiguousarray(tight[1::2]), lens, 82, 116, twin_rows=True)
        assert got.shape == want.shape and (got == want).all()
        rng = np.random.default_rng(4)
        at = (rng.random(len(lens)) < 0.8).astype(np.uint8)
        want_m, _, _ = O.prefsuf(words, lens, 82, 116, None, at)
        got = m.prefsuf_host(words, lens, 82, 116, None, at)
        assert got.shape == want_m.shape and (got == want_m).all()
        # a bad pair is refused on every rank, and the handle works afterwards
        l2 = lens.copy(); l2[10] = l2[10] - 1
        with pytest.raises(alga_amd.AlgaError):
            m.prefsuf_host(np.ascontiguousarray(tight[1::2]), l2, 82, 116, twin_rows=True)
        got = m.prefsuf_host(tight, lens, 82, 116)
        assert got.shape == want.shape and (got == want).all()
    finally:
        m.close()
