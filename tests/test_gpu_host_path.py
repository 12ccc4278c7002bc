"""The host entry points end to end (SURVEY.md section 8(d)'s headline metric: packed host reads in, host edges out): what round 5 changed on
that path must not change a graph --
  * the graph in COMPACT form (alga_prefsuf_build_host_compact / alga_download_edges_compact: a degree byte per node, 5 bytes per edge) against
    the edge triples, and ALGA_ERR_UNSUPPORTED where an offset does not fit a byte;
  * lengths that cross PCIe as one or two bytes per node (node sets of 2^20 nodes and more) and are widened on the device;
  * node arrays in pinned host memory (alga_host_alloc), which go up without the staging copy;
  * the fused parallel length / twin check in front of the upload (a bad pair must still be refused).
Reference semantics of the graph itself: src/GraphCreators/GraphCreatorPrefSuf.cpp:73-488 (tests/test_gpu_parity.py holds those)."""
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


def _nodes(n, length, G, seed, err=0.0):
    codes, _ = gen_reads.sample_reads(n, length, G, seed, err)
    words, lens, _ = workload.make_nodes(codes)
    return words, lens


@pytest.mark.parametrize("length,err", [(150, 0.0), (100, 0.0), (150, 0.02)])
def test_compact_form_equals_the_triples_and_the_oracle(eng, length, err):
    words, lens = _nodes(12_000, length, 50_000, 40 + length, err)
    lo, rs = alga_amd.derive_params(float(length - 6))
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    got = eng.prefsuf_host(words, lens, lo, rs)
    assert got.shape == want.shape and (got == want).all()
    comp, _ = eng.prefsuf_host_compact(words, lens, lo, rs)
    assert comp.shape == want.shape and (comp == want).all()
    # twin rows (the odd nodes' rows alone) in compact form
    comp, _ = eng.prefsuf_host_compact(np.ascontiguousarray(words[1::2]), lens, lo, rs, twin_rows=True)
    assert comp.shape == want.shape and (comp == want).all()
    st = eng.last_stats()
    assert st["host_ms_upload"] > 0 and st["host_ms_build"] > 0 and st["host_ms_download"] > 0


def test_compact_form_is_refused_where_an_offset_does_not_fit_a_byte(eng):
    """reads of 400 nt with a minimal overlap of 60: offsets up to 340"""
    codes, _ = gen_reads.sample_reads(6000, 400, 800_000, 5)                # 3x coverage: the next read starts hundreds of nucleotides on
    rc = (3 - codes)[:, ::-1]
    both = np.stack([rc, codes], axis=1).reshape(-1, 400)
    lens = np.full(len(both), 400, dtype=np.int32)
    words = alga_amd.pack_reads(both, lens)
    want = eng.prefsuf_host(words, lens, 60, 120)
    assert int(want[:, 2].max()) > 255
    with pytest.raises(alga_amd.AlgaError) as ei:
        eng.prefsuf_host_compact(words, lens, 60, 120)
    assert ei.value.code == alga_amd.engine.ERR_UNSUPPORTED
    again = eng.prefsuf_host(words, lens, 60, 120)                   # the engine is as usable as before
    assert again.shape == want.shape and (again == want).all()


@pytest.mark.parametrize("length,n_reads", [(100, 700_000), (300, 560_000)])
def test_narrow_lengths_on_the_wire(eng, length, n_reads):
    """2^20 nodes and more: the lengths travel as one byte per node (reads of up to 255 nt) or two (longer ones) and are widened on the device;
    some nodes removed (length 0) and -- the 300-nt set -- reads of several lengths.  Against the build of the same node set resident on the
    device (uploaded by torch), edge for edge, through the triples, the compact form, twin rows and pinned arrays."""
    import torch
    from alga_amd.engine import device_view
    G = n_reads * length // 40
    codes, _ = gen_reads.sample_reads(n_reads, length, G, 7 + length)
    rc = (3 - codes)[:, ::-1]
    both = np.stack([rc, codes], axis=1).reshape(-1, length)
    lens = np.full(len(both), length, dtype=np.int32)
    rng = np.random.default_rng(3)
    dead = rng.random(len(lens) // 2) < 0.01
    lens[0::2][dead] = 0; lens[1::2][dead] = 0                       # removed pairs
    if length == 300:
        short = rng.random(len(lens) // 2) < 0.3
        cut = rng.integers(200, 300, size=len(lens) // 2).astype(np.int32)
        lens[1::2] = np.where(short & (lens[1::2] > 0), cut, lens[1::2])
        lens[0::2] = np.where(lens[0::2] > 0, lens[1::2], 0)
        # (the even row must stay the reverse complement of the odd one over the common length)
        idx = np.nonzero(short & ~dead)[0]
        L = lens[1::2][idx]
        for kk, ll in zip(idx, L):
            both[2 * kk, :ll] = (3 - both[2 * kk + 1, :ll])[::-1]
    assert len(lens) >= (1 << 20)
    words = alga_amd.pack_reads(both, lens)
    lo, rs = (55, 77) if length == 100 else (140, 200)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    ptr, m = eng.prefsuf_device(dw, dl, lo, rs)
    want = device_view(ptr, (m, 3), dw.device).cpu().numpy()
    del dw, dl
    assert m > 100_000
    got = eng.prefsuf_host(words, lens, lo, rs)
    assert got.shape == want.shape and (got == want).all()
    comp, _ = eng.prefsuf_host_compact(np.ascontiguousarray(words[1::2]), lens, lo, rs, twin_rows=True)
    assert comp.shape == want.shape and (comp == want).all()
    # pinned node arrays: no staging copy (the pointer is recognised), same graph
    pw = eng.host_array(words[1::2].shape, np.uint32)
    pl = eng.host_array(lens.shape, np.int32)
    pw[...] = words[1::2]; pl[...] = lens
    comp, _ = eng.prefsuf_host_compact(pw, pl, lo, rs, twin_rows=True)
    assert comp.shape == want.shape and (comp == want).all()
    del pw, pl


def test_a_bad_twin_pair_is_still_refused(eng):
    """the length check of twin rows runs on eight threads now (one pass with the maximum): a pair of different lengths anywhere must be refused"""
    n = (1 << 22) + 10
    lens = np.full(n, 100, dtype=np.int32)
    words = np.zeros((n // 2, 7), dtype=np.uint32)
    for bad in (4, n // 2 + 1, n - 2):
        l2 = lens.copy()
        l2[bad & ~1] = 99
        with pytest.raises(alga_amd.AlgaError):
            eng.prefsuf_host_compact(words, l2, 55, 77, twin_rows=True)
    with pytest.raises(alga_amd.AlgaError):                           # a read the rows cannot hold
        l2 = lens.copy(); l2[n - 1] = 500; l2[n - 2] = 500
        eng.prefsuf_host_compact(words, l2, 55, 77, twin_rows=True)
