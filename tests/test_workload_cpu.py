"""Host logic: the numpy workload builder used by bench.py hands the engine exactly the node set the
reference's input stages would (checked against the oracle's literal ingest of the same reads as FASTA)."""
import numpy as np
import pytest

import oracle_lib as O
from alga_amd import workload
import gen_reads


@pytest.mark.parametrize("n,L,G,seed,err", [(4000, 100, 3000, 5, 0.0), (3000, 150, 2500, 6, 0.01), (2000, 150, 400, 7, 0.0)])
def test_make_nodes_matches_oracle_ingest(tmp_path, n, L, G, seed, err):
    codes, _ = gen_reads.sample_reads(n, L, G, seed, err)
    # add exact duplicates, cross-strand duplicates, a palindromic read and STR reads
    codes[10] = codes[3]
    codes[11] = (3 - codes[4])[::-1]
    half = codes[5, : L // 2 - 3]
    mid = np.concatenate([half, (3 - half)[::-1]])
    codes[12, 3: 3 + len(mid)] = mid
    if 3 + len(mid) == L - 3:
        pass
    codes[13] = np.tile(codes[13, :7], L // 7 + 1)[:L]
    path = str(tmp_path / "reads.fasta")
    workload.write_fasta_fast(path, codes)
    nd = O.ingest(path)
    words, lens, ids = workload.make_nodes(codes)
    assert len(lens) == nd["n"]
    assert (lens == nd["len"]).all()
    assert (words == nd["words"]).all()
    lo, rs = workload.derive_params(float(L - 6))
    assert (lo, rs) == (nd["min_overlap"], nd["rsoemo"])


def test_write_fasta_fast_format(tmp_path):
    codes = np.array([[0, 1, 2, 3], [3, 3, 0, 0]], dtype=np.uint8)
    p = str(tmp_path / "x.fasta")
    workload.write_fasta_fast(p, codes)
    assert open(p).read() == ">r000000000\nACGT\n>r000000001\nTTAA\n"


@pytest.mark.parametrize("err,seed", [(0.0, 13), (0.02, 13), (0.02, 5)])
def test_device_generator_is_the_reference_node_set(tmp_path, err, seed):
    """alga_amd.workload.device_build (bench.py's generator; torch ops, here on the CPU device): the node set it packs must be, as a
    multiset of rows, what the reference's input stages make of the same reads -- the sample it hands to the CPU baseline IS the
    read set when the window covers the genome.  With errors two reads are duplicates when they start at the same position and
    carry the same errors in the part that survives the end trimming."""
    torch = pytest.importorskip("torch")
    wl = workload.device_build(20000, 150, 60000, seed, device="cpu", err=err, sample_reads=20000, chunk=4096)
    codes = wl["sample_codes"]
    assert codes.shape == (wl["unique_reads"], 150)
    path = str(tmp_path / "reads.fasta")
    workload.write_fasta_fast(path, codes)
    nd = O.ingest(path)
    mine = np.ascontiguousarray(wl["words"].numpy().view(np.uint32))
    assert mine.shape[0] == nd["n"] and (wl["lens"].numpy() == nd["len"][0]).all()
    W = nd["words"].shape[1]
    assert not mine[:, W:].any()
    key = lambda a: np.sort(np.ascontiguousarray(a[:, :W]).view([("", np.uint32)] * W).ravel())      # noqa: E731
    assert (key(mine) == key(nd["words"])).all()
    assert (wl["min_overlap"], wl["rsoemo"]) == (nd["min_overlap"], nd["rsoemo"])
