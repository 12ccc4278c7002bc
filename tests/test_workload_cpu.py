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
