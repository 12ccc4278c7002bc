"""Host logic on CPU: the C++ input stages of the product (alga_amd/host/ingest.cpp, through the C ABI) against the
oracle's literal restatement of the reference's input stages, on every golden fixture, single- and multi-threaded."""
import numpy as np
import pytest

import alga_amd
import oracle_lib as O


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", O.FIXTURES)
def test_cpp_ingest_matches_oracle(golden_dir, name, threads):
    fx = O.Fixture(golden_dir, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        want = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
        got = alga_amd.ingest_files(f1, f2, threads=threads, min_overlap=lo, rsoemo=rs)
    finally:
        fx.cleanup()
    assert got["n"] == want["n"] == fx.meta["nodes"]
    assert (got["min_overlap"], got["rsoemo"]) == (fx.meta["min_overlap"], fx.meta["rsoemo"])
    assert got["removed_prefix"] == fx.meta["removed_prefix_reads"]
    assert got["removed_short"] == fx.meta["removed_short_reads"]
    assert (got["len"] == want["len"]).all()
    assert (got["pair_off"] == want["pair_off"]).all()
    W = want["W"]
    assert got["stride"] % 4 == 0 and got["stride"] >= W
    assert (got["words"][:, :W] == want["words"]).all()
    assert (got["words"][:, W:] == 0).all()


def test_ingest_errors_are_reported(tmp_path):
    p = tmp_path / "bad.fasta"
    p.write_text(">r0\nACGTXACGTACGTACGTACGTACGT\n")
    with pytest.raises(alga_amd.AlgaError):
        alga_amd.ingest_files(str(p))
    with pytest.raises(alga_amd.AlgaError):
        alga_amd.ingest_files(str(tmp_path / "missing.fasta"))
