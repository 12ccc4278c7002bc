"""First simplifier step on the GPU (alga_cut_triangles_*: Graph::sortEdgesByIncreasingOffset + GraphSimplifier::
cutNonAndWeaklyMetricTriangles) against the dumps the reference's own code produced and against the oracle: byte identity of the
resulting graph, in-list order included."""
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


@pytest.mark.parametrize("name", ["f1_cfg1", "f2_err2", "f4_varlen", "f5_messy", "f7_pkb"])
def test_cut_triangles_equals_reference_dump(eng, golden_dir, name):
    c = json.load(open(os.path.join(golden_dir, "n3_aftercut.json")))[name]
    with gzip.open(os.path.join(golden_dir, c["graph_in"]), "rb") as f:
        n, e = O.parse_graph(f.read())
    got = eng.cut_triangles_host(n, e, c["max_offset_parallel_paths"])
    assert len(got) == c["edges_after"]
    with gzip.open(os.path.join(golden_dir, name + ".aftercut.graph.gz"), "rb") as f:
        assert O.graph_bytes(n, got) == f.read()


@pytest.mark.parametrize("n,length,G,seed,err,lo,rs,mopp", [
    (6000, 150, 20000, 51, 0.01, 82, 116, 262),
    (5000, 100, 8000, 52, 0.02, 55, 77, 40),          # a small weight cap: long edges stay
    (4000, 120, 4000, 53, 0.0, 40, 60, 250),          # high coverage, short minimum overlap: long lists
])
def test_cut_triangles_equals_oracle_on_built_graphs(eng, n, length, G, seed, err, lo, rs, mopp):
    codes, lens = gen_reads.sample_reads(n, length, G, seed, err)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, length)
    lens = np.repeat(lens, 2).astype(np.int32)
    words = alga_amd.pack_reads(codes, lens)
    edges = eng.prefsuf_host(words, lens, lo, rs)
    want = O.cut_triangles(len(lens), edges, mopp)
    got = eng.cut_triangles_host(len(lens), edges, mopp)
    assert len(want) < len(edges)
    assert got.shape == want.shape and (got == want).all()


def test_device_form_and_degenerate_graphs(eng):
    import torch
    assert eng.cut_triangles_host(5, np.zeros((0, 3), np.int32), 250).shape == (0, 3)
    e = np.array([[0, 1, 10], [0, 2, 30], [1, 2, 20], [2, 0, 5]], np.int32)          # 0->2 (30) == 0->1->2 (10 + 20): cut
    got = eng.cut_triangles_host(3, e, 250)
    assert got.tolist() == [[0, 1, 10], [1, 2, 20], [2, 0, 5]]
    assert eng.cut_triangles_host(3, e, 25).tolist() == [[0, 1, 10], [0, 2, 30], [1, 2, 20], [2, 0, 5]]      # weight above the cap: stays
    d = torch.from_numpy(e).cuda()
    ptr, m, rem = eng.cut_triangles_device(3, d.data_ptr(), len(e), 250, stream=torch.cuda.current_stream().cuda_stream)
    assert (m, rem) == (3, 1)
    assert alga_amd.engine.device_edges_to_numpy(ptr, m).tolist() == [[0, 1, 10], [1, 2, 20], [2, 0, 5]]
    with pytest.raises(alga_amd.AlgaError):
        eng.cut_triangles_host(3, e[::-1], 250)                                        # not sorted by (src, dst)
