"""The in-process drop-in of INTEGRATION.md section 2, compiled and run: oracle/_ref/ref_adapter links the reference's own object
files (minus main.o), my driver of its creator call site (oracle/ref_adapter.cpp: the sequence of src/main.cpp:239-296 and
:300-347) and the reference-side adapters alga_amd/host/adapter/GraphCreatorPrefSufHIP.h / GraphCreatorLIHIP.h over the C ABI.

  cpu mode (no GPU): the driver with the REFERENCE's creators must reproduce the stock binary's dumps -- pins the driver's flow;
  hip mode (-m gpu): the same flow with the adapters must give the same bytes.
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_adapter")
needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/ref_adapter is not built (make -C oracle adapter needs /root/reference)")


def _write_nodes(path, words, lens):
    with open(path, "wb") as f:
        f.write(np.array([len(lens), words.shape[1]], dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(lens, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(words, dtype=np.uint32).tobytes())


def _fixture_nodes(golden_dir, name, tmp_path):
    fx = O.Fixture(golden_dir, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        nd = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
    finally:
        fx.cleanup()
    path = str(tmp_path / (name + ".nodes.bin"))
    _write_nodes(path, nd["words"], nd["len"])
    return fx, nd, path


def _run(mode, nodes, nd, out, extra=()):
    cmd = [EXE, mode, nodes, out, str(nd["min_overlap"]), str(nd["rsoemo"]), str(nd["li_kmer_length"])] + [str(x) for x in extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_makefile_builds_the_adapter_where_the_reference_is_present():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no /root/reference on this machine: the prebuilt binary is used")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "adapter"], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)
    syms = subprocess.run(["nm", "-C", EXE], capture_output=True, text=True).stdout
    assert "GraphCreatorPrefSufHIP::startAlignmentGraphCreation" in syms and "GraphCreatorLIHIP::startAlignmentGraphCreation" in syms
    # bound through the C ABI: one upload per process (alga_adapter::Session), the stages on the resident node set
    for sym in ("alga_upload_nodes", "alga_prefsuf_build_device", "alga_pkb_supplement_device", "alga_download_edges", "alga_engine_reserve"):
        assert sym in syms, sym


@needs_exe
@pytest.mark.parametrize("name", ["f1_cfg1", "f4_varlen"])
def test_driver_flow_with_the_reference_creator_reproduces_the_stock_dump(golden_dir, tmp_path, name):
    fx, nd, nodes = _fixture_nodes(golden_dir, name, tmp_path)
    out = str(tmp_path / "cpu.graph")
    _run("cpu", nodes, nd, out)
    assert open(out, "rb").read() == fx.ref_graph()


@needs_exe
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f1_cfg1", "f2_err2", "f3_paired", "f4_varlen", "f5_messy", "f6_l40"])
def test_adapter_in_the_reference_call_site_gives_the_reference_dump(golden_dir, tmp_path, name):
    fx, nd, nodes = _fixture_nodes(golden_dir, name, tmp_path)
    out = str(tmp_path / "hip.graph")
    _run("hip", nodes, nd, out)
    assert open(out, "rb").read() == fx.ref_graph()


@needs_exe
@pytest.mark.gpu
def test_supplement_adapter_gives_the_reference_post_supplement_graph(golden_dir, tmp_path):
    import json
    meta = json.load(open(os.path.join(golden_dir, "f7_pkb.json")))
    fx, nd, nodes = _fixture_nodes(golden_dir, "f2_err2", tmp_path)
    out, out2 = str(tmp_path / "hip.graph"), str(tmp_path / "hip2.graph")
    log = _run("hip", nodes, nd, out, (meta["error_rate_percent"], meta["kmer_length_bucket"], out2))
    assert open(out, "rb").read() == fx.ref_graph()
    assert "edges_after_supplement %d" % meta["edges_after"] in log
    with gzip.open(os.path.join(golden_dir, "f7_pkb.supplement.graph.gz"), "rb") as f:
        assert open(out2, "rb").read() == f.read()


@needs_exe
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f2_err2", "f4_varlen"])
def test_first_simplifier_step_through_the_adapter(golden_dir, tmp_path, name):
    import json
    c = json.load(open(os.path.join(golden_dir, "n3_aftercut.json")))[name]
    fx, nd, nodes = _fixture_nodes(golden_dir, name, tmp_path)
    out = str(tmp_path / "hipcut.graph")
    log = _run("hip", nodes, nd, out, (c["max_offset_parallel_paths"],))
    assert "edges_after_cut %d" % c["edges_after"] in log
    with gzip.open(os.path.join(golden_dir, name + ".aftercut.graph.gz"), "rb") as f:
        assert open(out, "rb").read() == f.read()


@needs_exe
@pytest.mark.gpu
def test_one_pair_that_is_no_reverse_complement_pair_travels_whole(golden_dir, tmp_path):
    """ALGA's reads come as reverse-complement pairs and the adapter uploads the odd rows only (alga_adapter::NodeArrays) -- after checking
    EVERY pair, not a sample: a vector whose lengths pair up but where ONE even read is some other sequence must give the
    graph the reference's creator gives for that very vector (cpu mode of the same driver), i.e. the even read as it is, not a rebuilt twin."""
    fx, nd, _ = _fixture_nodes(golden_dir, "f1_cfg1", tmp_path)
    words, lens = nd["words"].copy(), nd["len"].copy()
    pairs = len(lens) // 2
    k = next(q for q in range(7, pairs) if q % max(1, pairs // 256) != 0 and lens[2 * q] > 0 and lens[2 * q + 2] == lens[2 * q])
    words[2 * k] = words[2 * k + 2]                      # node 2k: now a copy of node 2k + 2 instead of the reverse complement of node 2k + 1
    nodes = str(tmp_path / "edited.nodes.bin")
    _write_nodes(nodes, words, lens)
    out_cpu, out_hip = str(tmp_path / "cpu.graph"), str(tmp_path / "hip.graph")
    _run("cpu", nodes, nd, out_cpu)
    _run("hip", nodes, nd, out_hip)
    assert open(out_hip, "rb").read() == open(out_cpu, "rb").read()
    assert open(out_cpu, "rb").read() != fx.ref_graph()          # (the edit did change the graph: the case tests something)
