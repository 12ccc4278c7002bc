"""Full-size bit-exactness: the engine's graph against the REAL reference's --serialize=1 dump, byte for byte, at the size of
BASELINE.json configs[1] (1 M x 150 bp), through both probes of the source-side form and the per-target form.

The reference binary (oracle/_ref/ALGA: the unmodified reference compiled from /root/reference by oracle/Makefile, a built
file that travels with the snapshot) is the checker here, not the thing measured.  Error-free data gives the same dump for
any thread count; data with sequencing errors is compared against --threads=1, the canonical order.
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import alga_amd
import gen_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alga_amd import workload  # noqa: E402

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref", "ALGA")


def _reference_dump(codes, threads, wd):
    workload.write_fasta_fast(os.path.join(wd, "s.fasta"), codes)
    p = subprocess.Popen([REF, "--file1=s.fasta", "--threads=%d" % threads, "--serialize=1", "--output=o.fasta"], cwd=wd,
                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace", bufsize=1)
    for line in p.stderr:                       # the dump is complete before this marker (src/main.cpp:293 precedes :380)
        if "Before first simplifier" in line:
            break
    p.kill()
    p.wait()
    dumps = glob.glob(os.path.join(wd, "*_beforeSimplifier.graph"))
    assert dumps, "the reference wrote no dump"
    return open(dumps[0], "rb").read()


def _engine_dump(eng, words, lens, lo, rs, wd, **kw):
    e = eng.prefsuf_host(words, lens, lo, rs, **kw)
    path = os.path.join(wd, "gpu.graph")
    eng.write_graph(path, len(lens), e)
    return open(path, "rb").read(), len(e)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/ALGA is not built")
@pytest.mark.parametrize("n,G,seed,err,threads", [
    (1_000_000, 3_000_000, 3, 0.0, 16),        # BASELINE configs[1]
    (1_000_000, 3_000_000, 5, 0.02, 1),        # 2 % substitutions: small-overlap ties, many survivors per source
])
def test_graph_dump_bytes_equal_reference(n, G, seed, err, threads):
    codes, _ = gen_reads.sample_reads(n, 150, G, seed, err)
    words, lens, _ = workload.make_nodes(codes)
    lo, rs = workload.derive_params(144.0)
    eng = alga_amd.Engine(0)
    try:
        with tempfile.TemporaryDirectory() as wd:
            want = _reference_dump(codes, min(threads, os.cpu_count() or 1), wd)
            for probe in ("table", "cluster"):
                eng.set_option("probe", probe)
                got, m = _engine_dump(eng, words, lens, lo, rs, wd, reduction="source_side")
                assert eng.last_stats()["probe_used"] == (2 if probe == "cluster" else 1)
                assert got == want, "source-side form, %s probe: %d edges" % (probe, m)
            eng.set_option("probe", "auto")
            got, m = _engine_dump(eng, words, lens, lo, rs, wd, reduction="per_target")
            assert got == want, "per-target form: %d edges" % m
    finally:
        eng.close()
