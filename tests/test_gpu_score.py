"""The stated tolerance of the approximate path (error_rate > 0.01; BASELINE configs[4]), asserted downstream: the engine's
post-supplement graph and the reference's go through the SAME unchanged simplifier and contig stages of stock ALGA and the contigs
are compared (tools/score_supplement.py; profiles/r03_score_*).

The engine's supplement is order independent (every group sees the round's start graph, ties by node id), the reference's is not
(groups in sequence, std::sort's tie order, races with --threads > 1), so the two edge sets differ -- 0.4 % of the edges at 1 M reads,
2.0 % at 10 M -- and the contig SETS differ with them (83 % / 49 % of the contig bp sit in contigs found identically).  What is
bounded is the quality of what comes out.  Measured at 1 M and at 10 M reads (configs[4] itself): contig count -0.3 % / +2.0 %,
total contig bp +2.2 % / +13.6 %, N50 +3.0 % / +14.9 % (engine relative to the reference; the reference against a second run of
itself: < 0.06 % on all three).  The tolerance asserted here, chosen to hold at both sizes:
    |contigs - ref| <= 3 %,   total bp >= 0.97 x ref,   N50 >= 0.97 x ref,   |post-supplement edges - ref| <= 3 %.

Those bounds are one-sided -- a false join RAISES N50 -- so the referee that knows the truth decides (round 4): both contig sets are placed
on the synthetic genome the reads were drawn from (tools/genome_score.py: full length, one diagonal, <= 2 % mismatches, either strand):
    engine misjoined contigs <= reference misjoined contigs,   engine genome fraction >= reference genome fraction - 0.5 %."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _edges_after(log):
    for line in log:
        m = re.search(r"After supplement G has (\d+) edges", line)
        if m:
            return int(m.group(1))
    return None


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ALGA")), reason="the reference binary (oracle/_ref/ALGA) is not built")
def test_supplement_tolerance_at_1m_reads(tmp_path, monkeypatch):
    import score_supplement
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    r = score_supplement.score(1_000_000, 3_000_000, T=16, seed=13, ref_runs=1)
    e, ref = r["engine"], r["ref1"]
    assert abs(e["contigs"] - ref["contigs"]) <= 0.03 * ref["contigs"], r
    assert e["total_bp"] >= 0.97 * ref["total_bp"], r
    assert e["n50"] >= 0.97 * ref["n50"], r
    ea, ra = _edges_after(r["engine_log"]), _edges_after(r["ref1_log"])
    assert ea and ra and abs(ea - ra) <= 0.03 * ra, r
    # against the genome: longer contigs must not be bought with false joins
    ge, gr = r["genome"]["engine"], r["genome"]["ref1"]
    assert ge["misjoined"] <= gr["misjoined"], (ge, gr)
    assert ge["genome_fraction"] >= gr["genome_fraction"] - 0.005, (ge, gr)
    assert ge["unplaced"] <= gr["unplaced"], (ge, gr)
