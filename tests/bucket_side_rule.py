"""Executable statement of the BUCKET-SIDE form of the transitive reduction (the seed-bucket-sharded N-GPU build) --
TEST INFRASTRUCTURE.

The N-GPU build shards the TARGETS by seed bucket: the rank that owns target C's bucket sees every source A with a raw overlap
(A -> C) -- they all reach C through C's one minimizer -- but it does not see A's other overlaps.  The source-side rule
(tests/source_side_rule.py) decides (A, C, d) from A's items; every via B it can use is itself a source of C (B -> C is a big raw
overlap), so the same decision can be taken from C's candidate list:

  candidates of C = raw overlaps (A -> C, offset d_A), L_A = |A| - d_A
  superseded      = another candidate of the same A with a smaller offset (whatever became of it: a kept one supersedes, a small
                    one the cap dropped has the larger L, so the cap drops this one as well)
  via B for A     = candidate (B, d_B), B != A, 0 < d_B <= d_A, delta = d_A - d_B (B sits at offset delta of A),
                    L_BC = |B| - d_B >= max(rsoemo, min_overlap), rho_B = |B| - (|A| - delta) with 0 <= rho_B <= rho_C = |C| - L_A,
                    (rho_B > 0 or B > A), alignTo[B], and A[delta:] == B[:|A| - delta]   (B is a raw overlap of A)
  kept            = L_A >= rsoemo, or (L_A, C) among the 3 largest small keys of A -- the ONLY part that needs A's other overlaps:
                    in the sharded build a small survivor is PENDING until the per-run top-3 small keys of its source have been
                    gathered from the bucket owners (the per-source cap, GraphCreatorPrefSuf.cpp:397-401)
  edge (A, C, d)  iff not superseded, no via, kept.
"""
import numpy as np


def raw_overlaps(seqs, min_overlap, align_from=None, align_to=None, cap=501):
    """-> {target: [(A, d, L)]}: every raw overlap, grouped by target"""
    n = len(seqs)
    af = [True] * n if align_from is None else [bool(x) for x in align_from]
    at = [True] * n if align_to is None else [bool(x) for x in align_to]
    table = {}
    for c, s in enumerate(seqs):
        if len(s) >= min_overlap and len(s) > 0 and at[c]:
            table.setdefault(s[:min_overlap], []).append(c)
    cands = {}
    for a, sa in enumerate(seqs):
        la = len(sa)
        if la < min_overlap or la == 0 or not af[a]:
            continue
        for d in range(max(0, la - cap), la - min_overlap + 1):
            L = la - d
            for c in table.get(sa[d: d + min_overlap], ()):
                sc = seqs[c]
                if c != a and len(sc) >= L and sc[:L] == sa[d:]:
                    cands.setdefault(c, []).append((a, d, L))
    return cands


def target_survivors(seqs, c, lst, min_overlap, rsoemo, align_to=None):
    """the candidates (A, d, L) of target c that no other candidate supersedes or removes (the cap is NOT applied)"""
    at = [True] * len(seqs) if align_to is None else [bool(x) for x in align_to]
    big_min = max(rsoemo, min_overlap)
    lc = len(seqs[c])
    out = []
    for (a, d, L) in lst:
        la = len(seqs[a])
        if any(a2 == a and d2 < d for (a2, d2, _) in lst):
            continue
        rho_c = lc - L
        removed = False
        for (b, db, Lb) in lst:
            if b == a or db <= 0 or db > d:
                continue
            delta = d - db
            lb = len(seqs[b])
            rho_b = lb - (la - delta)
            if lb - db >= big_min and 0 <= rho_b <= rho_c and (rho_b > 0 or b > a) and at[b] and seqs[a][delta:] == seqs[b][: la - delta]:
                removed = True
                break
        if not removed:
            out.append((a, d, L))
    return out


def bucket_side_edges(seqs, min_overlap, rsoemo, align_from=None, align_to=None, cap=501, want_pending=False):
    cands = raw_overlaps(seqs, min_overlap, align_from, align_to, cap)
    small = {}
    for c, lst in cands.items():
        for (a, d, L) in lst:
            if L < rsoemo:
                small.setdefault(a, []).append((L, c))
    top3 = {a: set(sorted(v, reverse=True)[:3]) for a, v in small.items()}
    edges, pending = [], 0
    for c, lst in cands.items():
        for (a, d, L) in target_survivors(seqs, c, lst, min_overlap, rsoemo, align_to):
            if L < rsoemo:
                pending += 1
                if (L, c) not in top3[a]:
                    continue
            edges.append((a, c, d))
    edges.sort()
    e = np.array(edges, dtype=np.int32).reshape(-1, 3)
    return (e, pending) if want_pending else e
