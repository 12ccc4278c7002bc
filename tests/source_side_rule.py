"""Executable statement of the SOURCE-SIDE form of the transitive reduction -- TEST INFRASTRUCTURE.

The reference removes implied overlaps per TARGET, replaying pushes in (L, source) order
(src/GraphCreators/GraphCreatorPrefSuf.cpp:397-483).  The engine's fast path decides the same thing per SOURCE A from
A's own raw out-overlaps only:

  items of A      = raw overlaps (A -> X, offset d_X), every L >= min_overlap
  kept(X)         = L_X >= rsoemo  or  (L_X, X) among the 3 largest small keys of A          (:397-401)
  overhang(X)     = X[L_X:]   (what X adds to the right of A's end), rho_X = |X| - L_X
  via(B, C)       = B != C, d_B < d_C, alignFrom[B], L_BC = |B| - (d_C - d_B) >= max(rsoemo, min_overlap),
                    rho_B <= rho_C, overhang(B) is a prefix of overhang(C), and (rho_B > 0 or B > A)
  edge (A, C, d)  iff kept(C), no kept instance of the same C with a smaller d, and no via(B, C)

This file is the slow, obviously-structured version used to check that statement against the oracle's literal replay;
the HIP kernel (prefsuf_device.h: local_reduce) is then checked against the oracle directly.
"""
import numpy as np


def decode_rows(words, lens):
    """2-bit packed rows -> list of bytes objects with codes 0..3"""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n, W = words.shape
    shifts = np.arange(0, 32, 2, dtype=np.uint32)
    codes = ((words[:, :, None] >> shifts[None, None, :]) & 3).astype(np.uint8).reshape(n, W * 16)
    return [codes[i, : int(lens[i])].tobytes() for i in range(n)]


def source_side_edges(seqs, min_overlap, rsoemo, align_from=None, align_to=None, cap=501):
    n = len(seqs)
    af = [True] * n if align_from is None else [bool(x) for x in align_from]
    at = [True] * n if align_to is None else [bool(x) for x in align_to]
    table = {}
    for c, s in enumerate(seqs):
        if len(s) >= min_overlap and len(s) > 0 and at[c]:
            table.setdefault(s[:min_overlap], []).append(c)
    big_min = max(rsoemo, min_overlap)
    edges = []
    for a, sa in enumerate(seqs):
        la = len(sa)
        if la < min_overlap or la == 0 or not af[a]:
            continue
        items = []  # (d, C, L, rho, overhang)
        for d in range(max(0, la - cap), la - min_overlap + 1):
            L = la - d
            for c in table.get(sa[d: d + min_overlap], ()):
                sc = seqs[c]
                if c != a and len(sc) >= L and sc[:L] == sa[d:]:
                    items.append((d, c, L, len(sc) - L, sc[L:]))
        small = sorted(((L, c) for (d, c, L, rho, o) in items if L < rsoemo), reverse=True)[:3]
        small = set(small)
        kept = [L >= rsoemo or (L, c) in small for (d, c, L, rho, o) in items]
        for i, (d, c, L, rho, o) in enumerate(items):
            if not kept[i]:
                continue
            removed = False
            for j, (dj, b, Lj, rhoj, oj) in enumerate(items):
                if dj >= d:
                    continue
                if b == c:
                    if kept[j]:
                        removed = True
                        break
                    continue
                Lv = len(seqs[b]) - (d - dj)
                if af[b] and Lv >= big_min and rhoj <= rho and (rhoj > 0 or b > a) and o[:rhoj] == oj:
                    removed = True
                    break
            if not removed:
                edges.append((a, c, d))
    edges.sort()
    return np.array(edges, dtype=np.int32).reshape(-1, 3)


def preconditions(lens, min_overlap, rsoemo, align_from=None, align_to=None, cap=500):
    """When the source-side form is exact (the engine checks the same on the device, engine.hip: local_ok)."""
    lens = np.asarray(lens)
    live = lens > 0
    mx = int(lens.max()) if len(lens) else 0
    if mx > cap:
        return False                                    # an overlap of A with its via may exceed the 501 cap
    if not (min_overlap <= rsoemo <= min(mx, cap) + 1):
        return False                                    # reversal quirks (GraphCreatorPrefSuf.cpp:288-296)
    if align_to is not None:
        f = np.ones(len(lens), bool) if align_from is None else np.asarray(align_from, bool)
        if np.any(live & f & ~np.asarray(align_to, bool) & (lens >= min_overlap)):
            return False                                # a via must itself be discoverable as a target of A
    return True
