"""N > 1 on CPU: the sharding driver (alga_amd/multigpu.py: source shards, record exchange by target owner, edge gather
and ordering) run with world_size 2 and 3 over gloo.  The engine needs a GPU, so a small pure-Python stand-in backend
(brute-force overlap discovery, literal replay of the per-target policy) takes its place here; what is under test is
the driver: shard bounds, split sizes, all_to_all / all_gather plumbing and the claim that the gathered graph equals
the single-process graph (checked against the CPU oracle)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

OL_SMALL = 1 << 31


def _seqs(words, lens):
    out = []
    for w, l in zip(words, lens):
        out.append("".join("ACGT"[(int(w[i >> 4]) >> ((i & 15) << 1)) & 3] for i in range(int(l))))
    return out


class PyBackend:
    """Stand-in for alga_amd.multigpu.HipBackend on CPU tensors."""

    def __init__(self, words, lens, lo, rs, source_side=False, decline_rank=None, rank=0, decline_phase=None):
        self.source_side, self.decline = source_side, decline_rank == rank
        self.decline_phase = decline_phase                  # "index" / "join": the declining rank declines that phase of the bucket-sharded form (and not the replicated build)
        self.seq = _seqs(words, lens)
        self.n = len(self.seq)
        self.lo, self.rs = lo, rs
        self.device = torch.device("cpu")
        self.stats = {}
        self.maxlen = max((len(s) for s in self.seq), default=0)

    def _records(self, a, b):
        pref = {}
        for c, s in enumerate(self.seq):
            for L in range(self.lo, len(s) + 1):
                pref.setdefault(s[:L], []).append(c)
        dst, val = [], []
        for B in range(a, b):
            s = self.seq[B]
            small = []
            for L in range(self.lo, min(len(s), self.maxlen, 500) + 1):
                for C in pref.get(s[len(s) - L:], ()):
                    if C == B:
                        continue
                    off = len(s) - L
                    if L < self.rs:
                        small.append((L, C, off))
                    else:
                        dst.append(C); val.append(((off | (L << 22)) << 32) | B)
            for L, C, off in sorted(small)[-3:]:
                dst.append(C); val.append(((off | (L << 22) | OL_SMALL) << 32) | B)
        return np.array(dst, dtype=np.int64), np.array(val, dtype=np.uint64)

    @staticmethod
    def _fake_keys(lo, hi):
        i = np.arange(lo, hi, dtype=np.int64)
        return ((i * 2654435761 + 12345) & 0x7FFFFFFF).astype(np.int32), ((i * 40503 + 7) & 0x7FFFFFFF).astype(np.int32)

    def node_keys(self, a, b, span):
        """Stand-in of alga_prefsuf_keys_device: per-node arrays of `span` entries with my range filled."""
        if not self.source_side:
            return None
        self.karr = [torch.full((span,), -1, dtype=torch.int32), torch.full((span,), -1, dtype=torch.int32)]
        self.kr = (a, b)
        k, m = self._fake_keys(a, b)
        self.karr[0][a:b] = torch.from_numpy(k)
        self.karr[1][a:b] = torch.from_numpy(m)
        return self.karr

    def build_range(self, a, b, collect_stats=False, keys_shared=0):
        """Stand-in of alga_prefsuf_build_range_device: the executable statement of the source-side rule, one source range."""
        if keys_shared == 2:                              # a later piece of the rank's range: the first piece came first
            assert self.first_piece_done
        elif keys_shared:                                 # the all-gather delivered every rank's slice
            self.first_piece_done = True
            k, m = self._fake_keys(0, self.n)
            assert (self.karr[0][:self.n].numpy() == k).all() and (self.karr[1][:self.n].numpy() == m).all()
        else:                                             # keys_shared == 0: the rank computes all keys itself (the plain form; the first piece of a two-rank run)
            self.first_piece_done = True
        if not self.source_side or (self.decline and self.decline_phase is None):
            return None
        from source_side_rule import source_side_edges
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        seqs = [bytes(code[c] for c in s) for s in self.seq]
        e = source_side_edges(seqs, self.lo, self.rs)
        e = e[(e[:, 0] >= a) & (e[:, 0] < b)]
        return torch.from_numpy(e.copy()).reshape(-1, 3)

    # ---- stand-ins of the bucket-sharded form's phases (alga_shard_*): the "bucket" of a target is a hash of its id, the
    #      per-target decision is tests/bucket_side_rule.py -- what is under test is the driver's five exchanges ----
    def _owner(self, c, world):
        return ((c * 2654435761) >> 7) % world

    def _seqs_bytes(self):
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        return [bytes(code[ch] for ch in s) for s in self.seq]

    def shard_index(self, rank, world):
        if self.decline_phase == "index" and self.decline:
            return None
        import bucket_side_rule as B
        self.world, self.rank_ = world, rank
        self.sb = self._seqs_bytes()
        self.cands = B.raw_overlaps(self.sb, self.lo)
        a, b = self.kr                                      # my node range (node_keys): a descriptor per (source, owner it reaches)
        rows, cnt = [[] for _ in range(world)], []
        reach = {}
        for c, lst in self.cands.items():
            for (src, d, L) in lst:
                if a <= src < b:
                    reach.setdefault(src, set()).add(self._owner(c, world))
        for src, owners in sorted(reach.items()):
            for q in owners:
                rows[q].append((q, src, 0))
        flat, off, cnt = [], [], []
        for q in range(world):
            off.append(len(flat) + 3 * q)                   # segments with slack between them, as the engine's have
            flat += [(-1, -1, -1)] * 3 + rows[q] if q else rows[q]
            cnt.append(len(rows[q]))
        off = [3 * q + sum(cnt[:q]) for q in range(world)]
        t = torch.tensor(flat, dtype=torch.int32).reshape(-1, 3)
        return t, cnt, off

    def shard_join(self, desc_in):
        if self.decline_phase == "join" and self.decline:
            return None
        import bucket_side_rule as B
        srcs = set(int(x) for x in desc_in[:, 1])
        assert all(int(x) == self.rank_ for x in desc_in[:, 0])              # only what falls into my buckets came
        self.final, self.pending = [], []
        for c, lst in self.cands.items():
            if self._owner(c, self.world) != self.rank_:
                continue
            assert all(a in srcs for (a, d, L) in lst)                         # every source that reaches my target sent its descriptor
            for (a, d, L) in B.target_survivors(self.sb, c, lst, self.lo, self.rs):
                (self.pending if L < self.rs else self.final).append((a, c, d, L))
        return torch.tensor([a for (a, c, d, L) in self.pending], dtype=torch.int32)

    def shard_small_keys(self, pending_all):
        want = set(int(x) for x in pending_all)
        rows = []
        for c, lst in self.cands.items():
            if self._owner(c, self.world) == self.rank_:
                rows += [(a, L, c) for (a, d, L) in lst if L < self.rs and a in want]
        return torch.tensor(rows, dtype=torch.int32).reshape(-1, 3)

    def shard_resolve(self, small_all, world):
        keys = {}
        for a, L, c in small_all.tolist():
            keys.setdefault(a, []).append((L, c))
        out = list(self.final)
        for (a, c, d, L) in self.pending:
            if sum(1 for k in keys[a] if k > (L, c)) < 3:
                out.append((a, c, d, L))
        from alga_amd.multigpu import shard_chunk
        ch = shard_chunk(self.n, world)
        rows = [[] for _ in range(world)]
        for (a, c, d, L) in out:
            rows[min(world - 1, a // ch)].append((a, c, d))
        cnt = [len(x) for x in rows]
        off = [sum(cnt[:q]) for q in range(world)]
        return torch.tensor([x for r in rows for x in r], dtype=torch.int32).reshape(-1, 3), cnt, off

    def shard_place(self, edges_in, a, b):
        e = edges_in.numpy().reshape(-1, 3)
        assert ((e[:, 0] >= a) & (e[:, 0] < b)).all()
        self.stats = {"edges": len(e)}
        return self.sort_edges(torch.from_numpy(e.copy())) if len(e) else torch.empty((0, 3), dtype=torch.int32)

    def discover_sorted(self, a, b, collect_stats=False):
        d, v = self._records(a, b)
        o = np.argsort(d, kind="stable")
        return torch.from_numpy(d[o].astype(np.int32)), torch.from_numpy(v[o].view(np.int64).copy())

    def reduce(self, rec_dst, rec_val, a, b, collect_stats=False):
        d = rec_dst.numpy().astype(np.int64)
        v = rec_val.numpy().view(np.uint64)
        edges = []
        for C in range(a, b):
            recs = []
            for x in v[d == C]:
                x = int(x)
                ol, src = x >> 32, x & 0xFFFFFFFF
                recs.append((0 if ol & OL_SMALL else 1, (ol >> 22) & 0x1FF, src, ol & 0x3FFFFF))
            recs.sort()
            live = []                                     # (A, offA, L_A)
            for big, L, B, off in recs:
                if not big:
                    hit = [i for i, e in enumerate(live) if e[0] == B]
                    if hit:
                        if off < live[hit[0]][1]:
                            live[hit[0]] = (B, off, L)
                    else:
                        live.append((B, off, L))
                    continue
                keep = []
                for A, offA, LA in live:
                    rm = A == B
                    if not rm and off > 0:
                        dd = offA - off
                        if dd >= 0 and L - LA >= 0 and self.seq[A][dd:dd + off] == self.seq[B][:off]:
                            rm = True
                    if not rm:
                        keep.append((A, offA, LA))
                live = keep + [(B, off, L)]
            edges += [(A, C, offA) for A, offA, _ in live]
        return torch.tensor(edges, dtype=torch.int32).reshape(-1, 3)

    def sort_edges(self, e):
        a = e.numpy()
        o = np.lexsort((a[:, 2], a[:, 1], a[:, 0]))
        return torch.from_numpy(a[o].copy())

    def build(self, collect_stats=False):
        d, v = self.discover_sorted(0, self.n)
        return self.sort_edges(self.reduce(d, v, 0, self.n))

    def sync(self):
        pass


def _worker(rank, world, port, words, lens, lo, rs, out_dir, source_side=False, decline_rank=None, replicate=False, opt_in=True):
    import torch.distributed as dist
    from alga_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # opt_in: the sharded key pass + key all-gather and several pieces per rank; else the driver's defaults (one piece, own keys)
        kw = dict(shard_keys=True, pieces=None) if opt_in else {}
        run = multigpu.ShardedPrefSuf(PyBackend(words, lens, lo, rs, source_side, decline_rank, rank), rank, world, dist, replicate=replicate, **kw)
        m, st = run.step(collect_stats=True)
        e = run.edges_numpy()
        assert m == len(e) or (rank != 0 and not replicate and len(e) == 0)
        np.save(os.path.join(out_dir, "count_%d.npy" % rank), np.array([m]))
        np.save(os.path.join(out_dir, "edges_%d.npy" % rank), e)
    finally:
        dist.destroy_process_group()


def _worker_bucket(rank, world, port, words, lens, lo, rs, out_dir, decline_rank, decline_phase):
    import torch.distributed as dist
    from alga_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = PyBackend(words, lens, lo, rs, True, decline_rank, rank, decline_phase)
        run = multigpu.ShardedPrefSuf(be, rank, world, dist, bucket_sharded=True)
        for _ in range(2):
            m, st = run.step()
        np.save(os.path.join(out_dir, "count_%d.npy" % rank), np.array([m]))
        np.save(os.path.join(out_dir, "edges_%d.npy" % rank), run.edges_numpy())
        with open(os.path.join(out_dir, "form_%d.txt" % rank), "w") as f:
            f.write(run.form_used)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,decline_rank,decline_phase", [(2, None, None), (3, None, None), (3, 1, "index"), (3, 2, "join")])
def test_bucket_sharded_driver_over_gloo_equals_oracle(tmp_path, world, decline_rank, decline_phase):
    """The five exchanges of the bucket-sharded form (descriptor all-to-all, pending-source and small-key all-gathers of any length
    per rank -- also zero --, edge all-to-all, gather to rank 0) over gloo, with a stand-in engine that decides per target by
    tests/bucket_side_rule.py and applies the per-source cap from the gathered small keys; one rank declining a phase makes ALL
    ranks continue in the replicated form.  Low coverage on purpose: most survivors are small overlaps, the cap really decides."""
    import gen_reads
    import oracle_lib as O
    import alga_amd
    codes, lens = gen_reads.sample_reads(150, 60, 900, 79)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 60)
    lens = np.repeat(lens, 2).astype(np.int32)
    words = alga_amd.pack_reads(codes, lens)
    lo, rs = 25, 45
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    assert len(want) > 50
    mp.spawn(_worker_bucket, args=(world, _free_port(), words, lens, lo, rs, str(tmp_path), decline_rank, decline_phase), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(str(tmp_path / ("count_%d.npy" % r)))[0]) == len(want)
        assert open(str(tmp_path / ("form_%d.txt" % r))).read() == ("replicated" if decline_rank is not None else "bucket_sharded")
    got = np.load(str(tmp_path / "edges_0.npy"))
    assert got.shape == want.shape and (got == want).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,source_side,decline_rank,replicate,opt_in", [
    (2, False, None, True, True), (3, False, None, False, True), (2, True, None, False, True), (3, True, None, True, True), (3, True, 1, False, True),
    (2, True, None, False, False), (3, True, None, False, False), (3, True, 2, False, False)])
def test_sharded_driver_over_gloo_equals_oracle(tmp_path, world, source_side, decline_rank, replicate, opt_in):
    """Per-target form (record exchange), source-side form (no exchange), and one rank declining the source-side form
    (capacity case): every rank must fall back together.  opt_in: sharded key pass + key all-gather + several pieces per rank;
    else the driver's defaults (one piece, every rank its own keys).  The edge lists reach rank 0 as exact-length point-to-point
    transfers landed at their offsets of one list."""
    import gen_reads
    import oracle_lib as O
    import alga_amd
    codes, lens = gen_reads.sample_reads(150, 60, 400, 77)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 60)
    lens = np.repeat(lens, 2).astype(np.int32)
    words = alga_amd.pack_reads(codes, lens)
    lo, rs = 25, 40
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    assert len(want) > 50
    single = PyBackend(words, lens, lo, rs).build().numpy()          # the stand-in itself agrees with the oracle
    assert (single == want).all()
    mp.spawn(_worker, args=(world, _free_port(), words, lens, lo, rs, str(tmp_path), source_side, decline_rank, replicate, opt_in), nprocs=world, join=True)
    for r in range(world):
        got = np.load(str(tmp_path / ("edges_%d.npy" % r)))
        assert int(np.load(str(tmp_path / ("count_%d.npy" % r)))[0]) == len(want)      # every rank knows the size of the graph
        if r == 0 or replicate:
            assert got.shape == want.shape and (got == want).all()  # the complete, ordered graph (rank 0; every rank if replicated)
        else:
            assert len(got) == 0


def _worker_validated(rank, world, port, words, lens, lo, rs, out_dir, corrupt):
    import torch.distributed as dist
    from alga_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = PyBackend(words, lens, lo, rs, True, None, rank)
        if corrupt and rank == 1:
            inner = be.build_range

            def bad(a, b, collect_stats=False, keys_shared=0):      # the sharded form (keys_shared != 0) of one rank loses an edge
                e = inner(a, b, collect_stats, keys_shared=keys_shared)
                return e[1:] if keys_shared and len(e) else e
            be.build_range = bad
        run, form = multigpu.validated_runner(be, rank, world, dist)
        m, _ = run.step()
        with open(os.path.join(out_dir, "form_%d.txt" % rank), "w") as f:
            f.write("%d|%d|%s|%s" % (m, run.pieces, form["form"], form["validated"]))
        np.save(os.path.join(out_dir, "edges_%d.npy" % rank), run.edges_numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_validated_runner_takes_the_sharded_form_only_when_it_reproduces_the_plain_one(tmp_path, corrupt):
    """bench.py --gpus N: the sharded key pass + pieces are timed only after their graph equalled the plain form's in the same
    process group (alga_amd.multigpu.validated_runner); a rank whose sharded form differs makes ALL ranks stay plain."""
    import gen_reads
    import oracle_lib as O
    import alga_amd
    codes, lens = gen_reads.sample_reads(150, 60, 400, 78)
    rc = (3 - codes)[:, ::-1]
    codes = np.stack([rc, codes], axis=1).reshape(-1, 60)
    lens = np.repeat(lens, 2).astype(np.int32)
    words = alga_amd.pack_reads(codes, lens)
    lo, rs = 25, 40
    want, _, _ = O.prefsuf(words, lens, lo, rs)
    mp.spawn(_worker_validated, args=(2, _free_port(), words, lens, lo, rs, str(tmp_path), corrupt), nprocs=2, join=True)
    for r in range(2):
        m, pieces, form, validated = open(str(tmp_path / ("form_%d.txt" % r))).read().split("|", 3)
        assert int(m) == len(want)
        if corrupt:
            assert form.startswith("plain") and "NOT taken" in validated and int(pieces) == 1
        else:
            assert form.startswith("all keys on every rank") and "byte-identical" in validated and int(pieces) > 1      # (two ranks: pieces without the key all-gather)
    got = np.load(str(tmp_path / "edges_0.npy"))
    assert got.shape == want.shape and (got == want).all()


def test_shard_bounds_keep_twins_together():
    from alga_amd.multigpu import shard_bounds
    for n in (0, 2, 10, 1700526, 99999998):
        for w in (1, 2, 3, 4, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:])) and all(x % 2 == 0 for x in b[:-1])


def test_piece_bounds_cover_the_rank_range():
    """the pieces a rank cuts its source range into (alga_amd/multigpu.py): consecutive, even-aligned, covering, for every world size"""
    from alga_amd.multigpu import shard_bounds, piece_bounds
    for n in (10, 2000, 1700526, 90621096):
        for w in (2, 3, 4, 8):
            b = shard_bounds(n, w)
            for r in range(w):
                for pieces in (1, 2, 4):
                    pb = piece_bounds(b[r], b[r + 1], pieces)
                    assert len(pb) == pieces + 1 and pb[0] == b[r] and pb[-1] == b[r + 1] and all(x <= y for x, y in zip(pb, pb[1:]))
                    assert all((x - b[r]) % 2 == 0 for x in pb[:-1])
                    if b[r + 1] - b[r] > 1000 and pieces > 1:          # each piece about three quarters of the one before it
                        sz = [y - x for x, y in zip(pb, pb[1:])]
                        assert all(0.70 < q / p < 0.80 for p, q in zip(sz, sz[1:]))


class PySupplementBackend:
    """Stand-in for the supplement phases of alga_amd.multigpu.HipBackend (alga_pkb_shard_*) on CPU tensors: a toy 'group join' whose groups are
    dealt out by hash -- a candidate edge (a, b, off) of round r belongs to rank mix(a * 131 + b) mod world -- with the engine's semantics (every
    addition is judged against the graph as it stood when the round began; the merge keeps the smallest offset per pair)."""

    def __init__(self, n_nodes, rounds_of_candidates):
        self.n, self.cands = n_nodes, rounds_of_candidates
        self.device = torch.device("cpu")

    def pkb_begin(self, edges, rank, world):
        self.rank, self.world, self.round = rank, world, 0
        e = edges.numpy().astype(np.int64)
        self.g = {(int(a), int(b)): int(o) for a, b, o in e}

    def pkb_round(self):
        out = []
        for a, b, off in self.cands[self.round]:
            if ((a * 131 + b) * 2654435761 >> 7) % self.world != self.rank:
                continue
            cur = self.g.get((a, b))
            if cur is None or cur > off:                      # addDirectedEdge keeps the minimum (src/DataStructures/Graph.cpp:53-71)
                out.append((a << 36) | (b << 9) | off)
        return torch.tensor(out, dtype=torch.int64)

    def pkb_merge(self, allk):
        for k in allk.tolist():
            a, b, off = k >> 36, (k >> 9) & ((1 << 27) - 1), k & 511
            if self.g.get((a, b), 1 << 30) > off:
                self.g[(a, b)] = off
        self.round += 1

    def pkb_end(self):
        e = sorted((a, b, o) for (a, b), o in self.g.items())
        return torch.tensor(e, dtype=torch.int32).reshape(-1, 3)


def _supp_inputs():
    rng = np.random.default_rng(77)
    n = 400
    pre = sorted({(int(a), int(b)): int(o) for a, b, o in zip(rng.integers(0, n, 900), rng.integers(0, n, 900), rng.integers(1, 60, 900))}.items())
    pre = np.array([(a, b, o) for (a, b), o in pre], dtype=np.int32)
    rounds = [[(int(a), int(b), int(o)) for a, b, o in zip(rng.integers(0, n, 700), rng.integers(0, n, 700), rng.integers(0, 50, 700))] for _ in range(4)]
    return n, pre, rounds


def _worker_supplement(rank, world, port, out_dir):
    import torch.distributed as dist
    from alga_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, pre, rounds = _supp_inputs()
        sup = multigpu.ShardedSupplement(PySupplementBackend(n, rounds), rank, world, dist)
        # the exact graph is complete on rank 0 only (the gather of the build); the driver broadcasts it
        out = sup.run(torch.from_numpy(pre) if rank == 0 else torch.empty((0, 3), dtype=torch.int32))
        np.save(os.path.join(out_dir, "supp_%d.npy" % rank), out.numpy())
        assert len(sup.exchange_bytes) == 4
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_supplement_driver_over_gloo(tmp_path, world):
    """alga_amd.multigpu.ShardedSupplement (what bench.py --gpus N runs on configs[4]): broadcast of the exact graph, four rounds of {own
    additions, variable-length all-gather, merge of all} over gloo -- every rank must end with the graph one process gives."""
    n, pre, rounds = _supp_inputs()
    one = PySupplementBackend(n, rounds)
    one.pkb_begin(torch.from_numpy(pre), 0, 1)
    for _ in range(4):
        one.pkb_merge(one.pkb_round())
    want = one.pkb_end().numpy()
    assert len(want) > len(pre)
    mp.spawn(_worker_supplement, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "supp_%d.npy" % r))
        assert got.shape == want.shape and (got == want).all(), r
