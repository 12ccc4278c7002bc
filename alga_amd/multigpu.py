"""One process per GPU: sharding of the PrefSuf build across ranks (torch.distributed = RCCL on ROCm).

Layout: the packed node set is replicated on every GPU (36-48 B/read; 4.8 GB at 100 M nodes, HBM is 288 GB).

Source-side form (alga_reduction; the normal case for short reads): rank r builds the FINAL edges of
the sources in its contiguous id range.  The build step is sharded as well: rank r computes the minimizer keys (and the
probe runs) of its own nodes, the per-node key arrays are all-gathered (8 bytes per node), and every rank sorts them into its
own copy of the bucket-ordered entry array.  After that nothing a rank computes depends on another rank; the remaining
collectives are one small all_gather (does every rank's input allow the form? edge counts) and the gather of the edge lists,
which arrive in (src, dst) order because the ranges are ascending.

Per-target form (any input; taken by all ranks when one of them cannot use the source-side form):
  1. discover  rank r probes the SOURCES of its contiguous id range against the full seed table and applies the
               per-source small-overlap cap locally (the cap is per source, so it needs no exchange), then orders
               its records by target id (device radix sort): the slice for every owner is contiguous
  2. exchange  overlap records go to the rank that owns the TARGET id range          (all_to_all_single x2)
  3. reduce    per-target transitive reduction of the owned targets                   (local)
  4. gather    edge lists of all ranks                                                 (all_gather, padded)
  5. order     every rank orders the gathered edges by (src, dst)                     (device radix sort)
so the result is byte-identical for every world size.  torch supplies buffers and collectives only; every compute
step is a C-ABI call into the HIP engine (`HipBackend`).  The same driver runs on CPU tensors over gloo with a
stand-in backend in tests/test_multigpu_gloo.py.
"""
import contextlib
import time

import numpy as np


def shard_chunk(n, world):
    """Ids per rank: equal, even (a read and its reverse complement, ids 2i and 2i+1, stay together)."""
    return 2 * ((n + 2 * world - 1) // (2 * world))


def shard_bounds(n, world):
    """Contiguous id ranges of `shard_chunk` ids (the last non-empty one may be shorter): rank r owns [b[r], b[r+1]).
    Equal chunks make the per-node arrays all-gatherable in place (slice r = [r * chunk, (r + 1) * chunk))."""
    c = shard_chunk(n, world)
    return [min(n, r * c) for r in range(world)] + [n]


PIECE_RATIO = 0.75      # a piece is this fraction of the one before it (piece_bounds)


def piece_bounds(lo, hi, pieces, ratio=PIECE_RATIO):
    """A rank's source range [lo, hi) in `pieces` consecutive pieces, even-aligned (a read and its reverse complement stay together), each
    `ratio` times the one before: the edges of a piece travel while the next piece is probed, so what the last piece ships is not hidden --
    shrinking pieces leave a short tail, and a piece's transfer (link time ~0.73 of its probe time at two ranks, ~0.79 at four and eight:
    DESIGN.md section 7) still ends before the next, smaller piece's probe does."""
    pairs = (hi - lo) // 2
    w = [ratio ** k for k in range(pieces)]
    tot = sum(w)
    cuts, acc = [lo], 0.0
    for k in range(pieces - 1):
        acc += w[k]
        cuts.append(lo + 2 * min(pairs, int(round(pairs * acc / tot))))
    cuts.append(hi)
    for k in range(1, len(cuts)):
        cuts[k] = max(cuts[k], cuts[k - 1])
    return cuts


KEY_ARRAY_SLACK = 1024      # the engine's per-node key arrays have room for n + 1024 entries (include/alga_amd.h): world * chunk <= n + 2 * world


class HipBackend:
    """The engine behind the driver: device tensors in, device tensors (views of engine memory) out.
    A step runs inside `stream_scope()`: one side stream of this backend is torch's CURRENT stream for its duration and the stream
    of every engine call, so collectives (RCCL), torch ops that produce an engine input and engine kernels that consume it are
    ordered by the stream itself.  (Not torch's default stream: its handle is 0, which the C ABI reads as "the engine's own
    stream", a non-blocking one that orders with nothing else -- an engine call that does not end in a host sync, like
    alga_prefsuf_keys_device, would then race with the torch ops around it.)"""

    def __init__(self, engine, d_words, d_lens, min_overlap, rsoemo):
        self.eng, self.w, self.l = engine, d_words, d_lens
        self.lo, self.rs = int(min_overlap), int(rsoemo)
        self.n = int(d_lens.shape[0])
        self.device = d_words.device
        self.stats = {}
        import torch
        self.stream = torch.cuda.Stream(device=self.device)

    def _stream(self):
        return self.stream.cuda_stream

    @contextlib.contextmanager
    def stream_scope(self):
        import torch
        outer = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(outer)                    # inputs made on the caller's stream
        with torch.cuda.stream(self.stream):
            yield
        outer.wait_stream(self.stream)                    # results read on the caller's stream

    def build(self, collect_stats=False):
        from .engine import device_view
        ptr, m = self.eng.prefsuf_device(self.w, self.l, self.lo, self.rs, collect_stats=collect_stats, stream=self._stream())
        self.stats = self.eng.last_stats()
        return device_view(ptr, (m, 3), self.device)

    def node_keys(self, node_begin, node_end, span):
        """Minimizer keys of my node range -> the per-node arrays to all-gather (views of engine memory, `span` entries each, my
        range filled), or None when the clustered probe does not take this input."""
        from .engine import device_view
        r = self.eng.keys_device(self.w, self.l, self.lo, self.rs, node_begin, node_end, stream=self._stream(), want_meta_flag=True)
        if r is None:
            return None
        assert span <= self.n + KEY_ARRAY_SLACK
        d_keys, d_meta, meta_needed = r
        # the meta array travels only when the build reads it (reads of different lengths, or an alignFrom mask): the same answer on every
        # rank for the same node set -- half the key traffic for every BASELINE configuration
        return [device_view(p, (span,), self.device) for p in ((d_keys, d_meta) if meta_needed else (d_keys,))]

    def build_range(self, src_begin, src_end, collect_stats=False, keys_shared=0):
        """Final edges of the sources in the range (tensor [m, 3]) or None when the source-side form is not exact here."""
        from .engine import device_view
        r = self.eng.build_range_device(self.w, self.l, self.lo, self.rs, src_begin, src_end, collect_stats=collect_stats, stream=self._stream(),
                                        keys_shared=keys_shared)
        if r is None:
            return None
        self.stats = self.eng.last_stats()
        return device_view(r[0], (r[1], 3), self.device)

    # ---- the bucket-sharded form, phase by phase (include/alga_amd.h: alga_shard_*): tensors are views of engine memory ----
    def shard_index(self, rank, world):
        """-> (descriptors [cap, 3] int32, counts, offsets) or None (the form does not take this input)"""
        from .engine import device_view
        r = self.eng.shard_index_device(self.w, self.l, self.lo, self.rs, rank, world, stream=self._stream())
        if r is None:
            return None
        ptr, cnt, off = r
        cap = max(o + c for o, c in zip(off, cnt))
        return device_view(ptr, (cap, 3), self.device), cnt, off

    def shard_join(self, desc_in):
        """received descriptors [m, 3] -> the sources of my pending edges (int32 tensor) or None (declined)"""
        from .engine import device_view
        r = self.eng.shard_join_device(self.w, self.l, desc_in, int(desc_in.shape[0]), stream=self._stream())
        if r is None:
            return None
        return device_view(r[0], (r[1],), self.device)

    def shard_small_keys(self, pending_all):
        from .engine import device_view
        ptr, k = self.eng.shard_small_keys_device(pending_all, int(pending_all.shape[0]), stream=self._stream())
        return device_view(ptr, (k, 3), self.device)

    def shard_resolve(self, small_all, world):
        from .engine import device_view
        ptr, cnt, off = self.eng.shard_resolve_device(small_all, int(small_all.shape[0]), world, stream=self._stream())
        return device_view(ptr, (sum(cnt), 3), self.device), cnt, off

    def shard_place(self, edges_in, src_begin, src_end):
        from .engine import device_view
        ptr, k = self.eng.shard_place_device(edges_in, int(edges_in.shape[0]), src_begin, src_end, stream=self._stream())
        self.stats = dict(self.eng.shard_stats())
        self.stats["edges"] = k
        return device_view(ptr, (k, 3), self.device)

    def discover_sorted(self, src_begin, src_end, collect_stats=False):
        from .engine import device_view
        d, v, k = self.eng.discover_device(self.w, self.l, self.lo, self.rs, src_begin, src_end, collect_stats=collect_stats, stream=self._stream())
        self.stats = self.eng.last_stats()
        ds, vs, nv = self.eng.sort_records_device(d, v, k, self.n, stream=self._stream())
        return device_view(ds, (nv,), self.device), device_view(vs, (nv,), self.device, "<i8")

    def reduce(self, rec_dst, rec_val, dst_begin, dst_end, collect_stats=False):
        from .engine import device_view
        ptr, m = self.eng.reduce_device(self.w, self.l, self.lo, self.rs, rec_dst, rec_val, int(rec_dst.shape[0]), dst_begin, dst_end,
                                        collect_stats=collect_stats, stream=self._stream())
        st = self.eng.last_stats()
        for k in ("ms_group", "ms_reduce", "ms_emit", "transitive_listed", "transitive_compares", "transitive_removed", "max_in_records"):
            self.stats[k] = st[k]
        return device_view(ptr, (m, 3), self.device)

    def sort_edges(self, edges):
        from .engine import device_view
        m = int(edges.shape[0])
        ptr = self.eng.sort_edges_device(edges, m, self.n, stream=self._stream())
        return device_view(ptr, (m, 3), self.device)

    # ---- the supplement on N ranks (ShardedSupplement); self.pkb = the alga_pkb_params (Engine.pkb_params), set by the caller ----
    pkb = None

    def pkb_begin(self, edges, rank, world):
        self._pkb_edges = edges.contiguous()                # (stays alive until the supplement has read it)
        self.eng.pkb_shard_begin(self.w, self.l, self._pkb_edges.data_ptr() if int(edges.shape[0]) else None, int(edges.shape[0]), self.pkb, rank, world, stream=self._stream())

    def pkb_round(self):
        import torch
        from .engine import device_view
        ptr, k = self.eng.pkb_shard_round(stream=self._stream())
        return device_view(ptr, (k,), self.device, typestr="<i8").clone() if k else torch.empty(0, dtype=torch.int64, device=self.device)

    def pkb_merge(self, allk):
        self._pkb_all = allk.contiguous()
        self.eng.pkb_shard_merge(self._pkb_all.data_ptr() if int(allk.shape[0]) else None, int(allk.shape[0]), stream=self._stream())

    def pkb_end(self):
        from .engine import device_view
        ptr, m = self.eng.pkb_shard_end(stream=self._stream())
        return device_view(ptr, (m, 3), self.device)

    def sync(self):
        import torch
        torch.cuda.synchronize()


class ShardedPrefSuf:
    """replicate=False (default): the complete graph is assembled on rank 0 only -- the one process that runs the unchanged
    simplifier / contig stages afterwards -- with direct sends over each peer's own xGMI link (`gather`); the other ranks keep
    an empty tensor.  replicate=True: every rank gets it (`all_gather`, 8x the traffic at 8 GPUs)."""

    def __init__(self, backend, rank=0, world=1, dist=None, replicate=False, shard_keys=False, pieces=1, bucket_sharded=False):
        self.be, self.rank, self.world, self.dist = backend, rank, world, dist
        self.replicate = replicate
        # Defaults = the form with the fewest moving parts (one piece, every rank computes all keys itself, no key all-gather): no
        # collective of this driver has run over RCCL on hardware yet, and the sharded key pass with several outstanding gathers is
        # exactly what a stand-in transport cannot validate.  Opt in once tools/multigpu_check.py has passed on an
        # N-GPU node:
        #   pieces     pieces of a rank's source range (the transfer of one overlaps the probe of the next).  A piece costs ~0.3 ms of
        #              its own (launches, host syncs, thinner kernels): 4 pieces up to 4 ranks, 2 at 8 were the measured optimum
        #   shard_keys every rank computes the minimizer keys of its own nodes only, the key arrays are all-gathered
        self.pieces = max(1, int(pieces)) if pieces is not None else min(4, max(1, 16 // max(1, world)))
        self.shard_keys = shard_keys or bucket_sharded
        # bucket_sharded: the INDEX is sharded by seed bucket (alga_shard_*): every rank builds 1 / world of the entry
        # array, run descriptors travel to the bucket's owner, the reduction is decided per target there, edges return to the source's
        # owner.  A build the form declines (any rank) continues in the replicated form with the gathered keys.
        self.bucket_sharded = bucket_sharded
        self.form_used = None
        self.exchange_bytes = {}
        self.n = backend.n
        self.bounds = shard_bounds(self.n, world)
        self.edges = None                       # tensor [m, 3] of the last step: the complete graph (rank 0, or every rank if replicate)
        self.total_edges = 0

    def step(self, collect_stats=False):
        """-> (n_edges of the complete graph, stats dict of this rank)."""
        with getattr(self.be, "stream_scope", contextlib.nullcontext)():
            if self.world == 1:
                self.edges = self.be.build(collect_stats)
                return int(self.edges.shape[0]), dict(self.be.stats)
            return self._step_sharded(collect_stats)

    def _step_sharded(self, collect_stats):
        import torch
        dist, be, r, nr, b = self.dist, self.be, self.rank, self.world, self.bounds
        dev = be.device
        # 0. source-side form: final edges of my sources; all ranks must agree to use it.  One small all_gather carries
        #    the "declined" flag and the edge count of every rank.
        #    The build step is sharded too: each rank computes the minimizer keys of its own nodes only, and the per-node key
        #    arrays (8 bytes per node) are all-gathered in place before every rank sorts them into its copy of the entry array.
        t_keys = time.perf_counter()
        chunk = shard_chunk(self.n, nr)
        karr = be.node_keys(b[r], b[r + 1], nr * chunk) if self.shard_keys else None
        if karr is not None:
            import torch
            # In place: rank r's slice already sits at its position of the engine's (slack-padded) array, the other slices arrive next to
            # it -- no staging tensors, no copy back.  Whether the backend takes aliased buffers is decided from the backend's NAME, the
            # same on every rank -- never from an exception one rank might be alone in raising while its peers are already inside the
            # collective: gloo gets one staging tensor, kept between steps.
            in_place = getattr(dist, "get_backend", lambda: "nccl")() != "gloo"
            for t in karr:
                if in_place:
                    dist.all_gather_into_tensor(t[:nr * chunk], t[r * chunk:(r + 1) * chunk])
                    continue
                full = getattr(self, "_keys_full", None)
                if full is None or full.shape[0] != nr * chunk or full.dtype != t.dtype or full.device != t.device:
                    full = self._keys_full = torch.empty(nr * chunk, dtype=t.dtype, device=dev)
                dist.all_gather_into_tensor(full, t[r * chunk:(r + 1) * chunk].clone())
                t.copy_(full)
        ms_keys = (time.perf_counter() - t_keys) * 1e3
        if self.bucket_sharded and karr is not None:
            done = self._step_bucket_sharded(ms_keys, collect_stats)
            if done is not None:
                return done
        self.form_used = "replicated"
        # The rank's source range in `pieces` consecutive pieces (the first one sorts the gathered keys into the entry array, the
        # others reuse it): the edges of a piece travel to rank 0 while the next piece is probed.  Per piece one small all_gather
        # carries every rank's "declined" flag and edge count.
        pieces = self.pieces                     # (without the sharded key pass the first piece computes every node's keys itself: keys_shared 0, then 2)
        pb = piece_bounds(b[r], b[r + 1], pieces)
        st, pending, declined = None, [], False
        t_wait = 0.0
        for k in range(pieces):
            mine = be.build_range(pb[k], pb[k + 1], collect_stats, keys_shared=(1 if karr is not None else 0) if k == 0 else 2)
            meta = torch.tensor([0 if mine is not None else 1, 0 if mine is None else int(mine.shape[0])], dtype=torch.int64, device=dev)
            allmeta = torch.empty(2 * nr, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allmeta, meta)
            allmeta = allmeta.cpu().view(nr, 2)
            if int(allmeta[:, 0].max()) != 0:
                declined = True
                break
            ps = dict(be.stats)
            if st is None:
                st = ps
            else:
                for kk, v in ps.items():
                    if kk.startswith("ms_") or kk in ("windows_probed", "slots_scanned", "raw_overlaps", "records", "transitive_compares", "generic_sources",
                                                       "big_sources", "deferred_sources", "edges"):
                        st[kk] = st.get(kk, 0) + v
            pending.append(self._gather_start(mine, [int(x) for x in allmeta[:, 1]], single=(pieces == 1)))
        if not declined:
            t2 = time.perf_counter()
            self.edges = self._gather_finish(pending)
            be.sync()
            st["ms_keys_shared"] = ms_keys if karr is not None else 0.0       # host clock: key pass of my nodes + all-gather (enqueue; the build waits for it on the stream)
            st["edges"] = self.total_edges
            st["ms_exchange"] = (time.perf_counter() - t2) * 1e3               # what was left of the gathers after the last piece's probe
            return self._finish(st, collect_stats)
        if pending:                                                            # a rank declined (capacity case): all take the general form;
            self._gather_finish(pending)                                       # what is already on its way is received and dropped
        del pending
        # 1. discover + order by target
        rdst, rval = be.discover_sorted(b[r], b[r + 1], collect_stats)
        st = dict(be.stats)
        t0 = time.perf_counter()
        # 2. exchange: the records for owner q are the contiguous slice [cut[q], cut[q+1]) of the sorted arrays
        cuts = torch.searchsorted(rdst, torch.tensor(b, dtype=rdst.dtype, device=dev))
        sc = (cuts[1:] - cuts[:-1]).to(torch.int64)
        rcnt = torch.empty_like(sc)
        dist.all_to_all_single(rcnt, sc)
        sc_l, rc_l = [int(x) for x in sc.cpu()], [int(x) for x in rcnt.cpu()]
        tot = sum(rc_l)
        recv_d = torch.empty(tot, dtype=rdst.dtype, device=dev)
        recv_v = torch.empty(tot, dtype=rval.dtype, device=dev)
        # send buffers are copied out of engine-owned memory into tensors the collective backend manages
        dist.all_to_all_single(recv_d, rdst.clone(), output_split_sizes=rc_l, input_split_sizes=sc_l)
        dist.all_to_all_single(recv_v, rval.clone(), output_split_sizes=rc_l, input_split_sizes=sc_l)
        be.sync()
        t1 = time.perf_counter()
        # 3. reduce the owned targets
        mine = be.reduce(recv_d, recv_v, b[r], b[r + 1], collect_stats)
        st.update({k: be.stats[k] for k in be.stats if k.startswith("ms_") or k.startswith("transitive") or k == "max_in_records"})
        t2 = time.perf_counter()
        # 4. gather the per-rank edge lists and 5. order them
        self.edges = self._gather(mine, ordered=False)
        be.sync()
        t3 = time.perf_counter()
        st["edges"] = self.total_edges
        st["ms_exchange"] = (t1 - t0) * 1e3 + (t3 - t2) * 1e3
        return self._finish(st, collect_stats)

    # ---- the variable-length exchanges of the bucket-sharded form ----
    def _agree(self, flag):
        """does ANY rank raise `flag`?  (one small all_reduce: every rank takes the same branch)"""
        import torch
        t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=self.be.device)
        self.dist.all_reduce(t)
        return int(t.item()) != 0

    def _all_to_all_rows(self, rows, cnt, off, name):
        """rows [cap, 3] int32 with the segment for rank q at rows[off[q] : off[q] + cnt[q]] -> what the ranks sent to me, rank by rank"""
        import torch
        dist, dev = self.dist, self.be.device
        send = torch.cat([rows[off[q]:off[q] + cnt[q]] for q in range(self.world)], dim=0).contiguous() if sum(cnt) else torch.empty((0, 3), dtype=torch.int32, device=dev)
        sc = torch.tensor(cnt, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc)
        rc_l = [int(x) for x in rc.cpu()]
        recv = torch.empty((sum(rc_l), 3), dtype=torch.int32, device=dev)
        dist.all_to_all_single(recv.view(-1), send.view(-1), output_split_sizes=[3 * x for x in rc_l], input_split_sizes=[3 * x for x in cnt])
        self.exchange_bytes[name] = 12 * (sum(cnt) - cnt[self.rank])
        return recv

    def _all_gather_rows(self, mine, width, name):
        """mine: [k] or [k, width] int32 of any length per rank -> the concatenation over the ranks"""
        import torch
        dist, dev, nr = self.dist, self.be.device, self.world
        k = int(mine.shape[0])
        allk = torch.empty(nr, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allk, torch.tensor([k], dtype=torch.int64, device=dev))
        ks = [int(x) for x in allk.cpu()]
        mx = max(max(ks), 1)
        local = torch.zeros((mx, width), dtype=torch.int32, device=dev)
        if k:
            local[:k] = mine.reshape(k, width)
        parts = torch.empty((nr, mx, width), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(parts.view(-1), local.view(-1))
        self.exchange_bytes[name] = 4 * width * k * (nr - 1)
        out = torch.cat([parts[q, :ks[q]] for q in range(nr)], dim=0).contiguous()
        return out if width > 1 else out.view(-1)

    def _step_bucket_sharded(self, ms_keys, collect_stats):
        """The build with the index sharded by seed bucket, keys already all-gathered; -> (edges, stats), or None when a rank's engine
        declined a phase (all ranks then continue in the replicated form)."""
        import torch
        be, r, nr, b = self.be, self.rank, self.world, self.bounds
        t0 = time.perf_counter()
        idx = be.shard_index(r, nr)
        if self._agree(idx is None):
            return None
        desc, cnt, off = idx
        recv = self._all_to_all_rows(desc, cnt, off, "descriptors")
        pend = be.shard_join(recv)
        if self._agree(pend is None):
            return None
        pend_all = self._all_gather_rows(pend, 1, "pending")
        small = be.shard_small_keys(pend_all)
        small_all = self._all_gather_rows(small, 3, "small_keys")
        eout, ecnt, eoff = be.shard_resolve(small_all, nr)
        ein = self._all_to_all_rows(eout, ecnt, eoff, "edges")
        mine = be.shard_place(ein, b[r], b[r + 1])
        st = dict(be.stats)
        st["ms_shard_wall"] = (time.perf_counter() - t0) * 1e3
        # edge lists of the ascending source ranges -> rank 0
        meta = torch.tensor([int(mine.shape[0])], dtype=torch.int64, device=be.device)
        allmeta = torch.empty(nr, dtype=torch.int64, device=be.device)
        self.dist.all_gather_into_tensor(allmeta, meta)
        t2 = time.perf_counter()
        self.edges = self._gather_finish([self._gather_start(mine, [int(x) for x in allmeta.cpu()], single=True)])
        be.sync()
        st["ms_keys_shared"] = ms_keys
        st["edges"] = self.total_edges
        st["ms_exchange"] = (time.perf_counter() - t2) * 1e3
        st["exchange_bytes"] = dict(self.exchange_bytes)
        self.form_used = "bucket_sharded"
        return self._finish(st, collect_stats)

    def _gather_start(self, mine, counts, single=False):
        """Start moving one piece of every rank's edge list to rank 0 (or to every rank): -> (works, buffers, counts, send buffer).
        Rank 0 receives every rank's piece with its exact length as point-to-point transfers (one per peer, each over that peer's own
        xGMI link), POSTED HERE -- at the program point where the peers post their sends: a send that finds no receive blocks what
        the sender's communicator is asked to do next (the small all_gather of the following piece), so the receives must not wait
        for `_gather_finish`.  single (the rank builds its range in one piece: the driver's default): the counts of this piece are
        the final layout, and the receives land at their offsets of ONE preallocated list -- no padding to the longest piece, no
        concatenation afterwards (round 2 copied the 1.1 GB list once more on rank 0).  Several pieces: one exact-length buffer per
        (peer, piece), concatenated at the end (the final offsets are unknown until the last piece is built)."""
        import torch
        dist, nr, dev = self.dist, self.world, self.be.device
        m = int(mine.shape[0])
        if self.replicate:
            mx = max(max(counts), 1)
            local = torch.zeros((mx, 3), dtype=torch.int32, device=dev)     # also the copy out of the engine's buffer, which the next piece reuses
            if m:
                local[:m] = mine
            parts = torch.empty((nr, mx, 3), dtype=torch.int32, device=dev)
            work = dist.all_gather_into_tensor(parts.view(-1), local.view(-1), async_op=True)
            return [work], parts, counts, local
        # one batch per rank and piece (ncclGroupStart / End under torch's NCCL backend): rank 0's receives run side by side, one per link
        if self.rank != 0:
            local = mine.clone() if m else torch.empty((0, 3), dtype=torch.int32, device=dev)   # the engine's edge buffer is reused by the next piece
            return (list(dist.batch_isend_irecv([dist.P2POp(dist.isend, local.view(-1), 0)])) if m else []), None, counts, local
        ops = []
        if single:
            out = torch.empty((sum(counts), 3), dtype=torch.int32, device=dev)
            off = 0
            parts = []
            for q in range(nr):
                dst = out[off:off + counts[q]]
                if q == 0:
                    dst.copy_(mine)
                elif counts[q]:
                    ops.append(dist.P2POp(dist.irecv, dst.view(-1), q))
                parts.append(dst)
                off += counts[q]
            return (list(dist.batch_isend_irecv(ops)) if ops else []), parts, counts, out
        parts = [mine.clone() if m else torch.empty((0, 3), dtype=torch.int32, device=dev)]
        for q in range(1, nr):
            buf = torch.empty((counts[q], 3), dtype=torch.int32, device=dev)
            if counts[q]:
                ops.append(dist.P2POp(dist.irecv, buf.view(-1), q))
            parts.append(buf)
        return (list(dist.batch_isend_irecv(ops)) if ops else []), parts, counts, None

    def _gather_finish(self, pending):
        """Wait for the pieces; the complete list in (src, dst) order: rank by rank, piece by piece (ascending source ranges)."""
        import torch
        nr, dev = self.world, self.be.device
        for w in pending:
            for x in w[0]:
                x.wait()
        self.total_edges = sum(sum(w[2]) for w in pending)
        if self.replicate:
            return torch.cat([w[1][q][:w[2][q]] for q in range(nr) for w in pending], dim=0).contiguous()
        if self.rank != 0:
            return torch.empty((0, 3), dtype=torch.int32, device=dev)
        if len(pending) == 1 and pending[0][3] is not None:
            return pending[0][3]                            # received in place
        return torch.cat([w[1][q] for q in range(nr) for w in pending], dim=0).contiguous()

    def _gather(self, mine, ordered, counts=None):
        """Edge lists of all ranks (padded to the longest) -> the complete list on rank 0 (or on every rank);
        `ordered`: the rank lists are consecutive (src, dst) runs, no sort needed."""
        import torch
        dist, be, nr = self.dist, self.be, self.world
        dev = be.device
        m = int(mine.shape[0])
        if counts is None:
            allm = torch.empty(nr, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allm, torch.tensor([m], dtype=torch.int64, device=dev))
            counts = [int(x) for x in allm.cpu()]
        self.total_edges = sum(counts)
        mx = max(max(counts), 1)
        local = torch.zeros((mx, 3), dtype=torch.int32, device=dev)
        if m:
            local[:m] = mine
        if self.replicate:
            gathered = torch.empty((nr, mx, 3), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(gathered.view(-1), local.view(-1))
        else:
            parts = [torch.empty((mx, 3), dtype=torch.int32, device=dev) for _ in range(nr)] if self.rank == 0 else None
            dist.gather(local, parts, dst=0)
            if self.rank != 0:
                return torch.empty((0, 3), dtype=torch.int32, device=dev)
            gathered = parts
        packed = torch.cat([gathered[q][:counts[q]] for q in range(nr)], dim=0).contiguous()
        return packed if ordered else be.sort_edges(packed)

    def _finish(self, st, collect_stats):
        import torch
        if collect_stats:   # whole-job counters for the roofline bookkeeping
            keys = ["windows_probed", "slots_scanned", "raw_overlaps", "records", "transitive_listed", "transitive_compares",
                    "transitive_removed"]
            t = torch.tensor([int(st.get(kk, 0)) for kk in keys], dtype=torch.int64, device=self.be.device)
            self.dist.all_reduce(t)
            for kk, v in zip(keys, t.cpu().tolist()):
                st[kk] = int(v)
        return st["edges"], st

    def edges_numpy(self):
        return self.edges.cpu().numpy().astype(np.int32).copy()


class ShardedSupplement:
    """The approximate supplement on N ranks (alga_pkb_shard_*, include/alga_amd.h; SURVEY.md section 8(e): the k-mer groups are dealt out by
    hash).  Every rank holds the node set and -- `run` broadcasts it from rank 0 -- the complete exact graph; per round a rank joins its own
    groups, the additions of all ranks are all-gathered (variable length: sizes first, then one padded all_gather) and every rank merges them
    all.  The graphs stay identical on all ranks and identical to the one-GPU supplement's.
    backend: pkb_begin(edges [m, 3] int32, rank, world), pkb_round() -> int64[k] addition keys, pkb_merge(int64[K]), pkb_end() -> edges [m', 3]
    (HipBackend below; tests/test_multigpu_gloo.py drives this class over gloo with a stand-in)."""

    def __init__(self, backend, rank, world, dist, rounds=4):
        self.be, self.rank, self.world, self.dist, self.rounds = backend, rank, world, dist, rounds
        self.exchange_bytes = []

    def _bcast_edges(self, edges):
        """the exact graph, complete on rank 0 (the gather of the build), to every rank"""
        import torch
        dev = self.be.device
        k = torch.tensor([int(edges.shape[0]) if self.rank == 0 else 0], dtype=torch.int64, device=dev)
        self.dist.broadcast(k, src=0)
        m = int(k.item())
        buf = edges.contiguous() if self.rank == 0 else torch.empty((m, 3), dtype=torch.int32, device=dev)
        if m:
            self.dist.broadcast(buf.view(-1), src=0)
        return buf

    def _all_gather_keys(self, mine):
        import torch
        dist, dev, nr = self.dist, self.be.device, self.world
        k = int(mine.shape[0])
        allk = torch.empty(nr, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allk, torch.tensor([k], dtype=torch.int64, device=dev))
        ks = [int(x) for x in allk.cpu()]
        mx = max(max(ks), 1)
        local = torch.zeros(mx, dtype=torch.int64, device=dev)
        if k:
            local[:k] = mine
        parts = torch.empty((nr, mx), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(parts.view(-1), local)
        self.exchange_bytes.append(8 * k * (nr - 1))
        return torch.cat([parts[q, :ks[q]] for q in range(nr)], dim=0).contiguous()

    def run(self, edges):
        """edges: the exact graph [m, 3] int32 on rank 0 (anything on the others) -> the post-supplement graph, on EVERY rank"""
        if self.world > 1:
            edges = self._bcast_edges(edges)
        self.be.pkb_begin(edges, self.rank, self.world)
        for _ in range(self.rounds):
            mine = self.be.pkb_round()
            allk = self._all_gather_keys(mine) if self.world > 1 else mine
            self.be.pkb_merge(allk)
        return self.be.pkb_end()


def edges_digest(e, chunk=1 << 25):
    """[count, position-weighted checksum] of an edge list [m, 3] (device tensor) as a device int64[2].  int64 wrap-around arithmetic:
    depends on the ORDER of the list, not only on its content -- two lists with equal digests are the same bytes but for a 2^-64 accident.
    In chunks, so that a 92 M-edge list costs 1 GB of temporaries, not 4."""
    import torch
    d = torch.zeros(2, dtype=torch.int64, device=e.device)
    k = int(e.shape[0])
    d[0] = k
    for s0 in range(0, k, chunk):
        e64 = e[s0:s0 + chunk].to(torch.int64)
        w = torch.arange(s0 + 1, s0 + int(e64.shape[0]) + 1, dtype=torch.int64, device=e.device) | 1
        d[1] += (((e64[:, 0] * 1000003 + e64[:, 1]) * 10007 + e64[:, 2]) * w).sum()
        del e64, w
    return d


def graph_digest(runner):
    """One step -> (edges, [count, position-weighted checksum] of the complete graph on rank 0, as every rank sees them)."""
    import torch
    m, _ = runner.step()
    e, dev = runner.edges, runner.be.device
    d = torch.zeros(2, dtype=torch.int64, device=dev)
    k = int(e.shape[0])
    if runner.rank == 0 and k:
        d = edges_digest(e).to(dev)
    if runner.world > 1:
        alld = torch.empty(2 * runner.world, dtype=torch.int64, device=dev)
        runner.dist.all_gather_into_tensor(alld, d)
        d = alld[:2]
    return int(m), [int(x) for x in d.cpu()]


def validated_runner(backend, rank, world, dist, plain=None, **fast_kw):
    """The driver to time at N > 1 -> (runner, {"form", "validated"}).  ShardedPrefSuf's defaults are its plainest form (every rank
    computes all keys, one piece per rank).  The faster one -- keys of the own nodes only + in-place key all-gather, the source range in
    pieces whose edge transfers overlap the next piece's probe -- is returned only after it has reproduced, in THIS process group and
    over its real transport, the plain form's complete graph on rank 0 byte for byte (count + position-weighted checksum); on any
    difference, or when a rank's collective raises, all ranks stay with the plain form.  Two steps (one per form) are spent on it."""
    import torch
    plain = plain if plain is not None else ShardedPrefSuf(backend, rank, world, dist)
    form = {"form": "plain (all keys on every rank, one piece per rank)", "validated": None}
    if world <= 1:
        return plain, form
    # Up to four ranks every rank computes all keys itself: the build then goes through the PILE path on every rank (round 5: k_pile_probe over
    # the rank's id range, include/alga_amd.h option pile_range) -- its key pass makes target keys alone (1.5 ms at the north-star size, less than
    # the all-gather of the other ranks' keys costs), and a rank's compute is 16.0 / 13.9 ms at two / four ranks against 23.4 / 14.8 ms through the
    # pairwise kernels (tools/emulate_rank.py).  From five ranks on the index every rank must hold (9.7 ms through the piles)
    # outweighs the probe it saves: keys of the own nodes + all-gather, pairwise kernels (10.1 against 12.5 ms at eight).
    kw = dict(shard_keys=world > 4, pieces=None)
    kw.update(fast_kw)
    # No exception handling around the collectives: a rank that raised alone inside a step would leave its peers in a collective it
    # never joins -- an error there ends the job (the process group's timeout sees to the peers).  Plain or fast is
    # decided from the all-reduced comparison of the two digests alone.
    fast = ShardedPrefSuf(backend, rank, world, dist, **kw)
    m_plain, d_plain = graph_digest(plain)
    m_fast, d_fast = graph_digest(fast)
    ok = int(m_plain == m_fast and d_plain == d_fast and d_plain[0] == m_plain)
    why = "" if ok else "graphs differ: plain %s / %d edges, sharded %s / %d edges" % (d_plain, m_plain, d_fast, m_fast)
    t = torch.tensor([ok], dtype=torch.int64, device=backend.device)
    dist.all_reduce(t)
    if int(t.item()) == world:
        what = "keys of own nodes + in-place key all-gather" if fast.shard_keys else "all keys on every rank (the rank's sources through the pile path where the build keeps it)"
        if getattr(fast, "bucket_sharded", False):
            what = "index sharded by seed bucket (%s)" % ("alga_shard_*: descriptors to the bucket's owner, per-target reduction there" if fast.form_used == "bucket_sharded"
                                                          else "DECLINED by the engine: replicated form with the gathered keys")
        form = {"form": "%s, %d pieces per rank (edge transfers overlap the next piece's probe)" % (what, fast.pieces),
                "validated": "complete graph on rank 0 byte-identical (count + position-weighted checksum) to the plain form's in this run: %d edges" % m_plain}
        return fast, form
    form["validated"] = "sharded form NOT taken: " + (why or "another rank failed its check")
    return plain, form

