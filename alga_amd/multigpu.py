"""One process per GPU: sharding of the PrefSuf build across ranks (torch.distributed = RCCL on ROCm).

Layout: the packed node set is replicated on every GPU (36 B/read; 3.6 GB at 100 M nodes, HBM is 288 GB).
  1. discover  rank r probes the SOURCES of its contiguous id range against the full seed table and applies the
               per-source small-overlap cap locally (the cap is per source, so it needs no exchange)
  2. exchange  overlap records go to the rank that owns the TARGET id range          (all_to_all_single)
  3. reduce    per-target transitive reduction of the owned targets                   (local)
  4. gather    edge lists of all ranks                                                 (all_gather, padded)
  5. order     every rank sorts the gathered edges by (src, dst, offset)              (local, HIP)
so the result is byte-identical for every world size.  torch supplies buffers and collectives only; all
compute steps are C-ABI calls into the HIP kernels.
"""
import numpy as np


class ShardedPrefSuf:
    def __init__(self, engine, d_words, d_lens, min_overlap, rsoemo, rank=0, world=1, dist=None):
        self.eng, self.w, self.l = engine, d_words, d_lens
        self.lo, self.rs = int(min_overlap), int(rsoemo)
        self.rank, self.world, self.dist = rank, world, dist
        self.n = int(d_lens.shape[0])
        b = [(self.n * r) // world for r in range(world + 1)]
        b = [x - (x & 1) for x in b[:-1]] + [self.n]          # keep a read and its reverse complement together
        self.bounds = b
        self.last_edges = None                                 # (device pointer, count) of the last step

    def step(self, collect_stats=False):
        """-> (n_edges of the complete graph, stats dict of this rank)."""
        if self.world == 1:
            ptr, m = self.eng.prefsuf_device(self.w, self.l, self.lo, self.rs, collect_stats=collect_stats)
            self.last_edges = (ptr, m)
            return m, self.eng.last_stats()
        return self._step_sharded(collect_stats)

    def _step_sharded(self, collect_stats):
        import time
        import torch
        from .engine import device_view
        dist, eng, r, nr = self.dist, self.eng, self.rank, self.world
        b = self.bounds
        dev = self.w.device
        d, v, k = eng.discover_device(self.w, self.l, self.lo, self.rs, b[r], b[r + 1], collect_stats=collect_stats)
        st = eng.last_stats()
        t0 = time.perf_counter()
        rdst, rval = device_view(d, (k,), dev), device_view(v, (k,), dev, "<i8")
        # owner of a record = rank whose target range holds dst; chunk padding (dst = -1 as int32) is dropped
        bounds_t = torch.tensor(b[1:], dtype=torch.int32, device=dev)
        valid = rdst >= 0
        rdst, rval = rdst[valid], rval[valid]
        owner = torch.searchsorted(bounds_t, rdst, right=True)
        order = torch.argsort(owner, stable=True)
        send_d, send_v = rdst[order].contiguous(), rval[order].contiguous()
        sc = torch.bincount(owner, minlength=nr)[:nr]
        rcnt = torch.empty_like(sc)
        dist.all_to_all_single(rcnt, sc)
        sc_l, rc_l = [int(x) for x in sc.cpu()], [int(x) for x in rcnt.cpu()]
        tot = sum(rc_l)
        recv_d = torch.empty(max(tot, 1), dtype=torch.int32, device=dev)
        recv_v = torch.empty(max(tot, 1), dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv_d[:tot], send_d, output_split_sizes=rc_l, input_split_sizes=sc_l)
        dist.all_to_all_single(recv_v[:tot], send_v, output_split_sizes=rc_l, input_split_sizes=sc_l)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ptr, m = eng.reduce_device(self.w, self.l, self.lo, self.rs, recv_d, recv_v, tot, b[r], b[r + 1],
                                   collect_stats=collect_stats)
        st2 = eng.last_stats()
        t2 = time.perf_counter()
        # gather the per-rank edge lists (padded to the longest), then order them by (src, dst, offset)
        mine = torch.tensor([m], dtype=torch.int64, device=dev)
        allm = torch.empty(nr, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allm, mine)
        ms = [int(x) for x in allm.cpu()]
        mx = max(max(ms), 1)
        local = torch.zeros((mx, 3), dtype=torch.int32, device=dev)
        if m:
            local[:m] = device_view(ptr, (m, 3), dev)
        gathered = torch.empty((nr, mx, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), local.view(-1))
        packed = torch.cat([gathered[q, :ms[q]] for q in range(nr)], dim=0)
        key = (packed[:, 0].to(torch.int64) << 32) | packed[:, 1].to(torch.int64)
        self.edges_sorted = packed[torch.argsort(key)].contiguous()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        m2 = int(self.edges_sorted.shape[0])
        self.last_edges = None
        for key_ in ("ms_group", "ms_reduce", "ms_emit", "transitive_listed", "transitive_compares", "transitive_removed",
                     "max_in_records"):
            st[key_] = st2[key_]
        st["edges"] = m2
        st["ms_exchange"] = (t1 - t0) * 1e3 + (t3 - t2) * 1e3
        if collect_stats:   # whole-job counters for the roofline bookkeeping
            keys = ["nodes_live", "windows_probed", "slots_scanned", "raw_overlaps", "records", "transitive_listed",
                    "transitive_compares", "transitive_removed"]
            t = torch.tensor([st[kk] for kk in keys], dtype=torch.int64, device=dev)
            live = st["nodes_live"]
            dist.all_reduce(t)
            for kk, v in zip(keys, t.cpu().tolist()):
                st[kk] = int(v)
            st["nodes_live"] = live
        return m2, st

    def edges_numpy(self):
        from .engine import device_edges_to_numpy
        if self.last_edges is None:
            return self.edges_sorted.cpu().numpy().copy()
        ptr, m = self.last_edges
        return device_edges_to_numpy(ptr, m)
