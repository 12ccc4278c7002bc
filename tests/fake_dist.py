"""A thread-based stand-in for torch.distributed (the handful of collectives alga_amd/multigpu.py uses), so that the real
sharded driver + the real HIP backend can run as N 'ranks' inside ONE process on ONE GPU -- TEST INFRASTRUCTURE.
Every rank is a thread holding a FakeDist(rank); collectives rendezvous on a shared barrier."""
import queue
import threading


class _World:
    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        # (src, dst) -> tensors in send order (point-to-point transfers).  Every queue exists before a rank thread runs: a defaultdict made
        # them on first touch, and a sender and a receiver touching a missing key at the same moment could each get a queue of their own -- the
        # message went into one, the receiver waited on the other until its timeout (seen once in a full-suite run, round 5)
        self.mail = {(s, d): queue.Queue() for s in range(n) for d in range(n)}


class ReduceOp:
    SUM, MAX = "sum", "max"


class FakeDist:
    ReduceOp = ReduceOp

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    @staticmethod
    def _sync():
        # every rank thread works on its own stream: what one thread enqueued must be complete before another thread's stream
        # reads it (and a reader must be done before the writer's buffers go back to the allocator)
        import torch
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()

    def _exchange(self, obj):
        self._sync()
        self.w.slots[self.rank] = obj
        self.w.barrier.wait()
        got = list(self.w.slots)
        self.w.barrier.wait()
        return got

    class _Done:
        def wait(self):
            return True

    def all_gather_into_tensor(self, out, inp, async_op=False):
        import torch
        parts = self._exchange(inp.detach().clone())
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).reshape(out.shape))
        self._sync()
        return self._Done() if async_op else None

    def gather(self, tensor, gather_list=None, dst=0, async_op=False):
        parts = self._exchange(tensor.detach().clone())
        if self.rank == dst:
            for g, p in zip(gather_list, parts):
                g.copy_(p)
        self._sync()
        return self._Done() if async_op else None

    def isend(self, tensor, dst):
        c = tensor.detach().clone()                          # the copy is enqueued on the sender's stream ...
        self._sync()                                         # ... and complete before the receiver (another thread, another stream) may read it
        self.w.mail[(self.rank, dst)].put(c)
        return self._Done()

    class _Recv:
        def __init__(self, outer, tensor, src):
            self.o, self.t, self.src = outer, tensor, src

        def wait(self):
            got = self.o.w.mail[(self.src, self.o.rank)].get(timeout=300)
            self.t.copy_(got.reshape(self.t.shape))
            self.o._sync()
            return True

    def irecv(self, tensor, src):
        return self._Recv(self, tensor, src)

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    @staticmethod
    def batch_isend_irecv(ops):
        return [o.op(o.tensor, o.peer) for o in ops]

    def barrier(self):
        self._exchange(None)

    def broadcast(self, tensor, src=0):
        parts = self._exchange(tensor.detach().clone() if self.rank == src else None)
        if self.rank != src:
            tensor.copy_(parts[src].reshape(tensor.shape))
        self._sync()

    def all_reduce(self, t, op=ReduceOp.SUM):
        import torch
        parts = self._exchange(t.detach().clone())
        st = torch.stack(parts)
        t.copy_(st.max(dim=0).values if op == ReduceOp.MAX else st.sum(dim=0))
        self._sync()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        import torch
        n = self.w.n
        if input_split_sizes is None:
            k = inp.shape[0] // n
            input_split_sizes = [k] * n
        offs = [0]
        for s in input_split_sizes:
            offs.append(offs[-1] + s)
        mine = [inp[offs[q]:offs[q + 1]].detach().clone() for q in range(n)]
        allp = self._exchange(mine)
        recv = [allp[r][self.rank] for r in range(n)]
        out.copy_(torch.cat(recv) if recv else out)
        self._sync()


def run_ranks(n, fn):
    """fn(rank, dist) in n threads; returns the list of results, re-raises the first exception."""
    world = _World(n)
    res, err = [None] * n, []

    def work(r):
        try:
            res[r] = fn(r, FakeDist(world, r))
        except BaseException as e:      # noqa: BLE001 -- surfaced below; a dead rank would deadlock the others
            err.append(e)
            world.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return res
