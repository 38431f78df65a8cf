"""In-process, thread-per-rank implementation of the reference's ``pace.util.Comm`` ABC.

Dev-container tool: lets the six cubed-sphere tile ranks of the reference run as six
Python threads (blocking receives, real all-reduce) so that grid generation, the
baroclinic initial state and halo exchanges execute with true multi-rank semantics
without MPI.  Not part of the product.
"""
import copy
import functools
import threading
from collections import defaultdict, deque

import numpy as np


class _Done:
    def wait(self):
        return None


class _Pending:
    def __init__(self, fn):
        self._fn = fn

    def wait(self):
        return self._fn()


class World:
    def __init__(self, n):
        self.n = n
        self.cond = threading.Condition()
        self.mail = defaultdict(deque)
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.children = {}


class ThreadComm:
    def __init__(self, world, rank):
        self.world = world
        self.rank = rank

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.world.n

    def barrier(self):
        self.world.barrier.wait()

    Barrier = barrier

    def Send(self, sendbuf, dest, tag=0, **kw):
        w = self.world
        with w.cond:
            w.mail[(self.rank, dest, tag)].append(copy.deepcopy(np.asarray(sendbuf)))
            w.cond.notify_all()

    def Isend(self, sendbuf, dest, tag=0, **kw):
        self.Send(sendbuf, dest, tag)
        return _Done()

    def Recv(self, recvbuf, source, tag=0, **kw):
        w = self.world
        key = (source, self.rank, tag)
        with w.cond:
            ok = w.cond.wait_for(lambda: len(w.mail[key]) > 0, timeout=120)
            if not ok:
                raise TimeoutError(f"rank {self.rank} waiting for {key}")
            data = w.mail[key].popleft()
        recvbuf[...] = data.reshape(recvbuf.shape)

    def Irecv(self, recvbuf, source, tag=0, **kw):
        return _Pending(lambda: self.Recv(recvbuf, source, tag))

    def _exchange(self, value):
        w = self.world
        w.slots[self.rank] = value
        w.barrier.wait()
        vals = list(w.slots)
        w.barrier.wait()
        return vals

    def allreduce(self, sendobj, op=None):
        vals = self._exchange(sendobj)
        if op is None:
            op = lambda a, b: a + b  # noqa: E731
        return functools.reduce(op, vals)

    def allgather(self, sendobj):
        return self._exchange(sendobj)

    def bcast(self, value, root=0):
        return copy.deepcopy(self._exchange(value)[root])

    def Split(self, color, key):
        vals = self._exchange((color, key, self.rank))
        members = sorted((k, r) for (c, k, r) in vals if c == color)
        ranks = [r for _, r in members]
        w = self.world
        with w.cond:
            ck = (tuple(ranks), "split")
            if ck not in w.children:
                w.children[ck] = World(len(ranks))
            child = w.children[ck]
        w.barrier.wait()
        return ThreadComm(child, ranks.index(self.rank))

    def Scatter(self, sendbuf, recvbuf, root=0, **kw):
        vals = self._exchange(sendbuf)
        recvbuf[...] = np.asarray(vals[root])[self.rank]

    def Gather(self, sendbuf, recvbuf, root=0, **kw):
        vals = self._exchange(np.asarray(sendbuf))
        if self.rank == root:
            for i, v in enumerate(vals):
                recvbuf[i, ...] = v

    def sendrecv(self, sendbuf, dest, **kw):
        self.Send(sendbuf, dest, tag=-7)
        out = np.empty_like(np.asarray(sendbuf))
        self.Recv(out, dest, tag=-7)
        return out


def run_ranks(n, fn):
    """Run fn(comm) on n threads; returns the list of results (re-raises the first error)."""
    world = World(n)
    results = [None] * n
    errors = []

    def target(r):
        try:
            results[r] = fn(ThreadComm(world, r))
        except BaseException as e:  # noqa: BLE001
            import traceback

            errors.append((r, e, traceback.format_exc()))
            world.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,), daemon=True, name=f"rank{r}") for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        r, e, tb = errors[0]
        raise RuntimeError(f"rank {r} failed: {e!r}\n{tb}") from e
    return results
