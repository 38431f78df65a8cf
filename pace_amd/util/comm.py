"""Point-to-point transports for the halo exchange.

The reference's ``HaloUpdater`` needs exactly ``Isend`` / ``Irecv`` of contiguous 1-D buffers plus
``Get_rank`` / ``Get_size`` (util/pace/util/comm.py:14-73, halo_updater.py:241-267).  Two transports provide that
over device tensors:

* ``TorchDistComm``  -- one process per GPU, ``torch.distributed`` (backend "nccl" is RCCL over xGMI; "gloo" is used by
  the CPU tests).  ``exchange`` posts all sends and receives of one halo update as ONE ``batch_isend_irecv`` group,
  so RCCL sees a single grouped launch per update and the transfers run on RCCL's stream while the compute stream
  keeps launching kernels until ``wait``.
* ``ThreadComm``     -- several tiles inside one process (one Python thread per tile, e.g. all six tiles of a small
  cubed sphere on a single 288 GB device).  A send enqueues a device-side copy; a receive copies it out.
"""
import threading
from collections import defaultdict, deque

import torch


class Request:
    def __init__(self, fn=None):
        self._fn = fn

    def wait(self):
        if self._fn is not None:
            self._fn()
            self._fn = None


class TorchDistComm:
    def __init__(self, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self._dist = dist
        self._group = group

    def Get_rank(self):
        return self._dist.get_rank(self._group)

    def Get_size(self):
        return self._dist.get_world_size(self._group)

    def barrier(self):
        self._dist.barrier(self._group)

    def exchange(self, sends, recvs, tag=0):
        """sends / recvs: lists of (1-D tensor, peer rank).  Returns a Request."""
        dist = self._dist
        ops = [dist.P2POp(dist.irecv, buf, peer, self._group, tag) for buf, peer in recvs]
        ops += [dist.P2POp(dist.isend, buf, peer, self._group, tag) for buf, peer in sends]
        works = dist.batch_isend_irecv(ops)

        def fin():
            for w in works:
                w.wait()

        return Request(fin)

    def allreduce_min(self, value: float) -> float:
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if self._dist.get_backend(self._group) == "nccl" else "cpu")
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self._group)
        return float(t.item())


class NullComm:
    """util/pace/util/null_comm.py:15-80: the rank of a communicator that has no peers -- what it "receives" is a fill value
    (zero as in the reference's Irecv; halo updates therefore zero the halos).  Lets one tile's operators run alone, as the
    reference's own drop-in test does (tests/main/fv3core/test_dycore_call.py)."""

    def __init__(self, rank, total_ranks, fill_value=0.0):
        self.rank, self.total_ranks, self._fill_value = rank, total_ranks, fill_value

    def __repr__(self):
        return f"NullComm(rank={self.rank}, total_ranks={self.total_ranks})"

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.total_ranks

    def barrier(self):
        return

    def exchange(self, sends, recvs, tag=0):
        def fin():
            for buf, _peer in recvs:
                buf[:] = 0.0  # NullAsyncResult.wait (null_comm.py:11-13)

        return Request(fin)

    def allreduce_min(self, value: float) -> float:
        return value


class LoopbackComm(NullComm):
    """NOT a reference communicator: a lone rank that receives from each peer what it sent to that peer (the cube's messages
    are symmetric in size).  The halos then hold values of the right magnitude -- not the neighbour's, and not in the
    neighbour's orientation -- so that one tile's step can be TIMED alone without the zeros of NullComm turning the state
    into NaNs (tools/dycore_bench.py --single).  Never used for results."""

    def exchange(self, sends, recvs, tag=0):
        sent = {peer: buf for buf, peer in sends}

        def fin():
            for buf, peer in recvs:
                src = sent.get(peer)
                if src is None or src.numel() != buf.numel():
                    buf[:] = 0.0
                else:
                    buf.copy_(src)

        return Request(fin)


class _World:
    """Shared state of the tile threads.  The condition's lock doubles as a run token: a tile thread holds it whenever it
    executes and gives it up only while it waits for a message (``Condition.wait_for`` releases and re-acquires it), so the
    tiles run as coroutines -- one at a time, in message-driven order -- and nothing below (PyTorch, the HIP runtime, the
    library) is ever entered from two threads at once."""

    def __init__(self, n):
        self.n = n
        self.cond = threading.Condition()  # RLock inside: re-entrant
        self.mail = defaultdict(deque)
        self.arrived = 0
        self.generation = 0
        self.failed = False


class ThreadComm:
    def __init__(self, world: _World, rank: int):
        self._world = world
        self._rank = rank

    def Get_rank(self):
        return self._rank

    def Get_size(self):
        return self._world.n

    def _wait(self, predicate, what):
        w = self._world
        if not w.cond.wait_for(lambda: predicate() or w.failed, timeout=600):
            raise TimeoutError(f"tile {self._rank} waiting for {what}")
        if w.failed and not predicate():
            raise RuntimeError(f"tile {self._rank}: another tile failed while this one waited for {what}")

    def barrier(self):
        w = self._world
        with w.cond:
            gen = w.generation
            w.arrived += 1
            if w.arrived == w.n:
                w.arrived = 0
                w.generation += 1
                w.cond.notify_all()
            else:
                self._wait(lambda: w.generation != gen, "barrier")

    def exchange(self, sends, recvs, tag=0):
        w = self._world
        with w.cond:
            for buf, peer in sends:
                w.mail[(self._rank, peer, tag)].append(buf.clone())
            w.cond.notify_all()

        def fin():
            for buf, peer in recvs:
                key = (peer, self._rank, tag)
                with w.cond:
                    self._wait(lambda: len(w.mail[key]) > 0, f"message {key}")
                    data = w.mail[key].popleft()
                buf.copy_(data)

        return Request(fin)


def run_tiles(n, fn):
    """Run ``fn(comm)`` for n tiles on n threads of this process; returns their results in tile order."""
    world = _World(n)
    results, errors = [None] * n, []
    # the current device is a per-thread setting: hand the caller's on to the tile threads (a rank of a multi-GPU run owns
    # device LOCAL_RANK, and a new thread would start on device 0).  Nothing here initialises the GPU.
    device = None
    try:
        import torch

        if torch.cuda.is_initialized():
            device = torch.cuda.current_device()
    except Exception:  # noqa: BLE001
        device = None

    def target(r):
        if device is not None:
            import torch

            torch.cuda.set_device(device)
        with world.cond:  # the run token (see _World)
            try:
                results[r] = fn(ThreadComm(world, r))
            except BaseException as e:  # noqa: BLE001
                import traceback

                errors.append((r, e, traceback.format_exc()))
                world.failed = True
                world.cond.notify_all()

    threads = [threading.Thread(target=target, args=(r,), daemon=True, name=f"tile{r}") for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        r, e, tb = errors[0]
        raise RuntimeError(f"tile {r} failed: {e!r}\n{tb}") from e
    return results
