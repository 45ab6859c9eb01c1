"""A thread-backed stand-in for torch.distributed -- TEST INFRASTRUCTURE.

The GPU box has ONE GPU, so a real multi-rank RCCL run cannot happen there.  This lets the real driver
(vdjer_amd/shard.py) and the real HIP phase engine run with world_size > 1 on a single device: every rank
is a Python thread with its own vdjx context/stream, and the collectives are tensor copies."""
import threading

import torch


class ThreadDist:
    class ReduceOp:
        MAX, MIN, SUM = "max", "min", "sum"

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def set_rank(self, r):
        self.local.rank = r

    def get_rank(self):
        return self.local.rank

    def get_world_size(self):
        return self.world

    def _deposit(self, x):
        self.slots[self.get_rank()] = x
        self.barrier.wait()

    def _done(self):
        self.barrier.wait()

    def all_reduce(self, t, op="sum"):
        self._deposit(t.clone())
        st = torch.stack(self.slots)
        if op == "max":
            t.copy_(st.max(dim=0).values)
        elif op == "min":
            t.copy_(st.min(dim=0).values)
        else:
            t.copy_(st.sum(dim=0))
        self._done()

    def all_gather(self, outs, t):
        self._deposit(t)
        for i in range(self.world):
            outs[i].copy_(self.slots[i])
        if t.is_cuda:
            torch.cuda.synchronize()
        self._done()

    def all_gather_into_tensor(self, full, t):
        self.all_gather(list(full.split(t.shape[0])), t)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        n = self.world
        if in_splits is None:
            in_splits = [inp.shape[0] // n] * n
            out_splits = [out.shape[0] // n] * n
        self._deposit((inp, in_splits))
        me = self.get_rank()
        dst = list(out.split(out_splits))
        for src in range(n):
            s_in, s_spl = self.slots[src]
            dst[src].copy_(list(s_in.split(s_spl))[me])
        if out.is_cuda:
            torch.cuda.synchronize()
        self._done()
