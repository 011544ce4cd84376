"""Test infrastructure: N "ranks" as N threads of ONE process, each driving its own model replica on the same GPU, with
torch.distributed's collectives replaced by an in-process sum.  A GPU box allows at most 6 processes on the card, so the
8-rank forms of BASELINE configs[3] / [4] cannot run as 8 gloo processes there; this runs the REAL per-rank code path
(`VFM.plan(process_group=...)`, `VFM.train_step(process_group=...)`: the row sharding, the sum of the batch normalisers,
the chunked statistics / gradient all-reduce, the prior-terms flag) for any N.  What it does not exercise is the
transport: tests/test_gpu_dist.py (2-4 gloo processes) and tests/test_dist_cpu.py do that.

Every rank launches on the same (default) stream: kernels run in launch order, and the barriers of the fake all-reduce
make that order the one a real collective would impose (everybody's contribution before the sum, the sum before anybody
reads it)."""
import threading

import torch


class ThreadGroup:
    def __init__(self, rank, shared):
        self.rank, self.shared = rank, shared


class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.bufs = [None] * world
        self.total = None
        self.n_allreduce = 0
        self.bytes_allreduce = 0


class _Done:
    def wait(self):
        return True


def _all_reduce(tensor, op=None, group=None, async_op=False):
    sh = group.shared
    sh.bufs[group.rank] = tensor
    sh.barrier.wait()
    if group.rank == 0:
        if op is not None and "MAX" in str(op).upper():
            total = torch.stack(list(sh.bufs)).max(dim=0).values
        else:
            total = sh.bufs[0].clone()
            for b in sh.bufs[1:]:                 # rank order: a fixed summation order
                total += b
        sh.total = total
        sh.n_allreduce += 1
        sh.bytes_allreduce += tensor.numel() * tensor.element_size()
    sh.barrier.wait()
    tensor.copy_(sh.total)
    sh.barrier.wait()                             # nobody starts the next collective before everybody has its copy
    return _Done() if async_op else None


def _all_gather(out_list, tensor, group=None, async_op=False):
    sh = group.shared
    sh.bufs[group.rank] = tensor
    sh.barrier.wait()
    for r in range(sh.world):
        out_list[r].copy_(sh.bufs[r])
    sh.barrier.wait()
    return _Done() if async_op else None


def run_ranks(world, fn, monkeypatch):
    """Run fn(rank, group) on `world` threads; returns (list of results by rank, the shared record)."""
    import torch.distributed as dist
    real = (dist.all_reduce, dist.get_rank, dist.get_world_size, dist.all_gather)
    monkeypatch.setattr(dist, "all_gather", lambda out, t, group=None, async_op=False:
                        _all_gather(out, t, group, async_op) if isinstance(group, ThreadGroup) else real[3](out, t, group=group, async_op=async_op))
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None, async_op=False:
                        _all_reduce(t, op, group, async_op) if isinstance(group, ThreadGroup) else real[0](t, op=op, group=group, async_op=async_op))
    monkeypatch.setattr(dist, "get_rank", lambda group=None: group.rank if isinstance(group, ThreadGroup) else real[1](group))
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: group.shared.world if isinstance(group, ThreadGroup) else real[2](group))
    sh = _Shared(world)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            torch.cuda.set_device(0)
            out[r] = fn(r, ThreadGroup(r, sh))
        except BaseException as e:               # noqa: BLE001  (re-raised in the caller's thread)
            err[r] = e
            sh.barrier.abort()
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    torch.cuda.synchronize()
    return out, sh
