"""Launcher-side plumbing of a GROUP BY over several partitions, one process per GPU.

The reference splits a GROUP BY into per-partition `PartialGroupByExpression`s and
merges their rows in `GroupByMergeExpression` (sql/statements/select/groupby.cc:231-714;
fan-out in server/sql/scheduler.cc:117-162).  Here the exchange step itself --
bucketing the group records by owner, moving them between GPUs, merging them in rank
order -- is `evql_query_exchange` behind the C ABI (csrc/exchange.cc).  Python only

  * assigns partitions to ranks (`partitions_for_rank`),
  * hands the ncclUniqueId of rank 0 to the other ranks through the launcher's
    rendezvous (`torch.distributed`, any backend) for the built-in RCCL transport, and
  * for rehearsals without one GPU per rank (`gloo`: several ranks on one GPU, or the
    CPU tests) supplies the two transport callbacks on top of torch.distributed,
    staging device memory through the host.
"""
import ctypes as C

import torch
import torch.distributed as dist


def partitions_for_rank(n_partitions, rank, world):
    """contiguous block assignment of table partitions to ranks"""
    per = (n_partitions + world - 1) // world
    lo = min(rank * per, n_partitions)
    return list(range(lo, min(lo + per, n_partitions)))


def _words(ptr, nwords, device):
    """int64 tensor view of `nwords` 8-byte words at raw address `ptr`"""
    nwords = int(nwords)
    if nwords == 0:
        return torch.zeros(0, dtype=torch.int64, device=device)
    if device == "cpu":
        buf = (C.c_int64 * nwords).from_address(int(ptr))
        return torch.frombuffer(buf, dtype=torch.int64)

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = dict(shape=(nwords,), typestr="<i8", data=(int(ptr), False),
                                      version=2)
    return torch.as_tensor(h, device=device)


class GlooTransport:
    """evql_transport_t callbacks over torch.distributed point-to-point transfers
    (gloo moves host memory: device words are staged through the host)"""

    def __init__(self, group=None, device="cuda"):
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather(self, send):
        t = torch.tensor([int(v) & 0x7FFFFFFFFFFFFFFF if v >= 1 << 63 else int(v) for v in send],
                         dtype=torch.int64)
        parts = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return [int(v) for p in parts for v in p.tolist()]

    def all_to_all(self, d_send, send_counts, d_recv, recv_counts, stream):
        if self.device != "cpu":
            torch.cuda.synchronize()
        send = _words(d_send, sum(send_counts), self.device)
        recv = _words(d_recv, sum(recv_counts), self.device)
        host_send = send.cpu() if self.device != "cpu" else send
        outs, reqs = [], []
        soff = roff = 0
        for r in range(self.world):
            sc, rc = int(send_counts[r]), int(recv_counts[r])
            piece = host_send[soff:soff + sc]
            if r == self.rank:
                outs.append((roff, piece.clone()))
            else:
                if sc:
                    reqs.append(dist.isend(piece.contiguous(), r, group=self.group))
                if rc:
                    buf = torch.zeros(rc, dtype=torch.int64)
                    reqs.append(dist.irecv(buf, r, group=self.group))
                    outs.append((roff, buf))
            soff += sc
            roff += rc
        for q in reqs:
            q.wait()
        for off, buf in outs:
            if buf.numel():
                recv[off:off + buf.numel()].copy_(buf)
        if self.device != "cpu":
            torch.cuda.synchronize()


def make_exchange(ctx, group=None):
    """the evql Exchange of this rank: RCCL when the process group is `nccl` (one GPU
    per rank; the id travels through the group), the gloo callbacks otherwise"""
    import eventql_amd as E
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        box = [E.Exchange.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return E.Exchange.rccl(ctx, world, rank, box[0])
    tr = GlooTransport(group, "cuda")
    x = E.Exchange.custom(ctx, world, rank, tr.all_gather, tr.all_to_all, name="gloo (host staged)")
    x._transport = tr
    return x
