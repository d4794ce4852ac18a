"""Partial-aggregate exchange across ranks (one process per GPU).

The reference splits a GROUP BY into per-partition `PartialGroupByExpression`s
and merges their (key, saved state) rows in `GroupByMergeExpression`
(sql/statements/select/groupby.cc:438-472, 553-615; fan-out in
server/sql/scheduler.cc:117-162).  Here the partitions live on different GPUs
and the rows travel as dense group records `[kind, identity, (first_row),
state words...]` (int64 words) over torch.distributed -- RCCL on GPUs, gloo in
the CPU tests:

  low cardinality   all_gather of every rank's records, each rank merges the
                    others' (`exchange_all_gather`)
  high cardinality  records are bucketed by `identity % world` and exchanged
                    with all_to_all so that every rank owns a disjoint key range
                    (`exchange_all_to_all`)
"""
import torch
import torch.distributed as dist


def partitions_for_rank(n_partitions, rank, world):
    """contiguous block assignment of table partitions to ranks"""
    per = (n_partitions + world - 1) // world
    lo = min(rank * per, n_partitions)
    return list(range(lo, min(lo + per, n_partitions)))


def _staged_on_host(records, group):
    """gloo moves host memory: device records are staged through the host (used
    to rehearse the multi-rank path on a single GPU; RCCL needs one GPU per rank)"""
    return records.device.type == "cuda" and dist.get_backend(group) == "gloo"


def exchange_buffers(record_words, max_groups, world, device):
    """persistent (send, recv) buffers for exchange_all_gather: word 0 of a rank's
    slice = its record count, the records follow -- export straight into
    send[1:] (data_ptr() + 8) and pass the pair as `buffers`"""
    width = max_groups * record_words + 1
    return (torch.zeros(width, dtype=torch.int64, device=device),
            torch.zeros(world * width, dtype=torch.int64, device=device))


def exchange_all_gather(records, n, record_words, max_groups, group=None, buffers=None):
    """records: int64 tensor holding >= n*record_words words (this rank's dense
    records), or None when they were exported into buffers[0][1:] already.
    Returns [(tensor_view, count)] for every rank, in rank order."""
    if records is not None and _staged_on_host(records, group):
        parts = exchange_all_gather(records[:n * record_words].cpu(), n, record_words,
                                    max_groups, group)
        return [(t.to(records.device), c) for t, c in parts]
    world = dist.get_world_size(group)
    width = max_groups * record_words + 1
    if n > max_groups:
        raise ValueError("more groups (%d) than the exchange buffer holds (%d)" % (n, max_groups))
    if buffers is not None:
        send, recv = buffers
        send[0] = n
        if records is not None:
            send[1:1 + n * record_words] = records[:n * record_words]
    else:
        send = torch.zeros(width, dtype=torch.int64, device=records.device)
        send[0] = n
        send[1:1 + n * record_words] = records[:n * record_words]
        recv = torch.zeros(world * width, dtype=torch.int64, device=records.device)
    if send.device.type == "cuda":
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        parts = [torch.zeros(width, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.cat(parts)
    out = []
    recv = recv.view(world, width)
    counts = recv[:, 0].tolist()  # one device->host sync for all ranks
    for r in range(world):
        cnt = int(counts[r])
        out.append((recv[r, 1:1 + cnt * record_words], cnt))
    return out


def gather_foreign(parts, rank, record_words):
    """the other ranks' records of an exchange_all_gather as ONE contiguous tensor
    (a single merge launch instead of one per rank); (tensor | None, count)"""
    others = [t for r, (t, c) in enumerate(parts) if r != rank and c]
    if not others:
        return None, 0
    cat = others[0].contiguous() if len(others) == 1 else torch.cat(others)
    if cat.device.type == "cuda":
        # the merge kernel runs on the library's own stream
        torch.cuda.current_stream(cat.device).synchronize()
    return cat, cat.numel() // record_words


def bucket_by_owner(records, n, record_words, world):
    """stable partition of dense records by owner rank = identity % world
    (records with kind != 0, i.e. the sentinel / NULL key, go to rank 0)"""
    rec = records[:n * record_words].view(n, record_words)
    ident = rec[:, 1]
    owner = torch.remainder(ident, world)
    owner = torch.where(rec[:, 0] != 0, torch.zeros_like(owner), owner)
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=world)
    return rec[order].contiguous().view(-1), counts


def exchange_all_to_all(records, n, record_words, group=None):
    """hash-partitioned exchange: returns (tensor, count) of the records this rank
    owns, received from all ranks (its own included)"""
    if _staged_on_host(records, group):
        recv, cnt = exchange_all_to_all(records[:n * record_words].cpu(), n, record_words, group)
        return recv.to(records.device), cnt
    world = dist.get_world_size(group)
    send, counts = bucket_by_owner(records, n, record_words, world)
    counts_cpu = counts.to("cpu")
    recv_counts = torch.zeros(world, dtype=torch.int64, device=records.device)
    if records.device.type == "cuda":
        dist.all_to_all_single(recv_counts, counts, group=group)
    else:
        outs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        ins = [counts[r:r + 1].clone() for r in range(world)]
        _all_to_all_lists(outs, ins, group)
        recv_counts = torch.cat(outs)
    in_split = [int(c) * record_words for c in counts_cpu.tolist()]
    out_split = [int(c) * record_words for c in recv_counts.to("cpu").tolist()]
    recv = torch.zeros(sum(out_split), dtype=torch.int64, device=records.device)
    if records.device.type == "cuda":
        dist.all_to_all_single(recv, send, out_split, in_split, group=group)
    else:
        outs = [torch.zeros(s, dtype=torch.int64) for s in out_split]
        ins = list(torch.split(send, in_split))
        _all_to_all_lists(outs, ins, group)
        recv = torch.cat(outs) if outs else recv
    return recv, sum(out_split) // record_words


def _all_to_all_lists(outs, ins, group):
    """gloo has no all_to_all: emulate with point-to-point transfers (empty
    messages are skipped on both sides; the sizes are known from the counts)"""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    reqs = []
    for r in range(world):
        if r == rank:
            outs[r].copy_(ins[r])
            continue
        if ins[r].numel():
            reqs.append(dist.isend(ins[r].contiguous(), r, group=group))
        if outs[r].numel():
            reqs.append(dist.irecv(outs[r], r, group=group))
    for q in reqs:
        q.wait()
