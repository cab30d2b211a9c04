"""Sharding of the flow path over several GPUs (one process per GPU).

The unit of sharding is the exporter: template state is keyed by exporter
address + source id (netflow-templates.c:47-66 builds its key from the
datagram's source address and the header's source id), so all datagrams of one
exporter must reach the same rank, in arrival order; datagrams of different
exporters are independent.  No record data crosses ranks.  The only exchange is
of results: per-object counters are plain wrapping u64 sums
(monit-objects.c:995-1007), so shards merge by addition.
"""
import numpy as np


def exporter_rank(src_ip4, world):
    """rank owning each exporter address (u32 array, any byte order)"""
    x = np.asarray(src_ip4, dtype=np.uint64)
    x = (x * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return ((x >> np.uint64(33)) % np.uint64(max(1, world))).astype(np.int64)


def select(packed, rank, world):
    """the datagrams of `packed` = (data, offsets, src, kinds) owned by `rank`,
    in arrival order, in the same packed form"""
    data, offs, src, kinds = packed
    data = np.asarray(data, dtype=np.uint8)
    offs = np.asarray(offs, dtype=np.uint64)
    src = np.asarray(src, dtype=np.uint32)
    if world <= 1:
        return data, offs, src, kinds
    mine = np.nonzero(exporter_rank(src, world) == rank)[0]
    lens = (offs[1:] - offs[:-1])[mine].astype(np.int64)
    new_offs = np.zeros(len(mine) + 1, dtype=np.uint64)
    np.cumsum(lens, out=new_offs[1:].view(np.int64))
    out = np.empty(int(new_offs[-1]), dtype=np.uint8)
    starts = offs[:-1][mine].astype(np.int64)
    # one gather: index of every byte that is kept
    if len(mine):
        idx = np.repeat(starts - new_offs[:-1].astype(np.int64), lens) + np.arange(out.size, dtype=np.int64)
        out[:] = data[idx]
    return out, new_offs, src[mine].copy(), (None if kinds is None else np.asarray(kinds, dtype=np.uint8)[mine].copy())


def merge_counters(counters, dist=None, device=None):
    """sum of the per-object counter tables of all ranks (wrapping u64);
    every rank gets the result.  `dist` is torch.distributed when a process
    group is up, else the input is returned unchanged."""
    c = np.ascontiguousarray(counters, dtype=np.uint64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return c.copy()
    import torch
    t = torch.from_numpy(c.view(np.int64).copy())   # two's-complement add == u64 wrap
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64)


def reduce_scalars(values, ops, dist=None, device=None):
    """all-reduce a few float64 scalars; ops[i] is 'sum' or 'max'"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    import torch
    out = []
    for v, op in zip(values, ops):
        t = torch.tensor([float(v)], dtype=torch.float64)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        out.append(float(t.item()))
    return out
