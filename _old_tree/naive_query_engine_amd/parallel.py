"""Row-range sharding of the hot operators across GPUs.

One process per GPU; torch.distributed (backend "nccl" = RCCL over xGMI, "gloo" in the CPU tests)
is used only as plumbing for the exchange steps the path really has:

  * filter / projection / join probe: rows are independent → contiguous row ranges per rank keep the
    reference's output order (rank order == row order); no collective unless one rank must
    materialise the whole result, in which case the variable-length batches are all-gathered in rank
    order (`all_gather_rows`).
  * hash aggregate: every rank aggregates its row range into partial state {count,sum,min,max} per
    group (nqe_aggregate_partial); the tiny partial tables are all-gathered and merged on every rank
    (nqe_aggregate_merge).  avg is finalised after the merge (sum/count), never averaged per rank.
  * hash join: the build side is replicated (every rank builds from its own copy), the probe side is
    range-split; outputs are per-rank batches in probe order.

The reference has no distributed code at all (single process, single thread); this module is the
build's own design and has no reference analogue.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def _single(group=None) -> bool:
    """True when there is nothing to exchange.  NQE_FORCE_EXCHANGE=1 keeps the exchange path even for one rank (used to
    exercise the RCCL collectives on a single-GPU box)."""
    import os

    import torch.distributed as dist

    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and not os.environ.get("NQE_FORCE_EXCHANGE")


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous row range [lo, hi) of rank `rank`: sizes differ by at most one row"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DeviceArray:
    """Zero-copy view of device memory for torch (`torch.as_tensor(DeviceArray(...), device='cuda')`)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner=None):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr if n else 0, False), "version": 2}
        self._owner = owner


def table_columns_as_tensors(table, device) -> list:
    """8-byte columns of an nqe table as int64 torch tensors (bit patterns), zero-copy."""
    import torch

    out = []
    for i in range(table.num_columns):
        info = table.column_info(i)
        n = int(info.length)
        if n == 0:
            out.append(torch.empty(0, dtype=torch.int64, device=device))
        else:
            out.append(torch.as_tensor(DeviceArray(info.values, n, "<i8", owner=table), device=device))
    return out


def all_gather_rows(cols: Sequence, group=None) -> Tuple[List[list], List[int]]:
    """Ordered variable-length all-gather.  `cols` = equally long 1-D tensors (one per column) on this
    rank.  Returns (per_rank_columns, counts): per_rank_columns[r][c] is rank r's column c.  Counts are
    exchanged first, then columns are padded to the maximum and gathered with one all_gather each."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = int(cols[0].numel()) if cols else 0
    dev = cols[0].device if cols else torch.device("cpu")
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # gloo has no device all_gather: stage the (tiny) partial tables through the host. RCCL ("nccl")
        # gathers device buffers directly.
        per_rank, counts = all_gather_rows([c.cpu() for c in cols], group)
        return [[t.to(dev) for t in rc] for rc in per_rank], counts
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts_t, cnt, group=group)
    counts = [int(c.item()) for c in counts_t]
    mx = max(counts) if counts else 0
    per_rank: List[list] = [[] for _ in range(world)]
    if not cols:
        return per_rank, counts
    # all columns are 8-byte words: pack them into ONE [ncols, mx] buffer → a single collective per exchange
    ncols = len(cols)
    packed = torch.zeros((ncols, mx), dtype=torch.int64, device=dev)
    for i, c in enumerate(cols):
        packed[i, :n_local] = c.view(torch.int64) if c.dtype != torch.int64 else c
    bufs = [torch.empty((ncols, mx), dtype=torch.int64, device=dev) for _ in range(world)]
    if mx:
        dist.all_gather(bufs, packed, group=group)
    if dev.type == "cuda":
        # collectives run on torch's stream; the consumers (nqe kernels) run on the context's own stream
        torch.cuda.synchronize(dev)
    for r in range(world):
        for i, c in enumerate(cols):
            col = bufs[r][i, : counts[r]].contiguous()
            per_rank[r].append(col if c.dtype == torch.int64 else col.view(c.dtype))
    return per_rank, counts


# rows of the fixed-size exchange buffer: partial states up to this many groups travel in ONE collective (header
# included); larger ones take the exact-size two-collective path
EXCHANGE_ROWS = 4096


def _shares_stream(ctx, dev) -> bool:
    """the context launches on torch's current stream (Context(device, stream=torch_stream.cuda_stream) under
    `torch.cuda.stream(torch_stream)`): kernels and collectives are ordered by the stream itself, no host synchronisation
    between them"""
    import torch

    return ctx.stream is not None and int(ctx.stream) == int(torch.cuda.current_stream(dev).cuda_stream)


def _all_gather_packed(buf, group=None):
    """all-gather of equally sized 1-D int64 buffers → [world, len] tensor (gloo + CUDA stages through the host)"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if buf.device.type == "cuda" and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        outs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(outs, host, group=group)
        return torch.stack(outs).to(buf.device)
    out = torch.empty(world * buf.numel(), dtype=buf.dtype, device=buf.device)  # flat: gloo accepts only the concatenated form
    dist.all_gather_into_tensor(out, buf, group=group)
    return out.view(world, buf.numel())


def sharded_aggregate(ctx, local_table, aggs, group_nodes=None, pred_nodes=None, group=None):
    """Aggregate over the union of every rank's `local_table`; returns (result_table, keys_table)
    on every rank (identical up to f64 summation order of the merge).

    Exchange: the partial (keys + {count,sum,min,max} per value column) is packed into one fixed-size device buffer whose
    last word is the group count, all-gathered with ONE collective, and merged straight from the gathered buffer
    (nqe_aggregate_merge_packed reads the counts from the headers on the device).  When the context launches on torch's
    current stream the whole exchange + merge costs the host ONE wait (the merged group count); with a private context
    stream the two streams are ordered by host synchronisations.  Partials with more than EXCHANGE_ROWS groups on any rank
    fall back to the exact-size path (counts first, then data)."""
    import torch
    import torch.distributed as dist

    state, keys = ctx.aggregate_partial(local_table, aggs, group_nodes=group_nodes, pred_nodes=pred_nodes)
    if _single(group):
        return ctx.aggregate_merge([state], [keys] if keys is not None else None, aggs)
    dev = torch.device("cuda", ctx.device)
    tables = ([keys] if keys is not None else []) + [state]
    dts = [d for t in tables for d in t.dtypes()]
    ncols = len(dts)
    nk = 1 if keys is not None else 0
    rows = state.num_rows
    stride = EXCHANGE_ROWS
    words = ncols * stride + 1
    buf = torch.empty(words, dtype=torch.int64, device=dev)
    fits = rows <= stride
    if fits:
        ctx.pack_words(tables, stride, buf.data_ptr())
    else:
        buf[-1] = rows  # header only: tells the peers to take the exact-size path
    shared = _shares_stream(ctx, dev)
    if not shared:
        ctx.synchronize()  # the pack ran on the context's stream, the collective is ordered after torch's current stream
    gathered = _all_gather_packed(buf, group)
    if not shared:
        torch.cuda.current_stream(dev).synchronize()  # ... and the merge runs on the context's stream again
    # the merge reads the row counts from the headers on the device: no count read-back, no unpack, one host wait in all
    merged = ctx.aggregate_merge_packed(gathered.data_ptr(), gathered.shape[0], stride, nk == 1, dts[0] if nk else 0, aggs)
    if merged is not None:
        if not shared:
            ctx.synchronize()  # `gathered` goes back to torch's allocator
        return merged
    # ---- exact-size path
    cols = []
    for t in tables:
        cols += table_columns_as_tensors(t, dev)
    per_rank, counts = all_gather_rows(cols, group)
    cat = [torch.cat([per_rank[r][i] for r in range(len(per_rank))]).contiguous() for i in range(ncols)]
    total = int(sum(counts))
    torch.cuda.synchronize(dev)
    keyt = ctx.table_from_device([(dts[0], total, cat[0].data_ptr() if total else None, None)]) if nk else None
    st = ctx.table_from_device([(dts[nk + i], total, cat[nk + i].data_ptr() if total else None, None) for i in range(ncols - nk)])
    out = ctx.aggregate_merge([st], [keyt] if keyt is not None else None, aggs)
    ctx.synchronize()
    del cat
    return out


def _gather_table(ctx, table, group=None):
    """All-gathers a per-rank result table (8-byte columns without validity) in rank order = row order; returns one table.
    Counts are exchanged first (the per-rank outputs differ in length), then every rank packs its table into one buffer of
    the common stride (nqe_table_pack_words), ONE all-gather moves the data, and nqe_table_unpack_words writes the
    concatenation: two device copies around the collective instead of per-column pads, slices and a concat."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", ctx.device)
    dts = table.dtypes()
    ncols = len(dts)
    world = dist.get_world_size(group)
    n_local = table.num_rows
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    if dist.get_backend(group) == "gloo":
        cl = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(cl, cnt.cpu(), group=group)
    else:
        cl = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(cl, cnt, group=group)
    counts = [int(c.item()) for c in cl]
    if min(counts) == max(counts) and counts[0] > 0 and dist.get_backend(group) != "gloo":
        # equal-length outputs (e.g. a PK-FK join over equal shards): gather every column straight from the table's memory
        # into its place in the result — the collective's output IS the concatenated column, no staging copies at all
        outs = []
        for t_in in table_columns_as_tensors(table, dev):
            o = torch.empty(world * n_local, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(o, t_in, group=group)
            outs.append(o)
        torch.cuda.synchronize(dev)
        res = ctx.table_from_device([(dts[i], world * n_local, outs[i].data_ptr(), None) for i in range(ncols)])
        res._keep = outs  # the table borrows the gathered tensors
        return res
    stride = max(max(counts), 1)
    buf = torch.empty(ncols * stride + 1, dtype=torch.int64, device=dev)
    ctx.pack_words([table], stride, buf.data_ptr())
    ctx.synchronize()  # the pack ran on the context's stream, the collective runs on torch's
    gathered = _all_gather_packed(buf, group)
    torch.cuda.synchronize(dev)
    out = ctx.unpack_words(gathered.data_ptr(), counts, dts, stride)
    ctx.synchronize()
    del gathered, buf
    return out


def sharded_hash_join(ctx, left_table, right_local_table, left_key: int, right_key: int, gather: bool = False, group=None, join_table=None):
    """Inner hash join with the build side replicated (every rank holds `left_table` and builds its own table)
    and the probe side range-split (`right_local_table` = this rank's contiguous row range).  The local output is
    already in probe order; concatenating the ranks' outputs in rank order reproduces the single-GPU row order.
    `gather=True` materialises that concatenation on every rank (variable-length all-gather over RCCL; for large
    outputs this dominates — SURVEY §8e); validity bitmaps / Utf8 columns are not gathered by this helper."""
    import torch.distributed as dist

    jt = join_table or ctx.hash_join_build(left_table, left_key)
    local = ctx.hash_join_probe(jt, right_local_table, right_key)
    if not gather or _single(group):
        return local
    return _gather_table(ctx, local, group)


def sharded_selection_projection(ctx, local_table, pred_nodes, exprs, gather: bool = False, group=None):
    """Filter + projection over a row-range shard; rows are independent, so there is no exchange unless one rank
    wants the whole result (`gather=True`: ordered variable-length all-gather, rank order == row order)."""
    import torch.distributed as dist

    local = ctx.selection_projection(local_table, pred_nodes, exprs)
    if not gather or _single(group):
        return local
    return _gather_table(ctx, local, group)


def pack_words_numpy(cols: Sequence[np.ndarray], stride: int) -> np.ndarray:
    """Host restatement of nqe_table_pack_words' layout (CPU gloo tests): [ncols*stride] words + the row count."""
    rows = len(cols[0]) if cols else 0
    assert rows <= stride
    buf = np.zeros(len(cols) * stride + 1, dtype=np.int64)
    for c, a in enumerate(cols):
        buf[c * stride : c * stride + rows] = np.ascontiguousarray(a).view(np.int64)
    buf[-1] = rows
    return buf


def unpack_words_numpy(gathered: np.ndarray, ncols: int, stride: int) -> Tuple[List[np.ndarray], List[int]]:
    """Host restatement of nqe_table_unpack_words: [world, ncols*stride+1] → per column the parts concatenated."""
    counts = [int(c) for c in gathered[:, -1]]
    return [np.concatenate([gathered[p, c * stride : c * stride + counts[p]] for p in range(gathered.shape[0])]) for c in range(ncols)], counts


def merge_partials_numpy(keys_list: Sequence[Optional[np.ndarray]], states_list: Sequence[Sequence[np.ndarray]]):
    """Host restatement of the partial merge rule (used by the CPU gloo tests to check the exchange;
    the product path merges on the GPU with nqe_aggregate_merge).  states = per distinct aggregate column
    (count u64, sum f64, min f64, max f64)."""
    grouped = keys_list[0] is not None
    acc = {}
    for keys, st in zip(keys_list, states_list):
        n = len(st[0])
        for r in range(n):
            k = int(keys[r]) if grouped else 0
            row = acc.setdefault(k, [[0, 0.0, np.finfo(np.float64).max, -np.finfo(np.float64).max] for _ in range(len(st) // 4)])
            for i in range(len(st) // 4):
                c, s, mn, mx = st[4 * i][r], st[4 * i + 1][r], st[4 * i + 2][r], st[4 * i + 3][r]
                row[i][0] += int(c)
                row[i][1] += float(s)
                row[i][2] = min(row[i][2], float(mn))
                row[i][3] = float("nan") if (np.isnan(mx) or np.isnan(row[i][3])) else max(row[i][3], float(mx))
    return acc
