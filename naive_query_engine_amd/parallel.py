"""Row-range sharding of the hot operators across GPUs — a thin caller of the C ABI's sharded entry points.

One process per GPU.  The exchange itself lives in libnqe_hip.so (csrc/exchange.hip: nqe_comm_*, nqe_sharded_*,
nqe_table_all_gather) on RCCL over xGMI, enqueued on the context's stream; this module only

  * bootstraps the communicator: rank 0 draws the RCCL unique id, torch.distributed (any backend) carries its 128 bytes
    (`make_comm`);
  * offers `HostStagedTransport`, a transport that stages through host memory over a torch.distributed group (gloo): it drives
    the very same C++ sharding code where RCCL cannot run — two ranks on ONE GPU in the tests;
  * keeps numpy restatements of the wire layout (`pack_words_numpy`, …) for the CPU (gloo, world size 2) protocol tests.

How the path shards:
  * filter / projection / join probe: rows are independent → contiguous row ranges per rank keep the reference's output order
    (rank order == row order); no collective unless every rank must materialise the whole result (`gather=True`: ordered
    variable-length all-gather).
  * hash aggregate: every rank aggregates its row range into partial state {count,sum,min,max} per group, the partial tables
    travel in one all-gather and are merged on every rank; avg is finalised after the merge (sum/count), never averaged per rank.
  * hash join: the build side is replicated (every rank builds from its own copy), the probe side is range-split.

The reference has no distributed code at all (single process, single thread); this is the build's own design.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .capi import COMM_ID_BYTES, EXCHANGE_ROWS, Comm  # noqa: F401  (re-exported)


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


def make_comm(ctx, group=None) -> Comm:
    """RCCL communicator over `ctx` for the ranks of a torch.distributed group: rank 0 draws the unique id, a broadcast carries
    it (host bytes over whatever backend the group has), every rank joins.  One device per rank (RCCL's rule)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        t = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8, device=torch.device("cuda", ctx.device))
    else:
        t = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8))
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return Comm.rccl(ctx, bytes(t.cpu().numpy().tobytes()), rank, world)


class HostStagedTransport:
    """nqe_transport over a torch.distributed group with host staging (gloo): device → host → collective → host → device.
    For hosts without RCCL between their ranks — here: the two-ranks-on-one-GPU tests of the C++ sharding code."""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist

        self.ctx = ctx
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _to_host(self, ptr: int, nbytes: int):
        import torch

        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8)
        return torch.as_tensor(DeviceArray(ptr, nbytes, "|u1"), device=torch.device("cuda", self.ctx.device)).cpu()

    def _to_device(self, ptr: int, host):
        import torch

        if host.numel():
            torch.as_tensor(DeviceArray(ptr, host.numel(), "|u1"), device=torch.device("cuda", self.ctx.device)).copy_(host)

    def all_gather(self, send: int, recv: int, nbytes: int):
        import torch
        import torch.distributed as dist

        self.ctx.synchronize()  # `send` was produced on the context's stream
        mine = self._to_host(send, nbytes)
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        self._to_device(recv, torch.cat(outs))
        torch.cuda.synchronize(self.ctx.device)

    def all_gather_v(self, send: int, send_bytes: int, recv: int, offsets: Sequence[int], sizes: Sequence[int]):
        import torch
        import torch.distributed as dist

        self.ctx.synchronize()
        assert sizes[self.rank] == send_bytes
        mx = max(sizes)
        mine = torch.zeros(mx, dtype=torch.uint8)
        mine[:send_bytes] = self._to_host(send, send_bytes)
        outs = [torch.empty(mx, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        for r in range(self.world):
            self._to_device(recv + offsets[r], outs[r][: sizes[r]].contiguous())
        torch.cuda.synchronize(self.ctx.device)


def make_staged_comm(ctx, group=None) -> Comm:
    import torch.distributed as dist

    return Comm.custom(ctx, HostStagedTransport(ctx, group), dist.get_rank(group), dist.get_world_size(group))


# ---- sharded operators: `comm` = capi.Comm over the rank's context; comm=None runs the single-rank operator
def sharded_aggregate(comm, local_table, aggs, group_nodes=None, pred_nodes=None, ctx=None):
    """Aggregate over the union of every rank's `local_table`; returns (result_table, keys_table) on every rank (identical up
    to f64 summation order of the merge).  nqe_sharded_aggregate_execute: partial → ONE all-gather of a fixed-size buffer whose
    last word is the group count → merge that reads the counts on the device; partials above EXCHANGE_ROWS groups take an
    exact-size two-step exchange."""
    if comm is None:
        return ctx.aggregate(local_table, aggs, group_nodes=group_nodes, pred_nodes=pred_nodes, with_keys=True)
    return comm.sharded_aggregate(local_table, aggs, group_nodes=group_nodes, pred_nodes=pred_nodes)


def sharded_hash_join(comm, left_table, right_local_table, left_key: int, right_key: int, gather: bool = False, join_table=None, ctx=None):
    """Inner hash join with the build side replicated (every rank holds `left_table` and builds its own table) and the probe side
    range-split (`right_local_table` = this rank's contiguous row range).  The local output is already in probe order;
    concatenating the ranks' outputs in rank order reproduces the single-GPU row order.  `gather=True` materialises that
    concatenation on every rank (for large outputs this dominates — SURVEY §8e)."""
    c = comm.ctx if comm is not None else ctx
    jt = join_table or c.hash_join_build(left_table, left_key)
    if comm is None:
        return c.hash_join_probe(jt, right_local_table, right_key)
    return comm.sharded_hash_join_probe(jt, right_local_table, right_key, gather=gather)


def sharded_selection_projection(comm, local_table, pred_nodes, exprs, gather: bool = False, ctx=None):
    """Filter + projection over a row-range shard; rows are independent, so there is no exchange unless one rank wants the whole
    result (`gather=True`: ordered variable-length all-gather, rank order == row order)."""
    if comm is None:
        return ctx.selection_projection(local_table, pred_nodes, exprs)
    return comm.sharded_selection_projection(local_table, pred_nodes, exprs, gather=gather)


# ---- torch-level helpers of the CPU protocol tests (tests/test_parallel_gloo.py): the same wire layout, restated on the host
def all_gather_rows(cols: Sequence, group=None) -> Tuple[List[list], List[int]]:
    """Ordered variable-length all-gather of equally long 1-D tensors (one per column): counts first, then the columns padded to
    the maximum.  Returns (per_rank_columns, counts).  Host restatement of nqe_table_all_gather's protocol for the gloo tests."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = int(cols[0].numel()) if cols else 0
    cnt = torch.tensor([n_local], dtype=torch.int64)
    counts_t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts_t, cnt, group=group)
    counts = [int(c.item()) for c in counts_t]
    mx = max(counts) if counts else 0
    per_rank: List[list] = [[] for _ in range(world)]
    if not cols:
        return per_rank, counts
    ncols = len(cols)
    packed = torch.zeros((ncols, mx), dtype=torch.int64)
    for i, c in enumerate(cols):
        packed[i, :n_local] = c.view(torch.int64) if c.dtype != torch.int64 else c
    bufs = [torch.empty((ncols, mx), dtype=torch.int64) for _ in range(world)]
    if mx:
        dist.all_gather(bufs, packed, group=group)
    for r in range(world):
        for i, c in enumerate(cols):
            col = bufs[r][i, : counts[r]].contiguous()
            per_rank[r].append(col if c.dtype == torch.int64 else col.view(c.dtype))
    return per_rank, counts


def _all_gather_packed(buf, group=None):
    """all-gather of equally sized 1-D int64 host buffers → [world, len] tensor"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty(world * buf.numel(), dtype=buf.dtype)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out.view(world, buf.numel())


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
