#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X, plus every BASELINE config in the same JSON line.

Headline (BASELINE.json `metric`: "rows/s + achieved HBM GB/s, 10^9-row filter→hash-agg"):

    select count(v), sum(v), avg(v), min(v), max(v) from t where id < N/2 group by id % 1024

over t(id Int64 = row number, v Float64 in [0,100)) with N = 10^9 rows PER GPU (weak scaling: rank r holds rows [r*N, (r+1)*N)
of a world*N-row table and the predicate is `id < world*N/2`), synthetic, generated on the device (SURVEY §8d generators),
HBM-resident before the timed region.  A step = one full pass of the fused filter→hash-aggregate over the rank's table, plus
(N>1) the all-gather + merge of the per-rank partial tables (nqe_sharded_aggregate_execute: RCCL on the context's stream).
Algorithmic bytes = 16 B/row (id + v each read once; no skip credit for filtered-out rows — the kernel loads unconditionally).

Launch: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one rank per device) when WORLD_SIZE is not
set, and exits non-zero when the box has fewer than N devices or WORLD_SIZE disagrees with --gpus: there is no silent 1-GPU
fallback.  Under the driver's own `python -m torch.distributed.run … bench.py --gpus N` it reads RANK/LOCAL_RANK/WORLD_SIZE.

Rank 0 prints ONE JSON line (kept under 8 KB so that a record which keeps only the tail of stdout still holds all of it): the
contract's keys for the headline, `roofline` (dominant kernel timed with HIP events on the launch stream), `cpu_baseline` (the
oracle — a C++ restatement of the reference's single-threaded algorithm — on a bounded sample, N=1 only), `parity_checked` (the
GPU result on that sample compared with the oracle's: counts exact, f64 within 1e-9) and `configs`: the other BASELINE configs
(C2 in both id orders, C3, C4 with build time), the random-key variants, the Int64-value and single-column forms of the headline,
C1's three-value-column query at scale, joins with a wide payload / a 10^7-row dim / sparse keys / duplicate build keys / a
partially matching foreign key, many-groups aggregates and a general predicate tree (N>1: the headline without its exchange and
C5 = the join range-split over the ranks with its output all-gathered).  EVERY side config is timed as the median of three
blocks (`ms` with `ms_min`/`ms_max`) and checked against the oracle on a sample in the same run (`parity`).  A config record is
compact: `frac` = SURVEY §8d's algorithmic bytes over the summed HIP-event time of the step's data kernels, as a fraction of the
8 TB/s peak — EXCEPT where the data lets a kernel skip reads (C2 over sorted ids): there `frac` counts only the bytes that move
and `frac_8d` is shown beside it; `frac_physical` = the bytes that physically move; `cold_ms` = the first execution of the query
shape in the process (no plan hints, no remembered join form).  The LAST key, `summary`, repeats {config: [ms, frac,
frac_physical, parity ok]} ({dropin_*: [ms, raw C-ABI ms, ms / raw, parity ok]}).  `dropin_headline` / `dropin_c2` / `dropin_c4` run the
same queries the way the reference's caller does — a fresh operator tree per execution through the mirrors of its PhysicalPlan surface,
the rewrite pass, root.execute(), tables registered once — and `upload` is what registering the headline's table from HOST memory costs
(pageable and page-locked numpy arrays through nqe_table_create).  Per-kernel maps and full workload descriptions go to `--details PATH` (default:
gpurun_out/bench_details.json when that directory exists).  `--workload X` runs one config as the main line; `--no-configs`
skips the block (used under rocprofv3).
"""
import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable)
PROFILE_TAG = "r06"    # profiles/<tag>/pmc_traffic_<config>.json are quoted as `roofline.traffic`
# kernels that do not belong to an operator's step (data generation, diagnostics)
NOT_STEP_KERNELS = ("synth_fill",)
LINE_LIMIT = 8000      # bytes of the JSON line


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "headline_int64", "headline_single", "agg3", "agg_readme", "headline_nullable", "tree_pred", "c2", "c2_random", "c2_tree", "c3",
                                                               "c4", "c4_sparse", "c4_wide", "c4_dup", "c4_partial", "agg_groups"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the BASELINE size of the workload)")
    ap.add_argument("--random-keys", action="store_true", help="c3/headline: group by a random id column instead of the row number")
    ap.add_argument("--groups", type=int, default=65536, help="agg_groups: distinct keys")
    ap.add_argument("--no-minmax", action="store_true", help="agg_groups: count / sum / avg only")
    ap.add_argument("--dim-rows", type=int, default=10**6, help="c4: build-side rows")
    ap.add_argument("--pass-frac", type=float, default=0.5, help="headline: fraction of rows passing `id < K` (diagnostics; the metric uses 0.5)")
    ap.add_argument("--gather", action="store_true", help="c4 with --gpus N: also all-gather every rank's output batch in rank order (BASELINE config C5)")
    ap.add_argument("--immutable", action="store_true", help="c4*: the probe table is created with NQE_TABLE_IMMUTABLE (an all-match join may share its columns)")
    ap.add_argument("--reserve-gb", type=float, default=32.0, help="nqe_ctx_reserve: device memory the context takes from the driver at start-up and serves its outputs "
                                                                    "and scratch from (0 = none: every first allocation is a hipMalloc)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the communicator self-test that runs before the big allocations")
    ap.add_argument("--preflight-timeout", type=float, default=180.0, help="seconds a preflight stage may take before the rank reports where it is stuck and exits")
    ap.add_argument("--no-configs", action="store_true", help="only the main workload's line (no `configs` block)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure `roofline.traffic` with rocprofv3 --pmc child passes after the timed region (the default run does; quoted figure only)")
    ap.add_argument("--only", default="", help="comma-separated side configs to run (default: all)")
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the OPTIONAL second CPU number of the headline — an optimised multi-core form, not the reference's "
                                                                "single-threaded algorithm (0 = skip, -1 = min(64, host cores))")
    ap.add_argument("--cpu-sample-rows", type=int, default=150_000_000, help="rows of the CPU baseline sample (about 10 s of single-thread work for the headline)")
    ap.add_argument("--details", default="", help="write the full per-config records (per-kernel times, workload text) to this file")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: one rank per device under torch.distributed.run; never returns"""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("NQE_BENCH_TRANSPORT") != "host":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this box has {have}; refusing to run on fewer\n")
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class F:
    def __init__(self, name):
        self.name = name


def preflight(B, parallel):
    """N > 1 ranks: the exchange layer's first contact with real peers, made cheap to diagnose (task 7).  Before the
    10^9-row allocations every rank (1) creates the library's communicator (nqe_comm_create: ncclCommInitRank with N ids), (2) runs
    one fixed-size collective — a sharded aggregate of 1000 rows per rank, i.e. ONE ncclAllGather of a packed partial — and (3) one
    ordered variable-length all-gather of a 3-column, (1000 + rank)-row table (p2p_all_gather_v: a group of N^2 ncclSend / ncclRecv
    per column), checks both results against what they must be, and reports rank / device / RCCL version on stderr.  A stage that
    raises names itself and the rank; a stage that HANGS is reported by a watchdog after --preflight-timeout seconds (the process
    exits 4).  Every rank learns every rank's verdict (all_gather_object) and all exit non-zero together."""
    import threading

    import numpy as np

    from naive_query_engine_amd import Column, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64

    torch, dist = B.torch, B.dist
    rank, world = dist.get_rank(), dist.get_world_size()
    t_start = time.perf_counter()
    stage = {"name": "start", "since": t_start}
    done = threading.Event()
    tag = f"bench.py preflight: rank {rank}/{world} (device {B.local_rank}, pid {os.getpid()})"

    def watchdog():
        if not done.wait(B.args.preflight_timeout):
            sys.stderr.write(f"{tag}: STUCK in stage '{stage['name']}' for {time.perf_counter() - stage['since']:.0f} s — a peer never joined, or the "
                             f"transport hangs (RCCL: NCCL_DEBUG=INFO shows the ring / p2p set-up); exiting 4\n")
            sys.stderr.flush()
            os._exit(4)

    threading.Thread(target=watchdog, daemon=True).start()

    def enter(name):
        stage["name"], stage["since"] = name, time.perf_counter()

    rec = {"rank": rank, "device": B.local_rank, "ok": False, "stage": None, "error": None}
    try:
        enter("device properties")
        rec["device_name"] = torch.cuda.get_device_name(B.local_rank)
        enter("communicator (nqe_comm_create: ncclCommInitRank)" if not B.host_transport else "communicator (host-staged transport)")
        B.comm = parallel.make_staged_comm(B.ctx) if B.host_transport else parallel.make_comm(B.ctx)
        rec["comm_world"], rec["comm_rank"] = B.comm.world, B.comm.rank
        rec["rccl_version"] = None if B.host_transport else B.capi.Comm.rccl_version()
        if B.comm.world != world or B.comm.rank != rank:
            raise RuntimeError(f"communicator says rank {B.comm.rank} of {B.comm.world}")
        m = 1000
        # (2) one fixed-size collective: `select count(v), sum(v) from t group by id % 3` over 1000 rows per rank, ids = global row numbers
        enter("fixed-size collective (sharded aggregate: one ncclAllGather of a packed partial)")
        ids = np.arange(rank * m, (rank + 1) * m, dtype=np.int64)
        v = (ids % 7).astype(np.float64)
        w = ids * 3 + 1
        t = B.ctx.table_from_host([Column.from_numpy(ids), Column.from_numpy(v), Column.from_numpy(w)])
        f = [F("id"), F("v"), F("w")]
        out, keys = B.comm.sharded_aggregate(t, [(AGG.Count, 1), (AGG.Sum, 1)], group_nodes=binop(col(0), Operator.Modulos, lit_i64(3)).flatten(f))
        B.ctx.synchronize()
        allid = np.arange(world * m, dtype=np.int64)
        cnt, sm = [c.to_numpy() for c in out.to_host()]
        kk = keys.to_host()[0].to_numpy()
        exp_cnt = np.array([(allid % 3 == g).sum() for g in range(3)])
        exp_sum = np.array([float((allid[allid % 3 == g] % 7).sum()) for g in range(3)])
        if not ((kk == np.arange(3)).all() and (cnt.astype(np.int64) == exp_cnt).all() and (sm == exp_sum).all()):
            raise RuntimeError(f"sharded aggregate returned keys {kk.tolist()} counts {cnt.tolist()} sums {sm.tolist()}, expected counts {exp_cnt.tolist()} sums {exp_sum.tolist()}")
        # (3) one ordered variable-length all-gather: rank r contributes 1000 + r rows of three columns
        enter("variable-length all-gather (p2p_all_gather_v: ncclSend / ncclRecv between every pair of ranks)")
        mr = m + rank
        base = sum(m + r for r in range(rank))
        gi = np.arange(base, base + mr, dtype=np.int64)
        tv = B.ctx.table_from_host([Column.from_numpy(gi), Column.from_numpy(gi * 0.5), Column.from_numpy(gi * 3 + 1)])
        g = B.comm.all_gather_table(tv)
        B.ctx.synchronize()
        total = sum(m + r for r in range(world))
        cols = [c.to_numpy() for c in g.to_host()]
        ei = np.arange(total, dtype=np.int64)
        if not (g.num_rows == total and (cols[0] == ei).all() and (cols[1] == ei * 0.5).all() and (cols[2] == ei * 3 + 1).all()):
            raise RuntimeError(f"gathered table has {g.num_rows} rows (expected {total}) or rows out of rank order")
        rec["ok"] = True
    except BaseException as e:  # noqa: BLE001 - reported with its stage and rank; every rank exits together below
        rec["stage"], rec["error"] = stage["name"], f"{type(e).__name__}: {e}"
    rec["seconds"] = r4(time.perf_counter() - t_start)
    sys.stderr.write(f"{tag}: {rec.get('device_name')}, RCCL version {rec.get('rccl_version')}, communicator {rec.get('comm_rank')}/{rec.get('comm_world')}: "
                     + ("ok" if rec["ok"] else f"FAILED in stage '{rec['stage']}': {rec['error']}") + f" ({rec['seconds']} s)\n")
    sys.stderr.flush()
    enter("verdict exchange (torch.distributed all_gather_object)")
    allrec = [None] * world
    dist.all_gather_object(allrec, rec)
    done.set()
    bad = [r for r in allrec if not r["ok"]]
    if bad:
        if rank == 0:
            for r in bad:
                sys.stderr.write(f"bench.py preflight: FAILED on rank {r['rank']} (device {r['device']}) in stage '{r['stage']}': {r['error']}\n")
        sys.exit(4)
    return {"ok": True, "ranks": world, "seconds": max(r["seconds"] for r in allrec), "rccl_version": rec.get("rccl_version"),
            "devices": [r.get("device_name") for r in allrec],
            "what": "communicator + one fixed-size collective (sharded aggregate) + one ordered variable-length all-gather (3 columns), results checked on every rank, before the big allocations"}


class Bench:
    def __init__(self, args, world, rank, local_rank):
        import torch
        import torch.distributed as dist

        from naive_query_engine_amd import capi, parallel

        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        self.torch, self.dist, self.capi = torch, dist, capi
        self.distributed = world > 1 or bool(os.environ.get("NQE_FORCE_EXCHANGE"))  # one rank through RCCL too (diagnostics)
        # NQE_BENCH_TRANSPORT=host: the ranks exchange through host memory over gloo (parallel.HostStagedTransport, plugged into the
        # library with nqe_comm_create_custom) and may therefore SHARE a device — `--gpus 2` on a one-GPU box runs this file's
        # multi-rank blocks (the sharded headline, C5) end to end, at reduced rows.  Not a scaling measurement: the line says so.
        self.host_transport = self.distributed and os.environ.get("NQE_BENCH_TRANSPORT") == "host"
        if self.host_transport:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.comm = None
        if self.distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.host_transport:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            if dist.get_world_size() != world:
                sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {world}")
        self.ctx = capi.Context(local_rank)
        self.preflight = None
        if self.distributed:
            # the data path's own communicator, on the context's stream: RCCL, or the host-staged transport — created and SELF-TESTED
            # (preflight) before anything large is allocated: the first contact of N real RCCL peers should cost seconds to diagnose
            if args.no_preflight:
                self.comm = parallel.make_staged_comm(self.ctx) if self.host_transport else parallel.make_comm(self.ctx)
            else:
                self.preflight = preflight(self, parallel)
            if self.comm.world != world:
                sys.stderr.write(f"bench.py: the library's communicator has {self.comm.world} ranks, --gpus asked for {world}\n")
                sys.exit(2)
        self.reserve_ms = None
        if args.reserve_gb > 0 and not self.host_transport:
            t0 = time.perf_counter()
            try:
                self.ctx.reserve(int(args.reserve_gb * (1 << 30)))
                self.reserve_ms = (time.perf_counter() - t0) * 1e3
            except Exception as e:  # noqa: BLE001 - a device without that much free memory: the pool and the driver serve every allocation
                sys.stderr.write(f"bench.py: nqe_ctx_reserve({args.reserve_gb} GB) failed ({e}); continuing without a reserved block\n")
        self.keep = []
        self.min_warm_s = 0.0

    def synth(self, kind, seed, rows, first_row=0, mod=1, base=0, dtype=None):
        """a synthetic column in a torch tensor, filled by the library's generator on the CONTEXT's stream.  Set-up code, fully
        synchronous on both sides: torch's caching allocator hands out blocks that earlier torch-stream work (the temporaries of a
        randperm, say) may still be using — safe for torch's own stream only — and torch ops that read the column later know
        nothing of the context's stream.  (Found the hard way: a 10^7-row randperm came back with duplicates.)"""
        self.torch.cuda.synchronize()
        t = self.torch.empty(rows, dtype=dtype or self.torch.int64, device=self.dev)
        self.ctx.synth_fill(kind, seed, first_row, rows, mod, base, t.data_ptr())
        self.ctx.synchronize()
        return t

    def barrier(self):
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.ctx.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if not self.distributed:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    @property
    def coll_dev(self):
        """where the tensors of torch.distributed collectives live: the device under RCCL, the host under gloo"""
        return self.torch.device("cpu") if self.host_transport else self.dev

    def _warm(self, step, warmup):
        """W untimed steps; the side configs (self.min_warm_s > 0) additionally warm up for a minimum wall time: they start right
        behind seconds of CPU-only work (the oracle), from a GPU that has clocked down."""
        t_w = time.perf_counter()
        for _ in range(warmup):
            r = step()
            del r
        if self.min_warm_s > 0:
            # the extra steps are a COUNT every rank agrees on (max over ranks): a step may hold a collective, and ranks that
            # looped on their own clocks would issue different numbers of them
            if warmup == 0:
                r = step()
                del r
            self.ctx.synchronize()
            elapsed = time.perf_counter() - t_w
            per = max(elapsed / max(warmup, 1), 1e-4)
            extra = min(2000, int(math.ceil(max(0.0, self.min_warm_s - elapsed) / per)))
            extra = int(self.max_over_ranks(float(extra)))
            for _ in range(extra):
                r = step()
                del r

    def _block(self, step, steps):
        """exactly K steps between barrier + synchronize on both sides; max over ranks.  Kernel times come from HIP events the
        library records around every launch on its stream while timing is enabled."""
        self.barrier()
        self.ctx.timing_enable(True)
        self.ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            del r
        self.barrier()
        dt = time.perf_counter() - t0
        self.ctx.timing_enable(False)
        kernels = {kn: {"ms_per_step": ms / steps, "launches_per_step": cnt / steps} for kn, (ms, cnt) in self.ctx.timing_report().items()
                   if kn not in NOT_STEP_KERNELS}
        return self.max_over_ranks(dt) / steps * 1e3, kernels

    def timed(self, step, steps, warmup, blocks=1, spread_blocks=None):
        """→ (ms per step, kernels, spread).  blocks = 1: the contract's W warm-up steps + exactly K timed steps.  blocks > 1 (the
        side configs): that many timed blocks of K steps behind one warm-up; the MEDIAN block is reported (its ms and its kernel
        times), with min/max over the blocks as `spread`.  `spread_blocks` more blocks of K steps follow the reported one(s) and
        feed ONLY the spread (the main line: its value is the contract's single block, its roofline says how far two more moved)."""
        if spread_blocks is None:
            spread_blocks = 2 if blocks == 1 else 0
        self._warm(step, warmup)
        runs = [self._block(step, steps) for _ in range(blocks)]
        more = [self._block(step, steps) for _ in range(spread_blocks)]
        every = runs + more
        runs.sort(key=lambda r: r[0])
        ms, kernels = runs[len(runs) // 2]
        return ms, kernels, {"ms_min": min(r[0] for r in every), "ms_max": max(r[0] for r in every), "blocks": len(every), "steps_per_block": steps,
                             "block_kernels": [r[1] for r in every]}

    def cold(self, step):
        """wall time of the FIRST execution of a query shape in this process (no plan hint, no remembered join form)"""
        self.barrier()
        t0 = time.perf_counter()
        r = step()
        self.ctx.synchronize()
        dt = time.perf_counter() - t0
        del r
        return dt * 1e3


def pick(kernels, prefixes):
    """the kernels of a step that stream the data (by name prefix), as opposed to its scans / table set-up / tails"""
    return sorted(k for k in kernels if any(k.startswith(p) for p in prefixes))


def kernel_ms(kernels, prefixes):
    return sum(kernels[k]["ms_per_step"] for k in pick(kernels, prefixes))


def roofline(algo_bytes, kernels, prefixes, phys_bytes=None, extra=None):
    """achieved = SURVEY §8d algorithmic bytes of one step ÷ the summed HIP-event time of the step's data kernels"""
    kms = kernel_ms(kernels, prefixes)
    achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": "+".join(pick(kernels, prefixes)), "kernel_ms_per_step": kms, "algorithmic_bytes_per_step": algo_bytes,
           "kernels": kernels}
    if phys_bytes is not None:
        out["physical_bytes_per_step"] = phys_bytes
        out["frac_physical"] = (phys_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0
    if extra:
        out.update(extra)
    return out


def attach_spread(res):
    """roofline.kernel_ms_min / _max (and frac_min / frac_max): the summed HIP-event time of the roofline's kernels in EVERY timed
    block of the config, not only the reported (median) one — three boxes disagreed by 8 % on the headline kernel last round"""
    roof = res["roofline"]
    # the same algorithmic bytes over the step's WALL time (the number the driver's own clock can vouch for; `frac` is over the HIP-event
    # time of the step's data kernels): launch gaps, scans, tails and the host's part of a step are inside it, so frac_step <= frac
    # (on the bytes `frac` counts: C2 over sorted ids quotes the bytes that move)
    if res.get("ms_per_step") and roof.get("kernel_ms_per_step"):
        roof["frac_step"] = roof["frac"] * roof["kernel_ms_per_step"] / res["ms_per_step"]
    blocks = res["spread"].pop("block_kernels", None)
    names = [k for k in roof.get("kernel", "").split("+") if k]
    if not blocks or not names or not roof.get("kernel_ms_per_step"):
        return
    kms = [sum(b[k]["ms_per_step"] for k in names if k in b) for b in blocks]
    kms = [x for x in kms if x > 0]
    if not kms:
        return
    roof["kernel_ms_min"], roof["kernel_ms_max"], roof["kernel_ms_blocks"] = min(kms), max(kms), len(kms)
    roof["frac_min"] = roof["frac"] * roof["kernel_ms_per_step"] / max(kms)
    roof["frac_max"] = roof["frac"] * roof["kernel_ms_per_step"] / min(kms)


def attach_traffic(roof, config_name):
    """HBM bytes per launch QUOTED from this round's PMC passes (tools/profile_round.sh → profiles/<tag>/pmc_traffic_<config>.json;
    rocprofv3 cannot run inside the timed process) — not a measurement of this run, hence `traffic_quoted_from`.  Only when the
    file was made from the csrc/ tree that is running."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG, f"pmc_traffic_{config_name}.json")
    if not os.path.exists(path):
        return
    try:
        with open(path) as f:
            rec = json.load(f)
        from tools.csrc_rev import csrc_rev

        if rec.get("csrc_rev") != csrc_rev():
            roof["traffic_quoted_from"] = f"{os.path.relpath(path, ROOT)} is stale (csrc rev {rec.get('csrc_rev')}, running {csrc_rev()}): not quoted"
            return
        roof["traffic"] = rec["hbm_bytes_per_step_corrected"]
        roof["traffic_ratio"] = rec["hbm_bytes_per_step_corrected"] / roof["algorithmic_bytes_per_step"] if roof["algorithmic_bytes_per_step"] else None
        roof["traffic_quoted_from"] = f"{os.path.relpath(path, ROOT)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this csrc revision, calibrated)"
        # what the counters count (MI355X_MICROARCH.md, HBM): the L2s' memory-side requests — Infinity-Cache hits included.  For a streamed
        # column that IS HBM traffic; for a join table of 25-80 MB gathered at random (c4_dim_1e7, c4_sparse_keys, c4_wide_payload) it is
        # 128-byte line traffic on the fabric, much of it served by the 256 MB Infinity Cache
        roof["traffic_counts"] = "L2 memory-side (fabric) requests; Infinity-Cache hits are included"
    except Exception as e:  # noqa: BLE001 - a missing/odd profile file must not fail the bench
        roof["traffic_quoted_from"] = f"unreadable {path}: {e}"


def _pmc_pass(counter, cmd, workdir, timeout_s):
    """one `rocprofv3 --pmc <counter> --kernel-trace` pass of `cmd` in a child process → {kernel name: (mean per dispatch, dispatches)}"""
    import csv
    import subprocess

    import signal

    env = dict(os.environ, TMPDIR="/tmp")
    d = os.path.join(workdir, counter.lower())
    # its own session: a pass that outlives its time limit is killed with everything it started (rocprofv3 is a launcher around the workload)
    proc = subprocess.Popen(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "live", "--"] + cmd,
                            cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        raise
    if rc != 0:
        raise RuntimeError(f"rocprofv3 --pmc {counter} exited with {rc}")
    acc = {}
    for root, _, files in os.walk(d):
        for fn in files:
            if fn.endswith("counter_collection.csv"):
                with open(os.path.join(root, fn)) as f:
                    for r in csv.DictReader(f):
                        if r.get("Counter_Name") == counter:
                            a = acc.setdefault(r["Kernel_Name"], [0.0, 0])
                            a[0] += float(r["Counter_Value"])
                            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


class LiveTraffic:
    """`roofline.traffic` MEASURED by the run that prints it.  After everything timed is done, child processes run a workload (3 steps)
    under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, as the guide's HBM section
    prescribes; the counters are reported in KB), and — when tools/stream_bench is built — two more calibrate the counters' factors
    on known byte counts in the same access pattern, once per run (else this round's committed factors are used:
    profiles/<tag>/pmc_calibration.json).  The step's kernels and their launches per step: tools/profile_configs.py, the list
    tools/summarize_profiles.py uses for the quoted files.  The quoted figure of attach_traffic stays beside the live one as
    `traffic_quoted`.  Any failure (no rocprofv3, a profiler already attached, a time-out) leaves the quoted figure in place and
    says why."""

    def __init__(self, timeout_s=60.0, budget_s=150.0):
        import shutil
        import tempfile

        # (a pass takes 2-4 s; `budget_s` bounds all of them together, and the first pass that fails ends the measuring for the run)
        self.timeout_s, self.why_not, self.factors, self.basis, self.work, self.deadline = timeout_s, None, None, None, None, time.time() + budget_s
        if shutil.which("rocprofv3") is None:
            self.why_not = "no rocprofv3 on PATH"
        elif any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
            self.why_not = "this process runs under a profiler"
        else:
            self.work = tempfile.mkdtemp(prefix="nqe_live_pmc_")

    def close(self):
        import shutil

        if self.work:
            shutil.rmtree(self.work, ignore_errors=True)

    def calibrate(self):
        if self.factors is not None:
            return
        sb = os.path.join(ROOT, "tools", "stream_bench")
        if os.path.exists(sb):
            try:
                cf = _pmc_pass("FETCH_SIZE", [sb, "calib"], os.path.join(self.work, "calib"), self.timeout_s)
                cw = _pmc_pass("WRITE_SIZE", [sb, "calib"], os.path.join(self.work, "calib"), self.timeout_s)
                rd = [v for k, v in cf.items() if "read2" in k]
                wr = [v for k, v in cw.items() if "copy_rw" in k and ("Li0ELi4E" in k or "copy_rw<4,0,4,1>" in k.replace(" ", ""))]
                if rd and wr:
                    self.factors = (16e9 / (rd[0][0] * 1024), 6.4e9 / (wr[0][0] * 1024))
                    self.basis = "calibrated in this run on tools/stream_bench (16e9 B read, 6.4e9 B written, 8-byte non-temporal accesses)"
                    return
            except Exception:  # noqa: BLE001 - fall back to the committed factors
                pass
        with open(os.path.join(ROOT, "profiles", PROFILE_TAG, "pmc_calibration.json")) as fh:
            c = json.load(fh)
        self.factors, self.basis = (c["fetch_factor"], c["write_factor"]), f"factors of profiles/{PROFILE_TAG}/pmc_calibration.json"

    def measure(self, roof, config_name, workload_args):
        """→ True when roof['traffic'] now holds this run's measurement"""
        if not self.why_not and time.time() + 10.0 > self.deadline:
            self.why_not = "the run's time budget for the counter passes is spent"
        if self.why_not:
            roof["traffic_live"] = "not measured: " + self.why_not
            return False
        t0 = time.time()
        self.timeout_s = max(5.0, min(self.timeout_s, self.deadline - time.time()))
        try:
            from tools.profile_configs import CONFIGS

            kernels = CONFIGS[config_name][0]
            self.calibrate()
            me = [sys.executable, os.path.join(ROOT, "bench.py")] + workload_args + ["--no-configs", "--no-cpu-baseline", "--no-live-traffic", "--steps", "3", "--warmup", "1"]
            sub = os.path.join(self.work, config_name)
            fetch = _pmc_pass("FETCH_SIZE", me, sub, self.timeout_s)
            write = _pmc_pass("WRITE_SIZE", me, sub, self.timeout_s)
            fb = wb = 0.0
            disp = 0
            for name, per_step in kernels.items():
                f = [v for k, v in fetch.items() if name in k]
                w = [v for k, v in write.items() if name in k]
                if not f or not w:
                    roof["traffic_live"] = f"not measured: no {name} dispatch in the counter files"
                    return False
                # (several instances of one kernel template: the dispatch-weighted mean, as tools/summarize_profiles.py takes it)
                fb += sum(v[0] * v[1] for v in f) / sum(v[1] for v in f) * 1024 * self.factors[0] * per_step
                wb += sum(v[0] * v[1] for v in w) / sum(v[1] for v in w) * 1024 * self.factors[1] * per_step
                disp += sum(v[1] for v in f)
            if roof.get("traffic") is not None:
                roof["traffic_quoted"] = roof["traffic"]  # (its source: `traffic_quoted_from`)
            roof["traffic"] = fb + wb
            roof["traffic_ratio"] = (fb + wb) / roof["algorithmic_bytes_per_step"] if roof.get("algorithmic_bytes_per_step") else None
            roof["traffic_fetch_write"] = [fb, wb]
            roof["traffic_live"] = (f"measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this workload after the timed region "
                                    f"({disp} dispatches per pass; fetch x {self.factors[0]:.4g}, write x {self.factors[1]:.4g}, {self.basis}; {time.time() - t0:.0f} s)")
            return True
        except Exception as e:  # noqa: BLE001 - the line must come out whatever the profiler does
            self.why_not = f"{type(e).__name__}: {str(e)[:120]}"  # … and no later config tries again
            roof["traffic_live"] = "not measured: " + self.why_not
            return False


AGG = None  # naive_query_engine_amd.AggregateFunc, filled in main (needs the package)


# ------------------------------------------------------------------------------------------------ aggregate workloads
# column generators: (name, synth kind, seed, modulus, base, dtype)
#   kind 0: row number; kind 1: base + splitmix64(seed + row) mod modulus; kind 2: Float64 in [0, 100)
def agg_shape(name, total, random_keys=False, groups=None):
    """→ dict(cols, aggs, key, pred(limit) builder, bytes per row, text) for one aggregate query shape"""
    from naive_query_engine_amd import Operator
    from naive_query_engine_amd.expression import binop, col, lit_f64, lit_i64

    idc = ("id", 1, 1, total, 0, "i64") if random_keys else ("id", 0, 0, 1, 0, "i64")
    five = lambda c: [(AGG.Count, c), (AGG.Sum, c), (AGG.Avg, c), (AGG.Min, c), (AGG.Max, c)]
    key_mod = lambda m: binop(col(0), Operator.Modulos, lit_i64(m))
    lt = lambda limit: binop(col(0), Operator.Lt, lit_i64(limit))
    if groups is not None and name == "no_minmax":   # … without min / max: 12-byte slots, one workgroup table up to 13632 keys (round 6)
        return dict(cols=[("k", 1, 7, groups, 0, "i64"), ("v", 2, 3, 1, 0, "f64")], aggs=[(AGG.Count, 1), (AGG.Sum, 1), (AGG.Avg, 1)], key=col(0), pred=None, bpr=16.0,
                    full=(groups, 1), text=f"select count(v),sum(v),avg(v) from t group by k; k random in [0, {groups})")
    if groups is not None:      # many distinct keys: key = a random Int64 column in [0, groups)
        return dict(cols=[("k", 1, 7, groups, 0, "i64"), ("v", 2, 3, 1, 0, "f64")], aggs=five(1), key=col(0), pred=None, bpr=16.0, full=(groups, 1),
                    text=f"select count(v),sum(v),avg(v),min(v),max(v) from t group by k; k random in [0, {groups})")
    if name == "v":             # the headline / C3: Float64 values
        return dict(cols=[idc, ("v", 2, 3, 1, 0, "f64")], aggs=five(1), key=key_mod(1024), pred=lt, bpr=16.0, full=(1024, 1),
                    text="select count(v),sum(v),avg(v),min(v),max(v) from t{w} group by id % 1024; t(id Int64, v Float64)")
    if name == "age":           # north_star's literal "10^9 Int64 rows": Int64 values (`val as f64` per row, sum.rs:86-101)
        return dict(cols=[idc, ("age", 1, 2, 60, 18, "i64")], aggs=five(1), key=key_mod(1024), pred=lt, bpr=16.0, full=(1024, 1),
                    text="select count(age),sum(age),avg(age),min(age),max(age) from t{w} group by id % 1024; t(id Int64, age Int64)")
    if name == "id":            # SURVEY §8d's single-column variant: key, predicate and value are ONE Int64 column = 8 B/row
        return dict(cols=[idc], aggs=five(0), key=key_mod(1024), pred=lt, bpr=8.0, full=(1024, 0),
                    text="select count(id),sum(id),avg(id),min(id),max(id) from t{w} group by id % 1024; t(id Int64)")
    if name == "three":         # C1's query shape (src/main.rs:36-40) at scale: three DIFFERENT value columns
        return dict(cols=[idc, ("age", 1, 2, 60, 18, "i64"), ("score", 2, 3, 1, 0, "f64")], aggs=[(AGG.Count, 0), (AGG.Sum, 1), (AGG.Avg, 2)],
                    key=key_mod(3), pred=None, bpr=24.0, full=(3, None), text="select count(id),sum(age),avg(score) from t group by id % 3; t(id Int64, age Int64, score Float64)")
    if name == "readme":        # the reference's own aggregate query (src/main.rs:36-40, README.md:105-111) at scale: three DIFFERENT value columns, one with min / max
        return dict(cols=[idc, ("age", 1, 2, 60, 18, "i64"), ("score", 2, 3, 1, 0, "f64")],
                    aggs=[(AGG.Count, 0), (AGG.Sum, 1), (AGG.Sum, 2), (AGG.Avg, 2), (AGG.Max, 2), (AGG.Min, 2)], key=key_mod(3), pred=None, bpr=24.0, full=(3, None),
                    text="select count(id),sum(age),sum(score),avg(score),max(score),min(score) from t group by id % 3; t(id Int64, age Int64, score Float64)")
    if name == "vnull":         # SURVEY 8d's correctness run at full size: the headline with 1 % NULLs in v (validity bitmap: 1 bit/row more)
        return dict(cols=[idc, ("v", 2, 3, 1, 0, "f64")], aggs=five(1), key=key_mod(1024), pred=lt, bpr=16.125, nullable={1: (4, 100)}, full=(1024, None),
                    text="select count(v),sum(v),avg(v),min(v),max(v) from t{w} group by id % 1024; t(id Int64, v Float64 with 1 % NULLs: u(i,4) mod 100 = 0)")
    if name == "tree":          # a predicate that is neither a chain nor a list of compares: `v < 20 or id % 3 == 0`
        tree = lambda limit: binop(binop(col(1), Operator.Lt, lit_f64(20.0)), Operator.Or,
                                   binop(binop(col(0), Operator.Modulos, lit_i64(3)), Operator.Eq, lit_i64(0)))
        # (host_pred: the same predicate over the downloaded columns in numpy, for the full-size check — Float64 `<` and Int64 `%` / `=` are exact on both sides)
        return dict(cols=[idc, ("v", 2, 3, 1, 0, "f64")], aggs=five(1), key=key_mod(1024), pred=tree, bpr=16.0, full=(1024, None),
                    host_pred=lambda c: (c[1] < 20.0) | (c[0] % 3 == 0),
                    text="select count(v),sum(v),avg(v),min(v),max(v) from t where v < 20 or id % 3 = 0 group by id % 1024; t(id Int64, v Float64)")
    raise ValueError(name)


def device_validity(B, seed, mod, n, first):
    """LSB-first validity bitmap (padded to whole 64-bit words) of `n` rows on the device: row i is NULL when u(first + i, seed) mod `mod` == 0
    (SURVEY 8d's 1 % nulls: seed 4, mod 100).  Built in chunks with torch from the library's own generator."""
    torch = B.torch
    nbytes = (n + 63) // 64 * 8
    out = torch.zeros(nbytes, dtype=torch.uint8, device=B.dev)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=B.dev)
    chunk = 1 << 27
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = B.synth(1, seed, m, first + lo, mod, 0)
        bits = (u != 0)
        del u
        if m % 8:
            bits = torch.cat([bits, torch.zeros(8 - m % 8, dtype=torch.bool, device=B.dev)])
        out[lo // 8: lo // 8 + (m + 7) // 8] = (bits.view(-1, 8).to(torch.uint8) * w).sum(dim=1, dtype=torch.uint8)
        del bits
    torch.cuda.synchronize()
    return out


def wl_aggregate(B, rows, with_filter, random_keys, steps, warmup, groups=None, exchange=True, shape="v", blocks=1, cold=False):
    """headline / C3 / their Int64-value, single-column, three-column and tree-predicate forms / many-groups aggregates"""
    from naive_query_engine_amd import DType

    n, total, first = rows, rows * B.world, B.rank * rows
    torch = B.torch
    sh = agg_shape(shape, total, random_keys, groups)
    fields = [F(c[0]) for c in sh["cols"]]
    tens = [B.synth(kind, seed, n, first, mod, base, dtype=torch.float64 if dt == "f64" else torch.int64) for (_, kind, seed, mod, base, dt) in sh["cols"]]
    valid = {j: device_validity(B, seed, mod, n, first) for j, (seed, mod) in sh.get("nullable", {}).items()}
    table = B.ctx.table_from_device([(DType.FLOAT64 if c[5] == "f64" else DType.INT64, n, t.data_ptr(), valid[j].data_ptr() if j in valid else None)
                                     for j, (c, t) in enumerate(zip(sh["cols"], tens))])
    key = sh["key"].flatten(fields)
    use_pred = with_filter and sh["pred"] is not None
    pred = sh["pred"](int(total * B.args.pass_frac)).flatten(fields) if use_pred else None

    def step():
        if B.comm is not None and exchange:
            return B.comm.sharded_aggregate(table, sh["aggs"], group_nodes=key, pred_nodes=pred)
        return B.ctx.aggregate(table, sh["aggs"], group_nodes=key, pred_nodes=pred)

    cold_ms = B.cold(step) if cold else None
    if shape == "tree":  # steady state = the run-time specialised streaming kernel (compiled on a worker thread during the first execution)
        r = step()
        del r
        B.ctx.jit_wait()
    ms, kernels, spread = B.timed(step, steps, warmup, blocks)
    where = " where id < N/2" if (use_pred and shape != "tree") else ""
    desc = sh["text"].replace("{w}", where) + f"; id = {'random' if random_keys else 'row number'}; {n} rows per GPU"
    # (agg_range: the range tier's transposing tail — it is part of the partitioned path's data flow, so it counts; round 4's dense tail did not)
    names = ["agg_grouped", "agg_merge_partials", "agg_fold_partials", "agg_partition", "agg_slab", "agg_segments", "agg_range", "agg_subpartition", "agg_sample", "expr_tree", "keep_from"]
    res = {"metric": "filter_hash_aggregate_rows_per_s" if use_pred else "hash_aggregate_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s",
           "ms_per_step": ms, "spread": spread, "cold_ms": cold_ms, "workload": desc, "rows_per_gpu": n, "roofline": roofline(sh["bpr"] * n, kernels, names)}
    return res, dict(table=table, tens=tens, valid=valid, sh=sh, key=key, fields=fields, n=n, total=total, use_pred=use_pred, random_keys=random_keys, groups=groups)


def parity_aggregate(B, st, sample_rows):
    """the same query on the first `sample_rows` rows: GPU (the same tensors, a prefix table) vs the oracle; also the cpu_baseline"""
    import numpy as np

    from naive_query_engine_amd import Column, DType
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    sh = st["sh"]
    host = []
    for (_, kind, seed, mod, base, dt) in sh["cols"]:
        a = orc.synth_fill(kind, seed, 0, m, mod, base)
        host.append(a.view(np.float64) if dt == "f64" else a.view(np.int64))
    masks = {j: orc.synth_fill(1, seed, 0, m, mod, 0).view(np.int64) != 0 for j, (seed, mod) in sh.get("nullable", {}).items()}
    h = orc.upload([[Column.from_numpy(a, masks.get(j)) for j, a in enumerate(host)]])
    # `id < limit`: half of the id range of the SAMPLE (sorted ids: the prefix's own range; random ids: drawn from [0, total))
    plimit = st["total"] // 2 if st["random_keys"] else m // 2
    pred = sh["pred"](plimit).flatten(st["fields"]) if st["use_pred"] else None
    t0 = time.perf_counter()
    ref = orc.aggregate(h, sh["aggs"], group_nodes=st["key"], pred_nodes=pred)[0]
    dt_s = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.FLOAT64 if c[5] == "f64" else DType.INT64, m, t.data_ptr(), st["valid"][j].data_ptr() if j in st.get("valid", {}) else None)
                                      for j, (c, t) in enumerate(zip(sh["cols"], st["tens"]))])
    got = B.ctx.aggregate(prefix, sh["aggs"], group_nodes=st["key"], pred_nodes=pred).to_host()
    g = np.stack([c.to_numpy().astype(np.float64) for c in got], axis=1)
    e = np.stack([c.to_numpy().astype(np.float64) for c in ref], axis=1)
    ok = g.shape == e.shape
    if ok:
        g, e = g[np.lexsort(g.T[::-1])], e[np.lexsort(e.T[::-1])]
        exact = [i for i, (f, _) in enumerate(sh["aggs"]) if f == AGG.Count]
        ok = bool(all((g[:, i] == e[:, i]).all() for i in exact) and np.allclose(g, e, rtol=1e-9, atol=0))
    cpu = {"value": m / dt_s, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"same query on the first {m} rows (single thread; the reference is single-threaded; host has {os.cpu_count()} cores)", "seconds": dt_s}
    return {"rows": m, "ok": ok, "groups": int(e.shape[0]), "tolerance": "counts exact, f64 rtol 1e-9"}, cpu


def parity_aggregate_full(B, st, threads):
    """EVERY row of the config (task 1): the device columns are downloaded in chunks of 2^27 rows and aggregated on the host
    by oracle/nqe_oracle.cpp: orc_grouped_parallel (per-thread direct-mapped tables over row ranges — checked against the
    reference-faithful port in tests/test_oracle_golden.py::test_parallel_grouped_form_matches_the_port), the chunks merged, and every
    group compared with the GPU's result over the whole table: keys and counts exact, min / max exact, sum / avg within 1e-9."""
    import numpy as np

    from oracle import oracle as orc

    from naive_query_engine_amd import AggregateFunc as A

    sh, n = st["sh"], st["n"]
    modulus = sh["full"][0]
    vcols = sorted({c for _, c in sh["aggs"]})                  # every value column the query aggregates (k of them: oracle.grouped_columns_parallel)
    host_pred = sh.get("host_pred") if st["use_pred"] else None
    limit = int(st["total"] * B.args.pass_frac) if (st["use_pred"] and host_pred is None) else None
    bitmaps = {j: t.cpu().numpy() for j, t in st.get("valid", {}).items()}   # the validity bitmaps the GPU reads (LSB first)
    t0 = time.perf_counter()
    parts, cpu_s = [], 0.0
    for lo in range(0, n, 1 << 27):
        hi = min(n, lo + (1 << 27))
        ids = st["tens"][0][lo:hi].cpu().numpy()
        cols = {c: (ids if c == 0 else st["tens"][c][lo:hi].cpu().numpy()) for c in vcols}
        masks = {j: np.unpackbits(b[lo // 8:(hi + 7) // 8], bitorder="little")[:hi - lo] for j, b in bitmaps.items()}
        rows_in = masks.get(0)
        if host_pred is not None:   # a predicate the CPU form does not know: evaluated here, handed over as the mask of the rows that take part
            every = {j: (cols[j] if j in cols else (ids if j == 0 else st["tens"][j][lo:hi].cpu().numpy())) for j in range(len(sh["cols"]))}
            rows_in = host_pred(every) if rows_in is None else (rows_in.astype(bool) & host_pred(every))
        t1 = time.perf_counter()
        parts.append(orc.grouped_columns_parallel(ids, cols, limit, modulus, threads, valid={c: masks[c] for c in vcols if c in masks}, id_valid=rows_in))
        cpu_s += time.perf_counter() - t1
        del ids, cols, masks
    live, exp = orc.finalize_grouped(orc.merge_grouped_columns(parts), sh["aggs"])
    pred = sh["pred"](int(st["total"] * B.args.pass_frac)).flatten(st["fields"]) if st["use_pred"] else None
    out, keys = B.ctx.aggregate(st["table"], sh["aggs"], group_nodes=st["key"], pred_nodes=pred, with_keys=True)
    got = [c.to_numpy() for c in out.to_host()]
    k = keys.to_host()[0].to_numpy()
    ok = bool(len(k) == len(live) and (k == live).all())
    if ok:
        for (func, _), g, e in zip(sh["aggs"], got, exp):
            if func in (A.Count, A.Min, A.Max):
                ok = ok and bool((g.astype(np.float64) == e).all())
            else:
                ok = ok and bool(np.allclose(g, e, rtol=1e-9, atol=0, equal_nan=True))
    return {"rows": n, "ok": ok, "groups": int(len(live)), "value_columns": len(vcols), "nullable_columns": sorted(bitmaps), "tolerance": "keys, counts, min, max exact; sum, avg rtol 1e-9",
            "against": f"orc_grouped_parallel (k value columns, validity) over the downloaded device columns, {threads} threads (itself checked against the single-threaded port)",
            "cpu_seconds": r4(cpu_s), "seconds": r4(time.perf_counter() - t0)}


def parity_aggregate_both(B, st, sample_rows, threads):
    """→ (parity_checked, cpu_baseline): the reference-faithful port on a sample (the CPU baseline, and the pin of the semantics) AND,
    for the shapes the parallel CPU form covers, every row of the config at full size.  `rows` is what was compared with an independent
    CPU result: the full row count when the full-size check ran."""
    par, cpu = parity_aggregate(B, st, sample_rows)
    if "full" in st["sh"] and st["n"] > par["rows"] and B.world == 1:
        full = parity_aggregate_full(B, st, threads)
        par = {"rows": full["rows"], "ok": bool(par["ok"] and full["ok"]), "groups": full["groups"], "tolerance": full["tolerance"], "full_size": full,
               "port_sample": {"rows": par["rows"], "ok": par["ok"], "groups": par["groups"]}}
    return par, cpu


def cpu_multicore_headline(st, threads, sample_rows):
    """SURVEY 8d's optional last row: an OPTIMISED multi-core CPU form of the headline (oracle/nqe_oracle.cpp: orc_headline_parallel — per-thread
    direct-mapped tables over contiguous row ranges, merged), so that the GPU/CPU ratio is not only against the reference's naive
    single-threaded design.  Clearly NOT the reference's algorithm; `cpu_baseline` stays the port."""
    import numpy as np

    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    ids = orc.synth_fill(0, 0, 0, m).view(np.int64)
    v = orc.synth_fill(2, 3, 0, m).view(np.float64)
    orc.headline_parallel(ids[:1 << 20], v[:1 << 20], m // 2, 1024, threads)  # thread pool / page warm-up
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        orc.headline_parallel(ids, v, m // 2, 1024, threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": r4(m / best), "unit": "rows/s", "cores": threads, "kind": "optimised multi-core form (not the reference's algorithm)",
            "sample": f"headline on the first {m} rows, best of 3; host has {os.cpu_count()} cores", "seconds": r4(best)}


# ------------------------------------------------------------------------------------------------ C2
def wl_c2(B, rows, steps, warmup, random_ids=False, blocks=1, cold=False):
    from naive_query_engine_amd import DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64

    n, total, first = rows, rows * B.world, B.rank * rows
    ids = B.synth(1, 1, n, first, total, 0) if random_ids else B.synth(0, 0, n, first)
    age = B.synth(1, 2, n, first, 60, 18)
    table = B.ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None)])
    fields = [F("id"), F("age")]
    pred = binop(col(0), Operator.Lt, lit_i64(total // 2)).flatten(fields)
    proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(fields)]
    step = lambda: B.ctx.selection_projection(table, pred, proj)
    cold_ms = B.cold(step) if cold else None
    ms, kernels, spread = B.timed(step, steps, warmup, blocks)
    names = ["select_fused", "keep_from_simple", "scan_single", "scan_chunk", "scan_add", "compact_expr"]  # (the scan of the tile counts runs every step: 9.5 us of ~285)
    # 16 B/row read + 8 B per kept row written = 20 B/row at 50 %.  The compaction does not read the source words of
    # 4096-row tiles in which nothing was kept: with ids = row numbers the kept rows are the first half, so only that half of `age`
    # moves (16 B/row) — §8d forbids taking that as skip credit, so for sorted ids `frac` is quoted on the bytes that MOVE and the
    # 20 B/row figure is kept as `frac_8d`; with random ids every tile holds kept rows and the two coincide.
    algo = 20.0 * n
    phys = algo if random_ids else 16.0 * n
    roof = roofline(algo, kernels, names, phys_bytes=phys)
    if not random_ids:
        roof["frac_8d"] = roof["frac"]
        roof["frac"] = roof["frac_physical"]
        roof["achieved"] = roof["frac"] * HBM_PEAK_GBS
        roof["note"] = "sorted ids: tiles without kept rows are not read; frac counts the 16 B/row that move, frac_8d SURVEY 8d's 20 B/row"
    res = {"metric": "filter_project_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms, "spread": spread, "cold_ms": cold_ms,
           "workload": f"select age + 100 from t where id < N/2; t(id Int64 {'random' if random_ids else 'row number'}, age Int64), {n} rows per GPU",
           "rows_per_gpu": n, "roofline": roof}
    return res, dict(ids=ids, age=age, n=n, total=total, proj=proj, fields=fields, random_ids=random_ids)


def parity_c2(B, st, sample_rows):
    import numpy as np

    from naive_query_engine_amd import Column, DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    ids = (orc.synth_fill_mt(1, 1, 0, m, st["total"], 0) if st["random_ids"] else orc.synth_fill_mt(0, 0, 0, m)).view(np.int64)
    x = orc.synth_fill_mt(1, 2, 0, m, 60, 18).view(np.int64)
    h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(x)]])
    pred = binop(col(0), Operator.Lt, lit_i64(st["total"] // 2 if st["random_ids"] else m // 2)).flatten(st["fields"])
    t0 = time.perf_counter()
    sel = orc.selection(h, pred, raw=True)
    ref = orc.projection(sel, st["proj"])[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["ids"].data_ptr(), None), (DType.INT64, m, st["age"].data_ptr(), None)])
    got = B.ctx.selection_projection(prefix, pred, st["proj"]).to_host()
    ok = len(got) == len(ref) == 1 and got[0].length == ref[0].length and bool((got[0].to_numpy() == ref[0].to_numpy()).all())
    cpu = {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port", "sample": f"same query on the first {m} rows (single thread)", "seconds": dt}
    return {"rows": m, "ok": ok, "output_rows": int(ref[0].length), "tolerance": "bit-exact"}, cpu



def wl_c2_tree(B, rows, steps, warmup, blocks=1, cold=False):
    """C2's shape with expression TREES on both sides: `select v * v + v / 4, id from t where (id + 1) % 10 < 5` (50 % pass, spread
    evenly).  Steady state = the run-time specialised kernels (csrc/expr_jit.hpp): the predicate as straight-line code into a
    Boolean column, the whole projection list in one pass over the kept rows; `cold_ms` is the first execution, which interprets
    (compilation runs on a worker thread)."""
    from naive_query_engine_amd import DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_f64, lit_i64

    n, first = rows, B.rank * rows
    ids = B.synth(0, 0, n, first)
    v = B.synth(2, 3, n, first, dtype=B.torch.float64)
    table = B.ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
    fields = [F("id"), F("v")]
    pred = binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5)).flatten(fields)
    proj = [binop(binop(col(1), Operator.Multiply, col(1)), Operator.Plus, binop(col(1), Operator.Divide, lit_f64(4.0))).flatten(fields), col(0).flatten(fields)]
    step = lambda: B.ctx.selection_projection(table, pred, proj)
    cold_ms = B.cold(step) if cold else None
    r = step()
    del r
    B.ctx.jit_wait()
    ms, kernels, spread = B.timed(step, steps, warmup, blocks)
    names = ["select_project_jit", "expr_jit", "proj_jit", "keep_from_pred", "keep_from_simple", "keep_jit", "compact_jit", "scan_single", "scan_chunk", "scan_add", "expr_tree", "expr_tree_compact", "compact_column", "compact_expr"]
    algo = 24.0 * n  # 16 B/row read (id, v) + two 8-byte columns written for half of the rows
    one_pass = any(k.startswith("select_project_jit") for k in kernels)  # predicate, compaction and projection list in ONE kernel: every column read once
    roof = roofline(algo, kernels, names, phys_bytes=algo if one_pass else algo + 8.0 * n + n / 4.0)  # two kernels: + id read by both, the Boolean column written and read
    specialised = one_pass or any(k.startswith(("expr_jit", "proj_jit")) for k in kernels)
    res = {"metric": "filter_project_rows_per_s", "value": rows * B.world / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms, "spread": spread, "cold_ms": cold_ms,
           "workload": f"select v * v + v / 4, id from t where (id + 1) % 10 < 5; t(id Int64 row number, v Float64), {n} rows per GPU; run-time specialised kernels: {specialised}, one pass: {one_pass}",
           "rows_per_gpu": n, "roofline": roof}
    return res, dict(ids=ids, v=v, n=n, pred=pred, proj=proj, fields=fields)


def parity_c2_tree(B, st, sample_rows):
    import numpy as np

    from naive_query_engine_amd import Column, DType
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    ids = orc.synth_fill_mt(0, 0, 0, m).view(np.int64)
    v = orc.synth_fill_mt(2, 3, 0, m).view(np.float64)
    h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(v)]])
    t0 = time.perf_counter()
    sel = orc.selection(h, st["pred"], raw=True)
    ref = orc.projection(sel, st["proj"])[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["ids"].data_ptr(), None), (DType.FLOAT64, m, st["v"].data_ptr(), None)])
    got = B.ctx.selection_projection(prefix, st["pred"], st["proj"]).to_host()
    ok = len(got) == len(ref) == 2 and all(g.length == r.length and bool((g.to_numpy().view(np.uint64) == r.to_numpy().view(np.uint64)).all()) for g, r in zip(got, ref))
    cpu = {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port", "sample": f"same query on the first {m} rows (single thread)", "seconds": dt}
    return {"rows": m, "ok": ok, "output_rows": int(ref[0].length), "tolerance": "bit-exact (Float64 bit patterns)"}, cpu


# ------------------------------------------------------------------------------------------------ C4
def make_join_data(B, rows, nb, variant, first):
    """dim(id, attr) = LEFT/build, fact(key, val) = RIGHT/probe.
    dense:   id = a permutation of 0..nb-1 (a primary key; every fact row matches once);
    wide:    the same with an attribute spanning 2^62 (the key-ordered payload table cannot be bit-packed);
    sparse:  unique ids spread over a 2^20 x nb domain (the general hashed path);
    dup:     every build key FOUR times (nb/4 distinct keys), fact keys drawn from [0, nb): a quarter matches, 4 output rows each
             (the reference's general case: chains of build rows per key, hash_join.rs:86-101);
    partial: a gap-free primary key, fact keys drawn from [0, nb/0.9): 90 % of the foreign keys match."""
    torch = B.torch
    g = torch.Generator(device=B.dev).manual_seed(7)
    perm = torch.randperm(nb, device=B.dev, generator=g).to(torch.int64)
    attr = B.synth(1, 4, nb, 0, (1 << 62) if variant == "wide" else (1 << 20), 0)
    fdom = int(math.ceil(nb / 0.9)) if variant == "partial" else nb
    fidx = B.synth(1, 5, rows, first, fdom, 0)  # which dim key a fact row references
    if variant == "sparse":
        # unique by construction: key j = j * 2^20 + (u(j, 11) mod 2^20): span 2^20 x nb, far beyond the direct-address limit
        dom = (torch.arange(nb, device=B.dev, dtype=torch.int64) << 20) + B.synth(1, 11, nb, 0, 1 << 20, 0)
        dkey = dom[perm].contiguous()          # build rows in shuffled order
        fkey = dom[fidx].contiguous()
    elif variant == "dup":
        dkey = (perm % (nb // 4)).contiguous()
        fkey = fidx
    else:
        dkey = perm
        fkey = fidx
    val = B.synth(2, 3, rows, first, dtype=torch.float64)
    torch.cuda.synchronize()
    return dkey, attr, fkey, val


def wl_c4(B, rows, nb, variant, steps, warmup, gather=False, blocks=1, cold=False, immutable=False):
    from naive_query_engine_amd import DType

    torch = B.torch
    n, first = rows, B.rank * rows
    dkey, attr, fkey, val = make_join_data(B, n, nb, variant, first)
    dim = B.ctx.table_from_device([(DType.INT64, nb, dkey.data_ptr(), None), (DType.INT64, nb, attr.data_ptr(), None)])
    # the fact table is BORROWED torch memory: by default every output column is written (outputs are the library's own memory,
    # SURVEY 8b); `immutable` = created with NQE_TABLE_IMMUTABLE (the caller keeps it alive and unmodified, as Arc-shared Arrow
    # buffers are in the reference): an all-match join may then hand the probe table's own columns on
    fact = B.ctx.table_from_device([(DType.INT64, n, fkey.data_ptr(), None), (DType.FLOAT64, n, val.data_ptr(), None)], immutable=immutable, keepalive=(fkey, val))
    # the first HashJoin::execute of the process: build + probe, nothing remembered
    cold_ms = B.cold(lambda: B.ctx.hash_join(dim, fact, 0, 0)) if cold else None
    # build (HashJoin::build, hash_join.rs:124-166): timed on its own — replicated on every rank, once per query
    build_ms, _, _ = B.timed(lambda: B.ctx.hash_join_build(dim, 0), max(3, steps // 2), 1, spread_blocks=0)
    jt = B.ctx.hash_join_build(dim, 0)

    def probe():
        if B.comm is not None and gather:  # C5: ordered variable-length all-gather of the per-rank outputs
            return B.comm.sharded_hash_join_probe(jt, fact, 0, gather=True)
        return B.ctx.hash_join_probe(jt, fact, 0)

    r0 = probe()
    out_rows = int(r0.num_rows)
    del r0
    ms, kernels, spread = B.timed(probe, steps, warmup, blocks)
    names = ["join_probe", "join_fused_write", "compact_gather", "compact_column"]
    total = n * B.world
    # 16 B/probe row read + 32 B/output row written + the build side once.  The output's two key columns are one shared
    # buffer when the join can alias them (unique build keys): then 24 B/output row are written
    m_out = out_rows if not gather else out_rows // B.world
    algo = 16.0 * n + 32.0 * m_out + 16.0 * nb
    phys = 16.0 * n + 24.0 * m_out + 16.0 * nb
    # unique build keys and EVERY probe row matches (output row = probe row): the probe-side columns of the output are the probe
    # table's own buffers (tables are immutable, columns may share buffers) — the probe keys are read and the build payload written,
    # nothing else moves.  `frac` is then quoted on those bytes and SURVEY 8d's figure kept as `frac_8d` (as for C2's skipped tiles)
    shared_probe = immutable and variant in ("dense", "wide", "sparse") and m_out == n and not os.environ.get("NQE_JOIN_NO_SHARED_PROBE_COLUMNS")
    if shared_probe:
        phys = 8.0 * n + 8.0 * m_out + 16.0 * nb
    kms = kernel_ms(kernels, names)
    extra = {"probe_kernels_ms": kms, "build_ms": build_ms, "probe_ms": ms, "execute_ms": ms + build_ms, "output_rows": m_out,
             "frac_end_to_end": algo / ((ms + build_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "frac = SURVEY 8d's bytes (16 B/probe row + 32 B/output row) over the probe kernels; frac_physical = with the shared key "
                     "column written once (24 B/output row); frac_end_to_end = 8d bytes over build + probe wall time (HashJoin::execute)"}
    if variant == "partial":
        # what the reference's call pattern costs: HashJoin::execute = a fresh build + probe per call (nqe_hash_join_execute, the entry
        # point integration/rust/gpu.rs calls), against the remembered two-pass form of the reused join table above
        ex_ms, _, exs = B.timed(lambda: B.ctx.hash_join(dim, fact, 0, 0), steps, 1, blocks)
        extra.update({"execute_call_ms": ex_ms, "execute_call_ms_min": exs["ms_min"], "execute_call_ms_max": exs["ms_max"],
                      "two_pass_ms": ms + build_ms, "execute_over_two_pass": ex_ms / (ms + build_ms)})
    gather_check = None
    if B.comm is not None and gather:
        # no oracle at this size: size-independent properties of the GATHERED output on every rank — every probe row matches once,
        # so the output has n x world rows in rank order: this rank's slice of the fact-key column is its own fact keys, the whole
        # column sums to the sum of every rank's keys, and the dim key column equals the fact key column row by row
        from naive_query_engine_amd.parallel import table_columns_as_tensors

        out_t = probe()
        B.ctx.synchronize()
        ok = out_t.num_rows == n * B.world and out_t.num_columns == 4
        if ok:
            cols_t = table_columns_as_tensors(out_t, B.dev)
            mine = cols_t[2][B.rank * n:(B.rank + 1) * n]
            ok = bool(torch.equal(mine, fkey)) and bool(torch.equal(cols_t[0], cols_t[2]))
            sums = torch.stack([fkey.sum(), val.view(torch.int64).sum()]).to(B.coll_dev)
            if B.distributed:
                B.dist.all_reduce(sums)  # int64 sums wrap identically everywhere
            ok = ok and int(cols_t[2].sum()) == int(sums[0]) and int(cols_t[3].sum()) == int(sums[1])
            del cols_t, mine
        okt = torch.tensor([1 if ok else 0], device=B.coll_dev)
        if B.distributed:
            B.dist.all_reduce(okt, op=B.dist.ReduceOp.MIN)
        gather_check = {"ok": bool(int(okt.item())), "rows": int(out_t.num_rows),
                        "what": "gathered join output on every rank: n x world rows, own slice == own fact keys, dim key == fact key, column sums == all-rank sums"}
        del out_t
        if not gather_check["ok"]:
            sys.stderr.write("bench.py: the gathered join output failed its check\n")
            sys.exit(3)
    shape = {"dense": "dense unique keys, attr of 20-bit range", "wide": "dense unique keys, attr of 62-bit range", "sparse": "sparse unique keys over a 2^40 domain",
             "dup": "every build key 4 times, 25 % of the fact keys match", "partial": "gap-free primary key, 90 % of the fact keys match"}[variant]
    res = {"metric": "hash_join_probe_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms, "spread": spread, "cold_ms": cold_ms,
           "workload": f"dim(id,attr) {nb} rows ({shape}; LEFT/build) join fact(key,val) {n} rows per GPU (RIGHT/probe) -> {m_out} rows x 4 columns"
                       f"{'; outputs all-gathered in rank order' if gather else ''}",
           "rows_per_gpu": n, "build_rows": nb, "roofline": roofline(algo, kernels, names, phys_bytes=phys, extra=extra)}
    if shared_probe:
        roof = res["roofline"]
        roof["frac_8d"] = roof["frac"]
        roof["frac"] = roof["frac_physical"]
        roof["achieved"] = roof["frac"] * HBM_PEAK_GBS
        roof["note"] = ("every probe row matches: the output's probe-side columns share the probe table's buffers; frac counts the bytes that move "
                        "(8 B/row of keys read + 8 B/row of build payload written), frac_8d SURVEY 8d's 16 + 32 B/row; " + roof["note"])
    if gather_check:
        res["gather_check"] = gather_check
    return res, dict(dim=dim, jt=jt, dkey=dkey, attr=attr, fkey=fkey, val=val, n=n, nb=nb, fact=fact, variant=variant)


def parity_c4(B, st, sample_rows):
    import numpy as np

    from naive_query_engine_amd import Column, DType
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    left = [Column.from_numpy(st["dkey"].cpu().numpy()), Column.from_numpy(st["attr"].cpu().numpy())]
    right = [Column.from_numpy(st["fkey"][:m].cpu().numpy()), Column.from_numpy(st["val"][:m].cpu().numpy())]
    t0 = time.perf_counter()
    ref = orc.hash_join([left], [right], 0, 0)[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["fkey"].data_ptr(), None), (DType.FLOAT64, m, st["val"].data_ptr(), None)])
    same = lambda got: len(got) == len(ref) and all(g.length == r.length and bool((g.to_numpy().view(np.int64) == r.to_numpy().view(np.int64)).all()) for g, r in zip(got, ref))
    ok = same(B.ctx.hash_join_probe(st["jt"], prefix, 0).to_host())
    if st["variant"] == "partial":   # the one-call entry point too (fresh build, the optimistic form and its fallback)
        ok = ok and same(B.ctx.hash_join(st["dim"], prefix, 0, 0).to_host())
    cpu = {"value": m / dt, "unit": "probe rows/s", "cores": 1, "kind": "port",
           "sample": f"HashJoin build ({st['nb']} rows) + probe of the first {m} fact rows (single thread)", "seconds": dt}
    return {"rows": m, "ok": ok, "output_rows": int(ref[0].length), "tolerance": "bit-exact, row order included"}, cpu


def parity_c4_full(B, st, threads, sample_rows=5_000_000):
    """EVERY probe row of a C4 variant against the reference-faithful port (oracle HashJoin, hash_join.rs:124-254): the port itself on
    the first `sample_rows` probe rows, single thread (the cpu_baseline, as before), and then over ALL probe rows in `threads` chunks of
    consecutive probe rows, each chunk a complete build + probe of the port on a thread of its own — the output is probe-major and a
    probe row's matches depend on no other probe row, so the chunks' outputs concatenated ARE the port's output over the whole table.
    Compared with the GPU's output bit for bit, row order included."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    from naive_query_engine_amd import Column
    from oracle import oracle as orc

    par, cpu = parity_c4(B, st, sample_rows)
    n = st["n"]
    if n <= par["rows"] or B.world != 1:
        return par, cpu
    t0 = time.perf_counter()
    left = [Column.from_numpy(st["dkey"].cpu().numpy()), Column.from_numpy(st["attr"].cpu().numpy())]
    fkey, val = st["fkey"].cpu().numpy(), st["val"].cpu().numpy()
    got = [c.to_numpy().view(np.int64) for c in B.ctx.hash_join_probe(st["jt"], st["fact"], 0).to_host()]
    # (a large build side: fewer chunks at a time — every chunk builds the port's hash map of the whole build side)
    big_build = st["nb"] > 2_000_000
    threads = max(1, min(threads, 8 if big_build else threads))
    nchunks = threads if big_build else max(threads, 32)   # (a 10^7-row build side: one port build per thread — 32 of them took 97 s)
    bounds = [n * i // nchunks for i in range(nchunks + 1)]

    def one(i):
        lo, hi = bounds[i], bounds[i + 1]
        ref = orc.hash_join([left], [[Column.from_numpy(fkey[lo:hi]), Column.from_numpy(val[lo:hi])]], 0, 0)[0]
        return [r.to_numpy().view(np.int64) for r in ref]

    ok, at = len(got) == 4, 0
    with ThreadPoolExecutor(max_workers=threads) as pool:
        for ref in pool.map(one, range(nchunks)):
            m = len(ref[0])
            ok = ok and len(ref) == len(got) and at + m <= len(got[0]) and all(bool((g[at:at + m] == r).all()) for g, r in zip(got, ref))
            at += m
    ok = bool(ok and at == len(got[0]))
    full = {"rows": n, "ok": ok, "output_rows": int(at), "tolerance": "bit-exact, row order included",
            "against": f"the port's HashJoin over all probe rows in {nchunks} chunks of consecutive rows on {threads} threads (a probe row's matches depend on no other probe row)",
            "seconds": r4(time.perf_counter() - t0)}
    return ({"rows": n, "ok": bool(par["ok"] and ok), "output_rows": int(at), "tolerance": full["tolerance"], "full_size": full,
             "port_sample": {"rows": par["rows"], "ok": par["ok"]}}, cpu)


def parity_c4_property(B, st):
    """a build side the oracle cannot hold in minutes (10^8 rows): the whole output checked at full size against what it must be for
    a primary-key join in which every fact row matches once — computed independently with torch on the device: same row count and
    order (fact key and value columns bit-identical to the input), dim id == fact key, attr == the attribute stored under that key.
    (The partitioned build itself has oracle parity at reduced size: tests/test_gpu_parity.py::test_join_partitioned_dense_build.)"""
    from naive_query_engine_amd.parallel import table_columns_as_tensors

    torch = B.torch
    out_t = B.ctx.hash_join_probe(st["jt"], st["fact"], 0)
    B.ctx.synchronize()
    ok = out_t.num_rows == st["n"] and out_t.num_columns == 4
    if ok:
        cols_t = table_columns_as_tensors(out_t, B.dev)
        by_key = torch.empty(st["nb"], dtype=torch.int64, device=B.dev)
        by_key[st["dkey"]] = st["attr"]
        ok = (bool(torch.equal(cols_t[0], st["fkey"])) and bool(torch.equal(cols_t[2], st["fkey"])) and bool(torch.equal(cols_t[1], by_key[st["fkey"]]))
              and bool(torch.equal(cols_t[3].view(torch.int64), st["val"].view(torch.int64))))
        del cols_t, by_key
    del out_t
    return ({"rows": st["n"], "ok": ok, "output_rows": st["n"], "tolerance": "bit-exact, row order included",
             "what": "full-size property check (torch, device): order, dim id == fact key, attr == attribute under that key"},
            {"value": None, "unit": "probe rows/s", "cores": 1, "kind": "port", "sample": "none: a 10^8-row build side is minutes of oracle time"})


# ------------------------------------------------------------------------------------------------ the drop-in path and the ingest
def mem_available_bytes():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def upload_record(B, table, ncols, n):
    """what registering the table costs when it starts in HOST memory (MemTable::try_create over real
    RecordBatches: nqe_table_create with NQE_HOST columns).  The device-generated columns are downloaded to numpy arrays (pageable
    memory), uploaded as they are, then page-locked in place (hipHostRegister) and uploaded again.  All `ncols` 8-byte columns of the
    first `rows` rows; rows = the whole table when host memory allows (3x the table must be available), else 10^8.
    → (record, the library-owned device table of the LAST upload or None)"""
    import numpy as np

    from naive_query_engine_amd import Column

    torch = B.torch
    need = 3 * ncols * 8 * n
    rows = n if mem_available_bytes() >= need else min(n, 10**8)
    t0 = time.perf_counter()
    prefix = table if rows == n else B.ctx.slice(table, 0, rows)
    host = [prefix.download_column(i).to_numpy() for i in range(ncols)]
    dl_s = time.perf_counter() - t0
    del prefix
    nbytes = float(ncols * 8 * rows)
    cols = [Column.from_numpy(a) for a in host]
    rec = {"rows": rows, "bytes": nbytes, "download_GBps": r4(nbytes / dl_s / 1e9), "what": "nqe_table_create over NQE_HOST columns (numpy arrays), blocking; pinned = the same arrays page-locked in place"}
    up = None
    for kind in ("pageable", "pinned"):
        registered = []
        if kind == "pinned":
            try:
                rt = torch.cuda.cudart()
                for a in host:
                    if int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) != 0:
                        raise RuntimeError("hipHostRegister failed")
                    registered.append(a)
            except Exception as e:  # noqa: BLE001 - a host that cannot page-lock this much memory still reports the pageable figure
                rec["pinned_error"] = str(e)[:120]
                for a in registered:
                    rt.cudaHostUnregister(a.ctypes.data)
                break
        best = None
        for _ in range(2):
            del up
            up = None
            B.ctx.synchronize()
            t0 = time.perf_counter()
            up = B.ctx.table_from_host(cols)
            B.ctx.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec[f"{kind}_ms"] = r4(best * 1e3)
        rec[f"{kind}_GBps"] = r4(nbytes / best / 1e9)
        for a in registered:
            rt.cudaHostUnregister(a.ctypes.data)
    return rec, (up if rows == n else None)


def dropin_record(B, make_tree, raw_step, steps, warmup, blocks, same):
    """The reference's call pattern over the mirrors of its own operator surface (physical_plan.py): per step a FRESH operator tree as
    QueryPlanner::create_physical_plan builds it (planner/mod.rs:42-182), the rewrite pass, root.execute() (db.rs:34-36) — tables
    registered once, results left in HBM — beside the raw C-ABI call of the same query (`raw_ms`), timed back to back here."""
    from naive_query_engine_amd.rewrite import rewrite

    step = lambda: rewrite(make_tree()).execute()
    got, exp = step(), raw_step()
    B.ctx.synchronize()
    ok = bool(same(got[0].table, exp))
    nrows = exp.num_rows
    del got, exp
    raw_ms, _, _ = B.timed(raw_step, steps, warmup, blocks)
    ms, _, spread = B.timed(step, steps, warmup, blocks)
    return {"ms": r4(ms), "ms_min": r4(spread["ms_min"]), "ms_max": r4(spread["ms_max"]), "raw_ms": r4(raw_ms), "over_raw": r4(ms / raw_ms),
            "parity": {"ok": ok, "rows": int(nrows), "what": "rewrite(tree).execute() == the raw C-ABI call, bit for bit, at full size"}}


def same_tables_device(B):
    from naive_query_engine_amd.parallel import table_columns_as_tensors

    def same(a, b):
        if a.num_rows != b.num_rows or a.num_columns != b.num_columns:
            return False
        ca, cb = table_columns_as_tensors(a, B.dev), table_columns_as_tensors(b, B.dev)
        B.ctx.synchronize()
        return all(bool(B.torch.equal(x, y)) for x, y in zip(ca, cb))
    return same


def dropin_aggregate(B, table, sh, total, steps, warmup, blocks, with_filter=True):
    """dropin_headline: PhysicalAggregatePlan(SelectionPlan(ScanPlan(MemTable))) exactly as the planner builds it"""
    from naive_query_engine_amd import DType
    from naive_query_engine_amd.arrow_host import Field
    from naive_query_engine_amd.expression import ColumnExpr
    from naive_query_engine_amd.physical_plan import Avg, Count, Max, MemTable, Min, PhysicalAggregatePlan, ScanPlan, SelectionPlan, Sum

    schema = [Field(c[0], DType.FLOAT64 if c[5] == "f64" else DType.INT64, False) for c in sh["cols"]]
    mt = MemTable.from_device(schema, [table])
    ops = {AGG.Count: Count, AGG.Sum: Sum, AGG.Avg: Avg, AGG.Min: Min, AGG.Max: Max}
    pred_expr = sh["pred"](int(total * B.args.pass_frac)) if (with_filter and sh["pred"] is not None) else None

    def make_tree():
        below = ScanPlan.create(mt, None)
        if pred_expr is not None:
            below = SelectionPlan.create(below, pred_expr)
        return PhysicalAggregatePlan.create([sh["key"]], [ops[f].create(ColumnExpr.try_create(None, c)) for f, c in sh["aggs"]], below)

    fields = [F(c[0]) for c in sh["cols"]]
    key, pred = sh["key"].flatten(fields), (pred_expr.flatten(fields) if pred_expr is not None else None)
    raw = lambda: B.ctx.aggregate(table, sh["aggs"], group_nodes=key, pred_nodes=pred)
    def same(a, b):
        # f64 sums are accumulated with atomics in whatever order the waves arrive: two executions of the SAME call differ in the last
        # bits, so aggregates are compared the way they are against the oracle — counts exact, Float64 within 1e-9 relative
        import numpy as np

        ca, cb = [c.to_numpy() for c in a.to_host()], [c.to_numpy() for c in b.to_host()]
        return len(ca) == len(cb) and all(x.shape == y.shape and x.dtype == y.dtype and
                                          (bool((x == y).all()) if x.dtype != np.float64 else bool(np.allclose(x, y, rtol=1e-9, atol=0, equal_nan=True)))
                                          for x, y in zip(ca, cb))

    rec = dropin_record(B, make_tree, raw, steps, warmup, blocks, same)
    rec["parity"]["what"] = "rewrite(tree).execute() == the raw C-ABI call at full size: same groups in the same (key) order, counts exact, Float64 within 1e-9"
    return rec


def dropin_c2(B, rows, steps, warmup, blocks):
    from naive_query_engine_amd import DType, Operator
    from naive_query_engine_amd.arrow_host import Field
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from naive_query_engine_amd.physical_plan import MemTable, ProjectionPlan, ScanPlan, SelectionPlan

    n = rows
    ids, age = B.synth(0, 0, n, 0), B.synth(1, 2, n, 0, 60, 18)
    table = B.ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None)])
    schema = [Field("id", DType.INT64, False), Field("age", DType.INT64, False)]
    mt = MemTable.from_device(schema, [table])
    pred_e, proj_e = binop(col("id"), Operator.Lt, lit_i64(n // 2)), binop(col("age"), Operator.Plus, lit_i64(100))
    make_tree = lambda: ProjectionPlan.create(SelectionPlan.create(ScanPlan.create(mt, None), pred_e), [Field("age + 100", DType.INT64, True)], [proj_e])
    fields = [F("id"), F("age")]
    raw = lambda: B.ctx.selection_projection(table, pred_e.flatten(fields), [proj_e.flatten(fields)])
    rec = dropin_record(B, make_tree, raw, steps, warmup, blocks, same_tables_device(B))
    del ids, age
    return rec


def dropin_c4(B, rows, nb, steps, warmup, blocks):
    """HashJoin(ScanPlan(dim), ScanPlan(fact)) — a fresh tree per step, so a fresh build per execute (hash_join.rs:280-284) — beside
    nqe_hash_join_execute.  The tables are registered from HOST columns (library-owned memory, as a drop-in's tables are)."""
    from naive_query_engine_amd import Column, DType
    from naive_query_engine_amd.arrow_host import Field
    from naive_query_engine_amd.physical_plan import ColumnRef, HashJoin, JoinType, MemTable, ScanPlan

    dkey, attr, fkey, val = make_join_data(B, rows, nb, "dense", 0)
    dim = B.ctx.table_from_host([Column.from_numpy(dkey.cpu().numpy()), Column.from_numpy(attr.cpu().numpy())])
    fact = B.ctx.table_from_host([Column.from_numpy(fkey.cpu().numpy()), Column.from_numpy(val.cpu().numpy())])
    del dkey, attr, fkey, val
    ls = [Field("id", DType.INT64, False), Field("attr", DType.INT64, False)]
    rs = [Field("key", DType.INT64, False), Field("val", DType.FLOAT64, False)]
    ml, mr = MemTable.from_device(ls, [dim]), MemTable.from_device(rs, [fact])
    make_tree = lambda: HashJoin.create(ScanPlan.create(ml, None), ScanPlan.create(mr, None), [(ColumnRef(None, "id"), ColumnRef(None, "key"))], JoinType.Inner, ls + rs)
    raw = lambda: B.ctx.hash_join(dim, fact, 0, 0)
    return dropin_record(B, make_tree, raw, steps, warmup, blocks, same_tables_device(B))


# ------------------------------------------------------------------------------------------------ the compact record
def r4(x):
    """4 significant digits: the line must stay small"""
    if x is None:
        return None
    return float(f"{x:.4g}")


def compact(res):
    """a config's record in the line: numbers only (the text and the per-kernel map go to the details file)"""
    roof = res["roofline"]
    out = {"ms": r4(res["ms_per_step"]), "ms_min": r4(res["spread"]["ms_min"]), "ms_max": r4(res["spread"]["ms_max"]),
           "kernel_ms": r4(roof["kernel_ms_per_step"]), "frac": r4(roof["frac"]), "rows": res["rows_per_gpu"]}
    if roof.get("frac_physical") is not None:   # (only where the bytes that move differ from SURVEY 8d's: the line has a size limit)
        out["frac_physical"] = r4(roof["frac_physical"])
    for k in ("frac_step", "kernel_ms_min", "kernel_ms_max", "frac_8d", "frac_end_to_end", "build_ms", "execute_call_ms", "two_pass_ms", "execute_over_two_pass", "traffic_ratio"):
        if roof.get(k) is not None:
            out[k] = r4(roof[k])
    if res.get("cold_ms") is not None:
        out["cold_ms"] = r4(res["cold_ms"])
    if "gather_check" in res:
        out["gather_check"] = {"ok": res["gather_check"]["ok"], "rows": res["gather_check"]["rows"]}
    if "parity_checked" in res:
        p = res["parity_checked"]
        out["parity"] = {"ok": p["ok"], "rows": p["rows"]}
        out["cpu_rows_per_s"] = r4(res["cpu_baseline"]["value"])
    return out


def summary_of(out):
    s = {"headline": [r4(out["ms_per_step"]), r4(out["roofline"]["frac"]), r4(out["roofline"].get("frac_physical")),
                      (out.get("parity_checked") or {}).get("ok")]}
    for k, c in out.get("configs", {}).items():
        if "raw_ms" in c:      # a drop-in row: [ms, raw C-ABI ms, ms / raw ms, parity ok]
            s[k] = [c["ms"], c["raw_ms"], c["over_raw"], (c.get("parity") or {}).get("ok")]
        elif "ms" in c:
            s[k] = [c["ms"], c.get("frac"), c.get("frac_physical"), (c.get("parity") or {}).get("ok")]
    return s


def finish_line(out):
    """→ the JSON line, `summary` as its LAST key, shortened until it fits LINE_LIMIT (the details file holds what is shed)"""
    out.pop("summary", None)
    out["summary"] = summary_of(out)
    line = json.dumps(out, separators=(",", ":"))
    for k in ("rows", "ms_min", "ms_max", "cpu_rows_per_s", "cold_ms", "kernel_ms_min", "kernel_ms_max", "kernel_ms", "traffic_ratio", "execute_call_ms", "two_pass_ms"):
        if len(line) <= LINE_LIMIT:
            break
        for c in out.get("configs", {}).values():
            c.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def sharded_headline_check(B, st):
    """no oracle at this size: a size-independent check of the SHARDED result on every rank — ids are row numbers, so group g of
    `id % 1024` holds exactly the ids g, g + 1024, ... below total/2, and every value lies in [0, 100)"""
    import numpy as np

    rs, keys = B.comm.sharded_aggregate(st["table"], st["sh"]["aggs"], group_nodes=st["key"], pred_nodes=st["sh"]["pred"](st["total"] // 2).flatten(st["fields"]))
    cols = [c.to_numpy() for c in rs.to_host()]
    kk = keys.to_host()[0].to_numpy()
    half = st["total"] // 2
    exp_cnt = np.array([(half - g + 1023) // 1024 if g < half else 0 for g in range(1024)], dtype=np.uint64)
    ok = bool(len(kk) == 1024 and (kk == np.arange(1024)).all() and (cols[0] == exp_cnt).all() and (cols[3] >= 0).all() and (cols[4] < 100).all()
              and np.allclose(cols[2], cols[1] / cols[0].astype(np.float64), rtol=1e-12))
    okt = B.torch.tensor([1 if ok else 0], dtype=B.torch.int64, device=B.coll_dev)
    B.dist.all_reduce(okt, op=B.dist.ReduceOp.MIN)
    return {"ok": bool(int(okt.item())), "what": "sharded headline on every rank: 1024 keys, analytic counts, avg = sum / count, min/max in [0, 100)"}


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a different configuration than asked\n")
        sys.exit(2)
    import torch

    if torch.cuda.device_count() <= local_rank and os.environ.get("NQE_BENCH_TRANSPORT") != "host":
        sys.stderr.write(f"bench.py: rank {rank} needs device {local_rank}, this box has {torch.cuda.device_count()}\n")
        sys.exit(2)

    from naive_query_engine_amd import AggregateFunc

    global AGG
    AGG = AggregateFunc
    B = Bench(args, world, rank, local_rank)
    wl = args.workload
    default_rows = {"headline": 10**9, "headline_int64": 10**9, "headline_single": 10**9, "agg3": 10**9, "agg_readme": 10**9, "headline_nullable": 10**9, "tree_pred": 10**9,
                    "c3": 10**9}.get(wl, 10**8)
    n = args.rows or default_rows
    want_cpu = world == 1 and not args.no_cpu_baseline
    par_threads = args.cpu_threads if args.cpu_threads > 0 else max(1, min(64, os.cpu_count() or 1))  # the full-size CPU form of the parity checks
    csteps, cwarm, cblocks = max(3, min(args.steps, 10)), 2, 3  # the side configs: three blocks of a few steps each

    # ---- the main line
    agg_shapes = {"headline": ("v", True), "c3": ("v", False), "headline_int64": ("age", True), "headline_single": ("id", True), "agg3": ("three", False),
                  "tree_pred": ("tree", True), "agg_readme": ("readme", False), "headline_nullable": ("vnull", True)}
    if wl in agg_shapes:
        shape, filt = agg_shapes[wl]
        res, st = wl_aggregate(B, n, filt, args.random_keys, args.steps, args.warmup, shape=shape, cold=True)
        par = parity_aggregate_both(B, st, args.cpu_sample_rows if wl == "headline" else 20_000_000, par_threads) if want_cpu else None
        name = {"headline_int64": "headline_int64_values", "headline_single": "headline_single_column", "agg3": "agg_three_value_columns", "agg_readme": "agg_readme_shape",
                "tree_pred": "agg_tree_predicate"}.get(wl, wl) + ("_random_keys" if args.random_keys else "")
    elif wl == "agg_groups":
        res, st = wl_aggregate(B, n, False, False, args.steps, args.warmup, groups=args.groups, shape="no_minmax" if args.no_minmax else "v")
        par = parity_aggregate_both(B, st, min(args.cpu_sample_rows, 20_000_000), par_threads) if want_cpu else None
        name = f"agg_{args.groups}_groups" + ("_count_sum_avg" if args.no_minmax else "")
    elif wl == "c2_tree":
        res, st = wl_c2_tree(B, n, args.steps, args.warmup)
        par = parity_c2_tree(B, st, n) if want_cpu else None
        name = "c2_expression_trees"
    elif wl in ("c2", "c2_random"):
        res, st = wl_c2(B, n, args.steps, args.warmup, random_ids=wl == "c2_random")
        par = parity_c2(B, st, n) if want_cpu else None
        name = "c2" if wl == "c2" else "c2_random_ids"
    else:
        variant = {"c4": "dense", "c4_sparse": "sparse", "c4_wide": "wide", "c4_dup": "dup", "c4_partial": "partial"}[wl]
        res, st = wl_c4(B, n, args.dim_rows, variant, args.steps, args.warmup, gather=args.gather, immutable=args.immutable)
        par = parity_c4(B, st, n if (wl == "c4" and args.dim_rows <= 10**6) else 5_000_000) if want_cpu else None
        name = {"c4": "c4", "c4_sparse": "c4_sparse_keys", "c4_wide": "c4_wide_payload", "c4_dup": "c4_dup_keys", "c4_partial": "c4_partial_match"}[wl]
        if args.immutable and wl == "c4":
            name = "c4_shared_probe_columns"
        if wl == "c4" and args.dim_rows != 10**6:
            name = {10**7: "c4_dim_1e7", 10**8: "c4_dim_1e8"}.get(args.dim_rows, name)
    attach_traffic(res["roofline"], name)
    attach_spread(res)
    n_main = res["rows_per_gpu"]
    details = {"main": {"name": name, "workload": res["workload"], "kernels": res["roofline"].pop("kernels")}, "configs": {}}
    res["roofline"].pop("note", None)
    out = {
        "metric": res["metric"], "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": res["workload"], "rows_per_gpu": n_main, "total_rows": n_main * world, "parallelism": f"row-range x{world}"},
        "roofline": res["roofline"],
    }
    if B.reserve_ms is not None:
        out["reserved"] = {"GB": args.reserve_gb, "ms": r4(B.reserve_ms)}  # nqe_ctx_reserve at start-up: outputs and scratch come from this block
    if res.get("cold_ms") is not None:
        out["cold_ms"] = r4(res["cold_ms"])  # the first execution of the query in the process (run_sql is one-shot: db.rs:24-37)
    if B.host_transport:
        out["ranks"] = B.dist.get_world_size()
        out["exchange"] = ("nqe_sharded_* (C ABI) over the HOST-STAGED transport (gloo; NQE_BENCH_TRANSPORT=host), every rank on device "
                           f"{B.local_rank}: a functional run of the multi-rank code, NOT a scaling measurement")
    elif B.distributed:
        out["rccl_ranks"] = B.comm.world  # the library's own communicator (asserted == --gpus when it was created)
        out["rccl_version"] = B.capi.Comm.rccl_version()
        out["exchange"] = "nqe_sharded_* (C ABI) on RCCL, collectives on the context's stream"
    if B.preflight:
        out["preflight"] = {k: B.preflight[k] for k in ("ok", "ranks", "seconds")}
    if par:
        out["parity_checked"], out["cpu_baseline"] = par
    if par and wl == "headline" and not args.random_keys and args.cpu_threads != 0:
        out["cpu_optimised_multicore"] = cpu_multicore_headline(st, args.cpu_threads if args.cpu_threads > 0 else min(64, os.cpu_count() or 1), args.cpu_sample_rows)
    if B.comm is not None and wl == "headline" and not args.random_keys and args.pass_frac == 0.5:
        out["result_check"] = sharded_headline_check(B, st)
    main_state = st if (wl == "headline" and not args.no_configs and not args.random_keys and world == 1) else None
    del st

    # ---- every other config, in the same line
    if wl == "headline" and not args.no_configs and not args.random_keys:
        cfg = {}
        only = set(x for x in args.only.split(",") if x)
        B.min_warm_s = 0.15

        def add(cname, fn, parity=None):
            # (no allocator trimming between configs: memory handed back to the driver and mapped again came back slower — C3 over
            # re-allocated columns ran 8-10 % below the same kernel over the process's first allocations)
            if only and cname not in only:
                return
            r, s = fn()
            attach_traffic(r["roofline"], cname)
            attach_spread(r)
            if parity and want_cpu:
                r["parity_checked"], r["cpu_baseline"] = parity(s)
            del s
            details["configs"][cname] = {"workload": r["workload"], "kernels": r["roofline"].get("kernels"), "spread": r["spread"], "cold_ms": r.get("cold_ms"),
                                         "roofline": {k: v for k, v in r["roofline"].items() if k != "kernels"},
                                         "parity_checked": r.get("parity_checked"), "cpu_baseline": r.get("cpu_baseline")}
            cfg[cname] = compact(r)

        kw = dict(blocks=cblocks, cold=True)
        pa = lambda rows: (lambda s: parity_aggregate_both(B, s, rows, par_threads))  # + every row at full size where the parallel CPU form covers the shape
        pj = lambda s: parity_c4_full(B, s, par_threads)  # every probe row against the port (chunks of probe rows on threads); the port on 5 x 10^6 rows single-threaded = the cpu_baseline
        pj_full = pj
        # (NQE_BENCH_MULTI_CONFIGS with NQE_FORCE_EXCHANGE: the multi-rank block on one rank through RCCL — a dry run of that code)
        if world == 1 and not (B.distributed and os.environ.get("NQE_BENCH_MULTI_CONFIGS")):
            add("c3", lambda: wl_aggregate(B, n, False, False, csteps, cwarm, **kw), pa(20_000_000))
            add("headline_random_keys", lambda: wl_aggregate(B, n, True, True, csteps, cwarm, **kw), pa(20_000_000))
            add("c3_random_keys", lambda: wl_aggregate(B, n, False, True, csteps, cwarm, **kw), pa(20_000_000))
            add("headline_int64_values", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, shape="age", **kw), pa(20_000_000))
            add("headline_single_column", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, shape="id", **kw), pa(20_000_000))
            add("agg_three_value_columns", lambda: wl_aggregate(B, n, False, False, csteps, cwarm, shape="three", **kw), pa(20_000_000))
            add("agg_readme_shape", lambda: wl_aggregate(B, n, False, False, csteps, cwarm, shape="readme", **kw), pa(20_000_000))
            add("headline_nullable", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, shape="vnull", **kw), pa(20_000_000))
            add("agg_tree_predicate", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, shape="tree", **kw), pa(20_000_000))
            add("c2", lambda: wl_c2(B, 10**8, csteps, cwarm, **kw), lambda s: parity_c2(B, s, s["n"]))
            add("c2_random_ids", lambda: wl_c2(B, 10**8, csteps, cwarm, random_ids=True, **kw), lambda s: parity_c2(B, s, s["n"]))
            add("c2_expression_trees", lambda: wl_c2_tree(B, 10**8, csteps, cwarm, **kw), lambda s: parity_c2_tree(B, s, s["n"]))
            add("c4", lambda: wl_c4(B, 10**8, 10**6, "dense", csteps, cwarm, **kw), pj_full)
            # the same join over a probe table the caller declared immutable (NQE_TABLE_IMMUTABLE): every probe row matches, so the output's
            # probe-side columns are the probe table's own buffers and only the keys are read and the build payload written
            add("c4_shared_probe_columns", lambda: wl_c4(B, 10**8, 10**6, "dense", csteps, cwarm, immutable=True, **kw), pj)
            add("c4_wide_payload", lambda: wl_c4(B, 10**8, 10**6, "wide", csteps, cwarm, **kw), pj)  # attr spans 2^62: an 8 MB payload table
            add("c4_dim_1e7", lambda: wl_c4(B, 10**8, 10**7, "dense", csteps, cwarm, **kw), pj)
            # a build side as large as the probe side (the partitioned dense build, >= 2^25 rows): build_ms is the number to read
            add("c4_dim_1e8", lambda: wl_c4(B, 10**8, 10**8, "dense", csteps, cwarm, **kw), lambda s: parity_c4_property(B, s))
            add("c4_sparse_keys", lambda: wl_c4(B, 10**8, 10**6, "sparse", csteps, cwarm, **kw), pj)
            add("c4_dup_keys", lambda: wl_c4(B, 10**8, 10**6, "dup", csteps, cwarm, **kw), pj)
            add("c4_partial_match", lambda: wl_c4(B, 10**8, 10**6, "partial", csteps, cwarm, **kw), pj)
            for G in (4096, 5000, 6000, 11000, 65536, 1 << 20):  # 5000: one directly addressed table without key words (round 6); 6000, 11000: two key subsets, the two halves of the range (up to 2 x 5840 keys)
                add(f"agg_{G}_groups", lambda G=G: wl_aggregate(B, 10**8, False, False, csteps, cwarm, groups=G, **kw), pa(10_000_000))
            # count / sum / avg only: no min / max arrays in the workgroup table — 12 bytes per slot, ONE table up to 13632 keys (round 6)
            add("agg_12000_groups_count_sum_avg", lambda: wl_aggregate(B, 10**8, False, False, csteps, cwarm, groups=12000, shape="no_minmax", **kw), pa(10_000_000))
        else:
            # the headline without its exchange (every rank aggregates its shard only): step time with and without
            add("headline_local_only", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, exchange=False, blocks=cblocks))
            out["exchange_ms_per_step"] = r4(out["ms_per_step"] - cfg["headline_local_only"]["ms"])
            # BASELINE's metric as worded — "10^9-row filter->hash-agg, 1/2/4/8 MI355X": the SAME 10^9 rows (NQE_BENCH_STRONG_ROWS for
            # functional runs) split over the ranks, exchange included = STRONG scaling, beside the weak-scaled main line (10^9 per GPU)
            strong_total = int(os.environ.get("NQE_BENCH_STRONG_ROWS", 10**9))
            strong_state = {}

            def strong():
                r, s_ = wl_aggregate(B, strong_total // world, True, False, csteps, cwarm, blocks=cblocks)
                if args.pass_frac == 0.5:
                    strong_state["check"] = sharded_headline_check(B, s_)
                return r, s_

            add("headline_strong", strong)
            if "headline_strong" in cfg:
                hs = cfg["headline_strong"]
                hs.update({"scaling": "strong", "total_rows": strong_total // world * world, "rows_per_s_all_gpus": r4(strong_total // world * world / (hs["ms"] * 1e-3)),
                           "result_check": strong_state.get("check")})
                if strong_state.get("check") and not strong_state["check"]["ok"]:
                    out.setdefault("result_check", {"ok": False, "what": "strong-scaled headline"})["ok"] = False
            # C5: the C4 join strong-scaled — build replicated, the fact rows (10^8; NQE_BENCH_C5_ROWS for functional runs) range-split
            # over the ranks.  The consumer-local form (gather = 0: every rank keeps its own output rows, rank order == row order) is
            # C5's headline — SURVEY 8e: the ordered all-gather of 3.2 GB is bound by one xGMI link per peer pair and dominates the probe
            # by an order of magnitude — with the gathered form (BASELINE configs[4] as written) beside it.
            c5_rows = int(os.environ.get("NQE_BENCH_C5_ROWS", 10**8))
            shard = c5_rows // world
            add("c5_probe_only", lambda: wl_c4(B, shard, 10**6, "dense", csteps, cwarm, gather=False, blocks=cblocks))
            add("c5_probe_and_gather", lambda: wl_c4(B, shard, 10**6, "dense", csteps, cwarm, gather=True, blocks=cblocks))
            p, g = cfg["c5_probe_only"], cfg["c5_probe_and_gather"]
            gather_ms = g["ms"] - p["ms"]
            out_bytes = 24.0 * shard * world  # three distinct 8-byte output columns per row (the two key columns are one buffer and travel once)
            inbound = out_bytes * (world - 1) / world
            cfg["c5"] = {"fact_rows_per_gpu": shard, "ms": p["ms"], "probe_only_ms": p["ms"], "probe_rows_per_s_all_gpus": r4(shard * world / (p["ms"] * 1e-3)),
                         "frac": p["frac"], "gather": 0, "with_gather": {"gather_ms": r4(gather_ms), "end_to_end_ms": g["ms"], "gathered_bytes_per_rank_inbound": inbound,
                                                                         "xgmi_GBps_per_gpu_inbound": r4(inbound / (gather_ms * 1e-3) / 1e9) if gather_ms > 0 else None,
                                                                         "gather_check": g.get("gather_check")},
                         "gather_ms": r4(gather_ms), "end_to_end_ms": g["ms"],
                         "xgmi_GBps_per_gpu_inbound": r4(inbound / (gather_ms * 1e-3) / 1e9) if gather_ms > 0 else None}
        if world == 1 and not B.distributed and (not only or any(x.startswith(("dropin", "upload")) for x in only)):
            # ---- the drop-in path (rewrite(tree).execute() through the mirrors of the reference's operator surface) and the ingest
            B.torch.cuda.empty_cache()
            up_rec, up_table = upload_record(B, main_state["table"], len(main_state["sh"]["cols"]), main_state["n"])
            out["upload"] = up_rec
            details["upload"] = dict(up_rec)
            dtab = up_table if up_table is not None else main_state["table"]
            def add_dropin(cname, rec, **extra):
                details["configs"][cname] = dict(rec, **extra)           # the full record (what was compared, which table) goes to the details file
                rec["parity"] = {"ok": rec["parity"]["ok"], "rows": rec["parity"]["rows"]}
                cfg[cname] = rec

            add_dropin("dropin_headline", dropin_aggregate(B, dtab, main_state["sh"], main_state["total"], csteps, cwarm, cblocks),
                       table="registered from host columns (library-owned)" if up_table is not None else "borrowed device columns")
            del up_table, dtab
            add_dropin("dropin_c2", dropin_c2(B, 10**8, csteps, cwarm, cblocks))
            add_dropin("dropin_c4", dropin_c4(B, 10**8, 10**6, csteps, cwarm, cblocks))
            up_rec.pop("what", None)
        out["configs"] = cfg

    if rank == 0 and world == 1 and wl == "headline" and not (args.no_configs or args.random_keys or args.no_live_traffic or args.rows) and args.pass_frac == 0.5:
        # everything timed is done: the counters are collected by child processes (rocprofv3 cannot attach to this one) — the headline and
        # BASELINE's other single-GPU configs
        lt = LiveTraffic()
        if lt.measure(out["roofline"], "headline", ["--workload", "headline"]):
            # the line keeps the two numbers and the live pass's account; the quoted file's description goes to the details
            details["main"]["traffic_quoted_from"] = out["roofline"].pop("traffic_quoted_from", None)
            out["roofline"]["traffic_quoted_from"] = f"profiles/{PROFILE_TAG}/pmc_traffic_headline.json"
        for cname in ("c2", "c3", "c4"):
            if cname in out.get("configs", {}) and cname in details["configs"]:
                roof = details["configs"][cname]["roofline"]
                if lt.measure(roof, cname, ["--workload", cname]):
                    out["configs"][cname]["traffic_ratio"] = r4(roof["traffic_ratio"])
                    out["configs"][cname]["traffic_live"] = 1   # (`traffic_ratio` of this config is this run's measurement; the details file has the bytes)
        lt.close()
    line = finish_line(out)  # `summary` = the LAST key: what a record that keeps only the tail of the line still holds
    dpath = args.details or (os.path.join(ROOT, "gpurun_out", "bench_details.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "")
    if dpath and rank == 0:
        try:
            details["line"] = out
            with open(dpath, "w") as f:
                json.dump(details, f, indent=1)
        except OSError as e:
            sys.stderr.write(f"bench.py: could not write {dpath}: {e}\n")

    # RCCL writes a version banner through C stdio, which (redirected) is flushed at process exit — after Python's own output.
    # Everything buffered so far goes out on every rank first, so that rank 0's JSON line is the LAST line of the job's stdout.
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if B.distributed:
        B.dist.barrier()
    if rank == 0:
        print(line, flush=True)
    if par and not par[0]["ok"]:
        sys.stderr.write("bench.py: PARITY FAILURE against the oracle on the sample\n")
        sys.exit(3)
    if "result_check" in out and not out["result_check"]["ok"]:
        sys.stderr.write("bench.py: the sharded result failed its analytic check\n")
        sys.exit(3)
    bad = [k for k, c in out.get("configs", {}).items() if not c.get("parity", {"ok": True})["ok"]]
    if bad:
        sys.stderr.write(f"bench.py: PARITY FAILURE against the oracle in side config(s) {bad}\n")
        sys.exit(3)
    if B.distributed:
        B.dist.barrier()
        if B.comm is not None:
            B.comm.close()
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
