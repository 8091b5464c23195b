#!/usr/bin/env python
"""profiles/r04/README.md = the narrative below + the table tools/round_table.py makes from the committed files (run after
tools/summarize_profiles.py r04)."""
import subprocess
import sys

table = subprocess.check_output([sys.executable, "tools/round_table.py", "r04", "r03"], text=True)
text = f"""# Round 4 — results and evidence index

Produced on one MI355X by `tools/profile_round.sh r04` (through `gpurun`), summarised by `tools/summarize_profiles.py r04`; this
file by `tools/write_r04_readme.py`. Every `pmc_traffic_*.json` and `summary.json` records the `csrc_rev` it was made from
(`tools/csrc_rev.py`), and `bench.py` quotes a file as `roofline.traffic` only while that hash matches the running tree.

| File | What |
|---|---|
| `bench_default.json`, `bench_details.json` | the default `python bench.py` line (24 configs + 3 drop-in rows, `upload`, `cold_ms`, both CPU numbers), per-kernel details |
| `bench_no_plan_hints.json`, `bench_details_no_plan_hints.json` | the same line under `NQE_NO_PLAN_HINTS=1`: nothing remembered between executions (the key sample still runs) |
| `probe_cold.txt` | first execution / first execution over a second table / steady state of 12 query shapes, each in a FRESH process: default (code objects loaded and 24 GB reserved at context creation), without the reservation, with lazy code-object loading (`tools/probe_cold.py`) |
| `rocprofv3_kernel_stats_<config>.csv` | `rocprofv3 --kernel-trace --stats` of `bench.py --workload … --steps 20 --warmup 3` |
| `pmc_traffic_<config>.json`, `pmc_calibration.json`, `summary.json` | HBM bytes per step from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, calibrated in the same run on `tools/stream_bench` |
| `probe_*.txt`, `micro_bench.txt` | the diagnostic sweeps of earlier rounds, re-run on this revision; new: `probe_strings.txt` (Utf8 filter / group key / join key, 10⁷ rows) and `probe_csv.txt` (CSV parse on the GPU) — SURVEY §8f ranks 3 and 4 |

## The bench line

`ms` = whole step incl. host waits, median of three blocks; `frac` = SURVEY §8d bytes over the step's data kernels (HIP events) as a
fraction of 8 TB/s; first execution = `cold_ms` (the first run of the query shape in the bench process: hints cold, code objects
loaded at context creation, outputs from the reserved block; for the joins it is build + probe while `ms` is the probe); every row
is parity-checked against the oracle in the same run.

{table}
## What changed in round 4, and what each change bought

* **The reference-side binding is complete and checked** (`integration/rust/gpu.rs`, `tools/check_rust_shim.py`): resident
  `GpuMemTable` / `GpuBatch` path, every operator the planner builds rewritten, `from_reference` / `create` defined; the call
  pattern (fresh tree per query → rewrite → `execute()`) measured through the Python mirror as `dropin_*`: 1.001–1.016× the raw
  C-ABI calls. `upload`: 56.5 GB/s from pageable host memory, 57.3 GB/s page-locked.
* **First execution** (`probe_cold.txt`): code objects loaded at context creation (+22 ms per context; lazy loading cost the
  first query 2–10 ms per operator family), a 65536-key sample picks the aggregate's starting tier, `nqe_ctx_reserve` removes the
  first `hipMalloc`s. First ÷ steady in a fresh process: 1.01–1.27 for all 12 shapes (lazy modules, no reservation, abandoned
  tiers — the round-3 state — 1.2–12.9). Run-time specialised kernels are kept on disk with their source, so a new process takes a
  known tree's kernel on its first execution (`c2_expression_trees`' 1.67 ms first execution is a first-ever compilation).
* **The reference's own aggregate query in one pass** (`agg_readme_shape`, `src/main.rs:36-40`): the three-column instance with
  min / max on its last column: 7.21 ms (two passes) → 4.4–4.5 ms = 0.68–0.69 on 24 GB, PMC 1.005×. **1 % NULLs**
  (`headline_nullable`): 2.69–2.71 ms = 0.76, PMC 1.015×.
* **Outputs never alias memory the caller lent**: `c4` is now the written form (0.86–0.89 ms = 0.71–0.74 of peak on §8d's
  4.8 GB); `c4_shared_probe_columns` (library-owned or `NQE_TABLE_IMMUTABLE` probe table) keeps the 0.58 ms form. The wide-payload
  / 10⁷ / 10⁸-row-dim / sparse-key rows are the written forms too (round 3 quoted them with shared columns).
* **Selection + projection in one specialised pass** (`c2_expression_trees`): 0.70 → 0.57–0.59 ms, frac 0.45 → 0.54–0.56, PMC
  1.37× → 1.03×. For plain predicates the static two-kernel form stays (the one-pass kernel measured 0.42 vs 0.36 ms on C2).
* **Predicate trees in the aggregate through a lean specialised kernel** (`agg_tree_predicate`): 3.71 → 2.6 ms, 0.55 → 0.78–0.79; trees the static kernels can only
  materialise (column-with-column compares) take the same kernel once it is compiled. Duplicate-key join: a plain-words write pass, 2.40 → 2.03 ms.
* **Utf8 and CSV measured** (`probe_strings.txt`, `probe_csv.txt`; SURVEY §8f ranks 3-4): the string take with 4-64 lanes per
  string and 8-byte moves, 8 bytes per step in the dictionary hash / compare: Utf8-key join of 10⁷ rows 6.96 → 1.88 ms, Utf8 filter
  0.91 → 0.42 ms, 10⁵-string group-by 1.81 → 1.16 ms; CSV parse of a 56 MB image 1.28 ms (44 GB/s HBM-resident, 24 GB/s from host bytes).
* **Two ranks on one GPU** through the host-staged transport run `bench.py`'s multi-rank blocks (`world = 2`); C5's headline is
  the consumer-local join with the gathered form beside it. No scaling claim: the pool has one GPU per box.
* **The partitioned aggregate** (`agg_65536_groups` 1.26 → 1.14–1.15 ms per step, `agg_1048576_groups` 1.61 → 1.32–1.36;
  A/B of every switch on one box: `probe_switches.txt`): (i) its tail ranks the groups by key − min instead of sorting them when
  their keys lie in a compact range — one host wait and three launches instead of three and twelve; (ii) key-range partitions
  with LDS tables addressed by slot (`agg_slab_segments_direct_kernel`; 256 partitions where hashing needs 512); (iii) the
  prefetch of its second kernel never worked — odd-aligned 64-bit pairs out of a `dwordx3` load, a phi behind a conditional fetch and
  a FLAT load of an LDS flag each made the wave wait for the loads it had just issued (found in the ISA; DESIGN §3.3); (iv) where a
  tile's time goes, measured with shader-clock stamps (`tools/probe_slab_phases.py`): the scatter moves its read + write mix at
  3.9–4.6 TB/s whatever the partition count, the second kernel is bound by LDS read-modify-writes.
* **Mid-size tables with many groups** (2¹⁸ … a few million rows) never reached the partitioned path: the global table was grown
  and every workgroup folded its LDS table into it through device-scope atomics. The key sample now runs from 2¹⁸ rows and an
  overfull global table goes to partitions: 500 000 rows / 90 000 groups 1.5 → 0.13 ms (first execution 2.0 → 0.49).
* **What a context remembers is bound to the table handle** (`nqe_table::uid`): `bench.py`'s consecutive many-group configs got the
  same torch buffers, so the 2²⁰-group config started from the 65536-group config's plan — a 20 ms "first execution" of four
  abandoned attempts in one bench line.
* Tried and left out, with numbers: two scatter workgroups per CU in the partitioned aggregate (kernels 1.06 → 1.12 ms);
  drawing the look-back's next ticket early (0.60 → 0.82 ms); a status word per 512-row chunk (0.94 ms); the one-pass selection
  for plain predicates; an order-restoring radix join (DESIGN §3.7: ≈ 8 GB of streamed traffic to save ≤ 0.3 ms — the beyond-L2
  probes sit on the 120-B-per-lookup line-fetch floor).

## Open

More than one physical GPU (C5, the xGMI numbers); the partitioned aggregate (three passes: PMC traffic is within 4 % of
that floor; the scatter's read + write mix runs at 3.9–4.6 TB/s, the second kernel is bound by LDS read-modify-writes); sparse 4 K–8 K-group band (two key subsets = every row
issued twice); predicate trees over keys other than `col % m` (interpreted or materialised); joins beyond L2 (line-fetch floor).
"""
open("profiles/r04/README.md", "w").write(text)
print(len(text))
