#!/usr/bin/env python
"""Markdown table of a round's bench line: profiles/<tag>/bench_default.json (+ bench_no_plan_hints.json, summary.json's PMC traffic,
the previous round's line for comparison).  usage: python tools/round_table.py r05 [r04]

Per config the table also puts the rocprofv3 --kernel-trace --stats average (profiles/<tag>/rocprofv3_kernel_stats_<config>.csv: the step's data
kernels, average duration x launches per step) beside the HIP-event figure of the bench line and FLAGS a disagreement above 3 % — the two
come from different processes, often different boxes."""
import json
import os
import csv
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_configs import CONFIGS  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
prev = sys.argv[2] if len(sys.argv) > 2 else None
d = json.load(open(os.path.join("profiles", tag, "bench_default.json")))
nh_path = os.path.join("profiles", tag, "bench_no_plan_hints.json")
nh = json.load(open(nh_path)) if os.path.exists(nh_path) else {}
summ = json.load(open(os.path.join("profiles", tag, "summary.json")))
det_path = os.path.join("profiles", tag, "bench_details.json")
det = json.load(open(det_path)).get("configs", {}) if os.path.exists(det_path) else {}  # what the line sheds to stay under its size limit (cold_ms, …) is here
pv = json.load(open(os.path.join("profiles", prev, "bench_default.json"))) if prev and os.path.exists(os.path.join("profiles", prev, "bench_default.json")) else {}


def f(x, n=3):
    return "" if x is None else f"{x:.{n}g}" if isinstance(x, float) else str(x)


def rocprof_ms(name):
    """the step's data kernels in the rocprofv3 stats of the config's own run: sum of average duration x launches per step (None: no file)"""
    path = os.path.join("profiles", tag, f"rocprofv3_kernel_stats_{name}.csv")
    if name not in CONFIGS or not os.path.exists(path):
        return None
    tot, hit = 0.0, False
    stats = list(csv.DictReader(open(path)))
    for sub, per_step in CONFIGS[name][0].items():
        match = [r for r in stats if sub in r["Name"]]
        if not match:
            continue
        hit = True
        calls = sum(int(r["Calls"]) for r in match)
        tot += sum(float(r["TotalDurationNs"]) for r in match) / calls * per_step / 1e6
    return tot if hit else None


# what changed in a row's DEFINITION since the previous round (so that the "previous round" column is not read as a like-for-like A/B)
NOTES = {
    "agg_5000_groups": "new in round 6 (one directly addressed workgroup table without key words; the two-subset form it replaces: 0.60 ms per step, ab_wide_direct.txt)",
    "agg_12000_groups_count_sum_avg": "new in round 6 (count / sum / avg only: 12-byte slots, one workgroup table up to 13632 keys; with min / max the same keys take the range tier: 0.84 ms per step, probe_no_minmax.txt)",
    "agg_6000_groups": "round 6: plain (not non-temporal) loads in the subset instances, equal halves of the range (ab_sub_plain_loads.txt)",
    "agg_11000_groups": "new in round 6 (two key subsets, each half of the range in a table without key words: up to 2 x 5840 keys; before: the range tier, 0.86 ms per step)",
    "c2": "kernel time now includes the tile-count scan (13 us as one workgroup; 9.5 us since the scan runs one workgroup per 4096 counts)",
    "c2_random_ids": "kernel time now includes the tile-count scan (9.5 us)",
    "c2_expression_trees": "",
    "agg_three_value_columns": "parity now over all 10^9 rows (r05: 2 x 10^7)",
    "agg_readme_shape": "parity now over all 10^9 rows (r05: 2 x 10^7)",
    "headline_nullable": "parity now over all 10^9 rows (r05: 2 x 10^7)",
    "agg_tree_predicate": "parity now over all 10^9 rows (r05: 2 x 10^7)",
    "c4_dim_1e8": "build_ms: the two-level partitioned build (r05: the place pass)",
}
rows = [("headline", {"ms": d["ms_per_step"], "kernel_ms": d["roofline"].get("kernel_ms_per_step"), "kernel_ms_min": d["roofline"].get("kernel_ms_min"),
                      "kernel_ms_max": d["roofline"].get("kernel_ms_max"), "frac": d["roofline"]["frac"], "frac_step": d["roofline"].get("frac_step"), "cold_ms": d.get("cold_ms"), "parity": d.get("parity_checked", {})})]
rows += list(d.get("configs", {}).items())
print("| config | ms / step | kernel ms (HIP events) [min .. max over blocks] | kernel ms (rocprofv3 avg) | differ | frac of 8 TB/s (8d bytes) | frac_step (same bytes / step wall time) | frac physical | PMC traffic / algorithmic | first execution ms | no plan hints ms | parity (rows) | previous round ms | definition changes |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, c in rows:
    if "raw_ms" in c:
        continue
    tr = (summ.get(name) or {}).get("traffic_ratio")
    nhc = nh.get("ms_per_step") if name == "headline" else (nh.get("configs", {}).get(name) or {}).get("ms")
    pvc = pv.get("ms_per_step") if name == "headline" else (pv.get("configs", {}).get(name) or {}).get("ms")
    par = c.get("parity") or {}
    ok = par.get("ok")
    droof = (det.get(name) or {}).get("roofline") or {}   # (the line sheds kernel_ms* first when it outgrows its limit: the details file keeps them)
    km, rp = c.get("kernel_ms") or droof.get("kernel_ms_per_step"), rocprof_ms(name)
    kmin, kmax = c.get("kernel_ms_min") or droof.get("kernel_ms_min"), c.get("kernel_ms_max") or droof.get("kernel_ms_max")
    spread = f" [{f(kmin, 4)} .. {f(kmax, 4)}]" if kmin is not None else ""
    diff = ""
    if km and rp:
        pct = (rp - km) / km * 100.0
        diff = f"{pct:+.1f} %" + (" **(> 3 %)**" if abs(pct) > 3.0 else "")
    print(f"| `{name}` | {f(c.get('ms'), 4)} | {f(km, 4)}{spread} | {f(rp, 4)} | {diff} | {f(c.get('frac'))} | {f(c.get('frac_step'))} | {f(c.get('frac_physical'))} | {f(tr, 4)} | {f(c.get('cold_ms') if c.get('cold_ms') is not None else (det.get(name) or {}).get('cold_ms'), 4)} | {f(nhc, 4)} | "
          f"{'ok' if ok else ok} ({par.get('rows', '')}) | {f(pvc, 4)} | {NOTES.get(name, '')} |")
print()
print("| drop-in row | ms / step | raw C-ABI ms | ratio | equal to the raw call |")
print("|---|---|---|---|---|")
for name, c in rows:
    if "raw_ms" in c:
        print(f"| `{name}` | {f(c['ms'], 4)} | {f(c['raw_ms'], 4)} | {f(c['over_raw'], 4)} | {(c.get('parity') or {}).get('ok')} |")
for k in ("upload", "reserved", "cpu_baseline", "cpu_optimised_multicore"):
    if k in d:
        print(f"\n`{k}`: `{json.dumps(d[k])}`")
