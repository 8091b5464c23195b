#!/usr/bin/env python
"""Markdown table of a round's bench line: profiles/<tag>/bench_default.json (+ bench_no_plan_hints.json, summary.json's PMC traffic,
the previous round's line for comparison).  usage: python tools/round_table.py r04 [r03]"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = sys.argv[2] if len(sys.argv) > 2 else None
d = json.load(open(os.path.join("profiles", tag, "bench_default.json")))
nh_path = os.path.join("profiles", tag, "bench_no_plan_hints.json")
nh = json.load(open(nh_path)) if os.path.exists(nh_path) else {}
summ = json.load(open(os.path.join("profiles", tag, "summary.json")))
pv = json.load(open(os.path.join("profiles", prev, "bench_default.json"))) if prev and os.path.exists(os.path.join("profiles", prev, "bench_default.json")) else {}


def f(x, n=3):
    return "" if x is None else f"{x:.{n}g}" if isinstance(x, float) else str(x)


rows = [("headline", {"ms": d["ms_per_step"], "frac": d["roofline"]["frac"], "cold_ms": d.get("cold_ms"), "parity": d.get("parity_checked", {})})]
rows += list(d.get("configs", {}).items())
print("| config | ms / step | frac of 8 TB/s (8d bytes) | frac physical | PMC traffic / algorithmic | first execution ms | no plan hints ms | parity | previous round ms |")
print("|---|---|---|---|---|---|---|---|---|")
for name, c in rows:
    if "raw_ms" in c:
        continue
    tr = (summ.get(name) or {}).get("traffic_ratio")
    nhc = nh.get("ms_per_step") if name == "headline" else (nh.get("configs", {}).get(name) or {}).get("ms")
    pvc = pv.get("ms_per_step") if name == "headline" else (pv.get("configs", {}).get(name) or {}).get("ms")
    ok = (c.get("parity") or {}).get("ok")
    print(f"| `{name}` | {f(c.get('ms'), 4)} | {f(c.get('frac'))} | {f(c.get('frac_physical'))} | {f(tr, 4)} | {f(c.get('cold_ms'), 4)} | {f(nhc, 4)} | {'ok' if ok else ok} | {f(pvc, 4)} |")
print()
print("| drop-in row | ms / step | raw C-ABI ms | ratio | equal to the raw call |")
print("|---|---|---|---|---|")
for name, c in rows:
    if "raw_ms" in c:
        print(f"| `{name}` | {f(c['ms'], 4)} | {f(c['raw_ms'], 4)} | {f(c['over_raw'], 4)} | {(c.get('parity') or {}).get('ok')} |")
for k in ("upload", "reserved", "cpu_baseline", "cpu_optimised_multicore"):
    if k in d:
        print(f"\n`{k}`: `{json.dumps(d[k])}`")
