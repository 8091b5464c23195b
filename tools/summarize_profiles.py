#!/usr/bin/env python
"""gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) → profiles/<tag>/ (tracked)."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
summary = {}
for f in sorted(os.listdir(src)):
    if f.startswith("bench_") and f.endswith(".json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
        try:
            d = json.load(open(os.path.join(src, f)))
            summary[f[:-5]] = {"rows_per_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms_per_step"],
                               "achieved_GBps": d["roofline"]["achieved"], "frac_of_8TBps": d["roofline"]["frac"],
                               "cpu_baseline_rows_per_s": d.get("cpu_baseline", {}).get("value")}
        except Exception as e:  # noqa: BLE001
            summary[f[:-5]] = {"error": str(e)}
for f in sorted(os.listdir(src)):
    if f.startswith("probe_") and f.endswith(".txt"):  # tools/probe_paths.py / probe_exchange.py diagnostics quoted in DESIGN.md
        lines = [l for l in open(os.path.join(src, f), errors="replace").read().splitlines() if "amdgpu.ids" not in l and not l.startswith("Hostname") and "Librccl" not in l]
        open(os.path.join(dst, f), "w").write("\n".join(lines) + "\n")
for w in ("headline", "c2", "c3", "c4"):
    p = os.path.join(src, f"prof_{w}", f"{w}_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"rocprofv3_kernel_stats_{w}.csv"))
pmc = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    p = os.path.join(src, name, "headline_counter_collection.csv")
    if os.path.exists(p):
        rows = list(csv.DictReader(open(p)))
        ks = [float(r["Counter_Value"]) for r in rows if "agg_grouped" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        if ks:
            pmc[ctr] = {"per_launch_values_KB": ks, "mean_KB": sum(ks) / len(ks)}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    f, w = pmc["FETCH_SIZE"]["mean_KB"] * 1024, pmc["WRITE_SIZE"]["mean_KB"] * 1024
    pmc["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; kernel agg_grouped_fast_kernel, 10^9 rows. "
                   "gfx950 FETCH_SIZE tallies 64 B per 128-B request (MI355X_MICROARCH.md §HBM): doubled. Calibration: the kernel reads id + v "
                   "exactly once = 16.0e9 B.")
    pmc["hbm_bytes_per_launch_corrected"] = 2 * f + w
    pmc["algorithmic_bytes_per_launch"] = 16e9
    json.dump(pmc, open(os.path.join(dst, "pmc_traffic_headline.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
