#!/usr/bin/env python
"""gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) → profiles/<tag>/ (tracked).

Per config: the rocprofv3 --kernel-trace --stats summary, and pmc_traffic_<config>.json = HBM bytes per step from the separate
--pmc FETCH_SIZE / --pmc WRITE_SIZE passes, corrected by factors CALIBRATED IN THE SAME RUN on tools/stream_bench.hip's kernels
(known byte counts, the product kernels' own access pattern: 8-byte non-temporal loads and stores) — not on the kernel under test.
Every file records the csrc revision it was made from (tools/csrc_rev.py); bench.py refuses to quote a stale one."""
import csv
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
rev = open(os.path.join(src, "csrc_rev.txt")).read().strip() if os.path.exists(os.path.join(src, "csrc_rev.txt")) else None

from profile_configs import CONFIGS  # noqa: E402  (tools/profile_configs.py)


def counter_means(path, counter):
    """{kernel name: (mean counter value per dispatch, dispatches)} from a rocprofv3 counter_collection csv"""
    acc = {}
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        a = acc.setdefault(r["Kernel_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def find(means, sub):
    hits = [(k, v) for k, v in means.items() if sub in k]
    if not hits:
        return None
    tot = sum(v[0] * v[1] for _, v in hits)
    n = sum(v[1] for _, v in hits)
    return tot / n, n


# ---- calibration (FETCH_SIZE / WRITE_SIZE are reported in KB)
calib = {}
fm = counter_means(os.path.join(src, "pmc_fetch_calib", "calib_counter_collection.csv"), "FETCH_SIZE")
wm = counter_means(os.path.join(src, "pmc_write_calib", "calib_counter_collection.csv"), "WRITE_SIZE")
f_read2, w_fill = find(fm, "read2"), None
for k, v in wm.items():
    if "copy_rw" in k and ("Li0ELi4E" in k or "copy_rw<4,0,4,1>" in k.replace(" ", "")):
        w_fill = v
f_copy = None
for k, v in fm.items():
    if "copy_rw" in k and ("Li1ELi1E" in k or "copy_rw<4,1,1,1>" in k.replace(" ", "")):
        f_copy = v
if f_read2:
    calib["fetch_factor"] = 16e9 / (f_read2[0] * 1024)
    calib["fetch_basis"] = f"stream_bench read2<4,0>: 16e9 B streamed with 8-byte non-temporal loads, FETCH_SIZE reported {f_read2[0] * 1024:.4g} B"
if w_fill:
    calib["write_factor"] = 6.4e9 / (w_fill[0] * 1024)
    calib["write_basis"] = f"stream_bench copy_rw<4,0,4,1>: 6.4e9 B written with 8-byte non-temporal stores, WRITE_SIZE reported {w_fill[0] * 1024:.4g} B"
if f_copy and f_read2:
    calib["fetch_check_copy"] = f"copy_rw<4,1,1,1> reads 1.6e9 B: FETCH_SIZE x factor = {f_copy[0] * 1024 * calib['fetch_factor']:.4g} B"
gm = counter_means(os.path.join(src, "pmc_fetch_gather", "gather_counter_collection.csv"), "FETCH_SIZE")
g = find(gm, "gather")
if g and "fetch_factor" in calib:
    calib["random_8B_reads"] = (f"micro_bench gather (1e8 random 8-byte reads + the 0.8e9 B key stream, 1 GB and 32 MB tables, {g[1]} launches): mean FETCH_SIZE x factor = "
                                f"{g[0] * 1024 * calib['fetch_factor']:.4g} B per launch = {(g[0] * 1024 * calib['fetch_factor'] - 0.8e9) / 1e8:.1f} B per random read")
json.dump(calib, open(os.path.join(dst, "pmc_calibration.json"), "w"), indent=1)

# ---- per config
summary = {"csrc_rev": rev, "calibration": calib}
for name, (kernels, algo) in CONFIGS.items():
    st = os.path.join(src, f"prof_{name}", f"{name}_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, f"rocprofv3_kernel_stats_{name}.csv"))
    fm = counter_means(os.path.join(src, f"pmc_fetch_{name}", f"{name}_counter_collection.csv"), "FETCH_SIZE")
    wm = counter_means(os.path.join(src, f"pmc_write_{name}", f"{name}_counter_collection.csv"), "WRITE_SIZE")
    if not fm or not wm or "fetch_factor" not in calib or "write_factor" not in calib:
        continue
    per_kernel, fetch_b, write_b = {}, 0.0, 0.0
    for sub, per_step in kernels.items():
        f, w = find(fm, sub), find(wm, sub)
        if not f or not w:
            continue
        fb, wb = f[0] * 1024 * calib["fetch_factor"] * per_step, w[0] * 1024 * calib["write_factor"] * per_step
        per_kernel[sub] = {"fetch_bytes": fb, "write_bytes": wb, "dispatches_profiled": f[1]}
        fetch_b += fb
        write_b += wb
    rec = {"config": name, "csrc_rev": rev, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only; factors from "
           "pmc_calibration.json (stream_bench kernels, same run)", "fetch_factor": calib["fetch_factor"], "write_factor": calib["write_factor"],
           "kernels": per_kernel, "hbm_bytes_per_step_corrected": fetch_b + write_b, "fetch_bytes_per_step": fetch_b, "write_bytes_per_step": write_b,
           "algorithmic_bytes_per_step": algo, "traffic_ratio": (fetch_b + write_b) / algo}
    json.dump(rec, open(os.path.join(dst, f"pmc_traffic_{name}.json"), "w"), indent=1)
    summary[name] = {"traffic_GB": (fetch_b + write_b) / 1e9, "algorithmic_GB": algo / 1e9, "traffic_ratio": rec["traffic_ratio"]}

for f in sorted(os.listdir(src)):
    if (f.startswith("bench_") or f.startswith("cold_")) and f.endswith(".json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    if (f.startswith("probe_") or f == "micro_bench.txt") and f.endswith(".txt"):
        lines = [l for l in open(os.path.join(src, f), errors="replace").read().splitlines() if "amdgpu.ids" not in l and not l.startswith("Hostname") and "Librccl" not in l]
        open(os.path.join(dst, f), "w").write("\n".join(lines) + "\n")
try:
    d = json.load(open(os.path.join(src, "bench_default.json")))
    summary["bench_default"] = {"headline": {"ms_per_step": d["ms_per_step"], "frac": d["roofline"]["frac"], "parity": d.get("parity_checked")}}
    for k, v in d.get("configs", {}).items():
        summary["bench_default"][k] = {"ms_per_step": v.get("ms"), "frac": v.get("frac"), "frac_physical": v.get("frac_physical"), "parity": v.get("parity")}
except Exception as e:  # noqa: BLE001
    summary["bench_default"] = {"error": str(e)}
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
