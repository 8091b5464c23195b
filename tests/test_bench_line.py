"""bench.py's JSON line (CPU): it must stay under 8 KB with every side config in it and end with the compact `summary`, so that a
record which keeps only the tail of stdout still shows C2/C3/C4."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def fake_result(i, join=False):
    roof = {"bound": "hbm", "achieved": 6543.21987 + i, "peak": 8000.0, "unit": "GB/s", "frac": 0.8179024 + i * 1e-3, "traffic": None,
            "kernel": "agg_grouped_fast", "kernel_ms_per_step": 2.41298765 + i, "algorithmic_bytes_per_step": 16e9,
            "kernels": {f"k{j}": {"ms_per_step": 0.123456789 * j, "launches_per_step": 1.0} for j in range(8)},
            "frac_physical": 0.7123456, "physical_bytes_per_step": 4.0e9}
    if join:
        roof.update({"build_ms": 0.15123, "frac_end_to_end": 0.40123, "execute_call_ms": 1.2345, "two_pass_ms": 1.1234, "execute_over_two_pass": 1.0989,
                     "frac_8d": 0.81234, "traffic_ratio": 1.00234})
    roof["frac_step"] = roof["frac"] * roof["kernel_ms_per_step"] / (2.4612345 + i)
    return {"ms_per_step": 2.4612345 + i, "spread": {"ms_min": 2.4012345, "ms_max": 2.5912345, "blocks": 3, "steps_per_block": 10}, "cold_ms": 3.912345,
            "workload": "select ... " * 20, "rows_per_gpu": 10**9, "roofline": roof,
            "parity_checked": {"rows": 20_000_000, "ok": True, "groups": 1024, "tolerance": "counts exact, f64 rtol 1e-9"},
            "cpu_baseline": {"value": 1.3612345e7, "unit": "rows/s", "cores": 1, "kind": "port", "sample": "x" * 100, "seconds": 1.5}}


NAMES = ["c3", "headline_random_keys", "c3_random_keys", "headline_int64_values", "headline_single_column", "agg_three_value_columns", "agg_readme_shape", "headline_nullable", "agg_tree_predicate",
         "c2", "c2_random_ids", "c4", "c4_shared_probe_columns", "c4_wide_payload", "c4_dim_1e7", "c4_dim_1e8", "c4_sparse_keys", "c4_dup_keys", "c4_partial_match", "agg_4096_groups",
         "agg_5000_groups", "agg_6000_groups", "agg_11000_groups", "agg_65536_groups", "agg_1048576_groups", "agg_12000_groups_count_sum_avg"]


def make_out():
    main = fake_result(0)
    main["roofline"].pop("kernels")
    main["roofline"].update({"traffic": 16.03e9, "traffic_ratio": 1.002, "traffic_quoted_from": "profiles/r03/pmc_traffic_headline.json (" + "y" * 120 + ")"})
    out = {"metric": "filter_hash_aggregate_rows_per_s", "value": 4.06e11, "unit": "rows/s", "n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": 2.461,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 260, "rows_per_gpu": 10**9, "total_rows": 10**9, "parallelism": "row-range x1"}, "roofline": main["roofline"],
           "parity_checked": main["parity_checked"], "cpu_baseline": main["cpu_baseline"]}
    out["configs"] = {n: bench.compact(fake_result(i, join=n.startswith("c4") or n == "c2")) for i, n in enumerate(NAMES)}
    return out


def test_line_fits_and_ends_with_summary():
    out = make_out()
    line = bench.finish_line(out)
    assert len(line) <= bench.LINE_LIMIT, len(line)
    d = json.loads(line)
    assert list(d)[-1] == "summary"
    assert set(d["summary"]) == {"headline", *NAMES}
    for n in NAMES:
        ms, frac, fphys, ok = d["summary"][n]
        assert ms > 0 and 0 < frac < 1.2 and ok is True
        assert d["configs"][n]["parity"] == {"ok": True, "rows": 20_000_000}
        assert 0 < d["configs"][n]["frac_step"] <= d["configs"][n]["frac"]      # the step's wall time holds its kernels
    # the trailing 2 KB alone hold the whole summary
    assert line.rfind('"summary"') > len(line) - 2048
    # the contract's keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert "traffic_quoted_from" in d["roofline"]


def test_every_side_config_has_a_parity_check():
    """no `add(...)` call of the single-GPU config block without a parity lambda"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    block = src[src.index('add("c3"'):src.index("# the headline without its exchange")]
    adds = [l for l in block.splitlines() if l.strip().startswith("add(")]
    assert len(adds) >= 16
    for l in adds:
        assert l.rstrip().endswith(("pa(20_000_000))", "pa(10_000_000))", "pj)", "pj_full)", 'parity_c2(B, s, s["n"]))', 'parity_c2_tree(B, s, s["n"]))', "pj)  # attr spans 2^62: an 8 MB payload table",
                                  "parity_c4_property(B, s))")), l


def test_parse_defaults_finish_quickly():
    a = bench.parse([])
    assert a.gpus == 1 and a.steps <= 20 and a.warmup <= 5 and a.workload == "headline"


def test_live_traffic_falls_back_to_the_quoted_figure(monkeypatch):
    """no rocprofv3 (or a profiler already attached): the quoted figure stays, the line says why"""
    monkeypatch.setenv("PATH", "/nonexistent")
    lt = bench.LiveTraffic()
    roof = {"traffic": 16.03e9, "algorithmic_bytes_per_step": 16e9}
    assert lt.measure(roof, "headline", ["--workload", "headline"]) is False
    assert roof["traffic"] == 16.03e9 and roof["traffic_live"].startswith("not measured: no rocprofv3")
    lt.close()
    monkeypatch.undo()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    lt = bench.LiveTraffic()
    assert lt.why_not == "this process runs under a profiler"
    lt.close()


def test_live_traffic_arithmetic(monkeypatch, tmp_path):
    """counters are KB per dispatch; bytes per step = sum over the step's kernels of mean x 1024 x factor x launches per step"""
    fake_bin = tmp_path / "rocprofv3"
    fake_bin.write_text("#!/bin/sh\nexit 0\n")
    fake_bin.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path))
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    calls = []

    def fake_pass(counter, cmd, workdir, timeout_s):
        calls.append((counter, cmd[-1] if cmd[-1] == "calib" else cmd[3]))
        if cmd[-1] == "calib":   # stream_bench: 16e9 B read is reported as 8e9 B (factor 2); 6.4e9 B written as 6.4e9 B
            return {"read2<4,0>": (8e9 / 1024, 3)} if counter == "FETCH_SIZE" else {"void copy_rw<4, 0, 4, 1>(...)": (6.4e9 / 1024, 3)}
        if counter == "FETCH_SIZE":
            return {"keep_from_range_tile_kernel<..>": (0.2e9 / 1024, 4), "scan_redundant_kernel<unsigned int, unsigned long>": (1e3 / 1024, 4), "compact_staged_kernel<0>": (0.4e9 / 1024, 4)}
        return {"keep_from_range_tile_kernel<..>": (0.0125e9 / 1024, 4), "scan_redundant_kernel<unsigned int, unsigned long>": (1e3 / 1024, 4), "compact_staged_kernel<0>": (0.4e9 / 1024, 4)}

    monkeypatch.setattr(bench, "_pmc_pass", fake_pass)
    lt = bench.LiveTraffic()
    roof = {"traffic": 1.7e9, "algorithmic_bytes_per_step": 2.0e9}
    ok = lt.measure(roof, "c2", ["--workload", "c2"])
    lt.close()
    assert ok, roof
    if os.path.exists(os.path.join(ROOT, "tools", "stream_bench")):
        assert abs(lt.factors[0] - 2.0) < 1e-9 and abs(lt.factors[1] - 1.0) < 1e-9
        exp = (0.2e9 + 1e3 + 0.4e9) * 2.0 + (0.0125e9 + 1e3 + 0.4e9) * 1.0
        assert abs(roof["traffic"] - exp) < 1.0, (roof["traffic"], exp)
    assert roof["traffic_quoted"] == 1.7e9 and roof["traffic_live"].startswith("measured by this run")
    assert abs(roof["traffic_ratio"] - roof["traffic"] / 2.0e9) < 1e-12


def test_live_traffic_stops_at_its_time_budget_and_after_a_failed_pass(monkeypatch, tmp_path):
    fake_bin = tmp_path / "rocprofv3"
    fake_bin.write_text("#!/bin/sh\nexit 0\n")
    fake_bin.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path))
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    lt = bench.LiveTraffic(budget_s=0.0)
    roof = {"traffic": 1.0, "algorithmic_bytes_per_step": 1.0}
    assert lt.measure(roof, "c2", ["--workload", "c2"]) is False and "time budget" in roof["traffic_live"]
    lt.close()
    calls = []

    def failing_pass(counter, cmd, workdir, timeout_s):
        calls.append(counter)
        raise RuntimeError("rocprofv3 --pmc FETCH_SIZE exited with 1")

    monkeypatch.setattr(bench, "_pmc_pass", failing_pass)
    lt = bench.LiveTraffic()
    lt.factors, lt.basis = (2.0, 1.0), "test"
    r1, r2 = dict(roof), dict(roof)
    assert lt.measure(r1, "headline", ["--workload", "headline"]) is False
    assert lt.measure(r2, "c2", ["--workload", "c2"]) is False
    lt.close()
    assert len(calls) == 1, calls   # the second config did not try again
    assert r1["traffic"] == 1.0 and r2["traffic"] == 1.0 and "exited with 1" in r2["traffic_live"]
