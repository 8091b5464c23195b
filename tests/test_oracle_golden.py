"""Pins the CPU oracle against every known-answer vector the reference holds for the hot path
(tests/golden/expected.json cites each source).  CPU only."""
import struct

import numpy as np
import pytest

from naive_query_engine_amd import AggregateFunc, Column, DType, ErrorCode, Operator, Status
from naive_query_engine_amd.expression import binop, col, lit_f64, lit_i64
from oracle import oracle as orc


def cols(batch, names):
    idx = {f.name: i for i, f in enumerate(batch.fields)}
    return [batch.columns[idx[n]] for n in names]


def flat(e, batch):
    return e.flatten(batch.fields)


def test_scan_matches_csv(csv_tables, golden):
    t = csv_tables["test_data"]
    exp = golden["test_physical_scan"]
    out = orc.scan([t.columns])
    assert len(out) == 1 and len(out[0]) == 4
    for i, name in enumerate(["id", "name", "age", "score"]):
        assert out[0][i].to_list() == exp[name]
    assert [f.dtype for f in t.fields] == [DType.INT64, DType.UTF8, DType.INT64, DType.FLOAT64]


def test_scan_projection_mem_table():
    # memory.rs:59-90: projection [2, 1] picks columns c, b
    b = [Column.from_list([1, 2, 3], DType.INT64), Column.from_list([4, 5, 6], DType.INT64),
         Column.from_list([7, 8, 9], DType.INT64), Column.from_list([None, None, 9], DType.INT64)]
    out = orc.scan([b], [2, 1])
    assert [c.to_list() for c in out[0]] == [[7, 8, 9], [4, 5, 6]]


def test_selection_golden(csv_tables, golden):
    t = csv_tables["test_data"]
    # Projection[id, name, age](Scan)
    proj = orc.projection([t.columns], [flat(col(0), t), flat(col("name"), t), flat(col(2), t)])
    pred = binop(binop(col("id"), Operator.Plus, lit_i64(1)), Operator.Gt, lit_i64(5))
    out = orc.selection(proj, pred.flatten(t.fields[:3]))
    assert len(out) == 1
    assert out[0][0].to_list() == golden["test_selection"]["id"]
    assert out[0][1].to_list() == golden["test_selection"]["name"]


def test_projection_golden(csv_tables, golden):
    t = csv_tables["test_data"]
    add = binop(col("id"), Operator.Plus, lit_i64(1))
    out = orc.projection([t.columns], [flat(add, t), flat(col("name"), t)])
    assert out[0][0].to_list() == golden["test_projection"]["id_plus_1"]
    assert out[0][1].to_list() == golden["test_projection"]["name"]


def test_sql_where_id_gt_1(csv_tables, golden):
    t = csv_tables["test_data"]
    sel = orc.selection([t.columns], flat(binop(col(0), Operator.Gt, lit_i64(1)), t))
    out = orc.projection(sel, [flat(col(0), t), flat(col(1), t), flat(col(2), t)])
    exp = golden["sql_where_id_gt_1"]
    assert [c.to_list() for c in out[0]] == [exp["id"], exp["name"], exp["age"]]


def test_readme_filter_project_offset_limit(csv_tables, golden):
    t = csv_tables["test_data"]
    # Limit(Offset(Projection(Filter(Scan)))) (sql/planner.rs:58-81)
    sel = orc.selection([t.columns], flat(binop(col(0), Operator.Lt, lit_i64(9)), t))
    proj = orc.projection(sel, [flat(col(0), t), flat(col(1), t), flat(binop(col(2), Operator.Plus, lit_i64(100)), t)])
    out = orc.limit(orc.offset(proj, 2), 3)
    rows = list(map(list, zip(*[c.to_list() for c in out[0]])))
    assert rows == golden["readme_filter_project_offset_limit"]["rows"]


def test_readme_two_hash_joins(csv_tables, golden):
    emp, rank, dep = csv_tables["employee"], csv_tables["rank"], csv_tables["department"]
    # employee(id,name,department_id,rank) ⋈ rank(id,rank_name) on employee.rank = rank.id: LEFT = build
    j1 = orc.hash_join([emp.columns], [rank.columns], 3, 0)
    # (employee ⋈ rank) ⋈ department on employee.department_id = department.id
    j2 = orc.hash_join(j1, [dep.columns], 2, 0)
    # select id, name, rank_name, department_name: names resolve to the FIRST match (Q12)
    out = j2[0]
    rows = list(map(list, zip(out[0].to_list(), out[1].to_list(), out[5].to_list(), out[7].to_list())))
    assert rows == golden["readme_two_hash_joins"]["rows"]


def test_readme_group_by_id_mod_3(csv_tables, golden):
    t = csv_tables["test_data"]
    aggs = [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 2), (AggregateFunc.Sum, 3), (AggregateFunc.Avg, 3),
            (AggregateFunc.Max, 3), (AggregateFunc.Min, 3)]
    out = orc.aggregate([t.columns], aggs, group_nodes=flat(binop(col(0), Operator.Modulos, lit_i64(3)), t))
    assert len(out) == 1
    assert [c.dtype for c in out[0]] == [DType.UINT64] + [DType.FLOAT64] * 5
    rows = sorted(map(list, zip(*[c.to_list() for c in out[0]])))
    # exact f64 digits: 243.29000000000002 pins sequential row-order accumulation
    assert rows == sorted(golden["readme_group_by_id_mod_3"]["rows"])


def test_c1_plumbing(csv_tables, golden):
    t = csv_tables["test_data"]
    aggs = [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 2), (AggregateFunc.Avg, 3)]
    out = orc.aggregate([t.columns], aggs, group_nodes=flat(binop(col(0), Operator.Modulos, lit_i64(3)), t))
    rows = sorted(map(list, zip(*[c.to_list() for c in out[0]])))
    assert rows == sorted(golden["c1_plumbing"]["rows"])


def test_ungrouped_count_sum(csv_tables, golden):
    t = csv_tables["test_data"]
    out = orc.aggregate([t.columns], [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 0)])
    assert [c.to_list() for c in out[0]] == [[8], [42.0]]
    # quirk Q9: state is never cleared on the un-grouped path
    out2 = orc.aggregate([t.columns], [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 0)], executions=2)
    assert [c.to_list() for c in out2[0]] == [[16], [84.0]]


def test_xxhash64_matches_published_algorithm():
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(7)
    vals = [0, 1, -1, 2**63 - 1, -2**63] + [int(v) for v in rng.integers(-2**62, 2**62, 200)]
    for v in vals:
        ref = xxhash.xxh64(struct.pack("<q", v), seed=0).intdigest()
        assert orc.xxhash64_word(v) == ref
    for n in [0, 1, 3, 4, 7, 8, 15, 31, 32, 33, 64, 100]:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert orc.xxhash64(data) == xxhash.xxh64(data, seed=0).intdigest()


# ---- semantics the reference leaves un-asserted, restated from arrow-rs 13 (quirk ledger) ----
def test_null_predicate_emits_null_row_q4():
    ids = Column.from_list([1, None, 3, 4], DType.INT64)
    v = Column.from_list([1.5, 2.5, None, 4.5], DType.FLOAT64)
    fields_like = [type("F", (), {"name": "id"})(), type("F", (), {"name": "v"})()]
    pred = binop(col(0), Operator.Gt, lit_i64(2)).flatten(fields_like)
    out = orc.selection([[ids, v]], pred)
    assert out[0][0].to_list() == [None, 3, 4]
    assert out[0][1].to_list() == [None, None, 4.5]


def test_predicate_on_batch0_only_q3():
    b0 = [Column.from_list([1, 5, 9], DType.INT64)]
    b1 = [Column.from_list([7, 0, 8, 100], DType.INT64)]
    f = [type("F", (), {"name": "x"})()]
    out = orc.selection([b0, b1], binop(col(0), Operator.Gt, lit_i64(4)).flatten(f))
    assert out[0][0].to_list() == [5, 9]
    assert out[1][0].to_list() == [0, 8]  # batch-0 mask [F,T,T] zipped against batch 1, truncated


def test_type_mismatch_is_interval_error_q6():
    f = [type("F", (), {"name": "x"})()]
    with pytest.raises(ErrorCode) as e:
        orc.expr_evaluate([[Column.from_list([1, 2], DType.INT64)]], binop(col(0), Operator.Lt, lit_f64(4.5)).flatten(f))
    assert e.value.status == Status.IntervalError


def test_divide_by_zero_and_wrapping_q14():
    f = [type("F", (), {"name": "x"})()]
    x = Column.from_list([2**63 - 1, -7, None], DType.INT64)
    r = orc.expr_evaluate([[x]], binop(col(0), Operator.Plus, lit_i64(1)).flatten(f))
    assert r.to_list() == [-2**63, -6, None]
    r = orc.expr_evaluate([[x]], binop(col(0), Operator.Modulos, lit_i64(3)).flatten(f))
    assert r.to_list() == [(2**63 - 1) % 3, -1, None]  # truncated remainder keeps the dividend's sign
    with pytest.raises(ErrorCode) as e:
        orc.expr_evaluate([[x]], binop(col(0), Operator.Divide, lit_i64(0)).flatten(f))
    assert e.value.status == Status.ArrowError
    # a zero divisor under a NULL slot is not an error
    d = Column.from_list([1, 0], DType.INT64)
    n = Column.from_list([10, None], DType.INT64)
    ff = f + f
    r = orc.expr_evaluate([[n, d]], binop(col(0), Operator.Divide, col(1)).flatten(ff))
    assert r.to_list() == [10, None]


def test_kleene_logic():
    T, F, N = True, False, None
    a = Column.from_list([T, T, T, F, F, F, N, N, N], DType.BOOLEAN)
    b = Column.from_list([T, F, N, T, F, N, T, F, N], DType.BOOLEAN)
    f = [type("F", (), {"name": "a"})(), type("F", (), {"name": "b"})()]
    assert orc.expr_evaluate([[a, b]], binop(col(0), Operator.And, col(1)).flatten(f)).to_list() == [T, F, N, F, F, F, N, F, N]
    assert orc.expr_evaluate([[a, b]], binop(col(0), Operator.Or, col(1)).flatten(f)).to_list() == [T, T, T, T, F, N, T, N, N]


def test_aggregate_null_and_nan_semantics_q10():
    nan = float("nan")
    k = Column.from_list([0, 0, 0, 1, 1, None, 2], DType.INT64)
    v = Column.from_list([1.0, nan, 3.0, None, None, 5.0, -0.5], DType.FLOAT64)
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
    f = [type("F", (), {"name": "k"})(), type("F", (), {"name": "v"})()]
    out = orc.aggregate([[k, v]], aggs, group_nodes=col(0).flatten(f))
    rows = {r[0]: r for r in zip(*[c.to_list() for c in out[0]])}  # keyed by count(v): 3, 0, 1
    assert set(rows) == {3, 0, 1}  # NULL-key row dropped
    fmax = np.finfo(np.float64).max
    all_null = rows[0]  # key 1: every value NULL
    assert all_null[1] == 0.0 and np.isnan(all_null[2]) and all_null[3] == fmax and all_null[4] == -fmax
    with_nan = rows[3]  # key 0: 1.0, NaN, 3.0
    assert np.isnan(with_nan[1]) and np.isnan(with_nan[2]) and with_nan[3] == 1.0 and np.isnan(with_nan[4])
    assert rows[1][1:] == (-0.5, -0.5, -0.5, -0.5)


def test_join_duplicates_order_and_validity_ignored_q11():
    # duplicate build keys: matches come out in ascending build row index, probe-row-major
    lk = Column.from_list([7, 3, 7, None, 7], DType.INT64)  # the NULL slot holds raw 0
    lp = Column.from_list([10, 11, 12, 13, 14], DType.INT64)
    rk = Column.from_list([3, 7, 0, 9], DType.INT64)
    rp = Column.from_list([0.5, 1.5, 2.5, 3.5], DType.FLOAT64)
    out = orc.hash_join([[lk, lp]], [[rk, rp]], 0, 0)
    rows = list(zip(*[c.to_list() for c in out[0]]))
    assert rows == [(3, 11, 3, 0.5), (7, 10, 7, 1.5), (7, 12, 7, 1.5), (7, 14, 7, 1.5), (None, 13, 0, 2.5)]


def test_join_errors():
    a = [[Column.from_list([1.0], DType.FLOAT64)]]
    with pytest.raises(ErrorCode) as e:
        orc.hash_join(a, a, 0, 0)
    assert e.value.status == Status.NotImplemented
    with pytest.raises(ErrorCode) as e:
        orc.hash_join(a, a, -1, -1)
    assert e.value.status == Status.PlanError


def test_synth_generators_are_deterministic():
    a = orc.synth_fill(1, 2, 0, 1000, 60, 18).view(np.int64)
    b = orc.synth_fill(1, 2, 500, 500, 60, 18).view(np.int64)
    assert (a[500:] == b).all() and a.min() >= 18 and a.max() < 78
    f = orc.synth_fill(2, 3, 0, 1000).view(np.float64)
    assert (f >= 0).all() and (f < 100).all()
    assert (orc.synth_fill(0, 0, 5, 4).view(np.int64) == [5, 6, 7, 8]).all()


def test_optimised_multicore_headline_matches_the_port():
    """bench.py's optional second CPU number (orc_headline_parallel: per-thread direct-mapped tables) computes what the reference-faithful
    port computes for the headline query: counts exact, Float64 within 1e-9"""
    from naive_query_engine_amd import AggregateFunc as A
    from naive_query_engine_amd import Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from tests.helpers import fields

    n = 300_007
    ids = orc.synth_fill(0, 0, 0, n).view(np.int64)
    v = orc.synth_fill(2, 3, 0, n).view(np.float64)
    f = fields("id", "v")
    ref = orc.aggregate([[Column.from_numpy(ids), Column.from_numpy(v)]], [(A.Count, 1), (A.Sum, 1), (A.Min, 1), (A.Max, 1)],
                        group_nodes=binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f), pred_nodes=binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f))[0]
    e = np.stack([c.to_numpy().astype(np.float64) for c in ref], axis=1)
    e = e[np.lexsort(e.T[::-1])]
    for threads in (1, 3, 8):
        g = orc.headline_parallel(ids, v, n // 2, 1024, threads)
        g = g[np.lexsort(g.T[::-1])]
        assert g.shape == e.shape and (g[:, 0] == e[:, 0]).all() and np.allclose(g, e, rtol=1e-9, atol=0)


@pytest.mark.parametrize("form", ["filter_sorted", "no_filter_sorted", "filter_random", "no_filter_random", "int64_values", "single_column", "many_groups"])
def test_parallel_grouped_form_matches_the_port(form):
    """The full-size parity checks (tests/test_gpu_fullsize.py, bench.py's `parity_checked`) compare the GPU with orc_grouped_parallel —
    every BASELINE aggregate shape it is used for is first checked here against the reference-faithful single-threaded port
    (aggregate/mod.rs:113-222): keyed rows, counts exact, Float64 within 1e-9; chunked evaluation + merge_grouped included."""
    from naive_query_engine_amd import AggregateFunc as A
    from naive_query_engine_amd import Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from tests.helpers import fields

    n, mod = 200_003, 1024
    total = 10 * n
    random_ids = "random" in form
    ids = (orc.synth_fill(1, 1, 0, n, total, 0) if random_ids else orc.synth_fill(0, 0, 0, n)).view(np.int64)
    v = orc.synth_fill(2, 3, 0, n).view(np.float64)
    limit = None if form.startswith("no_filter") else (total // 2 if random_ids else n // 2)
    if form == "int64_values":
        v = orc.synth_fill(1, 2, 0, n, 60, 18).view(np.int64)
    if form == "single_column":
        v = ids
    if form == "many_groups":
        mod, limit = 70_000, None
        ids = orc.synth_fill(1, 7, 0, n, mod, 0).view(np.int64)
    assert (orc.synth_fill_mt(2, 3, 0, n, threads=5).view(np.float64) == orc.synth_fill(2, 3, 0, n).view(np.float64)).all()
    f = fields("id", "v")
    cols = [Column.from_numpy(ids), Column.from_numpy(v)]
    key = (col(0) if form == "many_groups" else binop(col(0), Operator.Modulos, lit_i64(mod))).flatten(f)
    ref, = orc.aggregate([cols], [(A.Count, 1), (A.Sum, 1), (A.Min, 1), (A.Max, 1)], group_nodes=key,
                         pred_nodes=None if limit is None else binop(col(0), Operator.Lt, lit_i64(limit)).flatten(f))
    e = np.stack([c.to_numpy().astype(np.float64) for c in ref], axis=1)
    e = e[np.lexsort(e.T[::-1])]
    for threads in (1, 7):
        whole = orc.grouped_parallel(ids, v, limit, mod, threads)
        h = n // 3
        merged = orc.merge_grouped([orc.grouped_parallel(ids[:h], v[:h], limit, mod, threads), orc.grouped_parallel(ids[h:], v[h:], limit, mod, threads)])
        for g in (whole, merged):
            g = g[g[:, 0] > 0]                      # keys without rows: not groups
            g = g[np.lexsort(g.T[::-1])]
            assert g.shape == e.shape and (g[:, 0] == e[:, 0]).all() and (g[:, 2:] == e[:, 2:]).all() and np.allclose(g, e, rtol=1e-9, atol=0), (form, threads)


@pytest.mark.parametrize("form", ["readme_three_columns", "nullable_values", "nullable_ids_and_values", "all_null_group"])
def test_parallel_grouped_columns_form_matches_the_port(form):
    """The k-value-column / validity form of the parallel CPU aggregate (oracle.grouped_columns_parallel + finalize_grouped: what the
    full-size checks of the reference's own README query — src/main.rs:36-40 — and of the nullable headline use) against the
    reference-faithful single-threaded port (aggregate/mod.rs:113-222, count.rs, sum.rs, avg.rs, max.rs, min.rs; selection.rs:46 for
    a NULL predicate): keys exact, counts exact, min / max exact, sum / avg within 1e-9 — NaN averages of value-less groups included"""
    from naive_query_engine_amd import AggregateFunc as A
    from naive_query_engine_amd import Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from tests.helpers import fields

    n = 150_001
    ids = orc.synth_fill(0, 0, 0, n).view(np.int64)
    age = orc.synth_fill(1, 2, 0, n, 60, 18).view(np.int64)
    score = orc.synth_fill(2, 3, 0, n).view(np.float64)
    v_ok = orc.synth_fill(1, 4, 0, n, 100, 0).view(np.int64) != 0          # 1 % NULLs
    id_ok = orc.synth_fill(1, 9, 0, n, 50, 0).view(np.int64) != 0
    if form == "readme_three_columns":
        mod, limit, valid, id_valid = 3, None, {}, None
        cols = {0: ids, 1: age, 2: score}
        aggs = [(A.Count, 0), (A.Sum, 1), (A.Sum, 2), (A.Avg, 2), (A.Max, 2), (A.Min, 2)]
    else:
        mod, limit = 1024, n // 2
        cols = {2: score}
        aggs = [(A.Count, 2), (A.Sum, 2), (A.Avg, 2), (A.Min, 2), (A.Max, 2)]
        valid, id_valid = {2: v_ok}, (id_ok if form == "nullable_ids_and_values" else None)
        if form == "all_null_group":     # every value of the keys 5 and 700 is NULL: the groups exist, count 0, avg NaN, min f64::MAX, max f64::MIN
            v_ok = v_ok & (ids % 1024 != 5) & (ids % 1024 != 700)
            valid = {2: v_ok}
    f = fields("id", "age", "score")
    host_cols = [Column.from_numpy(ids, id_valid), Column.from_numpy(age), Column.from_numpy(score, valid.get(2))]
    key = binop(col(0), Operator.Modulos, lit_i64(mod)).flatten(f)
    pred = None if limit is None else binop(col(0), Operator.Lt, lit_i64(limit)).flatten(f)
    ref, = orc.aggregate([host_cols], aggs, group_nodes=key, pred_nodes=pred)       # (the reference's output has no key column: rows as a multiset)
    e = np.stack([c.to_numpy().astype(np.float64) for c in ref], axis=1)
    e = e[np.lexsort(e.T[::-1])]
    exact = [i for i, (func, _) in enumerate(aggs) if func in (A.Count, A.Min, A.Max)]
    for threads in (1, 5):
        h = n // 3
        sl = lambda a, lo, hi: None if a is None else a[lo:hi]
        parts = [orc.grouped_columns_parallel(ids[lo:hi], {c: a[lo:hi] for c, a in cols.items()}, limit, mod, threads,
                                              valid={c: m[lo:hi] for c, m in valid.items()}, id_valid=sl(id_valid, lo, hi)) for lo, hi in ((0, h), (h, n))]
        live, g = orc.finalize_grouped(orc.merge_grouped_columns(parts), aggs)
        assert (live == np.arange(mod)).all(), (form, threads)                  # every key has rows at this size
        gm = np.stack(g, axis=1)
        gm = gm[np.lexsort(gm.T[::-1])]
        assert gm.shape == e.shape, (form, threads)
        assert all((gm[:, i] == e[:, i]).all() for i in exact), (form, threads)
        assert np.allclose(gm, e, rtol=1e-9, atol=0, equal_nan=True), (form, threads)
    if form == "all_null_group":
        i5 = int(np.nonzero(live == 5)[0][0])
        assert g[0][i5] == 0 and g[1][i5] == 0.0 and np.isnan(g[2][i5]) and g[3][i5] == np.finfo(np.float64).max and g[4][i5] == -np.finfo(np.float64).max
