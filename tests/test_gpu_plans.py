"""The reference's own plan-level tests, written against the Python host mirror
(naive_query_engine_amd/physical_plan.py) exactly as the reference writes them
(selection.rs:126-178, projection.rs:88-121, sql/planner.rs:645-713, README.md:69-112), plus checks that the
rewrite pass (rewrite.py: unfused reference-shaped tree → fused device operators) leaves every result as it was."""
import os

import numpy as np
import pytest

from naive_query_engine_amd import (AggregateFunc, Column, ColumnExpr, DType, ErrorCode, Field, Operator, PhysicalBinaryExpr,
                                    PhysicalLiteralExpr, RecordBatch, ScalarValue, Status)
from tests.helpers import assert_batches_equal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pp():
    from naive_query_engine_amd import physical_plan

    return physical_plan


def numeric_table(pp, batch, names):
    idx = {f.name: i for i, f in enumerate(batch.fields)}
    fields = [batch.fields[idx[n]] for n in names]
    return pp.MemTable.try_create(fields, [RecordBatch(fields, [batch.columns[idx[n]] for n in names])])


@pytest.fixture(scope="module")
def t1(pp, csv_tables):
    return numeric_table(pp, csv_tables["test_data"], ["id", "age", "score"])


def lit(v):
    return PhysicalLiteralExpr.create(ScalarValue.Int64(v))


def test_selection_like_the_reference(pp, t1, golden):
    schema = [t1.schema()[0], t1.schema()[1]]
    scan = pp.ScanPlan.create(t1, None)
    proj = pp.ProjectionPlan.create(scan, schema, [ColumnExpr.try_create(None, 0), ColumnExpr.try_create("age", None)])
    add_expr = PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None), Operator.Plus, lit(1))
    expr = PhysicalBinaryExpr.create(add_expr, Operator.Gt, lit(5))
    res = pp.SelectionPlan.create(proj, expr).execute()
    assert len(res) == 1
    assert res[0].column(0).to_list() == golden["test_selection"]["id"]


def test_projection_like_the_reference(pp, t1, golden):
    schema = [t1.schema()[0], t1.schema()[2]]
    add_expr = PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None), Operator.Plus, lit(1))
    res = pp.ProjectionPlan.create(pp.ScanPlan.create(t1, None), schema, [add_expr, ColumnExpr.try_create("score", None)]).execute()
    assert res[0].column(0).to_list() == golden["test_projection"]["id_plus_1"]
    assert res[0].to_host().fields[0].name == "id"


def test_readme_query1_rewritten_equals_unfused(pp, t1, golden):
    scan = pp.ScanPlan.create(t1, None)
    pred = PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Lt, lit(9))
    exprs = [ColumnExpr.try_create(None, 0), PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 1), Operator.Plus, lit(100))]
    schema = [t1.schema()[0], Field("age + 100", DType.INT64, True)]
    from naive_query_engine_amd.rewrite import plan_shape, rewrite

    chain = pp.ProjectionPlan.create(pp.SelectionPlan.create(scan, pred), schema, exprs)
    plan = pp.PhysicalLimitPlan.create(pp.PhysicalOffsetPlan.create(chain, 2), 3)
    assert plan_shape(plan) == ["PhysicalLimitPlan", "PhysicalOffsetPlan", "ProjectionPlan", "SelectionPlan", "ScanPlan"]
    fused = rewrite(plan)
    assert plan_shape(fused) == ["PhysicalLimitPlan", "PhysicalOffsetPlan", "FusedSelectionProjectionPlan", "ScanPlan"]
    for p in (plan, fused):
        out = p.execute()
        rows = list(zip(out[0].column(0).to_list(), out[0].column(1).to_list()))
        assert rows == [(r[0], r[2]) for r in golden["readme_filter_project_offset_limit"]["rows"]]
    assert_batches_equal(rewrite(chain).execute()[0].table.to_host(), chain.execute()[0].table.to_host())


def test_readme_aggregate_and_q9(pp, t1, golden):
    ops = [pp.Count.create(ColumnExpr.try_create(None, 0)), pp.Sum.create(ColumnExpr.try_create("age", None)), pp.Sum.create(ColumnExpr.try_create(None, 2)),
           pp.Avg.create(ColumnExpr.try_create(None, 2)), pp.Max.create(ColumnExpr.try_create(None, 2)), pp.Min.create(ColumnExpr.try_create(None, 2))]
    key = PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Modulos, lit(3))
    agg = pp.PhysicalAggregatePlan.create([key], ops, pp.ScanPlan.create(t1, None))
    res = pp.ProjectionPlan.create(agg, [], []).execute()  # empty schema → pass-through (projection.rs:47-48)
    host = res[0].to_host()
    assert [f.name for f in host.fields] == ["count(id)", "sum(age)", "sum(score)", "avg(score)", "max(score)", "min(score)"]
    rows = np.array(sorted(map(list, zip(*[c.to_list() for c in host.columns]))), dtype=np.float64)
    exp = np.array(sorted(golden["readme_group_by_id_mod_3"]["rows"]), dtype=np.float64)
    assert (rows[:, 0] == exp[:, 0]).all() and np.allclose(rows, exp, rtol=1e-9, atol=0)
    assert agg.schema() == t1.schema()  # quirk Q8: schema() is the INPUT schema
    # the rewrite pass folds the filter into the aggregate: same result as the plain chain
    from naive_query_engine_amd.rewrite import plan_shape, rewrite

    pred = PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Gt, lit(2))
    mk = lambda: [pp.Count.create(ColumnExpr.try_create(None, 2)), pp.Sum.create(ColumnExpr.try_create(None, 2))]
    chain = pp.PhysicalAggregatePlan.create([key], mk(), pp.SelectionPlan.create(pp.ScanPlan.create(t1, None), pred))
    fused = rewrite(pp.ProjectionPlan.create(chain, [], []))
    assert plan_shape(fused) == ["ProjectionPlan", "FusedSelectionAggregatePlan", "ScanPlan"]
    assert_batches_equal(fused.execute()[0].table.to_host(), chain.execute()[0].table.to_host(), rtol=1e-12)
    # quirk Q9: the un-grouped state is never cleared between execute() calls
    un = pp.PhysicalAggregatePlan.create([], [pp.Count.create(ColumnExpr.try_create(None, 0)), pp.Sum.create(ColumnExpr.try_create(None, 0))], pp.ScanPlan.create(t1, None))
    assert [c.to_list() for c in un.execute()[0].table.to_host()] == [[8], [42.0]]
    assert [c.to_list() for c in un.execute()[0].table.to_host()] == [[16], [84.0]]


def test_readme_joins_row_order(pp, csv_tables, golden):
    emp = numeric_table(pp, csv_tables["employee"], ["id", "department_id", "rank"])
    rank_b = csv_tables["rank"]
    rank = pp.MemTable.try_create([Field("rank_id", DType.INT64)], [RecordBatch([Field("rank_id", DType.INT64)], [rank_b.columns[0]])])
    dep_b = csv_tables["department"]
    dep = pp.MemTable.try_create([Field("dep_id", DType.INT64)], [RecordBatch([Field("dep_id", DType.INT64)], [dep_b.columns[0]])])
    j1 = pp.HashJoin.create(pp.ScanPlan.create(emp, None), pp.ScanPlan.create(rank, None), [(pp.ColumnRef("employee", "rank"), pp.ColumnRef("rank", "rank_id"))],
                            pp.JoinType.Inner, [])
    j2 = pp.HashJoin.create(j1, pp.ScanPlan.create(dep, None), [(pp.ColumnRef(None, "department_id"), pp.ColumnRef(None, "dep_id"))], pp.JoinType.Inner, [])
    out = j2.execute()
    assert len(out) == 1
    assert out[0].column(0).to_list() == [r[0] for r in golden["readme_two_hash_joins"]["rows"]]
    with pytest.raises(ErrorCode) as e:
        pp.HashJoin.create(j1, j1, [], pp.JoinType.Inner, []).execute()
    assert e.value.status == Status.PlanError


def test_hash_join_reexecution_duplicates_matches_q11(pp):
    """the reference never clears HashJoin's hash table: executing the same plan object again emits every match twice,
    then three times (hash_join.rs:58-78, :138-156); the oracle keeps the table across executes the same way"""
    from oracle import oracle as orc

    rng = np.random.default_rng(17)
    nb, npr = 300, 2000
    left = [Column.from_numpy(rng.integers(0, 120, nb).astype(np.int64)), Column.from_numpy(rng.random(nb))]      # duplicate build keys
    right = [Column.from_numpy(rng.integers(-5, 130, npr).astype(np.int64)), Column.from_numpy(rng.integers(0, 9, npr).astype(np.int64))]
    lf, rf = [Field("id", DType.INT64, False), Field("x", DType.FLOAT64, False)], [Field("key", DType.INT64, False), Field("y", DType.INT64, False)]
    lt = pp.MemTable.try_create(lf, [RecordBatch(lf, left)])
    rt = pp.MemTable.try_create(rf, [RecordBatch(rf, right)])
    join = pp.HashJoin.create(pp.ScanPlan.create(lt, None), pp.ScanPlan.create(rt, None), [(pp.ColumnRef(None, "id"), pp.ColumnRef(None, "key"))], pp.JoinType.Inner, lf + rf)
    for k in (1, 2, 3):
        got = join.execute()[0].to_host().columns
        exp = orc.hash_join([left], [right], 0, 0, executions=k)[0]
        assert got[0].length == k * orc.hash_join([left], [right], 0, 0)[0][0].length
        assert_batches_equal(got, exp, what=f"execute() #{k}")


def test_multi_batch_selection_q3_and_scan_projection(pp):
    f = [Field("x", DType.INT64, True)]
    b0 = RecordBatch(f, [Column.from_list([1, None, 9], DType.INT64)])
    b1 = RecordBatch(f, [Column.from_list([7, 0, 8, 100], DType.INT64)])
    t = pp.MemTable.try_create(f, [b0, b1])
    res = pp.SelectionPlan.create(pp.ScanPlan.create(t, None), PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Gt, lit(4))).execute()
    assert [r.column(0).to_list() for r in res] == [[None, 9], [None, 8]]
    f3 = [Field("a", DType.INT64), Field("b", DType.INT64), Field("c", DType.INT64)]
    m = pp.MemTable.try_create(f3, [RecordBatch(f3, [Column.from_list([1, 2, 3], DType.INT64), Column.from_list([4, 5, 6], DType.INT64), Column.from_list([7, 8, 9], DType.INT64)])])
    b = pp.ScanPlan.create(m, [2, 1]).execute()[0]
    assert [x.name for x in b.fields] == ["c", "b"] and b.column(0).to_list() == [7, 8, 9]


def test_reference_tests_with_utf8_names(pp, csv_tables, golden):
    """the same reference tests with the Utf8 `name` column kept (selection.rs:166-172, projection.rs:113-118,
    README.md:70-85)"""
    b = csv_tables["test_data"]
    t = pp.MemTable.try_create(b.fields, [b])
    scan = pp.ScanPlan.create(t, None)
    schema = [b.fields[0], b.fields[1], b.fields[2]]
    proj = pp.ProjectionPlan.create(scan, schema, [ColumnExpr.try_create(None, 0), ColumnExpr.try_create("name", None), ColumnExpr.try_create(None, 2)])
    expr = PhysicalBinaryExpr.create(PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None), Operator.Plus, lit(1)), Operator.Gt, lit(5))
    res = pp.SelectionPlan.create(proj, expr).execute()
    assert res[0].column(0).to_list() == golden["test_selection"]["id"]
    assert res[0].column(1).to_list() == golden["test_selection"]["name"]
    # README query 1 with names
    pred = PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Lt, lit(9))
    exprs = [ColumnExpr.try_create(None, 0), ColumnExpr.try_create(None, 1), PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 2), Operator.Plus, lit(100))]
    q1 = pp.PhysicalLimitPlan.create(pp.PhysicalOffsetPlan.create(pp.ProjectionPlan.create(pp.SelectionPlan.create(scan, pred), schema, exprs), 2), 3).execute()
    assert list(map(list, zip(*[q1[0].column(i).to_list() for i in range(3)]))) == golden["readme_filter_project_offset_limit"]["rows"]
    # README query 2: joins resolved by NAME; `id` is ambiguous and resolves to the first match (quirk Q12)
    emp, rank, dep = (pp.MemTable.try_create(csv_tables[k].fields, [csv_tables[k]]) for k in ("employee", "rank", "department"))
    j1 = pp.HashJoin.create(pp.ScanPlan.create(emp, None), pp.ScanPlan.create(rank, None), [(pp.ColumnRef("employee", "rank"), pp.ColumnRef("rank", "id"))], pp.JoinType.Inner, [])
    j2 = pp.HashJoin.create(j1, pp.ScanPlan.create(dep, None), [(pp.ColumnRef("employee", "department_id"), pp.ColumnRef("department", "id"))], pp.JoinType.Inner, [])
    out = j2.execute()[0]
    names = [f.name for f in out.fields]
    pick = [names.index("id"), names.index("name"), names.index("rank_name"), names.index("department_name")]
    assert list(map(list, zip(*[out.column(i).to_list() for i in pick]))) == golden["readme_two_hash_joins"]["rows"]


def test_rewrite_keeps_quirks_q3_q9_q11(pp):
    """the fused operators fall back to the plain chain where the reference's behaviour depends on the tree being unfused
    (several input batches: the predicate comes from batch 0, Q3) and keep the per-operator state (Q9, Q11)"""
    from naive_query_engine_amd.rewrite import NaiveDB, plan_shape, rewrite

    f = [Field("x", DType.INT64, True), Field("y", DType.INT64, True)]
    b0 = RecordBatch(f, [Column.from_list([1, None, 9], DType.INT64), Column.from_list([10, 20, 30], DType.INT64)])
    b1 = RecordBatch(f, [Column.from_list([7, 0, 8, 100], DType.INT64), Column.from_list([1, 2, 3, 4], DType.INT64)])
    db = NaiveDB()
    db.create_memory_table("t", f, [b0, b1])
    pred = PhysicalBinaryExpr.create(ColumnExpr.try_create(None, 0), Operator.Gt, lit(4))
    proj = lambda: pp.ProjectionPlan.create(pp.SelectionPlan.create(db.scan("t"), pred), [f[1]],
                                            [PhysicalBinaryExpr.create(ColumnExpr.try_create("y", None), Operator.Plus, lit(1))])
    plain = [r.column(0).to_list() for r in proj().execute()]
    assert plain == [[None, 31], [None, 4]]                                   # Q3: batch 0's mask zipped against batch 1
    assert [r.column(0).to_list() for r in db.run_plan(proj())] == plain      # run_plan = rewrite + execute
    assert plan_shape(rewrite(proj())) == ["FusedSelectionProjectionPlan", "ScanPlan"]
    # grouped aggregate over the filtered two-batch table
    agg = lambda: pp.PhysicalAggregatePlan.create([ColumnExpr.try_create("x", None)], [pp.Sum.create(ColumnExpr.try_create("y", None))],
                                                  pp.SelectionPlan.create(db.scan("t"), pred))
    a, b = agg().execute()[0].table.to_host(), db.run_plan(agg())[0].table.to_host()
    assert_batches_equal(a, b)
    # Q9 on a rewritten tree: the un-grouped state survives between execute() calls of the same (rewritten) plan object
    un = rewrite(pp.PhysicalAggregatePlan.create([], [pp.Count.create(ColumnExpr.try_create("y", None))],
                                                 pp.SelectionPlan.create(db.scan("t"), PhysicalBinaryExpr.create(ColumnExpr.try_create("y", None), Operator.Gt, lit(0)))))
    first = un.execute()[0].column(0).to_list()
    assert un.execute()[0].column(0).to_list() == [2 * first[0]]
    with pytest.raises(ErrorCode) as e:
        db.scan("nope")
    assert e.value.status == Status.NoSuchTable
    with pytest.raises(ErrorCode):
        db.run_sql("select 1")


def test_arrow_tables_enter_through_the_catalog(pp):
    pa = pytest.importorskip("pyarrow")
    from naive_query_engine_amd.rewrite import NaiveDB

    rb = pa.RecordBatch.from_arrays([pa.array([1, 2, 3, 4, 5], pa.int64()), pa.array([1.5, None, 3.5, 4.5, 5.5], pa.float64()), pa.array(["a", "b", None, "d", "e"])],
                                    names=["id", "v", "s"])
    db = NaiveDB()
    db.create_arrow_table("t", [rb])
    plan = pp.ProjectionPlan.create(pp.SelectionPlan.create(db.scan("t"), PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None), Operator.Gt, lit(2))),
                                    [Field("v", DType.FLOAT64, True), Field("s", DType.UTF8, True)], [ColumnExpr.try_create("v", None), ColumnExpr.try_create("s", None)])
    out = db.run_plan(plan)[0].table.to_arrow(names=["v", "s"])
    assert out.to_pydict() == {"v": [3.5, 4.5, 5.5], "s": [None, "d", "e"]}
