"""GPU parity tests: every hot-path operator through the C ABI (libnqe_hip.so) against the CPU
oracle on the same seeded inputs.  Bit-exact for integers/booleans/validity/row order; Float64
sums/avgs within 1e-9 relative (north_star tolerance)."""
import os

import numpy as np
import pytest

from naive_query_engine_amd import AggregateFunc, Column, DType, ErrorCode, Operator, Status
from naive_query_engine_amd.expression import binop, col, lit_bool, lit_f64, lit_i64, lit_u64, lit_utf8
from oracle import oracle as orc
from tests.helpers import assert_batches_equal, assert_column_equal, assert_rows_multiset_equal, fields, random_batch, random_utf8, rows_sorted

pytestmark = pytest.mark.gpu

FLD = fields("id", "k", "v", "u", "b")
RTOL = 1e-9
ALL_AGGS = lambda c: [(AggregateFunc.Count, c), (AggregateFunc.Sum, c), (AggregateFunc.Avg, c), (AggregateFunc.Min, c), (AggregateFunc.Max, c)]


@pytest.fixture(scope="module")
def ctx():
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def flat(e):
    return e.flatten(FLD)


# --------------------------------------------------------------------------- expressions
ARITH = [Operator.Plus, Operator.Minus, Operator.Multiply, Operator.Divide, Operator.Modulos]
CMP = [Operator.Eq, Operator.NotEq, Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq]


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_binary_ops_all_types(ctx, n, null_frac):
    rng = np.random.default_rng(n * 7 + int(null_frac * 10))
    cols = random_batch(rng, n, null_frac, with_bool=True)
    t = ctx.table_from_host(cols)
    lits = {1: lit_i64(-7), 2: lit_f64(3.25), 3: lit_u64(12)}
    for c, lit in lits.items():
        for op in ARITH + CMP:
            for e in (binop(col(c), op, lit), binop(lit, op, col(c)), binop(col(c), op, col(c))):
                try:
                    exp = orc.expr_evaluate([cols], flat(e))
                except ErrorCode as err:  # e.g. lit % col with a zero in col
                    with pytest.raises(ErrorCode) as g:
                        ctx.expr_evaluate(t, flat(e))
                    assert g.value.status == err.status
                    continue
                got = ctx.expr_evaluate(t, flat(e)).to_host()[0]
                assert_column_equal(got, exp, what=f"{e!r}")
    # Boolean compares and Kleene logic
    for op in CMP + [Operator.And, Operator.Or]:
        for e in (binop(col(4), op, col(4)), binop(col(4), op, lit_bool(True)), binop(lit_bool(False), op, col(4)),
                  binop(binop(col(1), Operator.Gt, lit_i64(0)), op, binop(col(2), Operator.Lt, lit_f64(10.0)))):
            assert_column_equal(ctx.expr_evaluate(t, flat(e)).to_host()[0], orc.expr_evaluate([cols], flat(e)), what=f"{e!r}")


def test_expression_edge_values(ctx):
    i64 = np.array([np.iinfo(np.int64).min, -1, 0, 1, np.iinfo(np.int64).max, -1024, 1023, 5], dtype=np.int64)
    f64 = np.array([np.nan, -0.0, 0.0, np.inf, -np.inf, 1e308, -1e-308, 2.5])
    cols = [Column.from_numpy(i64), Column.from_numpy(f64)]
    t = ctx.table_from_host(cols)
    f = fields("a", "x")
    cases = [binop(col(0), Operator.Plus, lit_i64(1)), binop(col(0), Operator.Minus, lit_i64(2)),
             binop(col(0), Operator.Multiply, lit_i64(3)), binop(col(0), Operator.Modulos, lit_i64(1024)),
             binop(col(0), Operator.Modulos, lit_i64(-1024)), binop(col(0), Operator.Divide, lit_i64(1024)),
             binop(col(0), Operator.Divide, lit_i64(-8)), binop(col(0), Operator.Modulos, lit_i64(7)),
             binop(col(0), Operator.Divide, lit_i64(7)), binop(col(0), Operator.Modulos, lit_i64(np.iinfo(np.int64).min)),
             binop(lit_i64(100), Operator.Minus, col(0))]
    cases += [binop(col(1), op, lit_f64(0.0)) for op in CMP] + [binop(col(1), op, col(1)) for op in CMP]
    cases += [binop(col(1), Operator.Plus, lit_f64(1.0)), binop(col(1), Operator.Multiply, lit_f64(-2.0)),
              binop(col(1), Operator.Divide, lit_f64(3.0)), binop(col(1), Operator.Modulos, lit_f64(2.0))]
    for e in cases:
        assert_column_equal(ctx.expr_evaluate(t, e.flatten(f)).to_host()[0], orc.expr_evaluate([cols], e.flatten(f)), what=f"{e!r}")
    # i64::MIN / -1 overflows in Rust → error on both sides
    for op in (Operator.Divide, Operator.Modulos):
        e = binop(col(0), op, lit_i64(-1)).flatten(f)
        with pytest.raises(ErrorCode) as a:
            orc.expr_evaluate([cols], e)
        with pytest.raises(ErrorCode) as b:
            ctx.expr_evaluate(t, e)
        assert a.value.status == b.value.status == Status.ArrowError


def test_divide_and_modulo_by_literal_divisors(ctx):
    """literal divisors take the magic-multiply path (non powers of two) or the shift/mask path (±2^k)"""
    rng = np.random.default_rng(2024)
    i64 = np.concatenate([rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, 4000, dtype=np.int64),
                          np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).min + 1, -1, 0, 1, np.iinfo(np.int64).max], dtype=np.int64),
                          rng.integers(-5000, 5000, 2000).astype(np.int64)])
    u64 = np.concatenate([rng.integers(0, np.iinfo(np.uint64).max, 4000, dtype=np.uint64), np.array([0, 1, (1 << 64) - 1, 1 << 63], dtype=np.uint64)])
    f = fields("a")
    ti, tu = ctx.table_from_host([Column.from_numpy(i64)]), ctx.table_from_host([Column.from_numpy(u64)])
    for d in [3, 5, 7, 10, 1000, 1023, 1025, (1 << 31) + 1, (1 << 62) + 3, np.iinfo(np.int64).max, -3, -1000, -((1 << 40) + 7), 2, -2, 1 << 20, 1]:
        for op in (Operator.Divide, Operator.Modulos):
            e = binop(col(0), op, lit_i64(int(d))).flatten(f)
            assert_column_equal(ctx.expr_evaluate(ti, e).to_host()[0], orc.expr_evaluate([[Column.from_numpy(i64)]], e), what=f"i64 {op.name} {d}")
    for d in [3, 7, 1000, (1 << 63) + 5, (1 << 64) - 1, (1 << 33) - 1, 1 << 63, 6]:
        for op in (Operator.Divide, Operator.Modulos):
            e = binop(col(0), op, lit_u64(int(d))).flatten(f)
            assert_column_equal(ctx.expr_evaluate(tu, e).to_host()[0], orc.expr_evaluate([[Column.from_numpy(u64)]], e), what=f"u64 {op.name} {d}")
    # as group keys (fast aggregate kernel, magic modulo) incl. negative dividends
    v = Column.from_numpy(rng.random(i64.size))
    t2 = ctx.table_from_host([Column.from_numpy(i64), v])
    f2 = fields("a", "v")
    for d in (3, 1000, -7):
        key = binop(col(0), Operator.Modulos, lit_i64(d)).flatten(f2)
        exp = orc.aggregate([[Column.from_numpy(i64), v]], ALL_AGGS(1), group_nodes=key)[0]
        got, gk = ctx.aggregate(t2, ALL_AGGS(1), group_nodes=key, with_keys=True)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"group by a % {d}")
        assert gk.to_host()[0].to_list() == sorted(set(int(np.fmod(float(0), 1)) if False else int(abs(int(x)) % abs(d) * (1 if x >= 0 else -1)) for x in i64.tolist()))


def _random_tree(rng, depth, want):
    """random well-typed tree over FLD: want in {'i','f','u','b'}; leaves are columns or non-zero literals"""
    leaf_col = {"i": [0, 1], "f": [2], "u": [3], "b": [4]}[want]
    def leaf():
        if rng.random() < 0.6:
            return col(int(rng.choice(leaf_col)))
        if want == "i":
            return lit_i64(int(rng.choice([-9, -2, 1, 3, 8, 1000])))
        if want == "f":
            return lit_f64(float(rng.choice([-2.5, 0.5, 3.0, 10.0])))
        if want == "u":
            return lit_u64(int(rng.choice([1, 2, 7, 1 << 20])))
        return lit_bool(bool(rng.random() < 0.5))
    if depth == 0:
        return leaf()
    if want == "b":
        if rng.random() < 0.5:
            return binop(_random_tree(rng, depth - 1, "b"), Operator.And if rng.random() < 0.5 else Operator.Or, _random_tree(rng, depth - 1, "b"))
        t = str(rng.choice(["i", "f", "u", "b"]))
        return binop(_random_tree(rng, depth - 1, t), CMP[int(rng.integers(0, len(CMP)))], _random_tree(rng, depth - 1, t))
    # keep Divide/Modulos to literal divisors so that DivideByZero cannot fire on random data
    op = ARITH[int(rng.integers(0, 3))]
    if rng.random() < 0.3:
        div = {"i": lit_i64(int(rng.choice([-7, 3, 16, 1000]))), "f": lit_f64(2.5), "u": lit_u64(int(rng.choice([3, 64, 1000])))}[want]
        return binop(_random_tree(rng, depth - 1, want), Operator.Divide if rng.random() < 0.5 else Operator.Modulos, div)
    return binop(_random_tree(rng, depth - 1, want), op, _random_tree(rng, depth - 1, want))


@pytest.mark.parametrize("null_frac", [0.0, 0.25])
def test_random_expression_trees_fused_and_deep(ctx, null_frac):
    """trees with >= 2 binary nodes run in the single-pass stack machine (depth <= 4, <= 24 nodes); deeper ones node-at-a-time"""
    rng = np.random.default_rng(77 + int(null_frac * 100))
    cols = random_batch(rng, 3000, null_frac, with_bool=True)
    t = ctx.table_from_host(cols)
    n_checked = 0
    for trial in range(120):
        e = _random_tree(rng, int(rng.integers(1, 6)), str(rng.choice(["i", "f", "u", "b"])))
        nodes = flat(e)
        try:
            exp = orc.expr_evaluate([cols], nodes)
        except ErrorCode as err:
            with pytest.raises(ErrorCode) as g:
                ctx.expr_evaluate(t, nodes)
            assert g.value.status == err.status, repr(e)
            continue
        assert_column_equal(ctx.expr_evaluate(t, nodes).to_host()[0], exp, what=repr(e))
        n_checked += 1
    assert n_checked > 60
    # the same trees as predicates / projections inside the fused operators
    pred = binop(binop(binop(col(1), Operator.Plus, col(0)), Operator.Modulos, lit_i64(5)), Operator.Lt, binop(col(1), Operator.Multiply, lit_i64(2)))
    proj = [binop(binop(col(2), Operator.Multiply, col(2)), Operator.Plus, binop(col(2), Operator.Divide, lit_f64(4.0))), col(3)]
    sel = orc.selection([cols], flat(pred))
    exp = orc.projection(sel, [flat(x) for x in proj])[0]
    assert_batches_equal(ctx.selection_projection(t, flat(pred), [flat(x) for x in proj]).to_host(), exp, what="tree predicate + tree projection")
    aggs = ALL_AGGS(2)
    expa = orc.aggregate([cols], aggs, group_nodes=flat(binop(binop(col(1), Operator.Plus, lit_i64(3)), Operator.Multiply, binop(col(1), Operator.Minus, lit_i64(1)))), pred_nodes=flat(pred))[0]
    gota = ctx.aggregate(t, aggs, group_nodes=flat(binop(binop(col(1), Operator.Plus, lit_i64(3)), Operator.Multiply, binop(col(1), Operator.Minus, lit_i64(1)))), pred_nodes=flat(pred))
    assert_rows_multiset_equal(gota.to_host(), expa, RTOL, exact_cols=[0], what="tree key + tree predicate aggregate")


def test_expression_errors(ctx):
    cols = [Column.from_list([1, 0, 3], DType.INT64), Column.from_list([1.0, 0.0, None], DType.FLOAT64),
            Column.from_list([True, None, False], DType.BOOLEAN)]
    t = ctx.table_from_host(cols)
    f = fields("a", "x", "b")
    bad = [
        (binop(col(0), Operator.Lt, lit_f64(4.5)), Status.IntervalError),       # Q6: no coercion
        (binop(col(0), Operator.And, col(0)), Status.IntervalError),            # and/or need Boolean
        (binop(col(2), Operator.Plus, col(2)), Status.NotSupported),            # arithmetic on Boolean panics
        (binop(lit_i64(6), Operator.Divide, col(0)), Status.ArrowError),        # DivideByZero
        (binop(col(1), Operator.Modulos, col(1)), Status.ArrowError),           # float zero divisor too
        (binop(col(0), Operator.Modulos, lit_i64(0)), Status.ArrowError),
    ]
    for e, status in bad:
        with pytest.raises(ErrorCode) as a:
            orc.expr_evaluate([cols], e.flatten(f))
        with pytest.raises(ErrorCode) as b:
            ctx.expr_evaluate(t, e.flatten(f))
        assert a.value.status == b.value.status == status, repr(e)
    # a zero divisor under a NULL slot is fine
    n = [Column.from_list([10, None], DType.INT64), Column.from_list([2, 0], DType.INT64)]
    e = binop(col(0), Operator.Divide, col(1)).flatten(f)
    assert_column_equal(ctx.expr_evaluate(ctx.table_from_host(n), e).to_host()[0], orc.expr_evaluate([n], e))
    # NULL literal and literal-only expressions
    for e in (binop(col(0), Operator.Plus, lit_i64(None)), binop(lit_i64(2), Operator.Multiply, lit_i64(21)), lit_f64(1.5), col(1)):
        assert_column_equal(ctx.expr_evaluate(t, e.flatten(f)).to_host()[0], orc.expr_evaluate([cols], e.flatten(f)), what=repr(e))


# --------------------------------------------------------------------------- filter / projection
PREDS = [
    binop(col(0), Operator.Lt, lit_i64(500)),                                                  # fused shape
    binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Gt, lit_i64(5)),                  # reference test shape
    binop(binop(col(1), Operator.Modulos, lit_i64(3)), Operator.Eq, lit_i64(0)),
    binop(col(2), Operator.GtEq, lit_f64(0.0)),
    binop(binop(col(0), Operator.Lt, lit_i64(700)), Operator.And, binop(col(2), Operator.Lt, lit_f64(50.0))),   # general
    binop(binop(col(1), Operator.Gt, lit_i64(0)), Operator.Or, binop(col(3), Operator.Lt, lit_u64(1 << 39))),
    binop(col(1), Operator.Gt, col(0)),
]


@pytest.mark.parametrize("null_frac", [0.0, 0.15])
@pytest.mark.parametrize("n", [0, 1, 64, 4095, 4096, 4097, 20000])
def test_selection_matches_oracle(ctx, n, null_frac):
    rng = np.random.default_rng(1000 + n)
    cols = random_batch(rng, n, null_frac, with_bool=True)
    t = ctx.table_from_host(cols)
    for p in PREDS:
        exp = orc.selection([cols], flat(p))[0]
        got = ctx.selection(t, flat(p)).to_host()
        assert_batches_equal(got, exp, what=f"selection {p!r} n={n}")


def test_filter_zip_truncation_q3(ctx):
    rng = np.random.default_rng(5)
    b0 = random_batch(rng, 100)
    b1 = random_batch(rng, 150)
    p = binop(col(1), Operator.Gt, lit_i64(0))
    exp = orc.selection([b0, b1], flat(p))
    t0, t1 = ctx.table_from_host(b0), ctx.table_from_host(b1)
    mask = ctx.expr_evaluate(t0, flat(p))                       # predicate from batch 0 only
    assert_batches_equal(ctx.filter(t0, mask).to_host(), exp[0])
    assert_batches_equal(ctx.filter(t1, mask).to_host(), exp[1])  # zipped against batch 1, truncated to 100 rows


@pytest.mark.parametrize("null_frac", [0.0, 0.15])
def test_selection_projection_fused_equals_chained(ctx, null_frac):
    rng = np.random.default_rng(77)
    cols = random_batch(rng, 30000, null_frac, with_bool=True)
    t = ctx.table_from_host(cols)
    exprs = [binop(col(1), Operator.Plus, lit_i64(100)), col(0), binop(col(2), Operator.Multiply, lit_f64(0.5)),
             binop(col(3), Operator.Modulos, lit_u64(1000)), binop(col(1), Operator.LtEq, lit_i64(3)), col(4)]
    general = exprs + [binop(binop(col(0), Operator.Plus, col(1)), Operator.Multiply, lit_i64(2)),
                       binop(lit_i64(100), Operator.Divide, col(0))]   # would divide by zero on the row id == 0
    for p in PREDS[:4]:
        sel = orc.selection([cols], flat(p))
        exp = orc.projection(sel, [flat(e) for e in exprs])[0]
        got = ctx.selection_projection(t, flat(p), [flat(e) for e in exprs]).to_host()
        assert_batches_equal(got, exp, what=f"fused {p!r}")
    # general expressions: evaluated after compaction, so the filter protects the divisor
    p = binop(col(0), Operator.Gt, lit_i64(0))
    if null_frac == 0.0:
        sel = orc.selection([cols], flat(p))
        exp = orc.projection(sel, [flat(e) for e in general])[0]
        got = ctx.selection_projection(t, flat(p), [flat(e) for e in general]).to_host()
        assert_batches_equal(got, exp, what="general projection")


def test_tree_projection_under_selection_errors_only_on_kept_rows(ctx):
    """two-column trees run fused with the compaction; a zero divisor in a dropped row must not raise (reference: projection sees the filtered batch)"""
    rng = np.random.default_rng(31)
    n = 20000
    cols = random_batch(rng, n, 0.1, key_mod=5, with_bool=True)   # k in 0..4 with nulls
    t = ctx.table_from_host(cols)
    proj = [binop(binop(col(0), Operator.Divide, col(1)), Operator.Plus, col(0)), binop(col(0), Operator.Modulos, col(1)),
            binop(binop(col(2), Operator.Lt, lit_f64(0.0)), Operator.Or, binop(col(0), Operator.Gt, col(1))), col(3)]
    ok_pred = binop(col(1), Operator.NotEq, lit_i64(0))
    sel = orc.selection([cols], flat(ok_pred))
    exp = orc.projection(sel, [flat(e) for e in proj])[0]
    assert_batches_equal(ctx.selection_projection(t, flat(ok_pred), [flat(e) for e in proj]).to_host(), exp, what="kept rows only")
    bad_pred = binop(col(1), Operator.LtEq, lit_i64(2))   # keeps k == 0 rows
    with pytest.raises(ErrorCode) as a:
        orc.projection(orc.selection([cols], flat(bad_pred)), [flat(e) for e in proj])
    with pytest.raises(ErrorCode) as b:
        ctx.selection_projection(t, flat(bad_pred), [flat(e) for e in proj])
    assert a.value.status == b.value.status
    # nothing kept / everything kept
    for pred in (binop(col(1), Operator.Gt, lit_i64(100)), binop(col(1), Operator.Gt, lit_i64(0))):
        sel = orc.selection([cols], flat(pred))
        exp = orc.projection(sel, [flat(e) for e in proj])[0]
        assert_batches_equal(ctx.selection_projection(t, flat(pred), [flat(e) for e in proj]).to_host(), exp, what="edge selectivity")


@pytest.mark.parametrize("null_frac", [0.0, 0.15])
def test_aggregate_with_boolean_and_tree_predicates(ctx, null_frac):
    """non-null Boolean predicates (a Boolean column, or a tree evaluated to a bitmap) feed the fast kernels bit-wise"""
    rng = np.random.default_rng(5 + int(null_frac * 100))
    n = 300_000
    cols = random_batch(rng, n, null_frac, key_mod=300, with_bool=True)
    t = ctx.table_from_host(cols)
    tree = binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5))
    both = binop(col(4), Operator.And, binop(col(2), Operator.Gt, lit_f64(-20.0)))
    for pred in (col(4), tree, both):
        for key in (col(1), binop(col(0), Operator.Modulos, lit_i64(64)), binop(col(0), Operator.Modulos, lit_i64(1000)), None):
            aggs = ALL_AGGS(2) if key is not None else ALL_AGGS(2) + ALL_AGGS(3)
            exp = orc.aggregate([cols], aggs, group_nodes=flat(key) if key is not None else None, pred_nodes=flat(pred))[0]
            got = ctx.aggregate(t, aggs, group_nodes=flat(key) if key is not None else None, pred_nodes=flat(pred))
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"pred {pred!r} key {key!r}")


def test_float_predicates_take_the_range_paths(ctx):
    """`x op c` on Float64 runs as an integer range test over the order-preserving image of the double (selection keep
    mask, fast aggregate kernels): NaN, ±0, ±inf, subnormals on both sides, all six operators, literal on either side"""
    special = np.array([np.nan, -np.nan, 0.0, -0.0, np.inf, -np.inf, 5e-324, -5e-324, 1.0, -1.0, 2.5, -2.5, 1e308, -1e308, 50.0, 49.99999999999999])
    rng = np.random.default_rng(8)
    x = np.concatenate([special, rng.random(20000) * 100.0 - 50.0, rng.choice(special, 2000)])
    n = x.size
    cols = [Column.from_numpy(x), Column.from_numpy(rng.integers(0, 50, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
    f = fields("x", "k", "v")
    t = ctx.table_from_host(cols)
    lits = [0.0, -0.0, 50.0, -1.0, 2.5, float("inf"), float("-inf"), float("nan"), 5e-324, 1e308]
    for c in lits:
        for op in CMP:
            for pred in (binop(col(0), op, lit_f64(c)), binop(lit_f64(c), op, col(0))):
                exp = orc.selection([cols], pred.flatten(f))[0]
                assert_batches_equal(ctx.selection(t, pred.flatten(f)).to_host(), exp, what=f"selection {pred!r}")
                expa = orc.aggregate([cols], ALL_AGGS(2), group_nodes=col(1).flatten(f), pred_nodes=pred.flatten(f))[0]
                gota = ctx.aggregate(t, ALL_AGGS(2), group_nodes=col(1).flatten(f), pred_nodes=pred.flatten(f))
                assert_rows_multiset_equal(gota.to_host(), expa, RTOL, exact_cols=[0], what=f"aggregate {pred!r}")
    # predicate on the aggregated column itself, and un-grouped
    pred = binop(col(2), Operator.Gt, lit_f64(0.25))
    for key in (col(1), None):
        expa = orc.aggregate([cols], ALL_AGGS(2), group_nodes=key.flatten(f) if key else None, pred_nodes=pred.flatten(f))[0]
        gota = ctx.aggregate(t, ALL_AGGS(2), group_nodes=key.flatten(f) if key else None, pred_nodes=pred.flatten(f))
        assert_rows_multiset_equal(gota.to_host(), expa, RTOL, exact_cols=[0], what="pred on the value column")


def test_projection_matches_oracle(ctx):
    rng = np.random.default_rng(9)
    cols = random_batch(rng, 5000, 0.1, with_bool=True)
    t = ctx.table_from_host(cols)
    exprs = [binop(col(0), Operator.Plus, lit_i64(1)), col(2), binop(binop(col(2), Operator.Multiply, col(2)), Operator.Minus, lit_f64(1.0)),
             binop(col(4), Operator.Or, binop(col(1), Operator.Eq, lit_i64(0))), lit_i64(7)]
    exp = orc.projection([cols], [flat(e) for e in exprs])[0]
    assert_batches_equal(ctx.projection(t, [flat(e) for e in exprs]).to_host(), exp)


# --------------------------------------------------------------------------- aggregate
ALL_AGGS = lambda c: [(AggregateFunc.Count, c), (AggregateFunc.Sum, c), (AggregateFunc.Avg, c), (AggregateFunc.Min, c), (AggregateFunc.Max, c)]


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
@pytest.mark.parametrize("n", [1, 100, 5000, 70000])
def test_group_by_matches_oracle(ctx, n, null_frac):
    rng = np.random.default_rng(31 + n)
    cols = random_batch(rng, n, null_frac, with_bool=True)
    t = ctx.table_from_host(cols)
    keys = [binop(col(0), Operator.Modulos, lit_i64(16)), col(1), binop(col(1), Operator.Multiply, lit_i64(1 << 40)),
            binop(col(3), Operator.Modulos, lit_u64(37)), binop(binop(col(0), Operator.Plus, col(1)), Operator.Modulos, lit_i64(5))]
    aggsets = [ALL_AGGS(2), ALL_AGGS(0) + ALL_AGGS(3), [(AggregateFunc.Count, 4), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 2), (AggregateFunc.Min, 3), (AggregateFunc.Avg, 0)]]
    for key in keys:
        for aggs in aggsets:
            exp = orc.aggregate([cols], aggs, group_nodes=flat(key))[0]
            got, gk = ctx.aggregate(t, aggs, group_nodes=flat(key), with_keys=True)
            counts = [i for i, (f, _) in enumerate(aggs) if f == AggregateFunc.Count]
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=counts, what=f"key {key!r} aggs {aggs}")
            kk = gk.to_host()[0].to_numpy()
            assert (np.sort(kk) == kk).all() and len(np.unique(kk)) == len(kk)


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
def test_filter_then_group_by_matches_oracle(ctx, null_frac):
    rng = np.random.default_rng(123)
    n = 50000
    cols = random_batch(rng, n, null_frac)
    t = ctx.table_from_host(cols)
    for p in PREDS[:6]:
        for key in (binop(col(0), Operator.Modulos, lit_i64(1024)), col(1)):
            exp = orc.aggregate([cols], ALL_AGGS(2), group_nodes=flat(key), pred_nodes=flat(p))[0]
            got = ctx.aggregate(t, ALL_AGGS(2), group_nodes=flat(key), pred_nodes=flat(p)).to_host()
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"{p!r} / {key!r}")
    # general key under a filter that protects its divisor
    p = binop(col(0), Operator.Gt, lit_i64(0))
    key = binop(binop(lit_i64(1000000), Operator.Divide, col(0)), Operator.Modulos, lit_i64(7))
    if null_frac == 0.0:
        exp = orc.aggregate([cols], ALL_AGGS(2), group_nodes=flat(key), pred_nodes=flat(p))[0]
        got = ctx.aggregate(t, ALL_AGGS(2), group_nodes=flat(key), pred_nodes=flat(p)).to_host()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0])


def test_ungrouped_aggregate_matches_oracle(ctx):
    rng = np.random.default_rng(8)
    for n, nf in [(0, 0.0), (1, 0.0), (1000, 0.3), (300000, 0.0), (300000, 0.1)]:
        cols = random_batch(rng, n, nf, with_bool=True)
        t = ctx.table_from_host(cols)
        aggs = ALL_AGGS(2) + ALL_AGGS(1) + [(AggregateFunc.Count, 4), (AggregateFunc.Sum, 3)]
        for p in (None, PREDS[0], PREDS[4]):
            pn = flat(p) if p is not None else None
            exp = orc.aggregate([cols], aggs, pred_nodes=pn)[0] if not (n == 0 and p is not None) else None
            if exp is None:
                continue  # the reference panics on input[0] of an empty Vec; nothing to compare
            got = ctx.aggregate(t, aggs, pred_nodes=pn).to_host()
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0, 5, 10], what=f"n={n} p={p!r}")


def test_aggregate_nan_null_and_sentinel_key(ctx):
    nan = float("nan")
    imin = np.iinfo(np.int64).min
    k = Column.from_list([0, 0, 0, 1, 1, None, 2, imin, imin, -0], DType.INT64)
    v = Column.from_list([1.0, nan, 3.0, None, None, 5.0, -0.5, 2.0, 4.0, -7.0], DType.FLOAT64)
    cols = [k, v]
    f = fields("k", "v")
    aggs = ALL_AGGS(1)
    exp = orc.aggregate([cols], aggs, group_nodes=col(0).flatten(f))[0]
    got, gk = ctx.aggregate(ctx.table_from_host(cols), aggs, group_nodes=col(0).flatten(f), with_keys=True)
    assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0])
    assert gk.to_host()[0].to_list() == [imin, 0, 1, 2]
    rows = {kk: r for kk, r in zip(gk.to_host()[0].to_list(), zip(*[c.to_list() for c in got.to_host()]))}
    fmax = np.finfo(np.float64).max
    assert rows[1][0] == 0 and rows[1][1] == 0.0 and np.isnan(rows[1][2]) and rows[1][3] == fmax and rows[1][4] == -fmax
    assert np.isnan(rows[0][1]) and rows[0][3] == -7.0 and np.isnan(rows[0][4])
    assert rows[imin] == (2, 6.0, 3.0, 2.0, 4.0)


def test_aggregate_many_groups_overflows_workgroup_table(ctx):
    # more distinct keys than a workgroup's LDS table holds → rows spill to the global table
    rng = np.random.default_rng(4)
    n = 200000
    k = rng.integers(0, 60000, n).astype(np.int64) * 1024  # common low bits
    v = rng.random(n)
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f = fields("k", "v")
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f))[0]
    got = ctx.aggregate(ctx.table_from_host(cols), ALL_AGGS(1), group_nodes=col(0).flatten(f)).to_host()
    assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0])


@pytest.mark.parametrize("groups", [1, 63, 64, 65, 1024, 4095, 4097, 8000, 8192, 8193, 9000])
@pytest.mark.parametrize("unsigned", [False, True])
def test_aggregate_small_table_tail_boundaries(ctx, groups, unsigned):
    """group counts around the entry/slot boundaries of the single-launch tail (rank_finalize_kernel: 64 entries per workgroup,
    8192-slot first-attempt table) and past it (TABLE_FULL → retry with the worst-case table): rows in key order (signed /
    unsigned), keys_out aligned with the rows, the table's EMPTY sentinel (i64::MIN) as a real key"""
    rng = np.random.default_rng(groups * 2 + int(unsigned))
    n = max(3 * groups, 50)
    if unsigned:
        base = rng.permutation(groups).astype(np.uint64) * np.uint64(0x0004000000000001)   # keys on both sides of 2^63
        base[0] = np.uint64(1) << np.uint64(63)
    else:
        base = (rng.permutation(groups).astype(np.int64) - groups // 2) * 0x0000100000000003
        base[0] = np.iinfo(np.int64).min
    k = base[rng.integers(0, groups, n)]
    k[:groups] = base                                                         # every group present
    v = k.astype(np.float64)
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    aggs = [(AggregateFunc.Min, 1), (AggregateFunc.Count, 0), (AggregateFunc.Max, 1)]
    exp = orc.aggregate([cols], aggs, group_nodes=col(0).flatten(f2))[0]
    got, gk = ctx.aggregate(ctx.table_from_host(cols), aggs, group_nodes=col(0).flatten(f2), with_keys=True)
    kk = gk.to_host()[0].to_numpy()
    assert got.num_rows == groups and (kk == np.unique(k)).all()
    g = got.to_host()
    assert (g[0].to_numpy() == kk.astype(np.float64)).all() and (g[2].to_numpy() == kk.astype(np.float64)).all()
    cnt = {int(a): int(b) for a, b in zip(*np.unique(k, return_counts=True))}
    assert [cnt[int(x)] for x in kk] == g[1].to_list()
    assert_rows_multiset_equal(g, exp, RTOL, exact_cols=[1])


@pytest.mark.parametrize("m", [700, 150_001])
def test_aggregate_two_step_integer_keys(ctx, m):
    """chains of one or two integer operations with literals as group keys (`(id + 1) % m`, `id / 7 * 3`, `(100 - id) % m`, …): the
    fast kernels' interpreted key variant, on the single-pass path (m = 700) and the hash-partitioned one (m = 150001: more
    groups than a workgroup table, >= 2^18 rows); signed and unsigned sources, negative dividends, a predicate on another column"""
    rng = np.random.default_rng(m)
    n = 400_000
    a = rng.integers(-3 * m, 3 * m, n).astype(np.int64)
    u = rng.integers(0, 6 * m, n).astype(np.uint64)
    v = rng.random(n) * 10 - 5
    w = rng.integers(-50, 50, n).astype(np.int64)
    cols = [Column.from_numpy(a), Column.from_numpy(u), Column.from_numpy(v), Column.from_numpy(w)]
    f4 = fields("a", "u", "v", "w")
    t = ctx.table_from_host(cols)
    A, U = col(0), col(1)
    keys = [binop(binop(A, Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(m)),
            binop(binop(A, Operator.Divide, lit_i64(7)), Operator.Multiply, lit_i64(3)),
            binop(binop(lit_i64(100), Operator.Minus, A), Operator.Modulos, lit_i64(m)),
            binop(binop(A, Operator.Multiply, lit_i64(3)), Operator.Plus, lit_i64(1)),
            binop(binop(A, Operator.Modulos, lit_i64(-m)), Operator.Divide, lit_i64(2)),
            binop(A, Operator.Minus, lit_i64(-5)),
            binop(binop(U, Operator.Plus, lit_u64(3)), Operator.Modulos, lit_u64(m)),
            binop(binop(U, Operator.Divide, lit_u64(5)), Operator.Minus, lit_u64(7)),       # wraps below zero: huge UInt64 keys
            # three- and four-step chains (SIMPLE_MAX_OPS = 4)
            binop(binop(binop(A, Operator.Plus, lit_i64(1)), Operator.Multiply, lit_i64(3)), Operator.Modulos, lit_i64(m)),
            binop(binop(binop(binop(A, Operator.Divide, lit_i64(3)), Operator.Plus, lit_i64(7)), Operator.Multiply, lit_i64(-2)), Operator.Modulos, lit_i64(m)),
            binop(lit_u64(5), Operator.Plus, binop(binop(binop(U, Operator.Modulos, lit_u64(m)), Operator.Multiply, lit_u64(2)), Operator.Divide, lit_u64(3)))]
    aggs = ALL_AGGS(2)
    for key in keys:
        for pred in (None, binop(col(3), Operator.GtEq, lit_i64(-20))):
            pn = pred.flatten(f4) if pred is not None else None
            exp = orc.aggregate([cols], aggs, group_nodes=key.flatten(f4), pred_nodes=pn)[0]
            got, gk = ctx.aggregate(t, aggs, group_nodes=key.flatten(f4), pred_nodes=pn, with_keys=True)
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"key {key!r} pred {pred!r}")
            kk = gk.to_host()[0].to_numpy()
            assert (np.sort(kk) == kk).all() and len(np.unique(kk)) == len(kk) == got.num_rows


@pytest.mark.parametrize("groups", [900, 120_000])
def test_aggregate_chain_predicates(ctx, groups):
    """integer chains ending in a comparison as predicates (`w % 7 >= 2`, `3 * k > -1000`, `100 - u / 3 != 67`, …): the fast kernels'
    interpreted predicate variant on the single-pass path (900 groups) and on the hash-partitioned path (120000 groups, >= 2^18
    rows), over the key column and over other columns, with plain and interpreted keys"""
    rng = np.random.default_rng(groups)
    n = 420_000
    k = rng.integers(-groups // 2, groups // 2, n).astype(np.int64)
    u = rng.integers(0, 1000, n).astype(np.uint64)
    v = rng.random(n) * 10 - 5
    w = rng.integers(-50, 50, n).astype(np.int64)
    cols = [Column.from_numpy(k), Column.from_numpy(u), Column.from_numpy(v), Column.from_numpy(w)]
    f4 = fields("k", "u", "v", "w")
    t = ctx.table_from_host(cols)
    K, U, W = col(0), col(1), col(3)
    preds = [binop(binop(W, Operator.Modulos, lit_i64(7)), Operator.GtEq, lit_i64(2)),
             binop(binop(lit_i64(3), Operator.Multiply, K), Operator.Gt, lit_i64(-1000)),
             binop(binop(U, Operator.Divide, lit_u64(3)), Operator.NotEq, lit_u64(67)),
             binop(lit_i64(5), Operator.LtEq, binop(K, Operator.Modulos, lit_i64(-16))),
             binop(binop(W, Operator.Plus, lit_i64(50)), Operator.Eq, lit_i64(50)),
             # three- and four-step chains: the probe's `(id + 1) % 10 < 5` shape and a longer one
             binop(binop(binop(K, Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5)),
             binop(binop(binop(binop(W, Operator.Multiply, lit_i64(3)), Operator.Minus, lit_i64(7)), Operator.Divide, lit_i64(4)), Operator.GtEq, lit_i64(-3))]
    # Float64 chains over the value column (with NaN / +-inf / -0.0 among the values): IEEE arithmetic, ordered compares,
    # division by a non-zero literal; literal on either side
    V = col(2)
    v[:7] = [np.nan, np.inf, -np.inf, -0.0, 0.0, 5e-324, -2.5]
    cols[2] = Column.from_numpy(v)
    t = ctx.table_from_host(cols)
    preds += [binop(binop(V, Operator.Multiply, lit_f64(2.0)), Operator.Gt, lit_f64(1.5)),
              binop(binop(binop(V, Operator.Plus, lit_f64(5.0)), Operator.Divide, lit_f64(3.0)), Operator.LtEq, lit_f64(2.0)),
              binop(lit_f64(0.0), Operator.Lt, binop(lit_f64(1.0), Operator.Minus, V)),
              binop(binop(binop(binop(V, Operator.Multiply, lit_f64(-1.0)), Operator.Minus, lit_f64(0.25)), Operator.Divide, lit_f64(-0.5)), Operator.NotEq, lit_f64(0.5))]
    keys = [K, binop(binop(K, Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(groups + 1))]
    aggs = ALL_AGGS(2)
    for pred in preds:
        for key in keys:
            exp = orc.aggregate([cols], aggs, group_nodes=key.flatten(f4), pred_nodes=pred.flatten(f4))[0]
            got, gk = ctx.aggregate(t, aggs, group_nodes=key.flatten(f4), pred_nodes=pred.flatten(f4), with_keys=True)
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"pred {pred!r} key {key!r}")
            kk = gk.to_host()[0].to_numpy()
            assert (np.sort(kk) == kk).all() and len(np.unique(kk)) == len(kk) == got.num_rows


def test_aggregate_partial_merge_equals_single_pass(ctx):
    rng = np.random.default_rng(99)
    n = 40000
    cols = random_batch(rng, n, 0.1)
    aggs = ALL_AGGS(2) + [(AggregateFunc.Count, 0)]
    key = flat(binop(col(0), Operator.Modulos, lit_i64(100)))
    pred = flat(PREDS[0])
    full = ctx.aggregate(ctx.table_from_host(cols), aggs, group_nodes=key, pred_nodes=pred)
    parts = []
    for lo, hi in [(0, 13000), (13000, 13001), (13001, 40000)]:
        sub = [Column.from_numpy(c.to_numpy()[lo:hi], c.valid_mask()[lo:hi]) for c in cols]
        parts.append(ctx.aggregate_partial(ctx.table_from_host(sub), aggs, group_nodes=key, pred_nodes=pred))
    merged, mk = ctx.aggregate_merge([p[0] for p in parts], [p[1] for p in parts], aggs)
    assert_rows_multiset_equal(merged.to_host(), full.to_host(), RTOL, exact_cols=[0, 5])
    # un-grouped partials
    full_u = ctx.aggregate(ctx.table_from_host(cols), aggs)
    parts_u = [ctx.aggregate_partial(ctx.table_from_host([Column.from_numpy(c.to_numpy()[lo:hi], c.valid_mask()[lo:hi]) for c in cols]), aggs)[0]
               for lo, hi in [(0, 20000), (20000, 40000)]]
    merged_u, _ = ctx.aggregate_merge(parts_u, None, aggs)
    assert_rows_multiset_equal(merged_u.to_host(), full_u.to_host(), RTOL, exact_cols=[0, 5])


def test_aggregate_errors(ctx):
    cols = [Column.from_list([1.5, 2.5], DType.FLOAT64), Column.from_list([True, False], DType.BOOLEAN)]
    t = ctx.table_from_host(cols)
    f = fields("x", "b")
    with pytest.raises(ErrorCode) as e:  # Float64 group key (aggregate/mod.rs:217)
        ctx.aggregate(t, [(AggregateFunc.Sum, 0)], group_nodes=col(0).flatten(f))
    assert e.value.status == Status.NotSupported
    with pytest.raises(ErrorCode) as o:
        orc.aggregate([cols], [(AggregateFunc.Sum, 0)], group_nodes=col(0).flatten(f))
    assert o.value.status == Status.NotSupported
    with pytest.raises(ErrorCode) as e:  # sum over Boolean
        ctx.aggregate(t, [(AggregateFunc.Sum, 1)])
    assert e.value.status == Status.NotSupported


# --------------------------------------------------------------------------- hash join
def join_inputs(rng, nb, npr, key_space, null_frac=0.0, unique=False):
    if unique:
        # never materialise range(key_space): draw 2*nb candidates, de-duplicate, shuffle
        cand = np.unique(rng.integers(0, key_space, 2 * nb + 16)) if key_space > 4 * nb else np.arange(key_space)
        lk = rng.permutation(cand)[:nb].astype(np.int64)
        assert lk.size == nb
    else:
        lk = rng.integers(0, key_space, nb).astype(np.int64)
    left = [Column.from_numpy(lk), Column.from_numpy(rng.integers(-1000, 1000, nb).astype(np.int64), None if null_frac == 0 else rng.random(nb) > null_frac),
            Column.from_numpy(rng.random(nb) < 0.5)]
    rkey = rng.integers(-3, key_space + 3, npr).astype(np.int64)
    right = [Column.from_numpy(rng.random(npr), None if null_frac == 0 else rng.random(npr) > null_frac), Column.from_numpy(rkey)]
    return left, right


@pytest.mark.parametrize("unique", [True, False])
@pytest.mark.parametrize("nb,npr,space", [(0, 10, 5), (5, 0, 5), (1, 1, 1), (50, 1000, 40), (5000, 20000, 5000), (6000, 9000, 100), (20000, 30000, 1 << 40)])
def test_hash_join_matches_oracle(ctx, nb, npr, space, unique):
    if unique and space < nb:
        pytest.skip("cannot draw unique keys")
    rng = np.random.default_rng(nb * 3 + npr)
    left, right = join_inputs(rng, nb, npr, space, null_frac=0.1, unique=unique)
    exp = orc.hash_join([left], [right], 0, 1)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 1).to_host()
    assert_batches_equal(got, exp, what=f"join nb={nb} np={npr}")   # same rows in the SAME order


@pytest.mark.parametrize("nb,npr,holes", [(1, 5, False), (4096, 4096, False), (10000, 50000, True), (30000, 70001, False)])
def test_hash_join_pk_fk_fused_path(ctx, nb, npr, holes):
    """unique + dense build keys with plain payload on both sides → presence bitmap + fused write path"""
    rng = np.random.default_rng(nb + npr)
    space = nb * 2 if holes else nb            # holes: only half of the key range is present
    lk = (rng.permutation(space)[:nb] - 7).astype(np.int64)   # negative keys too
    # payloads: small signed range (packed as 32-bit offsets), Float64 and wide UInt64 (not packable), Int64 around i64::MIN
    # (packable, wrapping base), UInt64 with a range of exactly 2^32 - 1 and 2^32 (boundary of the packing rule)
    left = [Column.from_numpy(rng.integers(-9, 9, nb).astype(np.int64)), Column.from_numpy(lk), Column.from_numpy(rng.random(nb)),
            Column.from_numpy(rng.integers(0, 1 << 60, nb).astype(np.uint64)),
            Column.from_numpy((np.iinfo(np.int64).min + rng.integers(0, 1 << 31, nb)).astype(np.int64)),
            Column.from_numpy((np.uint64(1 << 63) + np.concatenate([[0, (1 << 32) - 1], rng.integers(0, 1 << 32, nb)])[:nb].astype(np.uint64))),
            Column.from_numpy((np.uint64(5) + np.concatenate([[0, 1 << 32], rng.integers(0, 1 << 32, nb)])[:nb].astype(np.uint64)))]
    rk = (rng.integers(-3, space + 3, npr) - 7).astype(np.int64)
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(npr)), Column.from_numpy(np.arange(npr, dtype=np.int64))]
    exp = orc.hash_join([left], [right], 1, 0)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 1, 0).to_host()
    assert_batches_equal(got, exp, what="pk-fk fused join")
    # same build table reused for a second probe batch
    jt = ctx.hash_join_build(ctx.table_from_host(left), 1)
    right2 = [Column.from_numpy(rk[::-1].copy()), Column.from_numpy(rng.random(npr)), Column.from_numpy(np.arange(npr, dtype=np.int64))]
    assert_batches_equal(ctx.hash_join_probe(jt, ctx.table_from_host(right2), 0).to_host(), orc.hash_join([left], [right2], 1, 0)[0])


def test_hash_join_heavy_duplicates_and_build_probe_reuse(ctx):
    rng = np.random.default_rng(2)
    lk = np.repeat(np.array([7, -1, 7, 3], dtype=np.int64), 1500)       # 6000 build rows, 3 distinct keys
    left = [Column.from_numpy(lk), Column.from_numpy(np.arange(lk.size, dtype=np.int64))]
    jt = ctx.hash_join_build(ctx.table_from_host(left), 0)
    for seed in (1, 2):
        rk = np.random.default_rng(seed).integers(-2, 9, 300).astype(np.int64)
        right = [Column.from_numpy(rk)]
        exp = orc.hash_join([left], [right], 0, 0)[0]
        got = ctx.hash_join_probe(jt, ctx.table_from_host(right), 0).to_host()
        assert_batches_equal(got, exp)


def test_hash_join_ignores_key_validity_q11_and_uint64(ctx):
    lk = Column.from_list([7, 3, 7, None, 7], DType.INT64)
    left = [lk, Column.from_list([10, 11, 12, 13, 14], DType.INT64)]
    right = [Column.from_list([3, 7, 0, 9], DType.INT64), Column.from_list([0.5, 1.5, 2.5, 3.5], DType.FLOAT64)]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    assert_batches_equal(ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0).to_host(), exp)
    ul = [Column.from_numpy(np.array([1 << 63, 5, (1 << 64) - 1], dtype=np.uint64))]
    ur = [Column.from_numpy(np.array([(1 << 64) - 1, 1 << 63, 6], dtype=np.uint64))]
    assert_batches_equal(ctx.hash_join(ctx.table_from_host(ul), ctx.table_from_host(ur), 0, 0).to_host(), orc.hash_join([ul], [ur], 0, 0)[0])


def test_hash_join_errors(ctx):
    a = ctx.table_from_host([Column.from_list([1.0], DType.FLOAT64)])
    i = ctx.table_from_host([Column.from_list([1], DType.INT64)])
    u = ctx.table_from_host([Column.from_list([1], DType.UINT64)])
    for l, r, lk, rk, st in [(a, a, 0, 0, Status.NotImplemented), (i, a, 0, 0, Status.NotImplemented), (i, u, 0, 0, Status.NotSupported),
                             (i, i, -1, -1, Status.PlanError)]:
        with pytest.raises(ErrorCode) as e:
            ctx.hash_join(l, r, lk, rk)
        assert e.value.status == st


# --------------------------------------------------------------------------- table utilities
def test_all_valid_bitmaps_are_dropped_at_table_creation(ctx):
    """a validity bitmap without a single null is dropped (counted on the device when null_count is unknown), so that
    schema-nullable data still takes the kernels specialised for non-null columns; results are unchanged"""
    rng = np.random.default_rng(12)
    for n in (1, 63, 64, 1000, 70001):
        allv = np.ones(n, dtype=bool)
        last_null = allv.copy(); last_null[-1] = False
        first_null = allv.copy(); first_null[0] = False
        ids = rng.integers(0, 50, n).astype(np.int64)
        v = rng.random(n)
        cols = [Column.from_numpy(ids, allv), Column.from_numpy(v, last_null), Column.from_numpy(v, first_null), Column.from_numpy(v, allv)]
        t = ctx.table_from_host(cols)
        infos = [t.column_info(i) for i in range(4)]
        assert not infos[0].validity and not infos[3].validity, n
        assert infos[1].validity and infos[2].validity, n
        assert [int(i.null_count) for i in infos] == [0, 1, 1, 0]
        got = ctx.aggregate(t, ALL_AGGS(1) + ALL_AGGS(3), group_nodes=col(0).flatten(fields("k", "a", "b", "c")))
        exp = orc.aggregate([cols], ALL_AGGS(1) + ALL_AGGS(3), group_nodes=col(0).flatten(fields("k", "a", "b", "c")))[0]
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0, 5], what=f"n={n}")


def test_memory_stats_and_trim(ctx):
    ctx.synchronize()
    live0, _ = ctx.memory_stats()
    t = ctx.table_from_host([Column.from_numpy(np.arange(1 << 20, dtype=np.int64))])
    live1, _ = ctx.memory_stats()
    assert live1 - live0 >= 8 << 20
    del t
    live2, pooled2 = ctx.memory_stats()
    assert live2 <= live0 and pooled2 >= 8 << 20          # the block went back to the pool
    ctx.trim()
    assert ctx.memory_stats()[1] == 0


def test_take_slice_concat_project(ctx):
    rng = np.random.default_rng(3)
    cols = random_batch(rng, 1000, 0.2, with_bool=True)
    t = ctx.table_from_host(cols)
    idx = rng.integers(0, 1000, 777).astype(np.int64)
    got = ctx.take(t, ctx.table_from_host([Column.from_numpy(idx)])).to_host()
    for g, c in zip(got, cols):
        exp = Column.from_numpy(c.to_numpy()[idx], c.valid_mask()[idx])
        assert_column_equal(g, exp)
    with pytest.raises(ErrorCode):
        ctx.take(t, ctx.table_from_host([Column.from_numpy(np.array([1000], dtype=np.int64))]))
    for off, ln in [(0, 1000), (3, 100), (64, 65), (999, 1), (10, 0)]:
        got = ctx.slice(t, off, ln).to_host()
        assert_batches_equal(got, orc.limit(orc.offset([cols], off), ln)[0] if ln else [Column.from_numpy(c.to_numpy()[0:0]) for c in cols])
    parts = [random_batch(rng, m, nf, with_bool=True) for m, nf in [(5, 0.0), (130, 0.3), (0, 0.0), (64, 0.0), (7, 0.5)]]
    got = ctx.concat([ctx.table_from_host(p) for p in parts]).to_host()
    for ci, g in enumerate(got):
        vals = np.concatenate([p[ci].to_numpy() for p in parts])
        mask = np.concatenate([p[ci].valid_mask() for p in parts])
        assert_column_equal(g, Column.from_numpy(vals, mask))
    pr = ctx.project(t, [2, 1]).to_host()
    assert_batches_equal(pr, [cols[2], cols[1]])


def test_synth_fill_matches_oracle_generator(ctx):
    import ctypes

    from naive_query_engine_amd import capi

    n = 100003
    for kind, seed, mod, base in [(0, 0, 1, 0), (1, 2, 60, 18), (1, 5, 1000000, 0), (2, 3, 1, 0)]:
        ptr = ctx.device_alloc(n * 8)
        ctx.synth_fill(kind, seed, 17, n, mod, base, ptr)
        t = ctx.table_from_device([(DType.UINT64, n, ptr, None)])
        got = t.download_column(0).to_numpy()
        assert (got == orc.synth_fill(kind, seed, 17, n, mod, base)).all()
        del t
        ctx.device_free(ptr)


# --------------------------------------------------------------------------- golden fixtures on the GPU
def test_golden_fixture_queries_on_gpu(ctx, csv_tables, golden):
    t1 = csv_tables["test_data"]
    num = [t1.columns[0], t1.columns[2], t1.columns[3]]  # id, age, score (Utf8 `name` stays on the host for now)
    f = fields("id", "age", "score")
    t = ctx.table_from_host(num)
    # test_selection: (id + 1) > 5
    got = ctx.selection(t, binop(binop(col("id"), Operator.Plus, lit_i64(1)), Operator.Gt, lit_i64(5)).flatten(f)).to_host()
    assert got[0].to_list() == golden["test_selection"]["id"]
    # test_projection: id + 1
    got = ctx.projection(t, [binop(col("id"), Operator.Plus, lit_i64(1)).flatten(f)]).to_host()
    assert got[0].to_list() == golden["test_projection"]["id_plus_1"]
    # select id, age from t1 where id > 1
    got = ctx.selection_projection(t, binop(col(0), Operator.Gt, lit_i64(1)).flatten(f), [col(0).flatten(f), col(1).flatten(f)]).to_host()
    assert got[0].to_list() == golden["sql_where_id_gt_1"]["id"] and got[1].to_list() == golden["sql_where_id_gt_1"]["age"]
    # README aggregate: exact digits are sequential-order artefacts → 1e-9 relative, counts exact
    aggs = [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 1), (AggregateFunc.Sum, 2), (AggregateFunc.Avg, 2), (AggregateFunc.Max, 2), (AggregateFunc.Min, 2)]
    got = ctx.aggregate(t, aggs, group_nodes=binop(col(0), Operator.Modulos, lit_i64(3)).flatten(f)).to_host()
    rows = np.array(sorted(map(list, zip(*[c.to_list() for c in got]))), dtype=np.float64)
    exp = np.array(sorted(golden["readme_group_by_id_mod_3"]["rows"]), dtype=np.float64)
    assert (rows[:, 0] == exp[:, 0]).all() and np.allclose(rows, exp, rtol=RTOL, atol=0)
    # README joins on the numeric columns: employee(id, department_id, rank) ⋈ rank(id) ⋈ department(id)
    emp, rank, dep = csv_tables["employee"], csv_tables["rank"], csv_tables["department"]
    e = ctx.table_from_host([emp.columns[0], emp.columns[2], emp.columns[3]])
    r = ctx.table_from_host([rank.columns[0]])
    d = ctx.table_from_host([dep.columns[0]])
    j2 = ctx.hash_join(ctx.hash_join(e, r, 2, 0), d, 1, 0).to_host()
    assert j2[0].to_list() == [row[0] for row in golden["readme_two_hash_joins"]["rows"]]  # employee ids in README order


# --------------------------------------------------------------------------- Utf8 payload columns
@pytest.mark.parametrize("null_frac", [0.0, 0.2])
@pytest.mark.parametrize("n", [0, 1, 65, 5000])
def test_utf8_columns_through_filter_take_slice_concat(ctx, n, null_frac):
    rng = np.random.default_rng(500 + n)
    cols = random_batch(rng, n, null_frac) + [random_utf8(rng, n, null_frac)]
    t = ctx.table_from_host(cols)
    for p in PREDS[:5]:
        assert_batches_equal(ctx.selection(t, flat(p)).to_host(), orc.selection([cols], flat(p))[0], what=f"utf8 selection {p!r}")
    # projection of a bare Utf8 column, plain and fused with a filter
    exprs = [col(4), binop(col(1), Operator.Plus, lit_i64(1))]
    f5 = fields("id", "k", "v", "u", "name")
    assert_batches_equal(ctx.projection(t, [e.flatten(f5) for e in exprs]).to_host(), orc.projection([cols], [e.flatten(f5) for e in exprs])[0])
    sel = orc.selection([cols], flat(PREDS[0]))
    assert_batches_equal(ctx.selection_projection(t, flat(PREDS[0]), [e.flatten(f5) for e in exprs]).to_host(),
                         orc.projection(sel, [e.flatten(f5) for e in exprs])[0])
    if n:
        idx = rng.integers(0, n, 33).astype(np.int64)
        got = ctx.take(t, ctx.table_from_host([Column.from_numpy(idx)])).to_host()[4]
        src = cols[4].to_list()
        assert got.to_list() == [src[i] for i in idx]
        for off, ln in [(0, n), (n // 3, n // 2), (n - 1, 1)]:
            assert ctx.slice(t, off, ln).to_host()[4].to_list() == src[off:off + ln]
    parts = [cols, random_batch(rng, 7, null_frac) + [random_utf8(rng, 7, null_frac)], random_batch(rng, 0) + [random_utf8(rng, 0)]]
    got = ctx.concat([ctx.table_from_host(p) for p in parts]).to_host()[4]
    assert got.to_list() == sum([p[4].to_list() for p in parts], [])
    # count(name) counts the non-null strings
    got = ctx.aggregate(t, [(AggregateFunc.Count, 4)]).to_host()[0].to_list()
    assert got == [sum(x is not None for x in cols[4].to_list())]


@pytest.mark.parametrize("unique", [True, False])
def test_hash_join_with_utf8_payload(ctx, unique):
    rng = np.random.default_rng(77 + unique)
    nb, npr = 3000, 8000
    lk = rng.permutation(nb).astype(np.int64) if unique else rng.integers(0, 400, nb).astype(np.int64)
    left = [random_utf8(rng, nb, 0.1), Column.from_numpy(lk), Column.from_numpy(rng.random(nb))]
    right = [Column.from_numpy(rng.integers(-5, nb + 5 if unique else 405, npr).astype(np.int64)), random_utf8(rng, npr, 0.1)]
    exp = orc.hash_join([left], [right], 1, 0)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 1, 0).to_host()
    assert_batches_equal(got, exp, what="utf8 join")


def test_readme_queries_end_to_end_with_names(ctx, csv_tables, golden):
    t1 = csv_tables["test_data"]
    f = fields("id", "name", "age", "score")
    t = ctx.table_from_host(t1.columns)
    # test_selection (selection.rs:166-172): ids AND names
    got = ctx.selection(t, binop(binop(col("id"), Operator.Plus, lit_i64(1)), Operator.Gt, lit_i64(5)).flatten(f)).to_host()
    assert got[0].to_list() == golden["test_selection"]["id"] and got[1].to_list() == golden["test_selection"]["name"]
    # README query 1: select id, name, age + 100 from t1 where id < 9 limit 3 offset 2
    proj = ctx.selection_projection(t, binop(col(0), Operator.Lt, lit_i64(9)).flatten(f),
                                    [col(0).flatten(f), col(1).flatten(f), binop(col(2), Operator.Plus, lit_i64(100)).flatten(f)])
    out = ctx.slice(proj, 2, 3).to_host()
    assert list(map(list, zip(*[c.to_list() for c in out]))) == golden["readme_filter_project_offset_limit"]["rows"]
    # README query 2: employee ⋈ rank ⋈ department with the Utf8 payloads, row order pinned
    emp, rank, dep = (ctx.table_from_host(csv_tables[k].columns) for k in ("employee", "rank", "department"))
    j2 = ctx.hash_join(ctx.hash_join(emp, rank, 3, 0), dep, 2, 0).to_host()
    rows = list(map(list, zip(j2[0].to_list(), j2[1].to_list(), j2[5].to_list(), j2[7].to_list())))
    assert rows == golden["readme_two_hash_joins"]["rows"]


@pytest.mark.parametrize("null_frac", [0.0, 0.2])
def test_utf8_comparisons_against_literals_and_columns(ctx, null_frac):
    """eq/neq/lt/lt_eq/gt/gt_eq on StringArrays (binary.rs:127-132 → arrow *_dyn: byte-wise lexicographic), either side a
    ScalarValue::Utf8 literal (also None and ""), as expression, predicate, inside and/or trees, and under an aggregate"""
    rng = np.random.default_rng(41 + int(null_frac * 10))
    n = 5000
    a, b = random_utf8(rng, n, null_frac), random_utf8(rng, n, null_frac)
    ids = Column.from_numpy(rng.integers(0, 40, n).astype(np.int64))
    v = Column.from_numpy(rng.random(n))
    cols = [a, b, ids, v]
    f = fields("a", "b", "id", "v")
    t = ctx.table_from_host(cols)
    lits = [lit_utf8("bob"), lit_utf8(""), lit_utf8("lynne5"), lit_utf8("véé"), lit_utf8("zzz"), lit_utf8(None), lit_utf8("x" * 70 + "1")]
    for op in CMP:
        exprs = [binop(col(0), op, col(1)), binop(col(1), op, col(1))]
        exprs += [binop(col(0), op, l) for l in lits] + [binop(l, op, col(1)) for l in lits[:3]] + [binop(lits[0], op, lits[2])]
        for e in exprs:
            assert_column_equal(ctx.expr_evaluate(t, e.flatten(f)).to_host()[0], orc.expr_evaluate([cols], e.flatten(f)), what=repr(e))
    pred = binop(binop(col(0), Operator.GtEq, lit_utf8("bob")), Operator.And, binop(binop(col(2), Operator.Lt, lit_i64(30)), Operator.Or, binop(col(1), Operator.Eq, col(0))))
    assert_batches_equal(ctx.selection(t, pred.flatten(f)).to_host(), orc.selection([cols], pred.flatten(f))[0], what="selection on a Utf8 predicate tree")
    exp = orc.aggregate([cols], ALL_AGGS(3), group_nodes=col(2).flatten(f), pred_nodes=pred.flatten(f))[0]
    got = ctx.aggregate(t, ALL_AGGS(3), group_nodes=col(2).flatten(f), pred_nodes=pred.flatten(f))
    assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what="aggregate under a Utf8 predicate")
    # a projected literal becomes an n-row StringArray (ScalarValue::into_array)
    proj = [lit_utf8("const"), col(0), lit_utf8(None)]
    assert_batches_equal(ctx.projection(t, [e.flatten(f) for e in proj]).to_host(), orc.projection([cols], [e.flatten(f) for e in proj])[0], what="literal projection")
    # type errors as in binary.rs:114-119
    for bad in (binop(col(0), Operator.Eq, lit_i64(1)), binop(col(0), Operator.Plus, col(1)), binop(col(0), Operator.And, col(1))):
        with pytest.raises(ErrorCode) as x:
            orc.expr_evaluate([cols], bad.flatten(f))
        with pytest.raises(ErrorCode) as y:
            ctx.expr_evaluate(t, bad.flatten(f))
        assert x.value.status == y.value.status, repr(bad)


def test_group_by_utf8_key(ctx):
    rng = np.random.default_rng(321)
    for n, nf in [(1, 0.0), (500, 0.0), (40000, 0.15)]:
        cols = random_batch(rng, n, nf) + [random_utf8(rng, n, nf)]
        t = ctx.table_from_host(cols)
        f5 = fields("id", "k", "v", "u", "name")
        aggs = ALL_AGGS(2) + [(AggregateFunc.Count, 4), (AggregateFunc.Sum, 0)]
        for pred in (None, PREDS[0], PREDS[4]):
            pn = pred.flatten(f5) if pred is not None else None
            exp = orc.aggregate([cols], aggs, group_nodes=col(4).flatten(f5), pred_nodes=pn)[0]
            got, gk = ctx.aggregate(t, aggs, group_nodes=col(4).flatten(f5), pred_nodes=pn, with_keys=True)
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0, 5], what=f"group by name n={n}")
            keys = gk.to_host()[0]
            assert keys.dtype == DType.UTF8 and len(set(keys.to_list())) == keys.length == got.num_rows


@pytest.mark.parametrize("distinct", [10000, 70000])
def test_group_by_utf8_key_many_strings_repeated(ctx, distinct):
    """`group by <Utf8 column>` with more distinct strings than the workgroup tables hold, on a table large enough for the plan hints
    (>= 2^18 rows), executed three times: the first execution measures the key range of the Int64 codes, later ones take the tiers
    that range selects (the range tier writes keys itself) — every one of them must hand back STRINGS, paired with their own groups"""
    rng = np.random.default_rng(77 + distinct)
    n = 300000
    code = rng.integers(0, distinct, n)
    v = rng.random(n)
    names = Column.from_list([f"name-{int(c):06d}" + ("é" if c % 5 == 0 else "") for c in code], DType.UTF8)
    t = ctx.table_from_host([names, Column.from_numpy(v)])
    f2 = fields("name", "v")
    cnt = np.bincount(code, minlength=distinct)
    sm = np.bincount(code, weights=v, minlength=distinct)
    present = np.nonzero(cnt)[0]
    exp = {f"name-{int(c):06d}" + ("é" if c % 5 == 0 else ""): (int(cnt[c]), float(sm[c])) for c in present}
    for rep in range(3):
        got, gk = ctx.aggregate(t, [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1)], group_nodes=col(0).flatten(f2), with_keys=True)
        keys = gk.to_host()[0]
        assert keys.dtype == DType.UTF8, f"execution {rep}: keys_out is {keys.dtype}, not strings"
        ks = keys.to_list()
        c_got, s_got = (x.to_numpy() for x in got.to_host())
        assert len(ks) == len(exp) == got.num_rows and len(set(ks)) == len(ks), f"execution {rep}"
        e_cnt = np.array([exp[k][0] for k in ks], dtype=np.uint64)
        e_sum = np.array([exp[k][1] for k in ks])
        assert np.array_equal(c_got, e_cnt), f"execution {rep}: counts"
        assert np.allclose(s_got, e_sum, rtol=RTOL, atol=0), f"execution {rep}: sums"


@pytest.mark.parametrize("unique", [True, False])
def test_hash_join_on_utf8_keys(ctx, unique):
    rng = np.random.default_rng(99 + unique)
    nb, npr = 2000, 6000
    if unique:
        lk = Column.from_list([f"key-{i:05d}" + ("é" if i % 7 == 0 else "") for i in rng.permutation(nb)], DType.UTF8)
    else:
        lk = random_utf8(rng, nb, 0.1)   # few distinct strings, duplicates, NULL slots (validity is ignored, quirk Q11)
    left = [lk, Column.from_numpy(rng.integers(0, 100, nb).astype(np.int64))]
    rk = Column.from_list([f"key-{i:05d}" + ("é" if i % 7 == 0 else "") for i in rng.integers(-5, nb + 5, npr)], DType.UTF8) if unique else random_utf8(rng, npr, 0.1)
    right = [Column.from_numpy(rng.random(npr)), rk]
    exp = orc.hash_join([left], [right], 0, 1)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 1).to_host()
    assert_batches_equal(got, exp, what="utf8-key join")
    with pytest.raises(ErrorCode) as e:
        ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0)  # Utf8 vs Float64 key
    assert e.value.status in (Status.NotImplemented, Status.NotSupported)


@pytest.mark.parametrize("groups", [3000, 5000, 6500, 60000, 400000])
def test_aggregate_partitioned_path_many_groups(ctx, groups):
    """more distinct keys than a workgroup table holds (and >= 2^18 rows): two key subsets (5000, 6500: each row range read by
    two workgroups that keep half of the keys each), then hash-partitioned aggregation"""
    rng = np.random.default_rng(groups)
    n = 600_000
    k = (rng.integers(0, groups, n).astype(np.int64) - groups // 2) * 7      # negative keys, common factor
    k[:3] = np.iinfo(np.int64).min                                            # the table's EMPTY sentinel as a real key
    v = rng.random(n) * 100 - 50
    v[5] = np.nan
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    f3 = fields("k", "v", "w")
    t = ctx.table_from_host(cols)
    for aggs in (ALL_AGGS(1), ALL_AGGS(1) + ALL_AGGS(2)):
        for key in (col(0), binop(col(0), Operator.Modulos, lit_i64(100003))):
            for pred in (None, binop(col(2), Operator.GtEq, lit_i64(-200))):
                pn = pred.flatten(f3) if pred is not None else None
                exp = orc.aggregate([cols], aggs, group_nodes=key.flatten(f3), pred_nodes=pn)[0]
                got = ctx.aggregate(t, aggs, group_nodes=key.flatten(f3), pred_nodes=pn).to_host()
                counts = [i for i, (fn, _) in enumerate(aggs) if fn == AggregateFunc.Count]
                assert_rows_multiset_equal(got, exp, RTOL, exact_cols=counts, what=f"partitioned groups={groups} key={key!r}")


@pytest.mark.parametrize("span_bits", [7, 16, 17, 24, 25, 40])
def test_join_payload_packing_widths(ctx, span_bits):
    """dense unique build keys with plain payloads: the key-ordered payload table holds value - min in 2, 3 or 4 bytes when the
    column's range allows (8 otherwise) — every width, at its boundary values, for Int64 (negative minimum), UInt64 and a
    Float64 payload (never packed) in one build"""
    rng = np.random.default_rng(span_bits)
    nb, n = 5000, 60000
    dk = rng.permutation(nb).astype(np.int64) + 1000
    top = (1 << span_bits) - 1
    a = rng.integers(0, top + 1, nb).astype(np.int64) - (1 << 20)       # range of exactly span_bits bits, negative minimum
    a[:2] = [-(1 << 20), top - (1 << 20)]
    b = rng.integers(0, top + 1, nb).astype(np.uint64) + np.uint64(1 << 62)
    b[:2] = [np.uint64(1 << 62), np.uint64((1 << 62) + top)]
    c = rng.random(nb)
    left = [Column.from_numpy(dk), Column.from_numpy(a), Column.from_numpy(b), Column.from_numpy(c)]
    rk = rng.integers(900, nb + 1100, n).astype(np.int64)           # some keys outside the build range
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0).to_host()
    assert_batches_equal(got, exp, what=f"payload range of {span_bits} bits")


def test_join_optimistic_all_match_form_and_its_fallback(ctx):
    """a build side whose keys fill their range (a primary key): the probe first assumes every probe row matches and writes the
    output in one pass; a probe key outside the range (first row, a middle tile, the very last row) must discard that output and
    produce the reference's rows through the two-pass form — on this and on every later probe of the same join table"""
    rng = np.random.default_rng(12)
    nb, n = 3000, 5 * 4096 + 33
    dk = rng.permutation(nb).astype(np.int64) + 50
    left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1000, nb).astype(np.int64)), Column.from_numpy(rng.random(nb))]
    lt = ctx.table_from_host(left)
    base = rng.integers(50, 50 + nb, n).astype(np.int64)
    for bad_at in ([], [0], [2 * 4096 + 17], [n - 1], [5, 9000, n - 2]):
        jt = ctx.hash_join_build(lt, 0)
        rk = base.copy()
        for j, i in enumerate(bad_at):
            rk[i] = [49, 50 + nb, -7][j % 3]                      # just below, just above, far outside
        right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
        rt = ctx.table_from_host(right)
        exp = orc.hash_join([left], [right], 0, 0)[0]
        for _ in range(2):                                      # the second probe starts from what the first one learnt
            got = ctx.hash_join_probe(jt, rt, 0).to_host()
            assert_batches_equal(got, exp, what=f"optimistic join, misses at {bad_at}")
        # the same join table, then an all-matching probe side
        right2 = [Column.from_numpy(base), Column.from_numpy(rng.random(n))]
        exp2 = orc.hash_join([left], [right2], 0, 0)[0]
        assert_batches_equal(ctx.hash_join_probe(jt, ctx.table_from_host(right2), 0).to_host(), exp2, what="all-match probe after a miss")


def test_clustered_predicates_leave_whole_tiles_empty(ctx):
    """a filter on sorted data keeps a contiguous range: the 4096-row tiles outside it hold no kept rows and the compaction / the
    join's fused write skip them before their loads — the result is the reference's, empty tiles or not (first, last and inner
    tiles empty; everything empty)"""
    n = 9 * 4096 + 17
    ids = np.arange(n, dtype=np.int64)
    rng = np.random.default_rng(4)
    v = rng.random(n) * 10
    w = rng.integers(0, 50, n).astype(np.int64)
    cols = [Column.from_numpy(ids), Column.from_numpy(v), Column.from_numpy(w), Column.from_numpy(w, rng.random(n) > 0.3)]
    f4 = fields("id", "v", "w", "wn")
    t = ctx.table_from_host(cols)
    ID = col(0)
    ranges = [(0, 5000), (3 * 4096, 5 * 4096), (n - 100, n), (20000, 20001), (n + 5, n + 9)]
    for lo, hi in ranges:
        pred = binop(binop(ID, Operator.GtEq, lit_i64(lo)), Operator.And, binop(ID, Operator.Lt, lit_i64(hi))).flatten(f4)
        exp = orc.selection([cols], pred)[0]
        assert_batches_equal(ctx.selection(t, pred).to_host(), exp, what=f"selection [{lo},{hi})")
        proj = [binop(col(2), Operator.Plus, lit_i64(100)).flatten(f4), binop(binop(col(1), Operator.Multiply, col(1)), Operator.Plus, col(1)).flatten(f4),
                col(3).flatten(f4)]
        expp = orc.projection(orc.selection([cols], pred), proj)[0]
        assert_batches_equal(ctx.selection_projection(t, pred, proj).to_host(), expp, what=f"selection + projection [{lo},{hi})")
    # join: the dim covers one key range only (dense keys) / scattered keys inside one range (hashed), probe keys ascending
    for lo, hi, stride in ((2 * 4096 + 5, 4 * 4096 + 9, 1), (5 * 4096, 7 * 4096, 97)):
        dk = np.arange(lo, hi, stride, dtype=np.int64)
        dk = dk[rng.permutation(len(dk))]
        left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1 << 20, len(dk)).astype(np.int64))]
        right = [Column.from_numpy(ids), Column.from_numpy(v)]
        exp = orc.hash_join([left], [right], 0, 0)[0]
        got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0).to_host()
        assert_batches_equal(got, exp, what=f"join, dim keys [{lo},{hi}) step {stride}")


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 70001])
def test_selection_two_test_predicates(ctx, n):
    """`A and B` / `A or B` of two compares with literals as a selection's predicate: one streaming pass over the tested
    column(s) builds the keep mask (no Boolean column); same rows, same order as the reference's filter"""
    rng = np.random.default_rng(n)
    k = rng.integers(-50, 50, n).astype(np.int64)
    v = rng.random(n) * 100 - 50
    v[: min(n, 5)] = [np.nan, np.inf, -np.inf, -0.0, 0.0][: min(n, 5)]
    u = rng.integers(0, 1 << 40, n).astype(np.uint64)
    z = rng.integers(0, 100, n).astype(np.int64)
    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(u), Column.from_numpy(z, rng.random(n) > 0.2)]
    f4 = fields("k", "v", "u", "z")
    t = ctx.table_from_host(cols)
    K, V, U, Z = col(0), col(1), col(2), col(3)
    A, O = Operator.And, Operator.Or
    preds = [binop(binop(K, Operator.GtEq, lit_i64(-10)), A, binop(K, Operator.Lt, lit_i64(25))),                # a range on one column
             binop(binop(K, Operator.Lt, lit_i64(0)), A, binop(V, Operator.Gt, lit_f64(-10.0))),
             binop(binop(lit_f64(0.0), Operator.LtEq, V), O, binop(U, Operator.Lt, lit_u64(1 << 38))),           # literal on the left, -0.0 == 0.0
             binop(binop(V, Operator.NotEq, lit_f64(float("nan"))), A, binop(V, Operator.Lt, lit_f64(float("inf")))),  # != NaN is true for every row
             binop(binop(V, Operator.Eq, lit_f64(float("nan"))), O, binop(K, Operator.Eq, lit_i64(7))),
             binop(binop(Z, Operator.Lt, lit_i64(50)), A, binop(K, Operator.Gt, lit_i64(-100)))]                  # nullable column: the general path (Kleene)
    l1, l2, l3 = binop(K, Operator.GtEq, lit_i64(-30)), binop(V, Operator.Lt, lit_f64(40.0)), binop(U, Operator.Gt, lit_u64(1 << 36))
    l4, l5 = binop(K, Operator.NotEq, lit_i64(3)), binop(V, Operator.GtEq, lit_f64(-45.0))
    preds += [binop(binop(l1, A, l2), A, l3),                       # three tests over three columns, left-deep
              binop(l1, A, binop(l2, A, binop(l3, A, l4))),         # four tests, right-deep
              binop(binop(l1, O, l2), O, binop(l3, O, l4)),         # four tests, balanced, or
              binop(binop(binop(binop(l1, A, l2), A, l3), A, l4), A, l5),   # five tests: the general path
              binop(binop(l1, A, l2), O, l3)]                       # mixed and / or: the truth-table form
    # tests with an arithmetic step, nested and / or (ConjPred's general form: straight-line tests + truth table)
    m1 = binop(binop(K, Operator.Modulos, lit_i64(3)), Operator.Eq, lit_i64(0))
    m2 = binop(binop(V, Operator.Multiply, lit_f64(2.0)), Operator.Gt, lit_f64(50.0))
    m3 = binop(binop(lit_i64(10), Operator.Minus, K), Operator.GtEq, lit_i64(7))
    m4 = binop(binop(U, Operator.Divide, lit_u64(1000)), Operator.Lt, lit_u64(1 << 28))
    m5 = binop(binop(K, Operator.Modulos, lit_i64(-4)), Operator.NotEq, lit_i64(-1))
    preds += [binop(l2, O, m1), binop(m1, A, m2), binop(binop(m3, O, l3), A, binop(m4, O, m5)), binop(m2, O, binop(l1, A, binop(m1, O, m4))),
              binop(binop(binop(V, Operator.Divide, lit_f64(4.0)), Operator.LtEq, lit_f64(-0.0)), O, m3)]
    for pred in preds:
        exp = orc.selection([cols], pred.flatten(f4))[0]
        got = ctx.selection(t, pred.flatten(f4)).to_host()
        assert_batches_equal(got, exp, what=f"n={n} pred={pred!r}")
        expp = orc.projection(orc.selection([cols], pred.flatten(f4)), [binop(K, Operator.Plus, lit_i64(100)).flatten(f4), U.flatten(f4)])[0]
        gotp = ctx.selection_projection(t, pred.flatten(f4), [binop(K, Operator.Plus, lit_i64(100)).flatten(f4), U.flatten(f4)]).to_host()
        assert_batches_equal(gotp, expp, what=f"fused n={n} pred={pred!r}")


@pytest.mark.parametrize("groups", [5, 900, 5000, 70000])
def test_aggregate_two_test_predicates(ctx, groups):
    """`A and B` / `A or B` of two compares with literals — the usual WHERE clause — run inside the streaming aggregate kernel
    when the tested columns are the key column, the first value column or one more column (5, 900 groups), and as a materialised
    Boolean column on the subset / partitioned tiers (5000, 70000 groups), for nullable columns and for two extra columns"""
    rng = np.random.default_rng(groups + 1)
    n = 400_000
    k = rng.integers(-groups // 2, groups - groups // 2, n).astype(np.int64)
    v = rng.random(n) * 100 - 50
    v[:5] = [np.nan, np.inf, -np.inf, -0.0, 0.0]
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    u = rng.integers(0, 1 << 40, n).astype(np.uint64)
    z = rng.integers(0, 100, n).astype(np.int64)
    zmask = rng.random(n) > 0.1
    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w), Column.from_numpy(u), Column.from_numpy(z, zmask)]
    f5 = fields("k", "v", "w", "u", "z")
    t = ctx.table_from_host(cols)
    K, V, W, U, Z = col(0), col(1), col(2), col(3), col(4)
    A, O = Operator.And, Operator.Or
    preds = [binop(binop(K, Operator.Lt, lit_i64(groups // 4)), A, binop(V, Operator.Gt, lit_f64(-10.0))),        # key column, value column
             binop(binop(lit_f64(25.0), Operator.GtEq, V), O, binop(K, Operator.Eq, lit_i64(1))),                  # literal on the left, or
             binop(binop(W, Operator.NotEq, lit_i64(0)), A, binop(K, Operator.GtEq, lit_i64(-3))),                 # another column + key
             binop(binop(W, Operator.Gt, lit_i64(-500)), A, binop(W, Operator.LtEq, lit_i64(500))),                # a range on one other column
             binop(binop(U, Operator.Lt, lit_u64(1 << 39)), O, binop(V, Operator.NotEq, lit_f64(0.0))),            # UInt64 + Float64 (!= is true for NaN)
             binop(binop(W, Operator.Lt, lit_i64(0)), A, binop(U, Operator.Gt, lit_u64(1 << 38))),                 # two other columns: materialised
             binop(binop(Z, Operator.Lt, lit_i64(50)), A, binop(K, Operator.Gt, lit_i64(-100)))]                   # nullable column: materialised (Kleene)
    l1, l2, l3, l4 = (binop(K, Operator.GtEq, lit_i64(-groups // 4)), binop(V, Operator.Lt, lit_f64(40.0)), binop(W, Operator.Gt, lit_i64(-900)),
                      binop(W, Operator.NotEq, lit_i64(5)))
    preds += [binop(binop(l1, A, l2), A, l3),                       # three tests: key, value, one more column
              binop(l1, A, binop(l2, A, binop(l3, A, l4))),         # four tests, right-deep
              binop(binop(l1, O, l2), O, binop(l3, O, l4)),         # four tests, or
              binop(binop(l1, A, l2), O, l3)]                       # mixed and / or: materialised
    for pred in preds:
        for aggs in (ALL_AGGS(1), ALL_AGGS(1) + ALL_AGGS(2)):
            for key in (K, binop(K, Operator.Modulos, lit_i64(1 << 20))):
                exp = orc.aggregate([cols], aggs, group_nodes=key.flatten(f5), pred_nodes=pred.flatten(f5))[0]
                got = ctx.aggregate(t, aggs, group_nodes=key.flatten(f5), pred_nodes=pred.flatten(f5)).to_host()
                counts = [i for i, (fn, _) in enumerate(aggs) if fn == AggregateFunc.Count]
                assert_rows_multiset_equal(got, exp, RTOL, exact_cols=counts, what=f"groups={groups} pred={pred!r} key={key!r}")
    # un-grouped: the materialised form
    exp = orc.aggregate([cols], ALL_AGGS(1), pred_nodes=preds[0].flatten(f5))[0]
    got = ctx.aggregate(t, ALL_AGGS(1), pred_nodes=preds[0].flatten(f5)).to_host()
    assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what="un-grouped, two-test predicate")


def test_two_test_predicate_runs_inside_the_streaming_kernel(ctx):
    """no expression-machine pass and no second kernel for `k < c and v > d`"""
    rng = np.random.default_rng(3)
    n = 300_000
    cols = [Column.from_numpy(rng.integers(0, 100, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
    f2 = fields("k", "v")
    t = ctx.table_from_host(cols)
    pred = binop(binop(col(0), Operator.Lt, lit_i64(50)), Operator.And, binop(col(1), Operator.Gt, lit_f64(0.25))).flatten(f2)
    ctx.timing_enable(True)
    ctx.timing_reset()
    got = ctx.aggregate(t, ALL_AGGS(1), group_nodes=col(0).flatten(f2), pred_nodes=pred).to_host()
    ctx.timing_enable(False)
    rep = ctx.timing_report()
    assert rep["agg_grouped_fast"][1] == 1 and not any(name.startswith("expr") for name in rep), rep
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f2), pred_nodes=pred)[0]
    assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what="two-test predicate, in-kernel")


@pytest.mark.parametrize("groups", [7, 1000, 3000])
def test_aggregate_special_float_values_under_random_keys(ctx, groups):
    """NaN, +-inf, +-0, subnormals as VALUES where every row's key differs from its neighbour's (the kernel's batch loop
    keeps min/max in LDS as doubles behind ordered compares): NaN is ignored by min and makes max NaN (OrderedFloat, max.rs:38-50),
    a group of nothing but +inf keeps the f64::MAX start of min (min.rs), counts are exact"""
    rng = np.random.default_rng(groups)
    n = 300_000
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 5e-324, -5e-324, 1.0, -1.0])  # (no +-DBL_MAX: their sum depends on the order)
    v = rng.random(n) * 200 - 100
    where = rng.random(n) < 0.02
    v[where] = special[rng.integers(0, len(special), int(where.sum()))]
    k = rng.integers(0, groups, n).astype(np.int64) * 3 - groups
    v[k == k.min()] = np.inf          # one group of +inf only
    v[k == k.max()] = np.nan          # one group of NaN only
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    t = ctx.table_from_host(cols)
    for key in (col(0), binop(col(0), Operator.Modulos, lit_i64(512))):
        exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=key.flatten(f2))[0]
        got = ctx.aggregate(t, ALL_AGGS(1), group_nodes=key.flatten(f2)).to_host()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"special values, groups={groups}, key={key!r}")


def test_aggregate_key_subset_tier_is_taken_between_one_table_and_partitioning(ctx):
    """5000 groups spread over a range no table addresses directly: the streaming kernel with two HASHED key subsets, not the partition
    kernels; 20000: partitioned.  (A range of up to 2 x 4096 values takes direct-mapped subsets, one of up to ~1.3 million the partitioned
    path's range tier: test_aggregate_measured_key_range_over_two_key_subsets.)"""
    rng = np.random.default_rng(11)
    n = 700_000
    f2 = fields("k", "v")
    v = rng.random(n)
    for groups, want_partition in ((5000, False), (20000, True)):
        k = rng.integers(0, groups, n).astype(np.int64) * 1_000_033 - 77  # (k % 1_000_003 below: 30 g - 77, as many groups)
        cols = [Column.from_numpy(k), Column.from_numpy(v)]
        t = ctx.table_from_host(cols)
        exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f2))[0]
        for _ in range(2):  # the second run starts from the plan hint
            ctx.timing_enable(True)
            ctx.timing_reset()
            got = ctx.aggregate(t, ALL_AGGS(1), group_nodes=col(0).flatten(f2)).to_host()
            ctx.timing_enable(False)
            assert (ctx.timing_query("agg_partition_scatter")[1] > 0) == want_partition, f"groups={groups}"
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"key subsets, groups={groups}")
        # Int64 values under an interpreted chain predicate, key `k % m` by magic multiply: the other instances of the tier
        w = rng.integers(-1000, 1000, n).astype(np.int64)
        cols3 = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
        f3 = fields("k", "v", "w")
        t3 = ctx.table_from_host(cols3)
        pred = binop(binop(binop(col(2), Operator.Plus, lit_i64(7)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(6)).flatten(f3)
        key = binop(col(0), Operator.Modulos, lit_i64(1_000_003)).flatten(f3)
        exp = orc.aggregate([cols3], ALL_AGGS(2), group_nodes=key, pred_nodes=pred)[0]
        got = ctx.aggregate(t3, ALL_AGGS(2), group_nodes=key, pred_nodes=pred).to_host()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"key subsets (chain predicate, Int64 values), groups={groups}")


def test_aggregate_two_level_partitioning_millions_of_groups(ctx):
    """more distinct keys per partition than a workgroup table holds → second partitioning level (512 x 64)"""
    rng = np.random.default_rng(5)
    n, groups = 3_000_000, 2_600_000
    k = rng.integers(0, groups, n).astype(np.int64) * 3 - groups
    v = rng.random(n)
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f2))[0]
    got, gk = ctx.aggregate(ctx.table_from_host(cols), ALL_AGGS(1), group_nodes=col(0).flatten(f2), with_keys=True)
    assert got.num_rows == len(np.unique(k))
    assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what="two-level partitioned aggregate")
    assert (gk.to_host()[0].to_numpy() == np.unique(k)).all()


@pytest.mark.parametrize("n", [1000, 4097, 70001])  # none a multiple of 64: a borrowed bitmap ends at ceil(n/8) bytes
def test_borrowed_boolean_predicate_column_is_not_read_past_its_end(ctx, n):
    """a caller-owned (NQE_DEVICE) Boolean column used directly as the aggregate predicate: the fast kernels read predicate
    bits as whole 64-bit words, which is only legal for padded (library-owned) bitmaps — a borrowed one whose length is not a
    multiple of 64 must take the general kernel."""
    rng = np.random.default_rng(n)
    ids = rng.integers(0, 50, n).astype(np.int64)
    v = rng.random(n) * 10.0
    b = rng.random(n) < 0.5
    cols = [Column.from_numpy(ids), Column.from_numpy(v), Column.from_numpy(b)]
    owner = ctx.table_from_host(cols)  # device memory the "caller" owns; the table below only borrows it
    tab = ctx.table_from_device([(DType(owner.column_info(i).dtype), n, owner.column_info(i).values, None) for i in range(3)])
    f = fields("id", "v", "b")
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
    for key in (col(0).flatten(f), None):
        got = ctx.aggregate(tab, aggs, group_nodes=key, pred_nodes=col(2).flatten(f)).to_host()
        exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=col(2).flatten(f))[0]
        assert_rows_multiset_equal(got, exp, exact_cols=(0,))


# --------------------------------------------------------------------------- hashed unique builds: buckets, filler, fall-backs
GOLD = 0x9E3779B97F4A7C15
GOLD_INV = pow(GOLD, -1, 1 << 64)


def _join_both_ways(ctx, left, right):
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0).to_host()
    exp = orc.hash_join([left], [right], 0, 0)[0]
    assert_batches_equal(got, exp, what="hash join")
    return got


@pytest.mark.parametrize("payload_cols", [1, 2])
@pytest.mark.parametrize("colliding", [5, 9, 60, 100, 300])
def test_hash_join_keys_that_all_land_in_one_bucket(ctx, colliding, payload_cols):
    """keys j * GOLD^-1 (mod 2^64) hash to slot 0 whatever the table size: a full 8-slot bucket (the wave-cooperative probe's slow
    path), probe sequences of up to `colliding` slots, and — beyond the insert's bound — the fall-back to the sort-based build"""
    rng = np.random.default_rng(colliding)
    bad = np.array([(j * GOLD_INV) % (1 << 64) for j in range(1, colliding + 1)], dtype=np.uint64)
    other = rng.integers(1 << 40, 1 << 62, 5000).astype(np.uint64)
    keys = np.unique(np.concatenate([bad, other]))
    rng.shuffle(keys)
    left = [Column.from_numpy(keys)] + [Column.from_numpy(rng.integers(0, 1 << 50, len(keys)).astype(np.int64)) for _ in range(payload_cols)]
    probe = np.concatenate([rng.choice(keys, 20000), bad, rng.integers(1 << 40, 1 << 62, 3000).astype(np.uint64),
                            np.array([(j * GOLD_INV) % (1 << 64) for j in range(colliding + 1, colliding + 40)], dtype=np.uint64)])  # misses that walk the same bucket
    rng.shuffle(probe)
    right = [Column.from_numpy(probe), Column.from_numpy(rng.random(len(probe)))]
    _join_both_ways(ctx, left, right)


@pytest.mark.parametrize("build_has", ["max_key", "zero_and_max", "zero"])
def test_hash_join_filler_value_of_the_payload_table(ctx, build_has):
    """the {key, payload} table marks empty slots with a value outside the build keys' range (max + 1, else min - 1; none when the
    keys span 0 .. 2^64-1): probing exactly that value must not match, and keys 0 / 2^64-1 must work"""
    rng = np.random.default_rng(1)
    keys = np.unique(rng.integers(1 << 45, 1 << 62, 3000).astype(np.uint64))
    extra = {"max_key": [(1 << 64) - 1], "zero_and_max": [0, (1 << 64) - 1], "zero": [0]}[build_has]
    keys = np.concatenate([keys, np.array(extra, dtype=np.uint64)])
    rng.shuffle(keys)
    left = [Column.from_numpy(keys), Column.from_numpy(rng.integers(-5, 5, len(keys)).astype(np.int64))]
    kmax, kmin = int(keys.max()), int(keys.min())
    specials = [0, 1, (1 << 64) - 1, (1 << 64) - 2, (kmax + 1) % (1 << 64), (kmin - 1) % (1 << 64)]
    probe = np.concatenate([rng.choice(keys, 10000), np.array(specials * 50, dtype=np.uint64), rng.integers(0, 1 << 63, 2000).astype(np.uint64)])
    rng.shuffle(probe)
    right = [Column.from_numpy(probe), Column.from_numpy(rng.random(len(probe)))]
    _join_both_ways(ctx, left, right)


@pytest.mark.parametrize("nb", [1, 7, 8, 9, 64, 65, 1000])
def test_hash_join_sort_free_build_small_and_signed_keys(ctx, nb):
    """unique Int64 keys incl. negatives (raw 8-byte slots compare; a mixed-sign key set is never 'dense'), tiny builds"""
    rng = np.random.default_rng(nb)
    keys = rng.permutation(np.arange(-(nb // 2), nb - nb // 2, dtype=np.int64) * 3)
    left = [Column.from_numpy(keys), Column.from_numpy(rng.random(nb)), Column.from_numpy(rng.integers(0, 9, nb).astype(np.int64))]
    probe = rng.integers(-3 * nb - 3, 3 * nb + 3, 5000).astype(np.int64)
    right = [Column.from_numpy(probe), Column.from_numpy(np.arange(5000, dtype=np.int64))]
    _join_both_ways(ctx, left, right)


@pytest.mark.parametrize("heavy_frac", [0.5, 0.9])
def test_aggregate_partitioned_path_with_a_heavy_key(ctx, heavy_frac):
    """skewed keys on the partitioned path: one key holds most rows, so its hash partition outgrows the fixed-capacity slabs of the
    count-free scatter (NQE_FLAG_SLAB_OVERFLOW) and the query is redone with exact partition sizes; the other 20000 keys keep the
    query off the single-pass path.  Executed twice: the second run starts from the recorded plan hint."""
    rng = np.random.default_rng(int(heavy_frac * 10))
    n = 600_000
    k = rng.integers(0, 20000, n).astype(np.int64)
    k[rng.random(n) < heavy_frac] = 777
    v = rng.random(n) * 100 - 50
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    t = ctx.table_from_host(cols)
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f2))[0]
    for _ in range(2):
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=col(0).flatten(f2), with_keys=True)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what="heavy-key partitioned aggregate")
        assert (gk.to_host()[0].to_numpy() == np.unique(k)).all()


def test_aggregate_plan_hint_is_per_query_shape(ctx):
    """a query that had to be partitioned leaves a hint keyed by its whole shape; the same key column under a different predicate /
    with nullable values must not inherit it blindly (it takes kernels that have no partitioned form)"""
    rng = np.random.default_rng(3)
    n = 500_000
    k = rng.integers(0, 30000, n).astype(np.int64)
    v = rng.random(n)
    vn = Column.from_numpy(rng.random(n), rng.random(n) > 0.1)
    b = Column.from_numpy(rng.random(n) < 0.5)
    cols = [Column.from_numpy(k), Column.from_numpy(v), vn, b]
    f4 = fields("k", "v", "vn", "b")
    t = ctx.table_from_host(cols)
    key = col(0).flatten(f4)
    shapes = [(ALL_AGGS(1), None), (ALL_AGGS(1), None), (ALL_AGGS(2), None), (ALL_AGGS(1), col(3).flatten(f4)),
              (ALL_AGGS(1), binop(col(1), Operator.Gt, lit_f64(0.25)).flatten(f4)), (ALL_AGGS(1) + ALL_AGGS(2), None), (ALL_AGGS(1), None)]
    for aggs, pred in shapes:
        exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
        got = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{aggs} {pred is not None}")


@pytest.mark.parametrize("pred_kind", ["none", "key_range", "other_column", "chain"])
def test_aggregate_partitioned_path_two_value_columns(ctx, pred_kind):
    """the slab form with two value columns in one pass (24-byte tuples, 4 rows per thread per tile) under every predicate variant
    of the scatter kernel; Int64 and Float64 values (`val as f64`)"""
    rng = np.random.default_rng(len(pred_kind))
    n, groups = 330_000, 45_000
    k = rng.integers(-groups // 2, groups // 2, n).astype(np.int64)
    v = rng.random(n) * 200 - 100
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    f3 = fields("k", "v", "w")
    t = ctx.table_from_host(cols)
    pred = {"none": None, "key_range": binop(col(0), Operator.GtEq, lit_i64(-groups // 4)), "other_column": binop(col(2), Operator.Lt, lit_i64(500)),
            "chain": binop(binop(col(2), Operator.Modulos, lit_i64(7)), Operator.NotEq, lit_i64(3))}[pred_kind]
    pn = pred.flatten(f3) if pred is not None else None
    aggs = ALL_AGGS(1) + ALL_AGGS(2)
    exp = orc.aggregate([cols], aggs, group_nodes=col(0).flatten(f3), pred_nodes=pn)[0]
    for _ in range(2):  # second run: from the plan hint
        got = ctx.aggregate(t, aggs, group_nodes=col(0).flatten(f3), pred_nodes=pn)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0, 5], what=f"two value columns, predicate {pred_kind}")


@pytest.mark.parametrize("nullable", [False, True])
@pytest.mark.parametrize("m", [1, 2, 3, 7, 16, 100, 511, 512, 513, 2047])
def test_aggregate_small_modulus_keys_replicated_table(ctx, m, nullable):
    """`group by id % m` with few groups (C1's own `id % 3`): the direct-mapped LDS table is replicated per lane group so that a
    wave's rows do not all update the same words (aggregate_fast_kernel.hpp, direct_rep), and the replicas are folded before the
    merge.  Signed ids on both sides of zero (keys in (-m, m)), keys that change on every row of a thread, one / two / three value
    columns (one and two passes), nullable values and keys, an unsigned source; counts exact, sums 1e-9"""
    rng = np.random.default_rng(1000 + m)
    n = 300_007
    ids = (np.arange(n, dtype=np.int64) - n // 3) * (1 if m % 2 else 3)
    age = rng.integers(-50, 50, n).astype(np.int64)
    v = rng.random(n) * 200.0 - 100.0
    u = rng.integers(0, 1 << 40, n).astype(np.uint64)
    mk = (lambda: rng.random(n) > 0.1) if nullable else (lambda: None)
    cols = [Column.from_numpy(ids, mk()), Column.from_numpy(age, mk()), Column.from_numpy(v, mk()), Column.from_numpy(u)]
    f4 = fields("id", "age", "v", "u")
    t = ctx.table_from_host(cols)
    for key_col in (0, 3):
        key = binop(col(key_col), Operator.Modulos, lit_i64(m) if key_col == 0 else lit_u64(m)).flatten(f4)
        for aggs in (ALL_AGGS(2), ALL_AGGS(1) + ALL_AGGS(2), [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 2)]):
            for pred in (None, binop(col(0), Operator.Lt, lit_i64(n // 4)).flatten(f4)):
                exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
                got, gk = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred, with_keys=True)
                exact = [i for i, (fn, _) in enumerate(aggs) if fn == AggregateFunc.Count]
                assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=exact, what=f"m={m} key_col={key_col} aggs={len(aggs)} pred={pred is not None}")
                kk = gk.to_host()[0].to_numpy()
                assert (np.diff(kk.astype(np.int64 if key_col == 0 else np.uint64)) > 0).all() and len(kk) == exp[0].length


def test_aggregate_slab_allocation_failure_falls_back_to_the_exact_form():
    """when the slab form's scatter buffers cannot be allocated the partitioned aggregate must take the exact
    count/scan/scatter form instead of failing (NQE_TEST_SLAB_OOM makes the allocation fail), and remember it per query shape"""
    import subprocess
    import sys

    code = r'''
import numpy as np
from naive_query_engine_amd import AggregateFunc, Column, capi
from naive_query_engine_amd.expression import col
from oracle import oracle as orc
from tests.helpers import assert_rows_multiset_equal, fields
rng = np.random.default_rng(3)
n = 600_000
cols = [Column.from_numpy(rng.integers(0, 50_000, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
f = fields("k", "v")
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
ctx = capi.Context(0)
t = ctx.table_from_host(cols)
exp = orc.aggregate([cols], aggs, group_nodes=col(0).flatten(f))[0]
for rep in range(2):   # the second execution starts from the plan hint (exact form)
    ctx.timing_enable(True); ctx.timing_reset()
    got = ctx.aggregate(t, aggs, group_nodes=col(0).flatten(f)).to_host()
    ctx.timing_enable(False)
    names = set(ctx.timing_report())
    assert_rows_multiset_equal(got, exp, 1e-9, exact_cols=[0])
    assert "agg_partition_count" in names, names      # the exact form ran
print("fallback ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NQE_TEST_SLAB_OOM="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0 and "fallback ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("groups", [3, 1000, 70000])
def test_aggregate_predicate_trees_run_inside_the_streaming_kernel(ctx, groups):
    """predicate trees that are neither a chain nor a list of range tests (`v < 20 or id % 3 = 0`, `a + w > 50 and …`, a literal on
    the left of a subtraction, column-with-column arithmetic and comparisons, three columns): the typed stack machine of
    aggregate_common.hpp (tree_pred_eval, PRED = 5) inside agg_grouped_fast_kernel — no materialised Boolean column (no expr_tree
    launch) while the groups fit one workgroup table; with 70000 groups (partitioned path) and for shapes the machine does not take
    (two arithmetic subtrees alive at once) the predicate is materialised.  Always equal to the oracle."""
    rng = np.random.default_rng(77 + groups)
    n = 400_003
    ids = rng.permutation(n).astype(np.int64) - n // 5
    k = rng.integers(0, groups, n).astype(np.int64)
    v = rng.random(n) * 100.0
    w = rng.integers(-100, 100, n).astype(np.int64)
    v[::97] = np.nan
    cols = [Column.from_numpy(ids), Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    f4 = fields("id", "k", "v", "w")
    t = ctx.table_from_host(cols)
    I, K, V, W = col(0), col(1), col(2), col(3)
    B = lambda a, op, b: binop(a, op, b)
    O = Operator
    trees = [
        (True, B(B(V, O.Lt, lit_f64(20.0)), O.Or, B(B(I, O.Modulos, lit_i64(3)), O.Eq, lit_i64(0)))),
        (True, B(B(B(lit_i64(100), O.Minus, W), O.Gt, lit_i64(120)), O.Or, B(B(V, O.Multiply, lit_f64(2.0)), O.GtEq, lit_f64(150.0)))),
        (True, B(B(B(W, O.Plus, K), O.Lt, lit_i64(50)), O.And, B(B(V, O.NotEq, V), O.Or, B(lit_f64(30.0), O.Lt, V)))),     # col + col, NaN != NaN, literal-left compare
        (True, B(B(B(B(W, O.Divide, lit_i64(7)), O.Multiply, lit_i64(3)), O.LtEq, K), O.Or, B(W, O.Eq, lit_i64(5)))),       # chain compared with a column
        (True, B(B(K, O.Lt, W), O.And, B(B(K, O.Gt, lit_i64(1)), O.Or, B(B(W, O.Modulos, lit_i64(-4)), O.Eq, lit_i64(-3))))),
        (False, B(B(B(W, O.Plus, lit_i64(1)), O.Lt, B(K, O.Multiply, lit_i64(2))), O.Or, B(W, O.Eq, lit_i64(5)))),         # two arithmetic subtrees alive
        (False, B(B(B(W, O.Modulos, lit_i64(0)), O.Lt, lit_i64(3)), O.Or, B(W, O.Eq, lit_i64(5)))),                           # faults: DivideByZero
    ]
    aggs = ALL_AGGS(2)
    key = K.flatten(f4)
    for in_kernel, tree in trees:
        pred = tree.flatten(f4)
        try:
            exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
        except ErrorCode as e:
            with pytest.raises(ErrorCode) as ge:
                ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)
            assert ge.value.status == e.status
            continue
        for rep in range(2):  # the second execution starts from the plan hint
            ctx.timing_enable(True)
            ctx.timing_reset()
            got = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred).to_host()
            ctx.timing_enable(False)
            names = set(ctx.timing_report())
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"groups={groups} tree={tree!r}")
            if groups <= 1000:
                assert ("expr_tree" not in names) == in_kernel, (names, repr(tree))


@pytest.mark.parametrize("groups", [7, 5000, 70000])
def test_aggregate_mixed_sign_zeros_across_kernel_paths(ctx, groups):
    """groups holding both +0.0 and -0.0 (and nothing else, or zeros as the extreme): on every tier (one workgroup table, two key
    subsets, slabs) min and max EQUAL the oracle's under `==` — which zero comes back is arrival order (the LDS tables compare plain
    doubles, the global table the total-order image; the reference keeps the first in row order): the documented divergence of
    DESIGN §4, pinned here so that nothing beyond the sign of a zero can ever differ"""
    rng = np.random.default_rng(groups)
    n = 400_000
    k = rng.integers(0, groups, n).astype(np.int64)
    v = np.where(rng.random(n) < 0.5, 0.0, -0.0)
    v[k % 3 == 1] = np.abs(rng.random(int((k % 3 == 1).sum())))          # zeros are the minimum there
    v[k % 3 == 2] = -np.abs(rng.random(int((k % 3 == 2).sum())))         # ... and the maximum here
    zero = rng.random(n) < 0.3
    v[zero] = np.where(rng.random(int(zero.sum())) < 0.5, 0.0, -0.0)
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1), (AggregateFunc.Sum, 1)]
    exp = orc.aggregate([cols], aggs, group_nodes=col(0).flatten(f2))[0]
    t = ctx.table_from_host(cols)
    for rep in range(2):
        got = ctx.aggregate(t, aggs, group_nodes=col(0).flatten(f2)).to_host()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"groups={groups}")   # -0.0 == 0.0 under allclose / ==
        g = rows_sorted(got)
        assert not np.isnan(g).any()


@pytest.mark.parametrize("span_bits", [3, 13, 25, 31, 40])
@pytest.mark.parametrize("shape", ["pk", "holes", "dup"])
def test_join_large_dense_builds_without_device_atomics(ctx, shape, span_bits):
    """builds of >= 2^16 rows take the two-kernel form (dense_scatter_rows_kernel + dense_finish_kernel: row numbers scattered with
    plain stores, presence words / occupancy count / bit-packed payloads produced in key order): a gap-free primary key, keys with
    holes (presence bitmap, entries of absent keys), a build size that is no multiple of 64, every payload width, and duplicate
    keys — detected by the occupancy count, after which the sort-based build takes over; probe order and duplicate order exact"""
    rng = np.random.default_rng(span_bits * 7 + len(shape))
    nb, n = 70_001, 200_000
    if shape == "pk":
        dk = rng.permutation(nb).astype(np.int64) - 500
    elif shape == "holes":
        dk = rng.permutation(3 * nb)[:nb].astype(np.int64) - 500
    else:
        dk = rng.permutation(nb).astype(np.int64)
        dk[rng.integers(0, nb, 50)] = dk[rng.integers(0, nb, 50)]      # a few duplicate keys
    top = (1 << span_bits) - 1
    a = rng.integers(0, top + 1, nb).astype(np.int64) - (1 << 10)
    a[:2] = [-(1 << 10), top - (1 << 10)]
    c = rng.random(nb)
    left = [Column.from_numpy(dk), Column.from_numpy(a), Column.from_numpy(c)]
    rk = rng.integers(int(dk.min()) - 10, int(dk.max()) + 10, n).astype(np.int64)
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    got = ctx.hash_join(lt, rt, 0, 0).to_host()
    assert_batches_equal(got, exp, what=f"{shape}, payload range of {span_bits} bits")
    jt = ctx.hash_join_build(lt, 0)                                 # the reused table: optimistic form / its fallback / two passes
    for _ in range(2):
        assert_batches_equal(ctx.hash_join_probe(jt, rt, 0).to_host(), exp, what=f"{shape} reused table")


@pytest.mark.parametrize("ncols", [0, 1, 2, 5, 9])
@pytest.mark.parametrize("shape", ["pk", "sorted", "holes", "dup"])
def test_join_partitioned_dense_build(ctx, monkeypatch, shape, ncols):
    """builds of >= 2^25 rows (here: the threshold lowered through NQE_JOIN_PART_BUILD_MIN) partition the rows by key range first
    (part_build_count / scatter / place kernels: tuples {key - min | row, payload words}, every tile width of the scatter), then
    finish in key order from the key-ordered payload copies: random and ascending primary keys, keys with holes, a row count that
    is no multiple of the tile, 0 … 9 payload columns of every packing, duplicates (occupancy count → sort-based build)"""
    monkeypatch.setenv("NQE_JOIN_PART_BUILD_MIN", "1000")
    if shape == "sorted":
        monkeypatch.setenv("NQE_JOIN_NO_ASCENDING", "1")     # (ascending keys normally skip the partitioning: next test)
    rng = np.random.default_rng(ncols * 11 + len(shape))
    nb, n = 150_001, 300_000
    if shape == "pk":
        dk = rng.permutation(nb).astype(np.int64) + 700
    elif shape == "sorted":
        dk = np.arange(nb, dtype=np.int64) + 5
    elif shape == "holes":
        dk = rng.permutation(3 * nb)[:nb].astype(np.int64) + 700
    else:
        dk = rng.permutation(nb).astype(np.int64)
        dk[rng.integers(0, nb, 50)] = dk[rng.integers(0, nb, 50)]
    left = [Column.from_numpy(dk)]
    for c in range(ncols):
        if c % 3 == 2:
            left.append(Column.from_numpy(rng.random(nb)))
        else:
            bits = [20, 31, 45, 7, 25, 2][c % 6]
            a = rng.integers(0, 1 << bits, nb).astype(np.int64) - (1 << 9)
            a[:2] = [-(1 << 9), (1 << bits) - 1 - (1 << 9)]
            left.append(Column.from_numpy(a))
    rk = rng.integers(int(dk.min()) - 10, int(dk.max()) + 10, n).astype(np.int64)
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    ctx.timing_enable(True); ctx.timing_reset()
    got = ctx.hash_join(lt, rt, 0, 0).to_host()
    took = ctx.timing_query("join_build_part_scatter")[1]
    ctx.timing_enable(False)
    assert took >= 1, "the partitioned build did not run"    # (non-negative keys: the unsigned key range is dense)
    assert_batches_equal(got, exp, what=f"{shape}, {ncols} payload columns")
    jt = ctx.hash_join_build(lt, 0)
    for _ in range(2):
        assert_batches_equal(ctx.hash_join_probe(jt, rt, 0).to_host(), exp, what=f"{shape} reused table")


@pytest.mark.parametrize("one_level", [False, True])
@pytest.mark.parametrize("shape", ["pk_int20", "pk_f64", "holes_int40", "key_only", "dup", "holes_key_only"])
def test_join_partitioned_dense_build_two_levels(ctx, monkeypatch, shape, one_level):
    """the two-level form of the partitioned build (round 6: count with a fine histogram, scatter into partitions, a second scatter
    by fine bin inside every partition, LDS fill of the final tables — key-only builds and one payload word) over a key range of
    several partitions (2.6 x 10^6 keys: 10 partitions, 320 fine bins, tiles that end inside a partition and inside a bin): bit-packed,
    32-bit and 8-byte payload columns, keys with holes (absent entries inside and at the end of the table), duplicates (occupancy
    count -> sort-based build) — and the same builds through the one-level form (NQE_JOIN_PART_ONE_LEVEL=1).  hash_join.rs:124-166"""
    monkeypatch.setenv("NQE_JOIN_PART_BUILD_MIN", "1000")
    if one_level:
        monkeypatch.setenv("NQE_JOIN_PART_ONE_LEVEL", "1")
    rng = np.random.default_rng(len(shape) + 77)
    nb, n = 2_600_003, 400_000
    holes = shape.startswith("holes")
    dk = (rng.permutation(3 * nb)[:nb] if holes else rng.permutation(nb)).astype(np.int64) + 12345
    if shape == "dup":
        dk[rng.integers(0, nb, 30)] = dk[rng.integers(0, nb, 30)]
    left = [Column.from_numpy(dk)]
    if shape in ("pk_int20", "dup"):
        left.append(Column.from_numpy(rng.integers(0, 1 << 20, nb).astype(np.int64) - 77))
    elif shape == "pk_f64":
        left.append(Column.from_numpy(rng.random(nb)))
    elif shape == "holes_int40":
        left.append(Column.from_numpy(rng.integers(0, 1 << 40, nb).astype(np.int64)))
    rk = rng.integers(int(dk.min()) - 10, int(dk.max()) + 10, n).astype(np.int64)
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    ctx.timing_enable(True); ctx.timing_reset()
    got = ctx.hash_join(lt, rt, 0, 0).to_host()
    names = ctx.timing_report()
    ctx.timing_enable(False)
    assert "join_build_part_scatter" in names, names
    assert ("join_build_part_fill" in names) == (not one_level) and ("join_build_part_place" in names) == one_level, names
    assert_batches_equal(got, exp, what=f"{shape}, one_level={one_level}")
    jt = ctx.hash_join_build(lt, 0)
    assert_batches_equal(ctx.hash_join_probe(jt, rt, 0).to_host(), exp, what=f"{shape} reused table")


@pytest.mark.parametrize("keys", ["dense", "sparse", "dense_gaps", "sparse_two_payloads"])
@pytest.mark.parametrize("all_match", [True, False])
def test_join_in_which_every_probe_row_matches_shares_the_probe_columns(ctx, keys, all_match):
    """unique build keys and every probe row matching: output row = probe row, so the probe-side columns of the output are the
    probe table's own buffers (dense keys: the optimistic one-pass form writes the build payload only; hashed keys with the payload
    in the slot: the words the lookup wrote are the payload column, no second pass) — same rows, same order, the same device
    pointers, and still readable after the probe table's handle is released; one absent key and everything is written as before"""
    import gc
    rng = np.random.default_rng(3 + len(keys))
    nb, n = 50_000, 130_001
    if keys == "dense":
        dk = rng.permutation(nb).astype(np.int64) + 2
    elif keys == "dense_gaps":                      # a primary key with gaps: the two-pass dense form, which finds n matches out of n
        dk = rng.permutation(2 * nb)[:nb].astype(np.int64) + 2
    else:
        dk = (np.arange(nb, dtype=np.int64) << 21) + rng.integers(0, 1 << 21, nb)
    dk = dk[rng.permutation(nb)]
    left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1 << 18, nb).astype(np.int64))]
    if keys == "sparse_two_payloads":               # no payload-in-the-slot table: the generic unique-key form (build rows recorded, gathered)
        left.append(Column.from_numpy(rng.random(nb)))
    rk = dk[rng.integers(0, nb, n)].copy()
    if not all_match:
        rk[n // 3] = dk.max() + 7
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    for attempt in range(2):                                   # (the second execution of the failing shape starts in the two-pass form)
        out = ctx.hash_join(lt, rt, 0, 0)
        nl = len(left)
        shared = [out.column_info(nl + j).values == rt.column_info(j).values for j in range(2)]
        assert out.column_info(0).values == out.column_info(nl).values           # the two key columns are one buffer either way
        assert shared == [all_match, all_match], (keys, all_match, shared)
        assert_batches_equal(out.to_host(), exp, what=f"{keys} keys, all_match={all_match}")
    del rt
    gc.collect()
    scratch = ctx.table_from_host([Column.from_numpy(rng.random(n)) for _ in range(4)])   # allocations that would reuse released blocks
    assert_batches_equal(out.to_host(), exp, what="after the probe table was released")
    del scratch


@pytest.mark.parametrize("nb", [70_001, 70_002, 300_000])
@pytest.mark.parametrize("order", ["random", "ascending", "one_descent"])
def test_join_build_over_borrowed_columns_that_are_only_word_aligned(ctx, nb, order):
    """the build's min / max / descents pass reads 16-byte-aligned columns in pairs of words; a BORROWED build column (NQE_DEVICE) need only be
    8-byte aligned (include/nqe.h) and then takes the word-by-word form — both must find the same range and the same number of descents (an
    ascending build side is built without the partition passes; a wrong count would only change the form, a wrong range the result), for even
    and odd row counts, with a descent exactly between two pairs"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    rng = np.random.default_rng(nb + len(order))
    dk = np.arange(nb, dtype=np.int64) * 3 - 1000
    if order == "random":
        dk = dk[rng.permutation(nb)]
    elif order == "one_descent":
        dk[[4001, 4002]] = dk[[4002, 4001]]  # rows 4001 > 4002: a descent between the pair (4000, 4001) and the pair (4002, 4003)
    dv = rng.integers(0, 1 << 18, nb).astype(np.int64)
    n = 200_003
    rk = dk[rng.integers(0, nb, n)].copy()
    rk[::7] = 5  # absent keys
    rv = rng.random(n)
    exp = orc.hash_join([[Column.from_numpy(dk), Column.from_numpy(dv)]], [[Column.from_numpy(rk), Column.from_numpy(rv)]], 0, 0)[0]
    rt = ctx.table_from_host([Column.from_numpy(rk), Column.from_numpy(rv)])
    bufs = [ctx.device_alloc(nb * 8 + 16) for _ in range(2)]
    try:
        for shift in (0, 8):  # aligned to 16 bytes; aligned to 8 only
            for buf, arr in zip(bufs, (dk, dv)):
                ctx.synchronize()
                assert hip.hipMemcpy(ctypes.c_void_p(buf + shift), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes), 1) == 0
            assert hip.hipDeviceSynchronize() == 0
            lt = ctx.table_from_device([(DType.INT64, nb, bufs[0] + shift, None), (DType.INT64, nb, bufs[1] + shift, None)])
            got = ctx.hash_join(lt, rt, 0, 0).to_host()
            assert_batches_equal(got, exp, what=f"{order} build keys at +{shift} bytes, {nb} rows")
            del lt
    finally:
        for b in bufs:
            ctx.device_free(b)


@pytest.mark.parametrize("keys", ["dense", "sparse", "dense_gaps", "sparse_two_payloads"])
@pytest.mark.parametrize("immutable", [False, True])
def test_join_output_never_aliases_borrowed_probe_memory(ctx, keys, immutable):
    """outputs are callee-allocated and owned by the caller.  A probe table over BORROWED device memory (nqe_table_create
    with NQE_DEVICE columns) is the caller's to free or overwrite once the join has returned: the all-match forms that share the
    probe table's columns apply only to memory the library owns or to tables created with NQE_TABLE_IMMUTABLE.  Here the borrowed
    probe buffers are overwritten right after the join and the output must still equal the oracle's (default), respectively alias
    them (immutable: same device pointers)."""
    rng = np.random.default_rng(11 + len(keys))
    nb, n = 40_000, 100_003
    if keys == "dense":
        dk = rng.permutation(nb).astype(np.int64) + 2
    elif keys == "dense_gaps":
        dk = rng.permutation(2 * nb)[:nb].astype(np.int64) + 2
    else:
        dk = (np.arange(nb, dtype=np.int64) << 21) + rng.integers(0, 1 << 21, nb)
    dk = dk[rng.permutation(nb)]
    left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1 << 18, nb).astype(np.int64))]
    if keys == "sparse_two_payloads":
        left.append(Column.from_numpy(rng.random(nb)))
    rk = dk[rng.integers(0, nb, n)].copy()
    rv = rng.random(n)
    exp = orc.hash_join([left], [[Column.from_numpy(rk), Column.from_numpy(rv)]], 0, 0)[0]
    lt = ctx.table_from_host(left)
    # the caller's own device memory: torch tensors
    import torch

    tk, tv = torch.from_numpy(rk).cuda(), torch.from_numpy(rv).cuda()
    torch.cuda.synchronize()
    pk, pv = tk.data_ptr(), tv.data_ptr()
    try:
        rt = ctx.table_from_device([(DType.INT64, n, pk, None), (DType.FLOAT64, n, pv, None)], immutable=immutable)
        nl = len(left)
        out = ctx.hash_join(lt, rt, 0, 0)
        ctx.synchronize()
        aliased = [out.column_info(nl + j).values == p for j, p in enumerate((pk, pv))]
        assert aliased == [immutable, immutable], (keys, immutable, aliased)
        # a projected bare column follows the same rule
        pr = ctx.projection(rt, [col(1).flatten(fields("key", "val"))])
        assert (pr.column_info(0).values == pv) == immutable
        if not immutable:
            # the caller reuses its buffers: every word overwritten
            ctx.synth_fill(1, 99, 0, n, 1 << 40, 0, pk)
            ctx.synth_fill(1, 98, 0, n, 1 << 40, 0, pv)
            ctx.synchronize()
        assert_batches_equal(out.to_host(), exp, what=f"{keys} keys, borrowed probe memory, immutable={immutable}")
        if not immutable:
            assert_column_equal(pr.to_host()[0], Column.from_numpy(rv), what="projected bare column of borrowed memory")
        del out, pr, rt
    finally:
        ctx.synchronize()
        del tk, tv


def test_join_duplicate_keys_when_the_sorted_payload_copy_does_not_fit(ctx, monkeypatch):
    """duplicate build keys: the key-ordered copy of each plain payload column is an optimisation on top of the table — when its
    allocation fails (NQE_TEST_SORTED_COLS_OOM stands in for the failed hipMalloc) the build still succeeds and the probe gathers
    the payload through the permutation: same rows, same order"""
    rng = np.random.default_rng(5)
    nb, n = 30_000, 80_000
    dk = rng.integers(0, nb // 4, nb).astype(np.int64)
    left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1 << 40, nb).astype(np.int64)), Column.from_numpy(rng.random(nb))]
    right = [Column.from_numpy(rng.integers(0, nb // 3, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    assert_batches_equal(ctx.hash_join(lt, rt, 0, 0).to_host(), exp, what="with the sorted payload copies")
    monkeypatch.setenv("NQE_TEST_SORTED_COLS_OOM", "1")
    assert_batches_equal(ctx.hash_join(lt, rt, 0, 0).to_host(), exp, what="without them (allocation failed)")


@pytest.mark.parametrize("order", ["ascending", "runs", "one_descent_in_60", "random"])
def test_join_dense_build_of_ascending_keys_skips_the_partitioning(ctx, monkeypatch, order):
    """the build's min/max pass also counts the rows whose key is below its predecessor's: with fewer than one in 64 (an ascending
    primary key, a few sorted runs) the scatter / finish form is coalesced at any size and the partitioning is skipped; one
    descent in 60 rows, or random order, partitions"""
    monkeypatch.setenv("NQE_JOIN_PART_BUILD_MIN", "1000")
    rng = np.random.default_rng(len(order))
    nb, n = 120_000, 200_000
    dk = np.arange(nb, dtype=np.int64) + 11
    if order == "runs":
        dk = np.concatenate([dk[2::3], dk[0::3], dk[1::3]])
    elif order == "one_descent_in_60":
        dk = dk.reshape(-1, 60).copy()
        dk[:, [0, 59]] = dk[:, [59, 0]]                                             # swap first and last of every 60: two descents per 60 rows
        dk = dk.reshape(-1)
    elif order == "random":
        dk = rng.permutation(dk)
    left = [Column.from_numpy(dk), Column.from_numpy(rng.integers(0, 1 << 33, nb).astype(np.int64)), Column.from_numpy(rng.random(nb))]
    right = [Column.from_numpy(rng.integers(0, nb + 30, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    ctx.timing_enable(True); ctx.timing_reset()
    got = ctx.hash_join(lt, rt, 0, 0).to_host()
    took = ctx.timing_query("join_build_part_scatter")[1]
    ctx.timing_enable(False)
    assert (took == 0) == (order in ("ascending", "runs")), (order, took)
    assert_batches_equal(got, exp, what=f"{order} build keys")


@pytest.mark.parametrize("nb", [4097, 6145, 200_003])
@pytest.mark.parametrize("keys", ["narrow", "wide_signed"])
def test_join_sort_based_build_order_of_duplicates(ctx, nb, keys):
    """duplicate build keys go through the stable LSD radix sort (sort.hip: 2048-key tiles sorted by digit in LDS, four waves, eight
    steps each): the matches of a probe row must come out in ascending build row after two (narrow keys) to six (keys over 2^41,
    both signs) digit passes, with a last tile that is not full and one that holds a single key"""
    rng = np.random.default_rng(nb)
    distinct = max(3, nb // 4)
    pool = rng.integers(0, 50_000, distinct) if keys == "narrow" else rng.integers(-(1 << 40), 1 << 40, distinct)
    lk = pool[rng.integers(0, distinct, nb)].astype(np.int64)
    left = [Column.from_numpy(lk), Column.from_numpy(np.arange(nb, dtype=np.int64)), Column.from_numpy(rng.random(nb))]
    rk = np.concatenate([pool[rng.integers(0, distinct, 20_000)], rng.integers(-5, 5, 100)]).astype(np.int64)
    right = [Column.from_numpy(rk), Column.from_numpy(np.arange(rk.size, dtype=np.int64))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    got = ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 0).to_host()
    assert_batches_equal(got, exp, what=f"duplicates, {keys} keys, {nb} build rows")


def test_join_partitioned_dense_build_falls_back_when_its_buffers_do_not_fit(ctx, monkeypatch):
    """the partitioned build's tuple stream and records are extra memory: an allocation failure (NQE_TEST_PART_BUILD_OOM) takes
    the forms that need none instead of failing the build"""
    monkeypatch.setenv("NQE_JOIN_PART_BUILD_MIN", "1000")
    monkeypatch.setenv("NQE_TEST_PART_BUILD_OOM", "1")
    rng = np.random.default_rng(5)
    nb, n = 100_003, 200_000
    left = [Column.from_numpy(rng.permutation(nb).astype(np.int64) + 3), Column.from_numpy(rng.integers(0, 1 << 40, nb).astype(np.int64))]
    right = [Column.from_numpy(rng.integers(0, nb + 10, n).astype(np.int64)), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    ctx.timing_enable(True); ctx.timing_reset()
    got = ctx.hash_join(lt, rt, 0, 0).to_host()
    took = ctx.timing_query("join_build_part_scatter")[1]
    ctx.timing_enable(False)
    assert took == 0
    assert_batches_equal(got, exp, what="partitioned build without memory")


@pytest.mark.parametrize("keys", ["small", "negative", "wide", "one_wide", "uint64_high"])
def test_aggregate_partitioned_path_twelve_byte_tuples_and_their_fallback(ctx, keys):
    """the slab form of the partitioned aggregate moves {int32 key, value} tuples (12 bytes) while every group key fits int32
    (agg_slab_scatter_kernel<…, K32>); a key outside int32 — all of them, a single one, UInt64 keys beyond 2^63 — raises
    NQE_FLAG_KEY32_OVERFLOW and the query is redone with 16-byte tuples, which the plan hint then remembers"""
    rng = np.random.default_rng(len(keys))
    n, groups = 600_000, 70_000
    k = rng.integers(0, groups, n).astype(np.int64)
    if keys == "negative":
        k = k - groups // 2 - (1 << 31) + groups      # down to exactly int32 min
        k[0] = -(1 << 31)
    elif keys == "wide":
        k = k * (1 << 34) - (1 << 50)
    elif keys == "one_wide":
        k[n // 2] = (1 << 31)                        # the first value beyond int32
    kc = Column.from_numpy(k.astype(np.uint64) + np.uint64(1 << 63)) if keys == "uint64_high" else Column.from_numpy(k)
    v = rng.random(n) * 10 - 5
    cols = [kc, Column.from_numpy(v)]
    f2 = fields("k", "v")
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=col(0).flatten(f2))[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=col(0).flatten(f2), with_keys=True)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{keys} rep {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert (kk == np.unique(kc.to_numpy())).all()


@pytest.mark.parametrize("groups", [3, 700, 1500, 3000, 50000])
def test_aggregate_three_value_columns_in_one_pass(ctx, groups):
    """C1's `count(id), sum(age), avg(score) … group by key` — three value columns, no min/max or min/max of the last one only: ONE launch
    of the three-column instance of agg_grouped_fast_kernel (2048-slot workgroup table) while the key and the predicate are ones the kernel
    computes itself and the groups fit; every other shape (a predicate on another column, a key chain, min/max of another column, a nullable column,
    more groups than the table holds — 3000: passes of one and two columns, 50000: the partitioned path) is redone in passes of
    one and two.  n >= 2^18 so that an overfull workgroup table asks for the other path instead of spilling to the global table.
    Second execution: from the plan hint."""
    rng = np.random.default_rng(4242 + groups)
    n = 300_011
    ids = np.arange(n, dtype=np.int64) - n // 7
    k = rng.integers(-(groups // 2), groups - groups // 2, n).astype(np.int64)
    age = rng.integers(-60, 60, n).astype(np.int64)
    score = rng.random(n) * 100.0
    score[::101] = np.nan
    cols = [Column.from_numpy(ids), Column.from_numpy(k), Column.from_numpy(age), Column.from_numpy(score), Column.from_numpy(score, rng.random(n) > 0.2)]
    f5 = fields("id", "k", "age", "score", "score_n")
    t = ctx.table_from_host(cols)
    C1 = lambda kc: [(AggregateFunc.Count, kc), (AggregateFunc.Sum, 2), (AggregateFunc.Avg, 3)]
    keys = [(True, 0, binop(col(0), Operator.Modulos, lit_i64(max(groups // 2, 2)))), (True, 1, col(1)),
            (True, 1, binop(col(1), Operator.Modulos, lit_i64(1 << 20))), (False, 1, binop(binop(col(1), Operator.Plus, lit_i64(5)), Operator.Multiply, lit_i64(3)))]
    for key_ok, kc, key in keys:
        kn = key.flatten(f5)
        shapes = [(True, C1(kc), None), (True, C1(kc), binop(col(kc), Operator.GtEq, lit_i64(-5))), (False, C1(kc), binop(col(2), Operator.Lt, lit_i64(10))),
                  (True, [(AggregateFunc.Avg, 3), (AggregateFunc.Sum, 3), (AggregateFunc.Count, 2), (AggregateFunc.Sum, 0)], None),
                  # min / max asked of the LAST value column only: the three-column instance that carries one pair of min / max arrays — the
                  # reference's own query shape (src/main.rs:36-40: count(id), sum(age), sum(score), avg(score), max(score), min(score))
                  (True, [(AggregateFunc.Count, kc), (AggregateFunc.Sum, 2), (AggregateFunc.Max, 3)], None),
                  (True, [(AggregateFunc.Count, kc), (AggregateFunc.Sum, 2), (AggregateFunc.Sum, 3), (AggregateFunc.Avg, 3), (AggregateFunc.Max, 3), (AggregateFunc.Min, 3)], None),
                  (True, [(AggregateFunc.Count, kc), (AggregateFunc.Sum, 2), (AggregateFunc.Min, 3), (AggregateFunc.Max, 3)], binop(col(kc), Operator.GtEq, lit_i64(-5))),
                  # … of another column: passes of one and two
                  (False, [(AggregateFunc.Count, kc), (AggregateFunc.Max, 2), (AggregateFunc.Sum, 3)], None),
                  (False, [(AggregateFunc.Count, kc), (AggregateFunc.Sum, 2), (AggregateFunc.Avg, 4)], None)]
        for ok, aggs, pred in shapes:
            pn = pred.flatten(f5) if pred is not None else None
            exp = orc.aggregate([cols], aggs, group_nodes=kn, pred_nodes=pn)[0]
            for rep in range(2):
                ctx.timing_enable(True)
                ctx.timing_reset()
                got = ctx.aggregate(t, aggs, group_nodes=kn, pred_nodes=pn).to_host()
                ctx.timing_enable(False)
                rep_ = ctx.timing_report()
                exact = [i for i, (fn, _) in enumerate(aggs) if fn == AggregateFunc.Count]
                assert_rows_multiset_equal(got, exp, RTOL, exact_cols=exact, what=f"groups={groups} key={key!r} aggs={aggs} pred={pred!r}")
                if ok and key_ok and groups <= 700:
                    assert rep_.get("agg_grouped_fast", (0, 0))[1] == 1, (rep_, repr(key), aggs)
                elif groups <= 700:  # passes of one and two columns (one each for a nullable column)
                    assert rep_.get("agg_grouped_fast", (0, 0))[1] == (3 if any(c == 4 for _, c in aggs) else 2), (rep_, repr(key), aggs)


def _packed_home(keys, nb):
    h = ((keys.astype(np.uint64) * np.uint64(GOLD)) >> np.uint64(32)).astype(np.uint64)
    return ((h * np.uint64(nb)) >> np.uint64(32)).astype(np.int64)


@pytest.mark.parametrize("case", ["20bit", "1bit", "63bits", "64bits", "negative_payload", "crowded_bucket", "last_bucket", "overfull"])
def test_hash_join_packed_key_payload_table(ctx, case):
    """sparse unique build keys with ONE integer payload column whose offset fits a word together with the key's offset: the 8-byte
    {key - min, payload - min} slots in 16-slot buckets (hash_join.hip, PackedPairs).  Payload widths 1 / 20 bits, key + payload
    = 63 bits (packed) and 64 bits (the 16-byte form), negative Int64 payloads, probe keys below / above / just outside the build
    range and ones whose offset has the empty word's bit pattern; 40 keys in ONE bucket (the probe's walk beyond a full bucket),
    the same in the table's LAST bucket (the walk wraps to slot 0), and more colliding keys than the insert's bound (fall-back
    to the sort-based build).  Output bit-exact incl. row order."""
    rng = np.random.default_rng(len(case) * 7 + 1)
    nk = 6000
    kbits = 40
    lo = 1 << 44
    pay_lo, pay_hi = 0, 1 << 20
    if case == "1bit":
        pay_hi = 2
    if case in ("63bits", "64bits"):
        pay_hi = 1 << (23 if case == "63bits" else 24)
    if case == "negative_payload":
        pay_lo, pay_hi = -(1 << 19), 1 << 19
    ends = np.array([lo, lo + (1 << kbits) - 1], dtype=np.uint64)  # pin the range: kbits exactly
    keys = np.setdiff1d(np.unique(rng.integers(lo, lo + (1 << kbits), nk).astype(np.uint64)), ends)
    if case in ("crowded_bucket", "last_bucket", "overfull"):
        # keys that share ONE bucket of the table this build will get (the bucket count follows from the final key count)
        want = 300 if case == "overfull" else 40
        cand = np.setdiff1d(np.unique(rng.integers(lo, lo + (1 << kbits), 4_000_000).astype(np.uint64)), np.concatenate([keys, ends]))
        nb = (len(keys) + want + 2) * 5 // 48 + 1
        b = _packed_home(cand, nb)
        target = nb - 1 if case == "last_bucket" else int(np.bincount(b, minlength=nb).argmax())
        pick = cand[b == target][:want]
        assert len(pick) == want
        keys = np.concatenate([keys, pick])
    keys = np.concatenate([keys, ends])
    keys = np.unique(keys)
    rng.shuffle(keys)
    pay = rng.integers(pay_lo, pay_hi, len(keys)).astype(np.int64)
    pay[:2] = [pay_lo, pay_hi - 1]
    left = [Column.from_numpy(keys), Column.from_numpy(pay)]
    pbits = max(1, int(pay_hi - 1 - pay_lo).bit_length())
    empty_pattern = lo + (1 << (64 - pbits)) - 1  # key whose offset, shifted, is all ones in the key field
    specials = [0, 1, lo - 1, lo + (1 << kbits), (1 << 64) - 1, empty_pattern % (1 << 64), int(keys.min()), int(keys.max())]
    probe = np.concatenate([rng.choice(keys, 30000), np.array(specials * 20, dtype=np.uint64), rng.integers(lo, lo + (1 << kbits), 5000).astype(np.uint64),
                            rng.integers(0, 1 << 63, 2000).astype(np.uint64)])
    rng.shuffle(probe)
    right = [Column.from_numpy(probe), Column.from_numpy(rng.random(len(probe)))]
    ctx.timing_enable(True)
    ctx.timing_reset()
    _join_both_ways(ctx, left, right)
    ctx.timing_enable(False)
    assert "join_probe_pairs" in ctx.timing_report() or case == "overfull"


@pytest.mark.parametrize("kind", ["i64_0", "i64_negative", "u64_high", "two_columns", "few", "too_wide"])
def test_aggregate_measured_key_range_addresses_the_table_directly(ctx, kind):
    """`group by k` over a plain integer column: the first execution of the query shape measures the column's min / max
    (agg_key_range) and remembers them; a RANGE that fits a workgroup table (4096 keys; two value columns: 2048) makes the streaming
    kernel address its table by key - min — no probe sequences at high load, no key subsets for the 4096 keys of [min, min + 4096),
    replicas for a handful of groups ("few").  A range that does not fit ("too_wide": every fifth integer) runs hashed as before.
    Then the column's CONTENTS change under the remembered range (a borrowed device buffer overwritten in place): keys outside it
    must be noticed by the kernel, the entry dropped, and the result still equal the oracle's."""
    rng = np.random.default_rng(len(kind))
    n = 400_000
    groups = 2048 if kind == "two_columns" else 5 if kind == "few" else 4096
    base = {"i64_0": 0, "i64_negative": -3000, "u64_high": (1 << 63) + 12345, "two_columns": 10**12, "few": -2, "too_wide": 7}[kind]
    dt = np.uint64 if kind == "u64_high" else np.int64
    k = (rng.integers(0, groups, n) * (5 if kind == "too_wide" else 1) + base).astype(dt) if kind != "u64_high" else \
        (rng.integers(0, groups, n).astype(np.uint64) + np.uint64(base))
    v = rng.random(n) * 100.0
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    aggs = ALL_AGGS(1) + ([(AggregateFunc.Sum, 2), (AggregateFunc.Max, 2)] if kind == "two_columns" else [])
    f3 = fields("k", "v", "w")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # (the runtime the library itself is linked against; torch would bring a second copy into this process)

    def upload(ptr, arr):
        arr = np.ascontiguousarray(arr)
        ctx.synchronize()
        assert hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes), 1) == 0  # hipMemcpyHostToDevice
        assert hip.hipDeviceSynchronize() == 0

    pk, pv, pw = (ctx.device_alloc(n * 8) for _ in range(3))
    upload(pk, k)
    upload(pv, v)
    upload(pw, w)
    kdt = DType.UINT64 if kind == "u64_high" else DType.INT64
    t = ctx.table_from_device([(kdt, n, pk, None), (DType.FLOAT64, n, pv, None), (DType.INT64, n, pw, None)])
    key = col(0).flatten(f3)

    def run():
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, aggs, group_nodes=key).to_host()
        ctx.timing_enable(False)
        return got, ctx.timing_report()

    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    exp = orc.aggregate([cols], aggs, group_nodes=key)[0]
    for rep in range(2):
        got, names = run()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"{kind} run {rep}")
        # (rep 0 measures the range — unless this buffer address still has the previous case's entry, which is then stale and dropped)
        assert rep == 0 or "agg_key_range" not in names, names
        if kind != "too_wide" and rep == 1:
            assert "agg_partition_scatter" not in names and names["agg_grouped_fast"][1] == 1, names
    # the same buffer, other contents: a few keys outside the remembered range
    k2 = k.copy()
    k2[::1000] = k2[::1000] + dt(100_000)
    k2[5::1000] = k2[5::1000] - dt(5000) if kind not in ("i64_0",) else k2[5::1000] + dt(7777)
    upload(pk, k2)
    cols2 = [Column.from_numpy(k2), Column.from_numpy(v), Column.from_numpy(w)]
    exp2 = orc.aggregate([cols2], aggs, group_nodes=key)[0]
    for rep in range(2):
        got, names = run()
        assert_rows_multiset_equal(got, exp2, RTOL, exact_cols=[0], what=f"{kind} after the contents changed, run {rep}")
    del t
    for p_ in (pk, pv, pw):
        ctx.device_free(p_)


@pytest.mark.parametrize("kind", ["i64_0", "i64_negative", "u64_high", "at_limit", "predicate", "int_values", "sparse", "odd_span", "wide_pair", "wide_pair_limit", "wide_pair_predicate"])
def test_aggregate_measured_key_range_over_two_key_subsets(ctx, kind):
    """More groups than one workgroup table holds, at most twice as many (5841 .. 11680 values between the column's min and max — 4097 .. 8192
    where the wide table is not taken), one value
    column: two workgroups share every row range and each keeps ONE HALF OF THE KEY RANGE in a direct-mapped table (AggArgs::direct_sub_width)
    — no hashing, no probing, no partition pass; the tables leave whole and agg_fold_partials_kernel folds them subset by subset.  The
    range comes from the first execution's key sample and is remembered; "sparse" (every third integer: 15000 values) takes the partitioned
    path's range tier.  Then the column's contents change under the remembered range: the kernel must notice, and the result still equal the oracle's."""
    rng = np.random.default_rng(len(kind) + 50)
    # ("predicate": no key sample under a filter — the first execution overflows its tables and asks for subsets, the second measures the
    # column's range; enough rows per workgroup for that overflow)
    n = 4_000_000 if kind in ("predicate", "wide_pair_predicate") else 1_000_000
    groups = {"i64_0": 6000, "i64_negative": 8000, "u64_high": 5000, "at_limit": 8192, "predicate": 7000, "int_values": 4500, "sparse": 5000, "odd_span": 6001, "wide_pair": 11000,
              "wide_pair_limit": 11680, "wide_pair_predicate": 10001}[kind]
    base = {"i64_0": 0, "i64_negative": -5000, "u64_high": (1 << 63) + 999, "at_limit": 10**11, "predicate": 17, "int_values": -1, "sparse": 3, "odd_span": -3000, "wide_pair": 12345,
            "wide_pair_limit": -11679, "wide_pair_predicate": 1}[kind]
    dt = np.uint64 if kind == "u64_high" else np.int64
    draw = rng.integers(0, groups, n)
    if kind in ("at_limit", "wide_pair_limit"):
        draw[:2] = [0, groups - 1]  # the whole range is there
    k = (draw.astype(np.uint64) + np.uint64(base)) if kind == "u64_high" else (draw * (3 if kind == "sparse" else 1) + base).astype(dt)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64) if kind == "int_values" else rng.random(n) * 100.0
    v_dt = DType.INT64 if kind == "int_values" else DType.FLOAT64
    w = rng.random(n)
    aggs = ALL_AGGS(1)
    f3 = fields("k", "v", "w")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def upload(ptr, arr):
        arr = np.ascontiguousarray(arr)
        ctx.synchronize()
        assert hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.nbytes), 1) == 0
        assert hip.hipDeviceSynchronize() == 0

    pk, pv, pw = (ctx.device_alloc(n * 8) for _ in range(3))
    upload(pk, k)
    upload(pv, v)
    upload(pw, w)
    kdt = DType.UINT64 if kind == "u64_high" else DType.INT64
    t = ctx.table_from_device([(kdt, n, pk, None), (v_dt, n, pv, None), (DType.FLOAT64, n, pw, None)])
    key = col(0).flatten(f3)
    pred = binop(col(2), Operator.Lt, lit_f64(0.5)).flatten(f3) if kind in ("predicate", "wide_pair_predicate") else None

    def run():
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred).to_host()
        ctx.timing_enable(False)
        return got, ctx.timing_report()

    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
    for rep in range(3):
        got, names = run()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"{kind} run {rep}")
        if rep >= 1 and kind == "sparse":  # 15000 values: the partitioned path's range tier (two HASHED subsets are slower than that)
            assert "agg_partition_scatter" in names and "agg_grouped_fast" not in names, names
        elif rep >= 1:
            # (NQE_NO_PLAN_HINTS: nothing is remembered — a query under a predicate overflows its single table on every execution first)
            launches = (1, 2) if os.environ.get("NQE_NO_PLAN_HINTS") and kind in ("predicate", "wide_pair_predicate") else (1,)
            assert "agg_partition_scatter" not in names and names["agg_grouped_fast"][1] in launches, names
        if rep == 2 and kind != "sparse":
            assert "agg_fold_partials" in names and "agg_range_emit" in names, names
    # the keys come out in order, and the per-shard states of the multi-GPU path (nqe_aggregate_partial: count, sum, min, max per group) take
    # the same route: merged with themselves they are the single-pass result with doubled counts and sums
    got, gk = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred, with_keys=True)
    kk = gk.to_host()[0].to_numpy()
    assert (np.sort(kk) == kk).all() and len(np.unique(kk)) == len(kk), f"{kind}: keys not in order"
    st, sk = ctx.aggregate_partial(t, aggs, group_nodes=key, pred_nodes=pred)
    assert (sk.to_host()[0].to_numpy() == kk).all()
    merged, mk = ctx.aggregate_merge([st, st], [sk, sk], aggs)
    assert (mk.to_host()[0].to_numpy() == kk).all()
    one, two = got.to_host(), merged.to_host()
    for i, (fn, _) in enumerate(aggs):
        a1, a2 = one[i].to_numpy().astype(np.float64), two[i].to_numpy().astype(np.float64)
        factor = 2.0 if fn in (AggregateFunc.Count, AggregateFunc.Sum) else 1.0
        assert np.allclose(a2, a1 * factor, rtol=1e-9, atol=0.0, equal_nan=True), f"{kind}: merged partial states, aggregate {i}"
    # the same buffer, other contents: a few keys outside the remembered range (above and below)
    k2 = k.copy()
    k2[::1000] = k2[::1000] + dt(100_000)
    k2[5::1000] = k2[5::1000] - dt(3000) if kind not in ("i64_0",) else k2[5::1000] + dt(7777)
    upload(pk, k2)
    cols2 = [Column.from_numpy(k2), Column.from_numpy(v), Column.from_numpy(w)]
    exp2 = orc.aggregate([cols2], aggs, group_nodes=key, pred_nodes=pred)[0]
    for rep in range(2):
        got, names = run()
        assert_rows_multiset_equal(got, exp2, RTOL, exact_cols=[0], what=f"{kind} after the contents changed, run {rep}")
    del t
    for p_ in (pk, pv, pw):
        ctx.device_free(p_)


@pytest.mark.parametrize("kind", ["range_5000", "range_5840_at_limit", "range_5841_two_subsets", "negative_base", "u64_mod_5000", "predicate", "nan_values", "int_values", "value_is_key",
                                  "count_and_sum_only", "count_and_sum_only_13000", "count_and_sum_only_13632_at_limit", "count_and_sum_only_13633_beyond", "count_and_sum_only_u64_mod_12000",
                                  "count_and_sum_only_predicate_9000", "count_and_sum_only_value_is_key_9000", "count_and_sum_only_20000_two_subsets",
                                  "count_and_sum_only_27264_two_subsets_at_limit", "count_and_sum_only_27265_beyond", "count_and_sum_only_int_values_19999_two_subsets"])
def test_aggregate_one_wide_direct_table(ctx, kind, monkeypatch):
    """4097 .. 5840 values between a key column's min and max (or `col % m`, m <= 5840, UInt64), one value column, no validity bitmaps:
    ONE directly addressed workgroup table (round 6 — the table carries no key words: 28 bytes per slot), where two workgroups per row
    range each read every row before; 5841 values still take two key subsets.  Against the oracle on the first, the remembered and a
    third execution, keys in order, and the same with the switch NQE_NO_WIDE_DIRECT=1 (the two-subset form).  count / sum / avg only: 12-byte
    slots — one table up to 13632 keys, and the two halves of a range of up to 27264 keys in two such tables (two key subsets, one launch).
    aggregate/mod.rs:113-222"""
    rng = np.random.default_rng(len(kind) + 600)
    n = 4_000_000 if "predicate" in kind else 1_200_000
    # (count / sum / avg only: the instance without min / max arrays — 12 bytes per slot, one table up to 13632 keys; under a predicate the
    # planner keeps the 28-byte limit, so 9000 keys take the next tier)
    groups = {"range_5840_at_limit": 5840, "range_5841_two_subsets": 5841, "count_and_sum_only_13000": 13000, "count_and_sum_only_13632_at_limit": 13632,
              "count_and_sum_only_13633_beyond": 13633, "count_and_sum_only_predicate_9000": 9000, "count_and_sum_only_value_is_key_9000": 9000,
              "count_and_sum_only_20000_two_subsets": 20000, "count_and_sum_only_27264_two_subsets_at_limit": 27264, "count_and_sum_only_27265_beyond": 27265,
              "count_and_sum_only_int_values_19999_two_subsets": 19999}.get(kind, 5000)
    base = {"negative_base": -2500, "range_5840_at_limit": 10**12, "count_and_sum_only_27264_two_subsets_at_limit": -20000}.get(kind, 0)
    draw = rng.integers(0, groups, n)
    draw[:2] = [0, groups - 1]
    if kind in ("u64_mod_5000", "count_and_sum_only_u64_mod_12000"):
        kc = Column.from_numpy(rng.integers(0, 1 << 62, n).astype(np.uint64))
        key = binop(col(0), Operator.Modulos, lit_u64(5000 if kind == "u64_mod_5000" else 12000))
    else:
        kc = Column.from_numpy((draw + base).astype(np.int64))
        key = col(0)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64) if "int_values" in kind else rng.random(n) * 100.0 - 30.0
    if kind == "nan_values":
        v[rng.integers(0, n, 50)] = np.nan
    cols = [kc, Column.from_numpy(v), Column.from_numpy(rng.random(n))]
    f3 = fields("k", "v", "w")
    kn = key.flatten(f3)
    pn = binop(col(2), Operator.Lt, lit_f64(0.5)).flatten(f3) if kind in ("predicate", "count_and_sum_only_predicate_9000") else None
    # (value_is_key: `sum(k) … group by k` — the single-load instance; count_and_sum_only: no aggregate asks for min / max)
    aggs = (ALL_AGGS(0) if kind == "value_is_key" else [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 0), (AggregateFunc.Avg, 0)] if kind == "count_and_sum_only_value_is_key_9000" else
            [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1)] if kind.startswith("count_and_sum_only") else ALL_AGGS(1))
    exp = orc.aggregate([cols], aggs, group_nodes=kn, pred_nodes=pn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, aggs, group_nodes=kn, pred_nodes=pn, with_keys=True)
        ctx.timing_enable(False)
        names = ctx.timing_report()
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{kind} run {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert len(np.unique(kk)) == len(kk) == got.num_rows
        if rep >= 1 and not os.environ.get("NQE_NO_PLAN_HINTS") and kind != "count_and_sum_only_27265_beyond":
            assert "agg_partition_scatter" not in names and names["agg_grouped_fast"][1] == 1, names
    monkeypatch.setenv("NQE_NO_WIDE_DIRECT", "1")
    monkeypatch.setenv("NQE_NO_PLAN_HINTS", "1")
    got = ctx.aggregate(t, aggs, group_nodes=kn, pred_nodes=pn)
    assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{kind}, NQE_NO_WIDE_DIRECT")


def test_expression_trees_specialised_at_run_time(ctx, monkeypatch):
    """trees of three or more operators are also compiled to straight-line kernels at run time (csrc/expr_jit.hpp: hipRTC on a worker
    thread; executions switch to the compiled form once it is ready).  Every tree here runs interpreted first, then — after
    nqe_ctx_jit_wait — specialised (an `expr_jit` launch, no `expr_tree`), and both results must be bit-identical and equal to the
    oracle's: integer wrap-around, literal and column divisors (DivideByZero and MIN / -1 must still raise), Float64 division and
    fmod, compares of every type, Boolean results, Kleene and/or over nullable columns, a NULL literal, a Boolean input column.
    A second literal value reuses the compiled kernel (literals other than baked divisors are arguments)."""
    monkeypatch.setenv("NQE_JIT_MIN_ROWS", "1000")
    rng = np.random.default_rng(2024)
    n = 70_001
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    b = rng.integers(1, 50, n).astype(np.int64) * rng.choice([-1, 1], n)
    u = rng.integers(0, 1 << 63, n).astype(np.uint64)
    v = rng.random(n) * 200.0 - 100.0
    v[::97] = np.nan
    v[::89] = 0.0
    bo = rng.random(n) < 0.5
    cols_plain = [Column.from_numpy(a), Column.from_numpy(b), Column.from_numpy(u), Column.from_numpy(v), Column.from_numpy(bo)]
    cols_null = [Column.from_numpy(a, rng.random(n) > 0.1), Column.from_numpy(b, rng.random(n) > 0.1), Column.from_numpy(u), Column.from_numpy(v, rng.random(n) > 0.2),
                 Column.from_numpy(bo, rng.random(n) > 0.3)]
    f5 = fields("a", "b", "u", "v", "bo")
    A, B, U, V, BO = (col(i) for i in range(5))
    O = Operator
    X = binop
    trees = [
        X(X(X(X(A, O.Plus, lit_i64(1)), O.Plus, lit_i64(2)), O.Multiply, lit_i64(3)), O.Minus, lit_i64(7)),
        X(X(X(A, O.Modulos, lit_i64(1000)), O.Multiply, lit_i64(3)), O.Plus, X(A, O.Divide, lit_i64(7))),
        X(X(X(A, O.Divide, lit_i64(-8)), O.Plus, X(A, O.Modulos, lit_i64(-16))), O.Minus, X(lit_i64(100), O.Minus, B)),
        X(X(X(A, O.Plus, B), O.Divide, B), O.Plus, X(A, O.Modulos, B)),                       # column divisors (never zero here)
        X(X(X(U, O.Divide, lit_u64(10)), O.Plus, X(U, O.Modulos, lit_u64(1 << 20))), O.Multiply, lit_u64(3)),
        X(X(X(V, O.Multiply, V), O.Plus, X(V, O.Divide, lit_f64(4.0))), O.Minus, X(V, O.Modulos, lit_f64(7.5))),
        X(X(X(V, O.Gt, lit_f64(50.0)), O.And, X(X(A, O.Modulos, lit_i64(3)), O.Eq, lit_i64(0))), O.Or, BO),
        X(X(X(A, O.Lt, B), O.Or, X(U, O.GtEq, lit_u64(1 << 62))), O.And, X(X(V, O.NotEq, V), O.Or, X(V, O.LtEq, lit_f64(0.0)))),
        X(X(X(A, O.Plus, lit_i64(5)), O.Multiply, B), O.Gt, X(B, O.Multiply, lit_i64(1000))),
    ]
    faulting = [
        X(X(X(A, O.Plus, lit_i64(1)), O.Divide, X(B, O.Minus, B)), O.Plus, lit_i64(1)),      # DivideByZero
        X(X(X(A, O.Minus, A), O.Plus, lit_i64(-(1 << 63))), O.Divide, lit_i64(-1)),         # i64::MIN / -1
        X(X(X(V, O.Plus, lit_f64(1.0)), O.Divide, X(V, O.Minus, V)), O.Plus, lit_f64(1.0)),  # Float64 zero divisor (NaN - NaN is not zero; 0 - 0 is)
    ]
    for cols in (cols_plain, cols_null):
        t = ctx.table_from_host(cols)
        for tree in trees + faulting:
            nodes = tree.flatten(f5)
            try:
                exp = orc.expr_evaluate([cols], nodes)
                err = None
            except ErrorCode as e:
                exp, err = None, e
            results = []
            for phase in ("interpreted", "specialised"):
                ctx.timing_enable(True)
                ctx.timing_reset()
                try:
                    got = ctx.expr_evaluate(t, nodes).to_host()[0]
                    gerr = None
                except ErrorCode as e:
                    got, gerr = None, e
                ctx.timing_enable(False)
                names = ctx.timing_report()
                if phase == "interpreted":
                    ctx.jit_wait()
                elif gerr is None or "expr_jit" in names:
                    assert "expr_jit" in names and "expr_tree" not in names, (names, repr(tree))
                results.append((got, gerr))
            (g0, e0), (g1, e1) = results
            assert (e0 is None) == (e1 is None), repr(tree)
            if e0 is not None:
                assert e0.status == e1.status and (tree in faulting), repr(tree)
                if err is not None:
                    assert err.status == e0.status
                continue
            assert_column_equal(g1, g0, what=f"specialised vs interpreted {tree!r}")
            if exp is not None:
                assert_column_equal(g1, exp, what=f"specialised vs oracle {tree!r}")
    # another literal, same shape: no new compilation (the kernel is ready at once), still right
    t = ctx.table_from_host(cols_plain)
    for k in (11, 12345):
        tree = X(X(X(X(A, O.Plus, lit_i64(k)), O.Plus, lit_i64(2)), O.Multiply, lit_i64(3)), O.Minus, lit_i64(7))
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.expr_evaluate(t, tree.flatten(f5)).to_host()[0]
        ctx.timing_enable(False)
        assert "expr_jit" in ctx.timing_report()
        assert (got.to_numpy() == ((a + np.int64(k) + 2) * 3 - 7)).all()


@pytest.mark.parametrize("null_frac", [0.0, 0.15])
def test_projection_list_behind_a_selection_specialised_at_run_time(ctx, null_frac, monkeypatch):
    """the whole projection list behind a selection in ONE run-time specialised kernel (expr_jit.hpp, nqe_jit_proj): bare columns,
    chains, trees and Boolean outputs in one pass over the kept rows, each referenced column read once.  First execution: the
    per-expression kernels; after nqe_ctx_jit_wait: a `proj_jit` launch and no compact_* / expr_tree* — both equal to the oracle's
    projection of the oracle's selection, bit for bit, validity included.  Nullable columns, a nullable predicate column (a NULL
    predicate emits an all-NULL row, literals stay valid under Kleene or), divisors that are zero only in dropped rows (must not
    raise) and in kept rows (must raise from both forms), lists the kernel does not take (a Boolean bare column, five outputs)."""
    monkeypatch.setenv("NQE_JIT_MIN_ROWS", "1000")
    rng = np.random.default_rng(5 + int(null_frac * 100))
    n = 50_003
    cols = random_batch(rng, n, null_frac, key_mod=7, with_bool=True)   # id, k in 0..6, v, u, b
    t = ctx.table_from_host(cols)
    ID, K, V, U, B = (col(i) for i in range(5))
    O = Operator
    X = binop
    lists = [
        [X(X(V, O.Multiply, V), O.Plus, X(V, O.Divide, lit_f64(4.0))), ID],
        [X(ID, O.Plus, lit_i64(100)), X(X(ID, O.Modulos, lit_i64(1000)), O.Multiply, lit_i64(3)), X(V, O.Lt, lit_f64(0.0)), U],
        [X(X(ID, O.Divide, K), O.Plus, ID), X(X(V, O.Gt, lit_f64(10.0)), O.Or, X(ID, O.Lt, K))],          # k == 0 rows are dropped by the predicates that allow this list
        [X(X(X(U, O.Divide, lit_u64(7)), O.Plus, U), O.Modulos, lit_u64(1 << 20))],
        [X(X(B, O.And, X(V, O.Lt, lit_f64(0.0))), O.Or, lit_bool(True)), X(X(B, O.Or, X(K, O.Eq, lit_i64(3))), O.And, X(V, O.GtEq, V))],
    ]
    not_taken = [[B, ID], [ID, K, V, U, X(ID, O.Plus, lit_i64(1))], [ID, K, V], [X(ID, O.Plus, lit_i64(100)), V]]  # Boolean bare column, five outputs, nothing to specialise
    preds = [X(K, O.NotEq, lit_i64(0)), X(X(X(ID, O.Plus, lit_i64(1)), O.Modulos, lit_i64(10)), O.Lt, X(K, O.Plus, lit_i64(1))),
             X(X(K, O.Gt, lit_i64(0)), O.And, X(V, O.Lt, lit_f64(50.0))), X(B, O.And, X(K, O.GtEq, lit_i64(1)))]
    for p in preds:
        sel = orc.selection([cols], flat(p))
        for exprs in lists + not_taken:
            nodes = [flat(e) for e in exprs]
            try:
                exp = orc.projection(sel, nodes)[0]
            except ErrorCode as oe:   # a zero divisor in a kept row: both forms must raise the same
                for phase in (0, 1):
                    with pytest.raises(ErrorCode) as ge:
                        ctx.selection_projection(t, flat(p), nodes)
                    assert ge.value.status == oe.status
                    ctx.jit_wait()
                continue
            for phase in (0, 1):
                ctx.timing_enable(True)
                ctx.timing_reset()
                got = ctx.selection_projection(t, flat(p), nodes).to_host()
                ctx.timing_enable(False)
                names = ctx.timing_report()
                assert_batches_equal(got, exp, what=f"phase {phase} pred {p!r} list {exprs!r}")
                if phase == 0:
                    ctx.jit_wait()
                elif exprs in not_taken:
                    assert "proj_jit" not in names, names
                else:
                    # (a tree predicate over inputs without NULLs and word-typed outputs: predicate, compaction and list in ONE kernel)
                    assert ("proj_jit" in names or "select_project_jit" in names) and not any(k.startswith(("compact", "expr_tree_compact")) for k in names), (names, repr(exprs))
    # a divisor that is zero in KEPT rows raises from both forms
    bad = X(K, O.LtEq, lit_i64(2))
    nodes = [flat(e) for e in (X(X(ID, O.Divide, K), O.Plus, ID), X(X(V, O.Gt, lit_f64(10.0)), O.Or, X(ID, O.Lt, K)))]
    for phase in (0, 1):
        with pytest.raises(ErrorCode) as ge:
            ctx.selection_projection(t, flat(bad), nodes)
        assert ge.value.status == Status.ArrowError
        ctx.jit_wait()


@pytest.mark.parametrize("n", [1 << 12, 70_001, 512 * 300 + 7])
@pytest.mark.parametrize("keep", ["first_half", "spread", "nothing", "everything", "one_row", "last_row"])
def test_selection_and_projection_in_one_specialised_pass(ctx, monkeypatch, n, keep):
    """a tree predicate + a projection list over inputs without NULLs: ONE run-time specialised kernel (expr_jit.hpp, nqe_jit_selproj)
    evaluates the predicate, compacts by decoupled look-back over 512-row chunks and writes the projected kept rows — stable order,
    every column read once, no mask, no scan.  After nqe_ctx_jit_wait the step is one `select_project_jit` launch; results equal the
    oracle's (and the two-kernel form's, NQE_NO_FUSED_SELECT) bit for bit for kept rows at the front, spread out, none, all, a single
    row and the very last row, at sizes around the chunk size; a zero divisor only in dropped rows does not raise, one in a kept row does."""
    monkeypatch.setenv("NQE_JIT_MIN_ROWS", "1000")
    rng = np.random.default_rng(n + len(keep))
    ids = np.arange(n, dtype=np.int64)
    v = rng.random(n) * 100.0
    k = rng.integers(0, 5, n).astype(np.int64)
    cols = [Column.from_numpy(ids), Column.from_numpy(v), Column.from_numpy(k)]
    f3 = fields("id", "v", "k")
    ID, V, K = col(0), col(1), col(2)
    O, X = Operator, binop
    pred = {"first_half": X(X(X(ID, O.Plus, lit_i64(0)), O.Multiply, lit_i64(2)), O.Lt, lit_i64(n)),
            "spread": X(X(X(ID, O.Plus, lit_i64(1)), O.Modulos, lit_i64(10)), O.Lt, lit_i64(5)),
            "nothing": X(X(X(ID, O.Plus, lit_i64(1)), O.Multiply, lit_i64(1)), O.Lt, lit_i64(0)),
            "everything": X(X(X(ID, O.Plus, lit_i64(1)), O.Multiply, lit_i64(1)), O.Gt, lit_i64(0)),
            "one_row": X(X(X(ID, O.Plus, lit_i64(1)), O.Multiply, lit_i64(3)), O.Eq, lit_i64(3 * (n // 3 + 1))),
            "last_row": X(X(X(ID, O.Plus, lit_i64(1)), O.Multiply, lit_i64(1)), O.GtEq, lit_i64(n))}[keep]
    exprs = [X(X(V, O.Multiply, V), O.Plus, X(V, O.Divide, lit_f64(4.0))), ID, X(X(ID, O.Modulos, lit_i64(1000)), O.Multiply, lit_i64(3))]
    pn, en = pred.flatten(f3), [e.flatten(f3) for e in exprs]
    exp = orc.projection(orc.selection([cols], pn), en)[0]
    t = ctx.table_from_host(cols)
    for phase in (0, 1, 2):
        if phase == 2:
            monkeypatch.setenv("NQE_NO_FUSED_SELECT", "1")
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.selection_projection(t, pn, en).to_host()
        ctx.timing_enable(False)
        names = ctx.timing_report()
        assert_batches_equal(got, exp, what=f"phase {phase} keep={keep} n={n}")
        if phase == 0:
            ctx.jit_wait()
        elif phase == 1:
            assert names.get("select_project_jit", (0, 0))[1] == 1 and not any(x.startswith(("keep_from", "compact", "proj_jit", "expr_jit")) for x in names), names
        else:
            assert "select_project_jit" not in names, names
    monkeypatch.delenv("NQE_NO_FUSED_SELECT")
    if keep == "spread":
        # k == 0 rows are dropped by the predicate: `id / k` must not raise; with them kept it must (from the fused kernel too)
        safe = X(X(X(K, O.Plus, lit_i64(0)), O.Multiply, lit_i64(1)), O.Gt, lit_i64(0))
        div = [X(X(ID, O.Divide, K), O.Plus, ID).flatten(f3)]
        exp2 = orc.projection(orc.selection([cols], safe.flatten(f3)), div)[0]
        for phase in (0, 1):
            assert_batches_equal(ctx.selection_projection(t, safe.flatten(f3), div).to_host(), exp2, what=f"divisor zero in dropped rows only, phase {phase}")
            ctx.jit_wait()
        unsafe = X(X(X(K, O.Plus, lit_i64(0)), O.Multiply, lit_i64(1)), O.GtEq, lit_i64(0))
        for phase in (0, 1):
            with pytest.raises(ErrorCode) as ge:
                ctx.selection_projection(t, unsafe.flatten(f3), div)
            assert ge.value.status == Status.ArrowError
            ctx.jit_wait()


def test_reserved_block_serves_allocations_and_falls_back(monkeypatch):
    """nqe_ctx_reserve: a context with a reserved block sub-allocates operator outputs and scratch from it (best fit, freed ranges
    coalesced) and falls back to the pool / the driver for what does not fit — same results either way, the free part is reported
    as pooled bytes, and released tables give their ranges back."""
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    try:
        c.reserve(48 << 20)
        with pytest.raises(ErrorCode):
            c.reserve(1 << 20)                       # once per context
        live0, pooled0 = c.memory_stats()
        assert pooled0 >= 48 << 20
        rng = np.random.default_rng(77)
        n = 400_003
        cols = random_batch(rng, n, 0.0, key_mod=1000)
        t = c.table_from_host(cols)                  # 4 x 3.2 MB: from the block
        live1, pooled1 = c.memory_stats()
        assert pooled1 < pooled0 and live1 > live0
        f = FLD[:4]
        pred = binop(col(1), Operator.Lt, lit_i64(500)).flatten(f)
        exp = orc.selection([cols], pred)[0]
        for _ in range(3):
            assert_batches_equal(c.selection(t, pred).to_host(), exp, what="selection from the reserved block")
        aggs = ALL_AGGS(2)
        expa = orc.aggregate([cols], aggs, group_nodes=col(1).flatten(f))[0]
        assert_rows_multiset_equal(c.aggregate(t, aggs, group_nodes=col(1).flatten(f)).to_host(), expa, RTOL, exact_cols=[0], what="aggregate")
        # more than the block holds: the driver serves it, results unchanged
        big = [Column.from_numpy(rng.integers(0, 1 << 40, 3_000_000).astype(np.int64)) for _ in range(3)]   # 72 MB
        tb = c.table_from_host(big)
        got = c.projection(tb, [binop(col(0), Operator.Plus, col(1)).flatten(fields("a", "b", "c"))]).to_host()[0]
        assert (got.to_numpy() == big[0].to_numpy() + big[1].to_numpy()).all()
        del t, tb, got
        import gc
        gc.collect()
        live2, pooled2 = c.memory_stats()
        assert live2 <= live0 + (1 << 20) and pooled2 >= pooled0   # everything came back (the driver's blocks to the pool)
    finally:
        c.close()


def test_specialised_kernels_are_found_on_disk_by_a_new_context(tmp_path, monkeypatch):
    """the code object hipRTC produces for a tree is kept under NQE_JIT_CACHE_DIR (source hash + ISA + runtime version; the generated
    source is stored with it and compared on load): a context that has never seen the tree — a new process, in production — takes the
    specialised kernel on its FIRST execution instead of interpreting until a compilation finishes.  A file whose source differs is
    ignored (recompiled), NQE_NO_JIT_DISK_CACHE switches the cache off."""
    from naive_query_engine_amd import capi

    monkeypatch.setenv("NQE_JIT_MIN_ROWS", "1000")
    monkeypatch.setenv("NQE_JIT_CACHE_DIR", str(tmp_path / "jit"))
    rng = np.random.default_rng(31)
    n = 40_000
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    cols = [Column.from_numpy(a)]
    O, X = Operator, binop
    tree = X(X(X(col(0), O.Modulos, lit_i64(977)), O.Multiply, lit_i64(3)), O.Plus, X(col(0), O.Divide, lit_i64(7)))
    nodes = tree.flatten(fields("a"))
    exp = orc.expr_evaluate([cols], nodes)

    def run(ctx):
        t = ctx.table_from_host(cols)
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.expr_evaluate(t, nodes).to_host()[0]
        ctx.timing_enable(False)
        assert_column_equal(got, exp, what="tree")
        return set(ctx.timing_report())

    c1 = capi.Context(0)
    try:
        assert "expr_jit" not in run(c1)          # nothing on disk yet: interpreted, compilation started
        c1.jit_wait()
        assert "expr_jit" in run(c1)
    finally:
        c1.close()
    files = list((tmp_path / "jit").glob("*.nqejit"))
    assert len(files) == 1 and files[0].stat().st_size > 1000
    c2 = capi.Context(0)
    try:
        assert "expr_jit" in run(c2)              # first execution of a fresh context: from the file
    finally:
        c2.close()
    # a file that does not hold this source is not used
    blob = bytearray(files[0].read_bytes())
    blob[24 + 10] ^= 0x20                         # one character of the stored source
    files[0].write_bytes(bytes(blob))
    c3 = capi.Context(0)
    try:
        assert "expr_jit" not in run(c3)
        c3.jit_wait()
        assert "expr_jit" in run(c3)
    finally:
        c3.close()
    monkeypatch.setenv("NQE_NO_JIT_DISK_CACHE", "1")
    c4 = capi.Context(0)
    try:
        assert "expr_jit" not in run(c4)
    finally:
        c4.jit_wait()
        c4.close()


@pytest.mark.parametrize("shape", ["few", "dense_4096", "dense_6000", "spread_6000", "sparse_3000", "sparse_6000", "sparse_20000", "groups_300000", "all_distinct",
                                   "mod_100000", "range_with_outliers", "skewed"])
@pytest.mark.parametrize("no_hints", [False, True])
def test_aggregate_first_execution_starts_in_the_tier_its_key_sample_picks(ctx, monkeypatch, shape, no_hints):
    """tables of 2^22 rows and more without a predicate: the FIRST execution samples 65536 keys (key_sample_kernel) and starts in the
    tier their distinct count calls for — one workgroup table (a range that fits: addressed by key - min), two key subsets (direct-mapped
    when the sampled range fits two tables; the partitioned path's range tier instead of HASHED subsets when the range fits that), the
    partitioned path (256 or 512 partitions) — instead of falling through abandoned tiers; later executions take the remembered
    tier (or, NQE_NO_PLAN_HINTS, sample again).  Results equal the oracle's in every case, including the ones the sample gets wrong:
    a plain key column whose few outliers the sample misses (the kernel's range check asks for the exact measurement), heavy skew
    (the sample's distinct count is far below the table's)."""
    if no_hints:
        monkeypatch.setenv("NQE_NO_PLAN_HINTS", "1")
    rng = np.random.default_rng(len(shape) * 7 + 1)
    # (a row count of its own per case: what the context remembers is keyed by buffer, rows and query shape, and the pool hands the
    # same buffers to consecutive cases)
    n = (1 << 22) + 12_345 + 64 * (sum(map(ord, shape)) + (1 if no_hints else 0))
    key_expr = col(0)
    if shape == "few":
        k = rng.integers(-3, 4, n)
    elif shape == "dense_4096":
        k = rng.integers(100, 4196, n)
    elif shape == "dense_6000":                                     # two key subsets over a direct-mapped table (round 5)
        k = rng.integers(-2000, 4000, n)
    elif shape == "spread_6000":                                    # 6000 keys over a range of 42000: the partitioned path's range tier, not two hashed subsets
        k = rng.integers(0, 6000, n) * 7 + 11
    elif shape.startswith("sparse_"):
        g = int(shape.split("_")[1])
        k = rng.integers(0, g, n) * 1_000_003 - 5
    elif shape == "groups_300000":
        k = rng.integers(0, 300_000, n) * 7
    elif shape == "all_distinct":
        k = rng.permutation(n).astype(np.int64) * 3 - n
    elif shape == "mod_100000":
        k = rng.integers(0, 1 << 40, n)
        key_expr = binop(col(0), Operator.Modulos, lit_i64(100_000))
    elif shape == "range_with_outliers":
        k = rng.integers(0, 1000, n)
        k[[5, n // 2, n - 3]] = [10**12, -7, 4095 + 10**6]         # three rows the sample will not see
    else:                                                           # skewed: 99.9 % of the rows in 50 keys, the rest in 200 000
        k = np.where(rng.random(n) < 0.999, rng.integers(0, 50, n), rng.integers(1000, 201_000, n))
    k = k.astype(np.int64)
    v = rng.random(n) * 100.0
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    kn = key_expr.flatten(f2)
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn).to_host()
        ctx.timing_enable(False)
        names = ctx.timing_report()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"{shape} rep {rep} no_hints={no_hints}")
        sampled = "agg_key_sample" in names
        assert sampled == (rep == 0 or no_hints or bool(os.environ.get("NQE_NO_PLAN_HINTS"))), (shape, rep, no_hints, sorted(names))
        if rep == 0 and shape == "dense_6000":
            assert names.get("agg_grouped_fast", (0, 0))[1] == 1 and "agg_range_emit" in names and "agg_partition_scatter" not in names, sorted(names)
        if rep == 0 and shape in ("sparse_20000", "groups_300000", "all_distinct", "mod_100000", "spread_6000"):
            # straight to the partitioned path: no abandoned streaming attempt before it
            assert "agg_partition_scatter" in names and "agg_grouped_fast" not in names, sorted(names)
        if rep == 0 and shape in ("few", "dense_4096", "sparse_3000"):
            assert names.get("agg_grouped_fast", (0, 0))[1] == 1 and "agg_key_range" not in names, sorted(names)
        if rep == 0 and shape == "range_with_outliers":
            assert "agg_key_range" in names                        # the sampled range was too narrow: measured exactly, once


@pytest.mark.parametrize("n", [70_001, 4096 * 3])
def test_aggregate_under_a_predicate_tree_through_the_specialised_streaming_kernel(ctx, monkeypatch, n):
    """predicate trees the static streaming kernel interprets (PRED 5 / 6), grouped by `col % m` with a direct-mapped table of 512 to
    4096 slots, one value column, no NULLs: a lean run-time specialised kernel (expr_jit.hpp: nqe_jit_agg — the predicate as
    straight-line code, run cache, native LDS f64 atomics) whose per-workgroup tables agg_merge_partials_kernel folds into the group
    table.  First execution: the interpreters; after nqe_ctx_jit_wait: one `agg_grouped_jit` + one `agg_merge_partials` launch and no
    `agg_grouped_fast`; NQE_NO_AGG_JIT: the interpreters again — all three equal to the oracle (counts exact, Float64 1e-9).  Signed
    keys on both sides of zero, unsigned keys, Float64 values with NaNs and +-0, Int64 / UInt64 values, the value column being the key
    column, predicates over a third column; shapes that do not qualify (few slots, a plain key, a nullable value) keep the static kernel."""
    monkeypatch.setenv("NQE_JIT_MIN_ROWS", "1000")
    rng = np.random.default_rng(n)
    ids = np.arange(n, dtype=np.int64) - n // 3
    u = rng.integers(0, 1 << 50, n).astype(np.uint64)
    v = rng.random(n) * 100.0
    v[::97] = np.nan
    v[::89] = -0.0
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    cols = [Column.from_numpy(ids), Column.from_numpy(u), Column.from_numpy(v), Column.from_numpy(w), Column.from_numpy(v, rng.random(n) > 0.1)]
    f5 = fields("id", "u", "v", "w", "vn")
    ID, U, V, W = col(0), col(1), col(2), col(3)
    O, X = Operator, binop
    trees = [X(X(V, O.Lt, lit_f64(20.0)), O.Or, X(X(ID, O.Modulos, lit_i64(3)), O.Eq, lit_i64(0))),
             X(X(X(W, O.Plus, lit_i64(7)), O.Multiply, lit_i64(3)), O.Gt, X(ID, O.Modulos, lit_i64(50))),
             X(X(X(V, O.Multiply, V), O.Lt, lit_f64(2500.0)), O.And, X(X(W, O.Lt, lit_i64(500)), O.Or, X(ID, O.GtEq, lit_i64(0)))),
             # chain predicates (`col op lit ... cmp lit`: the static kernel's operator-major interpreter, PRED 3)
             X(X(X(W, O.Plus, lit_i64(7)), O.Modulos, lit_i64(10)), O.Lt, lit_i64(6)), X(X(ID, O.Multiply, lit_i64(3)), O.GtEq, lit_i64(300))]
    # (key expression, value column, qualifies)
    # (None: whether the static path hands the tree to its in-kernel interpreters — the specialised kernel's hook — or materialises it
    # depends on the predicate / key / value combination: results are checked, the kernel taken is not)
    shapes = [(X(ID, O.Modulos, lit_i64(1024)), 2, True), (X(ID, O.Modulos, lit_i64(300)), 3, True), (X(U, O.Modulos, lit_u64(4096)), 1, None),
              (X(U, O.Modulos, lit_u64(1000)), 2, None), (X(ID, O.Modulos, lit_i64(-2048)), 0, None),
              (X(ID, O.Modulos, lit_i64(100)), 2, False), (X(U, O.Modulos, lit_u64(5000)), 2, False), (W, 2, False), (X(ID, O.Modulos, lit_i64(1024)), 4, False)]
    t = ctx.table_from_host(cols)
    for ti, tree in enumerate(trees):
        pn = tree.flatten(f5)
        for key, vc, qualifies in shapes:
            # (trees 1 and 2 — a column-with-column compare, a product of columns — are ones the static kernels can only take as a
            # materialised Boolean column: with the specialised kernel compiled they are evaluated in the aggregation pass instead)
            kn = key.flatten(f5)
            exp = orc.aggregate([cols], ALL_AGGS(vc), group_nodes=kn, pred_nodes=pn)[0]
            for phase in (0, 1, 2):
                if phase == 2:
                    monkeypatch.setenv("NQE_NO_AGG_JIT", "1")
                ctx.timing_enable(True)
                ctx.timing_reset()
                got = ctx.aggregate(t, ALL_AGGS(vc), group_nodes=kn, pred_nodes=pn).to_host()
                ctx.timing_enable(False)
                names = ctx.timing_report()
                assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"phase {phase} tree {tree!r} key {key!r} value {vc}")
                if phase == 0:
                    ctx.jit_wait()
                elif phase == 1 and qualifies:
                    assert names.get("agg_grouped_jit", (0, 0))[1] == 1 and names.get("agg_merge_partials", (0, 0))[1] == 1 and "agg_grouped_fast" not in names and \
                        not any(x.startswith(("expr_jit", "expr_tree", "binary")) for x in names), (sorted(names), repr(key), ti)
                elif phase == 2 or qualifies is False:
                    assert "agg_grouped_jit" not in names, (sorted(names), repr(key), phase)
            monkeypatch.delenv("NQE_NO_AGG_JIT")
    # interpreted chain KEYS (`(id + 1) % 1000`, `u / 7 % 4096`: KEY 3 of the static kernel) take the specialised kernel whatever the
    # predicate — none, a plain range test, a tree
    for key in (X(X(ID, O.Plus, lit_i64(1)), O.Modulos, lit_i64(1000)), X(X(U, O.Divide, lit_u64(7)), O.Modulos, lit_u64(4096)),
                X(X(X(ID, O.Multiply, lit_i64(3)), O.Minus, lit_i64(5)), O.Modulos, lit_i64(-700))):
        kn = key.flatten(f5)
        for pred in (None, X(ID, O.Lt, lit_i64(n // 3)), trees[0]):
            pn = pred.flatten(f5) if pred is not None else None
            exp = orc.aggregate([cols], ALL_AGGS(2), group_nodes=kn, pred_nodes=pn)[0]
            for phase in (0, 1):
                ctx.timing_enable(True)
                ctx.timing_reset()
                got = ctx.aggregate(t, ALL_AGGS(2), group_nodes=kn, pred_nodes=pn).to_host()
                ctx.timing_enable(False)
                names = ctx.timing_report()
                assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0], what=f"chain key {key!r} pred {pred!r} phase {phase}")
                if phase == 0:
                    ctx.jit_wait()
                else:
                    assert names.get("agg_grouped_jit", (0, 0))[1] == 1 and "agg_grouped_fast" not in names and "agg_grouped" not in names, (sorted(names), repr(key), repr(pred))
    # two value columns under a tree the static path would materialise: one specialised launch per column, no Boolean column
    tree = trees[1]
    pn, kn = tree.flatten(f5), X(ID, O.Modulos, lit_i64(1024)).flatten(f5)
    aggs2 = ALL_AGGS(2) + [(AggregateFunc.Sum, 3), (AggregateFunc.Max, 3), (AggregateFunc.Count, 3)]
    exp = orc.aggregate([cols], aggs2, group_nodes=kn, pred_nodes=pn)[0]
    for phase in (0, 1):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, aggs2, group_nodes=kn, pred_nodes=pn).to_host()
        ctx.timing_enable(False)
        names = ctx.timing_report()
        assert_rows_multiset_equal(got, exp, RTOL, exact_cols=[0, 7], what=f"two value columns, phase {phase}")
        if phase == 0:
            ctx.jit_wait()
        else:
            assert names.get("agg_grouped_jit", (0, 0))[1] == 2 and not any(x.startswith(("expr_jit", "expr_tree", "agg_grouped_fast")) for x in names), sorted(names)


@pytest.mark.parametrize("shape", ["dense", "gaps", "negative", "uint64", "at_limit", "past_limit", "outlier", "modulus_key"])
def test_aggregate_dense_table_tail_ranks_by_key_range(ctx, shape, monkeypatch):
    """the partitioned path leaves its groups in no order; when their keys lie in a compact range (at most 8 G + 65536 values) the tail
    ranks them by key - min (dense_key_range → dense_rank_mark → dense_rank_emit: no sort, one host wait) — keys and aggregates must
    come out exactly as the radix-sort tail (NQE_NO_RANGE_TAIL=1) and the oracle give them, whole and as partial states; a wider range
    (one outlying key, keys with a large common factor) keeps the sort.  aggregate/mod.rs:113-222 (the output order is the map's:
    unobservable; this build emits key order)"""
    rng = np.random.default_rng(len(shape) * 17)
    n, G = 600_000, 70_000
    base = rng.integers(0, G, n).astype(np.int64)
    expect_range = True
    if shape == "dense":
        k = base
    elif shape == "gaps":
        k = base * 5 - 1234                      # span 5 G: inside 8 G + 65536
    elif shape == "negative":
        k = base - G                             # every key negative, -1 (the all-ones word) among them
    elif shape == "uint64":
        k = base
    elif shape == "at_limit":
        k = base.copy()
        k[0] = 8 * len(np.unique(base[1:])) + 65536 - 2   # max - min just inside the limit whatever the distinct count turns out to be
    elif shape == "past_limit":
        k = base * 40                            # span 40 G
        expect_range = False
    elif shape == "outlier":
        k = base.copy()
        k[n // 3] = 1 << 45
        expect_range = False
    else:
        k = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
    kc = Column.from_numpy(k.astype(np.uint64) + np.uint64(1 << 63)) if shape == "uint64" else Column.from_numpy(k)
    v = rng.random(n) * 100 - 50
    v[7] = np.nan
    cols = [kc, Column.from_numpy(v)]
    f2 = fields("k", "v")
    key = binop(col(0), Operator.Modulos, lit_i64(77_777)) if shape == "modulus_key" else col(0)
    kn = key.flatten(f2)
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn)[0]
    t = ctx.table_from_host(cols)
    results = []
    # "tier": as the library runs it (round 5: key-range partitions end in the range tier's transposing tail, agg_range_emit);
    # "range" / "sort": hashed partitions (NQE_NO_RANGE_PARTITION), whose densely written table is ranked by key - min or sorted
    for tail in ("tier", "range", "sort"):
        if tail != "tier":
            monkeypatch.setenv("NQE_NO_RANGE_PARTITION", "1")
        if tail == "sort":
            monkeypatch.setenv("NQE_NO_RANGE_TAIL", "1")
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn, with_keys=True)
        st, sk = ctx.aggregate_partial(t, ALL_AGGS(1), group_nodes=kn)
        ctx.timing_enable(False)
        ranked = ctx.timing_query("agg_dense_rank_emit")[1]
        assert ctx.timing_query("agg_segments")[1] > 0, "the partitioned path was expected"
        if tail != "tier":  # ("tier": a sampled range may miss a key — one call then goes through hashed partitions, the next through the range tier)
            assert (ranked > 0) == (expect_range and tail == "range"), f"{shape} {tail}: {ranked} ranked tails"
        kk = gk.to_host()[0].to_numpy()
        order = np.argsort(kk, kind="stable")
        assert (order == np.arange(len(kk))).all(), f"{shape} {tail}: keys not in order"
        assert (sk.to_host()[0].to_numpy() == kk).all()
        h = got.to_host()
        assert_rows_multiset_equal(h, exp, RTOL, exact_cols=[0], what=f"{shape} {tail}")
        results.append((kk, [c.to_numpy() for c in h], [c.to_numpy() for c in st.to_host()]))
    (ka, ha, sa) = results[0]
    for (kb, hb, sb) in results[1:]:
        assert (ka == kb).all()
        for x, y in zip(ha + sa, hb + sb):
            assert np.allclose(x, y, rtol=1e-9, atol=0, equal_nan=True)   # f64 sums: LDS atomics in no fixed order


@pytest.mark.parametrize("shape", ["dense", "negative", "gaps", "wide_512_parts", "too_wide", "mod_key", "mod_key_mixed_signs", "sorted_ids_mod", "int64_values", "predicate",
                                   "skewed", "uint64_keys"])
def test_aggregate_partitioned_path_range_partitions(ctx, shape, monkeypatch):
    """the slab form of the partitioned aggregate picks a row's partition from its key's position in the KEY RANGE when it knows one of at most 2^21 values —
    the first execution's key sample (tables of 2^22 rows and more without a predicate; `col % m`: what the modulus allows), or the
    exact range the dense tail of an earlier execution measured — and the second kernel then addresses its LDS table by key - base
    (agg_slab_segments_direct_kernel).  Every execution must equal the oracle and the hashed form (NQE_NO_RANGE_PARTITION=1); a range
    that turns out wrong or lopsided (a key the sample missed, half the rows in one key) falls back to hashed partitions.
    aggregate/mod.rs:113-222"""
    rng = np.random.default_rng(len(shape) * 31 + 5)
    n = (1 << 22) + 4321 + 128 * sum(map(ord, shape))     # (a row count per case: remembered plans are keyed by buffer and rows)
    G = 70_000
    base = rng.integers(0, G, n).astype(np.int64)
    key = col(0)
    pred = None
    first_direct, later_direct = True, True
    v = rng.random(n) * 100 - 50
    v[11] = np.nan
    if shape == "dense":
        k = base
    elif shape == "negative":
        k = base - G - 5
    elif shape == "gaps":
        k = base * 5 + 1000
    elif shape == "wide_512_parts":
        k = base * 25                                     # 1.75 M values: 512 tables of 3418 slots
    elif shape == "too_wide":
        k = base * 40                                     # 2.8 M values: hashed partitions
        first_direct = later_direct = False
    elif shape in ("mod_key", "mod_key_mixed_signs"):
        k = rng.integers(0, 1 << 40, n).astype(np.int64)
        key = binop(col(0), Operator.Modulos, lit_i64(60_000))
        if shape == "mod_key_mixed_signs":
            # a thousand negative keys: a sample that sees none takes [0, m) and meets a key outside it (hashed partitions this time, the
            # measured range from then on); one that sees some takes (-m, m)
            k[:1000] -= 1 << 41
            first_direct = None
    elif shape == "sorted_ids_mod":
        # a row number modulo m: consecutive keys must land in different partitions (the low bits pick the partition), or a scatter
        # workgroup's whole chunk goes to a few of them and their slabs overflow
        k = np.arange(n, dtype=np.int64)
        key = binop(col(0), Operator.Modulos, lit_i64(65_536))
    elif shape == "int64_values":
        k = base
        v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    elif shape == "predicate":
        k = base
        pred = binop(col(1), Operator.Gt, lit_f64(-20.0))
        # no key sample under a predicate: the streaming attempt overflows, the key range is measured exactly (one pass) and the range
        # tier cuts its partitions from it in the SAME execution (round 6: ADVICE r05, react_to_flags / NEED_PARTITION)
    elif shape == "skewed":
        k = base.copy()
        k[rng.random(n) < 0.5] = 12_345                   # half the rows in one key: its interval's slabs overflow
        first_direct = later_direct = None
    else:
        k = base + 3
    kc = Column.from_numpy(k.astype(np.uint64)) if shape == "uint64_keys" else Column.from_numpy(k)
    cols = [kc, Column.from_numpy(v)]
    f2 = fields("k", "v")
    kn = key.flatten(f2)
    pn = pred.flatten(f2) if pred is not None else None
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn, pred_nodes=pn)[0]
    t = ctx.table_from_host(cols)
    keys_seen = None
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn, pred_nodes=pn, with_keys=True)
        ctx.timing_enable(False)
        direct = ctx.timing_query("agg_segments_direct")[1] > 0
        want = first_direct if rep == 0 else later_direct
        assert ctx.timing_query("agg_partition_scatter")[1] > 0, f"{shape}: the partitioned path was expected"
        if want is not None:
            assert direct == want, f"{shape} rep {rep}: direct segments kernel {direct}, expected {want}"
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{shape} rep {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert (np.diff(kk.astype(np.int64) if shape != "uint64_keys" else kk.astype(np.uint64).astype(np.float64)) > 0).all(), f"{shape}: keys not in order"
        if keys_seen is not None:
            assert (kk == keys_seen).all()
        keys_seen = kk
    monkeypatch.setenv("NQE_NO_RANGE_PARTITION", "1")
    ctx.timing_enable(True)
    ctx.timing_reset()
    got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn, pred_nodes=pn, with_keys=True)
    ctx.timing_enable(False)
    assert ctx.timing_query("agg_segments_direct")[1] == 0
    assert (gk.to_host()[0].to_numpy() == keys_seen).all()
    assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{shape} hashed")


@pytest.mark.parametrize("sample", [True, False])
def test_aggregate_mid_size_table_with_many_groups_takes_the_partitioned_path(ctx, monkeypatch, sample):
    """half a million rows over 90 000 groups: no workgroup sees more distinct keys than its LDS table holds, so nothing ever asked
    for the partitioned path — the global table was grown instead and every workgroup folded its LDS table into it through
    device-scope atomics (1.3 ms for a 0.13 ms query).  The first execution's key sample (tables of 2^18 rows and more) starts the
    query partitioned; without the sample (NQE_NO_KEY_SAMPLE=1) the overfull global table of the two-subset tier sends it there.
    aggregate/mod.rs:113-222"""
    if not sample:
        monkeypatch.setenv("NQE_NO_KEY_SAMPLE", "1")
    rng = np.random.default_rng(77 + int(sample))
    n, G = 500_000 + 64 * int(sample), 90_000
    k = rng.integers(0, G, n).astype(np.int64) * 3 - 1000
    v = rng.random(n) * 10
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    kn = col(0).flatten(f2)
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn)
        ctx.timing_enable(False)
        assert ctx.timing_query("agg_partition_scatter")[1] > 0, f"rep {rep}: the partitioned path was expected"
        if (rep > 0 and not os.environ.get("NQE_NO_PLAN_HINTS")) or sample:   # (the whole suite also runs under NQE_NO_PLAN_HINTS=1: nothing is remembered then)
            assert ctx.timing_query("agg_grouped_fast")[1] == 0, f"rep {rep}: no streaming attempt was expected"
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"mid-size table, rep {rep}")


@pytest.mark.parametrize("shape", ["keys_only", "count_utf8", "count_bool", "count_utf8_and_values"])
def test_aggregate_many_groups_over_shapes_the_streaming_kernel_does_not_cover(ctx, shape):
    """the first execution's key sample put EVERY query with > 8192 sampled keys on the partitioned path with a
    densely written table — also `select k from t group by k` (no aggregates) and count() over a Utf8 / Boolean column, which only
    the general (hashed) kernel evaluates: NQE_ERR_NOT_SUPPORTED "… handed a densely laid out group table", remembered for every
    later execution.  aggregate/mod.rs:113-222, count.rs:33-82"""
    rng = np.random.default_rng(991)
    n, G = (1 << 18) + 4321, 40_000
    k = rng.integers(0, G, n).astype(np.int64) * 7 - 12345
    v = rng.random(n) * 10
    s = random_utf8(rng, n, 0.1)
    b = Column.from_numpy(rng.random(n) < 0.5, rng.random(n) >= 0.1)
    cols = [Column.from_numpy(k), Column.from_numpy(v), s, b]
    aggs = {"keys_only": [], "count_utf8": [(AggregateFunc.Count, 2)], "count_bool": [(AggregateFunc.Count, 3)],
            "count_utf8_and_values": [(AggregateFunc.Count, 2)] + ALL_AGGS(1)}[shape]
    kn = col(0).flatten(fields("k", "v", "s", "b"))
    exp = orc.aggregate([cols], aggs, group_nodes=kn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(2):                                  # the second execution runs from what the first remembered
        got, gk = ctx.aggregate(t, aggs, group_nodes=kn, with_keys=True)
        keys = gk.to_host()[0].to_numpy()
        assert len(keys) == len(np.unique(k)) and (np.sort(keys) == np.unique(k)).all(), f"{shape} rep {rep}: keys"
        if aggs:
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{shape} rep {rep}")
        else:
            assert got.num_rows == len(keys) and got.num_columns == 0


def test_aggregate_range_partitions_with_keys_beyond_int32(ctx):
    """key-range partitions kept the key truncated to int32 in the 12-byte tuple; keys around 5e9 whose RANGE is
    small made the second kernel address LDS by garbage (and cost a retry).  The tuple now holds key - range_min."""
    rng = np.random.default_rng(5)
    n, G = 600_000, 70_000
    k = rng.integers(0, G, n).astype(np.int64) + 5_000_000_000
    v = rng.random(n) * 10 - 3
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    kn = col(0).flatten(fields("k", "v"))
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn, with_keys=True)
        ctx.timing_enable(False)
        assert ctx.timing_query("agg_segments_direct")[1] > 0, f"rep {rep}: key-range partitions were expected"
        assert (np.sort(gk.to_host()[0].to_numpy()) == np.unique(k)).all()
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"wide keys, rep {rep}")


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", ["readme", "c1", "five_on_one", "count_key_only", "uint64_key", "key_range_pred", "nan_values", "negative_keys"])
def test_aggregate_at_most_four_groups_in_registers(ctx, m, shape):
    """`group by id % m`, m <= 4 — the reference's own aggregate query (src/main.rs:36-40, README.md:105-111: count(id), sum(age), sum(score),
    avg(score), max(score), min(score) … group by id % 3) — goes through the register-resident kernel (aggregate_tiny.hip) from 2^20 rows on:
    every arrangement of value columns it takes, the key column as a value column, UInt64 keys, a range predicate on the key, NaN values
    (max.rs:38-50), and negative keys (outside [0, m): the streaming kernel takes over, now and on the next execution).
    aggregate/mod.rs:113-222, sum.rs, avg.rs, count.rs, max.rs, min.rs"""
    rng = np.random.default_rng(100 * m + len(shape))
    n = (1 << 20) + 777
    ids = rng.integers(0, 1 << 40, n).astype(np.int64)
    if shape == "negative_keys":
        ids[n // 2] = -5
        ids[7] = -(1 << 33)
    age = rng.integers(18, 78, n).astype(np.int64)
    score = rng.random(n) * 100.0
    if shape == "nan_values":
        score[rng.integers(0, n, 5)] = np.nan
    idc = Column.from_numpy(ids.astype(np.uint64)) if shape == "uint64_key" else Column.from_numpy(ids)
    cols = [idc, Column.from_numpy(age), Column.from_numpy(score)]
    f3 = fields("id", "age", "score")
    aggs = {"readme": [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 1), (AggregateFunc.Sum, 2), (AggregateFunc.Avg, 2), (AggregateFunc.Max, 2), (AggregateFunc.Min, 2)],
            "c1": [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 2)],
            "five_on_one": ALL_AGGS(2), "count_key_only": [(AggregateFunc.Count, 0), (AggregateFunc.Max, 0)]}.get(shape, ALL_AGGS(2) + [(AggregateFunc.Sum, 1)] if shape == "negative_keys" else
                                                                                                                    [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 2), (AggregateFunc.Min, 2)])
    lit = lit_u64 if shape == "uint64_key" else lit_i64
    key = binop(col(0), Operator.Modulos, lit(m)).flatten(f3)
    pred = binop(col(0), Operator.Lt, lit_i64(1 << 39)).flatten(f3) if shape == "key_range_pred" else None
    exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
    t = ctx.table_from_host(cols)
    for rep in range(2):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred, with_keys=True)
        ctx.timing_enable(False)
        tiny = ctx.timing_query("agg_grouped_tiny")[1]
        last_needs_minmax_only = shape != "negative_keys"      # (its aggregate list asks min / max of a column that is not the pass's last: not this kernel's shape)
        if shape == "negative_keys":
            assert tiny == 0 or rep == 0
        elif last_needs_minmax_only:
            assert tiny > 0, f"{shape} m={m} rep {rep}: the register-resident kernel was expected"
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[i for i, (fn, _) in enumerate(aggs) if fn == AggregateFunc.Count], what=f"{shape} m={m} rep {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert (np.sort(kk.astype(np.int64)) == np.unique(np.fmod(ids, m) if shape != "uint64_key" else (ids.astype(np.uint64) % np.uint64(m)).astype(np.int64))[
            np.isin(np.unique(np.fmod(ids, m) if shape != "uint64_key" else (ids.astype(np.uint64) % np.uint64(m)).astype(np.int64)), kk.astype(np.int64))]).all()


def test_register_kernel_is_not_taken_for_columns_on_an_odd_word(ctx):
    """the register kernel reads its tiles in 16-byte loads (round 6): a table whose columns start 8 bytes into a borrowed buffer (a
    slice: row 1 onwards) must take the streaming kernel instead — same result as the oracle's over those rows, and the aligned table
    over the same buffers still takes the register kernel"""
    rng = np.random.default_rng(4321)
    n = (1 << 20) + 1001
    ids = rng.integers(0, 1 << 40, n).astype(np.int64)
    age = rng.integers(18, 78, n).astype(np.int64)
    score = rng.random(n) * 100.0
    cols = [Column.from_numpy(ids), Column.from_numpy(age), Column.from_numpy(score)]
    whole = ctx.table_from_host(cols)
    ptrs = [int(whole.column_info(i).values) for i in range(3)]
    odd = ctx.table_from_device([(DType.INT64, n - 1, ptrs[0] + 8, None), (DType.INT64, n - 1, ptrs[1] + 8, None), (DType.FLOAT64, n - 1, ptrs[2] + 8, None)])
    f3 = fields("id", "age", "score")
    key = binop(col(0), Operator.Modulos, lit_i64(3)).flatten(f3)
    A = AggregateFunc
    aggs = [(A.Count, 0), (A.Sum, 1), (A.Sum, 2), (A.Avg, 2), (A.Max, 2), (A.Min, 2)]
    for t, sl, expect_tiny in ((odd, slice(1, None), False), (whole, slice(None), True)):
        exp = orc.aggregate([[Column.from_numpy(ids[sl]), Column.from_numpy(age[sl]), Column.from_numpy(score[sl])]], aggs, group_nodes=key)[0]
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.aggregate(t, aggs, group_nodes=key)
        ctx.timing_enable(False)
        assert (ctx.timing_query("agg_grouped_tiny")[1] > 0) == expect_tiny
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"README query, register kernel {expect_tiny}")
    del odd


@pytest.mark.parametrize("m", [3, 4])
def test_register_kernel_unpacks_its_packed_counters_mid_run(m, monkeypatch):
    """aggregate_tiny.hip keeps the per-key row counts of a lane in ONE packed word and unpacks it every 4096 tiles — a branch a workgroup
    first reaches beyond ~2 x 10^9 rows.  NQE_TINY_UNPACK_TILES (read once per context, AggSwitches::tiny_unpack_tiles) lowers the
    period to 2 tiles, so that the branch runs several times per lane at 6 x 10^6 rows: counts exact, everything else within 1e-9
    against the oracle, for the README's aggregate list, C1's and five aggregates of one column (count.rs:63-76, aggregate/mod.rs:113-222)"""
    from naive_query_engine_amd import capi

    monkeypatch.setenv("NQE_TINY_UNPACK_TILES", "2")
    c = capi.Context(0)
    try:
        rng = np.random.default_rng(900 + m)
        n = 6_000_000 + 321
        ids = rng.integers(0, 1 << 40, n).astype(np.int64)
        age = rng.integers(18, 78, n).astype(np.int64)
        score = rng.random(n) * 100.0
        cols = [Column.from_numpy(ids), Column.from_numpy(age), Column.from_numpy(score)]
        f3 = fields("id", "age", "score")
        key = binop(col(0), Operator.Modulos, lit_i64(m)).flatten(f3)
        t = c.table_from_host(cols)
        A = AggregateFunc
        for aggs, pred in (([(A.Count, 0), (A.Sum, 1), (A.Sum, 2), (A.Avg, 2), (A.Max, 2), (A.Min, 2)], None), ([(A.Count, 0), (A.Sum, 1), (A.Avg, 2)], None),
                           (ALL_AGGS(2), binop(col(0), Operator.Lt, lit_i64(1 << 39)).flatten(f3))):
            exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
            c.timing_enable(True)
            c.timing_reset()
            got = c.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)
            c.timing_enable(False)
            assert c.timing_query("agg_grouped_tiny")[1] > 0, "the register-resident kernel was expected"
            assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[i for i, (fn, _) in enumerate(aggs) if fn == A.Count], what=f"unpack period 2, m={m}, {len(aggs)} aggregates")
    finally:
        c.close()


@pytest.mark.parametrize("shape", ["range_1_5M_of_2M", "hashed_sparse_1M", "pred_on_other_column", "mod_key_70001"])
def test_aggregate_block_scatter_forms(ctx, shape):
    """the two-stream slabs written in whole 16-tuple blocks (round 5, agg_slab_scatter_soa_kernel): a key range beyond 2^20 values (512
    partitions, blocks of 8 tuples, the range tier with one workgroup per partition), hashed partitions of sparse keys that fit int32,
    a predicate on a column other than the key (PRED 2), a `col % m` key whose range comes from the modulus — against the oracle, first
    and remembered execution.  aggregate/mod.rs:113-222"""
    rng = np.random.default_rng(4242 + len(shape))
    n = 3_000_000 if shape != "mod_key_70001" else 900_000
    pred = None
    key = col(0)
    if shape == "range_1_5M_of_2M":
        k = rng.integers(0, 2_000_000, n).astype(np.int64) + 7_000_000_000       # beyond int32: the tuple holds key - min
    elif shape == "hashed_sparse_1M":
        k = (rng.integers(0, 1_000_000, n).astype(np.int64) * 2_001) - 1_000_000_000   # sparse over 2 x 10^9: no compact range, fits int32
    elif shape == "pred_on_other_column":
        k = rng.integers(0, 300_000, n).astype(np.int64)
        pred = binop(col(1), Operator.Gt, lit_f64(2.5))
    else:
        k = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
        key = binop(col(0), Operator.Modulos, lit_i64(70_001))
    v = rng.random(n) * 10 - 2
    cols = [Column.from_numpy(k), Column.from_numpy(v)]
    f2 = fields("k", "v")
    kn = key.flatten(f2)
    pn = pred.flatten(f2) if pred is not None else None
    exp = orc.aggregate([cols], ALL_AGGS(1), group_nodes=kn, pred_nodes=pn)[0]
    t = ctx.table_from_host(cols)
    for rep in range(3):
        ctx.timing_enable(True)
        ctx.timing_reset()
        got, gk = ctx.aggregate(t, ALL_AGGS(1), group_nodes=kn, pred_nodes=pn, with_keys=True)
        ctx.timing_enable(False)
        assert ctx.timing_query("agg_partition_scatter")[1] > 0, f"{shape} rep {rep}: the partitioned path was expected"
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{shape} rep {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert (np.diff(kk) > 0).all(), f"{shape}: keys not in order"


@pytest.mark.parametrize("tiles", [1, 4095, 4096, 4097, 8193, 24575, 24576, 24577, 30001])
def test_selection_tile_offsets_across_the_scan_forms(ctx, tiles):
    """The tile counts of a selection are scanned by ONE workgroup up to 6 x 4096 tiles (every thread 24 consecutive counts, one wave scan:
    round 6) and by the chunked recursive scan beyond: row counts around every boundary, ragged last tiles, kept rows checked against numpy
    (count, every kept value through a checksum, first and last).  selection.rs:58-107"""
    rng = np.random.default_rng(tiles)
    n = tiles * 4096 - int(rng.integers(0, 4095)) if tiles > 1 else 1234
    ids = rng.integers(0, 1000, n).astype(np.int64)
    v = np.arange(n, dtype=np.int64)
    t = ctx.table_from_host([Column.from_numpy(ids), Column.from_numpy(v)])
    f2 = fields("id", "v")
    for lim in (0, 137, 1000):
        pred = binop(col(0), Operator.Lt, lit_i64(lim)).flatten(f2)
        got = ctx.selection(t, pred)
        keep = ids < lim
        assert got.num_rows == int(keep.sum()), (tiles, lim)
        out = got.to_host()
        gv = out[1].to_numpy()
        assert (gv == v[keep]).all(), (tiles, lim)
        assert (out[0].to_numpy() == ids[keep]).all(), (tiles, lim)


@pytest.mark.parametrize("seed", range(int(os.environ.get("NQE_SWEEP_SEED_BASE", "0")), int(os.environ.get("NQE_SWEEP_SEED_BASE", "0")) + int(os.environ.get("NQE_SWEEP_SEEDS", "16"))))
def test_aggregate_direct_tables_random_sweep(ctx, seed):
    """Random spans around every limit of the directly addressed tables (one table 4096 / 5840 / 13632 keys, two subsets 2 x 5840 / 2 x 13632,
    the range tier beyond), random bases (negative, beyond 2^40, UInt64 above 2^63), with and without min / max, with and without a
    predicate, Int64 and Float64 values: three executions each (sampled plan, remembered plan, again) against the oracle, keys in order."""
    rng = np.random.default_rng(9000 + seed)
    limits = [4096, 5840, 8192, 11680, 13632, 16384, 27264]
    span = int(rng.choice(limits)) + int(rng.integers(-3, 4)) if seed % 2 == 0 else int(rng.integers(4097, 30000))
    span = max(span, 2)
    n = int(rng.integers(600_000, 1_500_000))
    unsigned = bool(rng.integers(0, 4) == 0)
    base = int(rng.choice([0, -span // 2, -10**9, 1 << 41, 7])) if not unsigned else int(rng.choice([0, (1 << 63) + 12345, 1 << 50]))
    draw = rng.integers(0, span, n)
    draw[:2] = [0, span - 1]
    if rng.integers(0, 3) == 0:  # every other value only: a range twice as wide as its keys
        draw = (draw // 2) * 2
        draw[:2] = [0, (span - 1) // 2 * 2]
    k = (draw.astype(np.uint64) + np.uint64(base)) if unsigned else (draw + base).astype(np.int64)
    int_values = bool(rng.integers(0, 2))
    v = rng.integers(-10**6, 10**6, n).astype(np.int64) if int_values else rng.random(n) * 200.0 - 100.0
    w = rng.random(n)
    nomm = bool(rng.integers(0, 2))
    with_pred = bool(rng.integers(0, 3) == 0)
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1)] if nomm else ALL_AGGS(1)
    f3 = fields("k", "v", "w")
    cols = [Column.from_numpy(k), Column.from_numpy(v), Column.from_numpy(w)]
    key = col(0).flatten(f3)
    pred = binop(col(2), Operator.Lt, lit_f64(0.6)).flatten(f3) if with_pred else None
    exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
    t = ctx.table_from_host(cols)
    what = f"seed {seed}: span {span} base {base} unsigned {unsigned} nomm {nomm} pred {with_pred} int values {int_values}"
    for rep in range(3):
        got, gk = ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred, with_keys=True)
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0], what=f"{what}, run {rep}")
        kk = gk.to_host()[0].to_numpy()
        assert len(np.unique(kk)) == len(kk) == got.num_rows, what


@pytest.mark.parametrize("seed", range(int(os.environ.get("NQE_SWEEP_SEED_BASE", "0")), int(os.environ.get("NQE_SWEEP_SEED_BASE", "0")) + int(os.environ.get("NQE_SWEEP_SEEDS", "12"))))
def test_join_two_level_build_random_sweep(ctx, monkeypatch, seed):
    """The two-level partitioned build over random sizes and key ranges (one partition ... hundreds, 1 ... 64 fine bins per partition, ranges
    that end inside a bin), key density 1 / 2 / 7 (holes), no payload / a bit-packed / a 32-bit / an 8-byte payload column, a few duplicate
    keys in some (-> the sort-based build): the join's rows, in order, against the oracle; the table reused for a second probe.
    hash_join.rs:124-254"""
    monkeypatch.setenv("NQE_JOIN_PART_BUILD_MIN", "1000")
    rng = np.random.default_rng(31000 + seed)
    nb = int(rng.choice([5_000, 70_000, 300_000, 1_100_000, 3_000_000])) + int(rng.integers(0, 999))
    density = int(rng.choice([1, 1, 2, 7]))
    base = int(rng.choice([0, 12345, -(10**9), 1 << 40]))
    dk = (rng.permutation(density * nb)[:nb]).astype(np.int64) + base
    dup = rng.integers(0, 5) == 0
    if dup:
        dk[rng.integers(0, nb, 5)] = dk[rng.integers(0, nb, 5)]
    left = [Column.from_numpy(dk)]
    payload = int(rng.integers(0, 4))
    if payload == 1:
        left.append(Column.from_numpy(rng.integers(0, 1 << int(rng.integers(2, 25)), nb).astype(np.int64) - 5))
    elif payload == 2:
        left.append(Column.from_numpy(rng.integers(0, 1 << 31, nb).astype(np.int64)))
    elif payload == 3:
        left.append(Column.from_numpy(rng.random(nb)))
    n = int(rng.integers(50_000, 300_000))
    rk = rng.integers(int(dk.min()) - 5, int(dk.max()) + 6, n).astype(np.int64)
    rk[: min(n, 1000)] = dk[rng.integers(0, nb, min(n, 1000))]
    right = [Column.from_numpy(rk), Column.from_numpy(rng.random(n))]
    exp = orc.hash_join([left], [right], 0, 0)[0]
    lt, rt = ctx.table_from_host(left), ctx.table_from_host(right)
    what = f"seed {seed}: {nb} build rows, density {density}, base {base}, payload {payload}, dup {bool(dup)}"
    assert_batches_equal(ctx.hash_join(lt, rt, 0, 0).to_host(), exp, what=what)
    jt = ctx.hash_join_build(lt, 0)
    rk2 = rk[::-1].copy()
    right2 = [Column.from_numpy(rk2), Column.from_numpy(rng.random(n))]
    exp2 = orc.hash_join([left], [right2], 0, 0)[0]
    assert_batches_equal(ctx.hash_join_probe(jt, ctx.table_from_host(right2), 0).to_host(), exp2, what=what + ", reused table")
