"""BASELINE.json's full sizes on one MI355X, checked through size-independent properties (the oracle cannot run
10^9 rows in seconds): analytic counts, additivity over row ranges, cross-kernel checksums, idempotence, stable order,
and an analytic key→payload relation for the join.  Data is generated on the device (SURVEY §8d generators)."""
import numpy as np
import pytest

from naive_query_engine_amd import AggregateFunc, DType, Operator
from naive_query_engine_amd.expression import binop, col, lit_f64, lit_i64
from tests.helpers import fields

pytestmark = pytest.mark.gpu
ALL = lambda c: [(AggregateFunc.Count, c), (AggregateFunc.Sum, c), (AggregateFunc.Avg, c), (AggregateFunc.Min, c), (AggregateFunc.Max, c)]


@pytest.fixture(scope="module")
def ctx():
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


class DevCols:
    """device buffers owned by the test (freed at the end)"""

    def __init__(self, ctx):
        self.ctx, self.ptrs = ctx, []

    def synth(self, kind, seed, n, first=0, mod=1, base=0):
        p = self.ctx.device_alloc(n * 8)
        self.ptrs.append(p)
        self.ctx.synth_fill(kind, seed, first, n, mod, base, p)
        return p

    def free(self):
        for p in self.ptrs:
            self.ctx.device_free(p)
        self.ptrs = []


def table(ctx, *cols):
    return ctx.table_from_device([(dt, n, p, None) for dt, n, p in cols])


def host(tab):
    return [c.to_numpy() for c in tab.to_host()]


@pytest.mark.timeout(600)
def test_headline_1e9_rows_properties(ctx):
    n = 10**9
    d = DevCols(ctx)
    try:
        ids, v = d.synth(0, 0, n), d.synth(2, 3, n)
        t = table(ctx, (DType.INT64, n, ids), (DType.FLOAT64, n, v))
        f = fields("id", "v")
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        out, keys = ctx.aggregate(t, ALL(1), group_nodes=key, pred_nodes=pred, with_keys=True)
        cnt, s, avg, mn, mx = host(out)
        k = host(keys)[0]
        assert (k == np.arange(1024)).all()
        # analytic counts: rows i < n/2 with i % 1024 == g
        exp_cnt = (n // 2 - np.arange(1024) + 1023) // 1024
        assert (cnt.astype(np.int64) == exp_cnt).all() and int(cnt.sum()) == n // 2
        assert (mn >= 0).all() and (mx < 100).all() and (mn <= avg).all() and (avg <= mx).all()
        assert np.allclose(avg, s / cnt, rtol=1e-15)
        assert np.allclose(s / cnt, 50.0, rtol=1e-2)                     # v uniform in [0, 100): 12 sigma of a 488k-sample mean
        # cross-kernel checksum: the grouped sums add up to the un-grouped aggregate of the filtered rows
        u = host(ctx.aggregate(t, ALL(1), pred_nodes=pred))
        assert int(u[0][0]) == n // 2 and abs(s.sum() - u[1][0]) <= 1e-9 * u[1][0]
        assert mn.min() == u[3][0] and mx.max() == u[4][0]
        # additivity over row ranges (what the multi-GPU path relies on): partials of the two halves merge to the same result
        h = n // 2
        t0 = table(ctx, (DType.INT64, h, ids), (DType.FLOAT64, h, v))
        t1 = table(ctx, (DType.INT64, n - h, ids + h * 8), (DType.FLOAT64, n - h, v + h * 8))
        p0, p1 = ctx.aggregate_partial(t0, ALL(1), group_nodes=key, pred_nodes=pred), ctx.aggregate_partial(t1, ALL(1), group_nodes=key, pred_nodes=pred)
        assert p1[0].num_rows == 0                                       # every row of the upper half fails `id < n/2`
        merged, _ = ctx.aggregate_merge([p0[0], p1[0]], [p0[1], p1[1]], ALL(1))
        m = host(merged)
        assert (m[0] == cnt).all() and np.allclose(m[1], s, rtol=1e-9) and (m[3] == mn).all() and (m[4] == mx).all()
        # idempotence: a second run gives the same counts/extremes and sums within the summation-order tolerance
        again = host(ctx.aggregate(t, ALL(1), group_nodes=key, pred_nodes=pred))
        assert (again[0] == cnt).all() and (again[3] == mn).all() and (again[4] == mx).all() and np.allclose(again[1], s, rtol=1e-12)
    finally:
        d.free()


@pytest.mark.timeout(600)
def test_group_by_1e9_rows_random_keys_matches_sorted_keys_total(ctx):
    """C3 with random ids: per-group results differ from the sorted case but the totals are the same multiset of v"""
    n = 10**9
    d = DevCols(ctx)
    try:
        rid, v = d.synth(1, 1, n, 0, n, 0), d.synth(2, 3, n)
        t = table(ctx, (DType.INT64, n, rid), (DType.FLOAT64, n, v))
        f = fields("id", "v")
        out = host(ctx.aggregate(t, ALL(1), group_nodes=binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)))
        u = host(ctx.aggregate(t, ALL(1)))
        assert int(out[0].sum()) == n == int(u[0][0])
        assert abs(out[1].sum() - u[1][0]) <= 1e-9 * u[1][0] and out[3].min() == u[3][0] and out[4].max() == u[4][0]
        assert out[0].min() > 0.9 * n / 1024 and out[0].max() < 1.1 * n / 1024
    finally:
        d.free()


@pytest.mark.timeout(600)
def test_filter_project_1e8_rows_properties(ctx):
    n = 10**8
    d = DevCols(ctx)
    try:
        ids, age = d.synth(0, 0, n), d.synth(1, 2, n, 0, 60, 18)
        t = table(ctx, (DType.INT64, n, ids), (DType.INT64, n, age))
        f = fields("id", "age")
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        out = ctx.selection_projection(t, pred, [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f), col(0).flatten(f)])
        assert out.num_rows == n // 2
        # stable order: the id column of the output is 0..n/2-1 (sortedness) and row j carries age[j] + 100
        f2 = fields("agep", "id")
        chk = host(ctx.aggregate(out, [(AggregateFunc.Min, 1), (AggregateFunc.Max, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Sum, 0)]))
        assert chk[0][0] == 0 and chk[1][0] == n // 2 - 1 and chk[2][0] == float((n // 2) * (n // 2 - 1) // 2)
        ref = ctx.slice(ctx.projection(t, [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f)]), 0, n // 2)
        both = ctx.table_from_device([(DType.INT64, n // 2, out.column_info(0).values, None), (DType.INT64, n // 2, ref.column_info(0).values, None)])
        diff = ctx.selection(both, binop(col(0), Operator.NotEq, col(1)).flatten(fields("a", "b")))
        assert diff.num_rows == 0
        ref_sum = host(ctx.aggregate(ref, [(AggregateFunc.Sum, 0)]))[0][0]
        assert chk[3][0] == ref_sum
        # un-fused chain gives the same batch
        sel = ctx.selection(t, pred)
        assert sel.num_rows == n // 2
        un = ctx.projection(sel, [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f)])
        both2 = ctx.table_from_device([(DType.INT64, n // 2, out.column_info(0).values, None), (DType.INT64, n // 2, un.column_info(0).values, None)])
        assert ctx.selection(both2, binop(col(0), Operator.NotEq, col(1)).flatten(fields("a", "b"))).num_rows == 0
    finally:
        d.free()


@pytest.mark.timeout(600)
def test_hash_join_1e8_x_1e6_properties(ctx):
    n, nb = 10**8, 10**6
    d = DevCols(ctx)
    try:
        # dim(id = a permutation of 0..nb-1, attr = id * 7 + 3); the 10^6-row build side comes from the host
        from naive_query_engine_amd import Column

        perm = np.random.default_rng(7).permutation(nb).astype(np.int64)
        dim = ctx.table_from_host([Column.from_numpy(perm), Column.from_numpy(perm * 7 + 3)])
        fkey, val = d.synth(1, 5, n, 0, nb, 0), d.synth(2, 3, n)
        fact = table(ctx, (DType.INT64, n, fkey), (DType.FLOAT64, n, val))
        out = ctx.hash_join(dim, fact, 0, 0)
        assert out.num_rows == n and out.num_columns == 4           # every probe key exists exactly once in dim
        fo = fields("id", "attr", "key", "val")
        # left key == right key, payload follows the analytic relation, probe columns are passed through in order
        assert ctx.selection(out, binop(col(0), Operator.NotEq, col(2)).flatten(fo)).num_rows == 0
        rel = binop(binop(binop(col(2), Operator.Multiply, lit_i64(7)), Operator.Plus, lit_i64(3)), Operator.NotEq, col(1))
        assert ctx.selection(out, rel.flatten(fo)).num_rows == 0
        # the probe-side output columns against FRESHLY generated copies of the fact columns, after the caller has overwritten its own
        # (borrowed) buffers: the output owns its memory and must not be a view of what the caller passed in
        assert out.column_info(2).values != fkey and out.column_info(3).values != val
        ctx.synth_fill(1, 77, 0, n, 1 << 40, 0, fkey)
        ctx.synth_fill(1, 78, 0, n, 1 << 40, 0, val)
        fkey2, val2 = d.synth(1, 5, n, 0, nb, 0), d.synth(2, 3, n)
        both = ctx.table_from_device([(DType.INT64, n, out.column_info(2).values, None), (DType.INT64, n, fkey2, None),
                                      (DType.FLOAT64, n, out.column_info(3).values, None), (DType.FLOAT64, n, val2, None)])
        fkey, val = fkey2, val2
        fact = table(ctx, (DType.INT64, n, fkey), (DType.FLOAT64, n, val))
        fb = fields("a", "b", "c", "d")
        assert ctx.selection(both, binop(col(0), Operator.NotEq, col(1)).flatten(fb)).num_rows == 0
        assert ctx.selection(both, binop(col(2), Operator.NotEq, col(3)).flatten(fb)).num_rows == 0
        # a build side with holes: only even ids present → exactly the probe rows with an even key survive, in order
        even = ctx.selection(dim, binop(binop(col(0), Operator.Modulos, lit_i64(2)), Operator.Eq, lit_i64(0)).flatten(fields("id", "attr")))
        out2 = ctx.hash_join(even, fact, 0, 0)
        exp_rows = ctx.selection(fact, binop(binop(col(0), Operator.Modulos, lit_i64(2)), Operator.Eq, lit_i64(0)).flatten(fields("key", "val")))
        assert out2.num_rows == exp_rows.num_rows
        both2 = ctx.table_from_device([(DType.INT64, out2.num_rows, out2.column_info(2).values, None), (DType.INT64, out2.num_rows, exp_rows.column_info(0).values, None)])
        assert ctx.selection(both2, binop(col(0), Operator.NotEq, col(1)).flatten(fields("a", "b"))).num_rows == 0
    finally:
        d.free()
