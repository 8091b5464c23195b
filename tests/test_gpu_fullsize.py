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


# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json's configs at FULL size against an independent CPU result (task 1): every group of the 10^9-row aggregates,
# every output row of C2 and C4.  The device columns themselves are downloaded (the CPU sees the very inputs the GPU saw); the
# aggregates are computed by oracle/nqe_oracle.cpp: orc_grouped_parallel — checked against the reference-faithful single-threaded port
# in tests/test_oracle_golden.py::test_parallel_grouped_form_matches_the_port — C2 and C4 by the port itself.
import os

from oracle import oracle as orc

CPU_THREADS = max(1, min(64, os.cpu_count() or 1))


def download_words(ctx, dtype, ptr, lo, hi):
    """rows [lo, hi) of a device column of 8-byte words → numpy"""
    t = ctx.table_from_device([(dtype, hi - lo, ptr + lo * 8, None)])
    return t.to_host()[0].to_numpy()


def cpu_grouped_full(ctx, ids_ptr, v_ptr, v_dtype, n, limit, modulus, chunk=1 << 27):
    parts = []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        ids = download_words(ctx, DType.INT64, ids_ptr, lo, hi)
        v = ids if v_ptr == ids_ptr else download_words(ctx, v_dtype, v_ptr, lo, hi)
        parts.append(orc.grouped_parallel(ids, v, limit, modulus, CPU_THREADS))
        del ids, v
    return orc.merge_grouped(parts)


def assert_groups_equal(out, keys, exp, what):
    """GPU result [count, sum, avg, min, max] + keys against float64[modulus, 4] = count, sum, min, max per key: every group"""
    cnt, s, avg, mn, mx = host(out)
    k = host(keys)[0]
    live = np.nonzero(exp[:, 0] > 0)[0]
    assert len(k) == len(live) and (k == live).all(), f"{what}: group keys differ"
    e = exp[live]
    assert (cnt.astype(np.float64) == e[:, 0]).all(), f"{what}: counts differ"
    assert np.allclose(s, e[:, 1], rtol=1e-9, atol=0), f"{what}: sums differ beyond 1e-9"
    assert np.allclose(avg, e[:, 1] / e[:, 0], rtol=1e-9, atol=0), f"{what}: averages differ beyond 1e-9"
    assert (mn == e[:, 2]).all() and (mx == e[:, 3]).all(), f"{what}: min / max differ"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("random_ids", [False, True])
def test_headline_and_c3_1e9_rows_every_group_against_the_cpu(ctx, random_ids):
    """headline (`where id < N/2`) and C3 (no filter) at 10^9 rows, ids = row numbers or random: all 1024 groups — counts exact, min / max
    exact, sum / avg within 1e-9 (aggregate/mod.rs:54-102, sum.rs, avg.rs, max.rs, min.rs)"""
    n = 10**9
    d = DevCols(ctx)
    try:
        ids = d.synth(1, 1, n, 0, n, 0) if random_ids else d.synth(0, 0, n)
        v = d.synth(2, 3, n)
        t = table(ctx, (DType.INT64, n, ids), (DType.FLOAT64, n, v))
        f = fields("id", "v")
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        for limit in (n // 2, None):
            exp = cpu_grouped_full(ctx, ids, v, DType.FLOAT64, n, limit, 1024)
            out, keys = ctx.aggregate(t, ALL(1), group_nodes=key, pred_nodes=pred if limit is not None else None, with_keys=True)
            assert int(exp[:, 0].sum()) == (n if limit is None else (n // 2 if not random_ids else int(exp[:, 0].sum())))
            assert_groups_equal(out, keys, exp, f"{'headline' if limit is not None else 'C3'}, {'random' if random_ids else 'sorted'} ids")
    finally:
        d.free()


@pytest.mark.timeout(600)
def test_headline_1e9_int64_values_and_single_column_against_the_cpu(ctx):
    """north_star's literal "10^9 Int64 rows": Int64 values accumulated `as f64`, and key = predicate = value = ONE column"""
    n = 10**9
    d = DevCols(ctx)
    try:
        ids, age = d.synth(0, 0, n), d.synth(1, 2, n, 0, 60, 18)
        f = fields("id", "age")
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        t = table(ctx, (DType.INT64, n, ids), (DType.INT64, n, age))
        out, keys = ctx.aggregate(t, ALL(1), group_nodes=key, pred_nodes=pred, with_keys=True)
        assert_groups_equal(out, keys, cpu_grouped_full(ctx, ids, age, DType.INT64, n, n // 2, 1024), "Int64 values")
        out, keys = ctx.aggregate(t, ALL(0), group_nodes=key, pred_nodes=pred, with_keys=True)
        assert_groups_equal(out, keys, cpu_grouped_full(ctx, ids, ids, DType.INT64, n, n // 2, 1024), "single column")
    finally:
        d.free()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("groups", [65536, 1 << 20])
def test_many_group_aggregate_1e8_rows_every_group_against_the_cpu(ctx, groups):
    """the partitioned path at bench size: 10^8 rows over 65536 / 2^20 random keys, every group"""
    n = 10**8
    d = DevCols(ctx)
    try:
        k, v = d.synth(1, 7, n, 0, groups, 0), d.synth(2, 3, n)
        t = table(ctx, (DType.INT64, n, k), (DType.FLOAT64, n, v))
        exp = cpu_grouped_full(ctx, k, v, DType.FLOAT64, n, None, groups)
        for rep in range(2):       # first execution (sampled plan) and the remembered one
            out, keys = ctx.aggregate(t, ALL(1), group_nodes=col(0).flatten(fields("k", "v")), with_keys=True)
            assert_groups_equal(out, keys, exp, f"{groups} groups, execution {rep}")
    finally:
        d.free()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("random_ids", [False, True])
def test_c2_1e8_rows_every_output_row_against_the_port(ctx, random_ids):
    """C2 at 10^8 rows: the whole 5x10^7-row output of `select age + 100 from t where id < N/2`, bit for bit and in order, against the
    oracle's SelectionPlan + ProjectionPlan (selection.rs:34-107, projection.rs:43-70)"""
    from naive_query_engine_amd import Column

    n = 10**8
    d = DevCols(ctx)
    try:
        ids = d.synth(1, 1, n, 0, n, 0) if random_ids else d.synth(0, 0, n)
        age = d.synth(1, 2, n, 0, 60, 18)
        t = table(ctx, (DType.INT64, n, ids), (DType.INT64, n, age))
        f = fields("id", "age")
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f), col(0).flatten(f)]
        h = orc.upload([[Column.from_numpy(download_words(ctx, DType.INT64, ids, 0, n)), Column.from_numpy(download_words(ctx, DType.INT64, age, 0, n))]])
        ref = orc.projection(orc.selection(h, pred, raw=True), proj)[0]
        got = ctx.selection_projection(t, pred, proj).to_host()
        assert len(got) == len(ref) == 2
        for g, r in zip(got, ref):
            assert g.length == r.length and (g.to_numpy() == r.to_numpy()).all()
        assert (not random_ids and ref[0].length == n // 2) or (random_ids and abs(ref[0].length - n // 2) < 10**5)
    finally:
        d.free()


@pytest.mark.timeout(900)
def test_c4_1e8_x_1e6_every_output_row_against_the_port(ctx):
    """C4 at full size: inner hash join of the 10^8-row fact table with the 10^6-row dimension — all four output columns of all 10^8
    rows, bit for bit and in the reference's order (probe-major, hash_join.rs:80-103, 168-254), against the oracle's HashJoin
    (single thread, ~25 s)"""
    from naive_query_engine_amd import Column

    n, nb = 10**8, 10**6
    d = DevCols(ctx)
    try:
        rng = np.random.default_rng(7)
        perm = rng.permutation(nb).astype(np.int64)
        attr = rng.integers(0, 1 << 20, nb).astype(np.int64)
        dim_cols = [Column.from_numpy(perm), Column.from_numpy(attr)]
        dim = ctx.table_from_host(dim_cols)
        fkey, val = d.synth(1, 5, n, 0, nb, 0), d.synth(2, 3, n)
        fact = table(ctx, (DType.INT64, n, fkey), (DType.FLOAT64, n, val))
        right = [Column.from_numpy(download_words(ctx, DType.INT64, fkey, 0, n)), Column.from_numpy(download_words(ctx, DType.FLOAT64, val, 0, n))]
        ref = orc.hash_join([dim_cols], [right], 0, 0)[0]
        got = ctx.hash_join(dim, fact, 0, 0).to_host()
        assert len(got) == len(ref) == 4
        for j, (g, r) in enumerate(zip(got, ref)):
            assert g.length == r.length == n, f"column {j}: {g.length} rows, expected {r.length}"
            assert (g.to_numpy().view(np.int64) == r.to_numpy().view(np.int64)).all(), f"column {j} differs"
    finally:
        d.free()


# ------------------------------------------------------------------------------------------------------------------------------------
# Round 6: the same for the reference's OWN aggregate query (src/main.rs:36-40, README.md:105-111: three different value columns, one
# with max / min — served by the register-resident kernel of aggregate_tiny.hip) and for the headline over a value column with 1 % NULLs
# (SURVEY §8d's correctness run), through the k-column / validity form of the parallel CPU aggregate (oracle.grouped_columns_parallel,
# pinned against the port in tests/test_oracle_golden.py::test_parallel_grouped_columns_form_matches_the_port).
def cpu_grouped_columns_full(ctx, ids_ptr, cols, n, limit, modulus, valid_bits=None, chunk=1 << 27):
    """cols = {column index: (dtype, device pointer)}; valid_bits = {column index: uint8[] LSB-first validity bitmap on the host}"""
    parts = []
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        ids = download_words(ctx, DType.INT64, ids_ptr, lo, hi)
        host_cols = {c: (ids if p == ids_ptr else download_words(ctx, dt, p, lo, hi)) for c, (dt, p) in cols.items()}
        masks = {c: np.unpackbits(b[lo // 8:(hi + 7) // 8], bitorder="little")[:hi - lo] for c, b in (valid_bits or {}).items()}
        parts.append(orc.grouped_columns_parallel(ids, host_cols, limit, modulus, CPU_THREADS, valid=masks))
        del ids, host_cols, masks
    return orc.merge_grouped_columns(parts)


def assert_agg_columns_equal(out, keys, state, aggs, what):
    live, exp = orc.finalize_grouped(state, aggs)
    k = host(keys)[0]
    assert len(k) == len(live) and (k == live).all(), f"{what}: group keys differ"
    for (func, c), g, e in zip(aggs, host(out), exp):
        if func in (AggregateFunc.Count, AggregateFunc.Min, AggregateFunc.Max):
            assert (g.astype(np.float64) == e).all(), f"{what}: {func.name}({c}) differs"
        else:
            assert np.allclose(g, e, rtol=1e-9, atol=0, equal_nan=True), f"{what}: {func.name}({c}) differs beyond 1e-9"


@pytest.mark.timeout(900)
def test_readme_aggregate_query_1e9_rows_every_group_against_the_cpu(ctx):
    """`select count(id), sum(age), sum(score), avg(score), max(score), min(score) from t group by id % 3` (the reference's own query,
    src/main.rs:36-40) at 10^9 rows, and C1's `count(id), sum(age), avg(score)`: every output value against the CPU"""
    n = 10**9
    d = DevCols(ctx)
    try:
        ids, age, score = d.synth(0, 0, n), d.synth(1, 2, n, 0, 60, 18), d.synth(2, 3, n)
        t = table(ctx, (DType.INT64, n, ids), (DType.INT64, n, age), (DType.FLOAT64, n, score))
        f = fields("id", "age", "score")
        key = binop(col(0), Operator.Modulos, lit_i64(3)).flatten(f)
        state = cpu_grouped_columns_full(ctx, ids, {0: (DType.INT64, ids), 1: (DType.INT64, age), 2: (DType.FLOAT64, score)}, n, None, 3)
        A = AggregateFunc
        for aggs in ([(A.Count, 0), (A.Sum, 1), (A.Sum, 2), (A.Avg, 2), (A.Max, 2), (A.Min, 2)], [(A.Count, 0), (A.Sum, 1), (A.Avg, 2)]):
            out, keys = ctx.aggregate(t, aggs, group_nodes=key, with_keys=True)
            assert_agg_columns_equal(out, keys, state, aggs, f"README query, {len(aggs)} aggregates")
    finally:
        d.free()


@pytest.mark.timeout(900)
def test_headline_1e9_rows_with_null_values_every_group_against_the_cpu(ctx):
    """the headline over `v Float64` with 1 % NULLs (u(i,4) mod 100 == 0) at 10^9 rows: count skips NULLs, sum / min / max
    ignore them (count.rs:63, sum.rs:86-101, max.rs:38); the bitmap the GPU reads is the one the CPU unpacks"""
    n = 10**9
    d = DevCols(ctx)
    try:
        ids, v, u = d.synth(0, 0, n), d.synth(2, 3, n), d.synth(1, 4, n, 0, 100, 0)
        fu = fields("u")
        bm = ctx.expr_evaluate(table(ctx, (DType.INT64, n, u)), binop(col(0), Operator.NotEq, lit_i64(0)).flatten(fu))   # Boolean column: its values ARE an LSB-first bitmap
        info = bm.column_info(0)
        assert DType(info.dtype) == DType.BOOLEAN and info.length == n and not info.validity
        bits = bm.download_column(0).values
        assert int(np.unpackbits(bits[:1 << 20], bitorder="little").sum()) < (1 << 23)      # ~1 % zeros
        t = ctx.table_from_device([(DType.INT64, n, ids, None), (DType.FLOAT64, n, v, int(info.values))])
        f = fields("id", "v")
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
        pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
        state = cpu_grouped_columns_full(ctx, ids, {1: (DType.FLOAT64, v)}, n, n // 2, 1024, valid_bits={1: bits})
        nulls = n // 2 - int(state[1][:, 0].sum())
        assert 0.009 * (n // 2) < nulls < 0.011 * (n // 2) and int(state[1][:, 4].sum()) == n // 2
        out, keys = ctx.aggregate(t, ALL(1), group_nodes=key, pred_nodes=pred, with_keys=True)
        assert_agg_columns_equal(out, keys, state, ALL(1), "headline with 1 % NULLs")
        del t, bm
    finally:
        d.free()


@pytest.mark.timeout(900)
def test_join_build_2p25_rows_over_512_partitions(ctx):
    """the two-level partitioned build at its real threshold (2^25 build rows) and over a key range of 1.0 x 10^8 values with holes — 384
    partitions, 12 288 fine bins: keys 3 i + (0 | 1 | 2) shuffled, an Int64 payload 7 key + 1 (beyond 25 bits: the 32-bit column form)
    and a Float64 payload key / 8; probed with 2 x 10^7 keys of which two thirds exist: row count, every matched payload analytic, order
    = probe order (hash_join.rs:124-254).  The one-level form (NQE_JOIN_PART_ONE_LEVEL=1) must give the same table."""
    from naive_query_engine_amd import Column

    nb, npr = 1 << 25, 20_000_000
    rng = np.random.default_rng(25)
    keys = (np.arange(nb, dtype=np.int64) * 3 + rng.integers(0, 3, nb)) + 1000
    keys = keys[rng.permutation(nb)]
    present = np.zeros(3 * nb + 1000 + 8, dtype=bool)
    present[keys] = True
    pk = rng.integers(1000, 3 * nb + 1000, npr).astype(np.int64)
    exp_keys = pk[present[pk]]
    right = ctx.table_from_host([Column.from_numpy(pk)])
    for payload in ("int", "f64"):
        pay = keys * 7 + 1 if payload == "int" else keys.astype(np.float64) / 8.0
        left = ctx.table_from_host([Column.from_numpy(keys), Column.from_numpy(pay)])
        outs = []
        for one_level in (False, True):
            if one_level:
                os.environ["NQE_JOIN_PART_ONE_LEVEL"] = "1"
            try:
                ctx.timing_enable(True)
                ctx.timing_reset()
                got = ctx.hash_join(left, right, 0, 0)
                names = ctx.timing_report()
                ctx.timing_enable(False)
            finally:
                os.environ.pop("NQE_JOIN_PART_ONE_LEVEL", None)
            assert ("join_build_part_fill" in names) == (not one_level) and ("join_build_part_place" in names) == one_level, names
            k, p, rk = host(got)
            assert len(k) == len(exp_keys) and (k == exp_keys).all() and (rk == exp_keys).all(), f"{payload}: keys / order"
            assert (p == (exp_keys * 7 + 1 if payload == "int" else exp_keys.astype(np.float64) / 8.0)).all(), f"{payload}: payload"
            outs.append(len(k))
        assert outs[0] == outs[1]
